"""Multi-process path on CPU (gloo, world_size 2): sharding + call-count all-reduce.

The GPU bench uses the same functions with the nccl (= RCCL) backend; here the per-rank "calls" are
produced by the CPU oracle so that no GPU is needed -- what is under test is warpdemux_amd.dist.
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, out_q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch

    from oracle import wdx_oracle as orc
    from warpdemux_amd import dist, synth

    r, lr, w = dist.init_process_group("gloo")
    assert (r, w) == (rank, world)
    lo, hi = dist.shard_range(n_total, rank, world)
    spec = synth.SynthSpec(n_barcodes=4)
    K = 25
    refs = np.random.default_rng(0).normal(size=(4, K))
    sig, off, a_s, a_e, bc = synth.generate_packed(spec, lo, hi - lo)
    fpt, dwell, stats, status = orc.fingerprint_packed(sig, off, a_s, a_e, orc.SegParams(barcode_num_events=K))
    ok = status == 0
    call = np.full(hi - lo, 4, dtype=np.int64)
    call[ok] = orc.argmin_rows(orc.dtw_matrix(fpt[ok], refs, 15, 0.1))
    counts = torch.from_numpy(np.bincount(call, minlength=5).astype(np.int64))
    local = counts.clone()
    dist.barrier()
    red = dist.CountReducer(None)          # no GPU context here: the process-group road of the shim
    assert red.mode == "torch"
    red(counts)
    assert dist.min_over_ranks(float(rank + 1)) == 1.0
    t = dist.max_over_ranks(float(rank + 1))
    out_q.put((rank, lo, hi, local.numpy(), counts.numpy(), t))
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_count_allreduce():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    n_total = 37   # odd on purpose: ragged shards
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, lo0, hi0, l0, g0, t0), (r1, lo1, hi1, l1, g1, t1) = res
    assert (lo0, hi0, lo1, hi1) == (0, 19, 19, 37)
    assert np.array_equal(g0, g1) and np.array_equal(g0, l0 + l1)
    assert g0.sum() == n_total
    assert t0 == t1 == 2.0

    # the same reads in ONE process give the same histogram: sharding is by global read index
    sys.path.insert(0, ROOT)
    from oracle import wdx_oracle as orc
    from warpdemux_amd import synth

    spec = synth.SynthSpec(n_barcodes=4)
    refs = np.random.default_rng(0).normal(size=(4, 25))
    sig, off, a_s, a_e, bc = synth.generate_packed(spec, 0, n_total)
    fpt, dwell, stats, status = orc.fingerprint_packed(sig, off, a_s, a_e, orc.SegParams(barcode_num_events=25))
    ok = status == 0
    call = np.full(n_total, 4, dtype=np.int64)
    call[ok] = orc.argmin_rows(orc.dtw_matrix(fpt[ok], refs, 15, 0.1))
    assert np.array_equal(np.bincount(call, minlength=5), g0)


def _fallback_worker(rank, world, port, scenario, out_q):
    """prefer="rccl" on a gloo group with the RCCL step failing in a controlled way on ONE rank: the
    collectives of all ranks must still match (no hang) and every rank must end on the same road."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch

    from warpdemux_amd import _lib, dist

    dist.init_process_group("gloo")

    class Reducer(dist.CountReducer):
        def _rccl_unavailable(self):
            self._L = None
            if scenario == "rank0_unavailable" and rank == 0:
                return "[wdx -2] RCCL unavailable: librccl not found (test)"
            if scenario == "rank1_unavailable" and rank == 1:
                return "[wdx -2] RCCL unavailable: librccl not found (test)"
            return ""

        def _draw_id(self):
            raise _lib.WdxError("[wdx -3] ncclGetUniqueId failed (test)")

    class FakeCtx:          # only its presence matters before wdx_comm_init
        handle = None

    counts = torch.tensor([rank + 1, 10 * (rank + 1)], dtype=torch.int64)
    try:
        red = Reducer(FakeCtx(), prefer="rccl")
        red(counts)
        out_q.put((rank, red.mode, red.note, counts.tolist(), None))
    except Exception as e:  # noqa: BLE001
        out_q.put((rank, "raised", "", counts.tolist(), f"{type(e).__name__}: {e}"))
    dist.barrier()          # the groups' collectives still line up after the failure path
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("scenario", ["rank0_unavailable", "rank1_unavailable", "rank0_id_error"])
def test_rccl_failure_paths_do_not_hang_and_agree(scenario):
    """rank 0 failing before the id broadcast used to leave the other ranks waiting in it, and any
    exception became a silent torch.distributed run.  Now: RCCL missing (WDX_ERR_NO_DEVICE) on any rank ->
    ALL ranks use the process group and say so; any other failure is raised on EVERY rank."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fallback_worker, args=(r, 2, port, scenario, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    if scenario.endswith("unavailable"):
        for rank, mode, note, counts, err in res:
            assert err is None and mode == "torch" and "torch.distributed used on all ranks" in note
            assert counts == [3, 30]
        culprit = 0 if scenario.startswith("rank0") else 1
        assert "this rank" in res[culprit][2] and "another rank" in res[1 - culprit][2]
    else:
        for rank, mode, note, counts, err in res:
            assert mode == "raised" and "rank 0 could not create the RCCL id" in err and "ncclGetUniqueId failed" in err
            assert counts == [rank + 1, 10 * (rank + 1)]


def test_single_process_helpers_are_noops():
    import torch

    from warpdemux_amd import dist

    c = torch.arange(5, dtype=torch.int64)
    assert torch.equal(dist.reduce_counts(c.clone()), c)
    red = dist.CountReducer(None)
    assert red.mode == "single" and torch.equal(red(c.clone()), c)
    assert dist.max_over_ranks(3.5) == 3.5
    dist.barrier()
    assert dist.env_rank_world() == (int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)),
                                     int(os.environ.get("WORLD_SIZE", 1)))


def _worker8(rank, world, port, n_total, out_q):
    """C4's shape on CPU: world 8, one global read range, synthetic reads by GLOBAL index."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GLOO_SOCKET_IFNAME="lo", OMP_NUM_THREADS="1")
    import torch

    torch.set_num_threads(1)
    from oracle import wdx_oracle as orc
    from warpdemux_amd import dist, synth

    dist.init_process_group("gloo")
    lo, hi = dist.shard_range(n_total, rank, world)
    spec = synth.SynthSpec(n_barcodes=10)
    K = 25
    refs = np.random.default_rng(0).normal(size=(10, K))
    call = np.zeros(0, dtype=np.int64)
    if hi > lo:
        sig, off, a_s, a_e, bc = synth.generate_packed(spec, lo, hi - lo)
        fpt, dwell, stats, status = orc.fingerprint_packed(sig, off, a_s, a_e, orc.SegParams(barcode_num_events=K))
        ok = status == 0
        call = np.full(hi - lo, 10, dtype=np.int64)
        call[ok] = orc.argmin_rows(orc.dtw_matrix(fpt[ok], refs, 15, 0.1))
    counts = torch.from_numpy(np.bincount(call, minlength=11).astype(np.int64))
    red = dist.CountReducer(None)
    red(counts)
    per_rank = dist.gather_over_ranks(float(10 * rank + 1))
    out_q.put((rank, lo, hi, counts.numpy(), per_rank, dist.max_over_ranks(float(rank))))
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
def test_eight_rank_shards_tile_the_global_range_and_the_histogram_is_the_single_process_one():
    """BASELINE config 4's layout (8 ranks, contiguous shards of ONE global read range, one int64[11] sum) on CPU:
    the shards tile [0, n) in rank order -- including a total that leaves the last rank short -- every rank ends with
    the same histogram, it sums to n, and it equals the histogram of the same reads in one process; per-rank timings
    come back in rank order (bench.py's per_rank_ms_per_step)."""
    import torch.multiprocessing as mp

    from warpdemux_amd import dist as wdist

    world, n_total = 8, 8 * 6 - 5          # ceil(43 / 8) = 6: ranks 0..6 take 6 reads, rank 7 takes 1
    for total in (0, 1, 7, 8, 9, 43, 40_000_000):
        cover = [wdist.shard_range(total, r, world) for r in range(world)]
        assert cover[0][0] == 0 and cover[-1][1] == total
        assert all(cover[r][1] == cover[r + 1][0] for r in range(world - 1))
        assert all(0 <= hi - lo <= -(-total // world) for lo, hi in cover)
    assert [hi - lo for lo, hi in (wdist.shard_range(40_000_000, r, 8) for r in range(8))] == [5_000_000] * 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [(lo, hi) for _, lo, hi, _, _, _ in res] == [wdist.shard_range(n_total, r, world) for r in range(world)]
    g = res[0][3]
    assert all(np.array_equal(g, c) for _, _, _, c, _, _ in res) and g.sum() == n_total
    assert all(pr == [float(10 * r + 1) for r in range(world)] for _, _, _, _, pr, _ in res)
    assert all(mx == float(world - 1) for *_, mx in res)

    sys.path.insert(0, ROOT)
    from oracle import wdx_oracle as orc
    from warpdemux_amd import synth

    spec = synth.SynthSpec(n_barcodes=10)
    K = 25
    refs = np.random.default_rng(0).normal(size=(10, K))
    sig, off, a_s, a_e, bc = synth.generate_packed(spec, 0, n_total)
    fpt, dwell, stats, status = orc.fingerprint_packed(sig, off, a_s, a_e, orc.SegParams(barcode_num_events=K))
    ok = status == 0
    call = np.full(n_total, 10, dtype=np.int64)
    call[ok] = orc.argmin_rows(orc.dtw_matrix(fpt[ok], refs, 15, 0.1))
    assert np.array_equal(g, np.bincount(call, minlength=11))

#!/usr/bin/env python3
"""Headline benchmark: reads/s demultiplexed on synthetic RNA004 adapter signals.

One "step" = one pass of the fused hot path (fingerprint -> banded DTW against the barcode
references -> call -> count histogram [-> all-reduce]) over the rank's whole batch of raw adapter
rows, which is resident in HBM when the timed region starts (generated on the device by the
"wdx-synth v1" kernels).  Workload at N=1: BASELINE.json configs[2] (C3) -- 10 M reads, WDX10
shape (10 barcodes x 110-point fingerprints, window 15, penalty 0.1).  For N>1 every rank holds
its own shard of the same size (weak scaling, no data-path collective; one int64[11] all-reduce
of the call histogram per step).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--reads R]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` (dominant kernel,
HIP-event timed on the launch stream) and `cpu_baseline` (the CPU oracle -- a port of the
reference's CPU path -- timed on this host on a bounded sample of the same reads).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
N_BARCODES = 10
K_FPT = 110
WINDOW = 15
PENALTY = 0.1


def _cpu_minibatch(args):
    """Worker of the CPU baseline: one 1000-read minibatch, driven like file_proc.py:418-450."""
    sig, off, a_s, a_e, refs = args
    from oracle import wdx_oracle as orc

    p = orc.SegParams(barcode_num_events=K_FPT)
    fpt, dwell, stats, status = orc.fingerprint_packed(sig, off, a_s, a_e, p)
    ok = status == 0
    D = orc.dtw_matrix(fpt[ok], refs, WINDOW, PENALTY)
    call = np.full(status.size, -1, dtype=np.int32)
    call[ok] = orc.argmin_rows(D)
    return call, status, D


def cpu_baseline(sig_h, off_h, a_s_h, a_e_h, refs, gpu_call, gpu_status, gpu_dist):
    """Oracle (kind="port") on all host cores, minibatches farmed to a process pool like
    file_proc.run_demux, + a 1-core figure; also returns whether the GPU results on the same sample
    are identical."""
    from concurrent.futures import ProcessPoolExecutor

    n = off_h.size - 1
    cores = os.cpu_count() or 1
    mb = max(100, min(1000, -(-n // cores)))  # reference minibatch is 1000 reads; shrink to occupy every core
    jobs = []
    for lo in range(0, n, mb):
        hi = min(n, lo + mb)
        o = off_h[lo:hi + 1]
        jobs.append((sig_h[o[0]:o[-1]], (o - o[0]).copy(), a_s_h[lo:hi], a_e_h[lo:hi], refs))
    # 1 core: a few minibatches in this process
    n1 = max(1, min(len(jobs), 2000 // mb))
    t0 = time.perf_counter()
    one = [_cpu_minibatch(j) for j in jobs[:n1]]
    t1 = time.perf_counter()
    single = sum(j[2].size for j in jobs[:n1]) / (t1 - t0)
    del one
    with ProcessPoolExecutor(max_workers=cores) as ex:
        list(ex.map(_cpu_noop, range(cores * 2)))  # start the workers before the clock
        t0 = time.perf_counter()
        res = list(ex.map(_cpu_minibatch, jobs))
        t1 = time.perf_counter()
    multi = n / (t1 - t0)
    call = np.concatenate([r[0] for r in res])
    status = np.concatenate([r[1] for r in res])
    D = np.concatenate([r[2] for r in res])
    ok = status == 0
    parity = bool(np.array_equal(call, gpu_call) and np.array_equal(status, gpu_status)
                  and np.array_equal(D, gpu_dist[ok]))
    return multi, cores, single, parity, mb


def _cpu_noop(i):
    return i


def make_refs(spec_clean, synth, sig_proc):
    """Barcode reference fingerprints: one low-noise template read per barcode through the HIP
    fingerprint kernel (host-buffer entry point)."""
    ids, rid = {}, 0
    while len(ids) < N_BARCODES:
        b = int(synth.read_layout(spec_clean, np.array([rid]))[0][0])
        ids.setdefault(b, rid)
        rid += 1
    rows = [synth.generate_read(spec_clean, ids[b])[0] for b in range(N_BARCODES)]
    stride = max(r.size for r in rows)
    mb = np.full((N_BARCODES, stride), np.nan, dtype=np.float32)
    for b, r in enumerate(rows):
        mb[b, : r.size] = r
    a_s = np.full(N_BARCODES, synth.PAD, dtype=np.int32)
    a_e = np.array([r.size - synth.PAD for r in rows], dtype=np.int32)
    # the 10 template reads go through the exact one-kernel path so that every launch of the fast kernel
    # seen by a profiler belongs to the timed workload (same results either way)
    os.environ["WDX_FORCE_SLOW"] = "1"
    try:
        fb = sig_proc.fingerprint_batch(mb, a_s, a_e, sig_proc.SegParams(barcode_num_events=K_FPT))
    finally:
        del os.environ["WDX_FORCE_SLOW"]
    if not (fb.status == 0).all():
        raise RuntimeError("template fingerprinting failed")
    return fb.fpt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU (C3 = 10 M)")
    ap.add_argument("--cpu-sample", type=int, default=64000, help="reads timed on the CPU oracle")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--calib", action="store_true", help="also stream the signal buffer once with the calibration "
                    "kernel (known byte count for the FETCH_SIZE counter; tools/collect_traffic.py)")
    args = ap.parse_args()

    import torch

    from warpdemux_amd import _lib, dist, sig_proc, synth
    from warpdemux_amd.engine import DemuxEngine

    # one process per GPU; backend "nccl" (= RCCL over xGMI).  WDX_BENCH_BACKEND=gloo lets the
    # multi-process logic be exercised on a box with fewer GPUs than ranks (ranks then share devices).
    backend = os.environ.get("WDX_BENCH_BACKEND") or None
    n_dev = torch.cuda.device_count()
    if backend == "gloo" and n_dev > 0:
        os.environ["LOCAL_RANK"] = str(int(os.environ.get("LOCAL_RANK", "0")) % n_dev)
    rank, local_rank, world = dist.init_process_group(backend)
    if world != args.gpus:
        if rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    torch.cuda.set_device(local_rank)
    tdev = torch.device("cuda", local_rank)
    host_collectives = backend == "gloo"

    spec = synth.SynthSpec(n_barcodes=N_BARCODES)
    clean = synth.SynthSpec(n_barcodes=N_BARCODES, noise_sigma=0.25, spikes=False)
    os.environ["WDX_DEVICE"] = str(local_rank)
    refs = make_refs(clean, synth, sig_proc)
    params = sig_proc.SegParams(barcode_num_events=K_FPT)
    eng = DemuxEngine(refs, WINDOW, PENALTY, params, device=local_rank)

    # ---- inputs: generated straight into HBM; shrink if the device cannot hold the batch --------
    n_reads = args.reads
    while True:
        try:
            first = rank * n_reads
            sig, off, a_s, a_e, bc, max_len = eng.synth_packed(spec, first, n_reads)
            res = eng.demux(sig, a_s, a_e, offsets=off, max_len=max_len)  # allocates outputs/workspace
            torch.cuda.synchronize()
            break
        except (torch.OutOfMemoryError, RuntimeError, _lib.WdxError) as e:  # noqa: PERF203
            if n_reads <= 100_000:
                raise
            sig = off = a_s = a_e = bc = res = None
            eng._work = None
            torch.cuda.empty_cache()
            if rank == 0:
                print(f"note: {n_reads} reads did not fit ({type(e).__name__}); halving", file=sys.stderr)
            n_reads //= 2
    total_samples = int(off[-1].item())

    def step():
        res.counts.zero_()
        eng.demux(sig, a_s, a_e, offsets=off, max_len=max_len, out=res)
        if host_collectives:
            c = res.counts.cpu()
            dist.reduce_counts(c)
            res.counts.copy_(c)
        else:
            dist.reduce_counts(res.counts)

    if args.calib:
        import ctypes as C

        dummy = torch.zeros(4, dtype=torch.float32, device=tdev)
        _lib.check(eng.L.wdx_calib_read_dev(eng.ctx.handle, C.c_void_p(sig.data_ptr()), total_samples,
                                            C.c_void_p(dummy.data_ptr()), None))
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    eng.kernel_time_reset()
    eng.kernel_timing(True)
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dist.barrier()
    t1 = time.perf_counter()
    eng.kernel_timing(False)
    elapsed = dist.max_over_ranks(t1 - t0, device=tdev if (world > 1 and not host_collectives) else None)

    fp_ms, fp_n = eng.kernel_time(_lib.K_FINGERPRINT)
    dtw_ms, dtw_n = eng.kernel_time(_lib.K_DTW)
    tr_ms, tr_n = eng.kernel_time(_lib.K_TRANSPOSE)
    cnt_ms, cnt_n = eng.kernel_time(_lib.K_COUNT)

    counts = res.counts.cpu().numpy()
    n_fail = int(counts[N_BARCODES])
    if rank == 0 and int(counts.sum()) != n_reads * world:
        raise RuntimeError(f"call histogram does not add up: {counts.sum()} != {n_reads * world}")
    if rank == 0 and n_fail > 0.01 * n_reads * world:
        raise RuntimeError(f"{n_fail} of {n_reads * world} reads failed: the synthetic workload should fingerprint cleanly")

    # ---- roofline of the dominant kernel (algorithmic bytes, DESIGN.md "Measurement") ------------
    n_ok = n_reads  # outputs are written for every read (NaN rows for failures)
    fp_bytes = 4.0 * total_samples + 8.0 * K_FPT * n_ok + 4.0 * n_reads
    dtw_bytes = (8.0 * K_FPT + 4.0 * N_BARCODES + 4.0) * n_reads
    if fp_ms >= dtw_ms:
        dom, dom_ms, dom_n, dom_bytes = "fingerprint_fast_kernel", fp_ms, fp_n, fp_bytes
    else:
        dom, dom_ms, dom_n, dom_bytes = "dtw_band_kernel<15>", dtw_ms, dtw_n, dtw_bytes
    avg_ms = dom_ms / max(dom_n, 1)
    dom_bytes = dom_bytes / max(dom_n // max(args.steps, 1), 1)  # per kernel launch (a step may be sliced)
    achieved = dom_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    # HBM traffic of that kernel from the PMC passes (profiles/traffic.json, written by
    # tools/collect_traffic.py on the same workload); null when not collected for this size
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as fh:
            tj = json.load(fh)
        if tj.get("kernel") == dom and tj.get("reads_per_launch") == n_reads // max(dom_n // max(args.steps, 1), 1):
            traffic = tj.get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        pass

    valu_busy = None
    try:
        with open(os.path.join(ROOT, "profiles", "r01i_sq_counters.json")) as fh:
            pl = json.load(fh)["kernels"][dom.split("<")[0]]["per_launch"]
        valu_busy = pl["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * pl["GRBM_GUI_ACTIVE"] / 8.0)
    except (OSError, ValueError, KeyError):
        pass

    out = None
    if rank == 0:
        reads_total = n_reads * world
        value = reads_total * args.steps / elapsed
        fused_bytes_per_read = 4.0 * total_samples / n_reads + 4.0 * N_BARCODES + 4.0
        out = {
            "metric": "reads/sec demuxed (110-pt DTW x 10 barcodes)",
            "value": value,
            "unit": "reads/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": ("C3: %d synthetic RNA004 adapter reads per GPU (wdx-synth v1, mean %.0f samples), "
                             "WDX10 shape: 10 barcodes x 110-pt fingerprints, window 15, penalty 0.1; fused "
                             "fingerprint+DTW+call+count, raw rows resident in HBM" % (n_reads, total_samples / n_reads)),
                "reads_per_gpu": n_reads,
                "reads_total": reads_total,
                "failed_reads": n_fail,
                "sharding": "contiguous read shards, one process per GPU, int64[11] count all-reduce per step",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": dom,
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "algorithmic_bytes_per_launch": dom_bytes,
                "avg_launch_ms": avg_ms,
                "launches": dom_n,
                # what actually bounds the kernel (float64 VALU issue), from the committed PMC pass
                "valu_busy_frac": valu_busy,
            },
            "kernels_ms_per_step": {   # HIP-event sums over all launches of a step (a step may be sliced)
                "fingerprint": fp_ms / max(args.steps, 1), "dtw": dtw_ms / max(args.steps, 1),
                "transpose": tr_ms / max(args.steps, 1), "count": cnt_ms / max(args.steps, 1),
            },
            "fused_path": {
                "algorithmic_bytes_per_read": fused_bytes_per_read,
                "hbm_frac_whole_job_per_gpu": value / world * fused_bytes_per_read / 1e9 / HBM_PEAK_GBS,
                "dtw_gcups_per_gpu": (n_reads * 29800.0) / (dtw_ms / max(dtw_n, 1) * 1e-3) / 1e9 if dtw_ms else None,
            },
        }

    # ---- CPU baseline + parity on a bounded sample (rank 0, N=1 only) ------------------------------
    if rank == 0 and world == 1 and not args.no_cpu:
        ns = min(args.cpu_sample, n_reads)
        end = int(off[ns].item())
        sig_h = sig[:end].cpu().numpy()
        off_h = off[: ns + 1].cpu().numpy()
        multi, cores, single, parity, mb = cpu_baseline(
            sig_h, off_h, a_s[:ns].cpu().numpy(), a_e[:ns].cpu().numpy(), refs,
            res.call[:ns].cpu().numpy(), res.status[:ns].cpu().numpy(), res.dist[:ns].cpu().numpy())
        out["cpu_baseline"] = {
            "value": multi,
            "unit": "reads/s",
            "cores": cores,
            "kind": "port",
            "sample": "first %d reads of the same workload, oracle/wdx_oracle.c driven as %d-read minibatches over "
                      "a ProcessPoolExecutor(%d) like file_proc.run_demux (IPC included); dtaidistance itself is "
                      "not available" % (ns, mb, cores),
            "single_core_value": single,
        }
        out["parity_on_sample"] = parity
        if not parity:
            print("ERROR: GPU results differ from the oracle on the CPU sample", file=sys.stderr)
    elif rank == 0:
        out["cpu_baseline"] = None

    if rank == 0:
        print(json.dumps(out))
    eng.close()
    if world > 1:
        import torch.distributed as tdist

        tdist.destroy_process_group()
    if out is not None and out.get("parity_on_sample") is False:
        sys.exit(2)


if __name__ == "__main__":
    main()

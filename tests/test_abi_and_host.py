"""CPU-side checks: the C-ABI library loads and exports every declared symbol, the host shims
validate like the reference, and -- with no GPU here -- every compute entry point fails loudly
instead of falling back to anything."""
import os
import re

import numpy as np
import pytest

from warpdemux_amd import _lib, dist, parallel_distances as pdist, sig_proc, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _have_gpu():
    return _lib.load().wdx_device_count() > 0


def test_header_symbols_are_exported():
    hdr = open(os.path.join(ROOT, "include", "wdx.h")).read()
    declared = set(re.findall(r"\b(wdx_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"wdx_seg_params"}
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    L = _lib.load()
    for name in sorted(declared):
        assert hasattr(L, name), name
    assert L.wdx_abi_version() == _lib.ABI_VERSION == 4


def test_option_and_kernel_constants_match_the_header():
    """every WDX_OPT_* / WDX_K_* / WDX_WANT_* the header defines has the same value in the ctypes layer (and vice versa)"""
    hdr = open(os.path.join(ROOT, "include", "wdx.h")).read()
    defs = {m.group(1): int(m.group(2), 0) for m in re.finditer(r"^#define\s+WDX_((?:OPT|K|WANT)_[A-Z0-9_]+)\s+(0x[0-9a-fA-F]+|\d+)u?\b", hdr, flags=re.M)}
    assert defs["OPT_DTW_UNFUSED"] == 16 and defs["OPT_NO_SPLIT_TAIL"] == 15
    mine = {k: getattr(_lib, k) for k in dir(_lib) if re.match(r"(OPT|K|WANT)_[A-Z0-9_]+$", k)}
    assert mine == defs, {k: (mine.get(k), defs.get(k)) for k in set(mine) ^ set(defs) | {k for k in mine if k in defs and mine[k] != defs[k]}}


def test_seg_params_struct_layout_matches_header():
    hdr = open(os.path.join(ROOT, "include", "wdx.h")).read()
    body = hdr[hdr.index("typedef struct wdx_seg_params {"): hdr.index("} wdx_seg_params;")]
    fields = re.findall(r"^\s*(int32_t|float|double)\s+(\w+);", body, flags=re.M)
    assert [f[1] for f in fields] == [f[0] for f in _lib.SegParamsC._fields_]
    import ctypes
    assert ctypes.sizeof(_lib.SegParamsC) == 48


def test_no_silent_fallback_without_gpu():
    if _have_gpu():
        pytest.skip("GPU present")
    X, Y = np.zeros((4, 25)), np.zeros((3, 25))
    with pytest.raises(_lib.WdxError, match="ROCm-capable|HIP|device"):
        pdist.distance_matrix_to(X, Y, window=15, penalty=0.1, n_jobs=1)
    with pytest.raises(_lib.WdxError):
        sig_proc.fingerprint_batch(np.zeros((2, 100), np.float32), [0, 0], [100, 100], sig_proc.SegParams())
    # the pipelined worker API and its page-locked buffers: no host fallback either
    from warpdemux_amd import pipeline

    with pytest.raises(_lib.WdxError):
        pipeline.pinned_empty((4, 16), np.float32)
    with pytest.raises(_lib.WdxError):
        pipeline.MinibatchPipeline(np.zeros((3, 25)), 15, 0.1)
    L = _lib.load()
    assert L.wdx_comm_available() in (0, _lib.WDX_ERR_NO_DEVICE)     # a local probe: never needs a device


def test_reference_error_behaviour_is_mirrored():
    X, Y = np.zeros((4, 25)), np.zeros((3, 25))
    # parallel_distances.py:70-73
    with pytest.raises(ValueError, match="block_size must be specified when using parallel."):
        pdist.distance_matrix_to(X, Y, window=15, penalty=0.1, n_jobs=4)
    with pytest.raises(ValueError, match="block_size must be specified"):
        pdist.distance_matrix_to(X, Y, n_jobs=-1)
    # np.vstack([X, Y]) column mismatch is a ValueError in the reference
    with pytest.raises(ValueError):
        pdist.distance_matrix_to(X, np.zeros((3, 24)), n_jobs=1)
    # sig_proc.py:131-134
    with pytest.raises(ValueError, match="not recognized"):
        sig_proc.SegParams(seg_norm="zscore").to_c()


def test_install_patches_reference_modules(monkeypatch):
    import sys
    import types

    import warpdemux_amd

    pkg = types.ModuleType("warpdemux")
    pd = types.ModuleType("warpdemux.parallel_distances")
    pd.distance_matrix_to = lambda *a, **k: "ref"
    models = types.ModuleType("warpdemux.models")
    svm = types.ModuleType("warpdemux.models.dtw_svm")
    svm.distance_matrix_to = pd.distance_matrix_to
    for name, m in (("warpdemux", pkg), ("warpdemux.parallel_distances", pd), ("warpdemux.models", models),
                    ("warpdemux.models.dtw_svm", svm)):
        monkeypatch.setitem(sys.modules, name, m)
    warpdemux_amd.install()
    assert pd.distance_matrix_to is pdist.distance_matrix_to
    assert svm.distance_matrix_to is pdist.distance_matrix_to
    assert pd.parallel_distance_matrix is pdist.parallel_distance_matrix


def test_synth_is_deterministic_and_sane():
    spec = synth.SynthSpec(n_barcodes=10)
    a, off, a_s, a_e, bc = synth.generate_packed(spec, 42, 12)
    b, off2, *_ = synth.generate_packed(spec, 42, 12)
    assert np.array_equal(a, b) and np.array_equal(off, off2)
    # shard independence: reads are a function of the GLOBAL read index only
    c, off3, *_ = synth.generate_packed(spec, 45, 3)
    assert np.array_equal(c, a[off[3]:off[6]])
    lens = np.diff(off)
    assert 3500 < lens.mean() < 6500 and a.dtype == np.float32
    assert set(bc.tolist()) <= set(range(10))
    assert np.all(a_s == 100) and np.array_equal(a_e, lens - 100)
    big = synth.read_layout(spec, np.arange(4000))[3]
    assert 4500 < big.mean() < 5100


def test_shard_ranges_partition_the_reads():
    for n in (0, 1, 7, 1000, 40_000_000):
        for world in (1, 2, 3, 8):
            parts = [dist.shard_range(n, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            assert max(hi - lo for lo, hi in parts) <= -(-n // world)
    with pytest.raises(ValueError):
        dist.shard_range(10, 2, 2)


def test_bench_launcher_builds_a_torchrun_command(monkeypatch):
    """`python bench.py --gpus N` without a torchrun environment starts N ranks itself, as children, before any
    GPU call in the parent (the driver invokes bench.py exactly like that)."""
    import importlib
    import subprocess
    import sys

    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    seen = {}

    class R:
        returncode = 0

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return R()

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2"])
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    assert "torch" not in bench.main.__code__.co_names   # the launcher path imports no torch
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "2"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_refinement_parameters_validate_like_the_reference():
    from types import SimpleNamespace as NS

    seg = NS(min_obs_per_base=9, running_stat_width=18, num_events=120, accept_less_cpts=False, normalization="mean",
             barcode_num_events=[25, 20], consensus_refinement=True, consensus_subseq_match_normalization="mean",
             consensus_subseq_match_penalty=1.5, consensus_subseq_match_psi=[5, 0, 40, 0],
             consensus_subseq_match_ub_start=18, consensus_subseq_match_lb_end=69, consensus_subseq_match_ub_end=97,
             refinement_optimal_cpts=False)
    spc = NS(sig_extract=NS(padding=100, normalization="none"), core=NS(sig_norm_outlier_thresh=5.0), segmentation=seg)
    q = np.linspace(-1, 1, 84)
    rp = sig_proc.RefineParams.from_spc(spc, q)
    assert (rp.barcode_segm_events, rp.barcode_keep_events, rp.psi, rp.penalty) == (25, 20, (5, 0, 40, 0), 1.5)
    assert sig_proc.SegParams.from_spc(spc).barcode_num_events == 20       # K of the outputs = barcode_num_events[1]
    c = rp.to_c()
    assert c.n_query == 84 and list(c.psi) == [5, 0, 40, 0] and c.barcode_keep_events == 20
    seg.barcode_num_events = 25          # sig_proc.py:455-459
    with pytest.raises(ValueError, match="use a tuple instead"):
        sig_proc.RefineParams.from_spc(spc, q)
    seg.barcode_num_events = [25, 25]
    seg.refinement_optimal_cpts = True   # ruptures KernelCPD: not offered
    with pytest.raises(NotImplementedError):
        sig_proc.RefineParams.from_spc(spc, q)
    seg.refinement_optimal_cpts = False
    with pytest.raises(ValueError):
        sig_proc.RefineParams.from_spc(spc, np.zeros((2, 2)))
    seg.consensus_subseq_match_normalization = "zscore"
    with pytest.raises(ValueError, match="not recognized"):
        sig_proc.RefineParams.from_spc(spc, q).to_c()


def test_a_configuration_that_admits_windows_beyond_the_engines_cap_is_refused_once():
    """adapter windows beyond WDX_MAX_ADAPTER_SAMPLES come back status 5 from the kernels, where the
    reference (sig_proc.py:382-391) fingerprints any length.  `--export core.max_obs_trace=...` is what makes such windows
    possible: SegParams.from_spc -- the start of every shim that takes the reference's config -- refuses that
    configuration with the limit in the message; the shipped values (10 000 / 15 000 + 2 x 100) pass."""
    from types import SimpleNamespace as NS

    seg = NS(min_obs_per_base=6, running_stat_width=12, num_events=110, accept_less_cpts=False, normalization="mean",
             barcode_num_events=25, consensus_refinement=False)
    for mot in (10000, 15000, 16184):
        spc = NS(sig_extract=NS(padding=100, normalization="none"), core=NS(sig_norm_outlier_thresh=5.0, max_obs_trace=mot), segmentation=seg)
        assert sig_proc.SegParams.from_spc(spc).num_events == 110
    spc = NS(sig_extract=NS(padding=100, normalization="none"), core=NS(sig_norm_outlier_thresh=5.0, max_obs_trace=16185), segmentation=seg)
    with pytest.raises(NotImplementedError, match="16384"):
        sig_proc.SegParams.from_spc(spc)
    with pytest.raises(NotImplementedError, match="max_obs_trace = 40000"):
        sig_proc.detect_results_to_fpt_batch(np.zeros((1, 100), np.float32), NS(sig_extract=NS(padding=100, normalization="none"),
                                             core=NS(sig_norm_outlier_thresh=5.0, max_obs_trace=40000), segmentation=seg),
                                             [sig_proc.DetectResults(True, "", 0, 50)])


def _feeder_ring(n_slots=2, max_reads=10, max_stride=100, n_refs=4, n_events=0, n_classes=0, K=25):
    import ctypes as C
    import mmap

    L = _lib.load()
    geo = _lib.FeederGeometryC(n_slots, n_events, n_classes, 0, max_reads, max_stride, n_refs)
    n = L.wdx_feeder_ring_bytes(C.byref(geo))
    assert n % 4096 == 0 and n >= 4096 + n_slots * max_reads * max_stride * 4
    m = mmap.mmap(-1, n)      # MAP_SHARED | MAP_ANONYMOUS: forked children see the same ring
    base = C.c_void_p(C.addressof(C.c_char.from_buffer(m)))
    pc = sig_proc.SegParams(barcode_num_events=n_events or K).to_c()
    return L, geo, n, m, base, pc


def test_feeder_ring_geometry_and_the_worker_side_without_a_feeder():
    """wdx_feeder_* (many worker processes, one GPU-facing process): the ring's size and layout, argument checks of the
    worker-side calls, and their answer when no feeder serves the ring -- none of which needs a GPU (the worker side makes
    no HIP call by design)."""
    import ctypes as C

    L = _lib.load()
    G = _lib.FeederGeometryC
    for bad in (G(0, 0, 0, 0, 10, 10, 4), G(33, 0, 0, 0, 10, 10, 4), G(2, 0, 0, 0, 0, 10, 4), G(2, 0, 17, 0, 10, 10, 4),
                G(2, -1, 0, 0, 10, 10, 4)):
        assert L.wdx_feeder_ring_bytes(C.byref(bad)) == 0
    assert L.wdx_feeder_ring_bytes(None) == 0
    L, geo, n, m, base, pc = _feeder_ring()
    sig = np.zeros((3, 50), np.float32)
    a = np.zeros(3, np.int32)
    d = np.zeros((3, 4), np.float32)
    c = np.zeros(3, np.int32)
    args = (sig.ctypes.data, 3, 50, a.ctypes.data, a.ctypes.data, None)
    with pytest.raises(ValueError, match="not an initialised ring"):
        _lib.check(L.wdx_feeder_demux(base, *args, 4, d.ctypes.data, c.ctypes.data, c.ctypes.data))
    with pytest.raises(ValueError):
        _lib.check(L.wdx_feeder_ring_init(base, n - 1, C.byref(geo), C.byref(pc)))
    with pytest.raises(ValueError):
        _lib.check(L.wdx_feeder_ring_init(base, n, C.byref(geo), None))
    Lk, geo_k, nk, mk, base_k, _ = _feeder_ring(n_events=30)    # n_events must be the parameters' barcode_num_events (25 here)
    with pytest.raises(ValueError, match="n_events"):
        _lib.check(L.wdx_feeder_ring_init(base_k, nk, C.byref(geo_k), C.byref(pc)))
    del base_k
    _lib.check(L.wdx_feeder_ring_init(base, n, C.byref(geo), C.byref(pc)))
    with pytest.raises(ValueError, match="do not fit"):
        _lib.check(L.wdx_feeder_demux(base, sig.ctypes.data, 11, 50, a.ctypes.data, a.ctypes.data, None, 4, d.ctypes.data,
                                      c.ctypes.data, c.ctypes.data))
    with pytest.raises(ValueError, match="references"):
        _lib.check(L.wdx_feeder_demux(base, *args, 5, d.ctypes.data, c.ctypes.data, c.ctypes.data))
    # outputs the ring has no room for (laid out with n_events = 0, n_classes = 0), unknown bits, missing destinations
    f = np.zeros((3, 25))
    for want, kw in ((_lib.WANT_FPT, dict(fpt=f)), (_lib.WANT_SVM, dict(prob=f, pred=c, conf=f)), (0x80, {})):
        job = _lib.FeederJobC(sig.ctypes.data, 3, 50, a.ctypes.data, a.ctypes.data, None, want, 0, c.ctypes.data, c.ctypes.data,
                              None, _lib.addr(kw.get("fpt")), None, None, _lib.addr(kw.get("prob")), _lib.addr(kw.get("pred")),
                              _lib.addr(kw.get("conf")))
        with pytest.raises(ValueError, match="without room"):
            _lib.check(L.wdx_feeder_run(base, C.byref(job)))
    job = _lib.FeederJobC(sig.ctypes.data, 3, 50, a.ctypes.data, a.ctypes.data, None, _lib.WANT_DIST, 0, c.ctypes.data,
                          c.ctypes.data, None, None, None, None, None, None, None)
    with pytest.raises(ValueError, match="no destination"):
        _lib.check(L.wdx_feeder_run(base, C.byref(job)))
    with pytest.raises(ValueError, match="without a model"):
        _lib.check(L.wdx_feeder_predict(base, f.ctypes.data, 3, f.ctypes.data, c.ctypes.data, f.ctypes.data))
    served, recl, free = C.c_int64(-1), C.c_int64(-1), C.c_int32(-1)
    _lib.check(L.wdx_feeder_served(base, C.byref(served)))
    _lib.check(L.wdx_feeder_stats(base, C.byref(served), C.byref(recl), C.byref(free)))
    assert (served.value, recl.value, free.value) == (0, 0, 2)
    assert L.wdx_feeder_alive(base) == 0          # nobody serves this ring
    _lib.check(L.wdx_feeder_stop(base))
    with pytest.raises(_lib.WdxNoDevice, match="feeder"):      # stopped, nobody serves: the worker is told, it does not hang
        _lib.check(L.wdx_feeder_demux(base, *args, 4, d.ctypes.data, c.ctypes.data, c.ctypes.data))
    del base


def test_feeder_ring_takes_back_the_slot_of_a_worker_that_died():
    """ADVICE r5 / VERDICT r5 weak 8b: a worker that is killed while it holds a ring slot must not cost the ring that slot.
    The owner's pid is part of the slot's state word; a claimant that finds the ring full gives slots of dead owners back.
    No GPU: the test hook wdx_feeder_selftest(ring, 1) claims a slot the way a worker does and leaves it FILLING."""
    import ctypes as C
    import signal
    import time

    L, geo, n, m, base, pc = _feeder_ring(n_slots=1)
    _lib.check(L.wdx_feeder_ring_init(base, n, C.byref(geo), C.byref(pc)))
    pid = os.fork()
    if pid == 0:      # the worker: claims the ring's only slot, then is killed with it in its hands
        rc = L.wdx_feeder_selftest(base, 1)
        os.kill(os.getpid(), signal.SIGKILL if rc == 0 else signal.SIGTERM)
        os._exit(3)
    t0 = time.monotonic()
    free = C.c_int32(-1)
    while time.monotonic() - t0 < 20:       # (not reaped: the dead worker is a zombie, which kill(pid, 0) calls alive)
        _lib.check(L.wdx_feeder_stats(base, None, None, C.byref(free)))
        if free.value == 0 and open(f"/proc/{pid}/stat").read().rsplit(") ", 1)[1][0] == "Z":
            break
        time.sleep(0.01)
    assert free.value == 0, "the child did not claim the slot"
    t0 = time.monotonic()
    s = L.wdx_feeder_selftest(base, 1)      # the ring is full: the claimant finds the owner dead and takes the slot over
    assert s == 0 and time.monotonic() - t0 < 10
    recl = C.c_int64(0)
    _lib.check(L.wdx_feeder_stats(base, None, C.byref(recl), C.byref(free)))
    assert recl.value == 1 and free.value == 0
    os.waitpid(pid, 0)
    del base


def test_a_dead_feeder_is_noticed_while_it_is_still_a_zombie():
    """kill(pid, 0) answers 0 for a zombie, and a feeder that dies under a parent blocked in pool.map
    is never reaped -- the workers read /proc/<pid>/stat's state (and the heartbeat) instead.  No GPU: the hook
    wdx_feeder_selftest(ring, 2) makes a child pose as the serving feeder; it is killed and NOT waited for."""
    import ctypes as C
    import signal
    import time

    L, geo, n, m, base, pc = _feeder_ring(n_slots=1)
    _lib.check(L.wdx_feeder_ring_init(base, n, C.byref(geo), C.byref(pc)))
    r, w = os.pipe()
    pid = os.fork()
    if pid == 0:
        os.close(r)
        L.wdx_feeder_selftest(base, 2)
        os.write(w, b"x")
        time.sleep(60)
        os._exit(0)
    os.close(w)
    assert os.read(r, 1) == b"x"
    assert L.wdx_feeder_alive(base) == 1
    assert L.wdx_feeder_selftest(base, 1) == 0          # a worker takes the only slot (and keeps it: this process lives)
    os.kill(pid, signal.SIGKILL)                           # ... the feeder dies, nobody reaps it
    t0 = time.monotonic()
    while L.wdx_feeder_alive(base) == 1 and time.monotonic() - t0 < 10:
        time.sleep(0.01)
    assert L.wdx_feeder_alive(base) == 0 and time.monotonic() - t0 < 5
    assert open(f"/proc/{pid}/stat").read().rsplit(") ", 1)[1][0] == "Z"     # still a zombie: kill(pid, 0) would say "alive"
    sig = np.zeros((3, 50), np.float32)
    a = np.zeros(3, np.int32)
    d = np.zeros((3, 4), np.float32)
    c = np.zeros(3, np.int32)
    t0 = time.monotonic()
    with pytest.raises(_lib.WdxNoDevice, match="died"):    # the ring is full and its feeder dead: told, not hung
        _lib.check(L.wdx_feeder_demux(base, sig.ctypes.data, 3, 50, a.ctypes.data, a.ctypes.data, None, 4, d.ctypes.data,
                                      c.ctypes.data, c.ctypes.data))
    assert time.monotonic() - t0 < 10
    os.waitpid(pid, 0)
    del base


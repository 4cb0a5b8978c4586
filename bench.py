#!/usr/bin/env python3
"""Headline benchmark: reads/s demultiplexed on synthetic RNA004 adapter signals.

One "step" = one pass of the fused hot path (fingerprint -> banded DTW against the barcode
references -> call -> count histogram -> count all-reduce) over the rank's whole shard of raw adapter
rows, which is resident in HBM when the timed region starts (generated on the device by the
"wdx-synth v1" kernels).

* N=1: BASELINE.json configs[2] (C3) -- 10 M reads, WDX10 shape (10 barcodes x 110-point
  fingerprints, window 15, penalty 0.1).
* N>1: BASELINE.json configs[3] (C4) -- 5 M reads per GPU (40 M at N=8), contiguous shards of one
  global read range (`dist.shard_range`), no data-path collective, ONE int64[11] all-reduce of the
  call histogram per step through the C ABI's RCCL communicator (`wdx_reduce_counts`).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--reads R]

With `--gpus N` (N>1) and no torchrun environment this process starts the N ranks itself
(`python -m torch.distributed.run --nproc-per-node N ... bench.py`) BEFORE anything touches the GPU
and forwards the ranks' output; under torchrun it is one rank.

Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` (dominant kernel,
HIP-event timed on the launch stream), `cpu_baseline` (the CPU oracle -- a port of the reference's
CPU path -- timed on this host on a bounded sample of the same reads, which doubles as the parity
gate) and, at N=1, `secondary` (the shipped-model DTW regime, host-buffer minibatches, live ticks
and DTW_SVM.predict, each with its own parity check).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
N_BARCODES = 10
K_FPT = 110
WINDOW = 15
PENALTY = 0.1
CELLS_110 = sum(min(110, i + 15) - max(0, i - 14) for i in range(110))  # 2 980 per pair
CELLS_25 = sum(min(25, i + 15) - max(0, i - 14) for i in range(25))     # 515 per pair


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=0, help="reads per GPU (default: C3 = 10 M at N=1, C4 = 5 M at N>1)")
    ap.add_argument("--cpu-seconds", type=float, default=33.0,
                    help="time budget of the CPU baseline / parity gate (oracle on all host cores)")
    ap.add_argument("--parity-reads", type=int, default=1_000_000,
                    help="upper bound of reads verified against the oracle inside the time budget; a run with "
                         "--cpu-seconds 0 verifies exactly this many")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--leg", default="", help="run ONE secondary leg alone and print its JSON (shipped_model_e2e)")
    ap.add_argument("--ctx-opt", action="append", default=[], metavar="ID=VALUE",
                    help="diagnostic: wdx_ctx_set_option(ID, VALUE) on the bench context (experiments only; the line "
                         "records it under config.ctx_options)")
    ap.add_argument("--calib", action="store_true", help="also stream the signal buffer once with the calibration "
                    "kernel (known byte count for the FETCH_SIZE counter; tools/collect_traffic.py)")
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` starts its own ranks (children first, no GPU call in this process)
# ----------------------------------------------------------------------------------------------------
def spawn_ranks(args) -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.run(cmd, env=env).returncode


# ----------------------------------------------------------------------------------------------------
# CPU baseline = the oracle (a port of the reference's CPU path) on the host cores, and the parity gate
# ----------------------------------------------------------------------------------------------------
def effective_cores() -> int:
    """CPUs this process may actually use: the affinity mask and the cgroup CPU quota both bound os.cpu_count()
    (the GPU boxes show 256 logical CPUs behind a 16-CPU quota)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
                q, per = int(fq.read()), int(fp.read())
            if q > 0:
                n = min(n, max(1, int(q / per + 0.5)))
        except (OSError, ValueError):
            pass
    return n


def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as fh:
            for ln in fh:
                if ln.lower().startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform

    return platform.processor() or platform.machine()


def _oracle_minibatch(job):
    """One <=1000-read minibatch driven like file_proc.py:418-450: per-read fingerprints, then one
    distance_matrix_to(n_jobs=1) call and the nearest-reference call.  Runs in a worker THREAD (ctypes
    drops the GIL inside the C oracle), so inputs are views -- no pickling, no fork."""
    import numpy as np

    from oracle import wdx_oracle as orc

    sig, off, a_s, a_e, refs, K = job
    p = orc.SegParams(barcode_num_events=K)
    fpt, dwell, stats, status = orc.fingerprint_packed(sig, off, a_s, a_e, p)
    ok = status == 0
    D = np.full((status.size, refs.shape[0]), np.nan, dtype=np.float32)
    D[ok] = orc.dtw_matrix(fpt[ok], refs, WINDOW, PENALTY)
    call = np.full(status.size, -1, dtype=np.int32)
    call[ok] = orc.argmin_rows(D[ok])
    return call, status, D, fpt, dwell


def gate_chunks(n_reads, max_reads, chunk):
    """Where the parity gate looks: chunks of `chunk` reads at evenly spaced offsets over the WHOLE shard -- the first
    and the last chunk always, then the remaining positions in bit-reversed (van der Corput) order, so that a run cut
    short by its time budget has still looked at both ends and the middle: every launch slice of the pass (a pass over
    more than 2^23 reads is cut into equal slices with block_base != 0) and the tail are compared, not just the front."""
    n_reads, chunk = int(n_reads), int(chunk)
    if n_reads < 2 * chunk:
        return [(0, n_reads)]
    k = max(2, min((min(max_reads, n_reads) + chunk - 1) // chunk, n_reads // chunk))
    starts = [round(i * (n_reads - chunk) / (k - 1)) for i in range(k)]
    bits = max(1, (k - 1).bit_length())
    order = sorted(range(k), key=lambda i: (0 if i in (0, k - 1) else 1, int(format(i, "0%db" % bits)[::-1], 2)))
    return [(starts[i], starts[i] + chunk) for i in order]


def cpu_baseline_and_parity(eng, sig, off, a_s, a_e, res, refs, n_reads, budget_s, max_reads):
    """Walks 32 768-read chunks spread over the whole shard (`gate_chunks`) until the time budget or `max_reads`
    is reached: oracle on all host cores (timed), GPU results of the same reads compared bit for bit --
    status, call, float32 distances, float64 fingerprints and the int64 dwell times (= the change-points).
    `res` is the result of the timed pass over the WHOLE shard, so a chunk at read 9.9 M checks what the second
    launch slice of that pass wrote.  Returns the cpu_baseline object and the parity record."""
    import numpy as np
    import torch
    from concurrent.futures import ThreadPoolExecutor

    cores = effective_cores()
    chunk = 32768
    mb = 1000 if cores * 1000 <= chunk else max(64, chunk // cores)
    t_cpu = 0.0
    n_done = 0
    single = None
    bad = {"status": 0, "call": 0, "dist": 0, "fpt": 0, "dwell": 0}
    max_rel = 0.0
    spans = []
    t_start = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        list(ex.map(lambda i: i, range(cores)))  # threads exist before the clock starts
        for lo, hi in gate_chunks(n_reads, max_reads, chunk):
            if n_done >= min(max_reads, n_reads):
                break
            if budget_s > 0 and n_done > 0 and (time.perf_counter() - t_start) > budget_s:
                break
            hi = min(hi, lo + min(max_reads, n_reads) - n_done)
            o = off[lo:hi + 1].cpu().numpy()
            sig_h = sig[int(o[0]):int(o[-1])].cpu().numpy()
            as_h, ae_h = a_s[lo:hi].cpu().numpy(), a_e[lo:hi].cpu().numpy()
            # GPU side of the gate: fingerprint + dwell of the same reads (the fused call keeps neither)
            g_fpt, g_dwell, _, g_status = eng.fingerprint(sig[int(o[0]):int(o[-1])], a_s[lo:hi], a_e[lo:hi],
                                                          offsets=off[lo:hi + 1] - off[lo], max_len=int((o[1:] - o[:-1]).max()))
            g_fpt, g_dwell, g_status = g_fpt.cpu().numpy(), g_dwell.cpu().numpy(), g_status.cpu().numpy()
            jobs = []
            for a in range(0, hi - lo, mb):
                b = min(hi - lo, a + mb)
                oo = o[a:b + 1]
                jobs.append((sig_h[oo[0] - o[0]:oo[-1] - o[0]], oo - oo[0], as_h[a:b], ae_h[a:b], refs, K_FPT))
            if single is None:  # 1-core figure: one minibatch in this thread
                t0 = time.perf_counter()
                _oracle_minibatch(jobs[0])
                single = (jobs[0][1].size - 1) / (time.perf_counter() - t0)
            t0 = time.perf_counter()
            out = list(ex.map(_oracle_minibatch, jobs))
            t_cpu += time.perf_counter() - t0
            call = np.concatenate([r[0] for r in out])
            status = np.concatenate([r[1] for r in out])
            D = np.concatenate([r[2] for r in out])
            fpt = np.concatenate([r[3] for r in out])
            dwell = np.concatenate([r[4] for r in out])
            okm = status == 0
            g_call = res.call[lo:hi].cpu().numpy()
            g_dist = res.dist[lo:hi].cpu().numpy()
            g_st = res.status[lo:hi].cpu().numpy()
            bad["status"] += int((status != g_st).sum() + (status != g_status).sum())
            bad["call"] += int((call != g_call).sum())
            bad["dist"] += int((D[okm].view(np.uint32) != g_dist[okm].view(np.uint32)).any(axis=1).sum())
            bad["fpt"] += int((fpt[okm].view(np.uint64) != g_fpt[okm].view(np.uint64)).any(axis=1).sum())
            bad["dwell"] += int((dwell[okm] != g_dwell[okm]).any(axis=1).sum())
            with np.errstate(invalid="ignore", divide="ignore"):
                rel = np.abs(D[okm].astype(np.float64) - g_dist[okm]) / np.abs(D[okm].astype(np.float64))
            if rel.size:
                max_rel = max(max_rel, float(np.nanmax(rel)))
            n_done += hi - lo
            spans.append((int(lo), int(hi)))
    parity_ok = not any(bad.values())
    cpu = {
        "value": n_done / t_cpu if t_cpu > 0 else None,
        "unit": "reads/s",
        "cores": cores,
        "logical_cpus": os.cpu_count(),
        "cpu_model": cpu_model(),
        "kind": "port",
        "sample": ("%d reads of the same workload in 32 768-read chunks spread evenly over the whole shard; oracle/wdx_oracle.c (C restatement of sig_proc.py:394-605 + "
                   "dtaidistance's banded DTW) driven as %d-read minibatches like file_proc.py:418-450, one worker thread "
                   "per host core on shared buffers (no IPC); dtaidistance itself: see `dtaidistance`" % (n_done, mb)),
        "single_core_value": single,
        "parallel_efficiency": (n_done / t_cpu) / (single * cores) if (single and t_cpu > 0) else None,
        "cpu_seconds": t_cpu,
    }
    parity = {
        "reads_checked": n_done,
        "read_ranges": sorted(spans),      # [lo, hi) of every chunk compared: spans the shard, both ends included
        "shard_reads": int(n_reads),
        "checked": "status, call (argmin), float32 distances, float64 fingerprints, int64 dwell (change-points): bitwise",
        "mismatching_reads": bad,
        "max_rel_dist_err": max_rel,
        "ok": parity_ok,
    }
    return cpu, parity


def try_dtaidistance(eng, refs, res_fpt_sample):
    """if the genuine reference library is importable on this box, compare it with the HIP
    matrix on >= 1e4 pairs (R1 and R2 shapes) and time it; otherwise record null (never an error)."""
    import numpy as np

    try:
        import dtaidistance
        from dtaidistance import dtw
    except Exception:  # noqa: BLE001
        return None
    import torch

    from warpdemux_amd import parallel_distances as pdist

    out = {"version": getattr(dtaidistance, "__version__", "?")}
    rng = np.random.default_rng(7)
    for name, X, Y in (("r1", res_fpt_sample[:1000], refs), ("r2", rng.normal(size=(64, 25)), rng.normal(size=(851, 25)))):
        X = np.ascontiguousarray(X, dtype=np.float64)
        Y = np.ascontiguousarray(Y, dtype=np.float64)
        stack = np.vstack([X, Y])
        nX = X.shape[0]
        t0 = time.perf_counter()
        ref = dtw.distance_matrix(stack, block=((0, nX), (nX, nX + Y.shape[0])), parallel=False, use_c=True,
                                  only_triu=True, window=WINDOW, penalty=PENALTY)[:nX, nX:].astype(np.float32)
        dt = time.perf_counter() - t0
        mine = pdist.distance_matrix_to(X, Y, window=WINDOW, penalty=PENALTY, n_jobs=1)
        with np.errstate(invalid="ignore", divide="ignore"):
            rel = float(np.nanmax(np.abs(mine.astype(np.float64) - ref) / np.abs(ref)))
        out[name] = {"pairs": int(mine.size), "bit_identical": bool(np.array_equal(mine, ref)), "max_rel": rel,
                     "argmin_identical": bool(np.array_equal(mine.argmin(1), ref.argmin(1))),
                     "reference_pairs_per_s_1core": mine.size / dt}
    return out


# ----------------------------------------------------------------------------------------------------
# secondary regimes (N=1 only): driver-timed, each with a parity check on its own sample
# ----------------------------------------------------------------------------------------------------
def _oracle_dtw_threads(X, Y, budget_s=3.0, rows_per_job=32):
    """oracle DTW of as many leading rows of X as fit the time budget, on all cores -> (D, rows)"""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor

    from oracle import wdx_oracle as orc

    cores = effective_cores()
    outs, done, t0 = [], 0, time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        while done < X.shape[0] and (done == 0 or time.perf_counter() - t0 < budget_s):
            hi = min(X.shape[0], done + rows_per_job * cores)
            parts = [X[a:min(hi, a + rows_per_job)] for a in range(done, hi, rows_per_job)]
            outs += list(ex.map(lambda x: orc.dtw_matrix(x, Y, WINDOW, PENALTY), parts))
            done = hi
    return np.concatenate(outs), done


def reference_model_parity(device):
    """The parity half of the shipped-model leg against a GENUINE reference model: fixture g6b holds the numbers of the
    reference's WDX10_rna004_v1_0.joblib (2 601 x 25 training fingerprints, 11 classes, thresholds) and what the
    reference's own DTW_SVM.predict (models/dtw_svm.py:54-98) returned for 256 query fingerprints when it was run in
    the build container (tests/golden/make_golden_svm.py).  The engine's DTW_SVM on the same queries: probabilities
    within 1e-5, labels identical wherever the reference's margin is not within 1e-4 of a tie or a threshold."""
    import numpy as np

    from warpdemux_amd.models import DTW_SVM

    path = os.path.join(ROOT, "tests", "golden", "g6b_dtw_svm_wdx10.npz")
    if not os.path.exists(path):
        return None
    g = np.load(path)
    label_mapper = {int(k): int(v) for k, v in zip(g["label_keys"], g["label_vals"])}
    m = DTW_SVM(g["X_train"], g["n_support"], g["support"], g["dual_coef"], -g["intercept"], g["probA"], g["probB"],
                label_mapper, g["thresholds"], window=int(g["window"]), penalty=float(g["penalty"]),
                gamma=float(g["gamma"]), pwr_dist=int(g["pwr_dist"]), block_size=int(g["block_size"]), device=device)
    pred, prob = m.predict(g["Xq"])
    err = float(np.abs(prob - g["y_prob"]).max())
    srt = np.sort(g["y_prob"], axis=1)
    conf = srt[:, -1] - srt[:, -2]
    safe = (conf > 1e-4) & (np.abs(conf - g["thresholds"][np.argmax(g["y_prob"], axis=1)]) > 1e-4)
    same = bool(np.array_equal(pred[safe], g["y_pred"][safe]))
    return {"model": "WDX10_rna004_v1_0 (reference model file; fixture g6b = the reference's DTW_SVM.predict run on it)",
            "queries": int(prob.shape[0]), "max_abs_prob_err": err, "labels_compared": int(safe.sum()),
            "labels_identical": same, "parity": bool(err <= 1e-5 and same), "parity_tolerance": 1e-5}


def secondary_shipped_model_e2e(device):
    import numpy as np
    import torch

    from oracle import wdx_oracle as orc
    from warpdemux_amd import parallel_distances as pdist
    from warpdemux_amd import sig_proc, synth
    from warpdemux_amd.engine import DemuxEngine
    from warpdemux_amd.models import DTW_SVM

    out = {}
    tdev = torch.device("cuda", device)

    def sync():
        torch.cuda.synchronize(tdev)

    # ---- shipped_model_e2e: the shipped models' WHOLE path in one device-resident call (wdx_demux_svm_dev): raw rows ->
    # fingerprint (K = 25) -> DTW against a WDX10-shaped training set (2 601 x 25-pt rows, 11 classes) -> SVM tail, the
    # distance matrix in row blocks that stay in the memory-side cache.  The model is TRAINED here on fingerprints of
    # synthetic reads of 11 barcodes (scikit-learn SVC on exp(-D), D from the engine), so the calls mean something:
    # accuracy on the 10^5 held-out reads is reported next to the throughput. ---------------------------------------------
    try:
        from sklearn.svm import SVC

        kcls, n_train, L, nq = 11, 2601, 25, 100_000
        specm = synth.SynthSpec(n_barcodes=kcls)
        pm = sig_proc.SegParams(barcode_num_events=L)
        eng0 = DemuxEngine(np.zeros((1, L)), WINDOW, PENALTY, pm, device=device)
        sgt, oft, st0, et0, bct, mlt = eng0.synth_packed(specm, 0, n_train + 64)
        ftr, _, _, sttr = eng0.fingerprint(sgt, st0, et0, offsets=oft, max_len=mlt)
        okt = (sttr == 0).cpu().numpy()
        Xtr = ftr.cpu().numpy()[okt][:n_train]
        ytr = bct.cpu().numpy()[okt][:n_train]
        eng0.close()
        Dtr = pdist.parallel_distance_matrix(Xtr, block_size=1000, n_jobs=1, window=WINDOW, penalty=PENALTY)
        svc = SVC(kernel="precomputed", probability=True, random_state=0).fit(np.exp(-Dtr.astype(np.float64)), ytr)
        sp = list(orc.svm_params(svc))
        n_sv_trained = int(sp[1].size)
        # the shipped models' shape: EVERY training row is a support vector (tests/helpers/kkt.py).  This easy synthetic
        # problem leaves most rows outside the margin, so the remaining rows of each class are appended as support vectors
        # with zero coefficients: the same decision function (parity against scikit-learn still holds), the shipped
        # models' amount of work -- the fused path only walks support vectors.
        n_support, support, dual_coef = sp[0], sp[1], sp[2]
        st = np.concatenate([[0], np.cumsum(n_support)])
        sup2, coef2, ns2 = [], [], []
        for c in range(kcls):
            own = support[st[c]:st[c + 1]]
            extra = np.setdiff1d(np.nonzero(ytr == svc.classes_[c])[0], own).astype(support.dtype)
            sup2.append(np.concatenate([own, extra]))
            coef2.append(np.concatenate([dual_coef[:, st[c]:st[c + 1]], np.zeros((kcls - 1, extra.size))], axis=1))
            ns2.append(own.size + extra.size)
        sp[0], sp[1], sp[2] = np.array(ns2, dtype=n_support.dtype), np.concatenate(sup2), np.ascontiguousarray(np.concatenate(coef2, axis=1))
        assert sp[1].size == n_train
        model = DTW_SVM(Xtr, *sp[:6], {i: i for i in range(kcls)}, None, WINDOW, PENALTY, block_size=1000, device=device)
        engm = DemuxEngine(Xtr, WINDOW, PENALTY, pm, device=device)
        engm.set_svm(model)
        sgq, ofq, sq, eq, bcq, mlq = engm.synth_packed(specm, 1_000_000, nq)
        res = engm.demux_svm(sgq, sq, eq, offsets=ofq, max_len=mlq)          # allocates outputs, workspaces, the block buffer
        for _ in range(2):
            engm.demux_svm(sgq, sq, eq, offsets=ofq, max_len=mlq, out=res)
        sync()
        walls = []
        engm.kernel_time_reset()
        engm.kernel_timing(True)
        for _ in range(5):
            sync()
            t0 = time.perf_counter()
            engm.demux_svm(sgq, sq, eq, offsets=ofq, max_len=mlq, out=res)
            sync()
            walls.append(time.perf_counter() - t0)
        engm.kernel_timing(False)
        from warpdemux_amd import _lib as _l
        kms = {nm: engm.kernel_time(kid)[0] / 5 for nm, kid in (("fingerprint", _l.K_FINGERPRINT), ("dtw", _l.K_DTW),
                                                                 ("transpose", _l.K_TRANSPOSE), ("svm_tail", _l.K_SVM))}
        # the same path as three separate calls with the whole (n, nY) matrix in HBM, for comparison
        dfull = torch.empty((nq, n_train), dtype=torch.float32, device=engm.tdev)
        walls3 = []
        for rep in range(4):
            sync()
            t0 = time.perf_counter()
            f3, _, _, s3 = engm.fingerprint(sgq, sq, eq, offsets=ofq, max_len=mlq)
            engm.dtw(f3, want_argmin=False, out=(dfull, None))
            p3 = engm.svm_predict(dfull)
            sync()
            if rep:
                walls3.append(time.perf_counter() - t0)
        prob, pred, conf, status = (t.cpu().numpy() for t in res[:4])
        same3 = bool(np.abs(p3[0].cpu().numpy()[status == 0] - prob[status == 0]).max() <= 1e-12)   # (another summation order)
        ns = 512
        o_h = ofq[: ns + 1].cpu().numpy()
        ofp, _, _, ost = orc.fingerprint_packed(sgq[: int(o_h[-1])].cpu().numpy(), o_h, sq[:ns].cpu().numpy(), eq[:ns].cpu().numpy(),
                                                orc.SegParams(barcode_num_events=L))
        oko = ost == 0
        pref = svc.predict_proba(np.exp(-orc.dtw_matrix(ofp[oko], Xtr, WINDOW, PENALTY)))
        err = float(np.abs(prob[:ns][oko] - pref).max())
        okq = status == 0
        dt = sum(walls) / len(walls)
        cells = 515.0 * n_train   # cells of the (25-pt, window 15) band x references, per read
        out["shipped_model_e2e"] = {
            "workload": "wdx_demux_svm_dev on 100 000 device-resident synthetic reads: fingerprint (K = 25) -> DTW vs 2 601 x 25-pt "
                        "training rows (WDX10 shape, 11 classes; model trained here on synthetic fingerprints, every row a support "
                        "vector like the shipped models) with the SVM decision sums in the DTW kernel's epilogue -> sigmoids, "
                        "coupling, process_probs; no distance matrix",
            "support_vectors": int(sp[1].size), "support_vectors_with_nonzero_coefficients": n_sv_trained,
            "reads_per_s": nq / dt, "ms": dt * 1e3, "ms_min": min(walls) * 1e3, "ms_max": max(walls) * 1e3, "reps": len(walls),
            "kernels_ms": kms,      # HIP events around the library's launches
            "three_separate_calls": {"reads_per_s": nq / (sum(walls3) / len(walls3)), "ms": 1e3 * sum(walls3) / len(walls3),
                                     "same_probabilities_1e-12": same3},
            "useful_cell_updates_per_s": nq * cells / dt,
            "accuracy_on_ok_reads": float((pred[okq] == bcq.cpu().numpy()[okq]).mean()), "ok_reads": int(okq.sum()),
            "parity": bool(err <= 1e-5 and same3 and np.array_equal(ost, status[:ns])), "max_abs_prob_err": err,
            "parity_tolerance": 1e-5, "parity_reads": ns}
        del dfull, res
        engm.close()
        # ---- the TIMED half on the genuine model: the arrays of the reference's WDX10_rna004_v1_0.joblib (fixture g6b: 2 601 x
        # 25 training fingerprints, every one a support vector, 11 classes, thresholds) resident instead of the model trained
        # above -- same call, same reads; the model trained here stays as the accuracy leg (the synthetic reads mean nothing
        # to the real model's barcodes).  Checked against the oracle (fingerprint -> DTW -> libsvm restatement) on 512 reads.
        g6b = os.path.join(ROOT, "tests", "golden", "g6b_dtw_svm_wdx10.npz")
        if os.path.exists(g6b):
            g = np.load(g6b)
            lm = {int(k): int(v) for k, v in zip(g["label_keys"], g["label_vals"])}
            mr = DTW_SVM(g["X_train"], g["n_support"], g["support"], g["dual_coef"], -g["intercept"], g["probA"], g["probB"], lm,
                         g["thresholds"], window=int(g["window"]), penalty=float(g["penalty"]), gamma=float(g["gamma"]),
                         pwr_dist=int(g["pwr_dist"]), block_size=int(g["block_size"]), device=device)
            engr = DemuxEngine(np.ascontiguousarray(g["X_train"], dtype=np.float64), int(g["window"]), float(g["penalty"]), pm, device=device)
            engr.set_svm(mr)
            resr = engr.demux_svm(sgq, sq, eq, offsets=ofq, max_len=mlq)
            for _ in range(2):
                engr.demux_svm(sgq, sq, eq, offsets=ofq, max_len=mlq, out=resr)
            sync()
            wr = []
            engr.kernel_time_reset()
            engr.kernel_timing(True)
            for _ in range(5):
                sync()
                t0 = time.perf_counter()
                engr.demux_svm(sgq, sq, eq, offsets=ofq, max_len=mlq, out=resr)
                sync()
                wr.append(time.perf_counter() - t0)
            engr.kernel_timing(False)
            kmr = {nm: engr.kernel_time(kid)[0] / 5 for nm, kid in (("fingerprint", _l.K_FINGERPRINT), ("dtw", _l.K_DTW),
                                                                     ("transpose", _l.K_TRANSPOSE), ("svm_tail", _l.K_SVM))}
            probr, predr, confr, statusr = (t.cpu().numpy() for t in resr[:4])
            Dr = orc.dtw_matrix(ofp[oko], np.ascontiguousarray(g["X_train"], dtype=np.float64), int(g["window"]), float(g["penalty"]))
            Kr = np.exp(-float(g["gamma"]) * np.power(Dr, int(g["pwr_dist"])))
            pr = orc.svm_predict_proba(Kr, g["n_support"].astype(np.int32), g["support"].astype(np.int32), g["dual_coef"], -g["intercept"],
                                       g["probA"], g["probB"])
            errr = float(np.abs(probr[:ns][oko] - pr).max())
            dtr = sum(wr) / len(wr)
            synth_leg = {k: out["shipped_model_e2e"][k] for k in ("reads_per_s", "ms", "ms_min", "ms_max", "kernels_ms", "support_vectors",
                                                                  "support_vectors_with_nonzero_coefficients", "accuracy_on_ok_reads",
                                                                  "max_abs_prob_err", "useful_cell_updates_per_s")}
            out["shipped_model_e2e"].update({
                "workload": "wdx_demux_svm_dev on 100 000 device-resident synthetic reads against the REFERENCE's WDX10_rna004_v1_0 model (fixture "
                            "g6b: 2 601 x 25-pt training fingerprints, every row a support vector, 11 classes, thresholds): fingerprint (K = 25) "
                            "-> DTW with the SVM decision sums in the kernel's epilogue -> sigmoids, coupling, process_probs; no distance "
                            "matrix.  `model_trained_here` = the same call on a model trained in this run on synthetic fingerprints (the "
                            "accuracy leg: the synthetic barcodes mean nothing to the real model)",
                "model": "WDX10_rna004_v1_0 (reference model file, tests/golden/g6b_dtw_svm_wdx10.npz)",
                "reads_per_s": nq / dtr, "ms": dtr * 1e3, "ms_min": min(wr) * 1e3, "ms_max": max(wr) * 1e3, "reps": len(wr),
                "kernels_ms": kmr, "support_vectors": int(g["support"].size), "useful_cell_updates_per_s": nq * 515.0 * g["X_train"].shape[0] / dtr,
                "max_abs_prob_err": errr, "ok_reads": int((statusr == 0).sum()),
                "parity": bool(out["shipped_model_e2e"]["parity"] and errr <= 1e-5 and np.array_equal(ost, statusr[:ns])),
                "model_trained_here": synth_leg})
            out["shipped_model_e2e"].pop("support_vectors_with_nonzero_coefficients", None)
            out["shipped_model_e2e"].pop("accuracy_on_ok_reads", None)
            del resr
            engr.close()
        del sgq
        out["shipped_model_e2e"]["reference_model"] = reference_model_parity(device)
        if out["shipped_model_e2e"]["reference_model"] is not None:
            out["shipped_model_e2e"]["parity"] = bool(out["shipped_model_e2e"]["parity"] and
                                                      out["shipped_model_e2e"]["reference_model"]["parity"])
    except ImportError:
        out["shipped_model_e2e"] = None
    return out["shipped_model_e2e"]


def secondary_regimes(device):
    import numpy as np
    import torch

    from oracle import wdx_oracle as orc
    from warpdemux_amd import _lib
    from warpdemux_amd import parallel_distances as pdist
    from warpdemux_amd import sig_proc, synth
    from warpdemux_amd.engine import DemuxEngine
    from warpdemux_amd.live import LiveDemux
    from warpdemux_amd.models import DTW_SVM

    out = {}
    rng = np.random.default_rng(0)
    tdev = torch.device("cuda", device)
    t_leg = [time.perf_counter()]

    def leg_done(name):   # wall time per leg on stderr: a slow box explains itself
        now = time.perf_counter()
        print("[bench] secondary.%s: %.1f s" % (name, now - t_leg[0]), file=sys.stderr, flush=True)
        t_leg[0] = now

    def sync():
        torch.cuda.synchronize(tdev)

    # ---- r2: shipped-model DTW shape (L=25, window 15), nY = 2601 (WDX10's training set size) ----------
    nY, nX = 2601, 100_000
    Y = rng.normal(size=(nY, 25))
    eng = DemuxEngine(Y, WINDOW, PENALTY, sig_proc.SegParams(barcode_num_events=25), device=device)
    Xh = rng.normal(size=(nX, 25))
    X = torch.from_numpy(Xh).to(tdev)
    # outputs allocated ONCE and touched before the clock starts: no allocation, no first touch inside the timed
    # loop (round 2's loop allocated a fresh 1 GB matrix per repetition right after ~190 GB had gone back to the
    # driver, and the driver's run of it came out 81x slower than the builder's)
    d = torch.zeros((nX, nY), dtype=torch.float32, device=tdev)
    am = torch.zeros(nX, dtype=torch.int32, device=tdev)
    for _ in range(3):
        eng.dtw(X, out=(d, am))
    sync()
    reps = 10
    eng.kernel_time_reset()
    eng.kernel_timing(True)
    walls = []
    for _ in range(reps):
        sync()
        t0 = time.perf_counter()
        eng.dtw(X, out=(d, am))
        sync()
        walls.append(time.perf_counter() - t0)
    eng.kernel_timing(False)
    k_ms, k_n = eng.kernel_time(_lib.K_DTW)
    dt = sum(walls) / reps
    Dref, rows = _oracle_dtw_threads(Xh, Y, budget_s=3.0)
    dh = d[:rows].cpu().numpy()
    out["r2"] = {"workload": "device-resident DTW, 100 000 reads x 2601 refs x 25 points (window 15, penalty 0.1)",
                 "reads_per_s": nX / dt, "gcups": nX * nY * CELLS_25 / dt / 1e9, "ms": dt * 1e3,
                 "reps": reps, "ms_min": min(walls) * 1e3, "ms_max": max(walls) * 1e3,
                 "hip_event_ms": k_ms / reps, "hip_event_launches_per_call": k_n / reps,
                 "timing": "wall clock per call, outputs preallocated; hip_event_ms = the library's own events around its DTW launches, per call",
                 "parity": bool(np.array_equal(dh.view(np.uint32), Dref.view(np.uint32))
                                and np.array_equal(am[:rows].cpu().numpy(), orc.argmin_rows(Dref))),
                 "parity_pairs": int(rows * nY)}
    del X, d, am
    eng.close()
    leg_done("r2")

    # ---- host_minibatch: what file_proc.py:418-450 would pass -- 1000 x 10 000 float32 rows, PCIe included -----
    spec = synth.SynthSpec(n_barcodes=N_BARCODES)
    mb, a_s, a_e, _ = synth.generate_minibatch(spec, 0, 1000, 10000)
    p110 = sig_proc.SegParams(barcode_num_events=K_FPT)
    Y10 = rng.normal(size=(N_BARCODES, K_FPT))
    for _ in range(2):
        fb = sig_proc.fingerprint_batch(mb, a_s, a_e, p110, device=device)
        Dm = pdist.distance_matrix_to(fb.fpt[fb.status == 0], Y10, window=WINDOW, penalty=PENALTY, n_jobs=1)
    reps = 20
    walls = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fb = sig_proc.fingerprint_batch(mb, a_s, a_e, p110, device=device)
        Dm = pdist.distance_matrix_to(fb.fpt[fb.status == 0], Y10, window=WINDOW, penalty=PENALTY, n_jobs=1)
        walls.append(time.perf_counter() - t0)
    dt = sum(walls) / reps
    ofpt, odw, ost, ostatus = orc.fingerprint_batch(mb, a_s, a_e, orc.SegParams(barcode_num_events=K_FPT))
    ook = ostatus == 0
    oD = orc.dtw_matrix(ofpt[ook], Y10, WINDOW, PENALTY)
    out["host_minibatch"] = {
        "workload": "fingerprint_batch + distance_matrix_to on one 1000 x 10 000 float32 minibatch (host buffers, PCIe included), 110-pt x 10 refs",
        "reads_per_s": 1000 / dt, "ms": dt * 1e3, "reps": reps, "ms_min": min(walls) * 1e3, "ms_max": max(walls) * 1e3,
        "parity": bool(np.array_equal(fb.status, ostatus) and np.array_equal(fb.fpt[ook], ofpt[ook])
                       and np.array_equal(fb.dwell[ook], odw[ook]) and np.array_equal(Dm.view(np.uint32), oD.view(np.uint32)))}

    leg_done("host_minibatch")
    # ---- host_workers: the reference's real calling pattern -- P forked workers x 1000-read minibatches sharing this GPU
    # (file_proc.py:380-454, 1197-1243).  Fresh interpreters (tools/host_workers.py forks before any GPU call); this
    # process keeps its context but is idle meanwhile.
    hw = {}
    t_hw = time.perf_counter()
    # (jitter 2900: rows carry whole reads, adapter_start ~ U{100..3000} per read -- page-locked minibatches then take the
    # packed staging, only the windows cross the bus; feeder: ONE GPU-facing process, P producer processes fill a shared
    # page-locked ring -- always with the producers' own 40 MB fill per minibatch, like "pipe refill")
    # (most telling first: the leg stops launching once 20 s are spent and says which configurations it skipped)
    for mode, refill, P, jit in (("feeder", False, 16, 0), ("sync", False, 16, 0), ("sync", False, 4, 0), ("pipe", False, 1, 2900),
                                 ("feeder", True, 16, 2900), ("feeder_full", False, 16, 0), ("feeder", False, 4, 0)):
        key = "%s%s%s_P%d" % (mode, "_refill" if refill else "", "_jitter" if jit else "", P)
        cmd = [sys.executable, os.path.join(ROOT, "tools", "host_workers.py"), "--workers", str(P), "--mode", mode,
               "--seconds", "1.5", "--jitter", str(jit)] + (["--refill"] if refill else [])
        if time.perf_counter() - t_hw > 20.0:   # the leg's budget: the rest is skipped, and says so
            hw[key] = {"skipped": "leg budget of 20 s spent"}
            continue
        try:
            pr = subprocess.run(cmd, capture_output=True, text=True, timeout=120, cwd=ROOT)
            rec = json.loads([ln for ln in pr.stdout.splitlines() if ln.startswith("{")][-1])
        except Exception as e:  # noqa: BLE001
            rec = {"error": f"{type(e).__name__}: {e}"}
        hw[key] = rec
    good = [v for v in hw.values() if "reads_per_s" in v]
    out["host_workers"] = {
        "workload": "P forked worker processes on ONE GPU, each driving 1000 x 10 000 float32 minibatches (host buffers, PCIe "
                    "included) through fingerprint + DTW (110-pt x 10 refs) + call; sync = sig_proc.demux_batch on a pageable "
                    "array, pipe = MinibatchPipeline (page-locked buffers, wdx_demux_submit / wdx_demux_wait); refill = the "
                    "worker copies a fresh minibatch into the buffer before every call; jitter = adapter_start ~ U{100..3000} per "
                    "read (rows carry whole reads; page-locked rows then go through the packed staging); feeder = warpdemux_amd.feeder.Feeder: one "
                    "GPU-facing process serving a shared-memory ring of 16 minibatch slots (wdx_feeder_serve; at most 8 in flight on the device), the P "
                    "workers call feeder.demux_batch like the sync mode calls sig_proc.demux_batch (wdx_feeder_run: no context, no HIP call, only "
                    "the adapter windows are copied into the ring); feeder_full = feeder.detect_and_predict: what the reference's worker needs "
                    "from its minibatch (file_proc.py:380-454) -- fingerprints, dwell times, six statistics AND DTW_SVM.predict on the "
                    "reference's WDX10_rna004_v1_0 model (2 601 x 25-pt training rows, 11 classes) -- from one pass",
        "host_cpus": effective_cores(), **hw,
        "best_reads_per_s": max((v["reads_per_s"] for v in good), default=None),
        "parity": bool(good) and all(v.get("parity") is True for v in hw.values() if "skipped" not in v)}

    leg_done("host_workers")
    # ---- other shipped parameter triples on the fast kernels (device-resident fingerprint stage) + the tRNA config's
    # consensus-refinement flow (host batch; HIP-event time of the fingerprint launches) ------------------------------
    trip = {}
    # (the third leg: the RNA002 triple on RNA002-LENGTH windows -- synth.dwell_table(2.5): dwell 15 + geometric(mean 71.75),
    # mean window ~11.5 k samples, 8.3 k .. 15.6 k -- which is what that triple is for; ASSUMPTION stated in synth.py.
    # These windows take the streaming fast kernel, fingerprint_fast_stream_kernel)
    for name, (E, d, W), n_t, dscale in (("rna002_110_15_30", (110, 15, 30), 262_144, 1.0), ("trna_120_9_18", (120, 9, 18), 262_144, 1.0),
                                         ("rna002_110_15_30_on_rna002_length_windows", (110, 15, 30), 65_536, 2.5)):
        pt = sig_proc.SegParams(num_events=E, min_obs_per_base=d, running_stat_width=W, barcode_num_events=25)
        engt = DemuxEngine(np.zeros((N_BARCODES, 25)), WINDOW, PENALTY, pt, device=device)
        spec_t = spec if dscale == 1.0 else synth.SynthSpec(n_barcodes=N_BARCODES, dwell_scale=dscale)
        sg, of, s0, e0, _, mlen = engt.synth_packed(spec_t, 0, n_t)
        for _ in range(2):
            g = engt.fingerprint(sg, s0, e0, offsets=of, max_len=mlen)
        sync()
        walls = []
        for _ in range(5):
            sync()
            t0 = time.perf_counter()
            g = engt.fingerprint(sg, s0, e0, offsets=of, max_len=mlen)
            sync()
            walls.append(time.perf_counter() - t0)
        ns_ = 4096
        o_h = of[:ns_ + 1].cpu().numpy()
        ofp, odw, ost, ostat = orc.fingerprint_packed(sg[:int(o_h[-1])].cpu().numpy(), o_h, s0[:ns_].cpu().numpy(), e0[:ns_].cpu().numpy(),
                                                      orc.SegParams(num_events=E, min_obs_per_base=d, running_stat_width=W, barcode_num_events=25))
        okt = ostat == 0
        rps = n_t / (sum(walls) / len(walls))
        bpr = 4.0 * float(of[-1].item()) / n_t + 8.0 * 25 + 4.0    # algorithmic bytes per read of the fingerprint stage (K = 25)
        trip[name] = {"reads_per_s": rps, "ms": 1e3 * sum(walls) / len(walls), "ms_min": 1e3 * min(walls),
                      "reads": n_t, "mean_window_samples": float(of[-1].item()) / n_t, "max_window_samples": int(mlen),
                      "roofline": {"bound": "hbm", "bytes_per_read": bpr, "achieved": rps * bpr / 1e9, "unit": "GB/s",
                                   "peak": HBM_PEAK_GBS, "frac": rps * bpr / 1e9 / HBM_PEAK_GBS},
                      "parity_reads": ns_,
                      "parity": bool(np.array_equal(g[3][:ns_].cpu().numpy(), ostat) and
                                     np.array_equal(g[0][:ns_].cpu().numpy()[okt].view(np.uint64), ofp[okt].view(np.uint64)) and
                                     np.array_equal(g[1][:ns_].cpu().numpy()[okt], odw[okt]) and
                                     np.array_equal(g[2][:ns_].cpu().numpy()[okt].view(np.uint64), ost[okt].view(np.uint64)))}
        del sg, of, s0, e0, g
        engt.close()
    try:
        cons = np.load(os.path.join(ROOT, "tests", "golden", "g8_refine.npz"))["consensus"]
        nr = 8192
        rows_r = []
        for i in range(nr):
            lv = np.concatenate([rng.normal(0, 1, int(rng.integers(2, 34))), cons, rng.normal(0, 1, 30)]) * 12.0 + 85.0
            dwl = rng.integers(12, 60, lv.size)
            rows_r.append((np.repeat(lv, dwl) + rng.normal(0, 1.5, int(dwl.sum()))).astype(np.float32))
        st_r = max(x.size for x in rows_r)
        mbr = np.full((nr, st_r), np.nan, dtype=np.float32)
        for i, x in enumerate(rows_r):
            mbr[i, :x.size] = x
        asr = np.full(nr, 100, dtype=np.int32)
        aer = np.array([x.size - 100 for x in rows_r], dtype=np.int32)
        kwr = dict(min_obs_per_base=9, running_stat_width=18, num_events=120, barcode_num_events=25)
        hp, hr = sig_proc.SegParams(**kwr), sig_proc.RefineParams(query=cons, barcode_segm_events=25, barcode_keep_events=25)
        cdef = _lib.default_context(device)
        Ll = _lib.load()
        fr = sig_proc.fingerprint_refine_batch(mbr, asr, aer, hp, hr, device=device)
        _lib.check(Ll.wdx_kernel_time_reset(cdef.handle))
        _lib.check(Ll.wdx_kernel_timing(cdef.handle, 1))
        for _ in range(3):
            fr = sig_proc.fingerprint_refine_batch(mbr, asr, aer, hp, hr, device=device)
        _lib.check(Ll.wdx_kernel_timing(cdef.handle, 0))
        import ctypes as C
        ms_, k_ = C.c_double(0), C.c_int64(0)
        _lib.check(Ll.wdx_kernel_time(cdef.handle, _lib.K_FINGERPRINT, C.byref(ms_), C.byref(k_)))
        ns_ = 1024
        ofp, odw, ostt, oidx, ostat = orc.fingerprint_refine_batch(
            mbr[:ns_], asr[:ns_], aer[:ns_], orc.SegParams(**kwr),
            orc.RefineParams(query=cons, barcode_segm_events=25, barcode_keep_events=25))
        okt = ostat == 0
        # the same flow device-resident at 8x the reads (wdx_fingerprint_refine_dev on the 8 192 rows repeated)
        engr = DemuxEngine(np.zeros((N_BARCODES, 25)), WINDOW, PENALTY, hp, device=device)
        rep = 8
        dmb = torch.from_numpy(mbr).to(tdev).repeat(rep, 1)
        das, dae = torch.from_numpy(asr).to(tdev).repeat(rep), torch.from_numpy(aer).to(tdev).repeat(rep)
        for _ in range(2):
            gdev = engr.fingerprint_refine(dmb, das, dae, hr, stride=st_r, max_len=int(aer.max()) + 100)
        sync()
        wl = []
        for _ in range(3):
            sync()
            t0 = time.perf_counter()
            gdev = engr.fingerprint_refine(dmb, das, dae, hr, stride=st_r, max_len=int(aer.max()) + 100)
            sync()
            wl.append(time.perf_counter() - t0)
        dev_same = bool(np.array_equal(gdev[4][:nr].cpu().numpy(), fr.status) and
                        np.array_equal(gdev[0][:nr].cpu().numpy().view(np.uint64), fr.fpt.view(np.uint64)))
        del dmb, gdev
        engr.close()
        # algorithmic bytes per read: the adapter window once + the barcode tail once more (the second segmentation) +
        # fingerprint, statistics, indices and status out
        bpr_r = 4.0 * float((aer - asr + 200).mean()) + 4.0 * 1100 + 8.0 * 25 + 8.0 * 6 + 12.0 + 4.0
        rps_r = nr * rep / (sum(wl) / len(wl))
        trip["trna_refine_flow"] = {
            "reads_per_s": nr / (ms_.value / 3 * 1e-3), "ms": ms_.value / 3, "reads": nr, "timing": "HIP events around the fingerprint launches (host copies excluded)",
            "device_resident": {"reads": nr * rep, "reads_per_s": rps_r, "ms": 1e3 * sum(wl) / len(wl),
                                "same_bits_as_host_call": dev_same,
                                "roofline": {"bound": "hbm", "bytes_per_read": bpr_r, "achieved": rps_r * bpr_r / 1e9, "unit": "GB/s",
                                             "peak": HBM_PEAK_GBS, "frac": rps_r * bpr_r / 1e9 / HBM_PEAK_GBS}},
            "ok_reads": int((fr.status == 0).sum()), "consensus_outliers": int((fr.status == 6).sum()), "parity_reads": ns_,
            "parity": bool(dev_same and np.array_equal(fr.status[:ns_], ostat) and np.array_equal(fr.fpt[:ns_][okt].view(np.uint64), ofp[okt].view(np.uint64))
                           and np.array_equal(fr.refine_idx[:ns_][okt], oidx[okt]))}
    except OSError:
        trip["trna_refine_flow"] = None
    out["other_triples"] = {
        "workload": "fingerprint stage of the other shipped parameter triples (num_events, min_obs_per_base, running_stat_width) on "
                    "262 144 device-resident synthetic reads (RNA004-length windows), the RNA002 triple on 65 536 RNA002-length "
                    "windows (dwell times x2.5: mean 11.5 k samples), and the tRNA config's consensus-refinement flow on 8 192 host reads",
        **trip, "parity": all(v.get("parity") is True for v in trip.values() if isinstance(v, dict))}

    leg_done("other_triples")
    # ---- live (C5): ticks of 1 / 64 / 512 reads through the live shim, WDX6 shape and the shipped WDX6 shape ------
    live = {}
    for K, nYl in ((K_FPT, 6), (25, 1368)):
        pl = sig_proc.SegParams(barcode_num_events=K)
        Yl = rng.normal(size=(nYl, K))
        ld = LiveDemux(Yl, WINDOW, PENALTY, pl, device=device, max_reads=512, max_samples=10000)
        oft, _, _, ost = orc.fingerprint_batch(mb[:512], a_s[:512], a_e[:512], orc.SegParams(barcode_num_events=K))
        okl = ost == 0
        oDl = orc.dtw_matrix(oft[okl], Yl, WINDOW, PENALTY)
        for n, ticks in ((1, 2000), (64, 2000), (512, 2000)):
            rows = [mb[i] for i in range(n)]
            for _ in range(20):
                r = ld.tick(rows, a_s[:n], a_e[:n])
            lat = np.empty(ticks)
            for i in range(ticks):
                t0 = time.perf_counter()
                r = ld.tick(rows, a_s[:n], a_e[:n])
                lat[i] = time.perf_counter() - t0
            lat *= 1e3
            okn = ost[:n] == 0
            par = bool(np.array_equal(r.status, ost[:n]) and
                       np.array_equal(r.dist[okn].view(np.uint32), oDl[:int(okl[:n].sum())].view(np.uint32)) and
                       np.array_equal(r.call[okn], orc.argmin_rows(oDl[:int(okl[:n].sum())])))
            live[f"nY{nYl}_K{K}_reads{n}"] = {"p50_ms": float(np.percentile(lat, 50)), "p99_ms": float(np.percentile(lat, 99)),
                                              "ticks": ticks, "reads_per_s": n / float(np.median(lat)) * 1e3, "parity": par}
        ld.close()
    out["live"] = {"workload": "LiveDemux.tick: all reads of one 100 ms chunk round -> fingerprint + DTW + nearest reference, "
                               "host buffers in/out through pinned staging on the context's stream (reference: 4.3 ms + 13.5 ms per read)",
                   **live}

    leg_done("live")
    # ---- dtw_svm_predict: WDX4-shaped DTW_SVM (851 x 25-pt training rows, 5 classes) on host fingerprints -------
    try:
        from sklearn.svm import SVC

        k, n_train, L = 5, 851, 25
        centers = rng.normal(size=(k, L))
        y = rng.integers(0, k, n_train)
        Xtr = centers[y] + 0.9 * rng.normal(size=(n_train, L))
        Ktr = np.exp(-_oracle_dtw_threads(Xtr, Xtr, budget_s=1e9)[0].astype(np.float64))
        svc = SVC(kernel="precomputed", probability=True, random_state=0).fit(Ktr, y)
        sp = orc.svm_params(svc)
        model = DTW_SVM(Xtr, *sp[:6], {i: i for i in range(k)}, None, WINDOW, PENALTY, block_size=1000, device=device)
        nq = 200_000
        yq = rng.integers(0, k, nq)
        Xq = centers[yq] + 0.9 * rng.normal(size=(nq, L))
        model.predict(Xq, nproc=1)     # same size as the timed calls: the context's device buffers exist and are touched
        walls = []
        for _ in range(10):
            t0 = time.perf_counter()
            y_pred, y_prob = model.predict(Xq, nproc=1)
            walls.append(time.perf_counter() - t0)
        dt = sorted(walls)[len(walls) // 2]      # the median of ten: a pageable 40 MB upload's wall time is noisy
        ns = 512
        Kq = np.exp(-orc.dtw_matrix(Xq[:ns], Xtr, WINDOW, PENALTY))  # float32 exp like the reference (dtw_svm.py:21-22)
        pref = svc.predict_proba(Kq)
        out["dtw_svm_predict"] = {
            "workload": "DTW_SVM.predict on 200 000 host fingerprints, WDX4-shaped model (851 x 25-pt rows, 5 classes), PCIe included",
            "reads_per_s": nq / dt, "ms": dt * 1e3, "statistic": "median", "reps": len(walls), "ms_min": min(walls) * 1e3,
            "ms_max": max(walls) * 1e3,
            "parity": bool(np.abs(y_prob[:ns] - pref).max() <= 1e-5), "max_abs_prob_err": float(np.abs(y_prob[:ns] - pref).max()),
            "parity_tolerance": 1e-5}
    except ImportError:
        out["dtw_svm_predict"] = None
    leg_done("dtw_svm_predict")
    out["shipped_model_e2e"] = secondary_shipped_model_e2e(device)
    leg_done("shipped_model_e2e")
    return out


def make_refs(spec_clean, synth, sig_proc, device=None, n_barcodes=None):
    """Barcode reference fingerprints: one low-noise template read per barcode through the HIP
    fingerprint kernel (host-buffer entry point)."""
    import numpy as np

    from warpdemux_amd import _lib

    N_BARCODES = n_barcodes or globals()["N_BARCODES"]   # (tests build the 4-barcode C2 set)
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
    ctx = _lib.default_context(device)
    ctx.set_option(_lib.OPT_EXACT_PATH, 1)
    try:
        fb = sig_proc.fingerprint_batch(mb, a_s, a_e, sig_proc.SegParams(barcode_num_events=K_FPT), device=device)
    finally:
        ctx.set_option(_lib.OPT_EXACT_PATH, 0)
    if not (fb.status == 0).all():
        raise RuntimeError("template fingerprinting failed")
    return fb.fpt


def run_rank(args):
    import numpy as np
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
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    torch.cuda.set_device(local_rank)
    tdev = torch.device("cuda", local_rank)
    host_collectives = backend == "gloo"

    spec = synth.SynthSpec(n_barcodes=N_BARCODES)
    clean = synth.SynthSpec(n_barcodes=N_BARCODES, noise_sigma=0.25, spikes=False)
    refs = make_refs(clean, synth, sig_proc, local_rank)
    params = sig_proc.SegParams(barcode_num_events=K_FPT)
    eng = DemuxEngine(refs, WINDOW, PENALTY, params, device=local_rank)
    for kv in args.ctx_opt:
        k, v = kv.split("=")
        eng.ctx.set_option(int(k), int(v))
    reducer = dist.CountReducer(eng.ctx, prefer="torch" if host_collectives else None)
    if world > 1 and not host_collectives and reducer.mode != "rccl":
        # the multi-GPU figure must be the C ABI's RCCL road; anything else is a configuration to fix, not to time
        raise SystemExit(f"bench.py: rank {rank}: --gpus {world} ended on the '{reducer.mode}' road ({reducer.note}); "
                         "set WDX_BENCH_BACKEND=gloo to time the process-group fallback on purpose")

    # ---- the global read range and this rank's contiguous shard of it ----------------------------
    per_gpu = args.reads or (10_000_000 if world == 1 else 5_000_000)
    workload = "C3" if world == 1 else "C4"
    while True:
        total = per_gpu * world
        lo, hi = dist.shard_range(total, rank, world)
        n_reads = hi - lo
        fit = 1
        try:
            sig, off, a_s, a_e, bc, max_len = eng.synth_packed(spec, lo, n_reads)
            res = eng.demux(sig, a_s, a_e, offsets=off, max_len=max_len)  # allocates outputs/workspace
            torch.cuda.synchronize()
        except torch.OutOfMemoryError:
            fit = 0
        except _lib.WdxError as e:
            if "hipMalloc" not in str(e):
                raise
            fit = 0
        # every rank takes the same decision (shards must tile the global range)
        fit = int(dist.min_over_ranks(fit, device=None if host_collectives else tdev))
        if fit:
            break
        if per_gpu <= 100_000:
            raise RuntimeError("the device cannot hold even 100 000 reads")
        sig = off = a_s = a_e = bc = res = None
        eng._work = None
        torch.cuda.empty_cache()
        if rank == 0:
            print(f"note: {per_gpu} reads per GPU did not fit; halving", file=sys.stderr)
        per_gpu //= 2
    total_samples = int(off[-1].item())

    def step():
        res.counts.zero_()
        eng.demux(sig, a_s, a_e, offsets=off, max_len=max_len, out=res)
        if host_collectives:
            c = res.counts.cpu()
            reducer(c)
            res.counts.copy_(c)
        else:
            reducer(res.counts, stream=eng._stream())

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
    coll_dev = tdev if (world > 1 and not host_collectives) else None
    elapsed = dist.max_over_ranks(t1 - t0, device=coll_dev)
    per_rank_ms = [v / args.steps * 1e3 for v in dist.gather_over_ranks(t1 - t0, device=coll_dev)]

    fp_ms, fp_n = eng.kernel_time(_lib.K_FINGERPRINT)            # the whole fingerprint chain
    fpm_ms, fpm_n = eng.kernel_time(_lib.K_FINGERPRINT_MAIN)     # its main kernel's launches alone
    fpc_ms, fpc_n = eng.kernel_time(_lib.K_FINGERPRINT_CLIP)     # clip_bounds_kernel (median / MAD ahead of the main kernel)
    fpt_ms, fpt_n = eng.kernel_time(_lib.K_FINGERPRINT_TAIL)     # split main kernel: its tail-kernel launches alone (part of MAIN)
    dtw_ms, dtw_n = eng.kernel_time(_lib.K_DTW)
    tr_ms, tr_n = eng.kernel_time(_lib.K_TRANSPOSE)
    cnt_ms, cnt_n = eng.kernel_time(_lib.K_COUNT)
    red_ms, red_n = eng.kernel_time(_lib.K_REDUCE)

    counts = res.counts.cpu().numpy()
    n_fail = int(counts[N_BARCODES])
    if int(counts.sum()) != total:
        raise RuntimeError(f"rank {rank}: all-reduced call histogram does not add up: {counts.sum()} != {total}")
    if rank == 0 and n_fail > 0.01 * total:
        raise RuntimeError(f"{n_fail} of {total} reads failed: the synthetic workload should fingerprint cleanly")
    # every read of the shard, on the device: the calls against the generator's true barcodes, per tenth of the shard
    # (a launch slice that wrote nothing, or wrote another slice's reads, cannot pass this; the bitwise gate below
    # samples the same range)
    okd = res.status == 0
    hit = ((res.call == bc) & okd).to(torch.float64)
    edges = [n_reads * i // 10 for i in range(11)]
    acc_tenths = [float(hit[a:b].sum().item() / max(int(okd[a:b].sum().item()), 1)) for a, b in zip(edges[:-1], edges[1:])]
    accuracy = float(hit.sum().item() / max(int(okd.sum().item()), 1))
    if min(acc_tenths) <= 0.75:
        raise RuntimeError(f"rank {rank}: calls disagree with the generator's barcodes: accuracy per tenth of the shard {acc_tenths}")
    del okd, hit

    # ---- roofline of the dominant kernel (algorithmic bytes, DESIGN.md "Measurement") ------------
    steps = max(args.steps, 1)
    fp_bytes = 4.0 * total_samples + 8.0 * K_FPT * n_reads + 4.0 * n_reads
    dtw_bytes = (8.0 * K_FPT + 4.0 * N_BARCODES + 4.0) * n_reads
    if fp_ms >= dtw_ms:
        # The fingerprint stage is a chain of launches: the main kernel takes every read whose adapter
        # window fits its instantiation and the list kernels the rest.  `achieved` prices the main kernel alone: the
        # algorithmic bytes of the reads IT completes (window within its capacity; the engine reports the capacity)
        # over ITS launches' HIP-event time -- the figure rocprofv3's average for that kernel must reproduce.
        main_cap = 5120 if max_len > 4096 and n_reads >= 2048 else (4096 if max_len <= 4096 else 6144)
        w_start = torch.clamp(a_s.to(torch.int64) - params.padding, min=0)
        w_stop = torch.minimum(a_e.to(torch.int64) + params.padding, off[1:] - off[:-1])
        lens = w_stop - w_start
        fits = lens <= main_cap
        main_reads = int(fits.sum().item())
        main_samples = int(lens[fits].sum().item())
        main_bytes = 4.0 * main_samples + (8.0 * K_FPT + 4.0) * main_reads
        dom, dom_ms, dom_n, dom_bytes = "fingerprint_fast_kernel", fpm_ms, fpm_n, main_bytes
        # Round 6: the main kernel is a PAIR per launch slice -- the workgroup-per-read tile kernel (fingerprint_fast_kernel<...,
        # SPLIT>: load, clip, t-score tiles, peak list, export of <= 256 peaks with prefix sums) and the wave-per-read tail
        # kernel (fingerprint_split_tail_kernel: suppression, top-E, boundaries, event means, normalisation, output).  The
        # MAIN events bracket the pairs, the TAIL events the tail kernel alone; the dominant kernel is the tile kernel, priced
        # on ITS algorithmic bytes (the samples it reads: 4 N per read -- the fingerprints are the tail kernel's output) over
        # ITS time (MAIN - TAIL), the figure rocprofv3's average for that kernel must reproduce.
        split_pair = None
        if fpt_n > 0 and fpt_ms > 0:
            split_pair = {"tile_kernel": "fingerprint_fast_kernel<20, false, 12, 1, true, true>", "tail_kernel": "fingerprint_split_tail_kernel<12>",
                          "launch_pairs": fpm_n, "pair_ms_per_step": fpm_ms / steps, "tail_ms_per_step": fpt_ms / steps,
                          "tile_ms_per_step": (fpm_ms - fpt_ms) / steps, "avg_tail_launch_ms": fpt_ms / fpt_n,
                          "pair_algorithmic_bytes_per_step": main_bytes,
                          "pair_achieved_gbs": main_bytes / (fpm_ms / steps * 1e-3) / 1e9,
                          "pair_frac": main_bytes / (fpm_ms / steps * 1e-3) / 1e9 / HBM_PEAK_GBS}
            dom_ms, dom_bytes = fpm_ms - fpt_ms, 4.0 * main_samples
    else:
        dom, dom_ms, dom_n, dom_bytes = "dtw_band_kernel<15>", dtw_ms, dtw_n, dtw_bytes
    launches_per_step = max(dom_n // steps, 1)
    avg_ms = dom_ms / max(dom_n, 1)
    dom_bytes = dom_bytes / launches_per_step  # per kernel launch (a step may be sliced)
    achieved = dom_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    # HBM traffic of that kernel from the PMC passes (profiles/traffic.json, written by
    # tools/collect_traffic.py on the same workload); null when not collected for this size
    # NOT measured by this run: PMC counters need rocprofv3 around the process, so these two figures are read from the
    # committed passes of the same workload and carry their source (file, round) in the line
    traffic = traffic_src = traffic_parts = None
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as fh:
            tj = json.load(fh)
        if tj.get("kernel") == dom and tj.get("reads_per_step") == n_reads:
            traffic = tj.get("hbm_bytes_per_step") / launches_per_step
            traffic_parts = {"fetched_per_launch": tj.get("fetch_bytes_per_launch_corrected"), "written_per_launch": tj.get("write_bytes_per_launch"),
                             "note": "the tile kernel reads each sample once (fetched = 1.02 x its algorithmic bytes); what it writes are the "
                                     "exported peak lists (<= 4.6 KB per read) that the tail kernel reads back"}
            traffic_src = {"file": "profiles/traffic.json", "round": tj.get("round"), "collected_by": "tools/collect_traffic.py "
                           "(separate rocprofv3 --pmc passes of this command; not measured by this run)"}
    except (OSError, ValueError, TypeError):
        pass
    valu_busy = valu_src = None
    try:
        with open(os.path.join(ROOT, "profiles", "sq_counters_latest.json")) as fh:
            sj = json.load(fh)
        pl = sj["kernels"][dom.split("<")[0]]["per_launch"]
        valu_busy = pl["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * pl["GRBM_GUI_ACTIVE"] / 8.0)
        valu_src = {"file": "profiles/sq_counters_latest.json", "round": sj.get("round"),
                    "collected_by": "tools/collect_sq.py (rocprofv3 --pmc pass of this command; not measured by this run)"}
    except (OSError, ValueError, KeyError):
        pass

    out = None
    if rank == 0:
        value = total * args.steps / elapsed
        fused_bytes_per_read = 4.0 * total_samples / n_reads + 4.0 * N_BARCODES + 4.0
        out = {
            "metric": "reads/sec demuxed (110-pt DTW x 10 barcodes)",
            "value": value,
            "unit": "reads/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "per_rank_ms_per_step": per_rank_ms,     # rank order; ms_per_step is their maximum
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": ("%s: %d synthetic RNA004 adapter reads per GPU (%d in all; wdx-synth v1, mean %.0f samples), "
                             "WDX10 shape: 10 barcodes x 110-pt fingerprints, window 15, penalty 0.1; fused "
                             "fingerprint+DTW+call+count, raw rows resident in HBM" % (workload, per_gpu, total, total_samples / n_reads)),
                "reads_per_gpu": per_gpu,
                "reads_total": total,
                "failed_reads": n_fail,
                "call_accuracy_vs_true_barcode": {"whole_shard": accuracy, "min_over_tenths": min(acc_tenths),
                                                  "asserted": "> 0.75 in every tenth of rank 0's shard (all reads, on the device)"},
                **({"ctx_options": list(args.ctx_opt)} if args.ctx_opt else {}),
                "sharding": "contiguous shards of one global read range (dist.shard_range), one process per GPU, "
                            "one int64[11] count all-reduce per step",
                "count_allreduce": {"single": "none (one process)", "rccl": "wdx_reduce_counts (C ABI, RCCL)",
                                    "torch": "torch.distributed.all_reduce (%s)" % (backend or "nccl")}[reducer.mode]
                                   + (("; " + reducer.note) if reducer.note else ""),
                "rccl_ranks": reducer.rccl_ranks,     # ncclCommCount of the C ABI's communicator (null: no communicator)
            },
            "roofline": {
                "bound": "hbm",
                "kernel": dom,
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "traffic_parts": traffic_parts,
                "algorithmic_bytes_per_launch": dom_bytes,
                "avg_launch_ms": avg_ms,
                "launches": dom_n,
                **({"kernel_instantiation": ("fingerprint_fast_kernel<%d, false, 12, 1, true, true> (%d-sample windows; the tile kernel of the split "
                                             "main pair: algorithmic bytes = the samples it reads)" % (main_cap // 256, main_cap)) if split_pair else
                                            "fingerprint_fast_kernel<%d, false> (%d-sample windows)" % (main_cap // 256, main_cap),
                    "split_main_pair": split_pair,
                    "reads_per_launch": main_reads / launches_per_step,
                    "share_of_reads": main_reads / n_reads,
                    "stage": {"what": "whole fingerprint chain (clip_bounds_kernel ahead of the main kernel + main kernel + "
                                      "list kernels for longer windows, peak-list overflows, exact-score retries, the "
                                      "exact general kernel)",
                              "ms_per_step": fp_ms / steps,
                              "algorithmic_gb_per_step": fp_bytes / 1e9,
                              "achieved_gbs": fp_bytes / (fp_ms / steps * 1e-3) / 1e9 if fp_ms else None},
                    # the launch ahead of the main kernel: one wave per read, the samples of the main kernel's reads once
                    # more from HBM (4 N bytes in, 16 bytes out per read) -- priced like the main kernel, on its own
                    # algorithmic bytes and its own HIP events
                    "clip_bounds_kernel": ({"ms_per_step": fpc_ms / steps, "launches": fpc_n,
                                            "algorithmic_bytes_per_step": 4.0 * main_samples + 16.0 * n_reads,
                                            "achieved": (4.0 * main_samples + 16.0 * n_reads) / (fpc_ms / steps * 1e-3) / 1e9,
                                            "frac": (4.0 * main_samples + 16.0 * n_reads) / (fpc_ms / steps * 1e-3) / 1e9 / HBM_PEAK_GBS}
                                           if fpc_ms else None)}
                   if dom == "fingerprint_fast_kernel" else {}),
                # what actually bounds the kernel (float64 VALU issue), from the committed PMC pass
                "valu_busy_frac": valu_busy,
                "valu_busy_source": valu_src,
            },
            "kernels_ms_per_step": {   # HIP-event sums over all launches of a step (a step may be sliced)
                "fingerprint": fp_ms / steps, "fingerprint_main_kernel": fpm_ms / steps,
                "fingerprint_tail_kernel": fpt_ms / steps, "fingerprint_clip_kernel": fpc_ms / steps, "dtw": dtw_ms / steps,
                "transpose": tr_ms / steps,
                "count": cnt_ms / steps, "count_allreduce": red_ms / steps,
            },
            "fused_path": {
                "algorithmic_bytes_per_read": fused_bytes_per_read,
                "hbm_frac_whole_job_per_gpu": value / world * fused_bytes_per_read / 1e9 / HBM_PEAK_GBS,
                "dtw_gcups_per_gpu": (n_reads * float(N_BARCODES * CELLS_110)) / (dtw_ms / max(dtw_n, 1) * 1e-3) / 1e9 if dtw_ms else None,
            },
        }

    # ---- CPU baseline + parity gate on a bounded sample (rank 0, N=1 only) ------------------------------
    parity_failed = False
    if rank == 0 and world == 1 and not args.no_cpu:
        cpu, parity = cpu_baseline_and_parity(eng, sig, off, a_s, a_e, res, refs, n_reads,
                                              args.cpu_seconds, args.parity_reads)
        out["cpu_baseline"] = cpu
        out["parity_gate"] = parity
        out["parity_on_sample"] = parity["ok"]
        parity_failed = not parity["ok"]
        if parity_failed:
            print("ERROR: GPU results differ from the oracle on the CPU sample: %r" % (parity,), file=sys.stderr)
        ns = min(1000, n_reads)
        g_fpt = eng.fingerprint(sig[: int(off[ns].item())], a_s[:ns], a_e[:ns], offsets=off[: ns + 1], max_len=max_len)[0]
        fs = g_fpt.cpu().numpy()
        out["dtaidistance"] = try_dtaidistance(eng, refs, fs[~np.isnan(fs).any(axis=1)])
    elif rank == 0:
        out["cpu_baseline"] = None

    # ---- secondary regimes, after the big buffers are gone ----------------------------------------------
    if rank == 0 and world == 1 and not args.no_secondary:
        sig = off = a_s = a_e = bc = res = None
        eng._work = None
        torch.cuda.synchronize()
        torch.cuda.empty_cache()          # ~190 GB back to the driver ...
        torch.cuda.synchronize()
        settle = torch.zeros(1 << 28, dtype=torch.uint8, device=tdev)   # ... and the unmapping done before any leg starts
        torch.cuda.synchronize()
        del settle
        sec = secondary_regimes(local_rank)
        out["secondary"] = sec
        for name, r in sec.items():
            if isinstance(r, dict):
                flags = [r.get("parity")] + [v.get("parity") for v in r.values() if isinstance(v, dict)]
                if any(f is False for f in flags):
                    parity_failed = True
                    print(f"ERROR: secondary regime {name} differs from the oracle", file=sys.stderr)

    if rank == 0:
        print(json.dumps(out), flush=True)
    reducer.close()
    eng.close()
    if world > 1:
        import torch.distributed as tdist

        tdist.destroy_process_group()
    if parity_failed:
        sys.exit(2)


def main():
    args = parse_args()
    if args.leg:
        fn = {"shipped_model_e2e": secondary_shipped_model_e2e}[args.leg]
        print(json.dumps({args.leg: fn(int(os.environ.get("LOCAL_RANK", "0")))}), flush=True)
        return
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))  # nothing in this process has touched the GPU
    run_rank(args)


if __name__ == "__main__":
    main()

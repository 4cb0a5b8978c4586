#!/usr/bin/env python3
"""Secondary measurements (not the headline metric): shipped-model DTW regime R2, host-buffer
(PCIe-inclusive) minibatch calls as file_proc would make them, and small-batch latency for the live
path (BASELINE config 5).  Run on the GPU box: python tools/bench_regimes.py"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from warpdemux_amd import parallel_distances as pdist  # noqa: E402
from warpdemux_amd import sig_proc, synth  # noqa: E402
from warpdemux_amd.engine import DemuxEngine  # noqa: E402

out = {}
rng = np.random.default_rng(0)


def timeit(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), float(np.min(ts))


# ---- R2: device-resident DTW, L=25, shipped-model reference counts -------------------------------
cells25 = sum(min(25, i + 15) - max(0, i - 14) for i in range(25))
for nY in (851, 1368, 2601):
    Y = rng.normal(size=(nY, 25))
    eng = DemuxEngine(Y, 15, 0.1, sig_proc.SegParams(barcode_num_events=25))
    for nX in (1000, 20000, 200000):
        X = torch.from_numpy(rng.normal(size=(nX, 25))).cuda()
        med, best = timeit(lambda: eng.dtw(X, want_argmin=False))
        out[f"R2_dev_nY{nY}_nX{nX}"] = {"ms": med * 1e3, "reads_per_s": nX / med,
                                         "gcups": nX * nY * cells25 / med / 1e9}
    eng.close()

# ---- host-buffer minibatch calls (PCIe inclusive), as file_proc.py:418-450 would issue them ------
spec = synth.SynthSpec(n_barcodes=10)
mb, a_s, a_e, _ = synth.generate_minibatch(spec, 0, 1000, 10000)
for K, nY in ((25, 2601), (110, 10)):
    p = sig_proc.SegParams(barcode_num_events=K)
    Y = rng.normal(size=(nY, K))
    fb = sig_proc.fingerprint_batch(mb, a_s, a_e, p)
    Xf = fb.fpt[fb.status == 0]
    t_fp, _ = timeit(lambda: sig_proc.fingerprint_batch(mb, a_s, a_e, p))
    t_dtw, _ = timeit(lambda: pdist.distance_matrix_to(Xf, Y, window=15, penalty=0.1, n_jobs=1))
    out[f"host_minibatch1000_K{K}_nY{nY}"] = {"fingerprint_ms": t_fp * 1e3, "dtw_ms": t_dtw * 1e3,
                                              "reads_per_s": 1000 / (t_fp + t_dtw)}

# ---- live path: a handful of reads per 100 ms tick, WDX6 shape and shipped WDX6 shape --------------
for K, nY in ((110, 6), (25, 1368)):
    p = sig_proc.SegParams(barcode_num_events=K)
    Y = rng.normal(size=(nY, K))
    for n in (1, 8, 64, 512):
        sub, s_, e_ = mb[:n] if n <= 1000 else mb, a_s[:n], a_e[:n]
        if n > 1000:
            continue

        def tick():
            fb = sig_proc.fingerprint_batch(sub, s_, e_, p)
            return pdist.nearest_reference(fb.fpt[fb.status == 0], Y, 15, 0.1)

        lat = []
        tick(); tick()
        for _ in range(30):
            t0 = time.perf_counter()
            tick()
            lat.append(time.perf_counter() - t0)
        lat = np.array(lat) * 1e3
        out[f"live_K{K}_nY{nY}_n{n}"] = {"p50_ms": float(np.percentile(lat, 50)), "p99_ms": float(np.percentile(lat, 99)),
                                          "reads_per_s": n / float(np.median(lat)) * 1e3}
# ---- the same ticks through the fused host call (one synchronisation) ---------------------------------
for K, nY in ((110, 6), (25, 1368)):
    p = sig_proc.SegParams(barcode_num_events=K)
    Y = rng.normal(size=(nY, K))
    sig_proc.set_references(Y, 15, 0.1)
    for n in (1, 8, 64, 512):
        sub, s_, e_ = mb[:n], a_s[:n], a_e[:n]
        lat = []
        for _ in range(3):
            sig_proc.demux_batch(sub, s_, e_, p, n_refs=nY)
        for _ in range(30):
            t0 = time.perf_counter()
            sig_proc.demux_batch(sub, s_, e_, p, n_refs=nY)
            lat.append(time.perf_counter() - t0)
        lat = np.array(lat) * 1e3
        out[f"live_fused_K{K}_nY{nY}_n{n}"] = {"p50_ms": float(np.percentile(lat, 50)), "p99_ms": float(np.percentile(lat, 99)),
                                                "reads_per_s": n / float(np.median(lat)) * 1e3}
print(json.dumps(out, indent=1))

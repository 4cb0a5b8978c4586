#!/usr/bin/env python3
"""Diagnostic: per-phase shader-clock shares and throughput of the EXACT general fingerprint kernel
(fp_process_read) for a parameter triple.  Run on the GPU box:
    python tools/profile_exact.py E d W [n_reads] [scale]      (scale stretches the synthetic dwell times)"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from warpdemux_amd import _lib, sig_proc, synth  # noqa: E402
from warpdemux_amd.engine import DemuxEngine, _dp  # noqa: E402

E, d, W = (int(v) for v in sys.argv[1:4])
n = int(sys.argv[4]) if len(sys.argv) > 4 else 8192
K = 25
spec = synth.SynthSpec(n_barcodes=10)
params = sig_proc.SegParams(num_events=E, min_obs_per_base=d, running_stat_width=W, barcode_num_events=K)
eng = DemuxEngine(np.zeros((10, K)), 15, 0.1, params)
sig, off, a_s, a_e, bc, max_len = eng.synth_packed(spec, 0, n)
status = torch.empty(n, dtype=torch.int32, device="cuda")
prof = torch.zeros((n, 32), dtype=torch.int64, device="cuda")
pc = params.to_c()
for _ in range(2):
    _lib.check(eng.L.wdx_fingerprint_profile_dev(eng.ctx.handle, _dp(sig), _dp(off), 0, max_len, n, _dp(a_s), _dp(a_e),
                                                 C.byref(pc), _dp(status), _dp(prof), n, 0, 0, None))
torch.cuda.synchronize()
p = prof.cpu().numpy()
ok = status.cpu().numpy() == 0
names = ["P0 load", "P1 median+clip", "P2 t-score", "P3-P6 fp_segment", "(4)", "(5)", "(6)", "refine / -", "P7 normalise/stats"]
p = p[ok]
st = p[:, :10].copy()
# unused stamps stay 0: carry the previous stamp forward
for i in range(1, 10):
    st[:, i] = np.where(st[:, i] == 0, st[:, i - 1], st[:, i])
dd = np.diff(st, axis=1)
tot = st[:, 9] - st[:, 0]
print(f"triple ({E},{d},{W}) reads ok {ok.sum()}/{n}  max_len {max_len}  median cycles/read {np.median(tot):.0f}")
for i in range(9):
    if dd[:, i].sum():
        print(f"  {names[i]:22s} median {np.median(dd[:, i]):9.0f}  share {dd[:, i].sum() / tot.sum() * 100:5.1f}%")
print("  suppression iterations: median %d  p99 %d" % (np.median(p[:, 10]), np.percentile(p[:, 10], 99)))
# throughput of the product entry point on this triple (whatever path the engine picks) and on the exact kernel
for exact in (0, 1):
    eng.ctx.set_option(_lib.OPT_EXACT_PATH, exact)
    for _ in range(2):
        eng.fingerprint(sig, a_s, a_e, offsets=off, max_len=max_len)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        eng.fingerprint(sig, a_s, a_e, offsets=off, max_len=max_len)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"  wdx_fingerprint_dev, exact_path={exact}: {n / dt / 1e6:.3f} M reads/s ({dt * 1e3:.2f} ms for {n} reads)")

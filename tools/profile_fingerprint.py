#!/usr/bin/env python3
"""Diagnostic: per-phase shader-clock shares of the fingerprint kernel (PROF instantiation).
Run on the GPU box:  python tools/profile_fingerprint.py [n_reads]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from warpdemux_amd import _lib, sig_proc, synth  # noqa: E402
from warpdemux_amd.engine import DemuxEngine, _dp  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
K = 110
spec = synth.SynthSpec(n_barcodes=10)
eng = DemuxEngine(np.zeros((10, K)), 15, 0.1, sig_proc.SegParams(barcode_num_events=K))
sig, off, a_s, a_e, bc, max_len = eng.synth_packed(spec, 0, n)
status = torch.empty(n, dtype=torch.int32, device="cuda")
prof = torch.zeros((n, 16), dtype=torch.int64, device="cuda")
pc = eng.params.to_c()
for _ in range(2):
    _lib.check(eng.L.wdx_fingerprint_profile_dev(eng.ctx.handle, _dp(sig), _dp(off), 0, max_len, n, _dp(a_s), _dp(a_e),
                                                 C.byref(pc), _dp(status), _dp(prof), n, None))
torch.cuda.synchronize()
p = prof.cpu().numpy()
ok = status.cpu().numpy() == 0
p = p[ok]
names = ["P0 load", "P1 median+clip", "A2+shrink", "P2 t-score", "P3a local maxima", "P3b suppression", "P4 top-E",
         "P5 boundaries", "P6 event means", "P7 normalise/stats"]
d = np.diff(p[:, :10], axis=1)
tot = p[:, 9] - p[:, 0]
print(f"reads {ok.sum()}  max_len {max_len}  median total cycles/read {np.median(tot):.0f}  mean {tot.mean():.0f}")
for i in range(9):
    print(f"  {names[i]:22s} median {np.median(d[:, i]):9.0f}  mean {d[:, i].mean():9.0f}  share {d[:, i].sum() / tot.sum() * 100:5.1f}%")
print("  suppression iterations: median %d  p99 %d  max %d" % (np.median(p[:, 10]), np.percentile(p[:, 10], 99), p[:, 10].max()))
print("  n samples median %d, score positions median %d" % (np.median(p[:, 11]), np.median(p[:, 12])))

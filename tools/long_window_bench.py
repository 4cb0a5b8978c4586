#!/usr/bin/env python3
"""Fingerprint-stage throughput by adapter-window length (device-resident minibatch layout), for one parameter triple:
which kernel a window reaches and what it costs there.   python tools/long_window_bench.py [E d W] [n_reads]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from warpdemux_amd import sig_proc  # noqa: E402
from warpdemux_amd.engine import DemuxEngine  # noqa: E402

E, d, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (110, 15, 30)
n = int(sys.argv[4]) if len(sys.argv) > 4 else 8192
rng = np.random.default_rng(1)
params = sig_proc.SegParams(padding=0, num_events=E, min_obs_per_base=d, running_stat_width=W, barcode_num_events=25)
eng = DemuxEngine(np.zeros((4, 25)), 15, 0.1, params)
print(f"triple ({E},{d},{W}), {n} reads per length")
for ln in (4600, 6000, 7500, 8192, 9000, 11200, 12000, 15200):
    dw = max(12, ln // 135)
    base = (np.repeat(rng.normal(80, 15, (64, ln // dw + 1)), dw, axis=1)[:, :ln] + rng.normal(0, 2, (64, ln))).astype(np.float32)
    mb = torch.from_numpy(np.tile(base, (n // 64, 1))).cuda()
    a_s = torch.zeros(n, dtype=torch.int32, device="cuda")
    a_e = torch.full((n,), ln, dtype=torch.int32, device="cuda")
    for _ in range(2):
        out = eng.fingerprint(mb, a_s, a_e, stride=ln, max_len=ln)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        out = eng.fingerprint(mb, a_s, a_e, stride=ln, max_len=ln)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    ok = int((out[3] == 0).sum().item())
    print(f"  window {ln:6d}: {n / dt / 1e6:7.3f} M reads/s  ({dt * 1e3:8.2f} ms, {ok} ok; {n * ln * 4 / dt / 1e9:7.1f} GB/s of samples)")

"""Fingerprint throughput on adapter windows of ~6.5k..8k samples (beyond the 6144-sample fast kernel):
python tools/long_adapter_bench.py [n_reads]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from warpdemux_amd.engine import DemuxEngine
from warpdemux_amd import sig_proc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
rng = np.random.default_rng(3)
L = 8000
base = np.empty((256, L), dtype=np.float32)
for i in range(256):
    ev = rng.integers(25, 55)
    base[i] = np.round((np.repeat(rng.normal(80, 15, L // ev + 1), ev)[:L] + rng.normal(0, 2, L)) * 8) / 8
sig = torch.from_numpy(base).cuda().repeat((n + 255) // 256, 1)[:n].contiguous()
lens = torch.randint(6400, 8000, (n,), dtype=torch.int32, device="cuda")
a_s = torch.zeros(n, dtype=torch.int32, device="cuda")
eng = DemuxEngine(np.zeros((10, 110)), 15, 0.1, sig_proc.SegParams(padding=0, barcode_num_events=110))
mode = "default"
if len(sys.argv) > 2 and sys.argv[2] == "exact":   # python tools/long_adapter_bench.py N exact
    from warpdemux_amd import _lib
    eng.ctx.set_option(_lib.OPT_EXACT_PATH, 1)
    mode = "exact path only"
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = eng.fingerprint(sig, a_s, lens, stride=L, max_len=L)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{mode}: {n / dt / 1e6:.3f} M reads/s "
          f"({dt * 1e3:.1f} ms), ok={(out[3] == 0).float().mean().item():.4f}")

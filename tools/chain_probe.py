#!/usr/bin/env python3
"""Diagnostic: the fingerprint launch chain on N synthetic RNA004 reads (device-resident) -- how many reads each
hand-over list received (WDX_OPT_DEBUG_OCCUPANCY) and the stage's throughput.  Under rocprofv3 --kernel-trace --stats
this gives the per-kernel split of the chain.   python tools/chain_probe.py [n_reads] [reps]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from warpdemux_amd import _lib, sig_proc, synth  # noqa: E402
from warpdemux_amd.engine import DemuxEngine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
K = 110
spec = synth.SynthSpec(n_barcodes=10)
params = sig_proc.SegParams(barcode_num_events=K)
eng = DemuxEngine(np.zeros((10, K)), 15, 0.1, params)
sig, off, a_s, a_e, bc, max_len = eng.synth_packed(spec, 0, n)
eng.ctx.set_option(_lib.OPT_DEBUG_OCCUPANCY, 1)
eng.fingerprint(sig, a_s, a_e, offsets=off, max_len=max_len)
eng.ctx.set_option(_lib.OPT_DEBUG_OCCUPANCY, 0)
for _ in range(2):
    eng.fingerprint(sig, a_s, a_e, offsets=off, max_len=max_len)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    out = eng.fingerprint(sig, a_s, a_e, offsets=off, max_len=max_len)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
st = out[3].cpu().numpy()
print(f"{n} reads, max_len {max_len}: {n / dt / 1e6:.3f} M reads/s ({dt * 1e3:.2f} ms), status ok {(st == 0).sum()}")

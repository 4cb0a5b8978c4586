"""Throughput of the fast fingerprint kernel with adapters truncated to <= 3900 samples (all reads take the
4096-sample instantiation), for the peak-list capacity given by WDX_FAST_CAPP (LDS per workgroup -> workgroups/CU)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from warpdemux_amd import _lib, synth
from warpdemux_amd.engine import DemuxEngine, _dp

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
trunc = int(sys.argv[2]) if len(sys.argv) > 2 else 3700
spec = synth.SynthSpec(seed=1234, n_barcodes=10)
eng = DemuxEngine(np.zeros((10, 110)), 15, 0.1)
sig, off, a_s, a_e, bc, max_len = eng.synth_packed(spec, 0, n)
a_e = torch.minimum(a_e, a_s + trunc).contiguous()
ml = trunc + 200
for rep in range(4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = eng.fingerprint(sig, a_s, a_e, offsets=off, max_len=ml)
    e1.record()
    torch.cuda.synchronize()
    print(f"capP={os.environ.get('WDX_FAST_CAPP', 'default')} rep {rep}: {e0.elapsed_time(e1):.2f} ms per {n} reads; "
          f"ok={(out[3] == 0).float().mean().item():.4f}")

#!/usr/bin/env python3
"""One parameter triple of the fingerprint stage on device-resident synthetic reads, a few repetitions -- made to be run
under `rocprofv3 --kernel-trace --stats` for the per-kernel split of that triple's launch chain.

    python tools/bench_triple.py E d W [n_reads] [dwell_scale] [reps]      e.g. 110 15 30 65536 2.5   (RNA002-length windows)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from warpdemux_amd import sig_proc, synth  # noqa: E402
from warpdemux_amd.engine import DemuxEngine  # noqa: E402

E, d, W = (int(v) for v in sys.argv[1:4])
n = int(sys.argv[4]) if len(sys.argv) > 4 else 65536
dscale = float(sys.argv[5]) if len(sys.argv) > 5 else 1.0
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 5
pt = sig_proc.SegParams(num_events=E, min_obs_per_base=d, running_stat_width=W, barcode_num_events=25)
eng = DemuxEngine(np.zeros((10, 25)), 15, 0.1, pt)
if os.environ.get("WDX_DEBUG_CHAIN"):   # the launch chain's hand-over counts on stderr (synchronises)
    from warpdemux_amd import _lib
    eng.ctx.set_option(_lib.OPT_DEBUG_OCCUPANCY, 1)
spec = synth.SynthSpec(n_barcodes=10) if dscale == 1.0 else synth.SynthSpec(n_barcodes=10, dwell_scale=dscale)
sg, of, s0, e0, _, mlen = eng.synth_packed(spec, 0, n)
for _ in range(2):
    g = eng.fingerprint(sg, s0, e0, offsets=of, max_len=mlen)
torch.cuda.synchronize()
walls = []
for _ in range(reps):
    t0 = time.perf_counter()
    g = eng.fingerprint(sg, s0, e0, offsets=of, max_len=mlen)
    torch.cuda.synchronize()
    walls.append(time.perf_counter() - t0)
dt = sum(walls) / len(walls)
print("triple (%d, %d, %d), %d reads, mean window %.0f samples (max %d): %.3f M reads/s (%.2f ms), %d ok" % (
    E, d, W, n, float(of[-1].item()) / n, int(mlen), n / dt / 1e6, dt * 1e3, int((g[3] == 0).sum().item())))

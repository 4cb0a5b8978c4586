#!/usr/bin/env python3
"""Does a fifth resident workgroup per CU pay?  Same reads (adapter windows cut to <= 4000 samples) through the
4096-sample instantiation of the fast fingerprint kernel and, by declaring max_len = 6144, through the 6144-sample one.
Run on the GPU box:  python tools/probes/occupancy_probe.py [n_reads]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from warpdemux_amd import _lib, sig_proc, synth  # noqa: E402
from warpdemux_amd.engine import DemuxEngine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
K = 110
spec = synth.SynthSpec(n_barcodes=10)
eng = DemuxEngine(np.zeros((10, K)), 15, 0.1, sig_proc.SegParams(barcode_num_events=K))
eng.ctx.set_option(_lib.OPT_DEBUG_OCCUPANCY, 1)
sig, off, a_s, a_e, bc, max_len = eng.synth_packed(spec, 0, n)
a_e = torch.minimum(a_e, a_s + 3800)
for ml, force, pcap in ((4096, 0, 0), (5120, 5120, 0), (5120, 5120, 976), (5120, 5120, 920), (6144, 6144, 0), (5120, 5120, 976)):
    eng.ctx.set_option(_lib.OPT_FAST_MAIN_CAP, force)
    eng.ctx.set_option(_lib.OPT_FAST_PEAK_CAP, pcap)
    ts = []
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = eng.fingerprint(sig, a_s, a_e, offsets=off, max_len=ml)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
        eng.ctx.set_option(_lib.OPT_DEBUG_OCCUPANCY, 0)
    print(f"max_len {ml} forced main {force} peak cap {pcap}: {min(ts):.2f} ms for {n} reads  ok={int((out[3] == 0).sum())}")
    eng.ctx.set_option(_lib.OPT_DEBUG_OCCUPANCY, 1)

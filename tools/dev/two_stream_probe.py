#!/usr/bin/env python3
"""Does the C3 step gain from two halves of the shard running as two launch chains on two streams (two contexts), so that
one half's DTW (float64 VALU, no HBM) runs beside the other half's clip / fingerprint kernels?
    python tools/dev/two_stream_probe.py [n_reads] [pieces]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from warpdemux_amd import sig_proc, synth
from warpdemux_amd.engine import DemuxEngine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
pieces = int(sys.argv[2]) if len(sys.argv) > 2 else 2
spec = synth.SynthSpec(n_barcodes=bench.N_BARCODES)
clean = synth.SynthSpec(n_barcodes=bench.N_BARCODES, noise_sigma=0.25, spikes=False)
refs = bench.make_refs(clean, synth, sig_proc, 0)
params = sig_proc.SegParams(barcode_num_events=bench.K_FPT)
engs = [DemuxEngine(refs, bench.WINDOW, bench.PENALTY, params, device=0) for _ in range(2)]
sig, off, a_s, a_e, bc, max_len = engs[0].synth_packed(spec, 0, n)
torch.cuda.synchronize()


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3


res_all = engs[0].demux(sig, a_s, a_e, offsets=off, max_len=max_len)
t_one = timed(lambda: engs[0].demux(sig, a_s, a_e, offsets=off, max_len=max_len, out=res_all))
print(f"{n} reads, one chain: {t_one:.2f} ms  {n / t_one / 1e3:.2f} M reads/s")
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
cuts = [n * i // pieces for i in range(pieces + 1)]
outs = [None] * pieces


def split():
    for i in range(pieces):
        lo, hi = cuts[i], cuts[i + 1]
        with torch.cuda.stream(streams[i % 2]):
            outs[i] = engs[i % 2].demux(sig, a_s[lo:hi], a_e[lo:hi], offsets=off[lo:hi + 1], max_len=max_len, out=outs[i])


split()
torch.cuda.synchronize()
t_two = timed(split)
print(f"{pieces} pieces on two streams: {t_two:.2f} ms  {n / t_two / 1e3:.2f} M reads/s")
call = torch.cat([o.call for o in outs])
print("same calls:", bool(torch.equal(call, res_all.call)))

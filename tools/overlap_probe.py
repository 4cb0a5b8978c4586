"""Does the chain gain from two halves of a batch in flight on two streams (clip / main kernel of one half beside the
DTW kernel of the other)?  tools/overlap_probe.py [n_reads] [reps]
  A: one wdx_demux_dev call over n reads
  B: two calls over the halves, one stream
  C: two calls over the halves, two contexts on two streams, enqueued from two host threads"""
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from warpdemux_amd import sig_proc, synth  # noqa: E402
from warpdemux_amd.engine import DemuxEngine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
spec = synth.SynthSpec(n_barcodes=10)
rng = np.random.default_rng(5)
refs = rng.normal(size=(10, 110))
params = sig_proc.SegParams(barcode_num_events=110)
e1 = DemuxEngine(refs, 15, 0.1, params, device=0)
e2 = DemuxEngine(refs, 15, 0.1, params, device=0)
sig, off, a_s, a_e, bc, max_len = e1.synth_packed(spec, 0, n)
h = n // 2
halves = [(off[:h + 1], a_s[:h], a_e[:h]), (off[h:], a_s[h:], a_e[h:])]
full = e1.demux(sig, a_s, a_e, offsets=off, max_len=max_len)
r1 = e1.demux(sig, halves[0][1], halves[0][2], offsets=halves[0][0], max_len=max_len)
r2 = e2.demux(sig, halves[1][1], halves[1][2], offsets=halves[1][0], max_len=max_len)
r1b = e1.demux(sig, halves[1][1], halves[1][2], offsets=halves[1][0], max_len=max_len)
torch.cuda.synchronize()
assert torch.equal(full.call[:h], r1.call) and torch.equal(full.call[h:], r2.call)


def timed(f):
    f()
    torch.cuda.synchronize()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        torch.cuda.synchronize()
        t.append(time.perf_counter() - t0)
    return min(t) * 1e3


def A():
    e1.demux(sig, a_s, a_e, offsets=off, max_len=max_len, out=full)


def B():
    e1.demux(sig, halves[0][1], halves[0][2], offsets=halves[0][0], max_len=max_len, out=r1)
    e1.demux(sig, halves[1][1], halves[1][2], offsets=halves[1][0], max_len=max_len, out=r1b)


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def C():
    def run(e, s, hv, out):
        with torch.cuda.stream(s):
            e.demux(sig, hv[1], hv[2], offsets=hv[0], max_len=max_len, out=out)
    th = [threading.Thread(target=run, args=(e1, s1, halves[0], r1)), threading.Thread(target=run, args=(e2, s2, halves[1], r2))]
    for t in th:
        t.start()
    for t in th:
        t.join()


for name, f in (("A one call", A), ("B two halves, one stream", B), ("C two halves, two streams", C), ("A one call", A)):
    ms = timed(f)
    print(f"{name:28s} {ms:8.2f} ms  {n / ms / 1e3:7.2f} M reads/s")

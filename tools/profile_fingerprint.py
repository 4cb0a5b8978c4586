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
fast = int(sys.argv[2]) if len(sys.argv) > 2 else 1
K = 110
spec = synth.SynthSpec(n_barcodes=10)
eng = DemuxEngine(np.zeros((10, K)), 15, 0.1, sig_proc.SegParams(barcode_num_events=K))
sig, off, a_s, a_e, bc, max_len = eng.synth_packed(spec, 0, n)
status = torch.empty(n, dtype=torch.int32, device="cuda")
prof = torch.zeros((n, 32), dtype=torch.int64, device="cuda")
pc = eng.params.to_c()
for _ in range(2):
    _lib.check(eng.L.wdx_fingerprint_profile_dev(eng.ctx.handle, _dp(sig), _dp(off), 0, max_len, n, _dp(a_s), _dp(a_e),
                                                 C.byref(pc), _dp(status), _dp(prof), n, fast, 0, None))
torch.cuda.synchronize()
p = prof.cpu().numpy()
ok = status.cpu().numpy() == 0
declined = int(p[0, 15]) if fast else 0
if fast == 2:
    # the split pair (round 6): tile kernel slots 0 / 3 / 4 / 5, tail kernel slots 6 .. 9 (two kernels: two clocks' worth of
    # differences, never across the pair)
    q = p[(p[:, 9] != 0) & (p[:, 5] != 0)]
    print(f"split pair, {q.shape[0]} of {n} reads finished by it (the others: longer windows, hand-overs)")
    tile_tot = q[:, 5] - q[:, 0]
    tail_tot = q[:, 9] - q[:, 6]
    rows = [("tile kernel: P0 load + clip", q[:, 3] - q[:, 0], tile_tot), ("tile kernel: P2+P3a t-score tiles + maxima", q[:, 4] - q[:, 3], tile_tot),
            ("tile kernel: export (pick, half-chunk prefixes, P at the peaks)", q[:, 5] - q[:, 4], tile_tot),
            ("tail kernel: window, suppression, top-E, boundaries", q[:, 7] - q[:, 6], tail_tot),
            ("tail kernel: event means, mean / sd", q[:, 8] - q[:, 7], tail_tot), ("tail kernel: normalise, output", q[:, 9] - q[:, 8], tail_tot)]
    for name, dd, tot in rows:
        print(f"  {name:64s} median {np.median(dd):8.0f}  mean {dd.mean():9.0f}  share of its kernel {dd.sum() / tot.sum() * 100:5.1f}%")
    print(f"  tile kernel: a workgroup's cycles median {np.median(tile_tot):.0f} mean {tile_tot.mean():.0f} (one-piece kernel: ~42 500); "
          f"tail kernel: a wave's cycles median {np.median(tail_tot):.0f} mean {tail_tot.mean():.0f}; exported peaks median {np.median(q[:, 12]):.0f} max {q[:, 12].max():.0f}")
    sys.exit(0)
if fast:
    dec = p[p[:, 9] == 0]
    print("handed on by the main kernel (0 window beyond its capacity / parameter gate, 1 NaN, 2 sums not provably exact, "
          "3 plateau / peak-list capacity, 4 nbr, 5 kept-list capacity, 6 tie at the top-E cut, 7 doubt -> exact-scores retry):",
          np.bincount(dec[:, 13].astype(int), minlength=8).tolist())
if fast:
    names = ["P0 load (+ clip: large batches)", "P1a median (small batches)", "P1b MAD+clip (small batches)",
             "P2+P3a t-score+maxima", "P3b suppression", "P4 top-E", "P5 boundaries", "P6 event means", "P7 normalise"]
    p = p[p[:, 9] != 0]   # reads the fast kernel finished (declined reads carry no end stamp)
else:
    names = ["P0 load", "P1 median+clip", "P2 t-score", "P3a local maxima", "P3b suppression", "P4 top-E",
             "P5 boundaries", "P6 event means", "P7 normalise/stats"]
d = np.diff(p[:, :10], axis=1)
tot = p[:, 9] - p[:, 0]
print(f"fast={fast} declined_to_slow={declined} profiled={p.shape[0]}")
print(f"reads {ok.sum()}  max_len {max_len}  median total cycles/read {np.median(tot):.0f}  mean {tot.mean():.0f}")
for i in range(9):
    print(f"  {names[i]:22s} median {np.median(d[:, i]):9.0f}  mean {d[:, i].mean():9.0f}  share {d[:, i].sum() / tot.sum() * 100:5.1f}%")
print("  suppression iterations: median %d  p99 %d  max %d" % (np.median(p[:, 10]), np.percentile(p[:, 10], 99), p[:, 10].max()))
print("  n samples median %d, score positions median %d" % (np.median(p[:, 11]), np.median(p[:, 12])))
if fast:
    fb = p[:, 25]
    print("  wave tiles that fell back to exact scores: %.2f per read on average (%.1f %% of the reads have one; max %d)"
          % (fb.mean(), 100.0 * (fb > 0).mean(), fb.max()))
    print("  score mode of the reads the main kernel finished (1 = approximate keys, 2 = exact scores):",
          np.bincount(p[:, 14].astype(int), minlength=3).tolist())

if fast and (p[:, 16] != 0).any():   # (in-kernel medians: small batches only)
    q = p[:, 16:23]
    lab = ["minmax", "zero+hist", "scan+locate", "gather", "rank", "evenfix"]
    base = p[:, 1]
    prev = base
    print("  median-1 internals (cycles, median over reads):")
    for i, name in enumerate(lab):
        cur = q[:, i]
        okm = cur > 0
        print(f"    {name:12s} {np.median((cur - prev)[okm]):8.0f}   (n={okm.sum()})")
        prev = np.where(okm, cur, prev)
    print("    small-list size median %d p99 %d" % (np.median(q[:, 6]), np.percentile(q[:, 6], 99)))

if fast and (p[:, 26] != 0).any():
    c = p[p[:, 24] != 0]
    order = [26, 27, 28, 29, 30, 31, 23, 24]
    lab = ["load + extremes", "median: zero + histogram", "median: find the bin", "median: gather + rank (+ even n)",
           "keys of |x - med|", "MAD: zero + histogram", "MAD: find bin + gather + rank"]
    print("  clip_bounds_kernel, one wave per read (100 MHz ticks: x24 = shader cycles at 2.4 GHz; median / mean over %d reads):" % len(c))
    for i, name in enumerate(lab):
        dd = c[:, order[i + 1]] - c[:, order[i]]
        print(f"    {name:34s} {np.median(dd):8.0f} {dd.mean():9.1f}")
    tt = c[:, 24] - c[:, 26]
    print(f"    {'total':34s} {np.median(tt):8.0f} {tt.mean():9.1f}")

if fast and len(sys.argv) > 3:
    # ablation: throughput of the kernel cut after phase k (no stamps: prof_reads = 0), steady state
    nbig = int(sys.argv[3])
    sig, off, a_s, a_e, bc, max_len = eng.synth_packed(spec, 0, nbig)
    status = torch.empty(nbig, dtype=torch.int32, device="cuda")
    prof1 = torch.zeros((1, 32), dtype=torch.int64, device="cuda")
    # large batches run the EXT instantiation (clip bounds from clip_bounds_kernel, no medians in the kernel): its cuts are
    # P0 (load + clip, stamp 3) and the phases behind it; the launch under test is the main kernel ALONE (stop_phase > 0
    # switches the rest of the chain off), clip_bounds_kernel runs ahead of it and is listed first
    names = {3: "P0 load + clip", 4: "P2+P3a t-score+maxima", 5: "P3b suppression", 6: "P4 top-E", 7: "P5 boundaries",
             8: "P6 event means", 9: "P7 normalise"}
    prev = 0.0
    print(f"ablation on {nbig} reads (ms per launch of clip_bounds_kernel + the main kernel cut after the phase, cumulative / marginal):")
    for k in range(3, 10):
        ts = []
        for rep in range(int(os.environ.get("WDX_PROF_REPS", "3"))):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _lib.check(eng.L.wdx_fingerprint_profile_dev(eng.ctx.handle, _dp(sig), _dp(off), 0, max_len, nbig, _dp(a_s),
                                                         _dp(a_e), C.byref(pc), _dp(status), _dp(prof1), 0, 1, k, None))
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        t = min(ts)
        print(f"  after {names[k]:24s} {t:8.2f}  +{t - prev:7.2f}")
        prev = t

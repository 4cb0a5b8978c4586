#!/usr/bin/env python3
"""Device time of the exact general kernel on the tRNA flow (120 events, d = 9, W = 18, consensus-guided refinement,
rna004_130bps@v1.0_tRNA.toml) against the same reads without the refinement branch.  HIP-event time of the
fingerprint launches (WDX_K_FINGERPRINT), host copies excluded.   python tools/bench_refine.py [n_reads]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from warpdemux_amd import _lib, sig_proc  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
consensus = np.load(os.path.join(ROOT, "tests", "golden", "g8_refine.npz"))["consensus"]
rng = np.random.default_rng(3)
rows = []
for i in range(n):
    n_lead = int(rng.integers(2, 34))
    lv = list(rng.normal(0, 1, n_lead)) + list(consensus) + list(rng.normal(0, 1, 30))
    lv = np.array(lv) * 12.0 + 85.0
    dw = rng.integers(12, 60, lv.size)
    rows.append((np.repeat(lv, dw) + rng.normal(0, 1.5, int(dw.sum()))).astype(np.float32))
stride = max(r.size for r in rows)
mb = np.full((n, stride), np.nan, dtype=np.float32)
for i, r in enumerate(rows):
    mb[i, : r.size] = r
a_s = np.full(n, 100, dtype=np.int32)
a_e = np.array([r.size - 100 for r in rows], dtype=np.int32)
seg = sig_proc.SegParams(min_obs_per_base=9, running_stat_width=18, num_events=120, barcode_num_events=25)
ref = sig_proc.RefineParams(query=consensus, barcode_segm_events=25, barcode_keep_events=25)
L = _lib.load()
ctx = _lib.default_context(None)


def timed(fn, reps=5):
    fn()
    _lib.check(L.wdx_kernel_time_reset(ctx.handle))
    _lib.check(L.wdx_kernel_timing(ctx.handle, 1))
    for _ in range(reps):
        out = fn()
    _lib.check(L.wdx_kernel_timing(ctx.handle, 0))
    ms, k = C.c_double(0), C.c_int64(0)
    _lib.check(L.wdx_kernel_time(ctx.handle, _lib.K_FINGERPRINT, C.byref(ms), C.byref(k)))
    return ms.value / reps, out


if os.environ.get("WDX_DEBUG_OCC"):
    ctx.set_option(_lib.OPT_DEBUG_OCCUPANCY, 1)
t_plain, fb = timed(lambda: sig_proc.fingerprint_batch(mb, a_s, a_e, seg))
t_ref, fr = timed(lambda: sig_proc.fingerprint_refine_batch(mb, a_s, a_e, seg, ref))
print(f"{n} reads, mean window {float((a_e - a_s + 200).mean()):.0f} samples")
print(f"  plain  (120, 9, 18): {t_plain:8.2f} ms  {n / t_plain / 1e3:7.3f} M reads/s   ok {int((fb.status == 0).sum())}")
print(f"  refine (tRNA flow) : {t_ref:8.2f} ms  {n / t_ref / 1e3:7.3f} M reads/s   ok {int((fr.status == 0).sum())}, "
      f"outliers {int((fr.status == 6).sum())}")

"""Development aid: the launch chain's hand-over counts and reasons (WDX_OPT_DEBUG_OCCUPANCY) for clip-heavy reads with
many level changes on two parameter triples.  python tools/dev/slowprobe.py"""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from warpdemux_amd import _lib, sig_proc  # noqa: E402

rng = np.random.default_rng(97)
consensus = np.load(__file__.rsplit("/tools/", 1)[0] + "/tests/golden/g8_refine.npz")["consensus"]
n = 4096
mb = np.full((n, 6144), np.nan, dtype=np.float32)
a_e = np.zeros(n, dtype=np.int32)
for i in range(n):
    lv = np.concatenate([rng.normal(0, 1, int(rng.integers(2, 34))), consensus, rng.normal(0, 1, 30)]) * 12.0 + 85.0
    dw = rng.integers(12, 60, lv.size)
    x = (np.repeat(lv, dw) + rng.normal(0, 1.5, int(dw.sum())))[:6100]
    mb[i, :x.size] = x
    a_e[i] = x.size
a_s = np.zeros(n, dtype=np.int32)
ctx = _lib.default_context(None)
ctx.set_option(_lib.OPT_DEBUG_OCCUPANCY, 1)
for E, d, w in ((110, 6, 12), (120, 9, 18)):
    print("triple", (E, d, w), file=sys.stderr, flush=True)
    fb = sig_proc.fingerprint_batch(mb, a_s, a_e, sig_proc.SegParams(padding=0, num_events=E, min_obs_per_base=d, running_stat_width=w))
    print("   ok", int((fb.status == 0).sum()), "of", n, file=sys.stderr, flush=True)

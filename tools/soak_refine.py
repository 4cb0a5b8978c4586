#!/usr/bin/env python3
"""Randomised soak of the refinement flow ON THE LAUNCH CHAIN (fast kernels -> fingerprint_refine_match_wave_kernel ->
fingerprint_refine_tail_wave_kernel, exact kernel for what they hand on): random consensus lengths, event counts,
relaxations, penalties, normalisations, barcode event counts and tail lengths per draw, 512 reads per draw, every output
against the oracle bit for bit.   python tools/soak_refine.py [draws] [seed0]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import wdx_oracle as orc  # noqa: E402
from warpdemux_amd import sig_proc  # noqa: E402


def same(a, b):
    return a.shape == b.shape and np.array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(b).view(np.uint8))


def main():
    draws = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    t0 = time.time()
    hist = np.zeros(8, dtype=np.int64)
    for k in range(draws):
        rng = np.random.default_rng(seed0 + k)
        w, dmax = [(12, 6), (18, 9), (30, 15)][int(rng.integers(0, 3))]
        d = int(rng.integers(max(2, dmax - 3), dmax + 1))
        nq = int(rng.integers(4, 97))
        E = int(rng.integers(nq + 6, 127))
        E2 = int(rng.integers(3, 90))
        keep = int(rng.integers(1, E2 + 2))
        coarse = rng.random() < 0.3
        query = rng.normal(0, 1, nq)
        if coarse:
            query = np.round(query * 2) / 2
        n = 512
        rows = []
        for i in range(n):
            n_lead = int(rng.integers(0, max(1, E - nq - 4)))
            n_tail = int(rng.integers(max(4, E2 - 6), E2 + 30))
            body = query + rng.normal(0, 0.05, nq) if (i % 4 and not coarse) else (query if i % 4 else rng.normal(0, 1, nq))
            lv = np.concatenate([rng.normal(0, 1, n_lead), body, rng.normal(0, 1, n_tail)]) * 12.0 + 85.0
            dw = rng.integers(2 * d + 1, 4 * d + 10, lv.size)
            x = np.repeat(lv, dw) + rng.normal(0, rng.uniform(0.5, 2.5), int(dw.sum()))
            if coarse:
                x = np.round(x * 2) / 2
            rows.append(x.astype(np.float32)[:15000])
        stride = max(r.size for r in rows)
        mb = np.full((n, stride), np.nan, dtype=np.float32)
        for i, r in enumerate(rows):
            mb[i, : r.size] = r
        pad = int(rng.choice([0, 50]))
        a_s = np.full(n, pad, dtype=np.int32)
        a_e = np.array([r.size - pad for r in rows], dtype=np.int32)
        seg = dict(padding=pad, min_obs_per_base=d, running_stat_width=w, num_events=E, seg_norm=str(rng.choice(["mean", "median"])),
                   outlier_thresh=float(rng.choice([3.0, 5.0])))
        ref = dict(subseq_norm=str(rng.choice(["mean", "median", "none"])), penalty=float(rng.choice([0.0, 0.5, 1.5, 3.0])),
                   psi=(int(rng.integers(0, 8)), 0, int(rng.integers(0, 60)), 0), ub_start=int(rng.integers(5, 120)),
                   lb_end=int(rng.integers(0, nq)), ub_end=int(rng.integers(nq, 160)), barcode_segm_events=E2, barcode_keep_events=keep)
        fb = sig_proc.fingerprint_refine_batch(mb, a_s, a_e, sig_proc.SegParams(barcode_num_events=keep, **seg),
                                               sig_proc.RefineParams(query=query, **ref))
        fpt, dwell, stats, idx, status = orc.fingerprint_refine_batch(mb, a_s, a_e, orc.SegParams(barcode_num_events=keep, **seg),
                                                                      orc.RefineParams(query=query, **ref))
        good, rep = status == 0, (status == 0) | (status == 6)
        ok = (np.array_equal(fb.status, status) and same(fb.fpt[good], fpt[good]) and same(fb.dwell[good], dwell[good]) and
              same(fb.stats[rep], stats[rep]) and same(fb.refine_idx[rep], idx[rep]))
        hist += np.bincount(status, minlength=8)[:8]
        if not ok:
            print(f"MISMATCH at draw {seed0 + k}: w={w} d={d} nq={nq} E={E} E2={E2} keep={keep} coarse={coarse}", flush=True)
            return 1
    print(f"{draws} draws x 512 reads (seeds {seed0}..{seed0 + draws - 1}): every status, fingerprint, dwell, statistic and index "
          f"bitwise equal to the oracle; status histogram {hist.tolist()}; {time.time() - t0:.0f} s")
    return 0


if __name__ == "__main__":
    sys.exit(main())

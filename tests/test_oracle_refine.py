"""N3 consensus-guided refinement (sig_proc.py:257-378, 452-521): oracle against fixture g8.

g8 comes from the REFERENCE's own detect_results_to_fpt run with consensus_refinement=True
(tests/golden/make_golden_refine.py); only the dtaidistance call inside `_get_subseq_match` is a pure-Python
stand-in there (library absent), so the subsequence match is cross-checked between two independent restatements
but stays parity-unpinned against the real library; everything around it is pinned to the reference."""
import os

import numpy as np

from oracle import wdx_oracle as orc

INV = {0: "none", 1: "mean", 2: "median"}


def params_from(g, k):
    pad, d, w, E, seg_norm, e2, keep, sub_norm, p0, p1, p2, p3, ub_s, lb_e, ub_e = (int(v) for v in g[f"seg_{k}"])
    thr, pen = (float(v) for v in g[f"fl_{k}"])
    seg = dict(padding=pad, min_obs_per_base=d, running_stat_width=w, num_events=E, seg_norm=INV[seg_norm],
               outlier_thresh=thr, barcode_num_events=keep)
    ref = dict(subseq_norm=INV[sub_norm], penalty=pen, psi=(p0, p1, p2, p3), ub_start=ub_s, lb_end=lb_e, ub_end=ub_e,
               barcode_segm_events=e2, barcode_keep_events=keep)
    return seg, ref


def _same(a, b):
    return np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)


def test_g8_refinement_end_to_end(golden_dir):
    g = np.load(os.path.join(golden_dir, "g8_refine.npz"))
    seen = set()
    for k in range(int(g["n"])):
        seg, ref = params_from(g, k)
        row = g[f"row_{k}"]
        a_s, a_e = (int(v) for v in g[f"args_{k}"])
        fpt, dwell, stats, idx, status = orc.fingerprint_refine_batch(
            row.reshape(1, -1), [a_s], [a_e], orc.SegParams(**seg), orc.RefineParams(query=g["consensus"], **ref))
        st = int(g[f"status_{k}"])
        tag = str(g[f"tag_{k}"])
        assert status[0] == st, f"case {k} ({tag}): status {status[0]} != {st}"
        seen.add(st)
        if st in (0, 6):
            assert _same(stats[0], g[f"stats_{k}"]), f"case {k} ({tag}) stats"
            assert _same(idx[0], g[f"idx_{k}"]), f"case {k} ({tag}) query start/end, barcode start"
        if st == 0:
            assert _same(fpt[0], g[f"fpt_{k}"]), f"case {k} ({tag}) fpt"
            assert _same(dwell[0], g[f"dwell_{k}"]), f"case {k} ({tag}) dwell"
    assert {0, 3, 5, 6} <= seen


def test_subseq_match_properties():
    """The unpinned piece on its own: invariants any faithful subsequence DTW must satisfy."""
    rng = np.random.default_rng(3)
    q = rng.normal(size=40)
    # the query embedded verbatim: perfect match, ends exactly where the copy ends, starts where it starts
    for off in (0, 3, 17, 40):
        s = np.concatenate([rng.normal(size=off) + 5.0, q, rng.normal(size=25) + 5.0])
        st, en = orc.subseq_match(q, s, penalty=1.5, psi=(0, 0, len(s), 0))
        assert (st, en) == (off, off + q.size - 1)
    # without series-begin relaxation the match is anchored at column 0
    s = np.concatenate([rng.normal(size=10) + 5.0, q, rng.normal(size=10) + 5.0])
    st, en = orc.subseq_match(q, s, penalty=0.0, psi=(0, 0, 0, 0))
    assert st == 0
    # psi on the query's beginning lets the match skip query points: a series that starts inside the query
    st, en = orc.subseq_match(q, q[4:], penalty=1.5, psi=(5, 0, 0, 0))
    assert (st, en) == (0, q.size - 5)
    # shifting both by a constant changes nothing; scaling the penalty to zero never lengthens the distance
    s = np.concatenate([rng.normal(size=12), q + 0.05 * rng.normal(size=q.size), rng.normal(size=12)])
    assert orc.subseq_match(q, s) == orc.subseq_match(q + 2.5, s + 2.5)


def test_subseq_match_hand_derived_cases():
    """Small cases worked out by hand from dtaidistance's published algorithm (warping_paths with penalty^2 on the
    two non-diagonal steps and psi relaxations, SubsequenceAlignment.best_match = argmin of the last row / len(query),
    best_path walking back over the sqrt'ed matrix with np.argmin([diag, up, left]) -- first minimum wins), as
    sig_proc.py:287-306 calls it.  D below is the squared accumulated cost, rows = query, columns = series (1-based).

    1. psi at the series' beginning.  q = [1,2,3], s = [9,9,1,2,3], penalty 0.  With psi_series_begin = 1 only
       D(0,0) = D(0,1) = 0: row 1 = [64,64,64,65,69], row 2 = [113,113,65,64,65], row 3 = [149,149,69,65,64]
       -> end index 4; back-trace (3,5) diag (2,4) diag (1,3) left (1,2) diag (0,1): first series column 2
       -> start index 1.  With psi_series_begin >= 2 the copy is matched for free: (2, 4).
    2. tie-break.  q = [0,1], s = [0,0.3,1], penalty 0, series fully relaxed: row 1 = [0,0.09,1], row 2 =
       [1,0.49,0.09] -> end index 2; (2,3) diag (1,2), whose three predecessors D(0,1) = D(0,2) = D(1,1) = 0 tie:
       argmin takes the diagonal -> start index 1 (a walk preferring `left` would reach start index 0).
    3. penalty is SQUARED.  q = [0,1,1], s = [5,0,1,5], series fully relaxed.  penalty 0: last row [57,2,0,16]
       -> end 2, path (3,3) up (2,3) diag (1,2) diag -> (1, 2).  penalty 5 (25 per non-diagonal step): rows
       [25,0,1,25], [66,26,0,17], [107,52,25,16] -> end 3, three diagonal steps -> (1, 3); an un-squared penalty
       (5 per step) would leave the vertical step at cost 5 < 16 and return (1, 2)."""
    assert orc.subseq_match([1, 2, 3], [9, 9, 1, 2, 3], penalty=0.0, psi=(0, 0, 1, 0)) == (1, 4)
    assert orc.subseq_match([1, 2, 3], [9, 9, 1, 2, 3], penalty=0.0, psi=(0, 0, 2, 0)) == (2, 4)
    assert orc.subseq_match([1, 2, 3], [9, 9, 1, 2, 3], penalty=0.0, psi=(0, 0, 5, 0)) == (2, 4)
    assert orc.subseq_match([0, 1], [0, 0.3, 1], penalty=0.0, psi=(0, 0, 3, 0)) == (1, 2)
    assert orc.subseq_match([0, 1, 1], [5, 0, 1, 5], penalty=0.0, psi=(0, 0, 4, 0)) == (1, 2)
    assert orc.subseq_match([0, 1, 1], [5, 0, 1, 5], penalty=5.0, psi=(0, 0, 4, 0)) == (1, 3)
    assert orc.subseq_match([0, 1, 1], [5, 0, 1, 5], penalty=np.sqrt(5.0), psi=(0, 0, 4, 0)) == (1, 2)   # 5 per step: the un-squared reading

"""Golden vectors of the consensus-refinement branch (sig_proc.py:257-378, 452-521).

Runs only in the build container.  The reference's own `detect_results_to_fpt` is executed with
`segmentation.consensus_refinement = True` -- its segmentation, re-segmentation of the score tail,
`normalize_wrt`, stats and outlier filter are the REFERENCE's code.  The one piece it cannot run here is the
dtaidistance call inside `_get_subseq_match` (library absent): `warping_paths_fast` and `SubsequenceAlignment`
are supplied by the small pure-Python stand-ins below, written from the library's published algorithm
(dtaidistance 2.3.x dtw.warping_paths / dtw.best_path / subsequence.dtw.SubsequenceAlignment) and independent of
oracle/wdx_oracle.c -- so the fixture cross-checks the oracle's restatement, but the subsequence match itself
stays PARITY UNPINNED against the real library.

    python tests/golden/make_golden_refine.py      # writes tests/golden/g8_refine.npz
"""
from __future__ import annotations

import os
import sys
import warnings
from types import SimpleNamespace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import REF, ROOT, import_reference, status_code  # noqa: E402


# ---- stand-ins for the two dtaidistance names sig_proc.py binds at import (sig_proc.py:10-12) --------------
def warping_paths_fast(s1, s2, penalty=None, psi=None, compact=False, psi_neg=True, window=None, **kw):
    assert not compact and not kw and window is None
    r, c = len(s1), len(s2)
    p1b, p1e, p2b, p2e = (int(v) for v in psi)
    pen = 0.0 if penalty is None else float(penalty) ** 2
    dtw = np.full((r + 1, c + 1), np.inf)
    dtw[0, : p2b + 1] = 0
    dtw[: p1b + 1, 0] = 0
    for i in range(r):
        for j in range(c):
            d = (s1[i] - s2[j]) ** 2
            dtw[i + 1, j + 1] = d + min(dtw[i, j], dtw[i, j + 1] + pen, dtw[i + 1, j] + pen)
    dtw = np.sqrt(dtw)
    return float(dtw[r, c]), dtw


class SubsequenceAlignment:
    def __init__(self, query, series, penalty=0.1, use_c=False):
        self.query, self.series, self.paths, self.matching = query, series, None, None

    def _compute_matching(self):
        matching = self.paths[-1, :]
        if len(matching) > len(self.series):
            matching = matching[-len(self.series):]
        self.matching = np.array(matching) / len(self.query)

    def best_match(self):
        idx = int(np.argmin(self.matching))
        paths = self.paths
        i, j = paths.shape[0] - 1, idx + 1
        p = [(i - 1, j - 1)]
        while i > 0 and j > 0:
            c = int(np.argmin([paths[i - 1, j - 1], paths[i - 1, j], paths[i, j - 1]]))
            if c == 0:
                i, j = i - 1, j - 1
            elif c == 1:
                i = i - 1
            else:
                j = j - 1
            p.append((i - 1, j - 1))
        p.pop()
        p.reverse()
        return SimpleNamespace(segment=[p[0][1], idx])


def make_spc(E=120, d=9, w=18, seg_norm="mean", bne=(25, 25), sub_norm="mean", penalty=1.5, psi=(5, 0, 40, 0),
             ub_start=18, lb_end=69, ub_end=97, thresh=5.0, padding=100):
    return SimpleNamespace(
        sig_extract=SimpleNamespace(padding=padding, normalization="none"),
        core=SimpleNamespace(sig_norm_outlier_thresh=thresh),
        segmentation=SimpleNamespace(
            num_events=E, min_obs_per_base=d, running_stat_width=w, accept_less_cpts=False,
            consensus_refinement=True, consensus_model="rna004_130bps_v1_0", normalization=seg_norm,
            barcode_num_events=list(bne), consensus_subseq_match_normalization=sub_norm,
            consensus_subseq_match_penalty=penalty, consensus_subseq_match_psi=list(psi),
            consensus_subseq_match_ub_start=ub_start, consensus_subseq_match_lb_end=lb_end,
            consensus_subseq_match_ub_end=ub_end, refinement_optimal_cpts=False,
        ),
    )


def main():
    sys.path.insert(0, ROOT)
    from warpdemux_amd import synth

    sp, DetectResults, _, _ = import_reference()
    sp.warping_paths_fast = warping_paths_fast
    sp.SubsequenceAlignment = SubsequenceAlignment
    sys.path.insert(0, REF)
    import importlib.util

    spec_ = importlib.util.spec_from_file_location("wdx_consensus", os.path.join(REF, "warpdemux", "_consensus.py"))
    cm = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(cm)
    consensus = np.asarray(cm.ALL["rna004_130bps_v1_0"], dtype=np.float64)   # 84-point adapter consensus (data)

    rng = np.random.Generator(np.random.PCG64(20250202))
    g = {"consensus": consensus}
    k = 0

    def make_read(seed, n_lead=70, embed=True, noise=1.5, dwell_lo=14, dwell_hi=60):
        """an adapter whose event levels follow: random leader | the consensus shape | 30 barcode events"""
        r = np.random.Generator(np.random.PCG64(seed))
        lv = list(r.normal(0, 1, n_lead))
        lv += list(consensus if embed else r.normal(0, 1, consensus.size))
        lv += list(r.normal(0, 1, 30))
        lv = np.array(lv) * 12.0 + 85.0
        dw = r.integers(dwell_lo, dwell_hi, lv.size)
        x = np.repeat(lv, dw) + r.normal(0, noise, int(dw.sum()))
        return x.astype(np.float32)

    def run_case(row, a_start, a_end, tag, **kw):
        nonlocal k
        spc = make_spc(**kw)
        dr = DetectResults(success=True, fail_reason="", adapter_start=a_start, adapter_end=a_end)
        work = np.array(row, dtype=np.float32, copy=True)
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                res = sp.detect_results_to_fpt(work, spc, dr, consensus)
            st = status_code(res) if res.fail_reason != "consensus query outlier" else 6
        except Exception:
            res, st = None, 5
        K = spc.segmentation.barcode_num_events[1]
        fpt, dwell, stats, idx = np.full(K, np.nan), np.zeros(K, np.int64), np.full(6, np.nan), np.full(3, -1, np.int64)
        if st in (0, 6):
            stats[:] = [res.adapter_dt_med, res.adapter_dt_mad, res.adapter_event_mean, res.adapter_event_std,
                        res.adapter_event_med, res.adapter_event_mad]
            idx[:] = [res.seg_cons_query_start, res.seg_cons_query_end, res.sig_barcode_start]
        if st == 0:
            fpt[:] = res.barcode_fpt
            dwell[:] = res.dwell_times
        s = spc.segmentation
        g[f"row_{k}"] = np.array(row, dtype=np.float32)
        g[f"args_{k}"] = np.array([a_start, a_end], dtype=np.int64)
        g[f"seg_{k}"] = np.array([spc.sig_extract.padding, s.min_obs_per_base, s.running_stat_width, s.num_events,
                                  {"none": 0, "mean": 1, "median": 2}[s.normalization], s.barcode_num_events[0],
                                  s.barcode_num_events[1], {"none": 0, "mean": 1, "median": 2}[s.consensus_subseq_match_normalization],
                                  *[int(v) for v in s.consensus_subseq_match_psi], s.consensus_subseq_match_ub_start,
                                  s.consensus_subseq_match_lb_end, s.consensus_subseq_match_ub_end], dtype=np.int64)
        g[f"fl_{k}"] = np.array([spc.core.sig_norm_outlier_thresh, s.consensus_subseq_match_penalty], dtype=np.float64)
        g[f"status_{k}"] = np.int64(st)
        g[f"fpt_{k}"], g[f"dwell_{k}"], g[f"stats_{k}"], g[f"idx_{k}"] = fpt, dwell, stats, idx
        g[f"tag_{k}"] = np.array(tag)
        k += 1
        return st

    sts = []
    for i in range(16):
        x = make_read(1000 + i, n_lead=int(rng.integers(2, 16)))
        sts.append(run_case(x, 100, x.size - 100, "embedded"))
    for i in range(6):   # consensus far into the read / absent: outliers of the filter
        x = make_read(2000 + i, n_lead=int(rng.integers(25, 40)))
        sts.append(run_case(x, 100, x.size - 100, "late_consensus"))
    for i in range(4):
        x = make_read(3000 + i, embed=False)
        sts.append(run_case(x, 100, x.size - 100, "no_consensus"))
    for i in range(4):
        x = make_read(4000 + i, n_lead=8)
        sts.append(run_case(x, 100, x.size - 100, "median_norms", seg_norm="median", sub_norm="median"))
    for i in range(3):
        x = make_read(5000 + i, n_lead=8)
        sts.append(run_case(x, 100, x.size - 100, "wide_filter_keep20", ub_start=60, lb_end=0, ub_end=200, bne=(25, 20)))
    for i in range(3):
        x = make_read(6000 + i, n_lead=8)
        sts.append(run_case(x, 100, x.size - 100, "penalty0_psi0", penalty=0.0, psi=(0, 0, 0, 0), ub_start=200, lb_end=0, ub_end=200))
    # short adapter: the window width shrinks below the configured one -> the tail's last boundary leaves the slice
    x = make_read(7000, n_lead=4, dwell_lo=8, dwell_hi=16)
    sts.append(run_case(x, 0, x.size, "shrunk_width", padding=0))
    # too few peaks in the adapter / in the barcode tail
    sts.append(run_case(rng.normal(90, 1, 3000).astype(np.float32), 0, 3000, "too_few_peaks", padding=0))
    x = make_read(8000, n_lead=8)
    sts.append(run_case(x[: x.size - 30 * 30], 100, x.size - 30 * 30 - 100, "short_tail", ub_start=60, lb_end=0, ub_end=200))
    g["n"] = np.int64(k)
    np.savez_compressed(os.path.join(HERE, "g8_refine.npz"), **g)
    print("G8 cases:", k, "status histogram:", {s: sts.count(s) for s in sorted(set(sts))})


if __name__ == "__main__":
    main()

"""Fixture g4b: detect_results_to_fpt of the REFERENCE on adapter windows of 9 000 .. 15 200 samples -- up to the
largest window the reference admits (max_obs_trace 15 000 + 2 x padding 100,
DEPRECATED/config_files/rna002_70bps@v0.4.4.toml:2) -- for the three shipped parameter triples
(num_events, min_obs_per_base, running_stat_width): RNA004 (110, 6, 12), RNA002 (110, 15, 30), tRNA (120, 9, 18).

Runs only in the build container (needs /root/reference); same recipe and record layout as make_golden.py's G4.

    python tests/golden/make_golden_long.py        # writes tests/golden/g4b_long_windows.npz
"""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import make_golden as mg  # noqa: E402


def long_row(rng, n, mean_dwell):
    """step signal with geometric-ish dwell times (>= 8 samples), white noise and a few flicker spikes"""
    levels, out = [], []
    while len(out) < n:
        d = 8 + int(rng.geometric(1.0 / (mean_dwell - 8)))
        lv = 80.0 + 15.0 * rng.normal()
        out.extend([lv] * d)
        levels.append(lv)
    s = np.array(out[:n]) + rng.normal(0, 2.0, n)
    idx = rng.integers(0, n, max(1, n // 1000))
    s[idx] += rng.choice([-60.0, 60.0], idx.size)
    return s.astype(np.float32)


def main():
    sp, DetectResults, _, _ = mg.import_reference()
    rng = np.random.Generator(np.random.PCG64(20261003))
    g, k = {}, 0

    def run_case(row, a_start, a_end, tag, **spc_kw):
        nonlocal k
        spc = mg.make_spc(**spc_kw)
        dr = DetectResults(success=True, fail_reason="", adapter_start=a_start, adapter_end=a_end)
        row_in = np.array(row, dtype=np.float32, copy=True)
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                res = sp.detect_results_to_fpt(row_in.copy(), spc, dr)
            st = mg.status_code(res)
        except Exception:  # barcode_fpt_wrapper -> "unknown"
            res, st = None, 5
        K = spc.segmentation.barcode_num_events
        fpt, dwell, stats = np.full(K, np.nan), np.zeros(K, dtype=np.int64), np.full(6, np.nan)
        if st == 0:
            fpt[:] = res.barcode_fpt
            dwell[:] = res.dwell_times
            stats[:] = [res.adapter_dt_med, res.adapter_dt_mad, res.adapter_event_mean, res.adapter_event_std,
                        res.adapter_event_med, res.adapter_event_mad]
        p = spc
        g[f"row_{k}"] = row_in
        g[f"args_{k}"] = np.array([a_start, a_end, 1], dtype=np.int64)
        g[f"params_{k}"] = np.array(
            [p.sig_extract.padding, {"none": 0, "mean": 1, "median": 2}[p.sig_extract.normalization],
             p.segmentation.min_obs_per_base, p.segmentation.running_stat_width, p.segmentation.num_events,
             int(p.segmentation.accept_less_cpts), {"none": 0, "mean": 1, "median": 2}[p.segmentation.normalization],
             K], dtype=np.int64)
        g[f"thresh_{k}"] = np.float64(p.core.sig_norm_outlier_thresh)
        g[f"clip64_{k}"] = np.int64(isinstance(p.core.sig_norm_outlier_thresh, np.float64))
        g[f"status_{k}"] = np.int64(st)
        g[f"fpt_{k}"], g[f"dwell_{k}"], g[f"stats_{k}"] = fpt, dwell, stats
        g[f"tag_{k}"] = np.array(tag)
        print(k, tag, "status", st)
        k += 1

    triples = {"rna004": dict(E=110, d=6, w=12), "rna002": dict(E=110, d=15, w=30), "trna": dict(E=120, d=9, w=18)}
    for n in (11201, 13000, 15200):
        row = long_row(rng, n, n / 135.0)
        for name, t in triples.items():
            run_case(row, 100, n - 100, f"{name}_{n}", **t)
    row = long_row(rng, 15200, 110.0)
    run_case(row, 100, 15100, "rna002_15200_K110", K=110, **triples["rna002"])
    run_case(row, 100, 15100, "rna002_15200_clip64", thresh=np.float64(2.7), **triples["rna002"])
    run_case(row, 100, 15100, "rna002_15200_signorm_mean", sig_norm="mean", **triples["rna002"])
    run_case(row, 100, 15100, "rna002_15200_segnorm_median", seg_norm="median", **triples["rna002"])
    nm = row.copy()
    nm[7000:7004] = np.nan
    run_case(nm, 100, 15100, "rna002_15200_nan_middle", **triples["rna002"])
    run_case(long_row(rng, 9000, 70.0), 100, 8900, "rna002_9000", **triples["rna002"])
    run_case(long_row(rng, 11200, 85.0), 100, 11100, "rna002_11200", **triples["rna002"])
    # the reference has no upper limit of its own: 15 200 is what its configs admit.  A flat, noise-only long window
    run_case((80 + rng.normal(0, 1, 15200)).astype(np.float32), 100, 15100, "rna002_15200_flat_noise", **triples["rna002"])
    g["n"] = np.int64(k)
    dst = os.path.join(HERE, "g4b_long_windows.npz")
    np.savez_compressed(dst, **g)
    print(dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()

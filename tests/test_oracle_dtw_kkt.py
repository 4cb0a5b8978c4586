"""Stage-B pin: the oracle DTW against the KKT conditions of the reference's five shipped DTW_SVM models.

The models were trained by libsvm on exp(-D) with D = dtaidistance's own distance matrix of `_X`
(models/dtw_svm.py:21-22, parallel_distances.py:59-67, 139-198), so their dual coefficients only satisfy
libsvm's optimality conditions (tests/helpers/kkt.py) under the distances the genuine library returned.
Fixture g9 (tests/golden/make_golden_kkt.py) holds the models' numbers; nothing here needs /root/reference.

What this pins, at which resolution (measured, `<model>__residuals` in g9):
* oracle DTW with the model's own (window=15, penalty=0.1): all ~23 000 free-vector equalities hold to
  5.3-5.6e-4 in each of the five models -- inside libsvm's eps = 1e-3, i.e. as well as the genuine
  distances themselves do;
* every structural alternative violates them by >= 17x that: un-squared penalty (0.57), penalty 0
  (0.10), no final sqrt (1.5), band one cell narrower / wider (0.05 / 0.04), window 5 (0.98), unbanded
  (0.03-0.43), penalty +-10 % (0.02);
* a uniform relative bias of 1e-4 moves the residual by ~2.5e-4, so systematic deviations above ~2e-4
  relative are excluded by the threshold alone; the least-squares FIT over the penalty and over a uniform scale
  (test_fitted_penalty_and_scale_are_the_models_own) puts the minimiser at (0.1 +- 0.05 %, 1 +- 2e-5);
  per-pair rounding-level differences (1e-7, float32 output) are below its resolution -- those are covered by
  the bit-exact HIP == oracle tests.  The claim, exactly: pinned to ~1e-4 by the shipped models (shape L = 25,
  window 15, penalty 0.1); bit-exact to the restatement; L = 110 by shared code.
"""
import os

import numpy as np
import pytest

from helpers import kkt
from oracle import wdx_oracle as orc

G9 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g9_kkt_models.npz")
G9B = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g9b_kkt_models_rna002.npz")
EPS = kkt.EPS_LIBSVM
RNA002 = ["WDX4", "WDX-DPC", "WDX6", "WDX8", "WDX10", "WDX12"]     # DEPRECATED/model_files/*_rna002_v0_4_4.joblib


@pytest.fixture(scope="module")
def g9():
    return np.load(G9)


@pytest.fixture(scope="module")
def g9b():
    return np.load(G9B)


def oracle_dtw(X, window, penalty):
    """all-vs-all oracle matrix, row blocks on the host cores (ctypes drops the GIL; WDX12-rna002 is 13 M pairs)"""
    from concurrent.futures import ThreadPoolExecutor

    with ThreadPoolExecutor(max_workers=min(os.cpu_count() or 1, 16)) as ex:
        return np.concatenate(list(ex.map(lambda a: orc.dtw_matrix(X[a:a + 128], X, window, penalty), range(0, X.shape[0], 128))))


@pytest.mark.parametrize("name", ["WDX4", "WDX4b", "WDX4c", "WDX6", "WDX10"])
def test_oracle_dtw_satisfies_the_shipped_models_kkt_conditions(g9, name):
    m = kkt.model_from_npz(g9, name)
    assert (m["window"], m["penalty"], m["gamma"], m["pwr_dist"]) == (15, 0.1, 1.0, 1)
    D = oracle_dtw(m["X"], m["window"], m["penalty"])
    r = kkt.kkt_residuals(D, m["n_support"], m["dual_coef"], m["intercept"], m["c_bound"], m["gamma"], m["pwr_dist"])
    assert r["n_free"] > 1800 and r["n_bound"] > 200 and r["n_zero"] > 1100
    assert r["free_max_abs"] < EPS, r          # 0 < alpha < C : y f(x) = 1 within libsvm's eps
    assert r["bound_max"] < EPS, r             # alpha = C     : y f(x) <= 1
    assert r["zero_min"] > -EPS, r             # alpha = 0     : y f(x) >= 1
    # and the numbers recorded when the fixture was made (same oracle, same model)
    ref = g9[f"{name}__residuals"][list(g9["variant_names"]).index("reference")]
    assert np.allclose([r["free_max_abs"], r["bound_max"], r["zero_min"]], ref, rtol=0, atol=1e-9)


@pytest.mark.parametrize("name", RNA002)
def test_oracle_dtw_satisfies_the_rna002_models_kkt_conditions(g9b, name):
    """The six rna002 v0.4.4 models the reference keeps under DEPRECATED/model_files (gamma 1.2, C 10, class_weight
    balanced; 851 .. 3 617 training fingerprints of 25 points, 5 .. 13 classes; WDX12 is the model of the reference's
    live run): ~60 000 more free-vector equalities and ~46 000 alpha = 0 inequalities that only the genuine library's
    distances satisfy.  With C = 10 no vector sits at its bound, so every support vector with alpha > 0 is an equality."""
    m = kkt.model_from_npz(g9b, name)
    assert (m["window"], m["penalty"], m["gamma"], m["pwr_dist"]) == (15, 0.1, 1.2, 1)
    D = oracle_dtw(m["X"], m["window"], m["penalty"])
    r = kkt.kkt_residuals(D, m["n_support"], m["dual_coef"], m["intercept"], m["c_bound"], m["gamma"], m["pwr_dist"])
    assert r["n_free"] > 2400 and r["n_zero"] > 1000 and r["n_bound"] == 0
    assert r["free_max_abs"] < EPS and r["zero_min"] > -EPS, r
    assert r["free_max_abs"] < 0.6 * EPS            # measured 5.4 .. 5.8e-4: libsvm's own stopping gap
    ref = g9b[f"{name}__residuals"][list(g9b["variant_names"]).index("reference")]
    assert np.allclose([r["free_max_abs"], r["zero_min"]], ref[[0, 2]], rtol=0, atol=1e-9)


def test_recorded_controls_of_the_rna002_models(g9b):
    """make_golden_kkt.py's residuals of every control on every rna002 model: >= 23 eps each (the worst: a band one cell
    wider on WDX6, 0.026), while the models' own parameters sit at 0.55 eps; a uniform 1e-4 scaling of the distances
    already costs 0.85 .. 1.02 eps."""
    names = list(g9b["variant_names"])
    total_free = 0
    for name in g9b["models"]:
        res = g9b[f"{name}__residuals"]
        w = np.maximum(res[:, 0], -res[:, 2])
        assert w[names.index("reference")] < 0.6 * EPS
        for vn in CONTROLS:
            assert w[names.index(vn)] > 23 * EPS, (name, vn)
        assert w[names.index("scaled_1e-4")] > 1.45 * w[names.index("reference")]
    assert sorted(g9b["models"]) == sorted(RNA002)


CONTROLS = ["penalty_not_squared", "penalty_zero", "no_final_sqrt", "window_minus_1", "window_plus_1",
            "window_5", "unbanded", "penalty_plus_10pct", "penalty_minus_10pct"]


@pytest.mark.parametrize("name", ["WDX4", "WDX6"])
def test_negative_controls_violate_the_kkt_conditions(g9, name):
    """The test has teeth: each single change to the recurrence breaks the conditions by >= 17 eps."""
    m = kkt.model_from_npz(g9, name)
    vs = kkt.variants(oracle_dtw, m["X"], m["window"], m["penalty"])
    for vn in CONTROLS:
        r = kkt.kkt_residuals(vs[vn](), m["n_support"], m["dual_coef"], m["intercept"], m["c_bound"], m["gamma"], m["pwr_dist"])
        assert kkt.worst(r) > 17 * EPS, (vn, r)


def test_recorded_controls_of_every_model(g9):
    """All five models' recorded residuals (make_golden_kkt.py): reference variant inside eps, every
    control outside by >= 17 eps, and the 1e-4 uniform scaling still inside (the stated resolution)."""
    names = list(g9["variant_names"])
    for name in g9["models"]:
        res = g9[f"{name}__residuals"]
        w = np.maximum(np.maximum(res[:, 0], res[:, 1]), -res[:, 2])
        assert w[names.index("reference")] < 0.6 * EPS
        for vn in CONTROLS:
            assert w[names.index(vn)] > 17 * EPS, (name, vn)
        assert w[names.index("reference")] < w[names.index("scaled_1e-4")] < EPS


PEN_GRID = (-0.005, -0.0025, 0.0, 0.0025, 0.005)      # relative changes of the penalty around the models' 0.1
SCALE_GRID = (-1e-4, -5e-5, 0.0, 5e-5, 1e-4)            # uniform relative scalings of the distances


def fit_penalty_and_scale(dtw, m):
    """A FIT instead of a threshold: the free vectors' ~2 000 equalities y f(x_s) = 1 as a least-squares
    problem in (i) the DTW penalty and (ii) a uniform scale of the distances.  Returns (relative penalty offset of the
    minimiser, scale offset of the minimiser, grid minimiser of the WORST residual for either)."""
    args = (m["n_support"], m["dual_coef"], m["intercept"], m["c_bound"], m["gamma"], m["pwr_dist"])
    ms_p, mx_p = [], []
    for rel in PEN_GRID:
        r = kkt.free_residuals(dtw(m["X"], m["window"], m["penalty"] * (1.0 + rel)), *args)
        ms_p.append(float((r * r).mean()))
        mx_p.append(float(np.abs(r).max()))
    D = dtw(m["X"], m["window"], m["penalty"]).astype(np.float64)
    ms_s, mx_s = [], []
    for sc in SCALE_GRID:
        r = kkt.free_residuals(D * (1.0 + sc), *args)
        ms_s.append(float((r * r).mean()))
        mx_s.append(float(np.abs(r).max()))
    return (kkt.parabola_vertex(PEN_GRID, ms_p), kkt.parabola_vertex(SCALE_GRID, ms_s),
            PEN_GRID[int(np.argmin(mx_p))], SCALE_GRID[int(np.argmin(mx_s))])


@pytest.mark.parametrize("name", ["WDX4", "WDX4b"])
def test_fitted_penalty_and_scale_are_the_models_own(g9, name):
    """The penalty and the distance scale that BEST explain the shipped model's free-vector equalities are the ones the
    restatement uses: least-squares minimiser at penalty = 0.1 within 0.5 % (measured: < 0.05 %) and at scale = 1 within
    1e-4 (measured: < 2e-5); the worst single residual is smallest at the grid's centre for the penalty (+-0.25 % already
    raises it 1.5x) and within one grid step (5e-5) of it for the scale.  This is the stated resolution of the stage-B
    pin: "pinned to ~1e-4 by the shipped models; bit-exact to the restatement; L = 110 by shared code"."""
    m = kkt.model_from_npz(g9, name)
    vp, vs, gp, gs = fit_penalty_and_scale(oracle_dtw, m)
    assert abs(vp) <= 0.005 and abs(vs) <= 1e-4, (vp, vs)
    assert abs(vp) <= 0.001 and abs(vs) <= 3e-5, (vp, vs)      # what was measured, with margin
    assert gp == 0.0 and abs(gs) <= 5e-5, (gp, gs)


# ---- dtaidistance's published examples (documentation, "DTW between set of series"; stated from the docs
# of 2.3.x as the builder knows them -- the pages are not fetchable here, so these are ANCHORS, not pins) ----

def test_documented_distance_matrix_example():
    s = [np.array([0.0, 0, 1, 2, 1, 0, 1, 0, 0]), np.array([0.0, 1, 2, 0, 0, 0, 0, 0, 0, 0, 0]),
         np.array([0.0, 0, 1, 2, 1, 0, 0, 0])]                 # unequal lengths 9 / 11 / 8
    got = np.array([[orc.dtw_distance(a, b) for b in s] for a in s])
    want = np.array([[0, 1.41421356, 1.0], [1.41421356, 0, 1.0], [1.0, 1.0, 0]])
    assert np.allclose(got, want, rtol=0, atol=5e-9)


def test_documented_block_example():
    """`distance_matrix_fast(series, block=((1, 4), (3, 5)))` of the six-series example: rows 1..3 x
    columns 3..4, upper triangle only -- the call shape parallel_distances.py:34-43 uses."""
    base = [[0.0, 0, 1, 2, 1, 0, 1, 0, 0], [0.0, 1, 2, 0, 0, 0, 0, 0, 0], [1.0, 2, 0, 0, 0, 0, 0, 1, 1]]
    s = np.array(base + base)
    D = orc.dtw_matrix(s[1:4], s[3:5])
    printed = {(1, 3): 1.4142, (1, 4): 0.0, (2, 3): 2.2360, (2, 4): 1.7320, (3, 4): 1.4142}   # the docs print 4 digits, truncated
    for (r, c), v in printed.items():
        assert abs(float(D[r - 1, c - 3]) - v) < 1e-4
    assert D[1, 0] == np.float32(np.sqrt(5.0)) and D[1, 1] == np.float32(np.sqrt(3.0))

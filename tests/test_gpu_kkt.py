"""Stage-B pin of the PRODUCT path: the HIP DTW kernels (through the C ABI and the reference-named shim)
against the KKT conditions of the reference's five shipped DTW_SVM models -- reference-held data, no
oracle in between (tests/helpers/kkt.py, tests/test_oracle_dtw_kkt.py explain the conditions and the
resolution).  Needs a real MI355X."""
import os

import numpy as np
import pytest

from helpers import kkt
from warpdemux_amd import parallel_distances as pdist

pytestmark = pytest.mark.gpu

G9 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g9_kkt_models.npz")
EPS = kkt.EPS_LIBSVM


def device_dtw(X, window, penalty):
    # the training call of the reference: the all-vs-all kernel matrix (parallel_distances.py:139-198)
    return pdist.parallel_distance_matrix(X, block_size=500, n_jobs=1, window=window, penalty=penalty)


@pytest.mark.parametrize("name", ["WDX4", "WDX4b", "WDX4c", "WDX6", "WDX10"])
def test_device_dtw_satisfies_the_shipped_models_kkt_conditions(name):
    g9 = np.load(G9)
    m = kkt.model_from_npz(g9, name)
    D = device_dtw(m["X"], m["window"], m["penalty"])
    assert D.dtype == np.float32 and D.shape == (m["X"].shape[0],) * 2
    r = kkt.kkt_residuals(D, m["n_support"], m["dual_coef"], m["intercept"], m["c_bound"], m["gamma"], m["pwr_dist"])
    assert r["free_max_abs"] < EPS and r["bound_max"] < EPS and r["zero_min"] > -EPS, r
    # same residuals as recorded with the CPU restatement when the fixture was made
    ref = g9[f"{name}__residuals"][list(g9["variant_names"]).index("reference")]
    assert np.allclose([r["free_max_abs"], r["bound_max"], r["zero_min"]], ref, rtol=0, atol=1e-9)
    # distance_matrix_to (the predict-time seam, dtw_svm.py:79-88) returns the same matrix
    D2 = pdist.distance_matrix_to(m["X"][:257], m["X"], window=m["window"], penalty=m["penalty"], n_jobs=1)
    assert np.array_equal(D2, D[:257])


def test_device_negative_controls_violate_the_kkt_conditions():
    g9 = np.load(G9)
    m = kkt.model_from_npz(g9, "WDX4")
    vs = kkt.variants(device_dtw, m["X"], m["window"], m["penalty"])
    for vn in ("penalty_not_squared", "penalty_zero", "window_minus_1", "window_plus_1", "window_5", "unbanded",
               "penalty_plus_10pct", "penalty_minus_10pct"):
        r = kkt.kkt_residuals(vs[vn](), m["n_support"], m["dual_coef"], m["intercept"], m["c_bound"], m["gamma"], m["pwr_dist"])
        assert kkt.worst(r) > 17 * EPS, (vn, r)


@pytest.mark.parametrize("name", ["WDX4", "WDX4b", "WDX4c", "WDX6", "WDX10"])
def test_device_fitted_penalty_and_scale_are_the_models_own(name):
    """The least-squares fit of tests/test_oracle_dtw_kkt.py with the HIP kernels' distances, all five models: the
    penalty / uniform scale that best explain the model's free-vector equalities are (0.1 within 0.5 %, 1 within 1e-4)."""
    from test_oracle_dtw_kkt import fit_penalty_and_scale

    g9 = np.load(G9)
    m = kkt.model_from_npz(g9, name)
    vp, vs, gp, gs = fit_penalty_and_scale(device_dtw, m)
    assert abs(vp) <= 0.005 and abs(vs) <= 1e-4, (vp, vs)
    assert gp == 0.0 and abs(gs) <= 5e-5, (gp, gs)

"""Stage-B pin of the PRODUCT path: the HIP DTW kernels (through the C ABI and the reference-named shim)
against the KKT conditions of the reference's eleven DTW_SVM models (five shipped rna004, six rna002 under
DEPRECATED/model_files) -- reference-held data, no oracle in between (tests/helpers/kkt.py,
tests/test_oracle_dtw_kkt.py explain the conditions and the resolution).  Every model twice: on
`dtw_short_kernel<25,15>` (what the shipped shape dispatches to) and, with WDX_OPT_NO_SHORT_DTW, on
`dtw_band_kernel<15>` -- the kernel that carries the headline's 110-point matrix.  Needs a real MI355X."""
import os

import numpy as np
import pytest

import contextlib

from helpers import kkt
from warpdemux_amd import _lib, parallel_distances as pdist

pytestmark = pytest.mark.gpu

G9 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g9_kkt_models.npz")
G9B = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g9b_kkt_models_rna002.npz")
EPS = kkt.EPS_LIBSVM
ALL_MODELS = [(G9, n) for n in ("WDX4", "WDX4b", "WDX4c", "WDX6", "WDX10")] + \
             [(G9B, n) for n in ("WDX4", "WDX-DPC", "WDX6", "WDX8", "WDX10", "WDX12")]
IDS = ["%s-%s" % (n, "rna004" if f == G9 else "rna002") for f, n in ALL_MODELS]


@contextlib.contextmanager
def band_kernel(on):
    """WDX_OPT_NO_SHORT_DTW: the 25-point shape on dtw_band_kernel<15> (the headline's kernel) instead of dtw_short_kernel"""
    ctx = _lib.default_context(None)
    ctx.set_option(_lib.OPT_NO_SHORT_DTW, int(on))
    try:
        yield
    finally:
        ctx.set_option(_lib.OPT_NO_SHORT_DTW, 0)


def device_dtw(X, window, penalty):
    # the training call of the reference: the all-vs-all kernel matrix (parallel_distances.py:139-198)
    return pdist.parallel_distance_matrix(X, block_size=500, n_jobs=1, window=window, penalty=penalty)


@pytest.mark.parametrize("band", [False, True], ids=["dtw_short_kernel", "dtw_band_kernel"])
@pytest.mark.parametrize("fixture,name", ALL_MODELS, ids=IDS)
def test_device_dtw_satisfies_the_reference_models_kkt_conditions(fixture, name, band):
    g9 = np.load(fixture)
    m = kkt.model_from_npz(g9, name)
    with band_kernel(band):
        D = device_dtw(m["X"], m["window"], m["penalty"])
        # distance_matrix_to (the predict-time seam, dtw_svm.py:79-88) returns the same matrix
        D2 = pdist.distance_matrix_to(m["X"][:257], m["X"], window=m["window"], penalty=m["penalty"], n_jobs=1)
    assert D.dtype == np.float32 and D.shape == (m["X"].shape[0],) * 2
    r = kkt.kkt_residuals(D, m["n_support"], m["dual_coef"], m["intercept"], m["c_bound"], m["gamma"], m["pwr_dist"])
    assert r["free_max_abs"] < EPS and r["bound_max"] < EPS and r["zero_min"] > -EPS, r
    # same residuals as recorded with the CPU restatement when the fixture was made
    ref = g9[f"{name}__residuals"][list(g9["variant_names"]).index("reference")]
    assert np.allclose([r["free_max_abs"], r["bound_max"], r["zero_min"]], ref, rtol=0, atol=1e-9)
    assert np.array_equal(D2, D[:257])


def test_short_and_band_kernels_return_the_same_bits():
    """dtw_short_kernel<25,15> and dtw_band_kernel<15> on the largest reference-held training set (WDX12-rna002,
    3 617 x 25): one matrix, bit for bit -- so the pin of either is the pin of both."""
    m = kkt.model_from_npz(np.load(G9B), "WDX12")
    D = device_dtw(m["X"], m["window"], m["penalty"])
    with band_kernel(True):
        Db = device_dtw(m["X"], m["window"], m["penalty"])
    assert np.array_equal(D.view(np.uint32), Db.view(np.uint32))


def test_device_negative_controls_violate_the_kkt_conditions():
    g9 = np.load(G9)
    m = kkt.model_from_npz(g9, "WDX4")
    vs = kkt.variants(device_dtw, m["X"], m["window"], m["penalty"])
    for vn in ("penalty_not_squared", "penalty_zero", "window_minus_1", "window_plus_1", "window_5", "unbanded",
               "penalty_plus_10pct", "penalty_minus_10pct"):
        r = kkt.kkt_residuals(vs[vn](), m["n_support"], m["dual_coef"], m["intercept"], m["c_bound"], m["gamma"], m["pwr_dist"])
        assert kkt.worst(r) > 17 * EPS, (vn, r)


@pytest.mark.parametrize("band", [False, True], ids=["dtw_short_kernel", "dtw_band_kernel"])
@pytest.mark.parametrize("name", ["WDX4", "WDX4b", "WDX4c", "WDX6", "WDX10"])
def test_device_fitted_penalty_and_scale_are_the_models_own(name, band):
    """The least-squares fit of tests/test_oracle_dtw_kkt.py with the HIP kernels' distances, all five models: the
    penalty / uniform scale that best explain the model's free-vector equalities are (0.1 within 0.5 %, 1 within 1e-4)."""
    from test_oracle_dtw_kkt import fit_penalty_and_scale

    g9 = np.load(G9)
    m = kkt.model_from_npz(g9, name)
    with band_kernel(band):
        vp, vs, gp, gs = fit_penalty_and_scale(device_dtw, m)
    assert abs(vp) <= 0.005 and abs(vs) <= 1e-4, (vp, vs)
    assert gp == 0.0 and abs(gs) <= 5e-5, (gp, gs)

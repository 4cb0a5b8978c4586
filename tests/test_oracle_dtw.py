"""Anchors for the DTW half of the oracle: the restated recurrence against the library's
documented example, a literal full-matrix form, exhaustive path enumeration and invariants.  dtaidistance 2.3.13 is
not in the reference tree nor installable here; the pin to reference-held data is tests/test_oracle_dtw_kkt.py (the
KKT conditions of the five shipped DTW_SVM models)."""
import itertools
import math

import numpy as np
import pytest

from oracle import wdx_oracle as orc


def dtw_literal(s1, s2, window=None, penalty=None):
    """Full (l1+1)x(l2+1) matrix form of App. A.1, pure Python."""
    l1, l2 = len(s1), len(s2)
    w = window if window else max(l1, l2)
    p2 = (penalty or 0.0) ** 2
    D = [[math.inf] * (l2 + 1) for _ in range(l1 + 1)]
    D[0][0] = 0.0
    for i in range(l1):
        j0 = max(0, i - max(l1 - l2, 0) - (w - 1))
        j1 = min(l2, i + w + max(l2 - l1, 0))
        for j in range(j0, j1):
            d = (s1[i] - s2[j]) ** 2
            D[i + 1][j + 1] = d + min(D[i][j], D[i][j + 1] + p2, D[i + 1][j] + p2)
    return math.sqrt(D[l1][l2])


def dtw_bruteforce(s1, s2, window, penalty):
    """Minimum over every monotone warping path inside the band (L <= 6)."""
    l1, l2 = len(s1), len(s2)
    w = window if window else max(l1, l2)
    p2 = penalty**2
    best = math.inf

    def rec(i, j, acc):
        nonlocal best
        if abs(i - j) > w - 1:
            return
        acc += (s1[i] - s2[j]) ** 2
        if i == l1 - 1 and j == l2 - 1:
            best = min(best, acc)
            return
        if i + 1 < l1 and j + 1 < l2:
            rec(i + 1, j + 1, acc)
        if i + 1 < l1:
            rec(i + 1, j, acc + p2)
        if j + 1 < l2:
            rec(i, j + 1, acc + p2)

    rec(0, 0, 0.0)
    return math.sqrt(best)


def test_documented_example():
    s1 = [0, 0, 1, 2, 1, 0, 1, 0, 0]
    s2 = [0, 1, 2, 0, 0, 0, 0, 0, 0]
    assert orc.dtw_distance(s1, s2) == 1.4142135623730951


def test_window1_is_euclidean_and_identity_zero():
    rng = np.random.default_rng(1)
    a, b = rng.normal(size=110), rng.normal(size=110)
    assert orc.dtw_distance(a, b, window=1) == pytest.approx(np.linalg.norm(a - b), rel=1e-15)
    assert orc.dtw_distance(a, a, window=15, penalty=0.1) == 0.0


def test_matches_literal_full_matrix_bitwise():
    rng = np.random.default_rng(2)
    for L in (1, 2, 5, 25, 110, 131):
        for (w, p) in ((15, 0.1), (None, None), (3, 0.0), (1, 2.0), (200, 0.1), (16, 1.5)):
            a, b = rng.normal(size=L), rng.normal(size=L)
            assert orc.dtw_distance(a, b, w, p) == dtw_literal(a.tolist(), b.tolist(), w, p)


def test_bruteforce_paths_small():
    rng = np.random.default_rng(3)
    for L in (1, 2, 3, 4, 5, 6):
        for w in (1, 2, 3, 6):
            for p in (0.0, 0.1, 1.0):
                a, b = rng.normal(size=L), rng.normal(size=L)
                assert orc.dtw_distance(a, b, w, p) == pytest.approx(dtw_bruteforce(a, b, w, p), rel=1e-14, abs=1e-300)


def test_invariants():
    rng = np.random.default_rng(4)
    a, b = rng.normal(size=110), rng.normal(size=110)
    d = orc.dtw_distance(a, b, 15, 0.1)
    # symmetry (exact: the transposed recurrence performs the same operations)
    assert orc.dtw_distance(b, a, 15, 0.1) == d
    # monotone: non-decreasing in penalty, non-increasing in window
    ps = [orc.dtw_distance(a, b, 15, p) for p in (0, 0.05, 0.1, 0.5, 2.0)]
    assert all(x <= y for x, y in zip(ps, ps[1:]))
    ws = [orc.dtw_distance(a, b, w, 0.1) for w in (1, 2, 5, 15, 40, 110, 300)]
    assert all(x >= y for x, y in zip(ws, ws[1:]))
    # p=0, w>=L is classical DTW
    assert orc.dtw_distance(a, b, 110, 0.0) == orc.dtw_distance(a, b, None, None)
    # translation of both series
    assert orc.dtw_distance(a + 3.0, b + 3.0, 15, 0.1) == pytest.approx(d, rel=1e-12)
    # window None/0 both mean unbanded (parallel_distances.py:52-53 defaults)
    assert orc.dtw_distance(a, b, 0, 0.1) == orc.dtw_distance(a, b, None, 0.1)


def test_nan_propagates():
    a, b = np.arange(30.0), np.arange(30.0)[::-1].copy()
    a2 = a.copy()
    a2[7] = np.nan
    assert math.isnan(orc.dtw_distance(a2, b, 15, 0.1))
    assert math.isnan(orc.dtw_distance(b, a2, 15, 0.1))


def test_matrix_and_argmin():
    rng = np.random.default_rng(5)
    X, Y = rng.normal(size=(37, 110)), rng.normal(size=(10, 110))
    D = orc.dtw_matrix(X, Y, 15, 0.1)
    assert D.dtype == np.float32 and D.shape == (37, 10)
    for r, c in itertools.product((0, 5, 36), (0, 9)):
        assert D[r, c] == np.float32(orc.dtw_distance(X[r], Y[c], 15, 0.1))
    assert np.array_equal(orc.argmin_rows(D), np.argmin(D, axis=1))
    D2 = D.copy()
    D2[3, 4] = np.nan
    D2[5, 2] = D2[5, 7] = D2[5].min()
    assert np.array_equal(orc.argmin_rows(D2), np.argmin(D2, axis=1))

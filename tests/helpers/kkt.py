"""KKT residuals of a shipped DTW_SVM model under a candidate DTW distance matrix (test infrastructure).

The reference's models (`warpdemux/models/model_files/*.joblib`) are scikit-learn `SVC(kernel=
"precomputed", C=1, class_weight="balanced", tol=1e-3)` objects whose training kernel was
`K = exp(-gamma * D)` (`models/dtw_svm.py:21-22`) with `D` = the genuine dtaidistance matrix of the
training fingerprints `_X` (`parallel_distances.py:59-67, 139-198`).  Every row of `_X` is a support
vector, so for each one-vs-one pair (i < j) libsvm's stopping rule left behind, at every training point
s of the two classes (y = +1 for class i, -1 for class j, f = decision value of that pair):

    0 < alpha_s < C_s   (free)     |y f(x_s) - 1| <  eps          eps = tol = 1e-3
    alpha_s = C_s       (bounded)   y f(x_s) - 1  <  eps
    alpha_s = 0                     y f(x_s) - 1  > -eps

(`C_s = C * class_weight_[class(s)]`; libsvm's working-set gap m(alpha) - M(alpha) < eps, rho = the
midpoint over the free vectors).  These are ~23 000 equalities over the five shipped models that only
hold if the candidate D reproduces the distances the reference trained on: they are the reference-held
data the DTW restatement is pinned to.
"""
from __future__ import annotations

import numpy as np

EPS_LIBSVM = 1e-3  # SVC(tol=1e-3), the value stored in every shipped model


def pair_layout(n_support):
    """(i, j, rows_of_class_i, rows_of_class_j, coefficient row of class i's vectors, of class j's)
    per one-vs-one pair in libsvm's order (svm.cpp svm_predict_values: sv_coef[j-1][si+k], sv_coef[i][sj+k])."""
    k = len(n_support)
    start = np.concatenate([[0], np.cumsum(n_support)]).astype(np.int64)
    for i in range(k):
        for j in range(i + 1, k):
            yield i, j, np.arange(start[i], start[i + 1]), np.arange(start[j], start[j + 1]), j - 1, i


def kkt_residuals(D, n_support, dual_coef, intercept, c_bound, gamma=1.0, pwr_dist=1):
    """D: (n, n) candidate distance matrix of the model's `_X` against itself (float32 like
    `distance_matrix_to` returns).  Returns the three worst residuals and the counts."""
    D = np.asarray(D, dtype=np.float32)
    K = np.exp(-gamma * np.power(D, pwr_dist)).astype(np.float64)     # pdist_kernel, dtw_svm.py:21-22
    free_max, bound_max, zero_min = 0.0, -np.inf, np.inf
    n_free = n_bound = n_zero = 0
    for p, (i, j, ri, rj, row_i, row_j) in enumerate(pair_layout(n_support)):
        idx = np.concatenate([ri, rj])
        coef = np.concatenate([dual_coef[row_i, ri], dual_coef[row_j, rj]])
        y = np.concatenate([np.ones(ri.size), -np.ones(rj.size)])
        cb = np.concatenate([np.full(ri.size, c_bound[i]), np.full(rj.size, c_bound[j])])
        alpha = coef * y
        if (alpha < -1e-12).any():
            raise AssertionError("dual coefficient with the wrong sign for its class")
        f = K[np.ix_(idx, idx)] @ coef + intercept[p]               # sklearn stores -rho
        r = y * f - 1.0
        zero = alpha <= 1e-12
        bound = alpha >= cb * (1.0 - 1e-9)
        free = ~zero & ~bound
        if free.any():
            free_max = max(free_max, float(np.abs(r[free]).max()))
        if bound.any():
            bound_max = max(bound_max, float(r[bound].max()))
        if zero.any():
            zero_min = min(zero_min, float(r[zero].min()))
        n_free += int(free.sum())
        n_bound += int(bound.sum())
        n_zero += int(zero.sum())
    return {"free_max_abs": free_max, "bound_max": bound_max, "zero_min": zero_min,
            "n_free": n_free, "n_bound": n_bound, "n_zero": n_zero}


def free_residuals(D, n_support, dual_coef, intercept, c_bound, gamma=1.0, pwr_dist=1):
    """y f(x_s) - 1 of every free vector (0 < alpha_s < C_s) of every one-vs-one pair under the candidate D: the
    ~2 000 (WDX4) .. ~9 000 (WDX10) EQUALITIES libsvm left behind, as one vector (for least-squares fits)."""
    D = np.asarray(D)
    K = np.exp(-gamma * np.power(D.astype(np.float32) if D.dtype == np.float32 else D, pwr_dist)).astype(np.float64)
    out = []
    for p, (i, j, ri, rj, row_i, row_j) in enumerate(pair_layout(n_support)):
        idx = np.concatenate([ri, rj])
        coef = np.concatenate([dual_coef[row_i, ri], dual_coef[row_j, rj]])
        y = np.concatenate([np.ones(ri.size), -np.ones(rj.size)])
        cb = np.concatenate([np.full(ri.size, c_bound[i]), np.full(rj.size, c_bound[j])])
        alpha = coef * y
        f = K[np.ix_(idx, idx)] @ coef + intercept[p]
        free = (alpha > 1e-12) & (alpha < cb * (1.0 - 1e-9))
        out.append((y * f - 1.0)[free])
    return np.concatenate(out)


def parabola_vertex(x, y):
    """Abscissa of the vertex of the least-squares parabola through (x, y)."""
    c2, c1, _ = np.polyfit(np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64), 2)
    if not c2 > 0:
        raise AssertionError("the residuals do not have a minimum inside the grid")
    return -c1 / (2.0 * c2)


def worst(res):
    """One number: the largest violation of any of the three conditions (<= eps when D is right)."""
    return max(res["free_max_abs"], res["bound_max"], -res["zero_min"])


def model_from_npz(z, name):
    g = lambda f: z[f"{name}__{f}"]  # noqa: E731
    return {"X": g("X"), "n_support": g("n_support"), "dual_coef": g("dual_coef"), "intercept": g("intercept"),
            "c_bound": g("c_bound"), "gamma": float(g("gamma")), "pwr_dist": int(g("pwr_dist")),
            "window": int(g("window")), "penalty": float(g("penalty"))}


# the candidate distance functions: `dtw(X, window, penalty)` -> float32 (n, n) is the implementation
# under test (oracle or device); each control changes ONE thing about how it is called / post-processed.
def variants(dtw, X, window, penalty):
    return {
        "reference": lambda: dtw(X, window, penalty),
        "penalty_not_squared": lambda: dtw(X, window, float(np.sqrt(penalty))),
        "penalty_zero": lambda: dtw(X, window, 0.0),
        "no_final_sqrt": lambda: dtw(X, window, penalty).astype(np.float64) ** 2,
        "window_minus_1": lambda: dtw(X, window - 1, penalty),
        "window_plus_1": lambda: dtw(X, window + 1, penalty),
        "window_5": lambda: dtw(X, 5, penalty),
        "unbanded": lambda: dtw(X, None, penalty),
        "penalty_plus_10pct": lambda: dtw(X, window, penalty * 1.1),
        "penalty_minus_10pct": lambda: dtw(X, window, penalty * 0.9),
        "scaled_1e-4": lambda: dtw(X, window, penalty).astype(np.float64) * (1 + 1e-4),
    }

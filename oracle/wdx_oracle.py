"""ctypes front-end of oracle/libwdx_oracle.so -- TEST INFRASTRUCTURE ONLY (see wdx_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libwdx_oracle.so")

NORM_CODES = {"none": 0, "mean": 1, "median": 2}
FAIL_REASONS = {
    0: "",
    1: "detect failed",
    2: "signal normalization failed",
    3: "event segmentation failed",
    4: "segment normalization failed",
    5: "unknown",
    6: "consensus query outlier",
}


class SegParamsC(C.Structure):
    _fields_ = [
        ("padding", C.c_int32),
        ("sig_norm", C.c_int32),
        ("outlier_thresh", C.c_float),
        ("min_obs_per_base", C.c_int32),
        ("running_stat_width", C.c_int32),
        ("num_events", C.c_int32),
        ("accept_less_cpts", C.c_int32),
        ("seg_norm", C.c_int32),
        ("barcode_num_events", C.c_int32),
        ("clip_bounds_f64", C.c_int32),
        ("outlier_thresh_f64", C.c_double),
    ]


@dataclass
class SegParams:
    """Values of config_files/rna004_130bps@v1.0.toml; outlier_thresh = ADAPTed default 5.0."""

    padding: int = 100
    sig_norm: str = "none"
    outlier_thresh: float = 5.0
    min_obs_per_base: int = 6
    running_stat_width: int = 12
    num_events: int = 110
    accept_less_cpts: bool = False
    seg_norm: str = "mean"
    barcode_num_events: int = 25
    clip_bounds_f64: bool = False   # True = NumPy 1.x promotion (the reference's pinned 1.26.4) / np.float64 threshold

    def to_c(self) -> SegParamsC:
        return SegParamsC(
            self.padding, NORM_CODES[self.sig_norm], self.outlier_thresh, self.min_obs_per_base,
            self.running_stat_width, self.num_events, int(self.accept_less_cpts),
            NORM_CODES[self.seg_norm], self.barcode_num_events, int(self.clip_bounds_f64),
            float(self.outlier_thresh),
        )


class RefineParamsC(C.Structure):
    _fields_ = [
        ("query", C.c_void_p),
        ("n_query", C.c_int32),
        ("subseq_norm", C.c_int32),
        ("penalty", C.c_double),
        ("psi", C.c_int32 * 4),
        ("ub_start", C.c_int32),
        ("lb_end", C.c_int32),
        ("ub_end", C.c_int32),
        ("barcode_segm_events", C.c_int32),
        ("barcode_keep_events", C.c_int32),
    ]


@dataclass
class RefineParams:
    """segmentation.consensus_* knobs (config/sig_proc.py:57-66; values of the tRNA config) + the consensus query."""

    query: np.ndarray = None
    subseq_norm: str = "mean"
    penalty: float = 1.5
    psi: tuple = (5, 0, 40, 0)
    ub_start: int = 18
    lb_end: int = 69
    ub_end: int = 97
    barcode_segm_events: int = 25
    barcode_keep_events: int = 25

    def to_c(self):
        self._q = np.ascontiguousarray(self.query, dtype=np.float64)
        return RefineParamsC(self._q.ctypes.data, self._q.size, NORM_CODES[self.subseq_norm], float(self.penalty),
                             (C.c_int32 * 4)(*[int(v) for v in self.psi]), self.ub_start, self.lb_end, self.ub_end,
                             self.barcode_segm_events, self.barcode_keep_events)


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "wdx_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libwdx_oracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        P = C.POINTER
        L.wdx_oracle_windowed_t_test.restype = C.c_int64
        L.wdx_oracle_windowed_t_test.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
        L.wdx_oracle_find_peaks.restype = C.c_int64
        L.wdx_oracle_find_peaks.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
        L.wdx_oracle_scores_to_cpts.restype = C.c_int64
        L.wdx_oracle_scores_to_cpts.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_void_p]
        L.wdx_oracle_new_means.restype = None
        L.wdx_oracle_new_means.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.wdx_oracle_fingerprint_one.restype = C.c_int
        L.wdx_oracle_fingerprint_one.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int, P(SegParamsC), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.wdx_oracle_normalize_f64.restype = C.c_int
        L.wdx_oracle_normalize_f64.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
        L.wdx_oracle_normalize_f32.restype = C.c_int
        L.wdx_oracle_normalize_f32.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
        L.wdx_oracle_nanmedian_mad_f32.restype = None
        L.wdx_oracle_nanmedian_mad_f32.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        L.wdx_oracle_normalize_wrt.restype = C.c_int
        L.wdx_oracle_normalize_wrt.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
        L.wdx_oracle_subseq_match.restype = C.c_int
        L.wdx_oracle_subseq_match.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_double, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
        L.wdx_oracle_fingerprint_refine_batch.restype = C.c_int
        L.wdx_oracle_fingerprint_refine_batch.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, P(SegParamsC), P(RefineParamsC), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.wdx_oracle_fingerprint_batch.restype = C.c_int
        L.wdx_oracle_fingerprint_batch.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, P(SegParamsC), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.wdx_oracle_fingerprint_packed.restype = C.c_int
        L.wdx_oracle_fingerprint_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, P(SegParamsC), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.wdx_oracle_dtw_distance.restype = C.c_double
        L.wdx_oracle_dtw_distance.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_double]
        L.wdx_oracle_dtw_matrix.restype = C.c_int
        L.wdx_oracle_dtw_matrix.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_double, C.c_void_p]
        L.wdx_oracle_argmin_rows.restype = None
        L.wdx_oracle_argmin_rows.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
        L.wdx_oracle_svm_predict_proba.restype = C.c_int
        L.wdx_oracle_svm_predict_proba.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def windowed_t_test(x: np.ndarray, w: int) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty(max(x.size - 2 * w, 0), dtype=np.float64)
    n = lib().wdx_oracle_windowed_t_test(_p(x), x.size, w, _p(out))
    return out[: max(n, 0)]


def find_peaks(x: np.ndarray, distance: int) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty(x.size // 2 + 2, dtype=np.int64)
    n = lib().wdx_oracle_find_peaks(_p(x), x.size, distance, _p(out))
    return out[:n]


def scores_to_cpts(scores, num_events=110, min_obs_per_base=6, running_stat_width=12, accept_less_cpts=False):
    scores = np.ascontiguousarray(scores, dtype=np.float64)
    out = np.empty(num_events + 2, dtype=np.int64)
    n = lib().wdx_oracle_scores_to_cpts(_p(scores), scores.size, num_events, min_obs_per_base, running_stat_width, int(accept_less_cpts), _p(out))
    if n < 0:
        raise ValueError("reference would raise")
    return out[:n]


def new_means(x, segs) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float64)
    segs = np.ascontiguousarray(segs, dtype=np.int64)
    out = np.empty(segs.size - 1, dtype=np.float64)
    lib().wdx_oracle_new_means(_p(x), _p(segs), segs.size - 1, _p(out))
    return out


def normalize(x, method="mean"):
    """sig_proc.normalize on a 1-D vector: float64 (NaN-free, accept_nan=False semantics) or float32
    (stage A2: accept_nan=True semantics)."""
    x = np.ascontiguousarray(x)
    if x.dtype == np.float32:
        out = np.empty_like(x)
        rc = lib().wdx_oracle_normalize_f32(_p(x), x.size, NORM_CODES[method], _p(out))
    else:
        x = x.astype(np.float64, copy=False)
        out = np.empty_like(x)
        rc = lib().wdx_oracle_normalize_f64(_p(x), x.size, NORM_CODES[method], _p(out))
    if rc:
        raise ValueError(f"Normalization method {method} not recognized.")
    return out


def nanmedian_mad_f32(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    med, mad = C.c_float(0), C.c_float(0)
    lib().wdx_oracle_nanmedian_mad_f32(_p(x), x.size, C.byref(med), C.byref(mad))
    return np.float32(med.value), np.float32(mad.value)


def normalize_wrt(to_norm, ref, method="mean"):
    to_norm = np.ascontiguousarray(to_norm, dtype=np.float64)
    ref = np.ascontiguousarray(ref, dtype=np.float64)
    out = np.empty_like(to_norm)
    if lib().wdx_oracle_normalize_wrt(_p(to_norm), to_norm.size, _p(ref), ref.size, NORM_CODES[method], _p(out)):
        raise ValueError(f"Normalization method {method} not recognized.")
    return out


def fingerprint_one(row, a_start, a_end, params: SegParams, ok=True):
    """-> dict(status, fpt, dwell, stats, cpts)"""
    row = np.ascontiguousarray(row, dtype=np.float32)
    K = params.barcode_num_events
    fpt = np.full(K, np.nan)
    dwell = np.zeros(K, dtype=np.int64)
    stats = np.full(6, np.nan)
    cpts = np.zeros(params.num_events + 2, dtype=np.int64)
    ncp = C.c_int64(0)
    pc = params.to_c()
    st = lib().wdx_oracle_fingerprint_one(_p(row), row.size, int(a_start), int(a_end), int(ok), C.byref(pc), _p(fpt), _p(dwell), _p(stats), _p(cpts), C.byref(ncp))
    return dict(status=st, fpt=fpt, dwell=dwell, stats=stats, cpts=cpts[: ncp.value])


def fingerprint_batch(sig, a_start, a_end, params: SegParams, ok=None):
    sig = np.ascontiguousarray(sig, dtype=np.float32)
    n, stride = sig.shape
    K = params.barcode_num_events
    a_start = np.ascontiguousarray(a_start, dtype=np.int32)
    a_end = np.ascontiguousarray(a_end, dtype=np.int32)
    okp = None if ok is None else np.ascontiguousarray(ok, dtype=np.uint8)
    fpt = np.full((n, K), np.nan)
    dwell = np.zeros((n, K), dtype=np.int64)
    stats = np.full((n, 6), np.nan)
    status = np.zeros(n, dtype=np.int32)
    pc = params.to_c()
    lib().wdx_oracle_fingerprint_batch(_p(sig), n, stride, _p(a_start), _p(a_end), None if okp is None else _p(okp), C.byref(pc), _p(fpt), _p(dwell), _p(stats), _p(status))
    return fpt, dwell, stats, status


def subseq_match(query, series, penalty=1.5, psi=(5, 0, 40, 0)):
    """(start, end) = SubsequenceAlignment(...).best_match().segment as sig_proc.py:287-306 obtains it."""
    q = np.ascontiguousarray(query, dtype=np.float64)
    s = np.ascontiguousarray(series, dtype=np.float64)
    st, en = C.c_int64(0), C.c_int64(0)
    if lib().wdx_oracle_subseq_match(_p(q), q.size, _p(s), s.size, float(penalty), int(psi[0]), int(psi[2]),
                                     C.byref(st), C.byref(en)):
        raise ValueError("empty query or series")
    return st.value, en.value


def fingerprint_refine_batch(sig, a_start, a_end, params: SegParams, rp: RefineParams, ok=None):
    """consensus-refinement branch -> (fpt (n,K), dwell (n,K), stats (n,6), idx (n,3), status)"""
    sig = np.ascontiguousarray(sig, dtype=np.float32)
    n, stride = sig.shape
    K = rp.barcode_keep_events
    a_start = np.ascontiguousarray(a_start, dtype=np.int32)
    a_end = np.ascontiguousarray(a_end, dtype=np.int32)
    okp = None if ok is None else np.ascontiguousarray(ok, dtype=np.uint8)
    fpt = np.full((n, K), np.nan)
    dwell = np.zeros((n, K), dtype=np.int64)
    stats = np.full((n, 6), np.nan)
    idx = np.full((n, 3), -1, dtype=np.int32)
    status = np.zeros(n, dtype=np.int32)
    pc, rc = params.to_c(), rp.to_c()
    lib().wdx_oracle_fingerprint_refine_batch(_p(sig), n, stride, _p(a_start), _p(a_end), None if okp is None else _p(okp),
                                              C.byref(pc), C.byref(rc), _p(fpt), _p(dwell), _p(stats), _p(idx), _p(status))
    return fpt, dwell, stats, idx, status


def fingerprint_packed(sig, off, a_start, a_end, params: SegParams):
    sig = np.ascontiguousarray(sig, dtype=np.float32)
    off = np.ascontiguousarray(off, dtype=np.int64)
    n = off.size - 1
    K = params.barcode_num_events
    a_start = np.ascontiguousarray(a_start, dtype=np.int32)
    a_end = np.ascontiguousarray(a_end, dtype=np.int32)
    fpt = np.full((n, K), np.nan)
    dwell = np.zeros((n, K), dtype=np.int64)
    stats = np.full((n, 6), np.nan)
    status = np.zeros(n, dtype=np.int32)
    pc = params.to_c()
    lib().wdx_oracle_fingerprint_packed(_p(sig), _p(off), n, _p(a_start), _p(a_end), C.byref(pc), _p(fpt), _p(dwell), _p(stats), _p(status))
    return fpt, dwell, stats, status


def dtw_distance(a, b, window=None, penalty=None) -> float:
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    return lib().wdx_oracle_dtw_distance(_p(a), a.size, _p(b), b.size, int(window or 0), float(penalty or 0.0))


def dtw_matrix(X, Y, window=None, penalty=None) -> np.ndarray:
    X = np.ascontiguousarray(X, dtype=np.float64)
    Y = np.ascontiguousarray(Y, dtype=np.float64)
    out = np.empty((X.shape[0], Y.shape[0]), dtype=np.float32)
    lib().wdx_oracle_dtw_matrix(_p(X), X.shape[0], _p(Y), Y.shape[0], X.shape[1], int(window or 0), float(penalty or 0.0), _p(out))
    return out


def argmin_rows(D) -> np.ndarray:
    D = np.ascontiguousarray(D, dtype=np.float32)
    out = np.empty(D.shape[0], dtype=np.int32)
    lib().wdx_oracle_argmin_rows(_p(D), D.shape[0], D.shape[1], _p(out))
    return out


def svm_params(svc):
    """The arrays libsvm's predict_probability needs, from a fitted sklearn SVC(kernel="precomputed",
    probability=True): (n_support i32[k], support i32[nSV], dual_coef f64[(k-1),nSV], rho f64[npairs],
    probA, probB)."""
    k = svc.classes_.size
    rho = -np.asarray(svc._intercept_, dtype=np.float64)          # libsvm's rho (sklearn stores -rho)
    return (np.ascontiguousarray(svc._n_support, dtype=np.int32), np.ascontiguousarray(svc.support_, dtype=np.int32),
            np.ascontiguousarray(svc._dual_coef_, dtype=np.float64), np.ascontiguousarray(rho),
            np.ascontiguousarray(svc._probA, dtype=np.float64), np.ascontiguousarray(svc._probB, dtype=np.float64), k)


def svm_predict_proba(K, n_support, support, dual_coef, rho, probA, probB, want_dec=False):
    K = np.ascontiguousarray(K, dtype=np.float64)
    n, n_train = K.shape
    k = n_support.size
    prob = np.empty((n, k), dtype=np.float64)
    dec = np.empty((n, k * (k - 1) // 2), dtype=np.float64) if want_dec else None
    lib().wdx_oracle_svm_predict_proba(_p(K), n, n_train, k, _p(n_support), _p(support), _p(dual_coef), _p(rho),
                                       _p(probA), _p(probB), _p(prob), None if dec is None else _p(dec))
    return (prob, dec) if want_dec else prob

"""MI355X drop-in for ``warpdemux.parallel_distances`` (reference file of the same name).

Same four public functions, argument meanings and error behaviour as the reference
(/root/reference/warpdemux/parallel_distances.py:24-198); the DTW arithmetic runs in the HIP
engine (libwdx_hip.so) instead of dtaidistance's C code.  ``n_jobs`` / ``block_size`` keep their
validation semantics but no process pool is started: one GPU launch covers the whole matrix.

Install in a WarpDemuX process with ``warpdemux_amd.install()``.
"""
from __future__ import annotations

import logging
from typing import Optional, Tuple

import numpy as np

from . import _lib


def _as_f64_2d(a, name):
    a = np.asarray(a)
    if a.ndim != 2:
        raise ValueError(f"{name} must be 2-dimensional, got shape {a.shape}")
    return np.ascontiguousarray(a, dtype=np.float64)


def _dtw(X: np.ndarray, Y: np.ndarray, window, penalty, want_argmin=False, device=None):
    if X.shape[1] != Y.shape[1]:
        # np.vstack([X, Y]) in the reference raises ValueError on a column mismatch
        raise ValueError(
            f"all the input array dimensions except for the concatenation axis must match exactly, "
            f"but got {X.shape[1]} and {Y.shape[1]} columns"
        )
    ctx = _lib.default_context(device)
    out = np.empty((X.shape[0], Y.shape[0]), dtype=np.float32)
    am = np.empty(X.shape[0], dtype=np.int32) if want_argmin else None
    L = _lib.load()
    _lib.check(
        L.wdx_dtw_matrix(
            ctx.handle, _lib.ptr(X), X.shape[0], _lib.ptr(Y), Y.shape[0], X.shape[1],
            int(window) if window else 0, float(penalty) if penalty else 0.0, _lib.ptr(out), _lib.ptr(am),
        )
    )
    return (out, am) if want_argmin else out


def compute_block_distance(
    block_indices: Tuple[np.ndarray, np.ndarray],
    X,
    window=None,
    penalty=None,
    **kwargs,
):
    """(i, j, float32 block) for the row-index blocks ``i`` x ``j`` of ``X``
    (reference: parallel_distances.py:24-45)."""
    if kwargs:
        raise NotImplementedError(f"unsupported dtaidistance options: {sorted(kwargs)}")
    i, j = block_indices
    X = _as_f64_2d(X, "X")
    return (i, j, _dtw(np.ascontiguousarray(X[i]), np.ascontiguousarray(X[j]), window, penalty))


def distance_matrix_to(
    X,
    Y,
    window: Optional[int] = None,
    penalty: Optional[float] = None,
    block_size: Optional[int] = None,
    n_jobs: int = -1,
    pbar: bool = False,
    pbar_kwargs: dict = {},
):
    """float32 (nX, nY) matrix of banded DTW distances (reference: parallel_distances.py:48-84)."""
    if n_jobs == 1:
        return _dtw(_as_f64_2d(X, "X"), _as_f64_2d(Y, "Y"), window, penalty)
    if block_size is None:
        msg = "block_size must be specified when using parallel."
        logging.error(msg)
        raise ValueError(msg)
    return parallel_distance_matrix_to(
        X, Y, block_size=block_size, n_jobs=n_jobs, window=window, penalty=penalty, pbar=pbar,
        pbar_kwargs=pbar_kwargs,
    )


def parallel_distance_matrix_to(
    X,
    Y,
    block_size: int = 1000,
    n_jobs: int = 6,
    window: Optional[int] = None,
    penalty: Optional[float] = None,
    pbar: bool = False,
    pbar_kwargs: dict = {},
    **kwargs,
):
    """Reference: parallel_distances.py:87-136 (stack X over Y, then the subset form)."""
    X = _as_f64_2d(X, "X")
    Y = _as_f64_2d(Y, "Y")
    if X.shape[1] != Y.shape[1]:
        raise ValueError("X and Y must have the same number of columns")
    return parallel_distance_matrix(
        np.vstack([X, Y]),
        block_size=block_size,
        n_jobs=n_jobs,
        subset=((0, X.shape[0]), (X.shape[0], X.shape[0] + Y.shape[0])),
        window=window,
        penalty=penalty,
        pbar=pbar,
        pbar_kwargs=pbar_kwargs,
        **kwargs,
    )


def parallel_distance_matrix(
    X,
    block_size: int = 1000,
    n_jobs: int = 6,
    subset: Optional[Tuple[Tuple[int, int], Tuple[int, int]]] = None,
    window: Optional[int] = None,
    penalty: Optional[float] = None,
    pbar: bool = False,
    pbar_kwargs: dict = {},
    **kwargs,
):
    """Rows r1 x rows r2 of ``X`` (all-vs-all when ``subset`` is None), float32
    (reference: parallel_distances.py:139-198).  The reference tiles the index space into
    ``block_size`` blocks for its process pool; every block is a full rectangle there
    (``only_triu`` never bites because block columns are offset past the rows), so one launch over
    the whole rectangle returns the same matrix."""
    if kwargs:
        raise NotImplementedError(f"unsupported dtaidistance options: {sorted(kwargs)}")
    X = _as_f64_2d(X, "X")
    if subset:
        (r1_start, r1_end), (r2_start, r2_end) = subset
    else:
        r1_start, r1_end, r2_start, r2_end = 0, X.shape[0], 0, X.shape[0]
    if block_size is None or block_size <= 0:
        raise ValueError("block_size must be a positive integer")
    A = np.ascontiguousarray(X[r1_start:r1_end])
    B = np.ascontiguousarray(X[r2_start:r2_end])
    if pbar:
        from tqdm import tqdm

        with tqdm(total=1, desc="Computing Kernel Matrix", **(pbar_kwargs or {})) as bar:
            out = _dtw(A, B, window, penalty)
            bar.update(1)
        return out
    return _dtw(A, B, window, penalty)


def nearest_reference(X, Y, window=None, penalty=None):
    """(float32 distances (nX,nY), int32 argmin per read) -- SURVEY.md §8 row B3."""
    return _dtw(_as_f64_2d(X, "X"), _as_f64_2d(Y, "Y"), window, penalty, want_argmin=True)

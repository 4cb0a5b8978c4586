"""MI355X counterpart of ``warpdemux.models.dtw_svm.DTW_SVM``.

Same ``predict`` signature and outputs as the reference (/root/reference/warpdemux/models/dtw_svm.py:54-98):
DTW distances to ``_X`` -> ``exp(-gamma * d**pwr_dist)`` -> ``SVC.predict_proba`` -> ``process_probs``
(models/utils.py:45-61) -> optional DataFrame (models/utils.py:36-43), with the whole chain on the device
(the (nX, len(_X)) distance matrix never leaves HBM).  Build one from a loaded reference model with
``DTW_SVM.from_reference(model)`` -- the fitted scikit-learn ``SVC`` is only read for its parameters.
"""
from __future__ import annotations

import ctypes as C
import logging
from typing import Dict, Optional, Tuple, Union

import numpy as np

from . import _lib


def predictions_to_df(y_pred, y_prob, conf, label_mapper):
    """models/utils.py:36-43"""
    import pandas as pd

    return pd.DataFrame(
        {
            "predicted_barcode": y_pred,
            "confidence_score": conf.round(3),
            **{f"p{label_mapper[i]:02d}": y_prob[:, i].round(4) for i in range(y_prob.shape[1])},
        }
    )


class DTW_SVM:
    """Holds the reference fingerprints and the SVC parameters resident on one GPU context."""

    def __init__(self, _X: np.ndarray, n_support, support, dual_coef, rho, probA, probB,
                 label_mapper: Dict[int, int], thresholds: Optional[np.ndarray], window: int, penalty: float,
                 gamma: float = 1.0, pwr_dist: int = 1, block_size: Optional[int] = None, device: Optional[int] = None):
        self._X = np.ascontiguousarray(_X, dtype=np.float64)
        self.window, self.penalty, self.block_size = window, penalty, block_size
        self.gamma, self.pwr_dist = float(gamma), int(pwr_dist)
        self.label_mapper = dict(label_mapper)
        self.thresholds = None if thresholds is None else np.ascontiguousarray(thresholds, dtype=np.float64)
        self._n_support = np.ascontiguousarray(n_support, dtype=np.int32)
        self._support = np.ascontiguousarray(support, dtype=np.int32)
        self._dual_coef = np.ascontiguousarray(dual_coef, dtype=np.float64)
        self._rho = np.ascontiguousarray(rho, dtype=np.float64)
        self._probA = np.ascontiguousarray(probA, dtype=np.float64)
        self._probB = np.ascontiguousarray(probB, dtype=np.float64)
        self.n_classes = int(self._n_support.size)
        self._label_arr = np.array([self.label_mapper[i] for i in range(self.n_classes)], dtype=np.int32)
        self._device = device

    @classmethod
    def from_reference(cls, model, device: Optional[int] = None) -> "DTW_SVM":
        """From a reference ``DTW_SVM`` instance (a loaded model_files/*.joblib)."""
        svc = model.model
        if getattr(svc, "kernel", None) != "precomputed" or not getattr(svc, "probability", False):
            raise ValueError("expected SVC(kernel='precomputed', probability=True)")
        return cls(
            _X=model._X, n_support=svc._n_support, support=svc.support_, dual_coef=svc._dual_coef_,
            rho=-np.asarray(svc._intercept_, dtype=np.float64), probA=svc._probA, probB=svc._probB,
            label_mapper=model.label_mapper, thresholds=model.thresholds, window=model.window,
            penalty=model.penalty, gamma=model.gamma, pwr_dist=model.pwr_dist, block_size=model.block_size,
            device=device,
        )

    @property
    def is_trained(self):
        return self._X is not None

    @property
    def num_bcs(self):
        return self.n_classes

    def to_c(self) -> "_lib.SvmModelC":
        """wdx_svm_model view of the host arrays (valid while ``self`` is alive)."""
        return _lib.SvmModelC(
            self.n_classes, int(self._support.size), int(self._X.shape[0]), self.pwr_dist, self.gamma,
            self._n_support.ctypes.data, self._support.ctypes.data, self._dual_coef.ctypes.data,
            self._rho.ctypes.data, self._probA.ctypes.data, self._probB.ctypes.data, self._label_arr.ctypes.data,
            None if self.thresholds is None else self.thresholds.ctypes.data,
        )

    def _ensure_resident(self):
        """References and SVM parameters on the process's context.  The context holds ONE reference set and ONE
        model at a time and other calls (distance_matrix_to, set_references, another DTW_SVM) may have replaced
        either: the reference set is re-submitted on every call (the library compares a content hash and uploads
        only on change), the model whenever this object is not the one the context last received."""
        ctx = _lib.default_context(self._device)
        L = _lib.load()
        _lib.check(L.wdx_set_refs(ctx.handle, _lib.ptr(self._X), self._X.shape[0], self._X.shape[1],
                                  int(self.window) if self.window else 0, float(self.penalty) if self.penalty else 0.0))
        if getattr(ctx, "_svm_owner", None) is not self:
            ctx._svm_owner = None
            m = self.to_c()
            _lib.check(L.wdx_svm_set_model(ctx.handle, C.byref(m)))
            ctx._svm_owner = self
        return ctx

    def predict(self, X: np.ndarray, nproc: int = -1, block_size: Optional[int] = None, pbar: bool = False,
                pbar_kwargs: dict = {}, return_df: bool = False) -> Union[Tuple[np.ndarray, np.ndarray], "object"]:
        """(y_pred, y_prob) or the predictions DataFrame -- dtw_svm.py:54-98.  ``nproc`` / ``block_size``
        keep the reference's validation (block_size required when nproc != 1) but nothing is forked."""
        if not self.is_trained:
            msg = "Model not trained yet."
            logging.error(msg)
            raise ValueError(msg)
        X = np.asarray(X)
        if X.ndim == 1:
            X = X.reshape(1, -1)
        if X.shape[1] != self._X.shape[1]:
            raise ValueError("X must have the same number of columns as the training data "
                             f" ({self._X.shape[1]}).")
        if nproc != 1 and (self.block_size if block_size is None else block_size) is None:
            msg = "block_size must be specified when using parallel."
            logging.error(msg)
            raise ValueError(msg)
        X = np.ascontiguousarray(X, dtype=np.float64)
        n = X.shape[0]
        ctx = self._ensure_resident()
        y_prob = np.empty((n, self.n_classes), dtype=np.float64)
        y_pred = np.empty(n, dtype=np.int32)
        conf = np.empty(n, dtype=np.float64)
        _lib.check(_lib.load().wdx_dtw_svm_predict(ctx.handle, _lib.ptr(X), n, _lib.ptr(y_prob), _lib.ptr(y_pred),
                                                   _lib.ptr(conf)))
        y_pred = y_pred.astype(np.int64)
        if return_df:
            return predictions_to_df(y_pred, y_prob, conf, self.label_mapper)
        return y_pred, y_prob

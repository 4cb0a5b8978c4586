"""Live path (BASELINE config 5): the reads of one 100 ms chunk round in one device call.

The reference runs two per-read worker loops (live_balancing/worker.py): ``segmentation_worker`` (:26-96 --
extract_adapter(0, polya_start), median/MAD clip, segment_signal, normalize, keep the last K events) and
``classification_worker`` (:99-131 -- ``model.predict(fpt, nproc=1)``).  Here

* :class:`LiveDemux` owns one engine context per thread (its own HIP stream and page-locked staging
  buffers) and turns a tick's reads into fingerprints, distances, calls and -- with a ``DTW_SVM`` -- class
  probabilities with ONE C-ABI call (``wdx_live_tick``);
* :func:`demux_worker` is the queue-to-queue mirror of the two reference workers: it drains whatever
  ``ReadObject``s the session queued during the tick, processes them as one batch and emits them with the
  fields the reference's ``balance_worker`` reads (``data_arr`` = ``y_prob.reshape(1, -1)``, ``is_outlier``,
  two ``time_per_step`` entries).

One LiveDemux per worker thread (live_balancing/session.py:162-169 starts thread pools): calls through
different contexts overlap on the device.
"""
from __future__ import annotations

import ctypes as C
import queue as _queue
import time
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

from . import _lib
from .sig_proc import SegParams


@dataclass
class TickResult:
    status: np.ndarray              # (n,) int32 WDX_READ_*
    call: np.ndarray                # (n,) int32 nearest reference, -1 for failed reads
    dist: Optional[np.ndarray]      # (n, nY) float32 (NaN rows for failed reads)
    fpt: Optional[np.ndarray]       # (n, K) float64
    prob: Optional[np.ndarray]      # (n, k) float64 -- y_prob of DTW_SVM.predict (with a model)
    pred: Optional[np.ndarray]      # (n,) int64 barcode label, -1 = outlier / failed read (with a model)
    conf: Optional[np.ndarray]      # (n,) float64 top1 - top2 margin (with a model)


class LiveDemux:
    """``refs``: (nY, K) reference fingerprints, or pass ``model`` = a :class:`warpdemux_amd.models.DTW_SVM`
    (its ``_X``/window/penalty become the references and its SVM tail runs on the device too)."""

    def __init__(self, refs=None, window=None, penalty=None, params: Optional[SegParams] = None, *, model=None,
                 device: int = 0, max_reads: int = 512, max_samples: int = 10000):
        self.L = _lib.load()
        self.ctx = _lib.Context(device)      # this object's own context = own stream + staging buffers
        self.model = model
        if model is not None:
            refs, window, penalty = model._X, model.window, model.penalty
        refs = np.ascontiguousarray(refs, dtype=np.float64)
        if refs.ndim != 2:
            raise ValueError("refs must be (nY, K)")
        self.nY, self.K = refs.shape
        self.params = params or SegParams(barcode_num_events=self.K)
        if self.params.barcode_num_events != self.K:
            raise ValueError(f"barcode_num_events ({self.params.barcode_num_events}) must equal the reference length ({self.K})")
        self._pc = self.params.to_c()
        _lib.check(self.L.wdx_set_refs(self.ctx.handle, _lib.ptr(refs), self.nY, self.K,
                                       int(window) if window else 0, float(penalty) if penalty else 0.0))
        self.k = 0
        if model is not None:
            self._m = model.to_c()
            _lib.check(self.L.wdx_svm_set_model(self.ctx.handle, C.byref(self._m)))
            self.k = model.n_classes
        self._cap = 0
        self._reserve(max_reads)
        # first tick at full size now: staging buffers and workspaces are allocated before the run starts
        if max_reads > 0 and max_samples > 0:
            z = np.zeros(max_samples, dtype=np.float32)
            self.tick([z] * max_reads, np.zeros(max_reads, np.int32), np.full(max_reads, max_samples, np.int32))

    def _reserve(self, n):
        if n <= self._cap:
            return
        self._cap = n
        self._rows = (C.c_void_p * n)()
        self._len = np.empty(n, dtype=np.int32)
        self._status = np.empty(n, dtype=np.int32)
        self._call = np.empty(n, dtype=np.int32)
        self._dist = np.empty((n, self.nY), dtype=np.float32)
        self._fpt = np.empty((n, self.K), dtype=np.float64)
        self._prob = np.empty((n, max(self.k, 1)), dtype=np.float64)
        self._pred = np.empty(n, dtype=np.int32)
        self._conf = np.empty(n, dtype=np.float64)

    def tick(self, rows: Sequence[np.ndarray], adapter_start, adapter_end, success=None, want_dist=True,
             want_fpt=False) -> TickResult:
        """rows: one float32 1-D array per read (ragged); adapter_start/end per read (the live caller passes 0 and
        ``polya_start``, worker.py:39-44).  Returned arrays are fresh copies."""
        n = len(rows)
        self._reserve(n)
        keep = []
        for i, r in enumerate(rows):
            if r.dtype != np.float32 or not r.flags.c_contiguous:
                r = np.ascontiguousarray(r, dtype=np.float32)
                keep.append(r)
            self._rows[i] = r.ctypes.data
            self._len[i] = r.size
        a_s = np.ascontiguousarray(adapter_start, dtype=np.int32)
        a_e = np.ascontiguousarray(adapter_end, dtype=np.int32)
        if a_s.shape != (n,) or a_e.shape != (n,):
            raise ValueError("adapter_start/adapter_end must have one entry per read")
        ok = None if success is None else np.ascontiguousarray(success, dtype=np.uint8)
        svm = self.k > 0
        _lib.check(self.L.wdx_live_tick(
            self.ctx.handle, self._rows, _lib.ptr(self._len), n, _lib.ptr(a_s), _lib.ptr(a_e), _lib.ptr(ok),
            C.byref(self._pc), self.nY, int(svm), _lib.ptr(self._fpt) if want_fpt else None,
            _lib.ptr(self._dist) if want_dist else None, _lib.ptr(self._call), _lib.ptr(self._status),
            _lib.ptr(self._prob) if svm else None, _lib.ptr(self._pred) if svm else None,
            _lib.ptr(self._conf) if svm else None))
        status = self._status[:n].copy()
        pred = None
        if svm:
            pred = self._pred[:n].astype(np.int64)
            pred[status != 0] = -1
        return TickResult(status, self._call[:n].copy(), self._dist[:n].copy() if want_dist else None,
                          self._fpt[:n].copy() if want_fpt else None, self._prob[:n, :self.k].copy() if svm else None,
                          pred, self._conf[:n].copy() if svm else None)

    def close(self):
        self.ctx.close()


def demux_worker(input_queue, output_queue, live: LiveDemux, tick_seconds: float = 0.1, max_reads: int = 512) -> None:
    """Queue-to-queue mirror of ``segmentation_worker`` + ``classification_worker`` (worker.py:26-131), batched per
    tick: blocks for the first ReadObject, then takes everything else already queued (at most ``max_reads``), runs
    ONE ``LiveDemux.tick`` and forwards each object with ``data_arr = y_prob.reshape(1, -1)``, ``is_outlier`` and
    two appended ``time_per_step`` entries (segmentation, classification: the tick's wall time split evenly -- the
    device does both in one call).  ``None`` stops the worker (and is forwarded).  Reads whose fingerprint fails
    are dropped like the reference's "no segments" branch (worker.py:75-79).  Needs ``live`` built with a model."""
    if live.k == 0:
        raise ValueError("demux_worker needs a LiveDemux with a DTW_SVM model")
    while True:
        first = input_queue.get()
        if first is None:
            output_queue.put(None)
            return
        batch = [first]
        stop = False
        while len(batch) < max_reads:
            try:
                nxt = input_queue.get_nowait()
            except _queue.Empty:
                break
            if nxt is None:
                stop = True
                break
            batch.append(nxt)
        t0 = time.time()
        rows = [np.asarray(o.data_arr, dtype=np.float32).ravel() for o in batch]
        a_e = np.array([o.polya_start for o in batch], dtype=np.int32)
        r = live.tick(rows, np.zeros(len(batch), np.int32), a_e, want_dist=False)
        dt = (time.time() - t0) / 2
        for i, o in enumerate(batch):
            if r.status[i] != 0:
                continue
            o.data_arr = r.prob[i].reshape(1, -1)
            o.is_outlier = bool(r.pred[i] == -1)
            o.time_per_step.append(dt)
            o.time_per_step.append(dt)
            output_queue.put(o)
        if stop:
            output_queue.put(None)
            return

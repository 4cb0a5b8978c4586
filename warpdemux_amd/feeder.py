"""Many worker processes, ONE GPU-facing process: the drop-in for the reference's `-j 8..16` forked workers
(file_proc.py:1197-1243: a ProcessPoolExecutor whose workers each run fingerprint + model on their minibatches,
file_proc.py:380-454).

Sixteen processes that each drive the GPU through their own context run at 40 % of the rate of four (the device
time-slices the processes' queues).  Here the PARENT creates a `Feeder` before it forks its workers: a ring of
minibatch slots in shared memory plus one forked process that owns the engine context, page-locks the ring and keeps
up to eight minibatches in flight (`wdx_feeder_serve`).  A worker's call copies the adapter windows of its minibatch
into a free slot and sleeps until the results are there (`wdx_feeder_run`: no context, no HIP call in the worker).

What comes back is what the reference's worker needs from a minibatch (file_proc.py:418-450):

* `fingerprint_batch(...)` -> `sig_proc.FingerprintBatch` (fingerprint, dwell times, six statistics, status): the
  ReadResults `save_fpts_signals` / `save_detected_boundaries` take (file_proc.py:707-754), = `sig_proc.fingerprint_batch`;
* `predict(X, return_df=...)` -> what `DTW_SVM.predict` returns (models/dtw_svm.py:54-98) for fingerprints the worker holds;
* `detect_and_predict(...)` -> both from ONE pass over the rows (the fingerprints never leave the device between the two);
* `demux_batch(...)` -> `sig_proc.DemuxBatch` (status, nearest-reference call, distance rows), = `sig_proc.demux_batch`.

    feeder = Feeder(model=DTW_SVM.from_reference(model), params=SegParams.from_spc(spc), max_reads=1000, stride=10000)
    with ProcessPoolExecutor(P, mp_context=multiprocessing.get_context("fork")) as pool:   # workers inherit `feeder`
        ... in a worker:  fb, preds = feeder.detect_and_predict(minibatch, adapter_start, adapter_end, success, return_df=True)
    feeder.close()

A worker that dies while it holds a slot does not cost the ring that slot, and a feeder process that dies is noticed
by the workers (`WdxNoDevice`) even while it is a zombie nobody has reaped (wdx_feeder.hip).
"""
from __future__ import annotations

import ctypes as C
import multiprocessing as mp
import os
from multiprocessing import shared_memory
from typing import Optional

import numpy as np

from . import _lib
from .sig_proc import DemuxBatch, FingerprintBatch, SegParams

MAX_SLOTS = 32      # ring slots (WDX_FEEDER_MAX_RING_SLOTS); the feeder keeps at most 8 of them in flight on the device


def _serve(shm_name: str, refs, window, penalty, model, device: int, ready):
    """The GPU-facing process (forked from a parent that never touched the GPU)."""
    shm = shared_memory.SharedMemory(name=shm_name)
    rc = 1
    try:
        L = _lib.load()
        ctx = _lib.Context(device)
        _lib.check(L.wdx_set_refs(ctx.handle, _lib.ptr(refs), refs.shape[0], refs.shape[1], int(window) if window else 0,
                                  float(penalty) if penalty else 0.0))
        if model is not None:
            m = model.to_c()
            _lib.check(L.wdx_svm_set_model(ctx.handle, C.byref(m)))
        base = C.addressof(C.c_char.from_buffer(shm.buf))
        # (wdx_feeder_serve announces itself in the ring -- server_pid + heartbeat -- once the ring is page-locked; the
        # parent polls wdx_feeder_alive after this event)
        ready.set()
        _lib.check(L.wdx_feeder_serve(ctx.handle, C.c_void_p(base)))
        ctx.close()
        rc = 0
    except BaseException:  # noqa: BLE001  (reported through the exit code and stderr; the ring is stopped below)
        import sys
        import traceback

        traceback.print_exc(file=sys.stderr)
        try:
            base = C.addressof(C.c_char.from_buffer(shm.buf))
            _lib.load().wdx_feeder_stop(C.c_void_p(base))
        except Exception:  # noqa: BLE001
            pass
        ready.set()
    finally:
        os._exit(rc)   # (no interpreter teardown in the forked child: the parent owns the shared memory)


class Feeder:
    """Create in the parent BEFORE forking the workers (the parent itself makes no GPU call); the workers use the
    inherited object.  `max_reads` x `stride` = the largest minibatch a slot holds (file_proc's 1000 x sig_preload_size).

    Either `refs` (+ `window`, `penalty`): nearest-reference calls only -- or `model`, a `warpdemux_amd.models.DTW_SVM`
    (`DTW_SVM.from_reference(loaded_model)`): its `_X` are the references and `predict` / `detect_and_predict` are served."""

    def __init__(self, refs=None, window=None, penalty=None, params: Optional[SegParams] = None, max_reads: int = 1000,
                 stride: int = 10000, n_slots: int = 16, device: int = 0, start_timeout: float = 120.0, model=None):
        if model is not None:
            if refs is not None:
                raise ValueError("pass either refs or model (whose _X are the references)")
            refs, window, penalty = model._X, model.window, model.penalty
        if refs is None:
            raise ValueError("refs or model is required")
        refs = np.ascontiguousarray(refs, dtype=np.float64)
        if refs.ndim != 2:
            raise ValueError("refs must be (nY, L)")
        if not 1 <= int(n_slots) <= MAX_SLOTS:
            raise ValueError(f"n_slots must be in [1, {MAX_SLOTS}]")
        self.params = params or SegParams(barcode_num_events=int(refs.shape[1]))
        if self.params.barcode_num_events != refs.shape[1]:
            raise ValueError("barcode_num_events must equal the reference length")
        self.model = model
        self.n_classes = int(model.n_classes) if model is not None else 0
        self.label_mapper = dict(model.label_mapper) if model is not None else None
        self.nY, self.K = (int(v) for v in refs.shape)
        self.max_reads, self.stride, self.n_slots = int(max_reads), int(stride), int(n_slots)
        self.L = _lib.load()
        geo = _lib.FeederGeometryC(self.n_slots, self.K, self.n_classes, 0, self.max_reads, self.stride, self.nY)
        nbytes = int(self.L.wdx_feeder_ring_bytes(C.byref(geo)))
        if nbytes == 0:
            raise ValueError("bad ring geometry")
        self._shm = shared_memory.SharedMemory(create=True, size=nbytes)
        self._owner = os.getpid()
        self._base = C.addressof(C.c_char.from_buffer(self._shm.buf))
        pc = self.params.to_c()
        _lib.check(self.L.wdx_feeder_ring_init(C.c_void_p(self._base), nbytes, C.byref(geo), C.byref(pc)))
        ctx = mp.get_context("fork")
        ready = ctx.Event()
        self._proc = ctx.Process(target=_serve, args=(self._shm.name, refs, window, penalty, model, int(device), ready),
                                 daemon=True)
        self._proc.start()
        import time

        up = ready.wait(start_timeout)
        t_end = time.monotonic() + 30.0
        while up and self._proc.is_alive() and self.L.wdx_feeder_alive(C.c_void_p(self._base)) != 1 and time.monotonic() < t_end:
            time.sleep(0.002)      # context, references, page-locking the ring: the feeder is up when it says so in the ring
        if not up or not self._proc.is_alive() or self.L.wdx_feeder_alive(C.c_void_p(self._base)) != 1:
            self.close()
            raise _lib.WdxError("the feeder process did not come up (see its stderr)")

    # ---- one minibatch ------------------------------------------------------------------------------------------------
    def _run(self, signals, adapter_start, adapter_end, success, want: int):
        sig = np.asarray(signals)
        if sig.ndim != 2:
            raise ValueError("signals must be a 2-D (n_reads, stride) array")
        sig = np.ascontiguousarray(sig, dtype=np.float32)
        n, stride = sig.shape
        a_s = np.ascontiguousarray(adapter_start, dtype=np.int32)
        a_e = np.ascontiguousarray(adapter_end, dtype=np.int32)
        if a_s.shape != (n,) or a_e.shape != (n,):
            raise ValueError("adapter_start/adapter_end must have one entry per read")
        ok = None if success is None else np.ascontiguousarray(success, dtype=np.uint8)
        if want & _lib.WANT_SVM and self.model is None:
            raise ValueError("this feeder was created without a model (Feeder(model=DTW_SVM...))")
        out = {
            "status": np.empty(n, dtype=np.int32),
            "call": np.empty(n, dtype=np.int32),
            "dist": np.empty((n, self.nY), dtype=np.float32) if want & _lib.WANT_DIST else None,
            "fpt": np.empty((n, self.K), dtype=np.float64) if want & _lib.WANT_FPT else None,
            "dwell": np.empty((n, self.K), dtype=np.int64) if want & _lib.WANT_DWELL else None,
            "stats": np.empty((n, 6), dtype=np.float64) if want & _lib.WANT_STATS else None,
            "prob": np.empty((n, self.n_classes), dtype=np.float64) if want & _lib.WANT_SVM else None,
            "pred": np.empty(n, dtype=np.int32) if want & _lib.WANT_SVM else None,
            "conf": np.empty(n, dtype=np.float64) if want & _lib.WANT_SVM else None,
        }
        job = _lib.FeederJobC(_lib.addr(sig), n, stride, _lib.addr(a_s), _lib.addr(a_e), _lib.addr(ok), int(want), 0,
                              *[_lib.addr(out[k]) for k in ("status", "call", "dist", "fpt", "dwell", "stats", "prob", "pred", "conf")])
        _lib.check(self.L.wdx_feeder_run(C.c_void_p(self._base), C.byref(job)))
        return out

    def demux_batch(self, signals, adapter_start, adapter_end, success=None, want_dist: bool = True) -> DemuxBatch:
        """Status, nearest-reference call and (optionally) the distance rows -- `sig_proc.demux_batch`'s result, bit for
        bit.  Callable from any process that inherited this object; blocks until the results are there."""
        o = self._run(signals, adapter_start, adapter_end, success, _lib.WANT_DIST if want_dist else 0)
        return DemuxBatch(o["status"], o["call"], o["dist"], None)

    def fingerprint_batch(self, signals, adapter_start, adapter_end, success=None) -> FingerprintBatch:
        """`sig_proc.fingerprint_batch`'s result (fingerprints, dwell times, the six statistics, status), bit for bit:
        what `sig_proc.read_results_from_batch` turns into the reference's ReadResult records."""
        o = self._run(signals, adapter_start, adapter_end, success, _lib.WANT_FPT | _lib.WANT_DWELL | _lib.WANT_STATS)
        return FingerprintBatch(o["fpt"], o["dwell"], o["stats"], o["status"])

    def detect_and_predict(self, signals, adapter_start, adapter_end, success=None, return_df: bool = False):
        """The two halves of the reference worker's minibatch (file_proc.py:418-450) from one pass over the rows:
        `(FingerprintBatch, predictions)` with predictions = `(y_pred, y_prob)` or, with ``return_df``, the predictions
        DataFrame of `DTW_SVM.predict(np.vstack(fpts), return_df=True)` -- one row per SUCCESSFUL read, in read order,
        like the reference, which only ever shows the model the successful fingerprints."""
        o = self._run(signals, adapter_start, adapter_end, success,
                      _lib.WANT_FPT | _lib.WANT_DWELL | _lib.WANT_STATS | _lib.WANT_SVM)
        fb = FingerprintBatch(o["fpt"], o["dwell"], o["stats"], o["status"])
        okr = o["status"] == 0
        y_pred, y_prob, conf = o["pred"][okr].astype(np.int64), o["prob"][okr], o["conf"][okr]
        if return_df:
            from .models import predictions_to_df

            return fb, predictions_to_df(y_pred, y_prob, conf, self.label_mapper)
        return fb, (y_pred, y_prob)

    def predict(self, X, return_df: bool = False):
        """`DTW_SVM.predict` (models/dtw_svm.py:54-98) through the feeder: (y_pred, y_prob) or the predictions DataFrame."""
        if self.model is None:
            raise ValueError("this feeder was created without a model (Feeder(model=DTW_SVM...))")
        X = np.asarray(X)
        if X.ndim == 1:
            X = X.reshape(1, -1)
        if X.shape[1] != self.K:
            raise ValueError(f"X must have the same number of columns as the training data  ({self.K}).")
        X = np.ascontiguousarray(X, dtype=np.float64)
        n = X.shape[0]
        y_prob = np.empty((n, self.n_classes), dtype=np.float64)
        y_pred = np.empty(n, dtype=np.int32)
        conf = np.empty(n, dtype=np.float64)
        _lib.check(self.L.wdx_feeder_predict(C.c_void_p(self._base), _lib.ptr(X), n, _lib.ptr(y_prob), _lib.ptr(y_pred),
                                             _lib.ptr(conf)))
        y_pred = y_pred.astype(np.int64)
        if return_df:
            from .models import predictions_to_df

            return predictions_to_df(y_pred, y_prob, conf, self.label_mapper)
        return y_pred, y_prob

    # ---- housekeeping -------------------------------------------------------------------------------------------------
    def alive(self) -> bool:
        """True while the feeder process serves the ring."""
        return self._base is not None and self.L.wdx_feeder_alive(C.c_void_p(self._base)) == 1

    def served(self) -> int:
        v = C.c_int64(0)
        _lib.check(self.L.wdx_feeder_served(C.c_void_p(self._base), C.byref(v)))
        return int(v.value)

    def stats(self) -> dict:
        """{'served': minibatches handed back, 'reclaimed': slots taken back from dead workers, 'free_slots': now}"""
        a, b, c = C.c_int64(0), C.c_int64(0), C.c_int32(0)
        _lib.check(self.L.wdx_feeder_stats(C.c_void_p(self._base), C.byref(a), C.byref(b), C.byref(c)))
        return {"served": int(a.value), "reclaimed": int(b.value), "free_slots": int(c.value), "n_slots": self.n_slots}

    def close(self):
        """Parent only: stop the feeder process and release the ring."""
        if self._shm is None or os.getpid() != self._owner:
            return
        try:
            self.L.wdx_feeder_stop(C.c_void_p(self._base))
        except Exception:  # noqa: BLE001
            pass
        if self._proc is not None:
            self._proc.join(30)
            if self._proc.is_alive():
                self._proc.terminate()
                self._proc.join(10)
        self._base = None
        shm, self._shm = self._shm, None
        try:
            shm.close()
        except BufferError:   # (ctypes views of the buffer are still referenced somewhere: unlink regardless)
            pass
        shm.unlink()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

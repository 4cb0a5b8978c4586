"""Pipelined minibatches for the reference's worker loop (file_proc.py:380-454, 1197-1243).

A WarpDemuX worker alternates "fill minibatch k+1" (pod5 reads -> one (1000, sig_preload_size) float32 array,
file_proc.py:244-260) with "process minibatch k".  `MinibatchPipeline` gives that loop two things the plain
`sig_proc.demux_batch` call cannot:

* page-locked minibatch buffers (`pinned_empty`) the worker fills in place of ``np.full(...)`` -- the GPU reads them
  by DMA at the bus rate instead of through the runtime's pageable staging;
* `submit(slot, ...)` / `wait(slot)` (C ABI: wdx_demux_submit / wdx_demux_wait): two minibatches in flight on two
  streams of one context, so the copy-in of k+1 overlaps the kernels and the copy-out of k while the worker's own
  thread fills the next buffer.

Results are bit-identical to `demux_batch` (same kernels).  INTEGRATION.md shows the four-line change in
``file_proc``'s loop.
"""
from __future__ import annotations

import ctypes as C
import weakref
from typing import Optional

import numpy as np

from . import _lib
from .sig_proc import DemuxBatch, SegParams


def pinned_empty(shape, dtype=np.float32, device: Optional[int] = None) -> np.ndarray:
    """Uninitialised NumPy array in page-locked host memory (wdx_host_alloc); freed with the array.  ``device``: the
    GPU whose context will read it (wdx_host_alloc_on: no stray HIP context on device 0 in a multi-GPU worker)."""
    dtype = np.dtype(dtype)
    shape = (int(shape),) if np.isscalar(shape) else tuple(int(v) for v in shape)
    nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
    L = _lib.load()
    p = C.c_void_p()
    if device is None:
        _lib.check(L.wdx_host_alloc(C.c_size_t(nbytes), C.byref(p)))
    else:
        _lib.check(L.wdx_host_alloc_on(int(device), C.c_size_t(nbytes), C.byref(p)))
    buf = (C.c_char * max(nbytes, 1)).from_address(p.value)
    arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape, dtype=np.int64))).reshape(shape)
    weakref.finalize(buf, L.wdx_host_free, C.c_void_p(p.value))   # the ctypes block lives as long as any view of it
    return arr


def pinned_full(shape, fill_value, dtype=np.float32, device: Optional[int] = None) -> np.ndarray:
    """``np.full`` in page-locked memory: the drop-in for file_proc.py:244 (``np.full((n, m), np.nan, float32)``)."""
    a = pinned_empty(shape, dtype, device)
    a.fill(fill_value)
    return a


MAX_SLOTS = 8   # WDX_MAX_SLOTS (include/wdx.h)


def register_host(arr: np.ndarray):
    """Page-lock memory the caller owns (wdx_host_register) -- e.g. a multiprocessing.shared_memory block that producer
    processes fill while ONE feeder process owns the context.  Returns a finalizer-like callable that unregisters."""
    L = _lib.load()
    p = C.c_void_p(arr.ctypes.data)
    _lib.check(L.wdx_host_register(p, C.c_size_t(arr.nbytes)))
    return lambda: L.wdx_host_unregister(p)


class MinibatchPipeline:
    """Minibatches in flight against one resident reference set (model._X): two slots for a worker's own loop, up to
    MAX_SLOTS for a feeder process that serves many producers."""

    N_SLOTS = 2

    def __init__(self, refs, window=None, penalty=None, params: Optional[SegParams] = None, device: int = 0,
                 n_slots: int = 2):
        if not 1 <= int(n_slots) <= MAX_SLOTS:
            raise ValueError(f"n_slots must be in [1, {MAX_SLOTS}]")
        self.N_SLOTS = int(n_slots)
        refs = np.ascontiguousarray(refs, dtype=np.float64)
        if refs.ndim != 2:
            raise ValueError("refs must be (nY, L)")
        self.params = params or SegParams(barcode_num_events=int(refs.shape[1]))
        if self.params.barcode_num_events != refs.shape[1]:
            raise ValueError("barcode_num_events must equal the reference length")
        self.nY, self.K = (int(v) for v in refs.shape)
        self.L = _lib.load()
        self.ctx = _lib.Context(device)
        _lib.check(self.L.wdx_set_refs(self.ctx.handle, _lib.ptr(refs), self.nY, self.K,
                                       int(window) if window else 0, float(penalty) if penalty else 0.0))
        self._pc = self.params.to_c()
        self._held = [None] * self.N_SLOTS     # the submitted arrays must outlive the copy-in

    def submit(self, slot: int, signals, adapter_start, adapter_end, success=None, want_dist=True, want_fpt=False):
        """Enqueue one minibatch on `slot` (0 or 1) and return.  `signals` must not be modified before `wait(slot)`."""
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
        if not 0 <= int(slot) < self.N_SLOTS:
            raise ValueError(f"slot must be in [0, {self.N_SLOTS})")
        _lib.check(self.L.wdx_demux_submit(self.ctx.handle, int(slot), _lib.ptr(sig), n, stride, _lib.ptr(a_s),
                                           _lib.ptr(a_e), _lib.ptr(ok), C.byref(self._pc), self.nY, int(want_fpt),
                                           int(want_dist)))
        self._held[slot] = (sig, a_s, a_e, ok, n, bool(want_dist), bool(want_fpt))

    def wait(self, slot: int) -> DemuxBatch:
        held = self._held[slot] if 0 <= int(slot) < self.N_SLOTS else None
        if held is None:
            raise ValueError(f"nothing was submitted on slot {slot}")
        n, want_dist, want_fpt = held[4:]
        dist = np.empty((n, self.nY), dtype=np.float32) if want_dist else None
        fpt = np.empty((n, self.K), dtype=np.float64) if want_fpt else None
        call = np.empty(n, dtype=np.int32)
        status = np.empty(n, dtype=np.int32)
        rc = self.L.wdx_demux_wait(self.ctx.handle, int(slot), _lib.ptr(fpt), _lib.ptr(dist), _lib.ptr(call),
                                   _lib.ptr(status))
        # WDX_ERR_INVALID (an argument error, or another thread already waiting on this slot) leaves the minibatch IN
        # FLIGHT in the slot (wdx.h): the copy-in may still be reading the arrays, so they stay referenced and the
        # caller can wait again.  Success and a HIP error both free the slot on the C side.
        if rc != _lib.WDX_ERR_INVALID:
            self._held[slot] = None
        _lib.check(rc)
        return DemuxBatch(status, call, dist, fpt)

    def run(self, minibatches):
        """Drive an iterable of (signals, adapter_start, adapter_end[, success]) through both slots; yields one
        DemuxBatch per minibatch, in order.  The iterable is advanced (= the caller's fill runs) while the previous
        minibatch is in flight.

        Order per minibatch k: wait(k - 2), THEN next(iterable), then submit(k) -- so a generator that refills two
        rotating page-locked buffers never writes into a buffer whose submit has not been waited
        for (submit()'s contract, wdx.h: inputs stay untouched until the matching wait); minibatch k - 1 is still in
        flight while the generator fills buffer k."""
        it = iter(minibatches)
        pending = []
        k = 0
        while True:
            slot = k % self.N_SLOTS
            if len(pending) == self.N_SLOTS:   # the slot (and the caller's buffer) about to be reused
                yield self.wait(pending.pop(0))
            try:
                mb = next(it)
            except StopIteration:
                break
            self.submit(slot, *mb)
            pending.append(slot)
            k += 1
        for slot in pending:
            yield self.wait(slot)

    def close(self):
        self.ctx.close()

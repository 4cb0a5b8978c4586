"""Many worker processes, ONE GPU-facing process: the drop-in for the reference's `-j 8..16` forked workers
(file_proc.py:1197-1243: a ProcessPoolExecutor whose workers each run fingerprint + model on their minibatches,
file_proc.py:380-454).

Sixteen processes that each drive the GPU through their own context run at 40 % of the rate of four (the device
time-slices the processes' queues).  Here the PARENT creates a `Feeder` before it forks its workers: a ring of
minibatch slots in shared memory plus one forked process that owns the engine context, page-locks the ring and keeps
up to eight minibatches in flight (`wdx_feeder_serve`).  A worker calls `feeder.demux_batch(signals, adapter_start,
adapter_end)` -- the arguments and the result of `sig_proc.demux_batch`, bit for bit -- which copies the minibatch into
a free slot and sleeps until the results are there (`wdx_feeder_demux`: no context, no HIP call in the worker).

    feeder = Feeder(model._X, window, penalty, params, max_reads=1000, stride=10000)   # parent, before the fork
    with ProcessPoolExecutor(P, mp_context=multiprocessing.get_context("fork")) as pool:   # workers inherit `feeder`
        ... in a worker:  res = feeder.demux_batch(minibatch, adapter_start, adapter_end)
    feeder.close()
"""
from __future__ import annotations

import ctypes as C
import multiprocessing as mp
import os
from multiprocessing import shared_memory
from typing import Optional

import numpy as np

from . import _lib
from .sig_proc import DemuxBatch, SegParams

MAX_SLOTS = 32      # ring slots (WDX_FEEDER_MAX_RING_SLOTS); the feeder keeps at most 8 of them in flight on the device


def _serve(shm_name: str, refs, window, penalty, pc_bytes: bytes, device: int, ready):
    """The GPU-facing process (forked from a parent that never touched the GPU)."""
    shm = shared_memory.SharedMemory(name=shm_name)
    rc = 1
    try:
        L = _lib.load()
        ctx = _lib.Context(device)
        _lib.check(L.wdx_set_refs(ctx.handle, _lib.ptr(refs), refs.shape[0], refs.shape[1], int(window) if window else 0,
                                  float(penalty) if penalty else 0.0))
        pc = _lib.SegParamsC.from_buffer_copy(pc_bytes)
        base = C.addressof(C.c_char.from_buffer(shm.buf))
        # (wdx_feeder_serve announces itself in the ring -- server_pid -- once the ring is page-locked; the parent polls
        # wdx_feeder_alive after this event)
        ready.set()
        _lib.check(L.wdx_feeder_serve(ctx.handle, C.c_void_p(base), C.byref(pc)))
        ctx.close()
        rc = 0
    except BaseException as e:  # noqa: BLE001  (reported through the exit code and stderr; the ring is stopped below)
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
    inherited object.  `max_reads` x `stride` = the largest minibatch a slot holds (file_proc's 1000 x sig_preload_size)."""

    def __init__(self, refs, window=None, penalty=None, params: Optional[SegParams] = None, max_reads: int = 1000,
                 stride: int = 10000, n_slots: int = 16, device: int = 0, start_timeout: float = 120.0):
        refs = np.ascontiguousarray(refs, dtype=np.float64)
        if refs.ndim != 2:
            raise ValueError("refs must be (nY, L)")
        if not 1 <= int(n_slots) <= MAX_SLOTS:
            raise ValueError(f"n_slots must be in [1, {MAX_SLOTS}]")
        self.params = params or SegParams(barcode_num_events=int(refs.shape[1]))
        if self.params.barcode_num_events != refs.shape[1]:
            raise ValueError("barcode_num_events must equal the reference length")
        self.nY, self.K = (int(v) for v in refs.shape)
        self.max_reads, self.stride, self.n_slots = int(max_reads), int(stride), int(n_slots)
        self.L = _lib.load()
        nbytes = int(self.L.wdx_feeder_ring_bytes(self.n_slots, self.max_reads, self.stride, self.nY))
        if nbytes == 0:
            raise ValueError("bad ring geometry")
        self._shm = shared_memory.SharedMemory(create=True, size=nbytes)
        self._owner = os.getpid()
        self._base = C.addressof(C.c_char.from_buffer(self._shm.buf))
        _lib.check(self.L.wdx_feeder_ring_init(C.c_void_p(self._base), nbytes, self.n_slots, self.max_reads, self.stride, self.nY))
        ctx = mp.get_context("fork")
        ready = ctx.Event()
        pc = self.params.to_c()
        self._proc = ctx.Process(target=_serve, args=(self._shm.name, refs, window, penalty, bytes(pc), int(device), ready),
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

    def demux_batch(self, signals, adapter_start, adapter_end, success=None, want_dist: bool = True) -> DemuxBatch:
        """One minibatch: status, nearest-reference call and (optionally) the distance rows -- `sig_proc.demux_batch`'s
        result, bit for bit.  Callable from any process that inherited this object; blocks until the results are there."""
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
        dist = np.empty((n, self.nY), dtype=np.float32) if want_dist else None
        call = np.empty(n, dtype=np.int32)
        status = np.empty(n, dtype=np.int32)
        _lib.check(self.L.wdx_feeder_demux(C.c_void_p(self._base), _lib.ptr(sig), n, stride, _lib.ptr(a_s), _lib.ptr(a_e),
                                           _lib.ptr(ok), self.nY, _lib.ptr(dist), _lib.ptr(call), _lib.ptr(status)))
        return DemuxBatch(status, call, dist, None)

    def alive(self) -> bool:
        """True while the feeder process serves the ring."""
        return self._base is not None and self.L.wdx_feeder_alive(C.c_void_p(self._base)) == 1

    def served(self) -> int:
        v = C.c_int64(0)
        _lib.check(self.L.wdx_feeder_served(C.c_void_p(self._base), C.byref(v)))
        return int(v.value)

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

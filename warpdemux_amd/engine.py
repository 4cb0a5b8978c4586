"""Device-resident fused pipeline: raw adapter rows -> fingerprint -> DTW -> barcode call.

PyTorch is used only as plumbing (HBM allocations, the current HIP stream); all arithmetic is in
libwdx_hip.so through the ``*_dev`` entry points of include/wdx.h.  This is the path bench.py
times and the one a long-running WarpDemuX worker would keep open per GPU.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _lib, synth
from .sig_proc import SegParams


def _torch():
    import torch

    return torch


def _dp(t):
    """device pointer of a torch tensor (None -> NULL)"""
    return None if t is None else C.c_void_p(t.data_ptr())


@dataclass
class DemuxResult:
    dist: "object"      # torch float32 (n, nY)
    call: "object"      # torch int32 (n,)  argmin column, -1 for failed reads
    status: "object"    # torch int32 (n,)  WDX_READ_*
    counts: "object"    # torch int64 (nY+1,) accumulated call histogram (slot nY = failed)
    fpt: Optional["object"] = None   # torch float64 (n, K) when requested


class DemuxEngine:
    """One engine per process per GPU.  ``refs``: (nY, L) float64 reference fingerprints
    (model._X in the reference, models/dtw_base.py:20)."""

    def __init__(self, refs: np.ndarray, window: Optional[int] = 15, penalty: Optional[float] = 0.1,
                 params: Optional[SegParams] = None, device: int = 0):
        torch = _torch()
        if not torch.cuda.is_available():
            raise _lib.WdxError("DemuxEngine needs a visible MI355X (torch.cuda.is_available() is False)")
        self.torch = torch
        self.device = int(device)
        self.tdev = torch.device("cuda", self.device)
        self.ctx = _lib.Context(self.device)
        self.L = _lib.load()
        self.params = params or SegParams(barcode_num_events=int(np.asarray(refs).shape[1]))
        self.set_refs(refs, window, penalty)
        self._work = None
        self._synth_tables = {}

    # -- reference set ---------------------------------------------------------------------------
    def set_refs(self, refs, window, penalty):
        refs = np.ascontiguousarray(refs, dtype=np.float64)
        if refs.ndim != 2:
            raise ValueError("refs must be (nY, L)")
        self.nY, self.K = refs.shape
        if self.params.barcode_num_events != self.K:
            raise ValueError(
                f"barcode_num_events ({self.params.barcode_num_events}) must equal the reference length ({self.K})")
        _lib.check(self.L.wdx_set_refs(self.ctx.handle, _lib.ptr(refs), self.nY, self.K,
                                       int(window) if window else 0, float(penalty) if penalty else 0.0))

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.tdev).cuda_stream)

    # -- fused path --------------------------------------------------------------------------------
    def demux(self, sig, a_start, a_end, *, offsets=None, stride=0, max_len: int, ok=None,
              counts=None, want_fpt=False, out: Optional[DemuxResult] = None) -> DemuxResult:
        """sig: float32 device tensor, packed (with int64 ``offsets`` of n+1) or a (n, stride)
        minibatch; a_start/a_end: int32 device tensors.  Enqueues on the current stream."""
        torch = self.torch
        n = int(a_start.shape[0])
        if out is None:
            out = DemuxResult(
                dist=torch.empty((n, self.nY), dtype=torch.float32, device=self.tdev),
                call=torch.empty(n, dtype=torch.int32, device=self.tdev),
                status=torch.empty(n, dtype=torch.int32, device=self.tdev),
                counts=counts if counts is not None else torch.zeros(self.nY + 1, dtype=torch.int64, device=self.tdev),
                fpt=torch.empty((n, self.K), dtype=torch.float64, device=self.tdev) if want_fpt else None,
            )
        need = int(self.L.wdx_demux_workspace_bytes(n, self.K))
        if self._work is None or self._work.numel() < need:
            self._work = torch.empty(need, dtype=torch.uint8, device=self.tdev)
        pc = self.params.to_c()
        _lib.check(self.L.wdx_demux_dev(
            self.ctx.handle, _dp(sig), _dp(offsets), None, int(stride), int(max_len), n,
            _dp(a_start), _dp(a_end), _dp(ok), C.byref(pc), _dp(out.fpt), None, None,
            _dp(out.status), _dp(out.dist), _dp(out.call), _dp(out.counts), _dp(self._work),
            self._stream()))
        return out

    def fingerprint(self, sig, a_start, a_end, *, offsets=None, stride=0, max_len: int, ok=None, want_stats: bool = True):
        """Fingerprint stage only -> (fpt f64 (n,K), dwell i64 (n,K), stats f64 (n,6) or None, status i32).  Without the six
        statistics a large RNA004 batch takes the split main kernel (tile kernel + tail kernel), like `demux` does."""
        torch = self.torch
        n = int(a_start.shape[0])
        fpt = torch.empty((n, self.K), dtype=torch.float64, device=self.tdev)
        dwell = torch.empty((n, self.K), dtype=torch.int64, device=self.tdev)
        stats = torch.empty((n, 6), dtype=torch.float64, device=self.tdev) if want_stats else None
        status = torch.empty(n, dtype=torch.int32, device=self.tdev)
        pc = self.params.to_c()
        _lib.check(self.L.wdx_fingerprint_dev(
            self.ctx.handle, _dp(sig), _dp(offsets), None, int(stride), int(max_len), n, _dp(a_start),
            _dp(a_end), _dp(ok), C.byref(pc), _dp(fpt), _dp(dwell), _dp(stats), _dp(status),
            self._stream()))
        return fpt, dwell, stats, status

    def fingerprint_refine(self, sig, a_start, a_end, refine, *, offsets=None, stride=0, max_len: int, ok=None):
        """Consensus-refinement branch on device-resident reads (wdx_fingerprint_refine_dev) ->
        (fpt f64 (n,K), dwell i64 (n,K), stats f64 (n,6), refine_idx i32 (n,3), status i32), K = refine.barcode_keep_events."""
        torch = self.torch
        n = int(a_start.shape[0])
        if offsets is None and not stride:
            if sig.dim() != 2:
                raise ValueError("packed reads need `offsets`, a minibatch needs 2-D `sig` or `stride`")
            stride = int(sig.shape[1])     # (stride 0 would make every row alias row 0)
        K = int(refine.barcode_keep_events)
        fpt = torch.empty((n, K), dtype=torch.float64, device=self.tdev)
        dwell = torch.empty((n, K), dtype=torch.int64, device=self.tdev)
        stats = torch.empty((n, 6), dtype=torch.float64, device=self.tdev)
        idx = torch.empty((n, 3), dtype=torch.int32, device=self.tdev)
        status = torch.empty(n, dtype=torch.int32, device=self.tdev)
        pc = self.params.to_c()
        rc = refine.to_c()
        _lib.check(self.L.wdx_fingerprint_refine_dev(
            self.ctx.handle, _dp(sig), _dp(offsets), None, int(stride), int(max_len), n, _dp(a_start), _dp(a_end), _dp(ok),
            C.byref(pc), C.byref(rc), _dp(fpt), _dp(dwell), _dp(stats), _dp(idx), _dp(status), self._stream()))
        return fpt, dwell, stats, idx, status

    def dtw(self, X, want_argmin=True, out=None):
        """Device DTW of (n, L) float64 rows against the resident refs.  ``out=(dist, argmin)`` reuses the
        caller's device tensors (float32 (n, nY), int32 (n,) or None) instead of allocating."""
        torch = self.torch
        n = int(X.shape[0])
        if out is not None:
            dist, am = out
            if tuple(dist.shape) != (n, self.nY) or dist.dtype != torch.float32 or not dist.is_contiguous():
                raise ValueError("out[0] must be a contiguous float32 (n, nY) device tensor")
            if am is not None and (tuple(am.shape) != (n,) or am.dtype != torch.int32):
                raise ValueError("out[1] must be an int32 (n,) device tensor")
        else:
            dist = torch.empty((n, self.nY), dtype=torch.float32, device=self.tdev)
            am = torch.empty(n, dtype=torch.int32, device=self.tdev) if want_argmin else None
        _lib.check(self.L.wdx_dtw_matrix_dev(self.ctx.handle, _dp(X), n, _dp(dist), _dp(am), self._stream()))
        return dist, am

    # -- classifier tail -----------------------------------------------------------
    def set_svm(self, model):
        """``model``: a warpdemux_amd.models.DTW_SVM whose ``_X`` is the resident reference set."""
        if model._X.shape != (self.nY, self.K):
            raise ValueError("the SVM's training set must be the engine's reference set")
        self._svm_model = model   # keeps the host arrays alive during the upload
        m = model.to_c()
        self.ctx._svm_owner = None
        _lib.check(self.L.wdx_svm_set_model(self.ctx.handle, C.byref(m)))
        self.ctx._svm_owner = model
        self.n_classes = model.n_classes

    def svm_predict(self, dist):
        """(prob f64 (n,k), pred i32 (n,), conf f64 (n,)) from a device (n, nY) float32 distance matrix:
        models/dtw_svm.py:90-93 + models/utils.py:45-61."""
        torch = self.torch
        n = int(dist.shape[0])
        prob = torch.empty((n, self.n_classes), dtype=torch.float64, device=self.tdev)
        pred = torch.empty(n, dtype=torch.int32, device=self.tdev)
        conf = torch.empty(n, dtype=torch.float64, device=self.tdev)
        _lib.check(self.L.wdx_svm_predict_dev(self.ctx.handle, _dp(dist), n, _dp(prob), _dp(pred), _dp(conf),
                                              self._stream()))
        return prob, pred, conf

    def demux_svm(self, sig, a_start, a_end, *, offsets=None, stride=0, max_len: int, ok=None, want_dist=False,
                  want_fpt=False, block_rows: int = 0, out=None):
        """The shipped models' whole path, device-resident (wdx_demux_svm_dev): raw rows -> fingerprint -> DTW against
        the resident training set -> SVM tail.  Returns (prob f64 (n,k), pred i32 (n,), conf f64 (n,), status i32 (n,),
        dist f32 (n,nY) or None, fpt f64 (n,K) or None); ``out`` = such a tuple from an earlier call is reused."""
        torch = self.torch
        n = int(a_start.shape[0])
        if out is None:
            out = (torch.empty((n, self.n_classes), dtype=torch.float64, device=self.tdev),
                   torch.empty(n, dtype=torch.int32, device=self.tdev),
                   torch.empty(n, dtype=torch.float64, device=self.tdev),
                   torch.empty(n, dtype=torch.int32, device=self.tdev),
                   torch.empty((n, self.nY), dtype=torch.float32, device=self.tdev) if want_dist else None,
                   torch.empty((n, self.K), dtype=torch.float64, device=self.tdev) if want_fpt else None)
        prob, pred, conf, status, dist, fpt = out
        need = int(self.L.wdx_demux_workspace_bytes(n, self.K))
        if self._work is None or self._work.numel() < need:
            self._work = torch.empty(need, dtype=torch.uint8, device=self.tdev)
        pc = self.params.to_c()
        _lib.check(self.L.wdx_demux_svm_dev(
            self.ctx.handle, _dp(sig), _dp(offsets), None, int(stride), int(max_len), n, _dp(a_start), _dp(a_end), _dp(ok),
            C.byref(pc), _dp(fpt), _dp(status), _dp(dist), _dp(prob), _dp(pred), _dp(conf), _dp(self._work), int(block_rows),
            self._stream()))
        return out

    # -- synthetic inputs, generated in HBM ----------------------------------------------------------
    def synth_packed(self, spec: synth.SynthSpec, first_read: int, n_reads: int):
        """(sig f32[total], offsets i64[n+1], a_start i32[n], a_end i32[n], barcode i32[n]) on device,
        bit-identical to synth.generate_packed()."""
        torch = self.torch
        key = (spec.seed, spec.n_barcodes)
        if key not in self._synth_tables:
            lead, bc, dt = spec.tables()
            self._synth_tables[key] = (
                torch.from_numpy(lead).to(self.tdev), torch.from_numpy(bc).to(self.tdev),
                torch.from_numpy(dt).to(self.tdev))
        lead, bc, dt = self._synth_tables[key]
        lens = torch.empty(n_reads, dtype=torch.int64, device=self.tdev)
        _lib.check(self.L.wdx_synth_lengths_dev(self.ctx.handle, spec.seed, first_read, n_reads,
                                                spec.n_barcodes, _dp(dt), _dp(lens), self._stream()))
        off = torch.zeros(n_reads + 1, dtype=torch.int64, device=self.tdev)
        torch.cumsum(lens, 0, out=off[1:])
        total = int(off[-1].item())
        sig = torch.empty(total, dtype=torch.float32, device=self.tdev)
        barcode = torch.empty(n_reads, dtype=torch.int32, device=self.tdev)
        _lib.check(self.L.wdx_synth_fill_dev(
            self.ctx.handle, spec.seed, first_read, n_reads, spec.n_barcodes, spec.n_bc_events,
            float(spec.noise_scale), int(spec.spikes), _dp(dt), _dp(lead), _dp(bc), _dp(off), _dp(sig),
            _dp(barcode), self._stream()))
        a_start = torch.full((n_reads,), synth.PAD, dtype=torch.int32, device=self.tdev)
        a_end = (lens - synth.PAD).to(torch.int32)
        return sig, off, a_start, a_end, barcode, int(lens.max().item())

    # -- measurement ----------------------------------------------------------------------------------
    def kernel_timing(self, enable: bool):
        _lib.check(self.L.wdx_kernel_timing(self.ctx.handle, int(enable)))

    def kernel_time_reset(self):
        _lib.check(self.L.wdx_kernel_time_reset(self.ctx.handle))

    def kernel_time(self, kernel_id: int):
        ms = C.c_double(0)
        n = C.c_int64(0)
        _lib.check(self.L.wdx_kernel_time(self.ctx.handle, kernel_id, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def close(self):
        self.ctx.close()

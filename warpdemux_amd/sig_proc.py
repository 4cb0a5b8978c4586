"""MI355X batched counterpart of ``warpdemux.sig_proc.detect_results_to_fpt``.

The reference fingerprints one read per Python call (sig_proc.py:394-605) from a loop over the
minibatch (file_proc.py:418-428).  Here the whole minibatch goes to the HIP engine in one call
(`detect_results_to_fpt_batch`), and thin shims rebuild per-read `ReadResult` objects with the
reference's field names and fail-reason strings so the callers' savers see the same records.

Both branches of the reference are covered: the plain one (``segmentation.consensus_refinement = false``, the
shipped RNA004 config) and the consensus-guided barcode refinement of the tRNA models (sig_proc.py:257-378,
452-521; `RefineParams`, `fingerprint_refine_batch`).  ``refinement_optimal_cpts`` (ruptures KernelCPD, false
in every shipped config) is not offered.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Sequence

import numpy as np

from . import _lib

MAX_ADAPTER_SAMPLES = 16384   # WDX_MAX_ADAPTER_SAMPLES (include/wdx.h)

# status code -> ReadResult.fail_reason (reference strings: sig_proc.py:400-407, 440-446, 538-544,
# 554-560; file_proc.py:224).  Codes 2 and 4 carry the exception text in the reference; the only
# exception reachable there is mean/mad_normalize's ValueError("Signal contains NaN values.").
FAIL_REASONS = {
    0: "",
    1: None,  # passthrough of detect_results.fail_reason
    2: "signal normalization failed: Signal contains NaN values.",
    3: "event segmentation failed",
    4: "segment normalization failed: Signal contains NaN values.",
    5: "unknown",
    6: "consensus query outlier",
}


@dataclass
class SegParams:
    """Hot-path knobs of SigProcConfig (config/sig_proc.py:16-70); defaults = shipped
    rna004_130bps@v1.0.toml, outlier threshold = ADAPTed's core.sig_norm_outlier_thresh default."""

    padding: int = 100
    sig_norm: str = "none"
    outlier_thresh: float = 5.0
    min_obs_per_base: int = 6
    running_stat_width: int = 12
    num_events: int = 110
    accept_less_cpts: bool = False
    seg_norm: str = "mean"
    barcode_num_events: int = 25
    # evaluation of the clip bounds `med -/+ thresh*mad` (sig_proc.py:426-431), the one NumPy-version-dependent
    # step of the path: "float32" (NumPy >= 2 with a Python-float threshold), "float64" (NumPy 1.x -- the
    # reference pins 1.26.4 -- or an np.float64 threshold), "auto" = the rule of the NumPy this process runs,
    # i.e. what the reference would compute here
    clip_bounds: str = "auto"

    @classmethod
    def from_spc(cls, spc) -> "SegParams":
        """From a reference-style SigProcConfig (attribute access as in sig_proc.py:414-534)."""
        seg = spc.segmentation
        k = seg.barcode_num_events
        if getattr(seg, "consensus_refinement", False):
            if isinstance(k, (int, np.integer)):
                # sig_proc.py:455-459
                raise ValueError("barcode_num_events is an integer in consensus refinement mode, use a tuple instead")
            k = int(k[1])
        elif not isinstance(k, (int, np.integer)):
            raise ValueError("barcode_num_events must be an int outside consensus refinement mode")
        # The engine takes adapter windows of at most WDX_MAX_ADAPTER_SAMPLES = 16 384 samples (the shipped configs admit
        # max_obs_trace + 2 * padding = 10 200 / 15 200); the reference has no such limit (sig_proc.py:382-391).  A
        # configuration that admits longer windows (`--export core.max_obs_trace=...`) is refused HERE, once, instead of
        # every such read coming back "unknown" from the kernels.
        mot = getattr(getattr(spc, "core", None), "max_obs_trace", None)
        if isinstance(mot, (int, np.integer)) and int(mot) + 2 * int(spc.sig_extract.padding) > MAX_ADAPTER_SAMPLES:
            raise NotImplementedError(
                f"core.max_obs_trace = {int(mot)} with padding {int(spc.sig_extract.padding)} admits adapter windows of "
                f"{int(mot) + 2 * int(spc.sig_extract.padding)} samples; the HIP engine takes at most {MAX_ADAPTER_SAMPLES} "
                "(WDX_MAX_ADAPTER_SAMPLES)")
        return cls(
            padding=int(spc.sig_extract.padding),
            sig_norm=str(spc.sig_extract.normalization),
            outlier_thresh=(spc.core.sig_norm_outlier_thresh if isinstance(spc.core.sig_norm_outlier_thresh, np.float64)
                            else float(spc.core.sig_norm_outlier_thresh)),
            min_obs_per_base=int(seg.min_obs_per_base),
            running_stat_width=int(seg.running_stat_width),
            num_events=int(seg.num_events),
            accept_less_cpts=bool(seg.accept_less_cpts),
            seg_norm=str(seg.normalization),
            barcode_num_events=int(k),
        )

    def to_c(self) -> _lib.SegParamsC:
        for name in (self.sig_norm, self.seg_norm):
            if name not in _lib.NORM_CODES:
                msg = f"Normalization method {name} not recognized."
                raise ValueError(msg)
        if self.clip_bounds not in ("auto", "float32", "float64"):
            raise ValueError("clip_bounds must be 'auto', 'float32' or 'float64'")
        f64 = (self.clip_bounds == "float64" or
               (self.clip_bounds == "auto" and (isinstance(self.outlier_thresh, np.float64)
                                                or int(np.__version__.split(".")[0]) < 2)))
        return _lib.SegParamsC(
            self.padding, _lib.NORM_CODES[self.sig_norm], float(self.outlier_thresh), self.min_obs_per_base,
            self.running_stat_width, self.num_events, int(self.accept_less_cpts),
            _lib.NORM_CODES[self.seg_norm], self.barcode_num_events, int(f64), float(self.outlier_thresh),
        )


@dataclass
class RefineParams:
    """segmentation.consensus_* knobs of the refinement branch (config/sig_proc.py:57-66) + the consensus query
    (``warpdemux._consensus.ALL[segmentation.consensus_model]``, passed in by the caller like the reference's
    ``detect_results_to_fpt(..., consensus_query)``)."""

    query: np.ndarray = None
    subseq_norm: str = "mean"
    penalty: float = 1.5
    psi: tuple = (5, 0, 40, 0)
    ub_start: int = 18
    lb_end: int = 69
    ub_end: int = 97
    barcode_segm_events: int = 25
    barcode_keep_events: int = 25

    @classmethod
    def from_spc(cls, spc, consensus_query) -> "RefineParams":
        seg = spc.segmentation
        if getattr(seg, "refinement_optimal_cpts", False):
            raise NotImplementedError("refinement_optimal_cpts (ruptures KernelCPD) is not offered by the HIP engine")
        k = seg.barcode_num_events
        if isinstance(k, (int, np.integer)):
            raise ValueError("barcode_num_events is an integer in consensus refinement mode, use a tuple instead")
        if not isinstance(k, (tuple, list, np.ndarray)):
            raise TypeError("barcode_num_events must be a tuple, list or numpy array when using multiple values")
        q = np.ascontiguousarray(consensus_query, dtype=np.float64)
        if q.ndim != 1 or q.size == 0:
            raise ValueError("consensus refinement needs a 1-D consensus query")
        return cls(query=q, subseq_norm=str(seg.consensus_subseq_match_normalization),
                   penalty=float(seg.consensus_subseq_match_penalty),
                   psi=tuple(int(v) for v in seg.consensus_subseq_match_psi),
                   ub_start=int(seg.consensus_subseq_match_ub_start), lb_end=int(seg.consensus_subseq_match_lb_end),
                   ub_end=int(seg.consensus_subseq_match_ub_end), barcode_segm_events=int(k[0]),
                   barcode_keep_events=int(k[1]))

    def to_c(self) -> "_lib.RefineParamsC":
        if self.subseq_norm not in _lib.NORM_CODES:
            raise ValueError(f"Normalization method {self.subseq_norm} not recognized.")
        self._q = np.ascontiguousarray(self.query, dtype=np.float64)   # kept alive with the object
        return _lib.RefineParamsC(self._q.ctypes.data, int(self._q.size), _lib.NORM_CODES[self.subseq_norm],
                                  float(self.penalty), (C.c_int32 * 4)(*[int(v) for v in self.psi]), self.ub_start,
                                  self.lb_end, self.ub_end, self.barcode_segm_events, self.barcode_keep_events)


@dataclass
class DetectResults:
    """The four fields of ADAPTed's DetectResults the hot path reads (sig_proc.py:400-418)."""

    success: bool = True
    fail_reason: str = ""
    adapter_start: Optional[int] = None
    adapter_end: Optional[int] = None


@dataclass
class ReadResult:
    """Same fields as warpdemux.sig_proc.ReadResult (sig_proc.py:26-62)."""

    read_id: Optional[str] = None
    success: bool = True
    fail_reason: str = ""
    detect_results: Any = None
    barcode_fpt: Optional[np.ndarray] = None
    dwell_times: Optional[np.ndarray] = None
    adapter_dt_med: Optional[float] = None
    adapter_dt_mad: Optional[float] = None
    adapter_event_mean: Optional[float] = None
    adapter_event_std: Optional[float] = None
    adapter_event_med: Optional[float] = None
    adapter_event_mad: Optional[float] = None
    seg_cons_query_start: Optional[int] = None
    seg_cons_query_end: Optional[int] = None
    sig_barcode_start: Optional[int] = None

    def to_summary_dict(self) -> Dict[str, Any]:
        return {
            "read_id": self.read_id,
            "success": self.success,
            "fail_reason": self.fail_reason,
            "adapter_dt_med": self.adapter_dt_med,
            "adapter_dt_mad": self.adapter_dt_mad,
            "adapter_event_mean": self.adapter_event_mean,
            "adapter_event_std": self.adapter_event_std,
            "adapter_event_med": self.adapter_event_med,
            "adapter_event_mad": self.adapter_event_mad,
            "seg_cons_query_start": self.seg_cons_query_start,
            "seg_cons_query_end": self.seg_cons_query_end,
            "sig_barcode_start": self.sig_barcode_start,
        }

    def set_read_id(self, read_id: str):
        self.read_id = read_id


@dataclass
class FingerprintBatch:
    """Struct-of-arrays result of one minibatch."""

    fpt: np.ndarray      # (n, K) float64, NaN rows for failed reads
    dwell: np.ndarray    # (n, K) int64
    stats: np.ndarray    # (n, 6) float64: dt_med, dt_mad, event_mean, event_std, event_med, event_mad
    status: np.ndarray   # (n,) int32, WDX_READ_*
    refine_idx: Optional[np.ndarray] = None   # (n, 3) int32 seg_cons_query_start / _end, sig_barcode_start (refinement)

    @property
    def success(self) -> np.ndarray:
        return self.status == 0


def fingerprint_batch(signals, adapter_start, adapter_end, params: SegParams, success=None, device=None) -> FingerprintBatch:
    """Fingerprint a (n_reads, stride) float32 minibatch (file_proc.py:244-260 layout, NaN tail)."""
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
    pc = params.to_c()
    K = params.barcode_num_events
    fpt = np.empty((n, K), dtype=np.float64)
    dwell = np.empty((n, K), dtype=np.int64)
    stats = np.empty((n, 6), dtype=np.float64)
    status = np.empty(n, dtype=np.int32)
    ctx = _lib.default_context(device)
    L = _lib.load()
    _lib.check(
        L.wdx_fingerprint_batch(
            ctx.handle, _lib.ptr(sig), n, stride, _lib.ptr(a_s), _lib.ptr(a_e), _lib.ptr(ok),
            C.byref(pc), _lib.ptr(fpt), _lib.ptr(dwell), _lib.ptr(stats), _lib.ptr(status),
        )
    )
    return FingerprintBatch(fpt, dwell, stats, status)


def fingerprint_refine_batch(signals, adapter_start, adapter_end, params: SegParams, refine: RefineParams, success=None,
                             device=None) -> FingerprintBatch:
    """Consensus-refinement branch on a (n_reads, stride) float32 minibatch; K = refine.barcode_keep_events."""
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
    pc, rc = params.to_c(), refine.to_c()
    K = refine.barcode_keep_events
    fpt = np.empty((n, K), dtype=np.float64)
    dwell = np.empty((n, K), dtype=np.int64)
    stats = np.empty((n, 6), dtype=np.float64)
    idx = np.empty((n, 3), dtype=np.int32)
    status = np.empty(n, dtype=np.int32)
    ctx = _lib.default_context(device)
    _lib.check(_lib.load().wdx_fingerprint_refine_batch(
        ctx.handle, _lib.ptr(sig), n, stride, _lib.ptr(a_s), _lib.ptr(a_e), _lib.ptr(ok), C.byref(pc), C.byref(rc),
        _lib.ptr(fpt), _lib.ptr(dwell), _lib.ptr(stats), _lib.ptr(idx), _lib.ptr(status)))
    return FingerprintBatch(fpt, dwell, stats, status, idx)


@dataclass
class DemuxBatch:
    """Result of the fused host call: fingerprint -> DTW -> nearest reference."""

    status: np.ndarray            # (n,) int32 WDX_READ_*
    call: np.ndarray              # (n,) int32 argmin reference index, -1 for failed reads
    dist: Optional[np.ndarray]    # (n, nY) float32, NaN rows for failed reads
    fpt: Optional[np.ndarray]     # (n, K) float64


def set_references(refs, window=None, penalty=None, device=None):
    """Upload the reference set once (model._X) for `demux_batch`; kept resident in the context."""
    refs = np.ascontiguousarray(refs, dtype=np.float64)
    if refs.ndim != 2:
        raise ValueError("refs must be (nY, L)")
    ctx = _lib.default_context(device)
    _submit_references(ctx, refs, window, penalty)
    ctx._demux_refs = (refs, window, penalty)   # re-submitted by demux_batch if another call replaces them
    return refs.shape


def _submit_references(ctx, refs, window, penalty):
    import ctypes as C

    L = _lib.load()
    _lib.check(L.wdx_set_refs(ctx.handle, _lib.ptr(refs), refs.shape[0], refs.shape[1],
                              int(window) if window else 0, float(penalty) if penalty else 0.0))
    gen = C.c_int64(0)
    _lib.check(L.wdx_refs_generation(ctx.handle, C.byref(gen)))
    ctx._demux_refs_gen = gen.value


def _ensure_references(ctx):
    """The context holds one reference set; distance_matrix_to / a DTW_SVM model may have replaced the one
    `set_references` installed.  One counter read per call tells (wdx_refs_generation)."""
    import ctypes as C

    held = getattr(ctx, "_demux_refs", None)
    if held is None:
        return
    gen = C.c_int64(0)
    _lib.check(_lib.load().wdx_refs_generation(ctx.handle, C.byref(gen)))
    if gen.value != ctx._demux_refs_gen:
        _submit_references(ctx, *held)


def demux_batch(signals, adapter_start, adapter_end, params: SegParams, success=None, want_dist=True,
                want_fpt=False, n_refs=None, device=None) -> DemuxBatch:
    """One call per minibatch / live tick: fingerprints, distances to the resident references
    (`set_references`) and the nearest-reference call, with a single device synchronisation."""
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
    pc = params.to_c()
    K = params.barcode_num_events
    ctx = _lib.default_context(device)
    held = getattr(ctx, "_demux_refs", None)
    if held is None:
        raise _lib.WdxError("demux_batch: no reference set -- call set_references() first")
    # the distance matrix is sized from the set this context holds, never from the caller (the C ABI checks
    # it against the resident set once more)
    n_held = int(held[0].shape[0])
    if n_refs is not None and int(n_refs) != n_held:
        raise ValueError(f"n_refs={n_refs} but set_references() installed {n_held} references")
    dist = np.empty((n, n_held), dtype=np.float32) if want_dist else None
    fpt = np.empty((n, K), dtype=np.float64) if want_fpt else None
    call = np.empty(n, dtype=np.int32)
    status = np.empty(n, dtype=np.int32)
    _ensure_references(ctx)
    _lib.check(_lib.load().wdx_demux_batch(
        ctx.handle, _lib.ptr(sig), n, stride, _lib.ptr(a_s), _lib.ptr(a_e), _lib.ptr(ok), C.byref(pc),
        n_held, _lib.ptr(fpt), _lib.ptr(dist), _lib.ptr(call), _lib.ptr(status)))
    return DemuxBatch(status, call, dist, fpt)


def detect_results_to_fpt_batch(calibrated_signals, spc, detect_results: Sequence, read_ids: Optional[Sequence[str]] = None,
                                device=None, consensus_query=None) -> List[ReadResult]:
    """Batched `detect_results_to_fpt`: one ReadResult per row, identical fields to the reference's
    per-read call (sig_proc.py:590-605) plus the `barcode_fpt_wrapper` read-id (file_proc.py:216).  With
    ``spc.segmentation.consensus_refinement`` the caller passes the consensus signal like the reference does."""
    params = SegParams.from_spc(spc)
    refine = None
    if getattr(spc.segmentation, "consensus_refinement", False):
        if consensus_query is None or np.asarray(consensus_query).size == 0:
            raise ValueError("consensus_model must be specified when consensus_refinement is True")
        refine = RefineParams.from_spc(spc, consensus_query)
    n = len(detect_results)
    ok = np.array([bool(d.success) for d in detect_results], dtype=np.uint8)
    a_s = np.array([d.adapter_start if (d.success and d.adapter_start is not None) else 0 for d in detect_results], dtype=np.int32)
    a_e = np.array([d.adapter_end if (d.success and d.adapter_end is not None) else 0 for d in detect_results], dtype=np.int32)
    if refine is None:
        fb = fingerprint_batch(calibrated_signals, a_s, a_e, params, success=ok, device=device)
    else:
        fb = fingerprint_refine_batch(calibrated_signals, a_s, a_e, params, refine, success=ok, device=device)
    return read_results_from_batch(fb, detect_results, read_ids, refined=refine is not None)


def read_results_from_batch(fb: FingerprintBatch, detect_results: Sequence, read_ids: Optional[Sequence[str]] = None,
                            refined: bool = False) -> List[ReadResult]:
    """The per-read ``ReadResult`` records of one fingerprinted minibatch (sig_proc.py:590-605 + the read id of
    ``barcode_fpt_wrapper``, file_proc.py:216) -- shared by `detect_results_to_fpt_batch` and the feeder's workers."""
    n = len(detect_results)
    out = []
    for i in range(n):
        st = int(fb.status[i])
        rid = None if read_ids is None else read_ids[i]
        extra = {}
        if refined and st in (0, 6):
            q = fb.refine_idx[i]
            extra = dict(seg_cons_query_start=int(q[0]), seg_cons_query_end=int(q[1]), sig_barcode_start=int(q[2]))
        if st == 0 or st == 6:
            s = fb.stats[i]
            out.append(ReadResult(
                read_id=rid, success=st == 0, fail_reason=FAIL_REASONS[st], detect_results=detect_results[i],
                barcode_fpt=fb.fpt[i].copy() if st == 0 else np.array([]),
                dwell_times=fb.dwell[i].copy() if st == 0 else np.array([]),
                adapter_dt_med=float(s[0]), adapter_dt_mad=float(s[1]), adapter_event_mean=float(s[2]),
                adapter_event_std=float(s[3]), adapter_event_med=float(s[4]), adapter_event_mad=float(s[5]), **extra,
            ))
        elif st == 5:
            # barcode_fpt_wrapper's except-branch builds a bare record (file_proc.py:220-224)
            out.append(ReadResult(read_id=rid, success=False, fail_reason="unknown"))
        else:
            reason = detect_results[i].fail_reason if st == 1 else FAIL_REASONS[st]
            out.append(ReadResult(
                read_id=rid, success=False, fail_reason=reason, detect_results=detect_results[i],
                barcode_fpt=np.array([]), dwell_times=np.array([]),
            ))
    return out


def detect_results_to_fpt(calibrated_signal, spc, detect_results, consensus_query=np.array([])) -> ReadResult:
    """Per-read signature of the reference (sig_proc.py:394-399); a batch of one."""
    sig = np.asarray(calibrated_signal, dtype=np.float32).reshape(1, -1)
    return detect_results_to_fpt_batch(sig, spc, [detect_results], consensus_query=consensus_query)[0]

"""ctypes binding of libwdx_hip.so (C ABI: include/wdx.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C warpdemux_amd/csrc``.
There is NO CPU fallback: if the shared object is missing or no MI355X is visible, every entry
point raises.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# $WDX_LIB_PATH: development only (tools/ab.sh benches two builds of the library in one GPU session)
LIB_PATH = os.environ.get("WDX_LIB_PATH") or os.path.join(_HERE, "csrc", "libwdx_hip.so")

WDX_SUCCESS = 0
WDX_ERR_INVALID = -1
WDX_ERR_NO_DEVICE = -2
WDX_ERR_HIP = -3
WDX_ERR_UNSUPPORTED = -4
WDX_ERR_NO_REFS = -5

K_FINGERPRINT, K_DTW, K_TRANSPOSE, K_COUNT, K_SVM, K_REDUCE, K_FINGERPRINT_MAIN, K_FINGERPRINT_CLIP, K_FINGERPRINT_TAIL = 0, 1, 2, 3, 4, 5, 6, 7, 8

# wdx_ctx_set_option selectors (diagnostics; the product path leaves all of them 0)
OPT_EXACT_PATH, OPT_NO_WAVEFRONT_DTW, OPT_NO_SHORT_DTW, OPT_SVM_SCALAR, OPT_DEBUG_OCCUPANCY, OPT_FAST_PEAK_CAP = 1, 2, 3, 4, 5, 6
OPT_FAST_EXACT_SCORES = 7
OPT_FAST_MAIN_CAP = 8
OPT_FAST_CHAIN_MIN_READS = 9
OPT_EXACT_NO_PEAK_LIST = 10
OPT_MAX_LAUNCH_SLICE = 11
OPT_NO_PEAK_FILTER = 12
OPT_NO_WAVE_CLIP_LONG = 13
OPT_NO_CLIP_REUSE = 14
OPT_NO_SPLIT_TAIL = 15
OPT_DTW_UNFUSED = 16
COMM_ID_BYTES = 128
ABI_VERSION = 4

NORM_CODES = {"none": 0, "mean": 1, "median": 2}

# every symbol include/wdx.h declares (tests check the .so exports each of them)
EXPORTS = [
    "wdx_abi_version", "wdx_last_error", "wdx_device_count", "wdx_ctx_create", "wdx_ctx_destroy",
    "wdx_ctx_synchronize", "wdx_ctx_stream", "wdx_ctx_set_option", "wdx_comm_available", "wdx_comm_info", "wdx_comm_unique_id", "wdx_comm_init",
    "wdx_comm_destroy", "wdx_reduce_counts", "wdx_reduce_counts_host", "wdx_dtw_matrix", "wdx_set_refs", "wdx_refs_generation", "wdx_dtw_matrix_dev",
    "wdx_fingerprint_batch", "wdx_fingerprint_refine_batch", "wdx_fingerprint_refine_dev", "wdx_fingerprint_dev", "wdx_demux_batch", "wdx_demux_submit", "wdx_demux_wait", "wdx_demux_submit_ex", "wdx_demux_wait_ex",
    "wdx_host_alloc", "wdx_host_alloc_on", "wdx_host_free", "wdx_host_register", "wdx_host_unregister", "wdx_live_tick", "wdx_svm_set_model",
    "wdx_svm_predict_dev", "wdx_dtw_svm_predict", "wdx_demux_svm_dev", "wdx_demux_workspace_bytes", "wdx_demux_dev",
    "wdx_kernel_timing", "wdx_kernel_time", "wdx_kernel_time_reset", "wdx_synth_lengths_dev",
    "wdx_synth_fill_dev", "wdx_fingerprint_profile_dev", "wdx_calib_read_dev", "wdx_selftest_score_dev", "wdx_selftest_clip_dev",
    "wdx_feeder_ring_bytes", "wdx_feeder_ring_init", "wdx_feeder_serve", "wdx_feeder_run", "wdx_feeder_demux", "wdx_feeder_predict", "wdx_feeder_stop",
    "wdx_feeder_served", "wdx_feeder_stats", "wdx_feeder_alive", "wdx_feeder_selftest",
]


class SegParamsC(C.Structure):
    """wdx_seg_params (include/wdx.h)"""

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


class RefineParamsC(C.Structure):
    """wdx_refine_params (include/wdx.h)"""

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


class SvmModelC(C.Structure):
    """wdx_svm_model (include/wdx.h)"""

    _fields_ = [
        ("n_classes", C.c_int32),
        ("n_sv", C.c_int32),
        ("n_train", C.c_int32),
        ("pwr_dist", C.c_int32),
        ("gamma", C.c_double),
        ("n_support", C.c_void_p),
        ("support", C.c_void_p),
        ("dual_coef", C.c_void_p),
        ("rho", C.c_void_p),
        ("probA", C.c_void_p),
        ("probB", C.c_void_p),
        ("label_map", C.c_void_p),
        ("thresholds", C.c_void_p),
    ]


WANT_FPT, WANT_DIST, WANT_DWELL, WANT_STATS, WANT_SVM = 0x01, 0x02, 0x04, 0x08, 0x10   # WDX_WANT_*


class MinibatchInC(C.Structure):
    """wdx_minibatch_in (include/wdx.h)"""

    _fields_ = [
        ("sig", C.c_void_p), ("n_reads", C.c_int64), ("stride", C.c_int64), ("row_off", C.c_void_p), ("row_len", C.c_void_p),
        ("a_start", C.c_void_p), ("a_end", C.c_void_p), ("ok", C.c_void_p),
    ]


class MinibatchOutC(C.Structure):
    """wdx_minibatch_out (include/wdx.h)"""

    _fields_ = [
        ("status", C.c_void_p), ("call", C.c_void_p), ("dist", C.c_void_p), ("fpt", C.c_void_p), ("dwell", C.c_void_p),
        ("stats", C.c_void_p), ("prob", C.c_void_p), ("pred", C.c_void_p), ("conf", C.c_void_p),
    ]


class FeederGeometryC(C.Structure):
    """wdx_feeder_geometry (include/wdx.h)"""

    _fields_ = [
        ("n_slots", C.c_int32), ("n_events", C.c_int32), ("n_classes", C.c_int32), ("pad_", C.c_int32),
        ("max_reads", C.c_int64), ("max_stride", C.c_int64), ("n_refs", C.c_int64),
    ]


class FeederJobC(C.Structure):
    """wdx_feeder_job (include/wdx.h)"""

    _fields_ = [
        ("sig", C.c_void_p), ("n_reads", C.c_int64), ("stride", C.c_int64), ("a_start", C.c_void_p), ("a_end", C.c_void_p),
        ("ok", C.c_void_p), ("want", C.c_uint32), ("pad_", C.c_uint32),
        ("status", C.c_void_p), ("call", C.c_void_p), ("dist", C.c_void_p), ("fpt", C.c_void_p), ("dwell", C.c_void_p),
        ("stats", C.c_void_p), ("prob", C.c_void_p), ("pred", C.c_void_p), ("conf", C.c_void_p),
    ]


def addr(a):
    """Address of a NumPy array as an int for a c_void_p FIELD of a ctypes structure (None -> NULL)."""
    return None if a is None else a.ctypes.data


class WdxError(RuntimeError):
    pass


class WdxNoDevice(WdxError):
    """WDX_ERR_NO_DEVICE: no usable HIP device / runtime / RCCL in this process."""


_lib = None
_lock = threading.Lock()


def _preload_hip_runtime():
    """One process must hold ONE HIP runtime.  PyTorch-ROCm wheels bundle their own libamdhip64.so
    (soname libamdhip64.so.7) but request it as "libamdhip64.so", so if libwdx_hip.so pulled in
    /opt/rocm's copy first a later `import torch` would load a second runtime and lose the GPU.
    Loading torch's copy first (when torch is installed) makes both resolve to the same object;
    without torch the RUNPATH of libwdx_hip.so finds /opt/rocm.  $WDX_HIP_RUNTIME overrides."""
    cand = os.environ.get("WDX_HIP_RUNTIME")
    if not cand:
        try:
            import importlib.util

            spec = importlib.util.find_spec("torch")
            if spec and spec.origin:
                c = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
                if os.path.exists(c):
                    cand = c
        except Exception:
            cand = None
    if cand:
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """dlopen the engine; raises if it has not been built."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise WdxError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C warpdemux_amd/csrc` (there is no CPU fallback)"
            )
        _preload_hip_runtime()
        L = C.CDLL(LIB_PATH)
        vp, i32, i64, u64, f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_double
        P = C.POINTER
        L.wdx_abi_version.restype = C.c_int
        L.wdx_abi_version.argtypes = []
        # (checked before any other symbol is bound: an older library then fails with this message, not with an
        # AttributeError on the first export it lacks)
        have = L.wdx_abi_version()
        if have != ABI_VERSION:
            raise WdxError(f"{LIB_PATH}: ABI version {have}, this package binds version {ABI_VERSION} -- rebuild it "
                           "(make -C warpdemux_amd/csrc)")
        L.wdx_last_error.restype = C.c_char_p
        L.wdx_last_error.argtypes = []
        L.wdx_device_count.restype = C.c_int
        L.wdx_device_count.argtypes = []
        L.wdx_ctx_create.restype = C.c_int
        L.wdx_ctx_create.argtypes = [C.c_int, P(vp)]
        L.wdx_ctx_destroy.restype = None
        L.wdx_ctx_destroy.argtypes = [vp]
        L.wdx_ctx_synchronize.restype = C.c_int
        L.wdx_ctx_synchronize.argtypes = [vp, vp]
        L.wdx_ctx_stream.restype = C.c_int
        L.wdx_ctx_stream.argtypes = [vp, P(vp)]
        L.wdx_ctx_set_option.restype = C.c_int
        L.wdx_ctx_set_option.argtypes = [vp, i32, i64]
        L.wdx_comm_available.restype = C.c_int
        L.wdx_comm_available.argtypes = []
        L.wdx_comm_info.restype = C.c_int
        L.wdx_comm_info.argtypes = [vp, P(i32), P(i32), P(i32)]
        L.wdx_comm_unique_id.restype = C.c_int
        L.wdx_comm_unique_id.argtypes = [vp]
        L.wdx_comm_init.restype = C.c_int
        L.wdx_comm_init.argtypes = [vp, vp, i32, i32]
        L.wdx_comm_destroy.restype = C.c_int
        L.wdx_comm_destroy.argtypes = [vp]
        L.wdx_reduce_counts.restype = C.c_int
        L.wdx_reduce_counts.argtypes = [vp, vp, i32, vp]
        L.wdx_reduce_counts_host.restype = C.c_int
        L.wdx_reduce_counts_host.argtypes = [vp, vp, i32]
        L.wdx_dtw_matrix.restype = C.c_int
        L.wdx_dtw_matrix.argtypes = [vp, vp, i64, vp, i64, i64, i32, f64, vp, vp]
        L.wdx_refs_generation.restype = C.c_int
        L.wdx_refs_generation.argtypes = [vp, P(C.c_int64)]
        L.wdx_set_refs.restype = C.c_int
        L.wdx_set_refs.argtypes = [vp, vp, i64, i64, i32, f64]
        L.wdx_dtw_matrix_dev.restype = C.c_int
        L.wdx_dtw_matrix_dev.argtypes = [vp, vp, i64, vp, vp, vp]
        L.wdx_fingerprint_batch.restype = C.c_int
        L.wdx_fingerprint_batch.argtypes = [vp, vp, i64, i64, vp, vp, vp, P(SegParamsC), vp, vp, vp, vp]
        L.wdx_fingerprint_refine_batch.restype = C.c_int
        L.wdx_fingerprint_refine_batch.argtypes = [vp, vp, i64, i64, vp, vp, vp, P(SegParamsC), P(RefineParamsC), vp, vp, vp, vp, vp]
        L.wdx_fingerprint_refine_dev.restype = C.c_int
        L.wdx_fingerprint_refine_dev.argtypes = [vp, vp, vp, vp, i64, i64, i64, vp, vp, vp, P(SegParamsC), P(RefineParamsC), vp, vp, vp, vp, vp, vp]
        L.wdx_svm_set_model.restype = C.c_int
        L.wdx_svm_set_model.argtypes = [vp, P(SvmModelC)]
        L.wdx_svm_predict_dev.restype = C.c_int
        L.wdx_svm_predict_dev.argtypes = [vp, vp, i64, vp, vp, vp, vp]
        L.wdx_dtw_svm_predict.restype = C.c_int
        L.wdx_dtw_svm_predict.argtypes = [vp, vp, i64, vp, vp, vp]
        L.wdx_demux_batch.restype = C.c_int
        L.wdx_demux_batch.argtypes = [vp, vp, i64, i64, vp, vp, vp, P(SegParamsC), i64, vp, vp, vp, vp]
        L.wdx_demux_submit.restype = C.c_int
        L.wdx_demux_submit.argtypes = [vp, i32, vp, i64, i64, vp, vp, vp, P(SegParamsC), i64, i32, i32]
        L.wdx_demux_wait.restype = C.c_int
        L.wdx_demux_wait.argtypes = [vp, i32, vp, vp, vp, vp]
        L.wdx_host_alloc.restype = C.c_int
        L.wdx_host_alloc.argtypes = [C.c_size_t, P(vp)]
        L.wdx_host_free.restype = C.c_int
        L.wdx_host_free.argtypes = [vp]
        L.wdx_host_alloc_on.restype = C.c_int
        L.wdx_host_alloc_on.argtypes = [C.c_int, C.c_size_t, P(vp)]
        L.wdx_host_register.restype = C.c_int
        L.wdx_host_register.argtypes = [vp, C.c_size_t]
        L.wdx_host_unregister.restype = C.c_int
        L.wdx_host_unregister.argtypes = [vp]
        L.wdx_live_tick.restype = C.c_int
        L.wdx_live_tick.argtypes = [vp, vp, vp, i64, vp, vp, vp, P(SegParamsC), i64, i32, vp, vp, vp, vp, vp, vp, vp]
        L.wdx_fingerprint_dev.restype = C.c_int
        L.wdx_fingerprint_dev.argtypes = [vp, vp, vp, vp, i64, i64, i64, vp, vp, vp, P(SegParamsC), vp, vp, vp, vp, vp]
        L.wdx_demux_workspace_bytes.restype = i64
        L.wdx_demux_workspace_bytes.argtypes = [i64, i32]
        L.wdx_demux_dev.restype = C.c_int
        L.wdx_demux_dev.argtypes = [vp, vp, vp, vp, i64, i64, i64, vp, vp, vp, P(SegParamsC), vp, vp, vp, vp, vp, vp, vp, vp, vp]
        L.wdx_demux_svm_dev.restype = C.c_int
        L.wdx_demux_svm_dev.argtypes = [vp, vp, vp, vp, i64, i64, i64, vp, vp, vp, P(SegParamsC), vp, vp, vp, vp, vp, vp, vp, i64, vp]
        L.wdx_kernel_timing.restype = C.c_int
        L.wdx_kernel_timing.argtypes = [vp, C.c_int]
        L.wdx_kernel_time.restype = C.c_int
        L.wdx_kernel_time.argtypes = [vp, C.c_int, P(f64), P(i64)]
        L.wdx_kernel_time_reset.restype = C.c_int
        L.wdx_kernel_time_reset.argtypes = [vp]
        L.wdx_fingerprint_profile_dev.restype = C.c_int
        L.wdx_fingerprint_profile_dev.argtypes = [vp, vp, vp, i64, i64, i64, vp, vp, P(SegParamsC), vp, vp, i64, i32, i32, vp]
        L.wdx_selftest_score_dev.restype = C.c_int
        L.wdx_selftest_score_dev.argtypes = [vp, vp, vp, i64, vp, vp, vp]
        L.wdx_selftest_clip_dev.restype = C.c_int
        L.wdx_selftest_clip_dev.argtypes = [vp, vp, vp, i64, i64, vp, vp, P(SegParamsC), i32, vp, vp]
        L.wdx_calib_read_dev.restype = C.c_int
        L.wdx_calib_read_dev.argtypes = [vp, vp, i64, vp, vp]
        L.wdx_synth_lengths_dev.restype = C.c_int
        L.wdx_synth_lengths_dev.argtypes = [vp, u64, i64, i64, i32, vp, vp, vp]
        L.wdx_synth_fill_dev.restype = C.c_int
        L.wdx_synth_fill_dev.argtypes = [vp, u64, i64, i64, i32, i32, C.c_float, i32, vp, vp, vp, vp, vp, vp, vp]
        L.wdx_demux_submit_ex.restype = C.c_int
        L.wdx_demux_submit_ex.argtypes = [vp, i32, P(MinibatchInC), P(SegParamsC), i64, C.c_uint32]
        L.wdx_demux_wait_ex.restype = C.c_int
        L.wdx_demux_wait_ex.argtypes = [vp, i32, P(MinibatchOutC)]
        L.wdx_feeder_ring_bytes.restype = C.c_size_t
        L.wdx_feeder_ring_bytes.argtypes = [P(FeederGeometryC)]
        L.wdx_feeder_ring_init.restype = C.c_int
        L.wdx_feeder_ring_init.argtypes = [vp, C.c_size_t, P(FeederGeometryC), P(SegParamsC)]
        L.wdx_feeder_serve.restype = C.c_int
        L.wdx_feeder_serve.argtypes = [vp, vp]
        L.wdx_feeder_run.restype = C.c_int
        L.wdx_feeder_run.argtypes = [vp, P(FeederJobC)]
        L.wdx_feeder_demux.restype = C.c_int
        L.wdx_feeder_demux.argtypes = [vp, vp, i64, i64, vp, vp, vp, i64, vp, vp, vp]
        L.wdx_feeder_predict.restype = C.c_int
        L.wdx_feeder_predict.argtypes = [vp, vp, i64, vp, vp, vp]
        L.wdx_feeder_stop.restype = C.c_int
        L.wdx_feeder_stop.argtypes = [vp]
        L.wdx_feeder_served.restype = C.c_int
        L.wdx_feeder_served.argtypes = [vp, P(i64)]
        L.wdx_feeder_stats.restype = C.c_int
        L.wdx_feeder_stats.argtypes = [vp, P(i64), P(i64), P(i32)]
        L.wdx_feeder_alive.restype = C.c_int
        L.wdx_feeder_alive.argtypes = [vp]
        L.wdx_feeder_selftest.restype = C.c_int
        L.wdx_feeder_selftest.argtypes = [vp, i32]
        _lib = L
        return L


def check(rc: int):
    """Map a WDX_ERR_* code to the exception the reference's Python would raise."""
    if rc == WDX_SUCCESS:
        return
    msg = load().wdx_last_error().decode("utf-8", "replace")
    if rc == WDX_ERR_INVALID:
        raise ValueError(msg)
    if rc == WDX_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    if rc == WDX_ERR_NO_DEVICE:
        raise WdxNoDevice(f"[wdx {rc}] {msg}")
    raise WdxError(f"[wdx {rc}] {msg}")


def ptr(a):
    """void* of a NumPy array (host) -- None passes NULL."""
    if a is None:
        return None
    return a.ctypes.data_as(C.c_void_p)


class Context:
    """Owns one wdx_ctx (device workspaces + resident reference set).  Created lazily per process
    and per device so that fork()ed workers (file_proc.py:1197) each initialise HIP themselves."""

    def __init__(self, device: int = 0):
        L = load()
        h = C.c_void_p()
        check(L.wdx_ctx_create(int(device), C.byref(h)))
        self._h = h
        self._L = L
        self.device = int(device)
        self.pid = os.getpid()

    @property
    def handle(self):
        if self._h is None:
            raise WdxError("context destroyed")
        if self.pid != os.getpid():
            raise WdxError("this context was created in another process (fork): create one per process")
        return self._h

    def set_option(self, option: int, value: int = 1):
        """Diagnostic switch (wdx_ctx_set_option); tests and profiling tools only."""
        check(self._L.wdx_ctx_set_option(self.handle, int(option), int(value)))

    def synchronize(self, stream=None):
        check(self._L.wdx_ctx_synchronize(self.handle, stream))

    def close(self):
        if self._h is not None and self.pid == os.getpid():
            self._L.wdx_ctx_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_ctx: dict[tuple[int, int, int], Context] = {}


def default_context(device: int | None = None) -> Context:
    """Per-(process, thread, device) context.  Device defaults to $WDX_DEVICE or 0."""
    if device is None:
        device = int(os.environ.get("WDX_DEVICE", "0"))
    key = (os.getpid(), threading.get_ident(), device)
    c = _ctx.get(key)
    if c is None:
        c = Context(device)
        _ctx[key] = c
    return c

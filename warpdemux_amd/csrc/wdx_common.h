// Shared declarations of the libwdx_hip translation units (internal; the public ABI is include/wdx.h).
#pragma once
#include <utility>
#include <vector>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/wdx.h"

namespace wdx {

// Thread-local error message backing wdx_last_error().
void set_error(const char *fmt, ...);

#define WDX_HIP_TRY(expr)                                                                 \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) {                                                           \
            ::wdx::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),       \
                             __FILE__, __LINE__);                                         \
            return WDX_ERR_HIP;                                                           \
        }                                                                                 \
    } while (0)

// Diagnostic switches of one context (wdx_ctx_set_option).  All off on the product path; nothing in the
// library reads the environment.
struct Knobs {
    bool exact_path = false;    // WDX_OPT_EXACT_PATH: fingerprint every read on the exact general kernel
    bool no_wavefront = false;  // WDX_OPT_NO_WAVEFRONT_DTW
    bool no_short_dtw = false;  // WDX_OPT_NO_SHORT_DTW
    bool svm_scalar = false;    // WDX_OPT_SVM_SCALAR
    bool debug_occ = false;     // WDX_OPT_DEBUG_OCCUPANCY
    int fast_peak_cap = 0;      // WDX_OPT_FAST_PEAK_CAP (0 = built-in capacity)
    int fast_chain_min = 0;     // WDX_OPT_FAST_CHAIN_MIN_READS: batch size from which the launch chain is used (0 = 2048)
    int fast_main_cap = 0;      // WDX_OPT_FAST_MAIN_CAP: 5120 / 6144 forces the main fast instantiation (0 = by batch)
    bool fast_exact_scores = false;  // WDX_OPT_FAST_EXACT_SCORES: fast fingerprint kernel without the approximate first attempt
    bool exact_no_list = false;      // WDX_OPT_EXACT_NO_PEAK_LIST: exact kernel's suppression / top-E in position space only
    bool no_clip_reuse = false;      // WDX_OPT_NO_CLIP_REUSE: the exact kernel behind the chain recomputes its clip bounds
    bool no_wave_clip_long = false;  // WDX_OPT_NO_WAVE_CLIP_LONG: long windows' clip bounds by clip_bounds_block_kernel alone
    bool no_peak_filter = false;     // WDX_OPT_NO_PEAK_FILTER: fast kernels append every local maximum (no threshold filter)
    int64_t max_launch_slice = 0;    // WDX_OPT_MAX_LAUNCH_SLICE: upper bound of one launch slice of the fingerprint chain (0 = built-in)
    bool no_split = false;           // WDX_OPT_NO_SPLIT_TAIL: the main fast kernel in one piece (A/B, tests)
    int dtw_unfused = 0;             // WDX_OPT_DTW_UNFUSED: 0 fused cells + settle | 1 six operations only | 2 / 3 tests, diagnostics
};

// A launch over more workgroups than grid.x admits is cut into slices (block_base != 0 from the second on).  The built-in
// slice sizes are millions of workgroups; WDX_OPT_MAX_LAUNCH_SLICE lowers them for the calling thread while one
// launch_fingerprint runs, so that the tests walk the multi-slice paths on batches the oracle finishes in seconds.
int64_t launch_slice_limit(int64_t builtin);
struct LaunchSliceScope {
    explicit LaunchSliceScope(int64_t cap);
    ~LaunchSliceScope();
    int64_t saved;
};

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device, size): a launch on the live
// path is ~0.1 ms, the attribute call must not be paid on each of them.  One static LdsAttr per kernel.
struct LdsAttr {
    int set[16] = {};
    template <class K>
    int ensure(K kern, size_t lds) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = -1;
        if (dev >= 0 && __atomic_load_n(&set[dev], __ATOMIC_RELAXED) >= (int)lds) return WDX_SUCCESS;
        WDX_HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (dev >= 0) __atomic_store_n(&set[dev], (int)lds, __ATOMIC_RELAXED);
        return WDX_SUCCESS;
    }
};

// ---- DTW (wdx_dtw.hip) -----------------------------------------------------------------------
// Largest Sakoe-Chiba window handled by the register-band kernel; wider / unbanded problems with
// L > this go to the scratch-row kernel.
constexpr int kMaxRegWindow = 32;

struct DtwRefs {  // resident reference set, both layouts (see DESIGN.md "DTW data layout")
    double *pad = nullptr;   // (nY, Lpad) row-major, Lpad = L + 2*halo, series starts at +halo
    double *T = nullptr;     // (L, ldT) read-minor (series along columns)
    uint8_t *has_nan = nullptr;  // nY flags
    int64_t nY = 0, L = 0, Lpad = 0, ldT = 0;
    int halo = 0;
    int window = 0;  // effective window (1..L), 0 = not set
    double penalty = 0;
    uint64_t content_hash = 0;
};

// lanes run over the columns of AT (L, ldA) [nA series] -- or, with a_rowmajor, over the rows of an (nA, L)
// row-major array (ldA ignored; a_nan may be null: the kernel flags NaN series itself); the uniform operand is
// Bpad (nB rows).
// out[a*sA + b*sB] = (float)dtw(a, b).  d_argmin (nullable) is only legal when the lanes are the
// reads (sA == nB, sB == 1): int32[nA].
int launch_dtw(const double *AT, int64_t ldA, int64_t nA, const uint8_t *a_nan /*nullable*/,
               const double *Bpad, int64_t Lpad, int halo, int64_t nB, const uint8_t *b_nan,
               int64_t L, int window, double penalty, float *out, int64_t sA, int64_t sB,
               int32_t *d_argmin, void *d_scratch, int64_t scratch_bytes, hipStream_t stream,
               const Knobs &knobs, bool a_rowmajor = false);
int64_t dtw_scratch_bytes(int64_t L, int window);
// The shipped models' DTW (25 points, window 15) over references in support-vector order with the SVM decision sums in
// its epilogue: P[slot][q][read] (wdx_dtw.hip: dtw_short_svm_kernel); gather of the resident references into that order.
int launch_dtw_svm_partial(const double *X, int64_t nA, const double *Ypad_sv, int64_t Lpad, int halo, const uint8_t *y_nan_sv,
                           int64_t L, int window, double penalty, const double *coefT, const int32_t *chunk_ref0,
                           const int32_t *chunk_slot, int n_chunks, int km1, int pwr, float ngamma, double *P,
                           hipStream_t stream, int unfused);
int launch_gather_rows(const double *src, const uint8_t *sflag, const int32_t *d_idx, int64_t n, int64_t ld, double *dst,
                       uint8_t *dflag, hipStream_t stream);
// anti-diagonal wavefront kernel for small problems (latency path)
bool dtw_wavefront_eligible(int64_t nX, int64_t nY, int64_t L, int window, const Knobs &knobs);
int launch_dtw_wavefront(const double *X, int64_t nX, const double *Ypad, int64_t Lpad, int halo,
                         int64_t nY, int64_t L, int window, double penalty, float *out,
                         hipStream_t stream);

// (n, L) row-major -> (L, ld) read-minor, plus per-series NaN flag (nullable)
int launch_transpose(const double *src, int64_t n, int64_t L, double *dstT, int64_t ld,
                     uint8_t *has_nan, hipStream_t stream);
int launch_nan_flags_T(const double *T, int64_t ld, int64_t n, int64_t L, uint8_t *flags,
                       hipStream_t stream);
// (n, L) row-major -> (n, Lpad) with `halo` zero entries either side, plus NaN flags
int launch_pad_rows(const double *src, int64_t n, int64_t L, double *dst, int64_t Lpad, int halo,
                    uint8_t *has_nan, hipStream_t stream);
// per-row argmin of a float32 (n, m) matrix with np.argmin semantics
int launch_argmin(const float *D, int64_t n, int64_t m, int32_t *out, hipStream_t stream);
// call[r] = status[r] ? -1 : call[r]; counts[call or m] += 1
int launch_count_calls(int32_t *call, const int32_t *status /*nullable*/, int64_t n, int64_t m,
                       int64_t *counts /*nullable*/, hipStream_t stream);

struct RefineDev;  // wdx_fingerprint.hip: device-side view of wdx_refine_params
int fill_refine_dev(const wdx_refine_params &rp, const double *d_query, int32_t *d_idx, struct RefineDev **out);
void free_refine_dev(struct RefineDev *rf);

// ---- fingerprint (wdx_fingerprint.hip) ---------------------------------------------------------
// Optional event pair recorded around the launches of the MAIN fast kernel only (WDX_K_FINGERPRINT_MAIN): the
// fingerprint stage is a chain of launches, and the roofline is quoted on its dominant kernel.
struct MainEvents {
    hipEvent_t first = nullptr, second = nullptr;
    bool recorded = false;
    // a third pair around clip_bounds_kernel (WDX_K_FINGERPRINT_CLIP): the launch ahead of the main kernel
    hipEvent_t c_first = nullptr, c_second = nullptr;
    bool c_recorded = false;
    // SPLIT main kernel: first / second bracket the whole sequence of (tile kernel, tail kernel) launch pairs; one more
    // pair per slice around the tail kernel alone (WDX_K_FINGERPRINT_TAIL), taken from the context's pool through `take`
    std::vector<std::pair<hipEvent_t, hipEvent_t>> tail;
    std::pair<hipEvent_t, hipEvent_t> (*take)(void *) = nullptr;
    void *take_arg = nullptr;
};
int launch_fingerprint(const float *d_sig, const int64_t *d_row_off, const int32_t *d_row_len,
                       int64_t stride, int64_t max_len, int64_t n_reads, const int32_t *d_a_start,
                       const int32_t *d_a_end, const uint8_t *d_ok, const wdx_seg_params &p,
                       double *d_fpt, int64_t *d_dwell, double *d_stats, int32_t *d_status,
                       hipStream_t stream, void *d_ws /* fingerprint_workspace_bytes(n) or null */,
                       const Knobs &knobs, int64_t *n_launches = nullptr, long long *d_prof = nullptr,
                       int64_t prof_reads = 0, int stop_phase = 0, const struct RefineDev *rf = nullptr,
                       MainEvents *main_ev = nullptr,
                       double *d_big = nullptr /* fingerprint_big_bytes(max_len) bytes, or null */);
int64_t fingerprint_workspace_bytes(int64_t n_reads);
// device bytes of the fast kernels' hand-over records for the refinement branch (RefineDev::ws), zero-initialised
int64_t fingerprint_refine_ws_bytes(int64_t n_reads);
void set_refine_ws(struct RefineDev *rf, void *d_ws);
// device bytes the exact kernel needs for the score curves of windows beyond its LDS capacity (0 when max_len fits)
int64_t fingerprint_big_bytes(int64_t max_len);
int launch_clip_bounds_selftest(const float *d_sig, const int64_t *d_row_off, int64_t stride, int64_t n_reads,
                                const int32_t *d_a_start, const int32_t *d_a_end, const wdx_seg_params &p, int cap,
                                void *d_rec, hipStream_t stream);
int launch_score_selftest(const double *dm, const double *vs, int64_t n, double *fast, double *ref, hipStream_t stream);

// ---- synthetic generator (wdx_synth.hip) -------------------------------------------------------
int launch_synth_lengths(uint64_t seed, int64_t first_read, int64_t n, int32_t n_barcodes,
                         const int32_t *dwell_table, int64_t *d_len, hipStream_t stream);
int launch_synth_fill(uint64_t seed, int64_t first_read, int64_t n, int32_t n_barcodes,
                      int32_t n_bc_events, float noise_scale, int32_t spikes,
                      const int32_t *dwell_table, const float *lead, const float *bc,
                      const int64_t *off, float *sig, int32_t *barcode, hipStream_t stream);

// ---- SVM tail (wdx_svm.hip) -----------------------------------------------------------------------
struct SvmDev {  // device-resident SVC(kernel="precomputed", probability=True) + label map / thresholds
    const int32_t *n_support, *support, *start, *label_map;
    const double *dual_coef, *rho, *probA, *probB, *thresholds;  // label_map / thresholds nullable
    int k, n_sv, n_train, pwr;
    float ngamma;  // -gamma rounded to float32 (NumPy: python float * float32 array -> float32)
};
int launch_svm_predict(const SvmDev &M, const float *d_dist, int64_t n, double *d_prob, int32_t *d_pred,
                       double *d_conf, hipStream_t stream, const Knobs &knobs);

int launch_svm_finish(const SvmDev &M, const double *d_P, int halves, int64_t n, double *d_prob, int32_t *d_pred,
                      double *d_conf, hipStream_t stream);
int launch_svm_mask_failed(const int32_t *d_status, int64_t n, int k, double *d_prob, int32_t *d_pred, double *d_conf,
                           hipStream_t stream);

// row r of a page-locked (n, stride) host minibatch, samples [st[r], st[r] + len[r]) -> dst + off[r] (device), read over
// the bus by a copy kernel
int launch_pack_windows(const float *src_dev, int64_t stride, int64_t n_reads, const int64_t *d_off, const int32_t *d_st,
                        const int32_t *d_len, float *dst, hipStream_t stream);
int launch_calib_read(const float *p, int64_t n, float *out, hipStream_t stream);

}  // namespace wdx

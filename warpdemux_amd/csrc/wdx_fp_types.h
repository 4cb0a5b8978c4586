// Device-side argument blocks of the fingerprint kernels, shared by wdx_fingerprint.hip (the exact general kernel, the
// fast kernels' launch chain) and wdx_clip.hip (clip_bounds_kernel).  Internal.
#pragma once
#include "wdx_common.h"

namespace wdx {

constexpr int kHBits = 11;       // radix digit of the float32 medians (2048-bin histogram)
constexpr int kHB = 1 << kHBits; // histogram bins

// consensus-guided refinement (sig_proc.py:257-378, 452-521); query == nullptr: plain branch
struct RefineDev {
    const double *query;   // consensus signal, DEVICE pointer
    int nq, norm;          // its length; consensus_subseq_match_normalization (WDX_NORM_*)
    double pen;            // consensus_subseq_match_penalty (un-squared)
    int psi1b, psi2b;      // relaxations at the beginning of the query / of the series
    int ub_start, lb_end, ub_end;
    int E2;                // barcode_num_events[0]
    int32_t *idx;          // (n_reads, 3) seg_cons_query_start, seg_cons_query_end, sig_barcode_start; nullable
    unsigned char *ws;     // n_reads RefineRec (device): the fast kernels' hand-over to fingerprint_refine_tail_kernel;
                           // null -> the refinement branch runs on the exact kernel only
};
// What a fast kernel leaves behind for a read of the refinement branch: the adapter's segmentation (bit-identical to
// the exact kernel's) and the clip bounds, so that the tail kernel can re-create the clipped samples of the barcode.
struct RefineRec {
    int32_t state;         // 0 untouched, 1 segmented by a fast kernel, 3 matched (fingerprint_refine_match_kernel),
                           // 4 reported by the match kernel, 2 handed on to the exact kernel (tail beyond kTailCap)
    int32_t n;             // adapter window length
    float lo, hi;          // clip bounds
    int32_t cpts[132];     // nseg + 1 boundaries (nseg <= 128)
    double ev[128];        // nseg event means
    double m[8];           // RefineMatch of the match kernel (bit copy)
};
static_assert(sizeof(RefineRec) == 1632, "fingerprint_refine_ws_bytes");
// what the subsequence match leaves for the barcode's segmentation (RefineRec::m, bit copy)
struct RefineMatch {
    double mean, sd, ev_med, ev_mad, dt_med, dt_mad;
    int32_t qs, qe, sbs, pad_;
};
static_assert(sizeof(RefineMatch) == 64, "RefineRec::m");
constexpr int kTailCap = 2048;        // barcode tails up to this many samples are segmented by the refinement tail kernels
constexpr int kRefineMaxQuery = 96;   // LDS budget of the subsequence DP (direction words + three fronts)
constexpr int kRefineMaxSeries = 128;

struct ClipRec;
struct FpArgs {
    const float *sig;
    const int64_t *row_off;
    const int32_t *row_len;
    int64_t stride;
    int64_t n_reads;
    const int32_t *a_start;
    const int32_t *a_end;
    const uint8_t *ok;
    wdx_seg_params p;
    double *fpt;
    int64_t *dwell;
    double *stats;
    int32_t *status;
    int cap;             // LDS capacity in samples
    int64_t block_base;  // first read of this launch (grid.x * block.x must stay below 2^32)
    long long *prof;     // diagnostic build only: 32 int64 per read (cycle stamps etc.)
    int64_t prof_reads;
    int stop_phase;      // diagnostic build only: leave the fast kernel after this phase (0 = run all)
    int exact_scores;    // fast kernel: exact t-scores from the first attempt (WDX_OPT_FAST_EXACT_SCORES)
    RefineDev rf;        // rf.query != nullptr: consensus-refinement branch (exact kernel only)
    double *big_scores;  // kBigSlots x kBigCap doubles: score curves of windows beyond the LDS capacity (nullable)
    int defer_big;       // 1: a window beyond `cap` is left to fingerprint_big_kernel (no status written here)
    int no_list;         // WDX_OPT_EXACT_NO_PEAK_LIST: fp_segment in position space only (diagnostic)
    int refine_record;   // exact kernel, refinement branch: 1 = leave a RefineRec for the refinement kernels where the read
                         // allows it (no NaN in the window, configured window width) instead of refining in place
    int peak_filter;     // fast kernels on approximate keys: 1 = drop peaks below kPeakTau at the append (WDX_OPT_NO_PEAK_FILTER)
    unsigned *dbg_reasons;  // WDX_OPT_DEBUG_OCCUPANCY: 16 counters, why the fast kernels handed reads to the exact kernel (else null)
    const ClipRec *clip; // exact kernel behind the launch chain: the reads' clip records (CLIP_OK -> the two medians are not redone); nullable
    unsigned e_magic1, e_magic2;   // ceil(2^32 / E), ceil(2^32 / 2E) (launch_fingerprint): the fast kernels' window parameters without a division
};

// (int)rint((double)n / (double)den) -- the reference's `round(n / den)` (sig_proc.py:526-533; banker's rounding) -- in integer
// arithmetic: magic = ceil(2^32 / den) gives the exact quotient for n * den < 2^32 (n <= 16 384 samples, den <= 2 * 253), and the
// double quotient can only differ from the rational one's rounding within 2^-53 of a tie, which is exact or >= 1 / (2 den) away.
// n is uniform: scalar instructions (s_mul_hi_u32) instead of a float64 division expanded on the vector ALU by every lane
// (checked against the float form for every den <= 506, n < 20 000).
__device__ __forceinline__ int rdiv_half_even(unsigned n, unsigned den, unsigned magic) {
    unsigned q = __umulhi(n, magic);
    const unsigned rem2 = 2u * (n - q * den);
    q += (rem2 > den || (rem2 == den && (q & 1u))) ? 1u : 0u;
    return (int)q;
}

// Clip bounds of one read, computed ahead of the fast kernels' launch chain by clip_bounds_kernel (one wave per read,
// wdx_fingerprint_clip.inc).  flag: CLIP_NONE = not computed (failed detection, window outside the kernel's range),
// CLIP_OK = bounds valid and every partial sum of the clipped samples exactly representable, CLIP_NAN_NEG = a
// negative sample / -0.0 / infinity / NaN in the window, CLIP_INEXACT = the exactness gate fails.
struct alignas(16) ClipRec {
    float lo, hi;   // med -/+ thresh * mad (float32, clip_bounds)
    float cmax;     // largest clipped sample (-> the variance floor of the approximate score keys)
    int32_t flag;
};
enum : int32_t { CLIP_NONE = 0, CLIP_OK = 1, CLIP_NAN_NEG = 2, CLIP_INEXACT = 3 };

// med -/+ thresh*mad of the outlier clip (sig_proc.py:426-431): in float32 (NumPy >= 2 with a Python-float
// threshold) or in float64 from the double threshold, rounded to float32 once (NumPy 1.x, np.float64 threshold)
__device__ __forceinline__ void clip_bounds(const wdx_seg_params &P, float med, float mad, float &lo, float &hi) {
    if (P.clip_bounds_f64) {
        const double tm = P.outlier_thresh_f64 * (double)mad;
        lo = (float)((double)med - tm);
        hi = (float)((double)med + tm);
    } else {
        const float tm = P.outlier_thresh * mad;
        lo = med - tm;
        hi = med + tm;
    }
}

__device__ __forceinline__ unsigned f32_key(float x) {
    unsigned u = __float_as_uint(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_f32(unsigned k) {
    unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}

// A1 for a whole batch ahead of the fast kernels' launch chain (wdx_clip.hip): one ClipRec per read of A; windows of
// 256 .. cap samples (cap = 4096, 5120 or 6144: the main fast instantiation's) are taken, the others are flagged CLIP_NONE.
int launch_clip_bounds(const FpArgs &A, ClipRec *d_rec, int cap, hipStream_t stream);
int launch_route_long_windows(const FpArgs &A, int cap, unsigned *d_count, int32_t *d_list, hipStream_t stream);
// the same for the first n_entries entries of a device-side read list (windows up to 6144 samples; reads that already
// have a record are skipped)
constexpr int kClipWaveLongCap = 13312;  // the longest window of the one-wave clip kernel (208 samples per lane)
int launch_clip_bounds_list(const FpArgs &A, ClipRec *d_rec, const unsigned *d_count, const int32_t *d_list, int64_t n_entries,
                            hipStream_t stream, int cap = 6144, bool defer = false);

// sqrt and quotient of the t-score without the range scaling of the compiler's general float64
// expansions.  The iterations are exactly the ones hipcc emits for sqrt() and '/' on gfx950
// (v_rsq_f64 / v_rcp_f64 seeds + the same fma chain), so the results are the same bits; what is dropped
// is v_ldexp / v_div_scale / v_div_fixup / v_cmp_class, which are identities on the fast path's value
// range: clipped samples are positive float32 (P1), hence a non-zero variance sum lies in
// [2^-402, 2^261] (squares of float64 differences of float32 values) and |m1 - m2| in {0} U [2^-201, 2^129]
// -- far inside the unscaled domain (x >= 2^-767 for sqrt; quotient and reciprocal normal for the
// division).  A zero variance sum is selected away by the caller.
__device__ __forceinline__ double fast_sqrt_mid(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    double d = __builtin_fma(-g, g, x);
    h = __builtin_fma(h, r, h);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    return __builtin_fma(d, h, g);
}
__device__ __forceinline__ double fast_div_mid(double num, double den) {
    double r = __builtin_amdgcn_rcp(den);
    double e = __builtin_fma(-den, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-den, r, 1.0);
    r = __builtin_fma(r, e, r);
    const double q = num * r;
    const double rem = __builtin_fma(-den, q, num);
    return __builtin_fma(rem, r, q);
}

// max(x, +0.0) as one raw v_max_f64 (no canonicalising pre-op): NaN -> 0, x >= 0 -> x
__device__ __forceinline__ double max0_f64(double x) {
    double r;
    asm("v_max_f64 %0, %1, 0" : "=v"(r) : "v"(x));
    return r;
}

// mean and sum of squared deviations of the W samples from x on (_c_segmentation.pyx:124-161; fp_process_read's
// operations in its order).  WT > 0: the configured width known at compile time (loads and conversions once)
// FASTM: positive samples (the value range of fast_div_mid): the quotient without the general expansion's range scaling
template <int WT, bool FASTM = false>
__device__ __forceinline__ void window_stats(const float *x, const int W, double &m, double &v) {
    if constexpr (WT > 0) {
        double xs[WT];
#pragma unroll
        for (int k = 0; k < WT; ++k) xs[k] = (double)x[k];
        m = 0.0;
#pragma unroll
        for (int k = 0; k < WT; ++k) m += xs[k];
        if constexpr (FASTM) m = fast_div_mid(m, (double)WT);
        else m /= (double)WT;
        v = 0.0;
#pragma unroll
        for (int k = 0; k < WT; ++k) {
            const double df = xs[k] - m;
            v += df * df;
        }
    } else {
        m = 0.0;
        for (int k = 0; k < W; ++k) m += (double)x[k];
        if constexpr (FASTM) m = fast_div_mid(m, (double)W);
        else m /= (double)W;
        v = 0.0;
        for (int k = 0; k < W; ++k) {
            const double df = (double)x[k] - m;
            v += df * df;
        }
    }
}

// the subsequence match of the consensus refinement for the reads a fast kernel segmented (RefineRec::state == 1), reads
// block_base .. block_base + n of A: one wave per group of reads (wdx_refine_match.hip)
int launch_refine_match_wave(FpArgs A, int64_t n, hipStream_t stream);
// the barcode's own segmentation for the matched reads (RefineRec::state == 3), one wave per read (wdx_refine_tail.hip);
// false: the parameters are outside what it takes (the workgroup-per-read kernel in wdx_fingerprint.hip serves those)
bool refine_tail_wave_takes(const FpArgs &A);
int launch_refine_tail_wave(FpArgs A, int64_t n, unsigned *back_count, int32_t *back_list, hipStream_t stream);

}  // namespace wdx

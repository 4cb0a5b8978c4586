// Per-read adapter fingerprinting for gfx950 -- stage A of the hot path.
//
// Replaces the per-read Python loop /root/reference/warpdemux/file_proc.py:418-428 over
// sig_proc.py:394-605 `detect_results_to_fpt` (non-refinement branch) and the Cython primitives
// segmentation/_c_segmentation.pyx:41-53 (c_new_means) and :124-161 (c_windowed_t_test), plus the
// scipy.signal.find_peaks(distance=) call at sig_proc.py:183.
//
// Mapping (DESIGN.md "Fingerprint kernel"): ONE WORKGROUP PER READ.  The adapter samples are read
// from HBM exactly once (coalesced), and everything else -- the clipped float32 signal, the
// float64 t-score curve, the peak state bytes, change-points and event means -- lives in LDS until
// the K-point fingerprint is written.  All float64 arithmetic is issued un-fused and in the
// reference's operation order, so t-scores, change-points and fingerprints are bit-identical to
// the CPU path.  Phases:
//   P0 load            HBM -> LDS (float32)
//   P1 MAD clip        two exact medians by range-adaptive radix select over LDS (A1)
//   P2 t-score         window mean/variance M[q],V[q] once per window start (m2(pos) == m1(pos+W)
//                      operation for operation), tiled through LDS, float64 (A3)
//   P3 peaks           local maxima incl. plateaus; greedy minimum-distance suppression as a
//                      fixed-point iteration over the position-space state bytes (A4, App. B)
//   P4 top-E           radix select on the float64 score bits (A4)
//   P5 boundaries      ordered compaction -> change-points
//   P6 event means     sequential float64 sums per segment (A5)
//   P7 normalise+stats numpy pairwise mean/std, medians by rank counting, tail-K extract (A6)
#include "wdx_common.h"
#include "wdx_fp_types.h"
#include "wdx_wave.h"

#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

namespace wdx {

constexpr int kMaxW = 64;        // cap on running_stat_width
constexpr int kTile = 512;       // t-score tile, positions
constexpr int kMaxEvents = 253;  // cap on num_events (E + 2 boundaries <= 255)
constexpr int kSegCap = kMaxEvents + 1;
// Exact kernel: windows up to kExactLdsCap samples keep their float64 score curve in LDS (12-13 B per sample of the
// 160 KB); longer ones (up to WDX_MAX_ADAPTER_SAMPLES = kBigCap; the reference's RNA002 config allows
// max_obs_trace + 2*padding = 15 200, DEPRECATED/config_files/rna002_70bps@v0.4.4.toml:2) run the SAME code with
// the score curve in a per-workgroup HBM slot (L2-resident: 128 KB per slot) -- fingerprint_big_kernel.
constexpr int kExactLdsCap = 11200;
constexpr int kBigCap = WDX_MAX_ADAPTER_SAMPLES;
constexpr int kBigSlots = 256;  // one workgroup per CU (97 KB of LDS each)
static_assert(kBigCap % 64 == 0 && kBigCap >= 15200, "the exact path must take every window the reference accepts");

enum : unsigned char { ST_NONE = 0, ST_UNDECIDED = 1, ST_KEPT = 2, ST_DROPPED = 3, ST_SELECTED = 4 };

struct alignas(8) FpShared {
    unsigned long long red64[16];
    unsigned red_a[16], red_b[16], red_c[16];
    unsigned sel_bin, sel_k;
    unsigned long long sel_key64;
    int flag;
    int count;
    float med, mad;
    double mean, sd;
    double stat[6];
};

// ---- block primitives ----------------------------------------------------------------------------

template <int BLOCK>
__device__ __forceinline__ void block_minmax_count(unsigned kmin, unsigned kmax, unsigned cnt,
                                                   FpShared &sh, unsigned &omin, unsigned &omax,
                                                   unsigned &ocnt) {
    for (int off = 32; off > 0; off >>= 1) {
        kmin = min(kmin, (unsigned)__shfl_xor((int)kmin, off));
        kmax = max(kmax, (unsigned)__shfl_xor((int)kmax, off));
        cnt += (unsigned)__shfl_xor((int)cnt, off);
    }
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        sh.red_a[wave] = kmin;
        sh.red_b[wave] = kmax;
        sh.red_c[wave] = cnt;
    }
    __syncthreads();
    omin = 0xffffffffu;
    omax = 0;
    ocnt = 0;
#pragma unroll
    for (int k = 0; k < BLOCK / 64; ++k) {
        omin = min(omin, sh.red_a[k]);
        omax = max(omax, sh.red_b[k]);
        ocnt += sh.red_c[k];
    }
}

template <int BLOCK>
__device__ __forceinline__ void block_minmax64_count(unsigned long long kmin,
                                                     unsigned long long kmax, unsigned cnt,
                                                     FpShared &sh, unsigned long long &omin,
                                                     unsigned long long &omax, unsigned &ocnt) {
    for (int off = 32; off > 0; off >>= 1) {
        unsigned long long a = __shfl_xor(kmin, off), b = __shfl_xor(kmax, off);
        kmin = a < kmin ? a : kmin;
        kmax = b > kmax ? b : kmax;
        cnt += (unsigned)__shfl_xor((int)cnt, off);
    }
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        sh.red64[wave] = kmin;
        sh.red_c[wave] = cnt;
    }
    __syncthreads();
    omin = ~0ull;
    ocnt = 0;
#pragma unroll
    for (int k = 0; k < BLOCK / 64; ++k) {
        omin = sh.red64[k] < omin ? sh.red64[k] : omin;
        ocnt += sh.red_c[k];
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh.red64[wave] = kmax;
    __syncthreads();
    omax = 0;
#pragma unroll
    for (int k = 0; k < BLOCK / 64; ++k) omax = sh.red64[k] > omax ? sh.red64[k] : omax;
}

// After the histogram of one digit is complete: wave 0 finds the bin holding rank k.
__device__ __forceinline__ void select_scan_bins(unsigned *hist, unsigned k, FpShared &sh) {
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        unsigned c0 = hist[4 * lane], c1 = hist[4 * lane + 1], c2 = hist[4 * lane + 2],
                 c3 = hist[4 * lane + 3];
        unsigned s = c0 + c1 + c2 + c3, incl = s;
        for (int off = 1; off < 64; off <<= 1) {
            unsigned t = (unsigned)__shfl_up((int)incl, off);
            if (lane >= off) incl += t;
        }
        unsigned e = incl - s;
        if (k >= e && k < e + s) {
            unsigned bin = 4 * lane, kk = k - e;
            if (kk >= c0) { kk -= c0; bin++;
                if (kk >= c1) { kk -= c1; bin++;
                    if (kk >= c2) { kk -= c2; bin++; } } }
            sh.sel_bin = bin;
            sh.sel_k = kk;
        }
    }
}

// k-th smallest (0-based) 32-bit key among the valid items; keyfn(i, key) -> valid.
// All threads must call; 0 <= k < n_valid; kmin/kmax are the exact extremes of the valid keys.
template <int BLOCK, class KeyFn>
__device__ unsigned block_select_u32(KeyFn keyfn, int n, unsigned k, unsigned kmin, unsigned kmax,
                                     unsigned *hist, FpShared &sh) {
    const unsigned range = kmax - kmin;
    int rb = range ? 32 - __clz((int)range) : 0;
    unsigned prefix = 0;
    while (rb > 0) {
        const int b = rb < 8 ? rb : 8;
        const int shift = rb - b;
        for (int t = threadIdx.x; t < 256; t += BLOCK) hist[t] = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += BLOCK) {
            unsigned key;
            if (keyfn(i, key)) {
                unsigned d = key - kmin;
                unsigned hi = rb >= 32 ? 0u : (d >> rb);
                if (hi == prefix) atomicAdd(&hist[(d >> shift) & ((1u << b) - 1u)], 1u);
            }
        }
        __syncthreads();
        select_scan_bins(hist, k, sh);
        __syncthreads();
        prefix = (prefix << b) | sh.sel_bin;
        k = sh.sel_k;
        rb = shift;
    }
    return kmin + prefix;
}

template <int BLOCK, class KeyFn>
__device__ unsigned long long block_select_u64(KeyFn keyfn, int n, unsigned k,
                                               unsigned long long kmin, unsigned long long kmax,
                                               unsigned *hist, FpShared &sh) {
    const unsigned long long range = kmax - kmin;
    int rb = range ? 64 - __clzll((long long)range) : 0;
    unsigned long long prefix = 0;
    while (rb > 0) {
        const int b = rb < 8 ? rb : 8;
        const int shift = rb - b;
        for (int t = threadIdx.x; t < 256; t += BLOCK) hist[t] = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += BLOCK) {
            unsigned long long key;
            if (keyfn(i, key)) {
                unsigned long long d = key - kmin;
                unsigned long long hi = rb >= 64 ? 0ull : (d >> rb);
                if (hi == prefix) atomicAdd(&hist[(unsigned)(d >> shift) & ((1u << b) - 1u)], 1u);
            }
        }
        __syncthreads();
        select_scan_bins(hist, k, sh);
        __syncthreads();
        prefix = (prefix << b) | sh.sel_bin;
        k = sh.sel_k;
        rb = shift;
    }
    return kmin + prefix;
}

// np.nanmedian over the float32 values val(i) (NaN = invalid): exact, float32 result.
template <int BLOCK, class ValFn>
__device__ float block_nanmedian_f32(ValFn val, int n, unsigned *hist, FpShared &sh) {
    unsigned kmin = 0xffffffffu, kmax = 0, cnt = 0;
    for (int i = threadIdx.x; i < n; i += BLOCK) {
        float v = val(i);
        if (v == v) {
            unsigned k = f32_key(v);
            kmin = min(kmin, k);
            kmax = max(kmax, k);
            cnt++;
        }
    }
    unsigned gmin, gmax, m;
    block_minmax_count<BLOCK>(kmin, kmax, cnt, sh, gmin, gmax, m);
    if (m == 0) return __builtin_nanf("");
    auto keyfn = [&](int i, unsigned &key) {
        float v = val(i);
        key = f32_key(v);
        return v == v;
    };
    const unsigned h = m / 2;
    if (m & 1) return key_f32(block_select_u32<BLOCK>(keyfn, n, h, gmin, gmax, hist, sh));
    // even: ranks h-1 and h.  khi = klo if more than h keys are <= klo, else the next larger key.
    const unsigned klo = block_select_u32<BLOCK>(keyfn, n, h - 1, gmin, gmax, hist, sh);
    unsigned nxt = 0xffffffffu, dummy = 0, le = 0;
    for (int i = threadIdx.x; i < n; i += BLOCK) {
        unsigned key;
        if (keyfn(i, key)) {
            if (key <= klo) le++;
            else nxt = min(nxt, key);
        }
    }
    unsigned gnxt, gdummy, gle;
    block_minmax_count<BLOCK>(nxt, dummy, le, sh, gnxt, gdummy, gle);
    const unsigned khi = gle > h ? klo : gnxt;
    const float s = key_f32(klo) + key_f32(khi);
    return s / 2.0f;
}

// numpy pairwise float64 summation (loops_utils.h.src::pairwise_sum), single thread.
__device__ __forceinline__ double np_pairwise_leaf(const double *p, int n) {  // n <= 128
    double res;
    if (n < 8) {
        res = 0.0;
        for (int i = 0; i < n; ++i) res += p[i];
    } else {
        double r0 = p[0], r1 = p[1], r2 = p[2], r3 = p[3], r4 = p[4], r5 = p[5], r6 = p[6],
               r7 = p[7];
        int i;
        for (i = 8; i < n - (n % 8); i += 8) {
            r0 += p[i]; r1 += p[i + 1]; r2 += p[i + 2]; r3 += p[i + 3];
            r4 += p[i + 4]; r5 += p[i + 5]; r6 += p[i + 6]; r7 += p[i + 7];
        }
        res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
        for (; i < n; ++i) res += p[i];
    }
    return res;
}
// n <= 256 (kSegCap = 254): at most one level of the recursion
__device__ double np_pairwise_sum_dev(const double *a, int n) {
    if (n <= 128) return np_pairwise_leaf(a, n);
    int n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise_leaf(a, n2) + np_pairwise_leaf(a + n2, n - n2);
}

// np.add.reduce over the float32 values val(0..n): NumPy's exact association -- the ufunc hands the inner loop at
// most 8192 elements (np.getbufsize()) at a time, each chunk is summed by the pairwise routine (leaves of <= 128
// elements with 8 accumulators, halves split at multiples of 8), chunk sums accumulate left to right.  The leaves
// are independent: thread 0 lists them, one thread sums each, thread 0 folds them back up the same tree.
// lstart: >= 258 unsigned, lsum: >= 257 floats, stk: >= 48 ints (all LDS).  Result to every thread.
template <int BLOCK, class ValFn>
__device__ float block_np_add_reduce_f32(ValFn val, int n, unsigned *lstart, float *lsum, int *stk, FpShared &sh) {
    constexpr int kBuf = 8192;
    __syncthreads();
    if (threadIdx.x == 0) {
        int nl = 0;
        for (int c = 0; c < n || c == 0; c += kBuf) {
            const int m = min(kBuf, n - c);
            // depth-first, left before right: leaves come out in increasing start order
            int sp = 0;
            stk[0] = c;
            stk[1] = m;
            sp = 1;
            while (sp > 0) {
                --sp;
                const int lo = stk[2 * sp], len = stk[2 * sp + 1];
                if (len <= 128) {
                    lstart[nl++] = (unsigned)lo;
                } else {
                    int n2 = len / 2;
                    n2 -= n2 % 8;
                    stk[2 * sp] = lo + n2;       // right half: popped second
                    stk[2 * sp + 1] = len - n2;
                    ++sp;
                    stk[2 * sp] = lo;            // left half: popped first
                    stk[2 * sp + 1] = n2;
                    ++sp;
                }
            }
            if (n == 0) break;
        }
        lstart[nl] = (unsigned)n;
        sh.count = nl;
    }
    __syncthreads();
    const int nl = sh.count;
    for (int l = threadIdx.x; l < nl; l += BLOCK) {
        const int lo = (int)lstart[l];
        // a leaf ends at the next leaf's start, and never crosses an 8192-element chunk
        const int hi = min((int)lstart[l + 1], (lo / kBuf + 1) * kBuf);
        const int len = hi - lo;
        float res;
        if (len < 8) {
            res = 0.0f;
            for (int i = 0; i < len; ++i) res += val(lo + i);
        } else {
            float r0 = val(lo), r1 = val(lo + 1), r2 = val(lo + 2), r3 = val(lo + 3), r4 = val(lo + 4),
                  r5 = val(lo + 5), r6 = val(lo + 6), r7 = val(lo + 7);
            int i;
            for (i = 8; i < len - (len % 8); i += 8) {
                r0 += val(lo + i); r1 += val(lo + i + 1); r2 += val(lo + i + 2); r3 += val(lo + i + 3);
                r4 += val(lo + i + 4); r5 += val(lo + i + 5); r6 += val(lo + i + 6); r7 += val(lo + i + 7);
            }
            res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
            for (; i < len; ++i) res += val(lo + i);
        }
        lsum[l] = res;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float total = 0.0f;
        int li = 0;
        for (int c = 0; c < n || c == 0; c += kBuf) {
            const int m = min(kBuf, n - c);
            // post-order fold: frame = (len, phase), value stack alongside (float bits in the int array)
            float ret = 0.0f;
            int sp = 0;
            stk[0] = m;
            stk[1] = 0;
            sp = 1;
            while (sp > 0) {
                int &len = stk[3 * (sp - 1)], &phase = stk[3 * (sp - 1) + 1], &left = stk[3 * (sp - 1) + 2];
                if (phase == 0) {
                    if (len <= 128) {
                        ret = lsum[li++];
                        --sp;
                    } else {
                        int n2 = len / 2;
                        n2 -= n2 % 8;
                        phase = 1;
                        stk[3 * sp] = n2;
                        stk[3 * sp + 1] = 0;
                        ++sp;
                    }
                } else if (phase == 1) {
                    left = __float_as_int(ret);
                    phase = 2;
                    int n2 = len / 2;
                    n2 -= n2 % 8;
                    stk[3 * sp] = len - n2;
                    stk[3 * sp + 1] = 0;
                    ++sp;
                } else {
                    ret = __int_as_float(left) + ret;
                    --sp;
                }
            }
            total = (c == 0) ? ret : total + ret;
            if (n == 0) break;
        }
        sh.med = total;
    }
    __syncthreads();
    const float out = sh.med;
    __syncthreads();
    return out;
}

// np.median of n <= kSegCap float64 values in LDS (no NaN) by rank counting; result via sh.stat[slot]
template <int BLOCK>
__device__ void block_small_median(const double *a, int n, FpShared &sh, int slot) {
    const int klo = (n - 1) / 2, khi = n / 2;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += BLOCK) {
        const double v = a[i];
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const double u = a[j];
            rank += (u < v) || (u == v && j < i);
        }
        if (rank == klo) sh.red64[0] = (unsigned long long)__double_as_longlong(v);
        if (rank == khi) sh.red64[1] = (unsigned long long)__double_as_longlong(v);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double lo = __longlong_as_double((long long)sh.red64[0]);
        double hi = __longlong_as_double((long long)sh.red64[1]);
        sh.stat[slot] = (n & 1) ? hi : (lo + hi) / 2.0;
    }
    __syncthreads();
}

// find_peaks(scores[0..ns), distance=d_eff) -> the E highest -> boundaries 0, peaks + W, n_end
// -> event means over sig (sig_proc.py:176-198, segmentation.py:48-74).  Used for the adapter and, in the
// consensus-refinement branch, once more for the barcode tail of the same score curve.  All threads call;
// returns a block-uniform WDX_READ_* status; on success cpts[0..nseg] and ev[0..nseg).
// block-wide exclusive scan of one int per thread (chunk counts); total = block sum.  Two barriers.
template <int BLOCK>
__device__ __forceinline__ int block_excl_scan(int v, FpShared &sh, int &total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
    }
    __syncthreads();  // sh.red_a may still be read by the previous user
    if (lane == 63) sh.red_a[wave] = (unsigned)incl;
    __syncthreads();
    int base = 0;
    total = 0;
#pragma unroll
    for (int k = 0; k < BLOCK / 64; ++k) {
        const int w = (int)sh.red_a[k];
        base += k < wave ? w : 0;
        total += w;
    }
    return base + incl - v;
}

template <int BLOCK>
__device__ int fp_segment(const double *scores, unsigned char *state, const int ns, const int d_eff, const int W,
                          const int E, const bool accept_less, const float *sig, const int n_end, int *cpts, double *ev,
                          unsigned *hist, FpShared &sh, int &nseg, int &nms_iters, const bool no_list = false,
                          const bool exact_sums = false) {
    const int tid = threadIdx.x;
    // ---- P3: find_peaks(scores, distance=d_eff) ----------------------------------
    for (int i = tid; i < ns; i += BLOCK) state[i] = ST_NONE;
    __syncthreads();
    for (int i = 1 + tid; i < ns - 1; i += BLOCK) {
        const double s = scores[i];
        if (scores[i - 1] < s) {
            int ia = i + 1;
            while (ia < ns - 1 && scores[ia] == s) ++ia;
            if (scores[ia] < s) state[(i + ia - 1) / 2] = ST_UNDECIDED;
        }
    }
    __syncthreads();
    // The local maxima (about ns / 5.6 of the positions) are compacted into a position-ordered LIST that overwrites
    // the state bytes in place -- uint16 positions, then one state byte per peak: 3 bytes per peak against one byte per
    // position, so it fits whenever np <= ns / 3 (else, and with WDX_OPT_EXACT_NO_PEAK_LIST, the position-space
    // code below runs; same decisions, same results).  Suppression, the top-E cut and the boundaries then walk
    // ~ns / 5.6 entries instead of ns positions: the suppression's neighbourhood scan is a few list entries instead
    // of 2 (d - 1) state bytes, and every pass of the select is 5.6x shorter.
    int nsel;
    bool use_list;
    int np = 0, lbase = 0;
    unsigned lbits = 0;
    const int lchunk = (ns + BLOCK - 1) / BLOCK;  // <= 32: cap <= 16 384, BLOCK >= 512
    const int lc0 = tid * lchunk;
    {
        const int lc1 = min(ns, lc0 + lchunk);
        for (int i = lc0; i < lc1; ++i) lbits |= (state[i] == ST_UNDECIDED ? 1u : 0u) << (i - lc0);
        lbase = block_excl_scan<BLOCK>(__popc(lbits), sh, np);  // (its barriers: every state byte has been read)
        use_list = !no_list && lchunk <= 32 && 3 * np + 8 <= ns;
    }
    if (use_list) {
        unsigned short *lpos = reinterpret_cast<unsigned short *>(state);
        unsigned char *lst = state + ((2 * np + 3) & ~3);
        {
            int o = lbase;
            unsigned b = lbits;
            while (b) {
                const int j = __ffs((int)b) - 1;
                b &= b - 1;
                lpos[o] = (unsigned short)(lc0 + j);
                lst[o] = ST_UNDECIDED;
                ++o;
            }
        }
        __syncthreads();
        // greedy suppression by priority (see the position-space form below for the rule): a peak's rivals are the list
        // neighbours closer than d_eff; ties -> the LATER peak outranks.  Only the rivals that OUTRANK a peak decide
        // about it (a lower-ranked rival stays undecided for as long as this peak is), and which those are never
        // changes: each thread finds them once -- a 16-bit mask over the up to eight list neighbours on either side
        // (local maxima are at least two positions apart, d_eff <= 17) -- and the rounds only re-read their state bytes.
        constexpr int kNmsPer = 4;   // list entries per thread held in registers
        if (d_eff <= 17 && np <= kNmsPer * BLOCK) {
            unsigned hp[kNmsPer];
            bool und[kNmsPer];
#pragma unroll
            for (int m = 0; m < kNmsPer; ++m) {
                const int k = tid + m * BLOCK;
                hp[m] = 0;
                und[m] = k < np;
                if (k < np) {
                    const int p = lpos[k];
                    const double s = scores[p];
                    for (int q = k - 1; q >= 0 && p - (int)lpos[q] < d_eff; --q)
                        if (scores[lpos[q]] > s) hp[m] |= 1u << (q - k + 8);
                    for (int q = k + 1; q < np && (int)lpos[q] - p < d_eff; ++q) {
                        const double sq = scores[lpos[q]];
                        if (sq > s || sq == s) hp[m] |= 1u << (q - k + 7);
                    }
                }
            }
            for (;;) {
                int pending = 0;
#pragma unroll
                for (int m = 0; m < kNmsPer; ++m) {
                    if (!und[m]) continue;
                    const int k = tid + m * BLOCK;
                    bool kept_near = false, wait = false;
                    unsigned mask = hp[m];
                    while (mask) {
                        const int b = __ffs((int)mask) - 1;
                        mask &= mask - 1;
                        const unsigned char st = lst[k + (b < 8 ? b - 8 : b - 7)];
                        kept_near |= st == ST_KEPT;
                        wait |= st == ST_UNDECIDED;
                    }
                    if (kept_near) { lst[k] = ST_DROPPED; und[m] = false; }
                    else if (!wait) { lst[k] = ST_KEPT; und[m] = false; }
                    else pending = 1;
                }
                ++nms_iters;
                if (!__syncthreads_or(pending)) break;
            }
        } else
        for (;;) {
            int pending = 0;
            for (int k = tid; k < np; k += BLOCK) {
                if (lst[k] != ST_UNDECIDED) continue;
                const int p = lpos[k];
                const double s = scores[p];
                bool kept_near = false, wait = false;
                for (int q = k - 1; q >= 0 && p - (int)lpos[q] < d_eff; --q) {
                    const unsigned char st = lst[q];
                    if (st == ST_KEPT) kept_near = true;
                    else if (st == ST_UNDECIDED && scores[lpos[q]] > s) wait = true;
                }
                for (int q = k + 1; q < np && (int)lpos[q] - p < d_eff; ++q) {
                    const unsigned char st = lst[q];
                    if (st == ST_KEPT) kept_near = true;
                    else if (st == ST_UNDECIDED) {
                        const double sq = scores[lpos[q]];
                        if (sq > s || sq == s) wait = true;
                    }
                }
                if (kept_near) lst[k] = ST_DROPPED;
                else if (!wait) lst[k] = ST_KEPT;
                else pending = 1;
            }
            ++nms_iters;
            if (!__syncthreads_or(pending)) break;
        }
        // ---- P4 on the list: keep the E highest peaks (sig_proc.py:185-188) --------------------------
        unsigned long long kmin = ~0ull, kmax = 0;
        unsigned cnt = 0;
        for (int k = tid; k < np; k += BLOCK) {
            if (lst[k] == ST_KEPT) {
                const unsigned long long key = (unsigned long long)__double_as_longlong(scores[lpos[k]]);
                kmin = key < kmin ? key : kmin;
                kmax = key > kmax ? key : kmax;
                cnt++;
            }
        }
        unsigned long long gmin, gmax;
        unsigned nk;
        block_minmax64_count<BLOCK>(kmin, kmax, cnt, sh, gmin, gmax, nk);
        if ((int)nk < E && !accept_less) return WDX_READ_FAIL_SEGMENT;
        if (nk == 0) return WDX_READ_FAIL_UNKNOWN;  // valid_cpts[0] on an empty array
        if ((int)nk <= E) {
            nsel = (int)nk;
            __syncthreads();
            for (int k = tid; k < np; k += BLOCK)
                if (lst[k] == ST_KEPT) lst[k] = ST_SELECTED;
        } else if ((int)nk <= kSegCap) {
            // few survivors (a barcode tail: ~100): their keys go, in list order, to a dense array and every survivor
            // counts the ones that outrank it (larger score; equal score and later) -- one pass instead of the seven
            // digit passes of the radix select below
            nsel = E;
            unsigned long long *dk = reinterpret_cast<unsigned long long *>(ev);   // (free until P6)
            unsigned short *dmap = reinterpret_cast<unsigned short *>(hist);
            const int c2 = (np + BLOCK - 1) / BLOCK;
            const int k0 = tid * c2, k1 = min(np, k0 + c2);
            int local = 0;
            for (int k = k0; k < k1; ++k) local += lst[k] == ST_KEPT;
            int tot;
            int o = block_excl_scan<BLOCK>(local, sh, tot);
            for (int k = k0; k < k1; ++k)
                if (lst[k] == ST_KEPT) {
                    dk[o] = (unsigned long long)__double_as_longlong(scores[lpos[k]]);
                    dmap[o] = (unsigned short)k;
                    ++o;
                }
            __syncthreads();
            for (int t = tid; t < (int)nk; t += BLOCK) {
                const unsigned long long key = dk[t];
                int rank = 0;
                for (int j = 0; j < (int)nk; ++j) {
                    const unsigned long long kj = dk[j];
                    rank += (kj > key) || (kj == key && j > t);
                }
                if (rank < E) lst[dmap[t]] = ST_SELECTED;
            }
        } else {
            nsel = E;
            auto keyfn = [&](int k, unsigned long long &key) {
                key = (unsigned long long)__double_as_longlong(scores[lpos[k]]);
                return lst[k] == ST_KEPT;
            };
            const unsigned long long T = block_select_u64<BLOCK>(keyfn, np, nk - (unsigned)E, gmin, gmax, hist, sh);
            unsigned gt = 0, eq = 0, z0 = 0xffffffffu;
            for (int k = tid; k < np; k += BLOCK) {
                if (lst[k] == ST_KEPT) {
                    const unsigned long long key = (unsigned long long)__double_as_longlong(scores[lpos[k]]);
                    gt += key > T;
                    eq += key == T;
                }
            }
            unsigned ggt, geq, dmy;
            block_minmax_count<BLOCK>(z0, gt, eq, sh, dmy, dmy, geq);  // geq = sum(eq)
            block_minmax_count<BLOCK>(z0, 0u, gt, sh, dmy, dmy, ggt);  // ggt = sum(gt)
            const unsigned need = (unsigned)E - ggt;  // 1 <= need <= geq
            __syncthreads();
            for (int k = tid; k < np; k += BLOCK) {
                if (lst[k] == ST_KEPT) {
                    const unsigned long long key = (unsigned long long)__double_as_longlong(scores[lpos[k]]);
                    if (key > T || (key == T && need == geq)) lst[k] = ST_SELECTED;
                }
            }
            __syncthreads();
            if (need != geq && tid == 0) {
                // exact score ties at the cut: the stable order keeps the LAST `need` of them
                unsigned left = need;
                for (int k = np - 1; k >= 0 && left; --k) {
                    if (lst[k] == ST_KEPT && (unsigned long long)__double_as_longlong(scores[lpos[k]]) == T) {
                        lst[k] = ST_SELECTED;
                        --left;
                    }
                }
            }
        }
        __syncthreads();
        // ---- P5 on the list: boundaries 0, peaks+W (ascending), n  (sig_proc.py:188-196) ---------------
        {
            const int c2 = (np + BLOCK - 1) / BLOCK;
            const int k0 = tid * c2, k1 = min(np, k0 + c2);
            int local = 0;
            for (int k = k0; k < k1; ++k) local += lst[k] == ST_SELECTED;
            int tot;
            int o = block_excl_scan<BLOCK>(local, sh, tot);
            for (int k = k0; k < k1; ++k)
                if (lst[k] == ST_SELECTED) cpts[1 + o++] = (int)lpos[k] + W;
            if (tid == 0) {
                cpts[0] = 0;
                cpts[nsel + 1] = n_end;
            }
            __syncthreads();
        }
    } else {
    {
        // greedy suppression by priority == fixed point of: a peak is KEPT once every higher-priority
        // peak closer than d_eff is DROPPED, and DROPPED as soon as one of them is KEPT.
        // priority order: score, ties -> larger position first (stable argsort read from the end).
        const int D1 = d_eff - 1;
        for (;;) {
            int pending = 0;
            for (int i = tid; i < ns; i += BLOCK) {
                if (state[i] != ST_UNDECIDED) continue;
                const double s = scores[i];
                bool kept_near = false, wait = false;
                const int lo = max(0, i - D1), hi = min(ns - 1, i + D1);
                for (int q = lo; q <= hi; ++q) {
                    if (q == i) continue;
                    const unsigned char st = state[q];
                    if (st == ST_KEPT) kept_near = true;
                    else if (st == ST_UNDECIDED) {
                        const double sq = scores[q];
                        if (sq > s || (sq == s && q > i)) wait = true;
                    }
                }
                if (kept_near) state[i] = ST_DROPPED;
                else if (!wait) state[i] = ST_KEPT;
                else pending = 1;
            }
            ++nms_iters;
            if (!__syncthreads_or(pending)) break;
        }
    }

    // ---- P4: keep the E highest peaks (sig_proc.py:185-188) -----------------------------------------
    {
        unsigned long long kmin = ~0ull, kmax = 0;
        unsigned cnt = 0;
        for (int i = tid; i < ns; i += BLOCK) {
            if (state[i] == ST_KEPT) {
                unsigned long long k = (unsigned long long)__double_as_longlong(scores[i]);
                kmin = k < kmin ? k : kmin;
                kmax = k > kmax ? k : kmax;
                cnt++;
            }
        }
        unsigned long long gmin, gmax;
        unsigned nk;
        block_minmax64_count<BLOCK>(kmin, kmax, cnt, sh, gmin, gmax, nk);
        if ((int)nk < E && !accept_less) return WDX_READ_FAIL_SEGMENT;
        if (nk == 0) return WDX_READ_FAIL_UNKNOWN;  // valid_cpts[0] on an empty array
        if ((int)nk <= E) {
            nsel = (int)nk;
            __syncthreads();
            for (int i = tid; i < ns; i += BLOCK)
                if (state[i] == ST_KEPT) state[i] = ST_SELECTED;
        } else {
            nsel = E;
            auto keyfn = [&](int i, unsigned long long &key) {
                key = (unsigned long long)__double_as_longlong(scores[i]);
                return state[i] == ST_KEPT;
            };
            const unsigned long long T =
                block_select_u64<BLOCK>(keyfn, ns, nk - (unsigned)E, gmin, gmax, hist, sh);
            unsigned gt = 0, eq = 0, z0 = 0xffffffffu;
            for (int i = tid; i < ns; i += BLOCK) {
                if (state[i] == ST_KEPT) {
                    unsigned long long k = (unsigned long long)__double_as_longlong(scores[i]);
                    gt += k > T;
                    eq += k == T;
                }
            }
            unsigned ggt, geq, dmy;
            block_minmax_count<BLOCK>(z0, gt, eq, sh, dmy, dmy, geq);  // geq = sum(eq)
            block_minmax_count<BLOCK>(z0, 0u, gt, sh, dmy, dmy, ggt);  // ggt = sum(gt)
            const unsigned need = (unsigned)E - ggt;  // 1 <= need <= geq
            __syncthreads();
            for (int i = tid; i < ns; i += BLOCK) {
                if (state[i] == ST_KEPT) {
                    unsigned long long k = (unsigned long long)__double_as_longlong(scores[i]);
                    if (k > T || (k == T && need == geq)) state[i] = ST_SELECTED;
                }
            }
            __syncthreads();
            if (need != geq && tid == 0) {
                // exact score ties at the cut: the stable order keeps the LAST `need` of them
                unsigned left = need;
                for (int i = ns - 1; i >= 0 && left; --i) {
                    if (state[i] == ST_KEPT &&
                        (unsigned long long)__double_as_longlong(scores[i]) == T) {
                        state[i] = ST_SELECTED;
                        --left;
                    }
                }
            }
        }
        __syncthreads();
    }

    // ---- P5: boundaries 0, peaks+W (ascending), n  (sig_proc.py:188-196) -----------------------------
    {
        const int chunk = (ns + BLOCK - 1) / BLOCK;
        const int c0 = tid * chunk, c1 = min(ns, c0 + chunk);
        int local = 0;
        for (int i = c0; i < c1; ++i) local += state[i] == ST_SELECTED;
        // block exclusive scan of `local`
        int incl = local;
        const int lane = tid & 63, wave = tid >> 6;
        for (int off = 1; off < 64; off <<= 1) {
            int t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        if (lane == 63) sh.red_a[wave] = (unsigned)incl;
        __syncthreads();
        int base = 0;
        for (int k = 0; k < wave; ++k) base += (int)sh.red_a[k];
        int o = base + incl - local;
        for (int i = c0; i < c1; ++i)
            if (state[i] == ST_SELECTED) cpts[1 + o++] = i + W;
        if (tid == 0) {
            cpts[0] = 0;
            cpts[nsel + 1] = n_end;
        }
        __syncthreads();
    }
    }
    nseg = nsel + 1;

    // ---- P6: event means (_c_segmentation.pyx:41-53), sequential float64 sums ------------------------
    // exact_sums (the caller's promise: every partial sum of the samples is exactly representable in float64 -- positive
    // float32 samples within a factor 2^16 of each other): the sum does not depend on its order, eight lanes share a segment
    if (exact_sums) {
        const int j = tid & 7;
        for (int s0 = 0; s0 < nseg; s0 += BLOCK / 8) {   // block-uniform trip count: the shuffles see whole waves
            const int s = s0 + (tid >> 3);
            double sum = 0.0;
            int b = 0, e = 1;
            if (s < nseg) {
                b = cpts[s];
                e = cpts[s + 1];
                for (int i = b + j; i < e; i += 8) sum += (double)sig[i];
            }
            sum += __shfl_xor(sum, 4);
            sum += __shfl_xor(sum, 2);
            sum += __shfl_xor(sum, 1);
            if (s < nseg && j == 0) ev[s] = sum / (double)(e - b);
        }
    } else
    for (int s = tid; s < nseg; s += BLOCK) {
        const int b = cpts[s], e = cpts[s + 1];
        double sum = 0.0;
        for (int i = b; i < e; ++i) sum += (double)sig[i];
        ev[s] = sum / (double)(e - b);
    }
    __syncthreads();

    return WDX_READ_OK;
}

// Consensus-guided barcode refinement, everything after the adapter's segmentation (sig_proc.py:287-378,
// 452-521; oracle: fingerprint_refine_one_impl).  The subsequence match (dtaidistance warping_paths_fast +
// SubsequenceAlignment.best_match, restated -- parity unpinned) is an anti-diagonal wavefront:
// thread i owns query row i, front k holds the cells (i, k - i); a cell needs the two previous fronts only, so
// three fronts of (cost, sqrt(cost)) pairs rotate through LDS, and what the back-trace needs -- argmin of the
// three predecessors' sqrt'ed costs per cell, first minimum -- is kept as 2-bit codes, 16 per word, one row of
// words per thread.  scratch: >= 3*(nq+1)*16 + nq*ceil(nseries/16)*4 bytes of LDS (the t-score tile buffers).
// `prep(sbs, ns2, scores_tail, sig_tail)` is called (block-uniform) once the barcode's first sample `sbs` is known and
// hands back the score curve and the clipped samples FROM `sbs` ON: the exact kernel has both in LDS already
// (scores + sbs, sig + sbs); fingerprint_refine_tail_kernel -- the refinement behind the fast kernels -- loads and
// scores only that tail.  Returning false leaves the read untouched (the caller hands it on).
// The branch comes in two halves so that the kernel behind the fast kernels can run them as two launches with very
// different LDS needs (the match is pure latency and wants many resident workgroups; the barcode's segmentation
// needs the tail's samples and scores): fp_refine_match -- adapter statistics, subsequence match, back-trace ->
// RefineMatch (false: the read has been reported, nothing more to do) -- and fp_refine_finish.
template <int BLOCK>
__device__ bool fp_refine_match(const FpArgs &A, const int64_t r, const int *cpts, double *ev, double *zz, double *tmp,
                                unsigned char *scratch, FpShared &sh, const int nseg, RefineMatch &M) {
    const int tid = threadIdx.x;
    const wdx_seg_params &P = A.p;
    const RefineDev &R = A.rf;
    const int K = P.barcode_num_events;  // = barcode_num_events[1], set by the host
    auto finish = [&](int st, bool with_stats) {
        if (st != WDX_READ_OK) {
            for (int i = tid; i < K; i += BLOCK) {
                if (A.fpt) A.fpt[r * K + i] = __builtin_nan("");
                if (A.dwell) A.dwell[r * K + i] = 0;
            }
            if (!with_stats) {
                if (A.stats && tid < 6) A.stats[r * 6 + tid] = __builtin_nan("");
                if (R.idx && tid < 3) R.idx[r * 3 + tid] = -1;
            }
        }
        if (tid == 0) A.status[r] = st;
    };
    // normalize(series, method, accept_nan=False) inside _get_subseq_match raises on NaN -> "unknown"
    {
        int has_nan = 0;
        for (int s = tid; s < nseg; s += BLOCK) has_nan |= (ev[s] != ev[s]);
        if (__syncthreads_or(has_nan)) {
            finish(WDX_READ_FAIL_UNKNOWN, false);
            return false;
        }
    }
    // adapter statistics (sig_proc.py:486-494) and the shift / scale of both normalisations
    if (tid == 0) sh.mean = np_pairwise_sum_dev(ev, nseg) / (double)nseg;
    __syncthreads();
    const double mean = sh.mean;
    for (int s = tid; s < nseg; s += BLOCK) {
        const double df = ev[s] - mean;
        tmp[s] = df * df;
    }
    __syncthreads();
    if (tid == 0) sh.sd = sqrt(np_pairwise_sum_dev(tmp, nseg) / (double)nseg);
    __syncthreads();
    const double sd = sh.sd;
    for (int s = tid; s < nseg; s += BLOCK) tmp[s] = (double)(cpts[s + 1] - cpts[s]);
    block_small_median<BLOCK>(tmp, nseg, sh, 0);
    const double dt_med = sh.stat[0];
    for (int s = tid; s < nseg; s += BLOCK) tmp[s] = fabs(tmp[s] - dt_med);
    block_small_median<BLOCK>(tmp, nseg, sh, 1);
    block_small_median<BLOCK>(ev, nseg, sh, 4);
    const double ev_med = sh.stat[4];
    for (int s = tid; s < nseg; s += BLOCK) tmp[s] = fabs(ev[s] - ev_med);
    block_small_median<BLOCK>(tmp, nseg, sh, 5);
    const double ev_mad = sh.stat[5];
    const double dt_mad = sh.stat[1];
    // series of the match: normalize(adapter_event_means, consensus_subseq_match_normalization)
    {
        double c0 = 0.0, c1 = 1.0;
        if (R.norm == WDX_NORM_MEAN) { c0 = mean; c1 = sd; }
        else if (R.norm == WDX_NORM_MEDIAN) { c0 = ev_med; c1 = ev_mad; }
        else if (R.norm != WDX_NORM_NONE) { finish(WDX_READ_FAIL_UNKNOWN, false); return false; }
        int bad = 0;
        for (int s = tid; s < nseg; s += BLOCK) {
            const double v = R.norm == WDX_NORM_NONE ? ev[s] : (ev[s] - c0) / c1;
            zz[s] = v;
            bad |= (v != v);
        }
        if (__syncthreads_or(bad)) {  // a constant series (0/0): the library's behaviour on NaN is not restated
            finish(WDX_READ_FAIL_UNKNOWN, false);
            return false;
        }
    }
    // ---- subsequence DTW of the consensus query against the normalised event means ------------------------
    const int nq = R.nq, c = nseg, WPR = (c + 15) >> 4;
    struct DS { double D, S; };
    DS *F = reinterpret_cast<DS *>(scratch);
    unsigned *dirw = reinterpret_cast<unsigned *>(F + 3 * (nq + 1));
    double *lastS = tmp;
    {
        const double p2 = R.pen * R.pen, inf = __builtin_huge_val();
        const double qi = (tid >= 1 && tid <= nq) ? R.query[tid - 1] : 0.0;
        unsigned acc = 0;
        for (int k = 0; k <= nq + c; ++k) {
            if (tid <= nq) {
                const int i = tid, j = k - i;
                if (j >= 0 && j <= c) {
                    DS out;
                    if (i == 0) {
                        out.D = j <= R.psi2b ? 0.0 : inf;
                        out.S = out.D;
                    } else if (j == 0) {
                        out.D = i <= R.psi1b ? 0.0 : inf;
                        out.S = out.D;
                    } else {
                        const DS dg = F[((k - 2) % 3) * (nq + 1) + i - 1], up = F[((k - 1) % 3) * (nq + 1) + i - 1],
                                 lf = F[((k - 1) % 3) * (nq + 1) + i];
                        double d = qi - zz[j - 1];
                        d = d * d;
                        double m = dg.D, t = up.D + p2;
                        if (t < m) m = t;
                        t = lf.D + p2;
                        if (t < m) m = t;
                        out.D = d + m;
                        out.S = sqrt(out.D);
                        unsigned code = 0;
                        double mv = dg.S;
                        if (up.S < mv) { mv = up.S; code = 1; }
                        if (lf.S < mv) code = 2;
                        acc |= code << (2 * ((j - 1) & 15));
                        if (((j - 1) & 15) == 15 || j == c) {
                            dirw[(i - 1) * WPR + ((j - 1) >> 4)] = acc;
                            acc = 0;
                        }
                        if (i == nq) lastS[j - 1] = out.S;
                    }
                    F[(k % 3) * (nq + 1) + i] = out;
                }
            }
            __syncthreads();
        }
    }
    if (tid == 0) {
        int best = 0;
        double bv = lastS[0] / (double)nq;
        for (int j = 1; j < c; ++j) {
            const double v = lastS[j] / (double)nq;
            if (v < bv) { bv = v; best = j; }
        }
        int i = nq, j = best + 1, sj = j;
        while (i > 0 && j > 0) {
            sj = j;
            const unsigned code = (dirw[(i - 1) * WPR + ((j - 1) >> 4)] >> (2 * ((j - 1) & 15))) & 3u;
            if (code == 0) { --i; --j; }
            else if (code == 1) --i;
            else --j;
        }
        sh.flag = sj - 1;   // seg_cons_query_start
        sh.count = best;    // seg_cons_query_end
    }
    __syncthreads();
    const int qs = sh.flag, qe = sh.count;
    const int sbs = cpts[qe];   // int(np.sum(adapter_dwell_times[:seg_query_end]))
    __syncthreads();            // cpts is rewritten below
    M = RefineMatch{mean, sd, ev_med, ev_mad, dt_med, dt_mad, qs, qe, sbs, 0};
    return true;
}

template <int BLOCK, class Prep>
__device__ void fp_refine_finish(const FpArgs &A, const int64_t r, const RefineMatch &M, unsigned char *state, const int ns,
                                 const int n, int *cpts, double *zz, unsigned *hist, FpShared &sh, Prep prep,
                                 const bool exact_sums = false) {
    const int tid = threadIdx.x;
    const wdx_seg_params &P = A.p;
    const RefineDev &R = A.rf;
    const int K = P.barcode_num_events;
    auto finish = [&](int st, bool with_stats) {
        if (st != WDX_READ_OK) {
            for (int i = tid; i < K; i += BLOCK) {
                if (A.fpt) A.fpt[r * K + i] = __builtin_nan("");
                if (A.dwell) A.dwell[r * K + i] = 0;
            }
            if (!with_stats) {
                if (A.stats && tid < 6) A.stats[r * 6 + tid] = __builtin_nan("");
                if (R.idx && tid < 3) R.idx[r * 3 + tid] = -1;
            }
        }
        if (tid == 0) A.status[r] = st;
    };
    const double mean = M.mean, sd = M.sd, ev_med = M.ev_med, ev_mad = M.ev_mad, dt_med = M.dt_med, dt_mad = M.dt_mad;
    const int qs = M.qs, qe = M.qe, sbs = M.sbs;
    // ---- the barcode tail: discrepenacy_curve_to_cpts(adapter_scores[sbs:], E2, config d, config W, False) -----
    int ns2 = ns - sbs;
    if (ns2 < 0) ns2 = 0;
    if (P.min_obs_per_base < 1) {  // scipy: `distance` must be >= 1
        finish(WDX_READ_FAIL_UNKNOWN, false);
        return;
    }
    int nseg2 = 0, it2 = 0;
    const int n_end2 = ns2 + 2 * P.running_stat_width;
    const double *sc_t = nullptr;
    const float *sg_t = nullptr;
    if (!prep(sbs, ns2, sc_t, sg_t)) return;
    {
        // a last boundary beyond the slice (window width shrunk below the configured one): the Cython loop of
        // compute_base_means indexes out of bounds -> exception -> "unknown"; checked BEFORE the means are summed
        if (n_end2 != n - sbs) {
            // still "event segmentation failed" when the tail has too few peaks (that return comes first)
            const int st0 = fp_segment<BLOCK>(sc_t, state, ns2, P.min_obs_per_base, P.running_stat_width, R.E2,
                                              false, sg_t, min(n_end2, n - sbs), cpts, zz, hist, sh, nseg2, it2, A.no_list != 0);
            finish(st0 == WDX_READ_FAIL_SEGMENT ? WDX_READ_FAIL_SEGMENT : WDX_READ_FAIL_UNKNOWN, false);
            return;
        }
        const int st = fp_segment<BLOCK>(sc_t, state, ns2, P.min_obs_per_base, P.running_stat_width, R.E2, false,
                                         sg_t, n_end2, cpts, zz, hist, sh, nseg2, it2, A.no_list != 0, exact_sums);
        if (st != WDX_READ_OK) {
            finish(st, false);
            return;
        }
    }
    // normalize_wrt(barcode_event_means, adapter_event_means, segmentation.normalization) (sig_proc.py:139-168)
    double shift, scale;
    if (P.seg_norm == WDX_NORM_MEAN) { shift = mean; scale = sd; }
    else if (P.seg_norm == WDX_NORM_MEDIAN) { shift = ev_med; scale = ev_mad; }
    else { finish(WDX_READ_FAIL_UNKNOWN, false); return; }  // "none" is not a normalize_wrt method: ValueError
    // The final status is decided first and stats / indices are written ONCE: "consensus query outlier" and
    // success report them, the nseg2 < K "unknown" path (the np.pad call subtracts an int from a tuple ->
    // TypeError) reports NaN / -1 -- no second write to the same addresses by other lanes.
    const bool outlier = qs > R.ub_start || qe < R.lb_end || qe > R.ub_end;
    if (!outlier && nseg2 < K) {
        finish(WDX_READ_FAIL_UNKNOWN, false);
        return;
    }
    if (tid == 0) {
        if (A.stats) {
            double *o = A.stats + r * 6;
            o[0] = dt_med; o[1] = dt_mad; o[2] = mean; o[3] = sd; o[4] = ev_med; o[5] = ev_mad;
        }
        if (R.idx) {
            R.idx[r * 3] = qs; R.idx[r * 3 + 1] = qe; R.idx[r * 3 + 2] = sbs;
        }
    }
    if (outlier) {
        finish(WDX_READ_FAIL_CONSENSUS, true);  // "consensus query outlier": stats and indices are reported
        return;
    }
    for (int i = tid; i < K; i += BLOCK) {
        const int s = nseg2 - K + i;
        if (A.fpt) A.fpt[r * K + i] = (zz[s] - shift) / scale;
        if (A.dwell) A.dwell[r * K + i] = (int64_t)(cpts[s + 1] - cpts[s]);
    }
    finish(WDX_READ_OK, true);
}


template <int BLOCK, class Prep>
__device__ void fp_refine_tail(const FpArgs &A, const int64_t r, unsigned char *state,
                               const int ns, const int W, const int n, int *cpts, double *ev,
                               double *zz, double *tmp, unsigned char *scratch, unsigned *hist, FpShared &sh,
                               const int nseg, Prep prep) {
    (void)W;
    RefineMatch M;
    if (!fp_refine_match<BLOCK>(A, r, cpts, ev, zz, tmp, scratch, sh, nseg, M)) return;
    fp_refine_finish<BLOCK>(A, r, M, state, ns, n, cpts, zz, hist, sh, prep);
}

// ---- the kernel -----------------------------------------------------------------------------------

// PROF = diagnostic instantiation with s_memtime stamps between phases (never used by the product
// entry points; see wdx_fingerprint_profile_dev).
#define WDX_STAMP(k)                                                                        \
    do {                                                                                    \
        if (PROF) {                                                                         \
            __syncthreads();                                                                \
            if (tid == 0 && r < A.prof_reads)                                               \
                A.prof[r * 32 + (k)] = (long long)__builtin_amdgcn_s_memtime();             \
        }                                                                                   \
    } while (0)

template <int BLOCK, bool PROF, bool BIG = false>
__device__ void fp_process_read(const FpArgs &A, const int64_t r, unsigned char *smem) {
    const int tid = threadIdx.x;
    const wdx_seg_params &P = A.p;
    const int K = P.barcode_num_events;
    const int E = P.num_events;

    // LDS carve-up (every region starts 8-byte aligned; cap is a multiple of 64).  BIG: the score curve lives in
    // this workgroup's HBM slot instead (same code, global loads/stores: the address space follows the
    // instantiation), everything else stays in LDS.
    double *scores = BIG ? A.big_scores + (size_t)blockIdx.x * kBigCap : reinterpret_cast<double *>(smem);  // cap
    double *Mt = BIG ? reinterpret_cast<double *>(smem) : scores + A.cap;  // kTile + kMaxW
    double *Vt = Mt + (kTile + kMaxW);                                    // kTile + kMaxW
    double *ev = Vt + (kTile + kMaxW);                                    // kSegCap
    double *zz = ev + kSegCap;                                            // kSegCap
    double *tmp = zz + kSegCap;                                           // kSegCap
    FpShared &sh = *reinterpret_cast<FpShared *>(tmp + kSegCap);
    float *sig = reinterpret_cast<float *>(&sh + 1);                      // cap
    unsigned *hist = reinterpret_cast<unsigned *>(sig + A.cap);           // 256
    int *cpts = reinterpret_cast<int *>(hist + 256);                      // kSegCap + 1
    // the peak-state bytes are live only after the last t-score tile: they reuse the tile buffers
    // whenever they fit, else a dedicated tail region
    unsigned char *state = (A.cap <= (int)((kTile + kMaxW) * 16))
                               ? reinterpret_cast<unsigned char *>(Mt)
                               : reinterpret_cast<unsigned char *>(cpts + kSegCap + 1);

    int status = WDX_READ_OK;

    auto finish = [&](int st) {
        // failed reads: NaN fingerprint / stats, zero dwell (block-uniform call)
        if (st != WDX_READ_OK) {
            for (int i = tid; i < K; i += BLOCK) {
                if (A.fpt) A.fpt[r * K + i] = __builtin_nan("");
                if (A.dwell) A.dwell[r * K + i] = 0;
            }
            if (A.stats && tid < 6) A.stats[r * 6 + tid] = __builtin_nan("");
        }
        if (tid == 0) A.status[r] = st;
    };

    if (A.ok && !A.ok[r]) {
        finish(WDX_READ_FAIL_DETECT);
        return;
    }

    WDX_STAMP(0);
    // ---- A0 extract_adapter (sig_proc.py:382-391) --------------------------------------------------
    const int64_t row_off = A.row_off ? A.row_off[r] : r * A.stride;
    const int64_t row_len = A.row_len ? (int64_t)A.row_len[r]
                                      : (A.row_off ? A.row_off[r + 1] - A.row_off[r] : A.stride);
    int64_t start = (int64_t)A.a_start[r] - P.padding;
    if (start < 0) start = 0;
    int64_t stop = (int64_t)A.a_end[r] + P.padding;
    if (stop > row_len) stop = row_len;
    int64_t n64 = stop - start;
    if (n64 < 0) n64 = 0;
    if (n64 > A.cap) {
        if (!BIG && A.defer_big && n64 <= kBigCap) return;  // fingerprint_big_kernel takes it
        finish(WDX_READ_FAIL_UNKNOWN);
        return;
    }
    const int n = (int)n64;

    // ---- P0: HBM -> LDS ----------------------------------------------------------------------------
    {
        const float *__restrict__ src = A.sig + row_off + start;
        for (int i = tid; i < n; i += BLOCK) sig[i] = src[i];
    }
    __syncthreads();

    WDX_STAMP(1);
    // ---- P1: MAD outlier clip (sig_proc.py:421-431), float32 ---------------------------------------
    int any_nan = 0;          // (this thread saw a NaN sample / the bounds are NaN: only the refinement hand-over asks)
    float clip_lo = 0.f, clip_hi = 0.f;
    {
        float lo, hi;
        // behind the launch chain the bounds exist already (clip_bounds_kernel: the same float32 numbers; CLIP_OK also
        // says that the window holds no NaN and lo <= hi) -- the two workgroup-wide selects are a quarter of this kernel
        const ClipRec *crp = A.clip;
        int cflag = CLIP_NONE;
        if (crp) {
            const ClipRec cr = crp[r];   // (block-uniform)
            cflag = cr.flag;
            lo = cr.lo;
            hi = cr.hi;
        }
        if (cflag != CLIP_OK) {
            const float med = block_nanmedian_f32<BLOCK>([&](int i) { return sig[i]; }, n, hist, sh);
            const float mad =
                block_nanmedian_f32<BLOCK>([&](int i) { return fabsf(sig[i] - med); }, n, hist, sh);
            clip_bounds(P, med, mad, lo, hi);
        }
        const bool bad = (lo != lo) || (hi != hi);
        __syncthreads();
        for (int i = tid; i < n; i += BLOCK) {
            float v = sig[i];
            if (v == v) {
                if (bad) v = __builtin_nanf("");
                else {
                    if (!(v > lo)) v = lo;
                    if (!(v < hi)) v = hi;
                }
                sig[i] = v;
            } else {
                any_nan = 1;
            }
        }
        __syncthreads();
        clip_lo = lo;
        clip_hi = hi;
        if (bad) any_nan = 1;
    }

    WDX_STAMP(2);
    // ---- A2: optional signal normalisation (sig_proc.py:433-446); "none" in every shipped config ----
    if (n > 0 && P.sig_norm == WDX_NORM_MEDIAN) {
        const float shift = block_nanmedian_f32<BLOCK>([&](int i) { return sig[i]; }, n, hist, sh);
        const float scale =
            block_nanmedian_f32<BLOCK>([&](int i) { return fabsf(sig[i] - shift); }, n, hist, sh);
        __syncthreads();
        for (int i = tid; i < n; i += BLOCK) sig[i] = (sig[i] - shift) / scale;
        __syncthreads();
    } else if (n > 0 && P.sig_norm == WDX_NORM_MEAN) {
        // mean_normalize on the float32 signal (sig_proc.py:99-111, accept_nan=True): np.mean / np.std, or
        // np.nanmean / np.nanstd when the window holds a NaN -- float32 sums in NumPy's association
        // (oracle: mean_normalize_f32).  LDS scratch: the score curve and the event arrays are still unused.
        float *sq = reinterpret_cast<float *>(scores);   // n floats
        float *lsum = reinterpret_cast<float *>(ev);       // leaf sums
        int *stk = reinterpret_cast<int *>(zz);            // traversal stack
        unsigned vcnt = 0;
        for (int i = tid; i < n; i += BLOCK) vcnt += (sig[i] == sig[i]);
        unsigned gmin_, gmax_, cnt;
        block_minmax_count<BLOCK>(0u, 0u, vcnt, sh, gmin_, gmax_, cnt);
        const bool has_nan = cnt != (unsigned)n;
        float shift, scale;
        if (!has_nan) {
            shift = block_np_add_reduce_f32<BLOCK>([&](int i) { return sig[i]; }, n, hist, lsum, stk, sh) / (float)n;
            for (int i = tid; i < n; i += BLOCK) {
                const float d = sig[i] - shift;
                sq[i] = d * d;
            }
            scale = sqrtf(block_np_add_reduce_f32<BLOCK>([&](int i) { return sq[i]; }, n, hist, lsum, stk, sh) / (float)n);
        } else {
            // _replace_nan(a, 0); sums in float32; _divide_by_count forms the quotient in float64 (float32 / intp)
            const float tot = block_np_add_reduce_f32<BLOCK>(
                [&](int i) { const float v = sig[i]; return v == v ? v : 0.0f; }, n, hist, lsum, stk, sh);
            shift = (float)((double)tot / (double)cnt);
            for (int i = tid; i < n; i += BLOCK) {
                const float v = sig[i];
                const float d = (v == v) ? v - shift : 0.0f;
                sq[i] = d * d;
            }
            const float var = (float)((double)block_np_add_reduce_f32<BLOCK>([&](int i) { return sq[i]; }, n, hist,
                                                                             lsum, stk, sh) / (double)cnt);
            scale = sqrtf(var);
        }
        __syncthreads();
        for (int i = tid; i < n; i += BLOCK) sig[i] = (sig[i] - shift) / scale;
        __syncthreads();
    } else if (n > 0 && P.sig_norm != WDX_NORM_NONE) {
        finish(WDX_READ_FAIL_SIGNORM);  // unknown codes fail here
        return;
    }

    // ---- parameter shrink (sig_proc.py:526-533), Python round() == rint() ---------------------------
    int d_eff = (int)rint((double)n / (double)E / 2.0);
    if (P.min_obs_per_base < d_eff) d_eff = P.min_obs_per_base;
    int W = (int)rint((double)n / (double)E);
    if (P.running_stat_width < W) W = P.running_stat_width;

    // ---- P2: windowed t-statistic (_c_segmentation.pyx:124-161) -------------------------------------
    int ns = n - 2 * W;
    if (ns < 0 || (ns > 0 && W == 0)) ns = 0;  // the Cython call raises -> zeros(0)
    if (d_eff < 1) {                            // scipy: `distance` must be >= 1 -> "unknown"
        finish(WDX_READ_FAIL_UNKNOWN);
        return;
    }
    {
        const double Wd = (double)W;
        const int nq = n - W + 1;  // window starts
        for (int t0 = 0; t0 < ns; t0 += kTile) {
            const int qend = min(t0 + kTile + W, nq);
            for (int q = t0 + tid; q < qend; q += BLOCK) {
                double m = 0.0;
                for (int k = 0; k < W; ++k) m += (double)sig[q + k];
                m /= Wd;
                double v = 0.0;
                for (int k = 0; k < W; ++k) {
                    const double df = (double)sig[q + k] - m;
                    v += df * df;
                }
                Mt[q - t0] = m;
                Vt[q - t0] = v;
            }
            __syncthreads();
            const int pend = min(t0 + kTile, ns);
            for (int pos = t0 + tid; pos < pend; pos += BLOCK) {
                const double m1 = Mt[pos - t0], m2 = Mt[pos - t0 + W];
                const double vs = Vt[pos - t0] + Vt[pos - t0 + W];
                double s;
                if (vs == 0) s = 0.0;
                else if (m1 > m2) s = (m1 - m2) / sqrt(vs);
                else s = (m2 - m1) / sqrt(vs);
                scores[pos] = s;
            }
            __syncthreads();
        }
    }

    WDX_STAMP(3);
    // ---- P3-P6: find_peaks + top-E + boundaries + event means (fp_segment) ----------------------------
    int nms_iters = 0, nseg = 0;
    {
        const int st = fp_segment<BLOCK>(scores, state, ns, d_eff, W, E, P.accept_less_cpts != 0, sig, n, cpts, ev,
                                         hist, sh, nseg, nms_iters, A.no_list != 0);
        if (st != WDX_READ_OK) {
            finish(st);
            return;
        }
    }
    WDX_STAMP(7);
    if (A.rf.query) {
        // The refinement kernels behind the fast kernels (one wave per three reads for the match, a quarter of this
        // workgroup for the barcode's segmentation) take this read too when they can reproduce its clipped samples from
        // the recorded bounds: no NaN in the window, samples as loaded, the configured window width.
        if (A.refine_record && A.rf.ws && P.sig_norm == WDX_NORM_NONE && W == P.running_stat_width && nseg <= 128 &&
            !__syncthreads_or(any_nan)) {
            RefineRec *rec = reinterpret_cast<RefineRec *>(A.rf.ws) + r;
            for (int s = tid; s < nseg; s += BLOCK) rec->ev[s] = ev[s];
            for (int s = tid; s <= nseg; s += BLOCK) rec->cpts[s] = cpts[s];
            if (tid == 0) {
                rec->n = n;
                rec->lo = clip_lo;
                rec->hi = clip_hi;
                rec->state = 1;
            }
            return;
        }
        fp_refine_tail<BLOCK>(A, r, state, ns, W, n, cpts, ev, zz, tmp, reinterpret_cast<unsigned char *>(Mt), hist, sh, nseg,
                              [&](int sbs, int, const double *&sc_t, const float *&sg_t) {
                                  sc_t = scores + sbs;   // the adapter pass's score curve and clipped samples are still here
                                  sg_t = sig + sbs;
                                  return true;
                              });
        return;
    }

    WDX_STAMP(8);
    // ---- P7: normalise, stats, tail (sig_proc.py:546-605) --------------------------------------------
    {
        int has_nan = 0;
        for (int s = tid; s < nseg; s += BLOCK) has_nan |= (ev[s] != ev[s]);
        if (__syncthreads_or(has_nan)) {
            finish(WDX_READ_FAIL_SEGNORM);  // normalize(..., accept_nan=False) raises
            return;
        }
        // np.mean / np.std of the event means (also adapter_event_mean / adapter_event_std)
        if (tid == 0) sh.mean = np_pairwise_sum_dev(ev, nseg) / (double)nseg;
        __syncthreads();
        const double mean = sh.mean;
        for (int s = tid; s < nseg; s += BLOCK) {
            const double df = ev[s] - mean;
            tmp[s] = df * df;
        }
        __syncthreads();
        if (tid == 0) sh.sd = sqrt(np_pairwise_sum_dev(tmp, nseg) / (double)nseg);
        __syncthreads();
        const double sd = sh.sd;

        if (P.seg_norm == WDX_NORM_MEAN) {
            for (int s = tid; s < nseg; s += BLOCK) zz[s] = (ev[s] - mean) / sd;
        } else if (P.seg_norm == WDX_NORM_MEDIAN) {
            block_small_median<BLOCK>(ev, nseg, sh, 4);
            const double m = sh.stat[4];
            for (int s = tid; s < nseg; s += BLOCK) tmp[s] = fabs(ev[s] - m);
            block_small_median<BLOCK>(tmp, nseg, sh, 5);
            const double sc = sh.stat[5];
            for (int s = tid; s < nseg; s += BLOCK) zz[s] = (ev[s] - m) / sc;
        } else if (P.seg_norm == WDX_NORM_NONE) {
            for (int s = tid; s < nseg; s += BLOCK) zz[s] = ev[s];
        } else {
            finish(WDX_READ_FAIL_SEGNORM);
            return;
        }
        __syncthreads();

        // stats (sig_proc.py:562-567) -- only when the caller takes them (four medians by rank counting)
        if (A.stats) {
            for (int s = tid; s < nseg; s += BLOCK) tmp[s] = (double)(cpts[s + 1] - cpts[s]);
            block_small_median<BLOCK>(tmp, nseg, sh, 0);
            const double dt_med = sh.stat[0];
            for (int s = tid; s < nseg; s += BLOCK) tmp[s] = fabs(tmp[s] - dt_med);
            block_small_median<BLOCK>(tmp, nseg, sh, 1);
            block_small_median<BLOCK>(ev, nseg, sh, 4);
            const double ev_med = sh.stat[4];
            for (int s = tid; s < nseg; s += BLOCK) tmp[s] = fabs(ev[s] - ev_med);
            block_small_median<BLOCK>(tmp, nseg, sh, 5);
        }

        if (nseg < K) {
            finish(WDX_READ_FAIL_UNKNOWN);  // np.pad(int64 dwell, NaN) raises in the reference
            return;
        }
        for (int i = tid; i < K; i += BLOCK) {
            const int s = nseg - K + i;
            if (A.fpt) A.fpt[r * K + i] = zz[s];
            if (A.dwell) A.dwell[r * K + i] = (int64_t)(cpts[s + 1] - cpts[s]);
        }
        if (A.stats && tid == 0) {
            double *o = A.stats + r * 6;
            o[0] = sh.stat[0];
            o[1] = sh.stat[1];
            o[2] = mean;
            o[3] = sd;
            o[4] = sh.stat[4];
            o[5] = sh.stat[5];
        }
    }
    WDX_STAMP(9);
    if (PROF && tid == 0 && r < A.prof_reads) {
        A.prof[r * 32 + 10] = nms_iters;
        A.prof[r * 32 + 11] = n;
        A.prof[r * 32 + 12] = ns;
    }
    finish(status);
}

// one workgroup per read
template <int BLOCK, bool PROF>
__global__ __launch_bounds__(BLOCK) void fingerprint_kernel(FpArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    fp_process_read<BLOCK, PROF>(A, A.block_base + blockIdx.x, smem);
}

// exact slow path: the reads the fast kernel declined (list built with atomics), grid-stride
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void fingerprint_list_kernel(FpArgs A, const unsigned *count,
                                                                 const int32_t *list) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned n = *count;
    for (unsigned k = blockIdx.x; k < n; k += gridDim.x) {
        __syncthreads();
        fp_process_read<BLOCK, false>(A, (int64_t)list[k], smem);
    }
}

// windows beyond the LDS capacity of the exact kernel (kExactLdsCap < n <= kBigCap): a fixed grid of <= kBigSlots
// workgroups strides over the slow list (or over all reads when there is none) and takes only those
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void fingerprint_big_kernel(FpArgs A, const unsigned *count, const int32_t *list,
                                                                int small_cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int64_t n = count ? (int64_t)*count : A.n_reads;
    for (int64_t k = blockIdx.x; k < n; k += gridDim.x) {
        const int64_t r = list ? (int64_t)list[k] : k;
        if (A.ok && !A.ok[r]) continue;  // the regular kernel reported it
        const int64_t row_len = A.row_len ? (int64_t)A.row_len[r]
                                          : (A.row_off ? A.row_off[r + 1] - A.row_off[r] : A.stride);
        int64_t start = (int64_t)A.a_start[r] - A.p.padding;
        if (start < 0) start = 0;
        int64_t stop = (int64_t)A.a_end[r] + A.p.padding;
        if (stop > row_len) stop = row_len;
        if (stop - start <= small_cap || stop - start > kBigCap) continue;
        __syncthreads();
        fp_process_read<BLOCK, false, true>(A, r, smem);
    }
}

// The refinement branch behind the FAST kernels (round 3; the match as a kernel of wave-sized workgroups and the exact
// kernel's hand-over since round 4): a fast kernel -- or the exact kernel, for a read of the slow list it can hand over --
// segments the adapter and leaves a RefineRec; two kernels do everything after it, bit for bit what fp_refine_match /
// fp_refine_finish (the exact kernel's own code) do:
//   fingerprint_refine_match_wave_kernel  (wdx_refine_match.hip) adapter statistics, subsequence DP, back-trace: one
//                                    wave per three reads;
//   fingerprint_refine_tail_kernel   the barcode's own segmentation: loads ONLY the samples from sig_barcode_start on,
//                                    re-clips them with the recorded bounds (the same v_med3_f32 on the same values),
//                                    computes their t-scores with the exact kernel's operations (every window's
//                                    statistics once, tiles of 256 window starts) and segments them (fp_segment).
// A barcode tail beyond kTailCap samples goes back to the exact kernel, which then refines the read in place.
// LDS of fingerprint_refine_tail_kernel<256>: FpShared | cpts | U | scores | samples, where U holds the event means, the
// select's histogram and the peak-state bytes of the segmentation and, before it, the window statistics of one t-score tile
constexpr int kTailBlock = 256;
constexpr size_t kTailU = ((size_t)kSegCap * 8 + 256 * 4 + kTailCap) > (size_t)kTailBlock * 16
                              ? ((size_t)kSegCap * 8 + 256 * 4 + kTailCap) : (size_t)kTailBlock * 16;
static_assert(kTailU % 16 == 0 && (kSegCap * 8) % 16 == 0, "alignment of the regions inside U");
static size_t refine_tail_lds_bytes() {
    size_t b = sizeof(FpShared) + (size_t)(kSegCap + 1) * 4;
    b = (b + 15) & ~(size_t)15;
    b += kTailU + (size_t)kTailCap * 8 + (size_t)(kTailCap + 64) * 4;
    return (b + 15) & ~(size_t)15;
}
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void fingerprint_refine_tail_kernel(FpArgs A, unsigned *slow_count, int32_t *slow_list) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int64_t r = A.block_base + blockIdx.x;
    if (r >= A.n_reads) return;
    RefineRec *rec = reinterpret_cast<RefineRec *>(A.rf.ws) + r;
    if (rec->state != 3) return;
    const wdx_seg_params &P = A.p;
    static_assert(BLOCK == kTailBlock, "refine_tail_lds_bytes");
    FpShared &sh = *reinterpret_cast<FpShared *>(smem);
    int *cpts = reinterpret_cast<int *>(&sh + 1);
    unsigned char *U = reinterpret_cast<unsigned char *>((reinterpret_cast<uintptr_t>(cpts + kSegCap + 1) + 15) & ~(uintptr_t)15);
    double *zz = reinterpret_cast<double *>(U);                                  // kSegCap
    unsigned *hist = reinterpret_cast<unsigned *>(zz + kSegCap);                 // 256
    unsigned char *state = reinterpret_cast<unsigned char *>(hist + 256);        // kTailCap
    double *Mt = reinterpret_cast<double *>(U), *Vt = Mt + BLOCK;                // one tile's window statistics (before these)
    double *t_scores = reinterpret_cast<double *>(U + kTailU);                   // kTailCap
    float *t_sig = reinterpret_cast<float *>(t_scores + kTailCap);               // kTailCap + 64
    RefineMatch M;
    memcpy(&M, rec->m, sizeof(M));
    const int n = rec->n, W = P.running_stat_width;
    const float lo = rec->lo, hi = rec->hi;
    const int64_t row_off = A.row_off ? A.row_off[r] : r * A.stride;
    int64_t start = (int64_t)A.a_start[r] - P.padding;
    if (start < 0) start = 0;
    const float *__restrict__ src = A.sig + row_off + start;
    const int ns = n - 2 * W;
    fp_refine_finish<BLOCK>(A, r, M, state, ns, n, cpts, zz, hist, sh,
                            [&](int sbs, int ns2, const double *&sc_t, const float *&sg_t) -> bool {
        const int nt = n - sbs;  // samples of the barcode tail
        if (nt > kTailCap || nt < 0) {  // block-uniform: hand the whole read to the exact kernel
            if (tid == 0) {
                rec->state = 2;
                slow_list[atomicAdd(slow_count, 1u)] = (int32_t)r;
            }
            return false;
        }
        for (int i = tid; i < nt; i += BLOCK) t_sig[i] = __builtin_amdgcn_fmed3f(src[sbs + i], lo, hi);
        __syncthreads();
        // windowed t-statistic of the tail (_c_segmentation.pyx:124-161; fp_process_read's operations): tiles of BLOCK
        // window starts -- every window's mean and squared deviations once, one thread each -- then the scores of the
        // BLOCK - W positions whose two windows the tile holds
        const int nwin = ns2 > 0 ? ns2 + W : 0;   // window starts 0 .. ns2 + W - 1 (the last one ends at sample nt - 1)
        const int TP = BLOCK - W;                 // W <= kMaxW = 64 < BLOCK
        for (int t0 = 0; t0 < ns2; t0 += TP) {
            const int q = t0 + tid;
            if (q < nwin) {
                double m, v;
                if (W == 18) window_stats<18>(t_sig + q, W, m, v);
                else if (W == 12) window_stats<12>(t_sig + q, W, m, v);
                else window_stats<0>(t_sig + q, W, m, v);
                Mt[tid] = m;
                Vt[tid] = v;
            }
            __syncthreads();
            const int pos = t0 + tid;
            if (tid < TP && pos < ns2) {
                const double m1 = Mt[tid], m2 = Mt[tid + W];
                const double vs = Vt[tid] + Vt[tid + W];
                double sc;
                if (vs == 0) sc = 0.0;
                else if (m1 > m2) sc = (m1 - m2) / sqrt(vs);
                else sc = (m2 - m1) / sqrt(vs);
                t_scores[pos] = sc;
            }
            __syncthreads();
        }
        sc_t = t_scores;
        sg_t = t_sig;
        return true;
    }, lo > 0.f && hi <= lo * 65536.f);   // (clipped samples in [lo, hi], at most kTailCap of them: their float64 sums are exact)
}

#ifndef WDX_DEV_KERNELS_ONLY  // (development: a TU that instantiates single kernels includes this file with the macro set)
static int launch_refine_tail(FpArgs A, unsigned *slow_count, int32_t *slow_list, hipStream_t stream) {
    static LdsAttr attr_t;
    const size_t lds_t = refine_tail_lds_bytes();
    if (int rc = attr_t.ensure(fingerprint_refine_tail_kernel<256>, lds_t)) return rc;
    const int64_t slice = launch_slice_limit(1 << 22);
    for (int64_t base = 0; base < A.n_reads; base += slice) {
        const int64_t n = A.n_reads - base < slice ? A.n_reads - base : slice;
        A.block_base = base;
        if (int rc = launch_refine_match_wave(A, n, stream)) return rc;
        if (refine_tail_wave_takes(A)) {
            if (int rc = launch_refine_tail_wave(A, n, slow_count, slow_list, stream)) return rc;
        } else {
            hipLaunchKernelGGL((fingerprint_refine_tail_kernel<256>), dim3((unsigned)n), dim3(256), lds_t, stream, A, slow_count,
                               slow_list);
        }
    }
    WDX_HIP_TRY(hipGetLastError());
    return WDX_SUCCESS;
}
int64_t fingerprint_refine_ws_bytes(int64_t n_reads) { return (int64_t)sizeof(RefineRec) * (n_reads > 0 ? n_reads : 0); }

#endif  // WDX_DEV_KERNELS_ONLY
#include "wdx_fingerprint_fast.inc"
#include "wdx_fingerprint_split.inc"
#ifndef WDX_DEV_KERNELS_ONLY

static size_t fp_lds_bytes(int cap) {
    size_t b = 0;
    b += (size_t)cap * 8;                   // scores
    b += (size_t)(kTile + kMaxW) * 8 * 2;   // Mt, Vt (+ state when it fits)
    b += (size_t)kSegCap * 8 * 3;           // ev, zz, tmp
    b += sizeof(FpShared);
    b += (size_t)cap * 4;                   // sig
    b += (size_t)256 * 4;                   // hist
    b += (size_t)(kSegCap + 1) * 4;         // cpts
    if (cap > (int)((kTile + kMaxW) * 16)) b += (size_t)cap;  // dedicated state
    return (b + 15) & ~(size_t)15;
}

static size_t fp_lds_bytes_big(int cap) { return fp_lds_bytes(cap) - (size_t)cap * 8; }  // no score curve in LDS

int64_t fingerprint_big_bytes(int64_t max_len) {
    return max_len > kExactLdsCap ? (int64_t)kBigSlots * kBigCap * 8 : 0;
}

static int launch_fp_big(FpArgs A, int small_cap, const unsigned *count, const int32_t *list, hipStream_t stream) {
    static LdsAttr attr;
    const size_t lds = fp_lds_bytes_big(kBigCap);
    if (int rc = attr.ensure(fingerprint_big_kernel<1024>, lds)) return rc;
    A.cap = kBigCap;
    A.defer_big = 0;
    const int64_t grid = A.n_reads < kBigSlots ? A.n_reads : kBigSlots;
    hipLaunchKernelGGL((fingerprint_big_kernel<1024>), dim3((unsigned)grid), dim3(1024), lds, stream, A, count, list,
                       small_cap);
    WDX_HIP_TRY(hipGetLastError());
    return WDX_SUCCESS;
}

template <int BLOCK, bool PROF>
static int launch_fp_chunks(FpArgs A, size_t lds, hipStream_t stream, int64_t *n_launches) {
    static LdsAttr attr;
    if (int rc = attr.ensure(fingerprint_kernel<BLOCK, PROF>, lds)) return rc;
    // HIP drops work when grid.x * block.x reaches 2^32: launch in slices of 2^21 reads
    const int64_t slice = launch_slice_limit(1 << 21);
    for (int64_t base = 0; base < A.n_reads; base += slice) {
        const int64_t n = A.n_reads - base < slice ? A.n_reads - base : slice;
        A.block_base = base;
        hipLaunchKernelGGL((fingerprint_kernel<BLOCK, PROF>), dim3((unsigned)n), dim3(BLOCK), lds,
                           stream, A);
        if (n_launches) ++*n_launches;
    }
    WDX_HIP_TRY(hipGetLastError());
    return WDX_SUCCESS;
}

template <int BLOCK>
static int launch_fp_list(const FpArgs &A, size_t lds, const unsigned *count, const int32_t *list,
                          hipStream_t stream, int max_grid = 2048) {
    static LdsAttr attr;
    if (int rc = attr.ensure(fingerprint_list_kernel<BLOCK>, lds)) return rc;
    const int64_t grid = A.n_reads < max_grid ? A.n_reads : max_grid;
    hipLaunchKernelGGL((fingerprint_list_kernel<BLOCK>), dim3((unsigned)grid), dim3(BLOCK), lds, stream,
                       A, count, list);
    WDX_HIP_TRY(hipGetLastError());
    return WDX_SUCCESS;
}

int fill_refine_dev(const wdx_refine_params &rp, const double *d_query, int32_t *d_idx, RefineDev **out) {
    if (rp.subseq_norm != WDX_NORM_NONE && rp.subseq_norm != WDX_NORM_MEAN && rp.subseq_norm != WDX_NORM_MEDIAN) {
        set_error("Normalization method %d not recognized.", (int)rp.subseq_norm);
        return WDX_ERR_INVALID;
    }
    if (rp.penalty != rp.penalty || rp.penalty < 0) {
        set_error("consensus refinement: penalty must be >= 0");
        return WDX_ERR_INVALID;
    }
    RefineDev *r = new RefineDev{d_query, rp.n_query, rp.subseq_norm, rp.penalty, rp.psi[0], rp.psi[2],
                                 rp.ub_start, rp.lb_end, rp.ub_end, rp.barcode_segm_events, d_idx, nullptr};
    *out = r;
    return WDX_SUCCESS;
}
void free_refine_dev(RefineDev *rf) { delete rf; }
void set_refine_ws(RefineDev *rf, void *d_ws) { rf->ws = reinterpret_cast<unsigned char *>(d_ws); }

// Self-test of the t-score's unscaled sqrt / quotient sequences (fast_sqrt_mid, fast_div_mid) against the
// compiler's general float64 sqrt() and '/': same bits on the documented value range is what the fast path's
// bit-identity rests on; a toolchain whose expansions change shows up here (tests/test_gpu_parity.py).
__global__ void score_selftest_kernel(const double *__restrict__ dm, const double *__restrict__ vs, int64_t n,
                                      double *__restrict__ fast, double *__restrict__ ref) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fast[i] = fast_div_mid(dm[i], fast_sqrt_mid(vs[i]));
    ref[i] = dm[i] / sqrt(vs[i]);
}
int launch_score_selftest(const double *dm, const double *vs, int64_t n, double *fast, double *ref, hipStream_t stream) {
    if (n <= 0) return WDX_SUCCESS;
    hipLaunchKernelGGL(score_selftest_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, dm, vs, n, fast, ref);
    WDX_HIP_TRY(hipGetLastError());
    return WDX_SUCCESS;
}

// eight counters (five used) | one ClipRec per read | five read lists (slow, big0, big1, retry, big2: see launch_fingerprint)
// | large batches: the split main kernel's peak lists for one launch slice (kSplitRecBytes per read, 16-byte aligned)
// | 16 diagnostic counters (WDX_OPT_DEBUG_OCCUPANCY: why reads were handed to the exact kernel) at the very end -- in the
// caller's workspace, i.e. per context and per call, ordered by the call's stream
static int64_t split_ws_offset(int64_t n_reads) { return (32 + 40 * (n_reads > 0 ? n_reads : 0) + 15) / 16 * 16; }
static int64_t dbg_ws_offset(int64_t n_reads) {
    // (the lists of one slice for every batch size: 4.6 KB per read -- the launch chain is for batches of 2048 reads and
    // more by default, but WDX_OPT_FAST_CHAIN_MIN_READS sends smaller ones through it: the randomised parameter tests)
    return split_ws_offset(n_reads) + (int64_t)kSplitRecBytes * std::min<int64_t>(n_reads > 0 ? n_reads : 0, kSplitSlice);
}
int64_t fingerprint_workspace_bytes(int64_t n_reads) { return dbg_ws_offset(n_reads) + 64; }

// The fast kernels exist for three (window width, suppression reach) combinations -- the shipped parameter triples:
//   1: W = 12, d <= 9  (RNA004: 110, 6, 12)      every instantiation of the launch chain
//   2: W = 18, d <= 9  (tRNA triple: 120, 9, 18; the tRNA config itself also refines -> exact kernel)
//   3: W = 30, d <= 17 (RNA002 triple: 110, 15, 30)
// 2 runs the 5120-sample instantiation as main kernel for large batches (then 6144 and 8192 behind it, like 1); 3 the
// 6144-sample one, and the 8192-sample one for longer windows and retries.
//   4 .. 6: W = 6 / 24 / 36, d <= 17 (round 5: the other multiples of the tile's six positions per lane -- configurations
//           nobody ships, but `--export segmentation.running_stat_width=...` is one flag away, and the exact kernel
//           behind this gate runs at a fifteenth of the rate); instantiated like 3 (NBT = 2, 6144-sample main kernel)
//   7, 8: W = 12 / 18 with 9 < d <= 17: the same, for the shipped widths with a longer suppression reach.
static int fast_combo(const wdx_seg_params &p) {
    if (p.min_obs_per_base < 1) return 0;
    if (p.running_stat_width == 12 && p.min_obs_per_base <= 9) return 1;
    if (p.running_stat_width == 18 && p.min_obs_per_base <= 9) return 2;
    if (p.running_stat_width == 30 && p.min_obs_per_base <= 17) return 3;
    if (p.running_stat_width == 6 && p.min_obs_per_base <= 17) return 4;
    if (p.running_stat_width == 24 && p.min_obs_per_base <= 17) return 5;
    if (p.running_stat_width == 36 && p.min_obs_per_base <= 17) return 6;
    if (p.running_stat_width == 12 && p.min_obs_per_base <= 17) return 7;   // (9 < d <= 17: the NBT = 2 form of the width)
    if (p.running_stat_width == 18 && p.min_obs_per_base <= 17) return 8;
    // 9 .. 37 (round 6): every other width from 7 to 35, d <= 17 -- widths that are not multiples of six, on the fast kernels'
    // EXACT-scores pass (the partner window's statistics come from another slot of a lane one further); no approximate keys,
    // no retry launch.  combo = 9 + (W - 7)
    if (p.min_obs_per_base <= 17 && p.running_stat_width >= 7 && p.running_stat_width <= 35 && p.running_stat_width % 6 != 0)
        return 9 + (p.running_stat_width - 7);
    return 0;
}
static bool fast_combo_exact_only(int combo) { return combo >= 9; }

int launch_fingerprint(const float *d_sig, const int64_t *d_row_off, const int32_t *d_row_len,
                       int64_t stride, int64_t max_len, int64_t n_reads, const int32_t *d_a_start,
                       const int32_t *d_a_end, const uint8_t *d_ok, const wdx_seg_params &p,
                       double *d_fpt, int64_t *d_dwell, double *d_stats, int32_t *d_status,
                       hipStream_t stream, void *d_ws, const Knobs &knobs, int64_t *n_launches,
                       long long *d_prof, int64_t prof_reads, int stop_phase, const RefineDev *rf,
                       MainEvents *main_ev, double *d_big) {
    if (n_reads == 0) return WDX_SUCCESS;
    if (n_reads > 0x7fffffffLL) {
        set_error("at most 2^31-1 reads per call");
        return WDX_ERR_INVALID;
    }
    LaunchSliceScope slice_scope(knobs.max_launch_slice);
    if (p.num_events < 1 || p.num_events > kMaxEvents) {
        set_error("num_events must be in [1, %d]", kMaxEvents);
        return p.num_events < 1 ? WDX_ERR_INVALID : WDX_ERR_UNSUPPORTED;
    }
    if (p.barcode_num_events < 1 || p.barcode_num_events > kSegCap) {
        set_error("barcode_num_events must be in [1, %d]", kSegCap);
        return WDX_ERR_INVALID;
    }
    if (p.running_stat_width < 0 || p.running_stat_width > kMaxW) {
        set_error("running_stat_width must be in [0, %d]", kMaxW);
        return WDX_ERR_UNSUPPORTED;
    }
    if (p.padding < 0) {
        set_error("padding must be >= 0");
        return WDX_ERR_INVALID;
    }
    int64_t cap64 = max_len;
    if (cap64 > kExactLdsCap) cap64 = kExactLdsCap;
    if (cap64 < 64) cap64 = 64;
    int cap = (int)((cap64 + 63) / 64 * 64);
    // windows of kExactLdsCap+1 .. kBigCap samples: left alone by every launch below and taken by
    // fingerprint_big_kernel at the end (needs the caller's fingerprint_big_bytes(max_len) buffer; without it they
    // are reported WDX_READ_FAIL_UNKNOWN as windows beyond WDX_MAX_ADAPTER_SAMPLES always are)
    const bool with_huge = max_len > kExactLdsCap && d_big != nullptr && !d_prof;
    FpArgs A{d_sig, d_row_off, d_row_len, stride, n_reads, d_a_start, d_a_end, d_ok,
             p,     d_fpt,     d_dwell,   d_stats, d_status, cap, 0, d_prof, prof_reads, stop_phase, 1, RefineDev{},
             d_big, with_huge ? 1 : 0, knobs.exact_no_list ? 1 : 0};
    {
        const uint64_t e1 = (uint64_t)(p.num_events > 0 ? p.num_events : 1);
        A.e_magic1 = (unsigned)std::min<uint64_t>(((1ull << 32) + e1 - 1) / e1, 0xffffffffull);   // (E = 1: 2^32 - 1 -> q = n - 1, rounded up to n)
        A.e_magic2 = (unsigned)(((1ull << 32) + 2 * e1 - 1) / (2 * e1));
    }
    if (rf) {
        if (!rf->query || rf->nq < 1 || rf->nq > kRefineMaxQuery || p.num_events + 1 > kRefineMaxSeries) {
            set_error("consensus refinement: the query must have 1..%d points and num_events + 1 <= %d", kRefineMaxQuery,
                      kRefineMaxSeries);
            return WDX_ERR_UNSUPPORTED;
        }
        if (rf->E2 < 1 || rf->E2 > kMaxEvents || rf->psi1b < 0 || rf->psi2b < 0) {
            set_error("consensus refinement: barcode_num_events[0] must be in [1, %d], psi >= 0", kMaxEvents);
            return WDX_ERR_INVALID;
        }
        A.rf = *rf;
    }
    const size_t lds = fp_lds_bytes(cap);
    if (lds > 160 * 1024) {
        set_error("fingerprint LDS carve-up (%zu B) exceeds 160 KiB", lds);
        return WDX_ERR_INVALID;
    }
    (void)hipGetLastError();  // do not inherit a stale error from an earlier failed call
    const bool small = lds <= 80 * 1024;
    if (d_prof && !d_ws) return small ? launch_fp_chunks<512, true>(A, lds, stream, n_launches)
                                      : launch_fp_chunks<1024, true>(A, lds, stream, n_launches);

    // fast path for the common case + exact slow path for whatever it declines
    // (the fast kernels take reads whose EFFECTIVE parameters are window width 12 and distance <= 9; with a configured
    // width other than 12 or a configured distance beyond 9 only a few very short reads would qualify -- sig_proc.py:
    // 526-533 shrinks the parameters for those -- and a launch chain whose main kernel declines nearly every read
    // costs more than it saves: 1.62 against 1.99 M reads/s on the RNA002 triple (110, 15, 30))
    const bool fast_ok = d_ws && p.sig_norm == WDX_NORM_NONE &&   // (accept_less_cpts: the fast kernels hand over the reads it concerns)
                         p.num_events <= kFSeg - 2 && p.barcode_num_events <= p.num_events + 1 &&
                         fast_combo(p) != 0 && (fast_combo(p) == 1 || !d_prof) && cap >= 512 && !knobs.exact_path &&
                         (!rf || (rf->ws && !d_prof && p.num_events + 1 <= 128));
    if (fast_ok) {
        // A chain of launches, each handing what it cannot take to the next through device-side lists:
        //   main    one workgroup per read; the instantiation follows the longest adapter window of the batch:
        //           4096 or (large batches) 5120 samples at FIVE workgroups per CU, else 6144 at four.  A fifth
        //           resident workgroup is worth 1.16x (tools/probes/occupancy_probe.py); 89 % of RNA004 adapter
        //           windows fit 5120 samples
        //   big0    windows (and peak lists) beyond the main instantiation -> 6144-sample list kernel
        //   big1    beyond that -> 8192-sample list kernel (three workgroups per CU)
        //   retry   reads whose approximate score keys left an order decision inside the error band (about 2 in
        //           1000) -> 8192-sample list kernel with exact scores
        //   slow    everything else (NaNs, other window widths, long plateaus, ...) -> the exact general kernel
        // Small batches (live ticks) skip the approximate keys, and with them the retry launch, and go straight
        // to the 6144-sample instantiation: there a launch costs more than the arithmetic saved.
        // Peak-list capacities: local maxima of the score curve run at ~N/5.6 (>= N/5.0 observed); the capacities
        // leave headroom within the LDS budget of the instantiation's occupancy; overflows move up the chain.
        const int64_t chain_min = knobs.fast_chain_min > 0 ? knobs.fast_chain_min : 2048;
        const bool large_batch = n_reads >= chain_min;
        const int combo = fast_combo(p);
        const bool approx = large_batch && !knobs.fast_exact_scores && !fast_combo_exact_only(combo);
        const int nbt = combo >= 3 ? 2 : 1;
        int capF = cap <= 4096 ? 4096 : (large_batch ? 5120 : 6144);
        if (knobs.fast_main_cap == 5120 || knobs.fast_main_cap == 6144) capF = knobs.fast_main_cap;  // experiments
        if (combo == 2) capF = large_batch ? 5120 : 6144;  // width 18: five workgroups per CU too (91 VGPRs)
        if (combo >= 3) capF = 6144;  // width 30 / reach 17 (and the round-5 widths): 115 VGPRs, four waves per SIMD either way
        // (LDS is allocated in 1280-byte granules: five workgroups per CU need <= 32 000 B each, four <= 40 960 B --
        // hipOccupancyMaxActiveBlocksPerMultiprocessor does not know and reports five at 32 640 B)
        // Large batches: the clip bounds of the MAIN kernel's reads are computed ahead of it by clip_bounds_kernel (one wave
        // per read, wdx_clip.hip) and the main kernel starts at the clip (EXT instantiation).  Small batches (live ticks:
        // every launch counts) keep the in-kernel radix selects.
        const bool ext = large_batch || capF == 5120;
        int capP = capF == 4096 ? 1152 : (capF == 5120 ? 980 : 1376);
        // with the threshold filter of the appends (kPeakTauLo) a read lists ~340 peaks instead of ~830: 512 entries leave the
        // 5120-sample main kernel at 26.8 KB of LDS -- six workgroups per CU (a longer list moves on to the list kernels).
        // (the width-30 instantiations, NBT = 2, are bound by their registers at four: they keep the long lists, except
        // the streaming form)
        const bool filt = approx && !knobs.no_peak_filter;   // (the launches on approximate keys below)
        if (ext && filt && nbt == 1 && capF <= 5120) capP = 512;
        // the NBT = 2 widths' 6144-sample main kernel at FIVE workgroups per CU: 96 VGPRs (launch bound) and the filtered
        // 512-entry list -- 31 000 B of LDS (five need <= 32 000); longer lists move on to the list kernel
        // (not width 6: its reach-3 lists are the longest -- 512 entries overflow for most reads, 23.3 -> 22.8 M reads/s)
        if (ext && filt && nbt == 2 && capF == 6144 && kWideWgPerCu == 5 && p.running_stat_width >= 12) capP = 512;
        if (knobs.fast_peak_cap > 0) capP = knobs.fast_peak_cap;  // experiment knob (wdx_ctx_set_option)
        const size_t flds = fast_lds_bytes(capF, capP, nbt);
        unsigned *count = reinterpret_cast<unsigned *>(d_ws);  // [0] slow, [1] big0, [2] big1, [3] retry, [4] big2, [5] back
        ClipRec *clip = reinterpret_cast<ClipRec *>(reinterpret_cast<unsigned char *>(d_ws) + 32);
        int32_t *list = reinterpret_cast<int32_t *>(reinterpret_cast<unsigned char *>(d_ws) + 32 + 16 * n_reads);
        int32_t *big0 = list + n_reads, *big1 = big0 + n_reads, *retry = big1 + n_reads, *big2 = retry + n_reads;
        WDX_HIP_TRY(hipMemsetAsync(count, 0, 32, stream));
        const bool chain = !(d_prof && stop_phase > 0);         // (ablation timing: the main kernel alone)
        const bool with_big0 = chain && capF == 5120;            // windows of 5121..6144 samples, peak-list overflows
        // windows beyond 6144 samples, up to WDX_MAX_ADAPTER_SAMPLES: the streaming fast kernel (at 8 000 samples it is
        // faster than the striding 8192-sample list kernel, which then only serves list overflows and exact-score retries)
        // (odd widths are instantiated without the streaming form: their windows beyond 8192 samples take the exact kernel)
        const bool has_stream = !(fast_combo_exact_only(combo) && (p.running_stat_width & 1));
        const bool with_stream = chain && ext && capF >= 5120 && max_len > 6144 && has_stream;
        const bool with_big1 = chain && capF >= 5120 && cap > 6144 && !with_stream;  // windows of 6145..8192 samples
        A.exact_scores = approx ? 0 : 1;
        A.peak_filter = knobs.no_peak_filter ? 0 : 1;   // (only the approximate-keys launches look at it)
        unsigned *d_reasons = reinterpret_cast<unsigned *>(reinterpret_cast<unsigned char *>(d_ws) + dbg_ws_offset(n_reads));
        if (knobs.debug_occ) {   // (diagnostic: 16 counters at the end of this call's workspace, zeroed on its stream)
            WDX_HIP_TRY(hipMemsetAsync(d_reasons, 0, 64, stream));
            A.dbg_reasons = d_reasons;
        }
        FastArgs F{A, capF, capP, count, list, nullptr, nullptr, nullptr, nullptr, 0u, approx ? count + 3 : nullptr,
                   approx ? retry : nullptr, nullptr};
        if (with_big0) {
            F.big_count = count + 1;
            F.big_list = big0;
        } else if (with_stream) {
            F.big_count = count + 4;
            F.big_list = big2;
        } else if (with_big1) {
            F.big_count = count + 2;
            F.big_list = big1;
        }
        F.clip = ext ? clip : nullptr;
        void (*kern_st)(FastArgs) = combo == 2 ? fingerprint_fast_stream_kernel<18, 1>
                                    : (combo == 3 ? fingerprint_fast_stream_kernel<30, 2> : fingerprint_fast_stream_kernel<kFW, 1>);
        void (*kern)(FastArgs) = nullptr;
        // the SPLIT pair of the main launch (tile kernel + tail kernel) and of the 6144-sample list launch, where they exist
        void (*kern_a)(FastArgs) = nullptr, (*kern_b)(FastArgs) = nullptr, (*kern_a1)(FastArgs) = nullptr;
        void (*kern_l1)(FastArgs) = fingerprint_fast_list1_kernel<kNptLarge>;   // 6144 samples, one workgroup per entry
        void (*kern_ls)(FastArgs) = fingerprint_fast_list_kernel<kNptHuge>;     // 8192 samples, striding
        int slot = 0;
        // (the NBT = 2 widths: one set of four kernels each)
        auto wide_set = [&](auto fw_t) {
            constexpr int FWx = decltype(fw_t)::value;
            kern = ext ? fingerprint_fast_kernel<kNptLarge, false, FWx, 2, true> : fingerprint_fast_kernel<kNptLarge, false, FWx, 2, false>;
            kern_l1 = fingerprint_fast_list1_kernel<kNptLarge, FWx, 2>;
            kern_ls = fingerprint_fast_list_kernel<kNptHuge, FWx, 2>;
            kern_st = fingerprint_fast_stream_kernel<FWx, 2>;
            slot = 2;
            if constexpr (FWx >= 12 && FWx % 6 == 0) {   // (the widths whose EXT main kernel runs the filtered 512-entry list)
                kern_a = fingerprint_fast_kernel<kNptLarge, false, FWx, 2, true, true>;
                kern_b = fingerprint_split_tail_kernel<FWx, 2>;
            }
        };
        if (combo == 2) {
            kern = capF == 5120 ? fingerprint_fast_kernel<kNptMid, false, 18, 1, true>
                                : (ext ? fingerprint_fast_kernel<kNptLarge, false, 18, 1, true>
                                       : fingerprint_fast_kernel<kNptLarge, false, 18, 1, false>);
            kern_l1 = fingerprint_fast_list1_kernel<kNptLarge, 18, 1>;
            kern_ls = fingerprint_fast_list_kernel<kNptHuge, 18, 1>;
            slot = capF == 5120 ? 1 : 2;
            if (capF == 5120) {
                kern_a = fingerprint_fast_kernel<kNptMid, false, 18, 1, true, true>;
                kern_b = fingerprint_split_tail_kernel<18, 1>;
                kern_a1 = fingerprint_fast_list1_kernel<kNptLarge, 18, 1, true>;
            }
        } else if (combo == 3) {
            wide_set(std::integral_constant<int, 30>{});
        } else if (combo == 4) {
            wide_set(std::integral_constant<int, 6>{});
        } else if (combo == 5) {
            wide_set(std::integral_constant<int, 24>{});
        } else if (combo == 6) {
            wide_set(std::integral_constant<int, 36>{});
        } else if (combo == 7) {
            wide_set(std::integral_constant<int, 12>{});
        } else if (combo == 8) {
            wide_set(std::integral_constant<int, 18>{});
        } else if (combo >= 9) {
            // the exact-scores-only widths: instantiated in wdx_fingerprint_w1.hip / _w2.hip
            FastKernelSet ks{};
            if (!exact_only_kernels_a(p.running_stat_width, ext, ks) && !exact_only_kernels_b(p.running_stat_width, ext, ks) &&
                !exact_only_kernels_c(p.running_stat_width, ext, ks) && !exact_only_kernels_d(p.running_stat_width, ext, ks)) {
                set_error("no fast kernels for running_stat_width %d", (int)p.running_stat_width);
                return WDX_ERR_INVALID;
            }
            kern = ks.main;
            kern_l1 = ks.l1;
            kern_ls = ks.ls;
            kern_st = ks.st;
            slot = 2;
        } else if (capF == 4096) {
            kern = ext ? (d_prof ? fingerprint_fast_kernel<kNptSmall, true, kFW, 1, true> : fingerprint_fast_kernel<kNptSmall, false, kFW, 1, true>)
                       : (d_prof ? fingerprint_fast_kernel<kNptSmall, true> : fingerprint_fast_kernel<kNptSmall, false>);
        } else if (capF == 5120) {
            kern = d_prof ? fingerprint_fast_kernel<kNptMid, true, kFW, 1, true> : fingerprint_fast_kernel<kNptMid, false, kFW, 1, true>;
            slot = 1;
            kern_a = fingerprint_fast_kernel<kNptMid, false, kFW, 1, true, true>;
            kern_b = fingerprint_split_tail_kernel<kFW, 1>;
            kern_a1 = fingerprint_fast_list1_kernel<kNptLarge, kFW, 1, true>;
        } else {
            kern = ext ? (d_prof ? fingerprint_fast_kernel<kNptLarge, true, kFW, 1, true> : fingerprint_fast_kernel<kNptLarge, false, kFW, 1, true>)
                       : (d_prof ? fingerprint_fast_kernel<kNptLarge, true> : fingerprint_fast_kernel<kNptLarge, false>);
            slot = 2;
        }
        static LdsAttr attr_fast[40][12];
        if (int rc = attr_fast[combo - 1][(ext ? 6 : 0) + (d_prof ? 3 : 0) + slot].ensure(kern, flds)) return rc;
        if (knobs.debug_occ) {
            int nb = 0;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)kern, FB, flds);
            fprintf(stderr, "[wdx] fast kernel capF=%d capP=%d lds=%zu B -> %d workgroups/CU%s\n", capF, capP, flds, nb,
                    approx ? ", approximate score keys" : "");
        }
        // one workgroup per read (or list entry); grid.x * block.x must stay below 2^32: equal launch slices of at
        // most 2^31 / FB workgroups
        auto launch_sliced = [&](void (*k)(FastArgs), FastArgs &fa, int64_t n_wg, size_t lds_bytes, bool counted) {
            const int64_t max_slice = launch_slice_limit((1ll << 31) / FB), n_slices = (n_wg + max_slice - 1) / max_slice;
            const int64_t slice = (n_wg + n_slices - 1) / n_slices;
            for (int64_t base = 0; base < n_wg; base += slice) {
                const int64_t n = n_wg - base < slice ? n_wg - base : slice;
                fa.a.block_base = base;
                hipLaunchKernelGGL(k, dim3((unsigned)n), dim3(FB), lds_bytes, stream, fa);
                if (counted && n_launches) ++*n_launches;
            }
        };
        if (ext) {
            // A1 for the main kernel's reads (windows of 256 .. capF samples; longer ones are flagged CLIP_NONE and the
            // main kernel hands them to the lists before it would look at their record)
            if (main_ev && main_ev->c_first) (void)hipEventRecord(main_ev->c_first, stream);
            if (int rc = launch_clip_bounds(A, clip, capF, stream)) return rc;
            if (F.big_list) {   // the windows beyond capF go on the main kernel's hand-over list here (one atomic per 64 reads)
                if (int rc = launch_route_long_windows(A, capF, F.big_count, F.big_list, stream)) return rc;
                F.routed = 1;
            }
            if (main_ev && main_ev->c_first) {
                (void)hipEventRecord(main_ev->c_second, stream);
                main_ev->c_recorded = true;
            }
        }
        // The SPLIT form of the main kernel (large batches on approximate keys with the filtered 512-entry list -- the RNA004
        // triple, width 18, and the NBT = 2 widths from 12 up): the workgroup-per-read kernel ends after the tile pass and exports
        // the <= 256 peaks that can matter, one WAVE per read does the rest (fingerprint_split_tail_kernel) -- launch pairs over
        // slices of kSplitSlice reads, whose lists live in the workspace behind the read lists.  Not for the diagnostic builds
        // (the one-piece kernel serves those).
        // (not the refinement branch: measured 2.36 against 2.15 ms per 32 768 tRNA-like reads with the one-piece kernel)
        // (stop_phase == -2: the diagnostic build of the RNA004 pair, wdx_fingerprint_profile_dev's fast_path = 2)
        const bool prof_split = d_prof && stop_phase == -2 && combo == 1 && capF == 5120;
        if (prof_split) kern_a = fingerprint_fast_kernel<kNptMid, true, kFW, 1, true, true>;
        const bool split = kern_a && ext && approx && filt && capP == 512 && chain && (!d_prof || prof_split) && !rf && !knobs.no_split;
        if (main_ev && main_ev->first) (void)hipEventRecord(main_ev->first, stream);
        if (split) {
            static LdsAttr attr_split[40], attr_split_prof;
            if (int rc = (prof_split ? attr_split_prof : attr_split[combo - 1]).ensure(kern_a, flds)) return rc;
            F.split_ws = reinterpret_cast<unsigned char *>(d_ws) + split_ws_offset(n_reads);
            const int64_t slice = launch_slice_limit(kSplitSlice);
            for (int64_t base = 0; base < n_reads; base += slice) {
                const int64_t m = std::min<int64_t>(slice, n_reads - base);
                F.split_base = base;
                F.split_n = m;
                F.a.block_base = base;
                hipLaunchKernelGGL(kern_a, dim3((unsigned)m), dim3(FB), flds, stream, F);
                if (n_launches) ++*n_launches;
                std::pair<hipEvent_t, hipEvent_t> tp{nullptr, nullptr};
                if (main_ev && main_ev->first && main_ev->take) tp = main_ev->take(main_ev->take_arg);
                if (tp.first) (void)hipEventRecord(tp.first, stream);
                F.a.block_base = 0;
                hipLaunchKernelGGL(kern_b, dim3((unsigned)((m + kSplitWaves - 1) / kSplitWaves)), dim3(kSplitWaves * 64), 0, stream, F);
                if (tp.first) {
                    (void)hipEventRecord(tp.second, stream);
                    main_ev->tail.push_back(tp);
                }
            }
            F.a.block_base = 0;
        } else {
            launch_sliced(kern, F, n_reads, flds, true);
        }
        if (main_ev && main_ev->first) {
            (void)hipEventRecord(main_ev->second, stream);
            main_ev->recorded = true;
        }
        // the list kernels
        const int64_t grid = n_reads < 1024 ? n_reads : 1024;  // striding kernels: every CU busy, nothing more
        const int capF1 = 6144, capP1 = 1376, capF2 = 8192, capP2 = 1856;
        const size_t flds1 = fast_lds_bytes(capF1, capP1, nbt), flds2 = fast_lds_bytes(capF2, capP2, nbt);
        // the same kernels behind filtered appends (the approximate-keys launches; the exact-scores retry keeps the long
        // lists): 512 entries -> five workgroups per CU at 6144 samples, four at 8192
        // (the striding 8192-sample kernel is NOT an EXT instantiation: its in-kernel medians put their 2112-word histogram
        // where the peak list will be, so its list region must hold 8448 bytes -- 768 entries, not 512)
        const int capP1f = filt && nbt == 1 ? 512 : capP1, capP2f = filt && nbt == 1 ? 768 : capP2;
        const size_t flds1f = fast_lds_bytes(capF1, capP1f, nbt), flds2f = fast_lds_bytes(capF2, capP2f, nbt);
        static LdsAttr attr_l1[40], attr_huge[40];
        if (with_big0 || (approx && chain))
            if (int rc = attr_l1[combo - 1].ensure(kern_l1, flds1)) return rc;
        if (with_big0 || with_big1 || with_stream || (approx && chain))
            if (int rc = attr_huge[combo - 1].ensure(kern_ls, flds2)) return rc;
        if (with_big0) {
            // 11 % of RNA004 adapter windows are longer than 5120 samples: a grid for a quarter of the batch, one
            // workgroup per list entry, and the striding 8192-sample kernel for whatever lies beyond it
            const int64_t g1 = std::min<int64_t>(n_reads, std::max<int64_t>(1024, n_reads / 4));
            FastArgs F1{A, capF1, capP1f, count, list, with_stream ? count + 4 : (with_big1 ? count + 2 : nullptr),
                        with_stream ? big2 : (with_big1 ? big1 : nullptr), count + 1, big0, 0u, F.retry_count, F.retry_list, clip};
            if (int rc = launch_clip_bounds_list(A, clip, count + 1, big0, g1, stream)) return rc;
            if (split && kern_a1 && capP1f == 512) {
                // the same pair over the list's entries (slot = workgroup of the slice; most of the grid lies past the list's end
                // and leaves at once, in both kernels)
                static LdsAttr attr_split1[40];
                if (int rc = attr_split1[combo - 1].ensure(kern_a1, flds1f)) return rc;
                F1.split_ws = F.split_ws;
                const int64_t slice = launch_slice_limit(kSplitSlice);
                for (int64_t base = 0; base < g1; base += slice) {
                    const int64_t m = std::min<int64_t>(slice, g1 - base);
                    F1.split_base = base;
                    F1.split_n = m;
                    F1.a.block_base = base;
                    hipLaunchKernelGGL(kern_a1, dim3((unsigned)m), dim3(FB), flds1f, stream, F1);
                    F1.a.block_base = 0;
                    hipLaunchKernelGGL(kern_b, dim3((unsigned)((m + kSplitWaves - 1) / kSplitWaves)), dim3(kSplitWaves * 64), 0, stream, F1);
                }
                F1.a.block_base = 0;
            } else {
                launch_sliced(kern_l1, F1, g1, flds1f, false);
            }
            if (g1 < n_reads) {
                FastArgs F1b{A, capF2, capP2f, count, list, with_stream ? count + 4 : nullptr, with_stream ? big2 : nullptr,
                             count + 1, big0, (unsigned)g1, F.retry_count, F.retry_list, nullptr};
                hipLaunchKernelGGL(kern_ls, dim3((unsigned)grid), dim3(FB), flds2f,
                                   stream, F1b);
            }
        }
        if (with_big1) {
            FastArgs F2{A, capF2, capP2f, count, list, nullptr, nullptr, count + 2, big1, 0u, F.retry_count, F.retry_list, nullptr};
            hipLaunchKernelGGL(kern_ls, dim3((unsigned)grid), dim3(FB), flds2f, stream,
                               F2);
        }
        if (with_stream) {
            // windows of 6145 .. 16384 samples (RNA002: max_obs_trace + 2 * padding = 15 200): their clip bounds by one
            // workgroup per list entry (samples in LDS), then the streaming form of the fast body -- its LDS is the peak
            // list plus two tile buffers, whatever the window length: four workgroups per CU up to 12 288 samples, three
            // up to 16 384.  One workgroup per entry; the list's length is only known on the device.  With windows beyond
            // 8192 samples in the batch (RNA002-length reads: every read is on this list) the grids cover the batch; up to
            // 8192 the long windows are a tail of the batch (687 of 10 M synthetic RNA004 reads) and, for batches of more
            // than 2 M reads, the grids cover a sixteenth of it -- a workgroup past the list's end leaves at once, but 10 M of
            // them cost a millisecond per launch -- with the striding 8192-sample kernel behind them for whatever lies
            // beyond.  Doubts and refusals go to the exact kernel.
            const int scap = max_len <= 8192 ? 8192 : (max_len <= 12288 ? 12288 : 16384);
            const int capPs = filt ? (scap == 8192 ? 1024 : (scap == 12288 ? 1280 : 1536))   // (the list from kPeakTauLo up)
                                   : (scap == 8192 ? 1700 : (scap == 12288 ? 2520 : 3400));
            const size_t lds_cb = clip_block_lds_bytes(scap), lds_st = fast_stream_lds_bytes(capPs, nbt);
            static LdsAttr attr_cb, attr_st[40];
            if (int rc = attr_cb.ensure(clip_bounds_block_kernel, lds_cb)) return rc;
            if (int rc = attr_st[combo - 1].ensure(kern_st, lds_st)) return rc;
            // (WDX_OPT_MAX_LAUNCH_SLICE, the tests' switch for the multi-launch paths, also selects the bounded grids)
            const int64_t g5 = (scap == 8192 && (n_reads > (1ll << 21) || knobs.max_launch_slice > 0))
                                   ? std::max<int64_t>(1, n_reads / 16) : n_reads;
            const int64_t max_slice = launch_slice_limit(1ll << 22);
            // the one-wave clip kernel first (samples in registers, no barriers: twice the block kernel's rate per sample at
            // two workgroups of four reads per CU) for the windows its register file holds -- 8192 samples at 128 per lane,
            // 13 312 at 208; the workgroup kernel then finds a record for those and serves the rest: longer windows, and
            // the ones the wave form may not decide (negative samples it cannot clamp away)
            if (!knobs.no_wave_clip_long)
                if (int rc = launch_clip_bounds_list(A, clip, count + 4, big2, g5, stream, scap == 8192 ? 8192 : kClipWaveLongCap, true))
                    return rc;
            for (int64_t base = 0; base < g5; base += max_slice) {
                ClipBlockArgs CB{A, clip, count + 4, big2, scap};
                CB.a.block_base = base;
                hipLaunchKernelGGL(clip_bounds_block_kernel, dim3((unsigned)std::min<int64_t>(max_slice, g5 - base)), dim3(FB),
                                   lds_cb, stream, CB);
            }
            FastArgs F5{A, 16384, capPs, count, list, nullptr, nullptr, count + 4, big2, 0u, nullptr, nullptr, clip};
            // batches of long windows (beyond 8192 samples: the grids cover the batch anyway): a read with a decision inside
            // the error band, or whose cut does not clear the append filter's threshold, is redone by a second launch of the
            // same kernel on exact scores with the unfiltered list capacity -- instead of the exact general kernel, which
            // serves such a window at a twentieth of the rate.  (The 8192-sample list's slots are free: big1 is not in use
            // when the streaming kernel is.)
            const bool st_retry = approx && scap > 8192;
            if (st_retry) {
                F5.retry_count = count + 2;
                F5.retry_list = big1;
            }
            launch_sliced(kern_st, F5, g5, lds_st, false);
            if (st_retry) {
                const int capPx = scap == 12288 ? 2520 : 3400;
                const size_t lds_x = fast_stream_lds_bytes(capPx, nbt);
                if (int rc = attr_st[combo - 1].ensure(kern_st, lds_x)) return rc;
                FastArgs F6{A, 16384, capPx, count, list, nullptr, nullptr, count + 2, big1, 0u, nullptr, nullptr, clip};
                F6.a.exact_scores = 1;
                launch_sliced(kern_st, F6, n_reads, lds_x, false);
            }
            if (g5 < n_reads) {
                FastArgs F5b{A, capF2, capP2f, count, list, nullptr, nullptr, count + 4, big2, (unsigned)g5, F.retry_count,
                             F.retry_list, nullptr};
                hipLaunchKernelGGL(kern_ls, dim3((unsigned)grid), dim3(FB), flds2f, stream, F5b);
            }
        }
        if (approx && chain) {
            // about 2 reads in 1000: a grid for 1/64 of the batch on the 6144-sample instantiation with exact scores
            // (a window beyond 6144 samples moves on to the slow path), the striding kernel beyond
            const int64_t g3 = std::min<int64_t>(n_reads, std::max<int64_t>(1024, n_reads / 64));
            FastArgs F3{A, capF1, capP1, count, list, nullptr, nullptr, count + 3, retry, 0u, nullptr, nullptr, clip};
            F3.a.exact_scores = 1;
            launch_sliced(kern_l1, F3, g3, flds1, false);
            if (g3 < n_reads) {
                FastArgs F3b{F3.a, capF2, capP2, count, list, nullptr, nullptr, count + 3, retry, (unsigned)g3, nullptr,
                             nullptr, nullptr};
                hipLaunchKernelGGL(kern_ls, dim3((unsigned)grid), dim3(FB), flds2,
                                   stream, F3b);
            }
        }
        WDX_HIP_TRY(hipGetLastError());
        if (!chain) return WDX_SUCCESS;  // (ablation timing of the main kernel: the lists are left unprocessed)
        // the exact kernels take the clip bounds of a read that has a CLIP_OK record (every read of the batch has a record by
        // now: the main clip kernel writes one per read, the list forms fill in the longer windows) instead of redoing the
        // two workgroup-wide medians
        if (ext && !knobs.no_clip_reuse) A.clip = clip;
        if (rf) {
            // refinement branch: the exact kernel segments the adapters of the slow list's reads and leaves them, like the
            // fast kernels theirs, to the refinement kernels (reads it cannot hand over it refines in place); barcode
            // tails beyond the tail kernel's capacity come back on a list of their own for the exact kernel's full form
            int32_t *back = big2 + n_reads;
            A.refine_record = 1;
            if (int rc = small ? launch_fp_list<512>(A, lds, count, list, stream)
                               : launch_fp_list<1024>(A, lds, count, list, stream))
                return rc;
            A.refine_record = 0;
            if (int rc = launch_refine_tail(A, count + 5, back, stream)) return rc;
            // (a grid-stride kernel: one workgroup per CU serves this list, which is empty unless barcodes are very long --
            // 2048 workgroups of ~100 KB that only find it empty cost 70 us)
            if (int rc = small ? launch_fp_list<512>(A, lds, count + 5, back, stream, 256)
                               : launch_fp_list<1024>(A, lds, count + 5, back, stream, 256))
                return rc;
        } else if (int rc = small ? launch_fp_list<512>(A, lds, count, list, stream)
                                  : launch_fp_list<1024>(A, lds, count, list, stream))
            return rc;
        if (with_huge)
            if (int rc = launch_fp_big(A, cap, count, list, stream)) return rc;
        if (knobs.debug_occ) {  // (diagnostic: synchronises)
            unsigned c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            WDX_HIP_TRY(hipMemcpyAsync(c, count, 32, hipMemcpyDeviceToHost, stream));
            WDX_HIP_TRY(hipStreamSynchronize(stream));
            // (counter 2 is the 8192-sample list when the streaming kernel is not in the chain, else the streaming kernel's
            // own exact-scores retry list)
            fprintf(stderr, "[wdx] of %lld reads: %u beyond the main instantiation, %u %s, %u to the "
                            "streaming kernel, %u redone with exact scores, %u on the exact general kernel\n", (long long)n_reads,
                    c[1], c[2], with_stream ? "redone by the streaming kernel on exact scores" : "to the 8192-sample list kernel",
                    c[4], c[3], c[0]);
            {
                unsigned h[16];
                WDX_HIP_TRY(hipMemcpyAsync(h, d_reasons, 64, hipMemcpyDeviceToHost, stream));
                WDX_HIP_TRY(hipStreamSynchronize(stream));
                fprintf(stderr, "[wdx] handed to the exact kernel by the fast kernels, by reason (0 parameter gate / window, 1 NaN or negative, "
                                "2 sums not provably exact, 3 plateau or peak-list capacity, 4 neighbourhood, 5 kept-list capacity, 6 tie at the "
                                "top-E cut, 7 doubt without a retry list, 8 fewer peaks than events with accept_less_cpts):");
                for (int i = 0; i < 10; ++i) fprintf(stderr, " %u", h[i]);
                fprintf(stderr, "\n");
            }
            if (rf)
                fprintf(stderr, "[wdx] refinement: %u reads back from the tail kernel to the exact kernel (%u for a run of equal scores across a "
                                "tile's end, %u beyond the peak list)\n", c[5], c[6], c[7]);
            if (ext) {
                std::vector<ClipRec> h((size_t)n_reads);
                WDX_HIP_TRY(hipMemcpy(h.data(), clip, sizeof(ClipRec) * (size_t)n_reads, hipMemcpyDeviceToHost));
                long long f[4] = {0, 0, 0, 0};
                for (const ClipRec &cr : h) ++f[cr.flag & 3];
                fprintf(stderr, "[wdx] clip_bounds_kernel flags: %lld not taken, %lld ok, %lld NaN / negative, %lld sums not "
                                "provably exact\n", f[0], f[1], f[2], f[3]);
            }
        }
        return WDX_SUCCESS;
    }
    // The exact kernel for the whole batch (a window width or suppression reach without a fast instantiation, a signal
    // normalisation, WDX_OPT_EXACT_PATH): large batches still get their clip bounds from the one-wave kernel first (windows
    // up to 13 312 samples; 4.5 ms per million reads against the exact kernel's ~100 for its two workgroup-wide medians)
    if (d_ws && !d_prof && !knobs.no_clip_reuse && n_reads >= (knobs.fast_chain_min > 0 ? knobs.fast_chain_min : 2048)) {
        ClipRec *clip = reinterpret_cast<ClipRec *>(reinterpret_cast<unsigned char *>(d_ws) + 32);
        const int ccap = max_len <= 4096 ? 4096 : (max_len <= 5120 ? 5120 : (max_len <= 6144 ? 6144 : (max_len <= 8192 ? 8192 : kClipWaveLongCap)));
        if (int rc = launch_clip_bounds(A, clip, ccap, stream)) return rc;
        A.clip = clip;
    }
    if (int rc = small ? launch_fp_chunks<512, false>(A, lds, stream, n_launches)
                       : launch_fp_chunks<1024, false>(A, lds, stream, n_launches))
        return rc;
    if (with_huge) return launch_fp_big(A, cap, nullptr, nullptr, stream);
    return WDX_SUCCESS;
}
#endif  // WDX_DEV_KERNELS_ONLY

}  // namespace wdx

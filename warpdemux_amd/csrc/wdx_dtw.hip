// Banded DTW distance matrix for gfx950 -- stage B of the hot path.
//
// Replaces dtaidistance.dtw.distance_matrix as called from
// /root/reference/warpdemux/parallel_distances.py:34-43 and :59-67 (SURVEY.md §8 rows B1-B3,
// recurrence in App. A):  D[i+1][j+1] = (a_i-b_j)^2 + min(D[i][j], D[i][j+1]+p2, D[i+1][j]+p2),
// band |i-j| <= w-1, result sqrt(D[L][L]), float64 arithmetic, float32 output.
//
// Mapping (DESIGN.md "DTW kernels"): one LANE per (series-a, series-b) pair.  Lanes of a wave run
// over different a-series (read-minor layout AT[L][ldA] -> every row load is one coalesced 512-B
// wave access); the b-series is UNIFORM across the wave, so its samples arrive through the scalar
// data path (s_load) and cost no vector issue slots.  The Sakoe-Chiba band of the previous DP row
// lives in 2W-1 float64 registers per lane and is updated in place, left to right; there is no
// cross-lane traffic and no LDS.  All float64 ops are issued un-fused (-ffp-contract=off) in the
// oracle's order, except min(up+p2, left+p2) == min(up,left)+p2, which is exact because rounding
// is monotone.  No MFMA: this is a min-plus recurrence.
#include "wdx_common.h"

#include <math.h>
#include <stdlib.h>
#include <utility>

namespace wdx {

#define WDX_INF __builtin_huge_val()

// v_min_f64 without the canonicalising v_max_f64 x,x that fmin() costs per call in IEEE mode when
// the compiler cannot prove an operand quiet (the loop-carried band registers).  The operands here
// are always results of float64 adds, +0.0 or +inf -- never signalling NaNs; NaN inputs are
// handled by the per-series flags, not by the recurrence.
__device__ __forceinline__ double min_f64(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// FMA: the cell as ONE v_fma_f64, (x - y)^2 + t with a single rounding -- five float64 operations per cell instead of
// six.  Not the reference's bits cell by cell: see dtw_unsettled() for how the kernels return the reference's float32.
template <bool FMA>
__device__ __forceinline__ double dtw_cell(double d, double t) {
    if constexpr (FMA) return __builtin_fma(d, d, t);
    else return d * d + t;
}

template <int W, bool MASKED, bool FMA = false>
__device__ __forceinline__ void dtw_row(double (&r)[2 * W - 1], double x,
                                        const double *__restrict__ yi, double p2, int jbase,
                                        int jlo, int jhi) {
    constexpr int B = 2 * W - 1;
    double left = WDX_INF;
#pragma unroll
    for (int c = 0; c < B; ++c) {
        const double d = x - yi[c];
        double up = (c + 1 < B) ? r[c + 1] : WDX_INF;
        double t = (c == 0 ? up : min_f64(up, left)) + p2;  // left of the first band cell is +inf
        t = min_f64(t, r[c]);
        double v = dtw_cell<FMA>(d, t);
        if (MASKED) {
            int j = jbase + c;
            v = (j < jlo || j > jhi) ? WDX_INF : v;
        }
        r[c] = v;
        left = v;
    }
}

// dtw_row with a compile-time range [CLO, CHI] of band cells that lie inside the matrix: the head and tail
// rows of the band without masks, and without touching the cells outside (head: they stay +inf; tail:
// they go stale but are never read again).
template <int W, int CLO, int CHI, bool FMA = false>
__device__ __forceinline__ void dtw_row_ct(double (&r)[2 * W - 1], double x, const double *__restrict__ yi,
                                           double p2) {
    constexpr int B = 2 * W - 1;
    double left = WDX_INF;
#pragma unroll
    for (int c = CLO; c <= CHI; ++c) {
        const double d = x - yi[c];
        const double up = (c + 1 < B) ? r[c + 1 < B ? c + 1 : 0] : WDX_INF;
        double t = (c == CLO ? up : min_f64(up, left)) + p2;
        t = min_f64(t, r[c]);
        const double v = dtw_cell<FMA>(d, t);
        r[c] = v;
        left = v;
    }
}
// head rows i = 0 .. W-2 (band cells c >= W-1-i), tail rows i = L-W+1+t, t = 0 .. W-2 (c <= 2W-3-t)
template <int W, bool FMA, int... Is>
__device__ __forceinline__ void dtw_head_rows(double (&r)[2 * W - 1], const double *__restrict__ xp, int64_t ldA,
                                              const double *__restrict__ y, double p2,
                                              std::integer_sequence<int, Is...>) {
    (dtw_row_ct<W, W - 1 - Is, 2 * W - 2, FMA>(r, xp[(int64_t)Is * ldA], y + Is, p2), ...);
}
template <int W, bool FMA, int... Ts>
__device__ __forceinline__ void dtw_tail_rows(double (&r)[2 * W - 1], const double *__restrict__ xp, int64_t ldA,
                                              const double *__restrict__ y, double p2, int i0,
                                              std::integer_sequence<int, Ts...>) {
    (dtw_row_ct<W, 0, 2 * W - 3 - Ts, FMA>(r, xp[(int64_t)(i0 + Ts) * ldA], y + i0 + Ts, p2), ...);
}

// The fused cell and the reference's float32 (dtw_band_kernel, dtw_short_kernel, dtw_short_svm_kernel).
// Every kernel first runs the pair on fused cells (FMA = true).  With u = 2^-53: the reference's cell is
// fl(fl(d^2) + t), the fused one fl(d^2 + t'), d the same number in both; all values are non-negative, min and the
// rounding are monotone, so if the three predecessor cells agree within a relative e the sums min(..) + p2 agree within
// e + 2u and the cells within e + 5u (first order) -- a cell's predecessors lie one step of i + j earlier, hence the final
// cells agree within (2 L - 1) 5u (1 + o(1)), and their correctly rounded square roots within half of that + 2u.  The
// kernels take delta = 32 (L + 1) u (more than three times that): when [res (1 - delta), res (1 + delta)] rounds to one
// float32, the reference's distance is that float32 (float conversion is monotone).  Otherwise -- a float32 rounding
// boundary inside the interval: about one pair in 10^5 at L = 110 -- or when the sum left [1e-280, 1e280] (underflow
// makes errors absolute; zero is exact: a zero fused sum means every term on its path was below 2^-1074 in both forms),
// the WAVE runs the pair again on the reference's six operations and the flagged lanes take that result.  Returns the
// lanes to settle.
__device__ __forceinline__ bool dtw_unsettled(double Dv, double res, float f, double delta) {
    const bool same = (float)(res * (1.0 - delta)) == f && (float)(res * (1.0 + delta)) == f;
    const bool mid = Dv == 0.0 || (Dv > 1e-280 && Dv < 1e280);
    return !(same && mid);
}
__device__ __forceinline__ double dtw_delta(int L) { return 32.0 * (double)(L + 1) * 0x1p-53; }
// WDX_OPT_DTW_UNFUSED (the kernels' `unfused` argument): 0 as above | 1 the reference's six operations only | 2 fused, and
// every pair is run again (tests: the second pass and the selection) | 3 fused, never run again (diagnostic: how often the
// fused float32 differs -- NOT the reference's results)
__device__ __forceinline__ bool dtw_redo(int mode, bool unsettled) {
    if (mode == 2) return true;
    if (mode == 3) return false;
    return __ballot(unsettled) != 0ull;
}

// np.argmin running update on float32 values (first minimum; first NaN wins outright)
struct ArgminAcc {
    float best = __builtin_huge_valf();
    int idx = 0;
    bool nan = false;
    bool any = false;
    __device__ __forceinline__ void push(float v, int i) {
        if (nan) return;
        if (v != v) {
            nan = true;
            idx = i;
        } else if (!any || v < best) {
            best = v;
            idx = i;
        }
        any = true;
    }
};

// true when the float64 row holds a NaN: lane-private sweep of a row-major series (16-byte loads)
__device__ __forceinline__ bool row_has_nan(const double *__restrict__ row, int L) {
    bool f = false;
    int i = 0;
    // (sixteen values per trip, their loads in flight together: two at a time the sweep was 55 memory latencies in a row per
    // wave of 110-point fingerprints -- 1.1 of the DTW kernel's 47.8 ms per 10 M reads)
    for (; i + 16 <= L; i += 16) {
        double v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = row[i + k];
#pragma unroll
        for (int k = 0; k < 16; ++k) f |= v[k] != v[k];
    }
    for (; i + 2 <= L; i += 2) {
        const double a = row[i], b = row[i + 1];
        f |= (a != a) | (b != b);
    }
    if (i < L) f |= row[i] != row[i];
    return f;
}

// ROWMAJOR: the a-series arrive as they are produced -- (nA, L) row-major, a lane reads its OWN row (ldA is
// ignored).  Loads of one lane are contiguous, so the body rows are fetched eight at a time with 16-byte
// loads (one 64-byte sector per lane and chunk instead of one per row) one chunk ahead of the recurrence,
// and the NaN flag of the series is computed here when the caller has none: no transposed copy of the
// fingerprints, no separate flag kernel.
template <int W, bool EXACT_W, bool ROWMAJOR>
__global__ __launch_bounds__(64) void dtw_band_kernel(
    const double *__restrict__ AT, int64_t ldA, int64_t nA, const uint8_t *__restrict__ a_nan,
    const double *__restrict__ Bpad, int64_t Lpad, int halo, int nB,
    const uint8_t *__restrict__ b_nan, int L, int w, double p2, float *__restrict__ out,
    int64_t sA, int64_t sB, int32_t *__restrict__ argmin, int refs_per_block, int unfused) {
    constexpr int B = 2 * W - 1;
    const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = a < nA;
    const int64_t al = active ? a : nA - 1;
    const int b0 = blockIdx.y * refs_per_block;
    const int b1 = min(nB, b0 + refs_per_block);
    if (ROWMAJOR) ldA = 1;
    const double *__restrict__ xp = ROWMAJOR ? AT + al * (int64_t)L : AT + al;
    // ROWMAJOR without flags: the NaN sweep of the lane's row is LAZY.  A NaN sample makes every cell of its row NaN, and from
    // there on no cell is finite again (a cell's three predecessors are NaN or +inf: v_min_f64 returns the operand that is not
    // NaN, +inf + p2 = +inf, d^2 + inf = inf) -- so a finite result proves a NaN-free row, and only a wave that sees a result
    // that is not finite (a failed read's NaN fingerprint, an overflow) sweeps its rows, once.  The eager sweep was 55 memory
    // latencies in a row per wave: 1.2 of the kernel's 47.8 ms per 10 M reads.
    bool anan = a_nan ? (a_nan[al] != 0) : false;
    bool swept = a_nan != nullptr || !ROWMAJOR;   // (uniform)
    const double delta = dtw_delta(L);
    ArgminAcc acc;

    for (int b = b0; b < b1; ++b) {
        const double *__restrict__ y = Bpad + (int64_t)b * Lpad + (halo - (W - 1));
        // the pair's accumulated cost D[L][L] on fused (FMA) or on the reference's cells
        auto pair_cost = [&](auto fma_t) -> double {
            constexpr bool FMA = decltype(fma_t)::value;
            double r[B];
#pragma unroll
            for (int c = 0; c < B; ++c) r[c] = WDX_INF;
            r[W - 1] = 0.0;
            double xn = xp[0];
            int i = 0;
            if (EXACT_W && L >= 2 * (W - 1)) {
                // head and tail rows from templates (no masks, no cells outside the matrix); loads of x are
                // independent of the recurrence and get hoisted by the compiler
                dtw_head_rows<W, FMA>(r, xp, ldA, y, p2, std::make_integer_sequence<int, W - 1>{});
                i = W - 1;
                if (ROWMAJOR) {
                    constexpr int CH = 8;
                    const int body_end = L - W + 1;
                    if (i + CH <= body_end) {
                        double xc[CH];
#pragma unroll
                        for (int k = 0; k < CH; ++k) xc[k] = xp[i + k];
                        for (; i + CH <= body_end; i += CH) {
                            double xq[CH];
                            const bool more = i + 2 * CH <= body_end;  // wave-uniform
                            if (more) {
#pragma unroll
                                for (int k = 0; k < CH; ++k) xq[k] = xp[i + CH + k];
                            }
#pragma unroll
                            for (int k = 0; k < CH; ++k) dtw_row<W, false, FMA>(r, xc[k], y + i + k, p2, 0, 0, 0);
                            if (more) {
#pragma unroll
                                for (int k = 0; k < CH; ++k) xc[k] = xq[k];
                            }
                        }
                    }
                    if (i < body_end) xn = xp[i];
                } else {
                    xn = xp[(int64_t)(W - 1) * ldA];
                }
                for (; i < L - W + 1; ++i) {
                    const double x = xn;
                    xn = xp[(int64_t)(i + 1) * ldA];  // i + 1 <= L - W + 1 < L
                    dtw_row<W, false, FMA>(r, x, y + i, p2, 0, 0, 0);
                }
                dtw_tail_rows<W, FMA>(r, xp, ldA, y, p2, L - W + 1, std::make_integer_sequence<int, W - 1>{});
            } else if (EXACT_W) {
                const int head_end = min(W - 1, L);          // rows whose band leaves [0, L)
                const int body_end = max(head_end, L - W + 1);
                for (; i < head_end; ++i) {
                    double x = xn;
                    if (i + 1 < L) xn = xp[(int64_t)(i + 1) * ldA];
                    dtw_row<W, true, FMA>(r, x, y + i, p2, i - (W - 1), 0, L - 1);
                }
                for (; i < body_end; ++i) {
                    double x = xn;
                    if (i + 1 < L) xn = xp[(int64_t)(i + 1) * ldA];
                    dtw_row<W, false, FMA>(r, x, y + i, p2, 0, 0, 0);
                }
                for (; i < L; ++i) {
                    double x = xn;
                    if (i + 1 < L) xn = xp[(int64_t)(i + 1) * ldA];
                    dtw_row<W, true, FMA>(r, x, y + i, p2, i - (W - 1), 0, L - 1);
                }
            } else {
                for (; i < L; ++i) {
                    double x = xn;
                    if (i + 1 < L) xn = xp[(int64_t)(i + 1) * ldA];
                    dtw_row<W, true, FMA>(r, x, y + i, p2, i - (W - 1), max(0, i - (w - 1)),
                                          min(L - 1, i + (w - 1)));
                }
            }
            return r[W - 1];
        };
        auto lazy_sweep = [&](double r_) __attribute__((always_inline)) {
            if (!swept && __ballot(!(r_ < WDX_INF)) != 0ull) {   // (rare: see above)
                anan = row_has_nan(xp, L);
                swept = true;
            }
        };
        const bool bnan = b_nan && b_nan[b];
        double res;
        bool redo = unfused == 1;   // (uniform)
        if (!redo) {
            const double Dv = pair_cost(std::true_type{});
            res = sqrt(Dv);
            lazy_sweep(res);
            redo = dtw_redo(unfused, !(anan || bnan) && dtw_unsettled(Dv, res, (float)res, delta));   // (the wave's decision)
        }
        if (redo) {
            res = sqrt(pair_cost(std::false_type{}));
            lazy_sweep(res);
        }
        const bool pnan = anan || bnan;
        if (pnan) res = __builtin_nan("");
        float f = (float)res;
        if (active) out[a * sA + (int64_t)b * sB] = f;
        acc.push(f, b);
    }
    if (argmin && active) argmin[a] = acc.idx;
}

// Short series, fully unrolled (the shipped DTW_SVM models: 25-point fingerprints, window 15, penalty 0.1
// -- every model under warpdemux/models/model_files/).  One lane per query series, the DP column
// vector D[L] and the query x[L] in registers, the reference uniform across the wave (scalar loads);
// rows and columns are compile-time, so the band limits cost nothing and only cells inside the band
// are computed (515 of the 725 that the rolling-band kernel evaluates for L = 25, all of them with
// masks there).  Same float64 operations per cell as dtw_row -> identical bits.
// row I of the short-series DP; everything about the band is a compile-time constant
template <int L, int W, int I, bool FMA>
__device__ __forceinline__ void dtw_short_row(double (&D)[L], const double xi, const double *__restrict__ y,
                                              const double p2) {
    constexpr int jlo = I - (W - 1) > 0 ? I - (W - 1) : 0;
    constexpr int jhi = I + (W - 1) < L - 1 ? I + (W - 1) : L - 1;
    double left = WDX_INF;
    double diag = jlo == 0 ? (I == 0 ? 0.0 : WDX_INF) : D[jlo > 0 ? jlo - 1 : 0];
#pragma unroll
    for (int j = jlo; j <= jhi; ++j) {
        const double up = D[j];
        const double d = xi - y[j];
        // min(min(up, left) + p2, diag) == min(min(up + p2, diag), left + p2) (rounding is monotone): the part that
        // does not depend on the cell to the left is off the row's dependency chain (3 dependent ops per cell, not 4)
        double t = __builtin_fmin(up + p2, diag);
        if (j != jlo) t = __builtin_fmin(t, left + p2);
        const double v = dtw_cell<FMA>(d, t);
        diag = up;
        D[j] = v;
        left = v;
    }
}
template <int L, int W, bool FMA, int... Is>
__device__ __forceinline__ void dtw_short_rows(double (&D)[L], const double (&x)[L], const double *__restrict__ y,
                                               const double p2, std::integer_sequence<int, Is...>) {
    (dtw_short_row<L, W, Is, FMA>(D, x[Is], y, p2), ...);
}
// the pair's distance as float64, the reference's float32 after conversion (see dtw_unsettled)
template <int L, int W>
__device__ __forceinline__ double dtw_short_pair(const double (&x)[L], const double *__restrict__ y, const double p2,
                                                 const bool pnan, const double delta, const int unfused) {
    double res;
    bool redo = unfused == 1;   // (uniform)
    if (!redo) {
        double D[L];
#pragma unroll
        for (int j = 0; j < L; ++j) D[j] = WDX_INF;
        dtw_short_rows<L, W, true>(D, x, y, p2, std::make_integer_sequence<int, L>{});
        res = sqrt(D[L - 1]);
        redo = dtw_redo(unfused, !pnan && dtw_unsettled(D[L - 1], res, (float)res, delta));
    }
    if (redo) {
        double D[L];
#pragma unroll
        for (int j = 0; j < L; ++j) D[j] = WDX_INF;
        dtw_short_rows<L, W, false>(D, x, y, p2, std::make_integer_sequence<int, L>{});
        res = sqrt(D[L - 1]);
    }
    return pnan ? __builtin_nan("") : res;
}

template <int L, int W, bool ROWMAJOR>
__global__ __launch_bounds__(64) void dtw_short_kernel(
    const double *__restrict__ AT, int64_t ldA, int64_t nA, const uint8_t *__restrict__ a_nan,
    const double *__restrict__ Bpad, int64_t Lpad, int halo, int nB,
    const uint8_t *__restrict__ b_nan, double p2, float *__restrict__ out, int64_t sA, int64_t sB,
    int32_t *__restrict__ argmin, int refs_per_block, int unfused) {
    const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = a < nA;
    const int64_t al = active ? a : nA - 1;
    const int b0 = blockIdx.y * refs_per_block;
    const int b1 = min(nB, b0 + refs_per_block);
    const double delta = dtw_delta(L);
    double x[L];
    bool anan = a_nan ? (a_nan[al] != 0) : false;
#pragma unroll
    for (int i = 0; i < L; ++i) x[i] = ROWMAJOR ? AT[al * (int64_t)L + i] : AT[(int64_t)i * ldA + al];
    if (ROWMAJOR && !a_nan) {
#pragma unroll
        for (int i = 0; i < L; ++i) anan |= x[i] != x[i];
    }
    ArgminAcc acc;
    for (int b = b0; b < b1; ++b) {
        const double *__restrict__ y = Bpad + (int64_t)b * Lpad + halo;
        const double res = dtw_short_pair<L, W>(x, y, p2, anan || (b_nan && b_nan[b]), delta, unfused);
        const float f = (float)res;
        if (active) out[a * sA + (int64_t)b * sB] = f;
        acc.push(f, b);
    }
    if (argmin && active) argmin[a] = acc.idx;
}

// Any window / any length: two rolling DP rows per lane in global scratch, element (row, j) of lane
// t at scratch[(row*(L+1)+j)*nT + t] (coalesced across lanes).  Slow path; correctness first.
__global__ __launch_bounds__(64) void dtw_scratch_kernel(
    const double *__restrict__ AT, int64_t ldA, int64_t a_base, int64_t nA,
    const uint8_t *__restrict__ a_nan, const double *__restrict__ Bpad, int64_t Lpad, int halo,
    int nB, const uint8_t *__restrict__ b_nan, int L, int w, double p2, float *__restrict__ out,
    int64_t sA, int64_t sB, double *__restrict__ scratch, int64_t nT) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t a = a_base + t;
    const bool active = a < nA && t < nT;
    const int64_t al = a < nA ? a : nA - 1;
    const int64_t tl = t < nT ? t : nT - 1;
    const bool anan = a_nan ? (a_nan[al] != 0) : false;
    double *row0 = scratch + tl;
    const int64_t rs = (int64_t)(L + 1) * nT;
    for (int b = 0; b < nB; ++b) {
        const double *__restrict__ y = Bpad + (int64_t)b * Lpad + halo;
        double *prev = row0, *cur = row0 + rs;
        for (int j = 0; j <= L; ++j) prev[(int64_t)j * nT] = WDX_INF;
        prev[0] = 0.0;
        for (int i = 0; i < L; ++i) {
            const double x = AT[(int64_t)i * ldA + al];
            const int j0 = max(0, i - (w - 1));
            const int j1 = min(L, i + w);
            cur[(int64_t)j0 * nT] = WDX_INF;
            if (j1 + 1 <= L) cur[(int64_t)(j1 + 1) * nT] = WDX_INF;
            double left = WDX_INF;
            double diag = prev[(int64_t)j0 * nT];
            for (int j = j0; j < j1; ++j) {
                double d = x - y[j];
                d = d * d;
                double up = prev[(int64_t)(j + 1) * nT];
                double tt = min_f64(up, left) + p2;
                tt = min_f64(tt, diag);
                double v = d + tt;
                cur[(int64_t)(j + 1) * nT] = v;
                left = v;
                diag = up;
            }
            double *sw = prev;
            prev = cur;
            cur = sw;
        }
        double res = sqrt(prev[(int64_t)L * nT]);
        if (anan || (b_nan && b_nan[b])) res = __builtin_nan("");
        if (active) out[a * sA + (int64_t)b * sB] = (float)res;
    }
}

// ---- anti-diagonal wavefront kernel (latency path: few pairs, e.g. the live 100 ms ticks) ----------
// One 16-lane DPP row per (read, reference) pair, four pairs per wave.  Front k = i + j is evaluated
// by all lanes at once: lane l holds band offset o = j - i = 2l - 14 on even fronts and 2l - 15 on odd
// ones, so  up   (i-1, j) = previous front, lane l+1 (even k) / lane l   (odd k)
//           left (i, j-1) = previous front, lane l   (even k) / lane l-1 (odd k)
//           diag (i-1,j-1) = front k-2, same lane
// i.e. one row_shl:1 or row_shr:1 DPP move per front and no LDS traffic for the band; the two series
// of a pair are staged in LDS once.  2L-1 dependent fronts instead of L*(2w-1) dependent cells: ~15x
// shorter critical path than the lane-per-pair kernel, at ~1.6x its instruction count per pair, hence
// only used when the whole problem fits the machine at once.  Same float64 operations per cell.
constexpr int kWfPairsPerBlock = 16;  // 256 threads

__device__ __forceinline__ double dpp_row_shift(double v, double fill, bool left) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    const int flo = __double2loint(fill), fhi = __double2hiint(fill);
    if (left) {  // lane l <- lane l+1
        lo = __builtin_amdgcn_update_dpp(flo, lo, 0x101, 0xf, 0xf, false);
        hi = __builtin_amdgcn_update_dpp(fhi, hi, 0x101, 0xf, 0xf, false);
    } else {     // lane l <- lane l-1
        lo = __builtin_amdgcn_update_dpp(flo, lo, 0x111, 0xf, 0xf, false);
        hi = __builtin_amdgcn_update_dpp(fhi, hi, 0x111, 0xf, 0xf, false);
    }
    return __hiloint2double(hi, lo);
}

__global__ __launch_bounds__(256) void dtw_wavefront_kernel(
    const double *__restrict__ X, int64_t nX, const double *__restrict__ Ypad, int64_t Lpad, int halo,
    int nY, int L, int w, double p2, float *__restrict__ out) {
    extern __shared__ double wf_lds[];  // per pair: x[L], y[L]
    __shared__ int nanflag[kWfPairsPerBlock];
    const int tid = threadIdx.x;
    const int row = tid >> 4, l = tid & 15;
    const int64_t npairs = nX * (int64_t)nY;
    const int64_t q = (int64_t)blockIdx.x * kWfPairsPerBlock + row;
    const bool active = q < npairs;
    const int64_t qq = active ? q : npairs - 1;
    const int64_t r = qq / nY;
    const int c = (int)(qq % nY);
    double *xs = wf_lds + (size_t)row * 2 * L, *ys = xs + L;
    if (l == 0) nanflag[row] = 0;
    __syncthreads();
    {
        const double *xg = X + r * L, *yg = Ypad + (int64_t)c * Lpad + halo;
        int bad = 0;
        for (int t = l; t < L; t += 16) {
            const double xv = xg[t], yv = yg[t];
            bad |= (xv != xv) | (yv != yv);
            xs[t] = xv;
            ys[t] = yv;
        }
        if (bad) atomicOr(&nanflag[row], 1);
    }
    __syncthreads();
    double prev2 = (l == 7) ? 0.0 : WDX_INF;  // front k-2 (holds the virtual D[0][0] = 0 for cell (0,0))
    double prev1 = WDX_INF;                     // front k-1
    double cur = WDX_INF;
    const int nfront = 2 * L - 1;
    for (int k = 0; k < nfront; ++k) {
        const bool odd = k & 1;
        const int o = 2 * l - (odd ? 15 : 14);
        const int i = (k - o) >> 1, j = (k + o) >> 1;  // k - o is even by construction
        const bool valid = (odd || l < 15) && i >= 0 && j >= 0 && i < L && j < L && o <= w - 1 && -o <= w - 1;
        const double up = odd ? prev1 : dpp_row_shift(prev1, WDX_INF, true);
        const double left = odd ? dpp_row_shift(prev1, WDX_INF, false) : prev1;
        const double xv = xs[min(max(i, 0), L - 1)], yv = ys[min(max(j, 0), L - 1)];
        double d = xv - yv;
        d = d * d;
        double t = min_f64(up, left) + p2;
        t = min_f64(t, prev2);
        const double v = valid ? d + t : WDX_INF;
        prev2 = prev1;
        prev1 = v;
        cur = v;
    }
    // cell (L-1, L-1): last front (even), o = 0 -> lane 7
    if (active && l == 7) {
        double res = sqrt(cur);
        if (nanflag[row]) res = __builtin_nan("");
        out[q] = (float)res;
    }
}

int64_t dtw_scratch_bytes(int64_t L, int window) {
    if (window <= kMaxRegWindow) return 0;
    return 2 * (L + 1) * 65536 * (int64_t)sizeof(double);
}

// Row-major X (nX, L) against padded refs; out (nX, nY) row-major.  Caller checked eligibility.
int launch_dtw_wavefront(const double *X, int64_t nX, const double *Ypad, int64_t Lpad, int halo,
                         int64_t nY, int64_t L, int window, double penalty, float *out,
                         hipStream_t stream) {
    const int w = (window <= 0 || window > L) ? (int)L : window;
    const int64_t npairs = nX * nY;
    if (npairs == 0) return WDX_SUCCESS;
    const size_t lds = (size_t)kWfPairsPerBlock * 2 * (size_t)L * sizeof(double);
    hipLaunchKernelGGL(dtw_wavefront_kernel, dim3((unsigned)((npairs + kWfPairsPerBlock - 1) / kWfPairsPerBlock)),
                       dim3(256), lds, stream, X, nX, Ypad, Lpad, halo, (int)nY, (int)L, w,
                       penalty * penalty, out);
    WDX_HIP_TRY(hipGetLastError());
    return WDX_SUCCESS;
}

bool dtw_wavefront_eligible(int64_t nX, int64_t nY, int64_t L, int window, const Knobs &knobs) {
    const int64_t w = (window <= 0 || window > L) ? L : window;
    return w <= 16 && L >= 1 && L <= 256 && nX * nY <= 16384 && !knobs.no_wavefront;
}

int launch_dtw(const double *AT, int64_t ldA, int64_t nA, const uint8_t *a_nan, const double *Bpad,
               int64_t Lpad, int halo, int64_t nB, const uint8_t *b_nan, int64_t L, int window,
               double penalty, float *out, int64_t sA, int64_t sB, int32_t *d_argmin,
               void *d_scratch, int64_t scratch_bytes, hipStream_t stream, const Knobs &knobs,
               bool a_rowmajor) {
    if (nA == 0 || nB == 0) return WDX_SUCCESS;
    if (L <= 0) {
        set_error("DTW series length must be positive");
        return WDX_ERR_INVALID;
    }
    int w = (window <= 0 || window > L) ? (int)L : window;  // |i-j| <= w-1 is vacuous beyond L
    const double p2 = penalty * penalty;
    if (w > kMaxRegWindow) {
        if (a_rowmajor) {
            set_error("the scratch-row DTW path takes the read-minor layout");
            return WDX_ERR_INVALID;
        }
        const int64_t nT = 65536;
        if (scratch_bytes < dtw_scratch_bytes(L, w) || !d_scratch) {
            set_error("DTW scratch too small for window %d", w);
            return WDX_ERR_INVALID;
        }
        if (d_argmin) {
            set_error("fused argmin is not available on the scratch-row DTW path");
            return WDX_ERR_INVALID;
        }
        for (int64_t a_base = 0; a_base < nA; a_base += nT) {
            int64_t n = nA - a_base < nT ? nA - a_base : nT;
            dim3 grid((unsigned)((n + 63) / 64));
            hipLaunchKernelGGL(dtw_scratch_kernel, grid, dim3(64), 0, stream, AT, ldA, a_base, nA,
                               a_nan, Bpad, Lpad, halo, (int)nB, b_nan, (int)L, w, p2, out, sA, sB,
                               (double *)d_scratch, nT);
        }
        WDX_HIP_TRY(hipGetLastError());
        return WDX_SUCCESS;
    }
    if (halo < kMaxRegWindow - 1) {
        set_error("reference rows need a halo of %d samples", kMaxRegWindow - 1);
        return WDX_ERR_INVALID;
    }
    const int64_t gx = (nA + 63) / 64;
    // one block walks `rpb` refs.  With enough a-series to fill 256 CUs x 8 waves a single block
    // walks them all and (if asked) folds the argmin in; otherwise the refs are split over grid.y
    // and the argmin is a second, tiny kernel.
    int32_t *fused_argmin = nullptr;
    int rpb = (int)nB;
    if (gx >= 8 * 3072 || nB == 1 || (gx >= 2048 && nB < 32)) {
        fused_argmin = d_argmin;
    } else {
        // A wave lives for the whole launch (one block = one wave walking its refs): with about as many waves as
        // the chip has slots (256 CUs x 4 SIMDs x 3 waves of this kernel = 3072) a few stragglers double the
        // launch time -- 1563 x 2 blocks ran at 67 % of the rate of the same kernel on a long grid.  Aim for
        // >= 8 waves per slot (dynamic balancing by the dispatcher), at least 16 refs per block.
        int64_t max_y = (nB + 15) / 16;
        int64_t want_y = (8 * 3072 + gx - 1) / gx;
        if (gx * max_y <= 3072) {  // fits the chip at once anyway: latency-bound -- as many short blocks as it takes
            want_y = (2048 + gx - 1) / gx;
            max_y = nB;
        }
        if (want_y > max_y) want_y = max_y;
        if (want_y < 1) want_y = 1;
        rpb = (int)((nB + want_y - 1) / want_y);
    }
    if (d_argmin && (sA != nB || sB != 1)) {
        set_error("argmin needs the row-major (nA, nB) output layout");
        return WDX_ERR_INVALID;
    }
    dim3 grid((unsigned)gx, (unsigned)((nB + rpb - 1) / rpb));
    if (L == 25 && w == 15 && !knobs.no_short_dtw) {
        if (a_rowmajor)
            hipLaunchKernelGGL((dtw_short_kernel<25, 15, true>), grid, dim3(64), 0, stream, AT, ldA, nA, a_nan, Bpad,
                               Lpad, halo, (int)nB, b_nan, p2, out, sA, sB, fused_argmin, rpb, knobs.dtw_unfused);
        else
            hipLaunchKernelGGL((dtw_short_kernel<25, 15, false>), grid, dim3(64), 0, stream, AT, ldA, nA, a_nan, Bpad,
                               Lpad, halo, (int)nB, b_nan, p2, out, sA, sB, fused_argmin, rpb, knobs.dtw_unfused);
        WDX_HIP_TRY(hipGetLastError());
        if (d_argmin && !fused_argmin) return launch_argmin(out, nA, nB, d_argmin, stream);
        return WDX_SUCCESS;
    }
#define WDX_LAUNCH_BAND(WW, EX)                                                                          \
    do {                                                                                                 \
        if (a_rowmajor)                                                                                  \
            hipLaunchKernelGGL((dtw_band_kernel<WW, EX, true>), grid, dim3(64), 0, stream, AT, ldA, nA,  \
                               a_nan, Bpad, Lpad, halo, (int)nB, b_nan, (int)L, w, p2, out, sA, sB,      \
                               fused_argmin, rpb, knobs.dtw_unfused);                                    \
        else                                                                                             \
            hipLaunchKernelGGL((dtw_band_kernel<WW, EX, false>), grid, dim3(64), 0, stream, AT, ldA, nA, \
                               a_nan, Bpad, Lpad, halo, (int)nB, b_nan, (int)L, w, p2, out, sA, sB,      \
                               fused_argmin, rpb, knobs.dtw_unfused);                                    \
    } while (0)
    if (w == 15) {
        WDX_LAUNCH_BAND(15, true);
    } else if (w <= 8) {
        WDX_LAUNCH_BAND(8, false);
    } else if (w <= 16) {
        WDX_LAUNCH_BAND(16, false);
    } else {
        WDX_LAUNCH_BAND(32, false);
    }
#undef WDX_LAUNCH_BAND
    WDX_HIP_TRY(hipGetLastError());
    if (d_argmin && !fused_argmin) return launch_argmin(out, nA, nB, d_argmin, stream);
    return WDX_SUCCESS;
}

// ---- the shipped models' DTW with the SVM decision sums in its epilogue (wdx_demux_svm_dev; round 4) --------------
// dtw_short_kernel's body over references in libsvm's SUPPORT-VECTOR order (grouped by class), one block (= one wave)
// per 64 reads and CHUNK of one class's vectors.  Instead of writing the distance, the lane turns it into the kernel
// value k = expf(-gamma d^p) (float32 like the reference, models/dtw_svm.py:21-22) and adds coef[q][s] * k to its k - 1
// decision sums of that class -- P[q][c] = sum over the class's vectors -- kept in LDS (lane-private slots [q][lane]: the
// DTW body uses no LDS and sits at 166 of the 168 VGPRs three waves per SIMD allow, so the sums cannot live in
// registers).  A chunk's sums go to P[slot][q][read] (slot = class * halves + half); svm_finish sums the halves, adds
// -rho and runs the sigmoids / coupling.  The (n, nY) distance matrix does not exist.  Summation order: the class's
// vectors in order -- deterministic, no atomics.
struct DtwSvmArgs {
    const double *X;            // (nA, L) row-major fingerprints
    int64_t nA;
    const double *Ypad;         // references in support-vector order, padded rows
    int64_t Lpad;
    int halo;
    const uint8_t *y_nan;       // per reference (support-vector order)
    double p2;
    const double *coefT;        // [n_sv][k - 1]: dual coefficients, vector-major
    const int32_t *chunk_ref0;  // [n_chunks + 1] first vector of each chunk (chunks never straddle a class)
    const int32_t *chunk_slot;  // [n_chunks] class * halves + half
    int km1, pwr;
    float ngamma;
    double *P;                  // [n_slots][k - 1][nA]
    int unfused;                // 1: the reference's six operations per cell only (WDX_OPT_DTW_UNFUSED)
};
template <int L, int W>
__global__ __launch_bounds__(64, 3) void dtw_short_svm_kernel(DtwSvmArgs G) {
    __shared__ double sums[16][64];
    const int lane = threadIdx.x;
    const int64_t a = (int64_t)blockIdx.x * 64 + lane;
    const bool active = a < G.nA;
    const int64_t al = active ? a : G.nA - 1;
    const int b0 = G.chunk_ref0[blockIdx.y], b1 = G.chunk_ref0[blockIdx.y + 1];
    const int km1 = G.km1;
    const double delta = dtw_delta(L);
    double x[L];
    bool anan = false;
#pragma unroll
    for (int i = 0; i < L; ++i) {
        x[i] = G.X[al * (int64_t)L + i];
        anan |= x[i] != x[i];
    }
    for (int q = 0; q < km1; ++q) sums[q][lane] = 0.0;
    for (int b = b0; b < b1; ++b) {
        const double *__restrict__ y = G.Ypad + (int64_t)b * G.Lpad + G.halo;
        const double res = dtw_short_pair<L, W>(x, y, G.p2, anan || (G.y_nan && G.y_nan[b]), delta, G.unfused);
        const float d = (float)res;   // the float32 distance distance_matrix_to returns (parallel_distances.py:59-67)
        const float t = G.pwr == 1 ? d : (G.pwr == 2 ? d * d : powf(d, (float)G.pwr));
        const double kv = (double)expf(G.ngamma * t);
        const double *__restrict__ cf = G.coefT + (int64_t)b * km1;   // uniform: scalar loads
        for (int q = 0; q < km1; ++q) sums[q][lane] += cf[q] * kv;
    }
    if (active) {
        double *__restrict__ Pq = G.P + ((int64_t)G.chunk_slot[blockIdx.y] * km1) * G.nA + a;
        for (int q = 0; q < km1; ++q) Pq[(int64_t)q * G.nA] = sums[q][lane];
    }
}

int launch_dtw_svm_partial(const double *X, int64_t nA, const double *Ypad_sv, int64_t Lpad, int halo, const uint8_t *y_nan_sv,
                           int64_t L, int window, double penalty, const double *coefT, const int32_t *chunk_ref0,
                           const int32_t *chunk_slot, int n_chunks, int km1, int pwr, float ngamma, double *P,
                           hipStream_t stream, int unfused) {
    if (nA == 0 || n_chunks == 0) return WDX_SUCCESS;
    if (L != 25 || window != 15 || km1 < 1 || km1 > 15) {
        set_error("the fused DTW + SVM path serves the shipped shape (25-point fingerprints, window 15, <= 16 classes)");
        return WDX_ERR_UNSUPPORTED;
    }
    DtwSvmArgs G{X, nA, Ypad_sv, Lpad, halo, y_nan_sv, penalty * penalty, coefT, chunk_ref0, chunk_slot, km1, pwr, ngamma, P, unfused};
    dim3 grid((unsigned)((nA + 63) / 64), (unsigned)n_chunks);
    hipLaunchKernelGGL((dtw_short_svm_kernel<25, 15>), grid, dim3(64), 0, stream, G);
    WDX_HIP_TRY(hipGetLastError());
    return WDX_SUCCESS;
}

// rows of `src` (n_src, ld) gathered by index into `dst` (n, ld), plus their byte flags: the resident references in
// support-vector order
__global__ void gather_rows_kernel(const double *__restrict__ src, const uint8_t *__restrict__ sflag, const int32_t *__restrict__ idx,
                                   int64_t n, int64_t ld, double *__restrict__ dst, uint8_t *__restrict__ dflag) {
    const int64_t r = blockIdx.x;
    const int64_t s = idx[r];
    for (int64_t i = threadIdx.x; i < ld; i += blockDim.x) dst[r * ld + i] = src[s * ld + i];
    if (threadIdx.x == 0 && dflag) dflag[r] = sflag ? sflag[s] : 0;
}
int launch_gather_rows(const double *src, const uint8_t *sflag, const int32_t *d_idx, int64_t n, int64_t ld, double *dst,
                       uint8_t *dflag, hipStream_t stream) {
    if (n == 0) return WDX_SUCCESS;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)n), dim3(64), 0, stream, src, sflag, d_idx, n, ld, dst, dflag);
    WDX_HIP_TRY(hipGetLastError());
    return WDX_SUCCESS;
}

// ---- layout helpers ----------------------------------------------------------------------------

__global__ void transpose_kernel(const double *__restrict__ src, int64_t n, int64_t L,
                                 double *__restrict__ dst, int64_t ld) {
    __shared__ double tile[32][33];
    const int64_t r0 = (int64_t)blockIdx.x * 32;  // series
    const int64_t c0 = (int64_t)blockIdx.y * 32;  // sample
    for (int k = threadIdx.y; k < 32; k += blockDim.y) {
        int64_t r = r0 + k, c = c0 + threadIdx.x;
        if (r < n && c < L) tile[k][threadIdx.x] = src[r * L + c];
    }
    __syncthreads();
    for (int k = threadIdx.y; k < 32; k += blockDim.y) {
        int64_t c = c0 + k, r = r0 + threadIdx.x;
        if (r < n && c < L) dst[c * ld + r] = tile[threadIdx.x][k];
    }
}

__global__ void nan_flags_T_kernel(const double *__restrict__ T, int64_t ld, int64_t n, int64_t L,
                                   uint8_t *__restrict__ flags) {
    int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n) return;
    bool f = false;
    for (int64_t i = 0; i < L; ++i) {
        double v = T[i * ld + a];
        f |= (v != v);
    }
    flags[a] = f ? 1 : 0;
}

int launch_transpose(const double *src, int64_t n, int64_t L, double *dstT, int64_t ld,
                     uint8_t *has_nan, hipStream_t stream) {
    if (n == 0) return WDX_SUCCESS;
    dim3 grid((unsigned)((n + 31) / 32), (unsigned)((L + 31) / 32));
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(32, 8), 0, stream, src, n, L, dstT, ld);
    if (has_nan)
        hipLaunchKernelGGL(nan_flags_T_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                           stream, dstT, ld, n, L, has_nan);
    WDX_HIP_TRY(hipGetLastError());
    return WDX_SUCCESS;
}

int launch_nan_flags_T(const double *T, int64_t ld, int64_t n, int64_t L, uint8_t *flags,
                       hipStream_t stream) {
    if (n == 0) return WDX_SUCCESS;
    hipLaunchKernelGGL(nan_flags_T_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                       T, ld, n, L, flags);
    WDX_HIP_TRY(hipGetLastError());
    return WDX_SUCCESS;
}

__global__ void pad_rows_kernel(const double *__restrict__ src, int64_t n, int64_t L,
                                double *__restrict__ dst, int64_t Lpad, int halo,
                                uint8_t *__restrict__ has_nan) {
    // one wave per series
    const int64_t r = (int64_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    const int lane = threadIdx.x & 63;
    if (r >= n) return;
    bool f = false;
    for (int64_t c = lane; c < Lpad; c += 64) {
        int64_t s = c - halo;
        double v = (s >= 0 && s < L) ? src[r * L + s] : 0.0;
        f |= (v != v);
        dst[r * Lpad + c] = v;
    }
    unsigned long long m = __ballot(f);
    if (has_nan && lane == 0) has_nan[r] = m ? 1 : 0;
}

int launch_pad_rows(const double *src, int64_t n, int64_t L, double *dst, int64_t Lpad, int halo,
                    uint8_t *has_nan, hipStream_t stream) {
    if (n == 0) return WDX_SUCCESS;
    hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream, src, n,
                       L, dst, Lpad, halo, has_nan);
    WDX_HIP_TRY(hipGetLastError());
    return WDX_SUCCESS;
}

// ---- B3: argmin + call histogram -----------------------------------------------------------------

__global__ void argmin_rows_kernel(const float *__restrict__ D, int64_t n, int64_t m,
                                   int32_t *__restrict__ out) {
    const int64_t r = (int64_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    const int lane = threadIdx.x & 63;
    if (r >= n) return;
    // key: (is_nan desc, value asc, index asc) -- np.argmin
    float bv = __builtin_huge_valf();
    int bi = 0x7fffffff;
    bool bn = false;
    for (int64_t c = lane; c < m; c += 64) {
        float v = D[r * m + c];
        bool isn = v != v;
        bool better = bn ? false : (isn ? true : (bi == 0x7fffffff || v < bv));
        if (better) {
            bv = v;
            bi = (int)c;
            bn = isn;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        float ov = __shfl_down(bv, off);
        int oi = __shfl_down(bi, off);
        int on = __shfl_down((int)bn, off);
        bool take;
        if (oi == 0x7fffffff) take = false;
        else if (bi == 0x7fffffff) take = true;
        else if (on != (int)bn) take = on != 0;
        else if (bn) take = oi < bi;
        else take = (ov < bv) || (ov == bv && oi < bi);
        if (take) {
            bv = ov;
            bi = oi;
            bn = on != 0;
        }
    }
    if (lane == 0) out[r] = bi == 0x7fffffff ? 0 : bi;
}

int launch_argmin(const float *D, int64_t n, int64_t m, int32_t *out, hipStream_t stream) {
    if (n == 0) return WDX_SUCCESS;
    hipLaunchKernelGGL(argmin_rows_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream, D,
                       n, m, out);
    WDX_HIP_TRY(hipGetLastError());
    return WDX_SUCCESS;
}

__global__ void count_calls_kernel(int32_t *__restrict__ call, const int32_t *__restrict__ status,
                                   int64_t n, int m, unsigned long long *__restrict__ counts) {
    extern __shared__ unsigned int hist[];
    const bool use_lds = m + 1 <= 4096;
    if (use_lds) {
        for (int k = threadIdx.x; k <= m; k += blockDim.x) hist[k] = 0;
        __syncthreads();
    }
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n;
         r += (int64_t)gridDim.x * blockDim.x) {
        int c = call[r];
        if (status && status[r] != 0) {
            c = -1;
            call[r] = -1;
        }
        int bin = (c < 0 || c >= m) ? m : c;
        if (counts) {
            if (use_lds) atomicAdd(&hist[bin], 1u);
            else atomicAdd(&counts[bin], 1ull);
        }
    }
    if (use_lds && counts) {
        __syncthreads();
        for (int k = threadIdx.x; k <= m; k += blockDim.x)
            if (hist[k]) atomicAdd(&counts[k], (unsigned long long)hist[k]);
    }
}

int launch_count_calls(int32_t *call, const int32_t *status, int64_t n, int64_t m, int64_t *counts,
                       hipStream_t stream) {
    if (n == 0) return WDX_SUCCESS;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    size_t lds = (m + 1 <= 4096) ? (size_t)(m + 1) * sizeof(unsigned int) : 0;
    hipLaunchKernelGGL(count_calls_kernel, dim3((unsigned)blocks), dim3(256), lds, stream, call,
                       status, n, (int)m, (unsigned long long *)counts);
    WDX_HIP_TRY(hipGetLastError());
    return WDX_SUCCESS;
}

}  // namespace wdx

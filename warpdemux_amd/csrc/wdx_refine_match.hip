// Consensus-guided barcode refinement behind the fast kernels: the subsequence match of a read's adapter event means
// against the consensus query (sig_proc.py:287-378, 452-521; dtaidistance subsequence alignment restated -- parity of
// that restatement: DESIGN.md 4.5), one WAVE per group of reads (round 4).
//
// What fp_refine_match<BLOCK> (wdx_fingerprint.hip, the exact kernel's own code) does with one workgroup per read --
// statistics, 205 anti-diagonal fronts of one thread per DP row with a barrier each, a single-thread argmin and
// back-trace: 30 k wave instructions per read, VALU-bound -- this kernel does with the same arithmetic, bit for bit, in
// a sixth of the instructions:
//   * statistics per read by the whole wave: NumPy's pairwise sums with its eight accumulators on eight lanes, the four
//     medians by a bitonic sort of two values per lane (np.median's value does not depend on how ties are ordered);
//   * the DP for up to three reads at once: a lane owns kRows = 4 consecutive query rows of one read (nq = 84 -> 21
//     lanes per read, 63 of the 64 lanes busy), lane l works on series column t - l at step t, so its upper neighbour's
//     last row arrives by one DPP shift per step and c + nq / 4 - 1 steps (141) replace the 205 fronts;
//   * no square root per cell: the back-trace's direction is the argmin of the three predecessors' ROOTS (first minimum),
//     and correctly rounded roots order like their arguments unless two arguments are within a few ulps of each other --
//     only then (top 32 bits within one of each other and a strict `<`) are the roots computed;
//   * arg-min of the last row in parallel, the (up to three) back-traces on three lanes side by side.
// Output: RefineRec::m / state exactly as fingerprint_refine_match_kernel leaves them.
#include "wdx_fp_types.h"
#include "wdx_wave.h"

namespace wdx {

namespace {

constexpr int kRows = 4;   // query rows per lane
constexpr int kMaxG = 3;   // reads per wave
static_assert(kRefineMaxSeries == 128, "two values per lane");

struct MwRead {
    double zz[kRefineMaxSeries];      // normalised series
    double lastD[kRefineMaxSeries];   // last DP row (squared costs)
    union {
        struct { double ev[kRefineMaxSeries], tmp[kRefineMaxSeries]; } s;   // statistics
        unsigned dirw[kRefineMaxQuery * (kRefineMaxSeries / 16)];           // 2-bit directions, 16 per word
    } u;
};

__device__ __forceinline__ double shfl_f64(double v, int src) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_ds_bpermute(src << 2, lo);
    hi = __builtin_amdgcn_ds_bpermute(src << 2, hi);
    return __hiloint2double(hi, lo);
}
// lane l <- lane l - 1 (lane 0 keeps its own): DPP wave_shr:1
__device__ __forceinline__ double wave_shr1_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// |difference of the top 32 bits| of two non-negative doubles
__device__ __forceinline__ unsigned hi_dist(double a, double b) {
    unsigned r;
    asm("v_sad_u32 %0, %1, %2, 0" : "=v"(r) : "v"(__double2hiint(a)), "v"(__double2hiint(b)));
    return r;
}
// NumPy's pairwise float64 sum of p[0..n), n <= 128 (np_pairwise_leaf in wdx_fingerprint.hip), by the whole wave: the
// eight accumulators run on lanes 0..7 (every lane runs accumulator lane & 7)
__device__ __forceinline__ double wave_pairwise_sum(const double *p, const int n, const int lane) {
    double res;
    if (n < 8) {
        res = 0.0;
        for (int i = 0; i < n; ++i) res += p[i];
        return res;
    }
    const int k = lane & 7;
    double rk = p[k];
    int i;
    for (i = 8; i < n - (n % 8); i += 8) rk += p[i + k];
    const double r0 = bcast_f64(rk, 0), r1 = bcast_f64(rk, 1), r2 = bcast_f64(rk, 2), r3 = bcast_f64(rk, 3),
                 r4 = bcast_f64(rk, 4), r5 = bcast_f64(rk, 5), r6 = bcast_f64(rk, 6), r7 = bcast_f64(rk, 7);
    res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; ++i) res += p[i];
    return res;
}
__device__ __forceinline__ double wave_min_f64(double v, const int lane) {
#pragma unroll
    for (int j = 1; j < 64; j <<= 1) v = vmin_f64(v, shfl_f64(v, lane ^ j));
    return v;
}

__global__ __launch_bounds__(64) void fingerprint_refine_match_wave_kernel(FpArgs A) {
    __shared__ MwRead sm[kMaxG];
    const int lane = threadIdx.x;
    const wdx_seg_params &P = A.p;
    const RefineDev &R = A.rf;
    const int K = P.barcode_num_events;
    const int nq = R.nq, c = P.num_events + 1, WPR = (c + 15) >> 4;
    const int LPR = (nq + kRows - 1) / kRows;   // lanes per read
    int G = 64 / LPR;
    if (G > kMaxG) G = kMaxG;
    const int64_t r0 = A.block_base + (int64_t)blockIdx.x * G;
    RefineRec *recs = reinterpret_cast<RefineRec *>(R.ws);
    const double inf = __builtin_huge_val();
    auto fail = [&](const int64_t r, const int st) {   // fp_refine_match's finish(st, false)
        for (int i = lane; i < K; i += 64) {
            if (A.fpt) A.fpt[r * K + i] = __builtin_nan("");
            if (A.dwell) A.dwell[r * K + i] = 0;
        }
        if (A.stats && lane < 6) A.stats[r * 6 + lane] = __builtin_nan("");
        if (R.idx && lane < 3) R.idx[r * 3 + lane] = -1;
        if (lane == 0) {
            A.status[r] = st;
            recs[r].state = 4;
        }
    };

    // ---- statistics and the normalised series, read by read ---------------------------------------------------------
    unsigned live = 0;
    for (int g = 0; g < G; ++g) {
        const int64_t r = r0 + g;
        if (r >= A.n_reads) break;
        RefineRec *rec = recs + r;
        if (rec->state != 1) continue;   // not segmented by a fast kernel: the exact kernel takes (or has reported) it
        MwRead &S = sm[g];
        const bool in0 = lane < c, in1 = lane + 64 < c;
        const double e0 = in0 ? rec->ev[lane] : inf, e1 = in1 ? rec->ev[lane + 64] : inf;
        const double d0 = in0 ? (double)(rec->cpts[lane + 1] - rec->cpts[lane]) : inf;
        const double d1 = in1 ? (double)(rec->cpts[lane + 65] - rec->cpts[lane + 64]) : inf;
        // normalize(series, method, accept_nan=False) inside _get_subseq_match raises on NaN -> "unknown"
        if (__any((in0 && e0 != e0) || (in1 && e1 != e1))) {
            fail(r, WDX_READ_FAIL_UNKNOWN);
            continue;
        }
        // adapter statistics (sig_proc.py:486-494)
        __syncthreads();
        S.u.s.ev[lane] = e0;
        S.u.s.ev[lane + 64] = e1;
        __syncthreads();
        const double mean = wave_pairwise_sum(S.u.s.ev, c, lane) / (double)c;
        {
            const double f0 = e0 - mean, f1 = e1 - mean;
            S.u.s.tmp[lane] = f0 * f0;
            S.u.s.tmp[lane + 64] = f1 * f1;
        }
        __syncthreads();
        const double sd = sqrt(wave_pairwise_sum(S.u.s.tmp, c, lane) / (double)c);
        const double dt_med = wave_median(d0, d1, c, lane);
        const double dt_mad = wave_median(in0 ? fabs(d0 - dt_med) : inf, in1 ? fabs(d1 - dt_med) : inf, c, lane);
        const double ev_med = wave_median(e0, e1, c, lane);
        const double ev_mad = wave_median(in0 ? fabs(e0 - ev_med) : inf, in1 ? fabs(e1 - ev_med) : inf, c, lane);
        // series of the match: normalize(adapter_event_means, consensus_subseq_match_normalization)
        double c0 = 0.0, c1 = 1.0;
        if (R.norm == WDX_NORM_MEAN) { c0 = mean; c1 = sd; }
        else if (R.norm == WDX_NORM_MEDIAN) { c0 = ev_med; c1 = ev_mad; }
        else if (R.norm != WDX_NORM_NONE) { fail(r, WDX_READ_FAIL_UNKNOWN); continue; }
        const double z0 = R.norm == WDX_NORM_NONE ? e0 : (e0 - c0) / c1, z1 = R.norm == WDX_NORM_NONE ? e1 : (e1 - c0) / c1;
        if (__any((in0 && z0 != z0) || (in1 && z1 != z1))) {   // a constant series (0/0): the library's NaN behaviour is not restated
            fail(r, WDX_READ_FAIL_UNKNOWN);
            continue;
        }
        S.zz[lane] = z0;
        S.zz[lane + 64] = z1;
        if (lane == 0) {
            double *m = rec->m;   // RefineMatch: mean, sd, ev_med, ev_mad, dt_med, dt_mad, {qs, qe}, {sbs, 0}
            m[0] = mean; m[1] = sd; m[2] = ev_med; m[3] = ev_mad; m[4] = dt_med; m[5] = dt_mad;
        }
        live |= 1u << g;
    }
    if (!live) return;
    __syncthreads();   // the statistics' arrays become the direction words

    // ---- subsequence DTW, all reads of the wave at once -------------------------------------------------------------
    {
        int g = lane / LPR;
        const int l = lane - g * LPR;
        const bool mine = g < G && ((live >> g) & 1u);
        if (g >= G) g = G - 1;
        MwRead &S = sm[g];
        const int i0 = kRows * l + 1;   // first of this lane's rows
        const double p2 = R.pen * R.pen;
        double q[kRows], lf[kRows];
        unsigned acc[kRows];
#pragma unroll
        for (int rr = 0; rr < kRows; ++rr) {
            q[rr] = i0 + rr <= nq ? R.query[i0 + rr - 1] : 0.0;
            lf[rr] = i0 + rr <= R.psi1b ? 0.0 : inf;   // column 0
            acc[rr] = 0;
        }
        double top_prev = l == 0 ? (0 <= R.psi2b ? 0.0 : inf) : (i0 - 1 <= R.psi1b ? 0.0 : inf);   // D(i0 - 1, 0)
        double last_out = inf;
        const int steps = c + LPR - 1;
        for (int t = 1; t <= steps; ++t) {
            const double nb = wave_shr1_f64(last_out);   // the upper neighbour's last row at this lane's column
            const int j = t - l;
            if (mine && j >= 1 && j <= c) {
                const double top_cur = l == 0 ? (j <= R.psi2b ? 0.0 : inf) : nb;
                const double z = S.zz[j - 1];
                const int shl = 2 * ((j - 1) & 15);
                double dg = top_prev, up = top_cur;
#pragma unroll
                for (int rr = 0; rr < kRows; ++rr) {
                    const double lfv = lf[rr];
                    double d = q[rr] - z;
                    d = d * d;
                    const double out = d + vmin_f64(vmin_f64(dg, up + p2), lfv + p2);
                    // direction: first minimum of (sqrt dg, sqrt up, sqrt lf)
                    const bool lt1 = up < dg;
                    const double m1 = vmin_f64(dg, up);
                    const bool lt2 = lfv < m1;
                    unsigned code = lt1 ? 1u : 0u;
                    code = lt2 ? 2u : code;
                    const bool close = (bool)((int)(lt1 & (hi_dist(dg, up) < 2u)) | (int)(lt2 & (hi_dist(m1, lfv) < 2u)));
                    if (__builtin_expect(close, 0)) {   // arguments within a few ulps: their roots may coincide
                        const double sdg = sqrt(dg), sup = sqrt(up), slf = sqrt(lfv);
                        code = 0;
                        double mv = sdg;
                        if (sup < mv) { mv = sup; code = 1; }
                        if (slf < mv) code = 2;
                    }
                    acc[rr] |= code << shl;
                    dg = lfv;
                    up = out;
                    lf[rr] = out;
                    if (i0 + rr == nq) S.lastD[j - 1] = out;
                }
                top_prev = top_cur;
                last_out = up;
                if (shl == 30 || j == c) {
#pragma unroll
                    for (int rr = 0; rr < kRows; ++rr) {
                        if (i0 + rr <= nq) S.u.dirw[(i0 + rr - 1) * WPR + ((j - 1) >> 4)] = acc[rr];
                        acc[rr] = 0;
                    }
                }
            }
        }
    }
    __syncthreads();

    // ---- best match (first minimum of sqrt(D[nq][j]) / nq) per read, then the back-traces side by side ----------------
    int best_g[kMaxG];
#pragma unroll
    for (int g = 0; g < kMaxG; ++g) {
        best_g[g] = 0;
        if (g < G && ((live >> g) & 1u)) {
            const MwRead &S = sm[g];
            const bool in0 = lane < c, in1 = lane + 64 < c;
            const double v0 = in0 ? sqrt(S.lastD[lane]) / (double)nq : inf;
            const double v1 = in1 ? sqrt(S.lastD[lane + 64]) / (double)nq : inf;
            double bv = v0;
            unsigned bi = in0 ? (unsigned)lane : 0x7fffffffu;
            if (in1 && v1 < bv) { bv = v1; bi = (unsigned)lane + 64u; }
            const double vm = wave_min_f64(bv, lane);
            best_g[g] = (int)wave_min_u32(bv == vm ? bi : 0x7fffffffu);
        }
    }
    {
        const int g = lane;
        const bool walk = g < G && ((live >> g) & 1u);
        int best = best_g[0];
#pragma unroll
        for (int k = 1; k < kMaxG; ++k) best = g == k ? best_g[k] : best;
        if (walk) {
            const MwRead &S = sm[g];
            int i = nq, j = best + 1, sj = j;
            while (i > 0 && j > 0) {
                sj = j;
                const unsigned code = (S.u.dirw[(i - 1) * WPR + ((j - 1) >> 4)] >> (2 * ((j - 1) & 15))) & 3u;
                if (code == 0) { --i; --j; }
                else if (code == 1) --i;
                else --j;
            }
            RefineRec *rec = recs + (r0 + g);
            int32_t *mi = reinterpret_cast<int32_t *>(rec->m + 6);
            mi[0] = sj - 1;           // seg_cons_query_start
            mi[1] = best;             // seg_cons_query_end
            mi[2] = rec->cpts[best];  // int(np.sum(adapter_dwell_times[:seg_query_end]))
            mi[3] = 0;
            rec->state = 3;
        }
    }
}

}  // namespace

int launch_refine_match_wave(FpArgs A, int64_t n, hipStream_t stream) {
    const int LPR = (A.rf.nq + kRows - 1) / kRows;
    int G = 64 / LPR;
    if (G > kMaxG) G = kMaxG;
    hipLaunchKernelGGL(fingerprint_refine_match_wave_kernel, dim3((unsigned)((n + G - 1) / G)), dim3(64), 0, stream, A);
    return WDX_SUCCESS;
}

}  // namespace wdx

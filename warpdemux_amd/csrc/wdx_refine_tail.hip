// Consensus-guided barcode refinement behind the fast kernels, last step: the barcode's own segmentation for a read whose
// subsequence match is done (RefineRec::state == 3) -- discrepenacy_curve_to_cpts(adapter_scores[sig_barcode_start:], ...)
// + compute_base_means + normalize_wrt + the outlier filter (sig_proc.py:308-378, 497-521) -- ONE WAVE per read (round 4).
//
// Same decisions and the same arithmetic, bit for bit, as fp_refine_finish / fp_segment (wdx_fingerprint.hip: the exact
// kernel's code, which fingerprint_refine_tail_kernel<256> runs with a 256-thread workgroup and ~20 barriers per read):
//   * the tail (<= 2 048 samples, ~1 100) is walked in tiles: 320 samples re-clipped with the recorded bounds into LDS, the
//     mean / squared deviations of 256 window starts once each (the exact kernel's operations), the scores of the
//     256 - W positions whose two windows the tile holds, the local maxima of its own 242 - W positions (scipy's rule: runs
//     of equal scores -- clipped stretches make them -- count at their midpoint) appended in position order by ballot /
//     prefix count;
//   * suppression on the list with per-peak masks of the outranking neighbours (fp_segment's rule), rounds separated by a
//     wave vote instead of a workgroup barrier;
//   * top-E of the survivors by rank counting over a dense key array; boundaries by ballot / prefix count;
//   * event means: eight lanes per segment when the sums are provably exact (positive samples within 2^16 of each other),
//     else one lane per segment in the reference's order; normalize_wrt, outlier filter, outputs.
// What it does not take goes back to the exact kernel, which refines the read in place (the hand-back list of
// launch_fingerprint): a run of equal scores that reaches more than kLook positions past a tile's end, more than kPeakCap (512) local maxima, a tail
// beyond kTailCap samples.  Parameters outside its range (refine_tail_wave_takes) keep the workgroup-per-read kernel.
#include "wdx_fp_types.h"
#include "wdx_wave.h"
#include <type_traits>

namespace wdx {

namespace {

constexpr int kWin = 256;        // window starts per tile (4 per lane)
constexpr int kLook = 12;        // scored positions past a tile's own: how far a run of equal scores may reach into the next tile
constexpr int kPeakCap = 512;    // local maxima of a tail (8 per lane); ~ns / 5.6 are typical (200 for 1 100 positions), ns / 4.5 happens
constexpr int kPeakPer = kPeakCap / 64;
constexpr int kSegMax = 127;     // segments of the barcode (E2 + 1)
enum : unsigned char { S_UNDECIDED = 1, S_KEPT = 2, S_DROPPED = 3, S_SELECTED = 4 };

struct TwLds {
    double Mt[kWin], Vt[kWin];            // window statistics of the tile; later: dense keys of the kept peaks (<= 512)
    union {
        double scl[kWin];                 // scores of the tile's positions t0 - 1 .. (index 0 = position t0 - 1)
        struct {                          // ... and once the peak list is complete:
            double ev[kSegMax + 1];
            int cpts[kSegMax + 2];
        } seg;
    } a;
    union {
        float sig[kWin + 64 + 8];         // the tile's clipped samples
        unsigned short dmap[kPeakCap];    // ... later: dense index -> list index
    } b;
    double pk_score[kPeakCap];
    unsigned short pk_pos[kPeakCap];
    unsigned char pk_st[kPeakCap];
};
static_assert(sizeof(TwLds) <= 13 * 1024 + 256, "twelve waves per CU");

__device__ __forceinline__ unsigned lanes_below(unsigned long long m) {   // set bits of m below this lane
    return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

__global__ __launch_bounds__(64) void fingerprint_refine_tail_wave_kernel(FpArgs A, unsigned *back_count, int32_t *back_list) {
    __shared__ TwLds S;
    const int lane = threadIdx.x;
    const int64_t r = A.block_base + blockIdx.x;
    if (r >= A.n_reads) return;
    RefineRec *rec = reinterpret_cast<RefineRec *>(A.rf.ws) + r;
    if (rec->state != 3) return;
    const wdx_seg_params &P = A.p;
    const RefineDev &R = A.rf;
    const int K = P.barcode_num_events, W = P.running_stat_width, d_eff = P.min_obs_per_base, E = R.E2;
    auto fail = [&](const int st, const bool with_stats) {   // fp_refine_finish's finish(st, with_stats), st != OK
        for (int i = lane; i < K; i += 64) {
            if (A.fpt) A.fpt[r * K + i] = __builtin_nan("");
            if (A.dwell) A.dwell[r * K + i] = 0;
        }
        if (!with_stats) {
            if (A.stats && lane < 6) A.stats[r * 6 + lane] = __builtin_nan("");
            if (R.idx && lane < 3) R.idx[r * 3 + lane] = -1;
        }
        if (lane == 0) A.status[r] = st;
    };
    auto hand_back = [&]() {   // the exact kernel redoes the read
        if (lane == 0) {
            rec->state = 2;
            back_list[atomicAdd(back_count, 1u)] = (int32_t)r;
        }
    };
    RefineMatch M;
    memcpy(&M, rec->m, sizeof(M));
    const int n = rec->n;
    const float lo = rec->lo, hi = rec->hi;
    const int sbs = M.sbs;
    const int ns = n - 2 * W;
    if (d_eff < 1) {   // scipy: `distance` must be >= 1
        fail(WDX_READ_FAIL_UNKNOWN, false);
        return;
    }
    const int nt = n - sbs;   // samples of the barcode tail
    if (nt > kTailCap || nt < 0) {
        hand_back();
        return;
    }
    if (ns - sbs < 0) {
        // the score curve ends before the barcode starts: no peak, "event segmentation failed" (fp_refine_finish: the
        // out-of-bounds case of compute_base_means comes after that return)
        fail(WDX_READ_FAIL_SEGMENT, false);
        return;
    }
    const int ns2 = ns - sbs, n_end2 = ns2 + 2 * W;   // (== nt)
    const int64_t row_off = A.row_off ? A.row_off[r] : r * A.stride;
    int64_t start = (int64_t)A.a_start[r] - P.padding;
    if (start < 0) start = 0;
    const float *__restrict__ src = A.sig + row_off + start + sbs;   // the tail's samples

    // ---- scores tile by tile, strict local maxima appended in position order -------------------------------------------
    const int nwin = ns2 > 0 ? ns2 + W : 0;   // window starts 0 .. ns2 + W - 1
    const int TP = kWin - W - 2 - kLook;      // owned positions per tile (W <= 64)
    const int NSC = kWin - W;                 // scored positions per tile: t0 - 1 .. t0 + TP + kLook
    int np = 0;
    bool plateau = false;
    // (a tile's samples are requested while the tile before it is being scored: index-clamped loads, five per lane)
    constexpr int kLd = (kWin + 64) / 64;
    float nxt[kLd];
    auto request = [&](const int w0) {
#pragma unroll
        for (int k = 0; k < kLd; ++k) nxt[k] = src[min(max(w0 + lane + 64 * k, 0), nt - 1)];
    };
    if (ns2 > 0) request(-1);
    auto tiles = [&](auto fast_t) __attribute__((always_inline)) {
    constexpr bool FASTM = decltype(fast_t)::value;
    for (int t0 = 0; t0 < ns2; t0 += TP) {
        const int w0 = t0 - 1;   // first window start / first scored position of the tile (-1: nothing there)
        __syncthreads();         // (the previous tile's readers are done)
#pragma unroll
        for (int k = 0; k < kLd; ++k) S.b.sig[lane + 64 * k] = __builtin_amdgcn_fmed3f(nxt[k], lo, hi);
        __syncthreads();
        if (t0 + TP < ns2) request(t0 + TP - 1);
#pragma unroll
        for (int k = 0; k < kWin / 64; ++k) {
            const int j = lane + 64 * k, q = w0 + j;
            if (q >= 0 && q < nwin) {
                double m, v;
                if (W == 18) window_stats<18, FASTM>(S.b.sig + j, W, m, v);
                else if (W == 12) window_stats<12, FASTM>(S.b.sig + j, W, m, v);
                else window_stats<0, FASTM>(S.b.sig + j, W, m, v);
                S.Mt[j] = m;
                S.Vt[j] = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kWin / 64; ++k) {
            const int j = lane + 64 * k, p = w0 + j;
            if (j < NSC && p >= 0 && p < ns2) {
                const double m1 = S.Mt[j], m2 = S.Mt[j + W];
                const double vs = S.Vt[j] + S.Vt[j + W];
                double sc;
                if constexpr (FASTM) {   // (rsq(0) = inf -> NaN all the way -> v_max_f64(NaN, 0) = 0: the reference's score of two flat windows)
                    sc = max0_f64(fast_div_mid(fabs(m1 - m2), fast_sqrt_mid(vs)));
                } else {
                    if (vs == 0) sc = 0.0;
                    else if (m1 > m2) sc = (m1 - m2) / sqrt(vs);
                    else sc = (m2 - m1) / sqrt(vs);
                }
                S.a.scl[j] = sc;
            }
        }
        __syncthreads();
        // find_peaks' local maxima among the tile's own positions (1 <= p <= ns2 - 2), scipy's _local_maxima_1d: a rising
        // edge, then the first different score must be lower; a run of equal scores (clipped stretches make them) is
        // recorded at its midpoint.  A run that leaves the tile's scores is the exact kernel's business
#pragma unroll
        for (int k = 0; k < kWin / 64; ++k) {
            const int j = 1 + lane + 64 * k, p = w0 + j;   // owned: j = 1 .. TP
            bool pk = false;
            double s = 0.0;
            int ppos = p;
            if (j <= TP && p >= 1 && p <= ns2 - 2) {
                s = S.a.scl[j];
                if (S.a.scl[j - 1] < s) {
                    int ja = j + 1;
                    while (ja < NSC && w0 + ja < ns2 - 1 && S.a.scl[ja] == s) ++ja;
                    if (ja >= NSC) plateau = true;
                    else if (S.a.scl[ja] < s) {
                        pk = true;
                        ppos = (p + (w0 + ja) - 1) / 2;
                    }
                }
            }
            const unsigned long long mk = __ballot(pk);
            const int o = np + (int)lanes_below(mk);
            if (pk && o < kPeakCap) {
                S.pk_pos[o] = (unsigned short)ppos;
                S.pk_score[o] = s;
                S.pk_st[o] = S_UNDECIDED;
            }
            np += __popcll(mk);
        }
    }
    };
    // positive clipped samples (every read a fast kernel segmented): the t-score's quotient and root without the range
    // scaling of the compiler's general float64 expansions -- the same bits (fast_div_mid / fast_sqrt_mid, wdx_fp_types.h)
    if (lo > 0.f && hi < 3.0e38f) tiles(std::true_type{});
    else tiles(std::false_type{});
    if (__any(plateau) || np > kPeakCap || d_eff > 17) {
        if (lane == 0) atomicAdd(back_count + (np > kPeakCap ? 2 : 1), 1u);   // (diagnostic counters [6] plateau, [7] list capacity)
        hand_back();
        return;
    }
    __syncthreads();

    // ---- greedy suppression by priority (fp_segment: a peak's rivals are the list neighbours closer than d_eff; ties ->
    // the later peak outranks; only the rivals that outrank a peak decide about it) ----------------------------------
    {
        unsigned hp[kPeakPer];
        bool und[kPeakPer];
#pragma unroll
        for (int m = 0; m < kPeakPer; ++m) {
            const int k = lane + 64 * m;
            hp[m] = 0;
            und[m] = k < np;
            if (k < np) {
                const int p = S.pk_pos[k];
                const double s = S.pk_score[k];
                for (int q = k - 1; q >= 0 && p - (int)S.pk_pos[q] < d_eff; --q)
                    if (S.pk_score[q] > s) hp[m] |= 1u << (q - k + 8);
                for (int q = k + 1; q < np && (int)S.pk_pos[q] - p < d_eff; ++q) {
                    const double sq = S.pk_score[q];
                    if (sq > s || sq == s) hp[m] |= 1u << (q - k + 7);
                }
            }
        }
        for (;;) {
            int pending = 0;
#pragma unroll
            for (int m = 0; m < kPeakPer; ++m) {
                if (!und[m]) continue;
                const int k = lane + 64 * m;
                bool kept_near = false, wait = false;
                unsigned mask = hp[m];
                while (mask) {
                    const int b = __ffs((int)mask) - 1;
                    mask &= mask - 1;
                    const unsigned char st = S.pk_st[k + (b < 8 ? b - 8 : b - 7)];
                    kept_near |= st == S_KEPT;
                    wait |= st == S_UNDECIDED;
                }
                if (kept_near) { S.pk_st[k] = S_DROPPED; und[m] = false; }
                else if (!wait) { S.pk_st[k] = S_KEPT; und[m] = false; }
                else pending = 1;
            }
            __syncthreads();
            if (!__any(pending)) break;
        }
    }

    // ---- keep the E highest (sig_proc.py:185-188): dense keys of the survivors in list order, rank counting -------------
    int nk = 0;
    unsigned long long *dk = reinterpret_cast<unsigned long long *>(S.Mt);   // 512 doubles: Mt + Vt
    {
#pragma unroll
        for (int m = 0; m < kPeakPer; ++m) {
            const int k = lane + 64 * m;
            const bool kept = k < np && S.pk_st[k] == S_KEPT;
            const unsigned long long mk = __ballot(kept);
            if (kept) {
                const int o = nk + (int)lanes_below(mk);
                dk[o] = (unsigned long long)__double_as_longlong(S.pk_score[k]);
                S.b.dmap[o] = (unsigned short)k;
            }
            nk += __popcll(mk);
        }
    }
    if (nk < E) {   // (nk == 0 -> "unknown" cannot be reached: E >= 1)
        fail(WDX_READ_FAIL_SEGMENT, false);
        return;
    }
    __syncthreads();
    const int nsel = E;   // nk >= E
    for (int t = lane; t < nk; t += 64) {
        bool sel = true;
        if (nk > E) {
            const unsigned long long key = dk[t];
            int rank = 0;
            for (int j = 0; j < nk; ++j) {
                const unsigned long long kj = dk[j];
                rank += (kj > key) || (kj == key && j > t);   // equal scores: the later peak outranks (stable order)
            }
            sel = rank < E;
        }
        if (sel) S.pk_st[S.b.dmap[t]] = S_SELECTED;
    }
    __syncthreads();
    // ---- boundaries 0, peaks + W (ascending), n_end (sig_proc.py:188-196) -----------------------------------------------
    {
        int o = 1;
#pragma unroll
        for (int m = 0; m < kPeakPer; ++m) {
            const int k = lane + 64 * m;
            const bool sel = k < np && S.pk_st[k] == S_SELECTED;
            const unsigned long long mk = __ballot(sel);
            if (sel) S.a.seg.cpts[o + (int)lanes_below(mk)] = (int)S.pk_pos[k] + W;
            o += __popcll(mk);
        }
        if (lane == 0) {
            S.a.seg.cpts[0] = 0;
            S.a.seg.cpts[nsel + 1] = n_end2;
        }
    }
    const int nseg2 = nsel + 1;
    __syncthreads();

    // ---- event means (_c_segmentation.pyx:41-53): sequential float64 sums; every partial sum of positive float32 samples
    // within a factor 2^16 of each other (<= 2 048 of them) is exactly representable, so then the order is free ----------
    if (lo > 0.f && hi <= lo * 65536.f) {
        const int j = lane & 7;
        for (int s0 = 0; s0 < nseg2; s0 += 8) {
            const int s = s0 + (lane >> 3);
            double sum = 0.0;
            int b = 0, e = 1;
            if (s < nseg2) {
                b = S.a.seg.cpts[s];
                e = S.a.seg.cpts[s + 1];
                for (int i0 = b + j; i0 < e; i0 += 64) {   // eight loads in flight per lane: one memory latency per 64 samples
                    float x[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) x[u] = src[min(i0 + 8 * u, e - 1)];
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (i0 + 8 * u < e) sum += (double)__builtin_amdgcn_fmed3f(x[u], lo, hi);
                }
            }
            sum += __shfl_xor(sum, 4);
            sum += __shfl_xor(sum, 2);
            sum += __shfl_xor(sum, 1);
            if (s < nseg2 && j == 0) S.a.seg.ev[s] = sum / (double)(e - b);
        }
    } else {
        for (int s = lane; s < nseg2; s += 64) {
            const int b = S.a.seg.cpts[s], e = S.a.seg.cpts[s + 1];
            double sum = 0.0;
            for (int i = b; i < e; ++i) sum += (double)__builtin_amdgcn_fmed3f(src[i], lo, hi);
            S.a.seg.ev[s] = sum / (double)(e - b);
        }
    }
    __syncthreads();

    // ---- normalize_wrt(barcode_event_means, adapter_event_means, segmentation.normalization), the outlier filter, outputs
    // (fp_refine_finish) ------------------------------------------------------------------------------------------------
    double shift, scale;
    if (P.seg_norm == WDX_NORM_MEAN) { shift = M.mean; scale = M.sd; }
    else if (P.seg_norm == WDX_NORM_MEDIAN) { shift = M.ev_med; scale = M.ev_mad; }
    else { fail(WDX_READ_FAIL_UNKNOWN, false); return; }   // "none" is not a normalize_wrt method: ValueError
    const bool outlier = M.qs > R.ub_start || M.qe < R.lb_end || M.qe > R.ub_end;
    if (!outlier && nseg2 < K) {
        fail(WDX_READ_FAIL_UNKNOWN, false);
        return;
    }
    if (lane == 0) {
        if (A.stats) {
            double *o = A.stats + r * 6;
            o[0] = M.dt_med; o[1] = M.dt_mad; o[2] = M.mean; o[3] = M.sd; o[4] = M.ev_med; o[5] = M.ev_mad;
        }
        if (R.idx) {
            R.idx[r * 3] = M.qs; R.idx[r * 3 + 1] = M.qe; R.idx[r * 3 + 2] = sbs;
        }
    }
    if (outlier) {
        fail(WDX_READ_FAIL_CONSENSUS, true);   // "consensus query outlier": stats and indices are reported
        return;
    }
    for (int i = lane; i < K; i += 64) {
        const int s = nseg2 - K + i;
        if (A.fpt) A.fpt[r * K + i] = (S.a.seg.ev[s] - shift) / scale;
        if (A.dwell) A.dwell[r * K + i] = (int64_t)(S.a.seg.cpts[s + 1] - S.a.seg.cpts[s]);
    }
    if (lane == 0) A.status[r] = WDX_READ_OK;
}

}  // namespace

bool refine_tail_wave_takes(const FpArgs &A) {
    return A.rf.E2 >= 1 && A.rf.E2 + 1 <= kSegMax && A.p.running_stat_width >= 1 && A.p.running_stat_width <= 64 &&
           A.p.min_obs_per_base <= 17;
}

int launch_refine_tail_wave(FpArgs A, int64_t n, unsigned *back_count, int32_t *back_list, hipStream_t stream) {
    hipLaunchKernelGGL(fingerprint_refine_tail_wave_kernel, dim3((unsigned)n), dim3(64), 0, stream, A, back_count, back_list);
    return WDX_SUCCESS;
}

}  // namespace wdx

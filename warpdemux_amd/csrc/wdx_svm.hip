// N1: classifier tail of DTW_SVM.predict on the device --
//   K = exp(-gamma * d^p)              /root/reference/warpdemux/models/dtw_svm.py:21-22, 90
//   SVC.predict_proba(K)               dtw_svm.py:92  (libsvm svm_predict_probability, precomputed kernel)
//   process_probs                      models/utils.py:45-61 (argmax, label map, top1-top2 margin, thresholds)
// so that the (nX, nY<=3617) distance matrix never has to leave HBM.
//
// One 256-thread workgroup per read: kernel values of the support vectors in LDS, the k(k-1)/2
// one-vs-one decision values by wave-level reductions, then Platt sigmoids and libsvm's pairwise-coupling
// fixed point on one 16-lane group (k <= 16).  float64 throughout except the kernel value itself,
// which the reference computes in float32 (np.exp of a float32 array).  Sums are tree-ordered, so
// probabilities agree with scikit-learn to ~1e-12 given the same kernel values; expf differs from
// NumPy's SIMD exp by <= 1 ulp(float32), i.e. ~1e-7 relative in K (tolerances in tests/).
#include "wdx_common.h"

#include <math.h>
#include <stdlib.h>

namespace wdx {

__device__ __forceinline__ double wave_sum_f64(double v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__global__ __launch_bounds__(256) void svm_predict_kernel(SvmDev M, const float *__restrict__ dist,
                                                          int64_t n, double *__restrict__ prob,
                                                          int32_t *__restrict__ pred,
                                                          double *__restrict__ conf) {
    extern __shared__ double svm_lds[];  // kv[n_sv] | dec[npairs] | pw[k*k] | Q[k*k] | p[k] | Qp[k]
    const int k = M.k, npairs = k * (k - 1) / 2;
    double *kv = svm_lds;
    double *dec = kv + M.n_sv;
    double *pw = dec + npairs;
    double *Q = pw + k * k;
    volatile double *pp = Q + k * k;  // shared between the lanes of wave 0 inside the coupling loop:
    volatile double *Qp = pp + k;     // volatile keeps every LDS access in program order
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t r = blockIdx.x;
    const float *drow = dist + r * M.n_train;

    // kernel values of the support vectors (float32 like the reference, then widened)
    for (int s = tid; s < M.n_sv; s += 256) {
        const float d = drow[M.support[s]];
        const float t = M.pwr == 1 ? d : (M.pwr == 2 ? d * d : powf(d, (float)M.pwr));
        kv[s] = (double)expf(M.ngamma * t);
    }
    __syncthreads();

    // one-vs-one decision values: pair index p enumerates (i<j) row-major; waves take pairs round-robin
    for (int p = wave; p < npairs; p += 4) {
        int i = 0, rem = p;
        while (rem >= k - 1 - i) {
            rem -= k - 1 - i;
            ++i;
        }
        const int j = i + 1 + rem;
        const int si = M.start[i], sj = M.start[j], ci = M.n_support[i], cj = M.n_support[j];
        const double *coef1 = M.dual_coef + (size_t)(j - 1) * M.n_sv, *coef2 = M.dual_coef + (size_t)i * M.n_sv;
        double sum = 0.0;
        for (int q = lane; q < ci; q += 64) sum += coef1[si + q] * kv[si + q];
        for (int q = lane; q < cj; q += 64) sum += coef2[sj + q] * kv[sj + q];
        sum = wave_sum_f64(sum);
        if (lane == 0) dec[p] = sum - M.rho[p];
    }
    __syncthreads();

    // Platt sigmoids -> pairwise probabilities
    for (int p = tid; p < npairs; p += 256) {
        int i = 0, rem = p;
        while (rem >= k - 1 - i) {
            rem -= k - 1 - i;
            ++i;
        }
        const int j = i + 1 + rem;
        const double fApB = dec[p] * M.probA[p] + M.probB[p];
        double v = fApB >= 0 ? exp(-fApB) / (1.0 + exp(-fApB)) : 1.0 / (1.0 + exp(fApB));
        v = fmin(fmax(v, 1e-7), 1.0 - 1e-7);
        pw[i * k + j] = v;
        pw[j * k + i] = 1.0 - v;
    }
    __syncthreads();

    // libsvm multiclass_probability on lanes 0..k-1 of wave 0 (lane t owns row t of Q)
    if (wave == 0) {
        const int t = lane;
        const bool on = t < k;
        if (on) {
            double qtt = 0.0;
            for (int j = 0; j < k; ++j)
                if (j != t) qtt += pw[j * k + t] * pw[j * k + t];
            for (int j = 0; j < k; ++j) Q[t * k + j] = (j == t) ? qtt : -pw[j * k + t] * pw[t * k + j];
            pp[t] = 1.0 / k;
        }
        __builtin_amdgcn_wave_barrier();
        const double eps = 0.005 / k;
        const int max_iter = k > 100 ? k : 100;
        for (int iter = 0; iter < max_iter; ++iter) {
            // recompute Qp and pQp (sequential order over j as in libsvm)
            double qp = 0.0;
            if (on)
                for (int j = 0; j < k; ++j) qp += Q[t * k + j] * pp[j];
            if (on) Qp[t] = qp;
            __builtin_amdgcn_wave_barrier();
            double pQp = 0.0;
            for (int j = 0; j < k; ++j) pQp += pp[j] * Qp[j];  // every lane the same sum, same order
            double err = on ? fabs(qp - pQp) : 0.0;
            for (int off = 8; off > 0; off >>= 1) err = fmax(err, __shfl_xor(err, off));
            err = __shfl(err, 0);
            if (err < eps) break;
            for (int u = 0; u < k; ++u) {
                const double Quu = Q[u * k + u];
                const double diff = (-Qp[u] + pQp) / Quu;
                __builtin_amdgcn_wave_barrier();
                if (on) {
                    double pt = pp[t];
                    if (t == u) pt += diff;
                    pQp = (pQp + diff * (diff * Quu + 2 * Qp[u])) / (1 + diff) / (1 + diff);
                    const double qpn = (Qp[t] + diff * Q[u * k + t]) / (1 + diff);
                    pt /= (1 + diff);
                    __builtin_amdgcn_wave_barrier();
                    Qp[t] = qpn;
                    pp[t] = pt;
                } else {
                    pQp = (pQp + diff * (diff * Quu + 2 * Qp[u])) / (1 + diff) / (1 + diff);
                    __builtin_amdgcn_wave_barrier();
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (on && prob) prob[r * k + t] = pp[t];
        if (lane == 0) {
            // process_probs: np.argmax (first maximum), margin top1 - top2, per-class threshold
            int best = 0;
            double b1 = pp[0], b2 = -1.0;
            for (int j = 1; j < k; ++j) {
                const double v = pp[j];
                if (v > b1) {
                    b2 = b1;
                    b1 = v;
                    best = j;
                } else if (v > b2) {
                    b2 = v;
                }
            }
            const double margin = b1 - b2;
            int label = M.label_map ? M.label_map[best] : best;
            if (M.thresholds && margin < M.thresholds[best]) label = -1;
            if (pred) pred[r] = label;
            if (conf) conf[r] = margin;
        }
    }
}

// ---- matrix-core variant (every supported model: k <= 16) ---------------------------------------------------------------
// The decision values are a dense contraction: for class c and coefficient row q,
//   P[r][q][c] = sum over the support vectors s of class c of dual_coef[q][s] * K[r][s],
// and the one-vs-one value of the pair (i < j) is P[j-1][i] + P[i][j] - rho.  One wave takes 16 reads:
// A = kernel values (16 reads x 4 support vectors, made on the fly: one float32 exp per lane and step),
// B = dual_coef^T (4 support vectors x 16 rows, zero-padded), v_mfma_f64_16x16x4_f64 accumulates class by
// class, so dual_coef is streamed once per 16 reads instead of once per read and every kernel value is
// computed exactly once.  Each accumulator entry belongs to exactly one pair; the two contributions of a
// pair arrive in different class passes of the same wave, so they are added into LDS without atomics.
// Platt + coupling then run as in svm_predict_kernel, four reads at a time on the wave's four 16-lane
// groups.  (Summation order differs from libsvm's; the float32 kernel values bound the agreement with
// scikit-learn at ~1e-7 relative anyway, tolerances in tests/.)
typedef double wdx_d4 __attribute__((ext_vector_type(4)));

// FROMP: the decision sums come from the DTW kernel's epilogue (dtw_short_svm_kernel: Psum[slot][q][read], slot =
// class * halves + half) instead of the distance matrix; phase 1 then only adds them up.
template <bool PWR1, bool FROMP = false>  // pwr_dist == 1 (every shipped model): no powf in the inner loop
__global__ __launch_bounds__(64) void svm_predict_mfma_kernel(SvmDev M, const float *__restrict__ dist,
                                                              int64_t n, double *__restrict__ prob,
                                                              int32_t *__restrict__ pred,
                                                              double *__restrict__ conf,
                                                              const double *__restrict__ Psum = nullptr, int halves = 0) {
    extern __shared__ double svm_lds[];  // dec[16][npairs] | 4 x pw[k*k]
    const int k = M.k, npairs = k * (k - 1) / 2;
    const int lane = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * 16;
    double *dec = svm_lds;
    double *grp_base = dec + 16 * npairs;

    for (int idx = lane; idx < 16 * npairs; idx += 64) dec[idx] = -M.rho[idx % npairs];
    __builtin_amdgcn_wave_barrier();

    if constexpr (FROMP) {
        // value of pair (i < j) = P[q = j - 1][class i] + P[q = i][class j] - rho  (svm.cpp svm_predict_values), each P the
        // sum of the class's halves in order
        for (int idx = lane; idx < 16 * npairs; idx += 64) {
            const int rl = idx / npairs, p = idx - rl * npairs;
            int i = 0, rem = p;
            while (rem >= k - 1 - i) {
                rem -= k - 1 - i;
                ++i;
            }
            const int j = i + 1 + rem;
            const int64_t rr = r0 + rl < n ? r0 + rl : n - 1;
            double a = 0.0, b = 0.0;
            for (int h = 0; h < halves; ++h) {
                a += Psum[((int64_t)(i * halves + h) * (k - 1) + (j - 1)) * n + rr];
                b += Psum[((int64_t)(j * halves + h) * (k - 1) + i) * n + rr];
            }
            dec[idx] += a + b;
        }
        __builtin_amdgcn_wave_barrier();
    } else
    // ---- phase 1: decision values through the matrix cores
    {
        const int ar = lane & 15, kk = lane >> 4;  // A: read ar, B: coefficient row ar; both: support vector kk of the step
        const int64_t rr = r0 + ar < n ? r0 + ar : n - 1;
        const float *drow = dist + rr * M.n_train;
        const bool qrow = ar < k - 1;
        const double *crow = M.dual_coef + (size_t)(qrow ? ar : 0) * M.n_sv;
        for (int c = 0; c < k; ++c) {
            const int s0 = M.start[c], cn = M.n_support[c];
            wdx_d4 acc = {0.0, 0.0, 0.0, 0.0};
            // 64 support vectors (16 matrix steps) at a time: their training-set indices by one coalesced
            // load + lane shuffles, then all 16 distance loads and all 16 coefficient loads of the lane in
            // flight together -- the loop is otherwise a chain of two dependent global loads per step
            for (int cb = 0; cb < cn; cb += 64) {
                const int rem = cn - cb;
                const int sup = M.support[s0 + (lane < rem ? cb + lane : 0)];
                float dv[16];
                double bv[16];
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const int l = 4 * t + kk;
                    const int sidx = __shfl(sup, l);
                    dv[t] = drow[sidx];
                    bv[t] = crow[s0 + (l < rem ? cb + l : 0)];
                }
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    if (4 * t < rem) {  // wave-uniform
                        const bool ok = 4 * t + kk < rem;
                        const float d = dv[t];
                        const float tt = PWR1 ? d : (M.pwr == 2 ? d * d : powf(d, (float)M.pwr));
                        const double a = ok ? (double)expf(M.ngamma * tt) : 0.0;
                        const double b = (ok && qrow) ? bv[t] : 0.0;
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
                    }
                }
            }
            // acc[reg] = P[read kk + 4 reg][q = ar][c]  ->  its pair
            if (qrow) {
                const int q = ar;
                const int i = q < c ? q : c, j = q < c ? c : q + 1;
                const int p = i * (k - 1) - i * (i - 1) / 2 + (j - i - 1);
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) dec[(kk + 4 * reg) * npairs + p] += acc[reg];
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");

    // ---- phase 2: Platt sigmoids + libsvm's pairwise coupling, four reads at a time (16 lanes each).
    // Lane t of a group owns row t of Q in REGISTERS (Q is symmetric, so Q[u][t] of the update step is
    // its own element u) together with Qp[t] and p[t]; what the sweep needs from other lanes (Qp[u], p[j])
    // comes by lane shuffle, the diagonal is replicated once per read.  No LDS round trip and no barrier
    // inside the iteration: it is a chain of ~k shuffles + divisions per sweep, and with only the decision
    // values and the pairwise table left in LDS more than twice as many waves fit a CU to hide it.
    const int g = lane >> 4, t = lane & 15, gbase = lane & 48;
    double *pw = grp_base + g * (k * k);
    const bool on = t < k;
    for (int rnd = 0; rnd < 4; ++rnd) {
        const int rl = rnd * 4 + g;  // read of this group within the tile
        const int64_t r = r0 + rl;
        const double *dr = dec + rl * npairs;
        for (int p = t; p < npairs; p += 16) {
            int i = 0, rem = p;
            while (rem >= k - 1 - i) {
                rem -= k - 1 - i;
                ++i;
            }
            const int j = i + 1 + rem;
            const double fApB = dr[p] * M.probA[p] + M.probB[p];
            double v = fApB >= 0 ? exp(-fApB) / (1.0 + exp(-fApB)) : 1.0 / (1.0 + exp(fApB));
            v = fmin(fmax(v, 1e-7), 1.0 - 1e-7);
            pw[i * k + j] = v;
            pw[j * k + i] = 1.0 - v;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        double qrow[16];  // (the diagonal element of the sweep step comes by shuffle: registers are what limits the wave count)
        {
            double qtt = 0.0;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const bool in = on && j < k && j != t;
                const double a = in ? pw[j * k + t] : 0.0;  // r[j][t]
                const double b = in ? pw[t * k + j] : 0.0;  // r[t][j]
                qtt += a * a;
                qrow[j] = -a * b;
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) qrow[j] = (j == t) ? qtt : qrow[j];
        }
        __builtin_amdgcn_wave_barrier();  // pw is rewritten for the next read only after every lane has its row
        double pt = 1.0 / k, qp = 0.0;
        const double eps = 0.005 / k;
        const int max_iter = k > 100 ? k : 100;
        bool live = true;  // per 16-lane group; the wave keeps sweeping until every group has converged
        for (int iter = 0; iter < max_iter; ++iter) {
            // Qp[t] = sum_j Q[t][j] p[j] and pQp = sum_t p[t] Qp[t], both in libsvm's index order
            double qn = 0.0;
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (j < k) qn += qrow[j] * __shfl(pt, gbase + j);
            if (live) qp = qn;
            const double pq = pt * qp;
            double pQp = 0.0;
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (j < k) pQp += __shfl(pq, gbase + j);
            double err = on ? fabs(qp - pQp) : 0.0;
            for (int off = 8; off > 0; off >>= 1) err = fmax(err, __shfl_xor(err, off));
            if (err < eps) live = false;
            if (!__ballot(live)) break;
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                if (u < k) {  // wave-uniform
                    // libsvm divides three times by (1 + diff); one reciprocal here, which moves the iterates
                    // by ~1e-16
                    const double Qpu = __shfl(qp, gbase + u);
                    const double Quu = __shfl(qrow[u], gbase + u);
                    const double diff = (-Qpu + pQp) / Quu;
                    const double inv = 1.0 / (1 + diff);
                    double pn = pt;
                    if (t == u) pn += diff;
                    pn *= inv;
                    const double qpn = (qp + diff * qrow[u]) * inv;
                    pQp = (pQp + diff * (diff * Quu + 2 * Qpu)) * inv * inv;
                    if (live) {
                        pt = pn;
                        qp = qpn;
                    }
                }
            }
        }
        // process_probs: np.argmax (first maximum), margin top1 - top2, per-class threshold
        int best = 0;
        double b1 = __shfl(pt, gbase), b2 = -1.0;
#pragma unroll
        for (int j = 1; j < 16; ++j) {
            if (j < k) {
                const double v = __shfl(pt, gbase + j);
                if (v > b1) {
                    b2 = b1;
                    b1 = v;
                    best = j;
                } else if (v > b2) {
                    b2 = v;
                }
            }
        }
        if (r < n) {
            if (on && prob) prob[r * k + t] = pt;
            if (t == 0) {
                const double margin = b1 - b2;
                int label = M.label_map ? M.label_map[best] : best;
                if (M.thresholds && margin < M.thresholds[best]) label = -1;
                if (pred) pred[r] = label;
                if (conf) conf[r] = margin;
            }
        }
    }
}

// reads whose fingerprint failed carry NaN distances: the reference never shows them to the model
// (file_proc.py:430-450 classifies the successful reads only) -> pred -1, probabilities / margin NaN
__global__ void svm_mask_failed_kernel(const int32_t *__restrict__ status, int64_t n, int k, double *prob, int32_t *pred,
                                       double *conf) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n || status[r] == WDX_READ_OK) return;
    if (pred) pred[r] = -1;
    if (conf) conf[r] = __builtin_nan("");
    if (prob)
        for (int c = 0; c < k; ++c) prob[r * k + c] = __builtin_nan("");
}
int launch_svm_mask_failed(const int32_t *d_status, int64_t n, int k, double *d_prob, int32_t *d_pred, double *d_conf,
                           hipStream_t stream) {
    if (n == 0) return WDX_SUCCESS;
    hipLaunchKernelGGL(svm_mask_failed_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_status, n, k, d_prob,
                       d_pred, d_conf);
    WDX_HIP_TRY(hipGetLastError());
    return WDX_SUCCESS;
}

// sigmoids + coupling + process_probs from the decision sums the DTW epilogue left (dtw_short_svm_kernel)
int launch_svm_finish(const SvmDev &M, const double *d_P, int halves, int64_t n, double *d_prob, int32_t *d_pred,
                      double *d_conf, hipStream_t stream) {
    if (n == 0) return WDX_SUCCESS;
    const int k = M.k;
    if (k < 2 || k > 16) {
        set_error("the fused SVM tail takes 2..16 classes");
        return WDX_ERR_UNSUPPORTED;
    }
    const size_t lds2 = sizeof(double) * ((size_t)16 * k * (k - 1) / 2 + 4 * (size_t)k * k);
    void (*kern)(SvmDev, const float *, int64_t, double *, int32_t *, double *, const double *, int) =
        svm_predict_mfma_kernel<true, true>;
    static LdsAttr attr;
    if (int rc = attr.ensure(kern, lds2)) return rc;
    const int64_t tiles = (n + 15) / 16;
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(64), lds2, stream, M, (const float *)nullptr, n, d_prob, d_pred, d_conf,
                       d_P, halves);
    WDX_HIP_TRY(hipGetLastError());
    return WDX_SUCCESS;
}

int launch_svm_predict(const SvmDev &M, const float *d_dist, int64_t n, double *d_prob, int32_t *d_pred,
                       double *d_conf, hipStream_t stream, const Knobs &knobs) {
    if (n == 0) return WDX_SUCCESS;
    const int k = M.k;
    if (k >= 2 && k <= 16 && !knobs.svm_scalar) {
        const size_t lds2 = sizeof(double) * ((size_t)16 * k * (k - 1) / 2 + 4 * (size_t)k * k);
        void (*kern)(SvmDev, const float *, int64_t, double *, int32_t *, double *, const double *, int) =
            M.pwr == 1 ? svm_predict_mfma_kernel<true, false> : svm_predict_mfma_kernel<false, false>;
        static LdsAttr attr[2];
        if (int rc = attr[M.pwr == 1].ensure(kern, lds2)) return rc;
        const int64_t tiles = (n + 15) / 16;
        const int64_t slice = (1ll << 31) / 64;
        for (int64_t base = 0; base < tiles; base += slice) {
            const int64_t m = tiles - base < slice ? tiles - base : slice;
            const int64_t rb = base * 16, rn = n - rb < m * 16 ? n - rb : m * 16;
            hipLaunchKernelGGL(kern, dim3((unsigned)m), dim3(64), lds2, stream, M,
                               d_dist + rb * M.n_train, rn, d_prob ? d_prob + rb * k : nullptr,
                               d_pred ? d_pred + rb : nullptr, d_conf ? d_conf + rb : nullptr, (const double *)nullptr, 0);
        }
        WDX_HIP_TRY(hipGetLastError());
        return WDX_SUCCESS;
    }
    const size_t lds = sizeof(double) * ((size_t)M.n_sv + (size_t)k * (k - 1) / 2 + 2 * (size_t)k * k + 2 * (size_t)k);
    if (lds > 150 * 1024) {
        set_error("SVM model too large for the LDS carve-up (%zu B)", lds);
        return WDX_ERR_UNSUPPORTED;
    }
    static LdsAttr attr_scalar;
    if (int rc = attr_scalar.ensure(svm_predict_kernel, lds)) return rc;
    const int64_t slice = (1ll << 31) / 256;
    for (int64_t base = 0; base < n; base += slice) {
        const int64_t m = n - base < slice ? n - base : slice;
        hipLaunchKernelGGL(svm_predict_kernel, dim3((unsigned)m), dim3(256), lds, stream, M,
                           d_dist + base * M.n_train, m, d_prob ? d_prob + base * k : nullptr,
                           d_pred ? d_pred + base : nullptr, d_conf ? d_conf + base : nullptr);
    }
    WDX_HIP_TRY(hipGetLastError());
    return WDX_SUCCESS;
}

}  // namespace wdx

// Wave-level primitives on DPP shared by the fingerprint translation units.  Internal.
#pragma once
#include <hip/hip_runtime.h>

namespace wdx {

// ---- wave primitives on DPP (row_shr 1/2/4/8 + row_bcast 15/31): 6 VALU ops per wave scan ----------
#define WDX_DPP(old, x, ctrl, rmask) \
    (unsigned)__builtin_amdgcn_update_dpp((int)(old), (int)(x), (ctrl), (rmask), 0xf, false)

__device__ __forceinline__ unsigned wave_incl_scan_u32(unsigned x) {
    x += WDX_DPP(0, x, 0x111, 0xf);
    x += WDX_DPP(0, x, 0x112, 0xf);
    x += WDX_DPP(0, x, 0x114, 0xf);
    x += WDX_DPP(0, x, 0x118, 0xf);
    x += WDX_DPP(0, x, 0x142, 0xa);
    x += WDX_DPP(0, x, 0x143, 0xc);
    return x;  // lane 63 holds the wave total
}
__device__ __forceinline__ unsigned wave_sum_u32(unsigned x) {
    return (unsigned)__builtin_amdgcn_readlane((int)wave_incl_scan_u32(x), 63);
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned x) {
    x = min(x, WDX_DPP(0xffffffffu, x, 0x111, 0xf));
    x = min(x, WDX_DPP(0xffffffffu, x, 0x112, 0xf));
    x = min(x, WDX_DPP(0xffffffffu, x, 0x114, 0xf));
    x = min(x, WDX_DPP(0xffffffffu, x, 0x118, 0xf));
    x = min(x, WDX_DPP(0xffffffffu, x, 0x142, 0xa));
    x = min(x, WDX_DPP(0xffffffffu, x, 0x143, 0xc));
    return (unsigned)__builtin_amdgcn_readlane((int)x, 63);
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned x) {
    x = max(x, WDX_DPP(0u, x, 0x111, 0xf));
    x = max(x, WDX_DPP(0u, x, 0x112, 0xf));
    x = max(x, WDX_DPP(0u, x, 0x114, 0xf));
    x = max(x, WDX_DPP(0u, x, 0x118, 0xf));
    x = max(x, WDX_DPP(0u, x, 0x142, 0xa));
    x = max(x, WDX_DPP(0u, x, 0x143, 0xc));
    return (unsigned)__builtin_amdgcn_readlane((int)x, 63);
}

// ---- float64 helpers shared by the one-wave-per-read kernels (refinement match, split tail) ------------------------
__device__ __forceinline__ double bcast_f64(double v, int src) {   // src wave-uniform
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}
__device__ __forceinline__ double vmin_f64(double a, double b) {   // no NaN here: `t < m ? t : m`
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double vmax_f64(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// value of lane (l ^ J), J = 1, 2, 4, 8, on DPP (quad permutes, a pair of bank-masked row shifts, row rotate)
template <int J>
__device__ __forceinline__ int xor_lane_dpp(int v) {
    if constexpr (J == 1) return __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false);        // quad_perm [1,0,3,2]
    else if constexpr (J == 2) return __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
    else if constexpr (J == 4) {
        int p = __builtin_amdgcn_update_dpp(v, v, 0x104, 0xf, 0x5, false);   // row_shl:4 -> lanes 0-3, 8-11 of a row
        return __builtin_amdgcn_update_dpp(p, v, 0x114, 0xf, 0xa, false);    // row_shr:4 -> lanes 4-7, 12-15
    } else {
        static_assert(J == 8, "xor_lane_dpp");
        return __builtin_amdgcn_update_dpp(v, v, 0x128, 0xf, 0xf, false);    // row_ror:8
    }
}
// (min, max) of v over the lane pair (l, l ^ J), on every lane of the pair
template <int J>
__device__ __forceinline__ void pair_minmax(const double v, double &mn, double &mx) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    if constexpr (J <= 8) {
        const double p = __hiloint2double(xor_lane_dpp<J>(hi), xor_lane_dpp<J>(lo));
        mn = vmin_f64(v, p);
        mx = vmax_f64(v, p);
    } else {
        // v_permlane{16,32}_swap of a register with a copy of itself leaves each lane pair's two values side by side
        double x0, x1;
        if constexpr (J == 16) {
            const auto l2 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
            const auto h2 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
            x0 = __hiloint2double(h2[0], l2[0]);
            x1 = __hiloint2double(h2[1], l2[1]);
        } else {
            static_assert(J == 32, "pair_minmax");
            const auto l2 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
            const auto h2 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
            x0 = __hiloint2double(h2[0], l2[0]);
            x1 = __hiloint2double(h2[1], l2[1]);
        }
        mn = vmin_f64(x0, x1);
        mx = vmax_f64(x0, x1);
    }
}
template <int K, int J>
__device__ __forceinline__ void sort_step(double &a, double &b, const int lane) {
    if constexpr (J == 64) {
        const double lo = vmin_f64(a, b), hi = vmax_f64(a, b);
        a = lo;
        b = hi;
    } else {
        const bool lower = (lane & J) == 0;
        const bool asc_a = K >= 64 ? true : (lane & K) == 0;                            // element index lane: bit K
        const bool asc_b = K == 128 ? true : (K == 64 ? false : (lane & K) == 0);       // element index 64 + lane
        double na, xa, nb, xb;
        pair_minmax<J>(a, na, xa);
        pair_minmax<J>(b, nb, xb);
        a = (lower == asc_a) ? na : xa;
        b = (lower == asc_b) ? nb : xb;
    }
    if constexpr (J > 1) sort_step<K, J / 2>(a, b, lane);
}
template <int K>
__device__ __forceinline__ void sort_stage(double &a, double &b, const int lane) {
    sort_step<K, K / 2>(a, b, lane);
    if constexpr (K < 128) sort_stage<2 * K>(a, b, lane);
}
// ascending bitonic sort of the 128 values {a of lane l = element l, b of lane l = element 64 + l}: 28 compare-exchange
// steps, all in registers (DPP for distances <= 8, v_permlane16/32_swap above, distance 64 inside the lane)
__device__ __forceinline__ void wave_sort128(double &a, double &b, const int lane) { sort_stage<2>(a, b, lane); }
// np.median of the c values (a: elements 0..63, b: 64..127; the others padded with +inf), no NaN
__device__ __forceinline__ double wave_median(double a, double b, const int c, const int lane) {
    wave_sort128(a, b, lane);
    const int klo = (c - 1) / 2, khi = c / 2;
    const double lo = klo < 64 ? bcast_f64(a, klo) : bcast_f64(b, klo - 64);
    const double hi = khi < 64 ? bcast_f64(a, khi) : bcast_f64(b, khi - 64);
    return (c & 1) ? hi : (lo + hi) / 2.0;
}

}  // namespace wdx

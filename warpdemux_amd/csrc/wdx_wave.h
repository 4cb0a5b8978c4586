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

}  // namespace wdx

// Fast fingerprint kernels of the ODD window widths 21, 23, 25, 27, 29, 31, 33, 35 (exact scores only, fast_body's kExactOnly; no
// streaming form) -- a translation unit of their own so that the build compiles the instantiations side by side.  The
// templates are wdx_fingerprint.hip's; nothing else of it is compiled here.
#define WDX_DEV_KERNELS_ONLY 1
#define WDX_EXTRA_TU 1
#include "wdx_fingerprint.hip"

namespace wdx {

bool exact_only_kernels_d(int fw, bool ext, FastKernelSet &k) {
    switch (fw) {
        case 21: fill_wide_set<21, false>(ext, k); return true;
        case 23: fill_wide_set<23, false>(ext, k); return true;
        case 25: fill_wide_set<25, false>(ext, k); return true;
        case 27: fill_wide_set<27, false>(ext, k); return true;
        case 29: fill_wide_set<29, false>(ext, k); return true;
        case 31: fill_wide_set<31, false>(ext, k); return true;
        case 33: fill_wide_set<33, false>(ext, k); return true;
        case 35: fill_wide_set<35, false>(ext, k); return true;
        default: return false;
    }
}

}  // namespace wdx

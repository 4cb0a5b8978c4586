// Fast fingerprint kernels of the ODD window widths 7, 9, 11, 13, 15, 17, 19 (exact scores only, fast_body's kExactOnly; no
// streaming form) -- a translation unit of their own so that the build compiles the instantiations side by side.  The
// templates are wdx_fingerprint.hip's; nothing else of it is compiled here.
#define WDX_DEV_KERNELS_ONLY 1
#define WDX_EXTRA_TU 1
#include "wdx_fingerprint.hip"

namespace wdx {

bool exact_only_kernels_c(int fw, bool ext, FastKernelSet &k) {
    switch (fw) {
        case 7: fill_wide_set<7, false>(ext, k); return true;
        case 9: fill_wide_set<9, false>(ext, k); return true;
        case 11: fill_wide_set<11, false>(ext, k); return true;
        case 13: fill_wide_set<13, false>(ext, k); return true;
        case 15: fill_wide_set<15, false>(ext, k); return true;
        case 17: fill_wide_set<17, false>(ext, k); return true;
        case 19: fill_wide_set<19, false>(ext, k); return true;
        default: return false;
    }
}

}  // namespace wdx

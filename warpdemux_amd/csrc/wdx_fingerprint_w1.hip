// Fast fingerprint kernels of the window widths 8, 10, 14, 16, 20 (even, not multiples of the tile's six positions per
// lane: exact scores only, fast_body's kExactOnly) -- a translation unit of their own so that the build compiles the
// instantiations side by side.  The templates are wdx_fingerprint.hip's; nothing else of it is compiled here.
#define WDX_DEV_KERNELS_ONLY 1
#define WDX_EXTRA_TU 1
#include "wdx_fingerprint.hip"

namespace wdx {

bool exact_only_kernels_a(int fw, bool ext, FastKernelSet &k) {
    switch (fw) {
        case 8: fill_wide_set<8>(ext, k); return true;
        case 10: fill_wide_set<10>(ext, k); return true;
        case 14: fill_wide_set<14>(ext, k); return true;
        case 16: fill_wide_set<16>(ext, k); return true;
        case 20: fill_wide_set<20>(ext, k); return true;
        default: return false;
    }
}

}  // namespace wdx

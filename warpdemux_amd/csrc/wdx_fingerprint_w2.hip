// Fast fingerprint kernels of the window widths 22, 26, 28, 32, 34 (even, not multiples of the tile's six positions per
// lane: exact scores only, fast_body's kExactOnly) -- a translation unit of their own so that the build compiles the
// instantiations side by side.  The templates are wdx_fingerprint.hip's; nothing else of it is compiled here.
#define WDX_DEV_KERNELS_ONLY 1
#define WDX_EXTRA_TU 1
#include "wdx_fingerprint.hip"

namespace wdx {

bool exact_only_kernels_b(int fw, bool ext, FastKernelSet &k) {
    switch (fw) {
        case 22: fill_wide_set<22>(ext, k); return true;
        case 26: fill_wide_set<26>(ext, k); return true;
        case 28: fill_wide_set<28>(ext, k); return true;
        case 32: fill_wide_set<32>(ext, k); return true;
        case 34: fill_wide_set<34>(ext, k); return true;
        default: return false;
    }
}

}  // namespace wdx

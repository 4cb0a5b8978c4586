#!/bin/bash
# Opcode-class tables of the t-score tile of the MAIN fingerprint kernel, fingerprint_fast_kernel<20,false,12,1,true>:
# the approximate-keys pass (what every wave tile executes) and the exact-scores pass (a wave tile's fall-back / the
# read-level retry), always-executed parts only (scratch copies of the sources with the other pass and the rare plateau
# block compiled out).  Compiles that ONE instantiation (WDX_DEV_KERNELS_ONLY): seconds, not the library's minutes.
# Usage: tools/isa_tile.sh [outdir]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-/tmp/isa_tile}
INST='fingerprint_fast_kernel<20, false, 12, 1, true>(wdx::FastArgs)'
SYM=fingerprint_fast_kernelILi20ELb0ELi12ELi1ELb1
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -save-temps=obj -Rpass-analysis=kernel-resource-usage"
for V in approx exact product; do
    D="$OUT/$V"
    rm -rf "$D" && mkdir -p "$D" && cp -r "$ROOT/warpdemux_amd/csrc" "$D/csrc" && cp -r "$ROOT/include" "$D/"
    cd "$D/csrc"
    sed -i "s#\"../../include/wdx.h\"#\"$D/include/wdx.h\"#" wdx_common.h
    MARK="v_rcp_f64 6"
    if [ $V != product ]; then
        sed -i 's/                if (wave_plateau) {$/                if (false \&\& wave_plateau) {/' wdx_fingerprint_fast.inc
    fi
    if [ $V = approx ]; then
        # a doubt hands the read to the slow path here, so the doubt logic stays live and the exact pass disappears
        sed -i 's/                bool wave_exact = exact_sc;/                const bool wave_exact = false;/; s/                if (!wave_exact) wave_exact = pass(std::false_type{}) != 0;  \/\/ wave-uniform/                if (pass(std::false_type{}) != 0) slow |= 1;/' wdx_fingerprint_fast.inc
    elif [ $V = exact ]; then
        sed -i 's/                bool wave_exact = exact_sc;/                const bool wave_exact = true;/; s/                if (!wave_exact) wave_exact = pass(std::false_type{}) != 0;  \/\/ wave-uniform//' wdx_fingerprint_fast.inc
        MARK="v_rsq_f64 6"
    fi
    cat > one.hip <<EOT
#define WDX_DEV_KERNELS_ONLY 1
#include "$D/csrc/wdx_fingerprint.hip"
template __global__ void wdx::$INST;
EOT
    /opt/rocm/bin/hipcc $FLAGS -I"$D/csrc" -c one.hip -o one.o 2> build.log || { grep -E "error" -A3 build.log | head -20; exit 1; }
    echo "==== $V ===="
    if [ $V = product ]; then
        python3 "$ROOT/tools/resource_usage.py" build.log | grep "$SYM"
    else
        python3 "$ROOT/tools/isa_opclass.py" one-hip-amdgcn-amd-amdhsa-gfx950.s $SYM $MARK | head -21
    fi
done

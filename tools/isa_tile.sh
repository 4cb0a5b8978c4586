#!/bin/bash
# Opcode-class tables of the t-score tile of fingerprint_fast_kernel<24,false>: the approximate-keys pass (what every
# wave tile executes) and the exact-scores pass (a wave tile's fall-back / the read-level retry), always-executed parts
# only (scratch copies of the sources with the other pass and the rare plateau block compiled out).
# Usage: tools/isa_tile.sh [outdir]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-/tmp/isa_tile}
SYM=_ZN3wdx23fingerprint_fast_kernelILi24ELb0ELi12ELi1EEEvNS_8FastArgsE
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -save-temps -Rpass-analysis=kernel-resource-usage"
for V in approx exact; do
    D="$OUT/$V"
    rm -rf "$D" && mkdir -p "$D" && cp -r "$ROOT/warpdemux_amd/csrc" "$D/csrc" && cp -r "$ROOT/include" "$D/"
    cd "$D/csrc"
    sed -i "s#\"../../include/wdx.h\"#\"$D/include/wdx.h\"#" wdx_common.h
    sed -i 's/                if (wave_plateau) {$/                if (false \&\& wave_plateau) {/' wdx_fingerprint_fast.inc
    if [ $V = approx ]; then
        # a doubt hands the read to the slow path here, so the doubt logic stays live and the exact pass disappears
        sed -i 's/                bool wave_exact = exact_sc;/                const bool wave_exact = false;/; s/                if (!wave_exact) wave_exact = pass(std::false_type{}) != 0;  \/\/ wave-uniform/                if (pass(std::false_type{}) != 0) slow |= 1;/' wdx_fingerprint_fast.inc
        MARK="v_rcp_f64 6"
    else
        sed -i 's/                bool wave_exact = exact_sc;/                const bool wave_exact = true;/; s/                if (!wave_exact) wave_exact = pass(std::false_type{}) != 0;  \/\/ wave-uniform//' wdx_fingerprint_fast.inc
        MARK="v_rsq_f64 6"
    fi
    /opt/rocm/bin/hipcc $FLAGS -c wdx_fingerprint.hip -o fp.o 2> build.log || { tail -20 build.log; exit 1; }
    echo "==== $V pass ===="
    python3 "$ROOT/tools/isa_opclass.py" wdx_fingerprint-hip-amdgcn-amd-amdhsa-gfx950.s $SYM $MARK | head -21
done
echo "==== product build ===="
cd "$ROOT/warpdemux_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Rpass-analysis=kernel-resource-usage \
    -c wdx_fingerprint.hip -o "$OUT/fp.o" 2>&1 | grep -A9 "fingerprint_fast_kernelILi24ELb0" | grep -E "VGPRs:|SGPRs:|Occupancy|ScratchSize" | head -5

#!/bin/bash
# Opcode-class table of the t-score tile of fingerprint_fast_kernel<24,false>, always-executed part only
# (the rare plateau block is compiled out of a scratch copy).  Usage: tools/isa_tile.sh [outdir]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-/tmp/isa_tile}
rm -rf "$OUT" && mkdir -p "$OUT" && cp -r "$ROOT/warpdemux_amd/csrc" "$OUT/csrc" && cp -r "$ROOT/include" "$OUT/"
cd "$OUT/csrc"
sed -i "s#\"../../include/wdx.h\"#\"$OUT/include/wdx.h\"#" wdx_common.h
sed -i 's/                if (wave_plateau) {$/                if (false \&\& wave_plateau) {/' wdx_fingerprint_fast.inc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -save-temps \
    -Rpass-analysis=kernel-resource-usage -c wdx_fingerprint.hip -o fp.o 2> build.log || { tail -20 build.log; exit 1; }
grep -A9 "fingerprint_fast_kernelILi24ELb0" build.log | grep -E "VGPRs:|SGPRs:|Occupancy|LDS Size|ScratchSize" | head -6
python3 "$ROOT/tools/isa_opclass.py" wdx_fingerprint-hip-amdgcn-amd-amdhsa-gfx950.s _ZN3wdx23fingerprint_fast_kernelILi24ELb0EEEvNS_8FastArgsE | head -24

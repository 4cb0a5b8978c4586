#!/bin/bash
# Development aid: compile ONE kernel instantiation of wdx_fingerprint.hip (seconds instead of minutes) and print its
# resource usage.  Usage: tools/dev/one_kernel.sh 'clip_bounds_kernel<80>(wdx::ClipArgs)' [extra hipcc flags]
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
INST="$1"; shift
mkdir -p /tmp/wdx_dev
cat > /tmp/wdx_dev/one.hip <<EOT
#define WDX_DEV_KERNELS_ONLY 1
#include "$ROOT/warpdemux_amd/csrc/wdx_fingerprint.hip"
template __global__ void wdx::$INST;
EOT
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -I"$ROOT/warpdemux_amd/csrc" \
    -Rpass-analysis=kernel-resource-usage -save-temps=obj -c /tmp/wdx_dev/one.hip -o /tmp/wdx_dev/one.o "$@" > /tmp/wdx_dev/one.txt 2>&1 || { grep -E "error" -A3 /tmp/wdx_dev/one.txt | head -40; exit 1; }
python3 "$ROOT/tools/resource_usage.py" /tmp/wdx_dev/one.txt | grep -v selftest

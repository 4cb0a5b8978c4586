#!/bin/bash
# A/B of two builds of libwdx_hip.so on the SAME GPU box (boxes differ by a few percent, so timings of separate
# gpurun calls cannot resolve 1-2 % effects).  Here (no GPU): tools/ab.sh build  -> builds A = git HEAD's csrc and
# B = the working tree's csrc into gpurun_out/ab/.  On the box: tools/ab.sh run [reps] -> alternates A, B.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
if [ "$1" = "build" ]; then
    rm -rf /tmp/ab_build && mkdir -p /tmp/ab_build/A/warpdemux_amd /tmp/ab_build/A/include ab_libs
    git archive HEAD warpdemux_amd/csrc include | tar -x -C /tmp/ab_build/A
    make -C /tmp/ab_build/A/warpdemux_amd/csrc -j4 -s 2>&1 | grep -E "error" || true
    cp /tmp/ab_build/A/warpdemux_amd/csrc/libwdx_hip.so ab_libs/libA.so
    make -C warpdemux_amd/csrc -j4 -s 2>&1 | grep -E "error" || true
    cp warpdemux_amd/csrc/libwdx_hip.so ab_libs/libB.so
    ls -la ab_libs
else
    REPS=${2:-2}
    for r in $(seq $REPS); do
        for v in A B; do
            WDX_LIB_PATH=$ROOT/ab_libs/lib$v.so timeout 150 python3 bench.py --steps 5 --warmup 1 --no-cpu --no-secondary 2>/dev/null | tail -1 |
                python3 -c "import sys,json; j=json.loads(sys.stdin.read()); k=j['kernels_ms_per_step']; print('$v', round(j['value']/1e6,3), 'M reads/s  fp', round(k['fingerprint'],2), 'dtw', round(k['dtw'],2))"
        done
    done
fi

#!/bin/bash
# Like tools/ab.sh run, for any set of prebuilt libraries in ab_libs/: tools/ab3.sh "A B C" [reps]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
REPS=${2:-2}
for r in $(seq $REPS); do
    for v in $1; do
        WDX_LIB_PATH=$ROOT/ab_libs/lib$v.so timeout 150 python3 bench.py --steps 5 --warmup 1 --no-cpu --no-secondary 2>/dev/null | tail -1 |
            python3 -c "import sys,json; j=json.loads(sys.stdin.read()); k=j['kernels_ms_per_step']; print('$v', round(j['value']/1e6,3), 'M reads/s  fp', round(k['fingerprint'],2), 'main', round(k.get('fingerprint_main_kernel',0),2), 'clip', round(k.get('fingerprint_clip_kernel',0),2), 'dtw', round(k['dtw'],2))"
    done
done

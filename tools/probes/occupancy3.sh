#!/bin/bash
# Sensitivity of the fast fingerprint kernel to resident workgroups per CU: the peak-list capacity option inflates
# the LDS request (capP 2048 -> 48 KB -> three workgroups per CU instead of four); the work per read is unchanged.
cd "$(dirname "$0")/../.."
for opt in "" "--ctx-opt 6=2048" "" "--ctx-opt 6=2048"; do
    python3 bench.py --steps 4 --warmup 1 --no-cpu --no-secondary --ctx-opt 5=1 $opt 2>&1 | grep -E "workgroups/CU|^\{" |
        python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('  ', round(j['value'] / 1e6, 3), 'M reads/s  fp', round(j['kernels_ms_per_step']['fingerprint'], 2))
    elif 'capF=6144' in l: print(l.strip())
" | sort -u
done

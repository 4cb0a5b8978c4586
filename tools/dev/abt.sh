#!/bin/bash
cd /root/repo
for r in 1 2; do for v in A B; do
  echo "$v: $(WDX_LIB_PATH=/root/repo/ab_libs/lib$v.so python3 tools/bench_triple.py 120 9 18 262144 1.0 5 2>&1 | tail -1 | sed 's/.*samples (max [0-9]*): //')  |  $(WDX_LIB_PATH=/root/repo/ab_libs/lib$v.so python3 tools/bench_refine.py 32768 2>&1 | tail -1)"
done; done
bash tools/ab3.sh "A B" 2

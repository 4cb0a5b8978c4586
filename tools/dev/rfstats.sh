#!/bin/bash
# per-kernel durations of the tRNA refinement flow (rocprofv3 --kernel-trace --stats over tools/bench_refine.py 32768)
set -u
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p "$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/rf
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/rf -- python3 $ROOT/tools/bench_refine.py 32768 > $ROOT/gpurun_out/rf.log 2>&1
tail -2 $ROOT/gpurun_out/rf.log
python3 - $(find $ROOT/gpurun_out/rf -name "*kernel_trace.csv" | head -1) <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# the last refinement call: from the last clip_bounds_kernel on
i=max(k for k,r in enumerate(rows) if "clip_bounds_kernel" in r["Kernel_Name"])
t=0
for r in rows[i:]:
    if "rocclr" in r["Kernel_Name"]: continue
    d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3; t+=d
    print("  %-72s %8.1f us"%(r["Kernel_Name"][:72], d))
print("  sum %.1f us"%t)
PY

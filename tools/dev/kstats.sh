#!/bin/bash
# per-kernel time of any command: tools/dev/kstats.sh <tag> <command ...>  -> gpurun_out/<tag>_kernel_stats.csv (+ the top rows)
set -u
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
TAG=$1; shift
mkdir -p "$ROOT/gpurun_out"
D="$ROOT/gpurun_out/${TAG}_kstats"; rm -rf "$D"
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d "$D" -- "$@") > "$ROOT/gpurun_out/${TAG}_kstats.log" 2>&1
cp "$(find "$D" -name '*kernel_stats.csv' | head -1)" "$ROOT/gpurun_out/${TAG}_kernel_stats.csv"
rm -rf "$D"
grep -v amdgpu.ids "$ROOT/gpurun_out/${TAG}_kstats.log" | tail -2
python3 - "$ROOT/gpurun_out/${TAG}_kernel_stats.csv" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print("  %-70s calls %4s  avg %10.1f us  total %10.2f ms  %5s %%" % (r["Name"].split("(")[0][-70:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY

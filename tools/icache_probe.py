#!/usr/bin/env python3
"""Instruction-cache counters of the hot kernels (rocprofv3 --pmc, one pass): python tools/icache_probe.py [reads]"""
import csv, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C = ["SQC_ICACHE_REQ", "SQC_ICACHE_HITS", "SQC_ICACHE_MISSES", "SQC_ICACHE_MISSES_DUPLICATE", "SQ_WAVE_CYCLES", "SQ_IFETCH", "SQ_WAIT_INST_ANY"]
reads = sys.argv[1] if len(sys.argv) > 1 else "1000000"
d = os.path.join(ROOT, "gpurun_out", "icache")
os.makedirs(d, exist_ok=True)
cmd = ["rocprofv3", "--kernel-trace", "--pmc", *C, "--output-format", "csv", "-d", d, "--", sys.executable,
       os.path.join(ROOT, "bench.py"), "--reads", reads, "--steps", "1", "--warmup", "0", "--no-cpu", "--no-secondary"]
with open(os.path.join(d, "run.log"), "w") as fh:
    subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=fh, stderr=subprocess.STDOUT, check=False)
acc = {}
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if int(r["Grid_Size"]) < (1 << 16):
            continue
        k = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").replace("wdx::", "")
        a = acc.setdefault(k, {})
        a[r["Counter_Name"]] = a.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for k, a in acc.items():
    print(k, {c: f"{v:.3g}" for c, v in a.items()})
    if a.get("SQC_ICACHE_REQ"):
        print("   icache miss rate %.3f%%  (misses incl. duplicates %.3f%%)" % (100 * a.get("SQC_ICACHE_MISSES", 0) / a["SQC_ICACHE_REQ"], 100 * (a.get("SQC_ICACHE_MISSES", 0) + a.get("SQC_ICACHE_MISSES_DUPLICATE", 0)) / a["SQC_ICACHE_REQ"]))

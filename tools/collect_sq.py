#!/usr/bin/env python3
"""Where the wave cycles of the two hot kernels go: rocprofv3 SQ/LDS counters -> profiles/<tag>_sq_counters.json.

Run on the GPU box from the repo root; each pass is a separate rocprofv3 run with --kernel-trace and
--pmc only (MI355X_MICROARCH.md, "rocprofv3 PMC slots"):

    python tools/collect_sq.py [--reads N] [--tag r01]

SQ_WAVE_CYCLES ~= SQ_WAIT_ANY (parked on s_waitcnt / barrier) + SQ_WAIT_INST_ANY (issue stall) +
SQ_ACTIVE_INST_ANY (issuing); all in quad-cycles summed over waves.
"""
import argparse
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = [
    ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
     "SQ_ACTIVE_INST_LDS", "SQ_INSTS_VALU", "SQ_INSTS_LDS"],
    ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_ADDR_CONFLICT", "SQ_WAIT_INST_LDS", "SQ_INSTS_SALU",
     "SQ_WAVES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"],
]


def run_pass(i, counters, outdir, reads):
    d = os.path.join(outdir, f"pass{i}")
    os.makedirs(d, exist_ok=True)
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", *counters, "--output-format", "csv", "-d", d, "--",
           sys.executable, os.path.join(ROOT, "bench.py"), "--reads", str(reads), "--steps", "1", "--warmup", "0",
           "--no-cpu", "--no-secondary"]
    with open(os.path.join(d, "bench.log"), "w") as fh:
        subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=fh, stderr=subprocess.STDOUT,
                       check=True)
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=1_000_000)
    ap.add_argument("--tag", default="r01")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "sq"))
    args = ap.parse_args()
    args.out = os.path.abspath(args.out)
    res = {}
    for i, counters in enumerate(PASSES):
        for r in run_pass(i, counters, args.out, args.reads):
            if int(r["Grid_Size"]) < (1 << 16):
                continue
            k = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").replace("wdx::", "")
            d = res.setdefault(k, {})
            c = d.setdefault(r["Counter_Name"], [0.0, 0])
            c[0] += float(r["Counter_Value"])
            c[1] += 1
    out = {"round": args.tag, "reads_per_launch": args.reads, "kernels": {}, "source": "rocprofv3 --kernel-trace --pmc (two passes), "
           "tools/collect_sq.py; per-launch averages, SQ cycle counters in quad-cycles summed over waves"}
    for k, d in res.items():
        avg = {c: v[0] / v[1] for c, v in d.items()}
        wc = avg.get("SQ_WAVE_CYCLES")
        e = {"per_launch": avg, "launches": max(v[1] for v in d.values())}
        if wc:
            e["share_of_wave_cycles"] = {
                "parked_waitcnt_or_barrier": avg["SQ_WAIT_ANY"] / wc,
                "issue_stall": avg["SQ_WAIT_INST_ANY"] / wc,
                "issuing": avg["SQ_ACTIVE_INST_ANY"] / wc,
                "issuing_valu": avg["SQ_ACTIVE_INST_VALU"] / wc,
                "issuing_lds": avg["SQ_ACTIVE_INST_LDS"] / wc,
            }
        if avg.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_bank_conflict_share_of_lds_cycles"] = avg["SQ_LDS_BANK_CONFLICT"] / avg["SQ_LDS_IDX_ACTIVE"]
        if avg.get("SQ_BUSY_CYCLES") and avg.get("SQ_ACTIVE_INST_VALU"):
            pass
        out["kernels"][k] = e
    for p in (os.path.join(ROOT, "gpurun_out", f"{args.tag}_sq_counters.json"),
              os.path.join(ROOT, "gpurun_out", "sq_counters_latest.json")):   # copy both under profiles/ afterwards
        with open(p, "w") as fh:
            json.dump(out, fh, indent=1)
    print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "per_launch"} for k, v in out["kernels"].items()},
                     indent=1))


if __name__ == "__main__":
    main()

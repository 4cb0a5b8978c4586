#!/usr/bin/env python3
"""SQ counters of named kernels under any command: rocprofv3 --kernel-trace --pmc (one pass per counter set), averaged
per launch of the largest grid of each kernel.   python tools/pmc_kernel.py <kernel-substring>[,<substring>...] -- <command...>"""
import csv
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = [["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_INSTS_SALU",
           "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS"],
          ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_ADDR_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_INST_LDS", "SQ_WAVES", "SQ_BUSY_CYCLES",
           "GRBM_GUI_ACTIVE"]]


def main():
    sep = sys.argv.index("--")
    names = sys.argv[1].split(",")
    cmd = sys.argv[sep + 1:]
    res = {}
    for i, counters in enumerate(PASSES):
        d = os.path.join(ROOT, "gpurun_out", "pmc_kernel", f"pass{i}")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "run.log"), "w") as fh:
            subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", *counters, "--output-format", "csv", "-d", d, "--", *cmd],
                           cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=fh, stderr=subprocess.STDOUT, check=True)
        rows = []
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            rows += list(csv.DictReader(open(f)))
        for r in rows:
            for nm in names:
                if nm in r["Kernel_Name"]:
                    res.setdefault(nm, {}).setdefault(int(r["Grid_Size"]), {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for nm, grids in res.items():
        g = max(grids)
        print(f"{nm}: grid {g} threads, {len(next(iter(grids[g].values())))} launches; per launch:")
        avg = {c: sum(v) / len(v) for c, v in grids[g].items()}
        for c, v in avg.items():
            print(f"    {c:24s} {v:16.0f}")
        wc = avg.get("SQ_WAVE_CYCLES")
        if wc:
            print("    shares of wave cycles: parked (waitcnt / barrier) %.2f, issue stall %.2f, issuing VALU %.2f, LDS %.2f" % (
                avg["SQ_WAIT_ANY"] / wc, avg["SQ_WAIT_INST_ANY"] / wc, avg["SQ_ACTIVE_INST_VALU"] / wc, avg["SQ_ACTIVE_INST_LDS"] / wc))
        if avg.get("GRBM_GUI_ACTIVE") and avg.get("SQ_ACTIVE_INST_VALU") is not None and "SQ_ACTIVE_INST_VALU" in avg:
            print("    VALU busy: %.2f of the SIMDs' issue time" % (avg["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * avg["GRBM_GUI_ACTIVE"] / 8.0)))


if __name__ == "__main__":
    main()

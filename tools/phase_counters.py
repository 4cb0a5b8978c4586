#!/usr/bin/env python3
"""Per-phase instruction / stall counters of fingerprint_fast_kernel: rocprofv3 --pmc around the phase
ablation of tools/profile_fingerprint.py (one launch per cut point), differences between cuts.

    python tools/phase_counters.py [n_reads] > gpurun_out/phase_counters.txt
"""
import csv
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COUNTERS = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU",
            "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS"]
CUTS = ["P0 load", "med1 hist", "med1 scan+locate", "med1 gather", "med1 rank", "P1a median (all)", "P1b MAD+clip",
        "P2+P3a t-score+maxima", "P3b suppression", "P4 top-E", "P5 boundaries", "P6 event means", "P7 normalise"]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
    d = os.path.join(ROOT, "gpurun_out", "phase_pmc")
    os.makedirs(d, exist_ok=True)
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", *COUNTERS, "--output-format", "csv", "-d", d, "--",
           sys.executable, os.path.join(ROOT, "tools", "profile_fingerprint.py"), "1024", "1", str(n)]
    with open(os.path.join(d, "run.log"), "w") as fh:
        subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, TMPDIR="/tmp", WDX_PROF_REPS="1"), stdout=fh,
                       stderr=subprocess.STDOUT, check=True)
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    disp = {}
    for r in rows:
        if "fingerprint_fast_kernel" not in r["Kernel_Name"] or int(r["Grid_Size"]) != n * 256:
            continue
        disp.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(disp)[-len(CUTS):]
    assert len(ids) == len(CUTS), (len(ids), len(CUTS))
    print(f"fingerprint_fast_kernel, {n} reads per launch; per read, marginal over the previous cut")
    print(f"{'phase':26s}" + "".join(f"{c.replace('SQ_', ''):>17s}" for c in COUNTERS))
    prev = {c: 0.0 for c in COUNTERS}
    for name, i in zip(CUTS, ids):
        cur = disp[i]
        if name == "P1a median (all)":
            name = "med1 evenfix+end"
        print(f"{name:26s}" + "".join(f"{(cur[c] - prev[c]) / n:17.0f}" for c in COUNTERS))
        prev = cur
    print(f"{'total':26s}" + "".join(f"{prev[c] / n:17.0f}" for c in COUNTERS))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""HBM traffic of the dominant kernel from rocprofv3 PMC passes -> profiles/traffic.json.

Run on the GPU box from the repo root (separate --pmc passes, never combined with trace domains
other than --kernel-trace, as MI355X_MICROARCH.md prescribes):

    python tools/collect_traffic.py [--reads N] [--out gpurun_out/traffic]

FETCH_SIZE on gfx950 is calibrated against a known byte count in the SAME access pattern (coalesced
dword-per-lane loads): bench.py --calib streams the signal buffer once with calib_read_dword_kernel.
"""
import argparse
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_pass(counter, outdir, reads, parse_only=False):
    d = os.path.join(outdir, counter)
    os.makedirs(d, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "--",
           sys.executable, os.path.join(ROOT, "bench.py"), "--reads", str(reads), "--steps", "1", "--warmup", "0",
           "--no-cpu", "--no-secondary", "--calib"]
    if not parse_only:
        with open(os.path.join(d, "bench.log"), "w") as fh:
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=fh, stderr=subprocess.STDOUT, check=True)
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    line = [l for l in open(os.path.join(d, "bench.log")) if l.startswith("{")][-1]
    return rows, json.loads(line)


def per_kernel(rows, counter, min_grid=1 << 16):
    """sum of the counter and number of dispatches per kernel name, full-size dispatches only (the
    10-read reference-fingerprint launch of bench.make_refs is not part of the workload)"""
    out = {}
    for r in rows:
        if r["Counter_Name"] != counter or int(r["Grid_Size"]) < min_grid:
            continue
        k = r["Kernel_Name"]
        v, n = out.get(k, (0.0, 0))
        out[k] = (v + float(r["Counter_Value"]), n + 1)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=10_000_000)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "traffic"))
    ap.add_argument("--tag", default="r05", help="round tag recorded in the JSON (bench.py prints it as the figure's source)")
    ap.add_argument("--parse-only", action="store_true", help="re-read existing CSVs under --out")
    args = ap.parse_args()
    args.out = os.path.abspath(args.out)
    fr, bj = run_pass("FETCH_SIZE", args.out, args.reads, args.parse_only)
    wr, _ = run_pass("WRITE_SIZE", args.out, args.reads, args.parse_only)
    fetch, write = per_kernel(fr, "FETCH_SIZE"), per_kernel(wr, "WRITE_SIZE")
    n_reads = bj["config"]["reads_per_gpu"]
    # calibration: known bytes = 4 * total samples of the batch
    calib_k = [k for k in fetch if "calib_read_dword" in k][0]
    dom_k = [k for k in fetch if "fingerprint_fast_kernel" in k][0]
    fp_bytes_total = (bj["fused_path"]["algorithmic_bytes_per_read"] - 44.0) * n_reads  # 4 * samples
    calib_kb, calib_n = fetch[calib_k]
    factor = fp_bytes_total / (calib_kb * 1024.0)          # true bytes per reported byte (dword pattern)
    # bench.py runs the fused pass twice here (allocation pass + 1 step), each possibly sliced into
    # several launches: average over the full-size launches
    fk, fn = fetch[dom_k]
    wk, wn = write[dom_k]
    fetch_b = fk * 1024.0 * factor / fn
    write_b = wk * 1024.0 / wn
    res = {
        "kernel": "fingerprint_fast_kernel",
        "round": args.tag,
        "reads_per_launch": n_reads // max(fn // 2, 1),
        # what bench.py looks up: the whole step (all launch slices of one pass over the batch)
        "reads_per_step": n_reads,
        "hbm_bytes_per_step": (fetch_b + write_b) * max(fn // 2, 1),
        "algorithmic_bytes_per_launch": bj["roofline"]["algorithmic_bytes_per_launch"],
        "hbm_bytes_per_launch": fetch_b + write_b,
        "fetch_bytes_per_launch_corrected": fetch_b,
        "write_bytes_per_launch": write_b,
        "fetch_size_raw_kb_per_launch": fk / fn,
        "fetch_calibration": {"kernel": calib_k.split("(")[0], "known_bytes": fp_bytes_total,
                              "reported_kb": calib_kb, "true_over_reported": factor},
        "launches_seen": fn,
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), tools/collect_traffic.py",
    }
    # the launch ahead of the main kernel: clip_bounds_kernel reads the main kernel's samples once more
    ck = [k for k in fetch if "clip_bounds_kernel" in k]
    if ck:
        cfk, cfn = fetch[ck[0]]
        cwk, cwn = write.get(ck[0], (0.0, 1))
        res["clip_bounds_kernel"] = {
            "fetch_bytes_per_launch_corrected": cfk * 1024.0 * factor / cfn, "write_bytes_per_launch": cwk * 1024.0 / max(cwn, 1),
            "launches_seen": cfn,
            "algorithmic_bytes_per_step": (bj["roofline"].get("clip_bounds_kernel") or {}).get("algorithmic_bytes_per_step")}
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    with open(os.path.join(ROOT, "profiles", "traffic.json"), "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Full-size parity gates, GPU engine vs CPU oracle (SURVEY 8(d) "parity gate run with every measurement", at the
sizes it names).  Run on the GPU box:  python tools/parity_gate.py [--r1 1000000] [--r2 100000] > log.json

  r1        >= 1e6 synthetic reads, WDX10 shape: status, call, float32 distances, float64 fingerprints, int64 dwell
            (= change-points) and the six statistics, bitwise (the fused device path for call/dist, the fingerprint
            entry point for the rest)
  r2        >= 1e5 fingerprints x 851 references x 25 points (WDX4's shape): float32 distances bitwise + argmin
  quantised 2e5 reads rounded to an ADC quantum (0.1755 pA) and to a coarse 2 pA grid: exact score ties and plateaus
            at scale through the fast kernel's plateau walk and its slow-path hand-over
  long      5e4 reads with 6.4-8 k-sample windows (8192-sample instantiation), 2e4 with 8.2-11.2 k (exact kernel), 1e4 with
            11.2-15.2 k (score curve in HBM)
  triples   1e5 reads each on the RNA002 (110, 15, 30) and tRNA (120, 9, 18) triples (fast kernels of widths 30 / 18),
            2.5e4 quantised reads on the RNA002 triple
  refine    5e4 tRNA-like reads through the consensus-refinement flow (fast kernel + match kernel + tail kernel; long
            barcode tails on the exact kernel): status, fingerprints, dwell, stats, query start / end, barcode start
"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from oracle import wdx_oracle as orc  # noqa: E402
from warpdemux_amd import sig_proc, synth  # noqa: E402
from warpdemux_amd.engine import DemuxEngine  # noqa: E402


def oracle_fp(mb, a_s, a_e, p, cores):
    n = mb.shape[0]
    step = max(64, -(-n // (cores * 4)))
    with ThreadPoolExecutor(cores) as ex:
        parts = list(ex.map(lambda a: orc.fingerprint_batch(mb[a:a + step], a_s[a:a + step], a_e[a:a + step], p),
                            range(0, n, step)))
    return [np.concatenate([q[i] for q in parts]) for i in range(4)]


def gate_minibatches(name, make, total, K, cores, chunk=16384):
    """make(first, n) -> (mb, a_s, a_e); engine's host-buffer fingerprint call vs oracle, chunk by chunk"""
    ph, po = sig_proc.SegParams(barcode_num_events=K), orc.SegParams(barcode_num_events=K)
    bad = {"status": 0, "fpt": 0, "dwell": 0, "stats": 0}
    hist = np.zeros(7, dtype=np.int64)
    t0 = time.perf_counter()
    for first in range(0, total, chunk):
        n = min(chunk, total - first)
        mb, a_s, a_e = make(first, n)
        fb = sig_proc.fingerprint_batch(mb, a_s, a_e, ph)
        fpt, dwell, stats, status = oracle_fp(mb, a_s, a_e, po, cores)
        ok = status == 0
        bad["status"] += int((fb.status != status).sum())
        bad["fpt"] += int((fb.fpt[ok].view(np.uint64) != fpt[ok].view(np.uint64)).any(axis=1).sum())
        bad["dwell"] += int((fb.dwell[ok] != dwell[ok]).any(axis=1).sum())
        bad["stats"] += int((fb.stats[ok].view(np.uint64) != stats[ok].view(np.uint64)).any(axis=1).sum())
        hist += np.bincount(status, minlength=7)[:7]
    return {"reads": total, "mismatching_reads": bad, "status_histogram": hist.tolist(), "ok": not any(bad.values()),
            "seconds": time.perf_counter() - t0}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--r1", type=int, default=1_000_000)
    ap.add_argument("--r2", type=int, default=100_000)
    ap.add_argument("--quant", type=int, default=200_000)
    ap.add_argument("--long", type=int, default=50_000)
    args = ap.parse_args()
    cores = bench.effective_cores()
    out = {"cores": cores}
    spec = synth.SynthSpec(n_barcodes=10)

    # ---- r1: the bench's own gate at full size ------------------------------------------------------------------
    clean = synth.SynthSpec(n_barcodes=10, noise_sigma=0.25, spikes=False)
    refs = bench.make_refs(clean, synth, sig_proc, 0)
    eng = DemuxEngine(refs, 15, 0.1, sig_proc.SegParams(barcode_num_events=110))
    sig, off, a_s, a_e, bc, max_len = eng.synth_packed(spec, 0, args.r1)
    res = eng.demux(sig, a_s, a_e, offsets=off, max_len=max_len)
    torch.cuda.synchronize()
    cpu, parity = bench.cpu_baseline_and_parity(eng, sig, off, a_s, a_e, res, refs, args.r1, 0.0, args.r1)
    out["r1"] = {"parity": parity, "oracle_reads_per_s": cpu["value"], "oracle_seconds": cpu["cpu_seconds"]}
    del sig, off, a_s, a_e, res
    eng.close()
    torch.cuda.empty_cache()

    # ---- r2: shipped-model DTW shape ------------------------------------------------------------------------------
    rng = np.random.default_rng(17)
    Y = rng.normal(size=(851, 25))
    X = rng.normal(size=(args.r2, 25))
    eng = DemuxEngine(Y, 15, 0.1, sig_proc.SegParams(barcode_num_events=25))
    d, am = eng.dtw(torch.from_numpy(X).cuda())
    dh, amh = d.cpu().numpy(), am.cpu().numpy()
    t0 = time.perf_counter()
    Dref, rows = bench._oracle_dtw_threads(X, Y, budget_s=1e9, rows_per_job=64)
    out["r2"] = {"pairs": int(Dref.size), "bitwise_equal": bool(np.array_equal(dh.view(np.uint32), Dref.view(np.uint32))),
                 "argmin_equal": bool(np.array_equal(amh, orc.argmin_rows(Dref))), "oracle_seconds": time.perf_counter() - t0}
    eng.close()

    # ---- quantised signals ------------------------------------------------------------------------------------------
    def quantised(q):
        def make(first, n):
            mb, a_s, a_e, _ = synth.generate_minibatch(spec, 5_000_000 + first, n, 9000)
            return (np.round(mb / np.float32(q)).astype(np.float32) * np.float32(q)), a_s, a_e
        return make
    out["quantised_adc_0.1755pA"] = gate_minibatches("adc", quantised(0.1755), args.quant, 110, cores)
    out["quantised_coarse_2pA"] = gate_minibatches("coarse", quantised(2.0), args.quant // 4, 110, cores)

    # ---- long adapter windows ----------------------------------------------------------------------------------------
    def long_reads(lo, hi, stride):
        def make(first, n):
            r = np.random.default_rng(900 + first)
            mb = np.full((n, stride), np.nan, dtype=np.float32)
            lens = r.integers(lo, hi, n)
            for i, ln in enumerate(lens):
                ev = int(r.integers(25, 55))
                mb[i, :ln] = (np.repeat(r.normal(80, 15, ln // ev + 1), ev)[:ln] + r.normal(0, 2, ln)).astype(np.float32)
            return mb, np.zeros(n, np.int32), lens.astype(np.int32)
        return make
    ph = dict(padding=0)
    g = gate_minibatches  # same gate, padding 0 through the params below

    def gate_long(make, total):
        ph_, po_ = sig_proc.SegParams(padding=0, barcode_num_events=110), orc.SegParams(padding=0, barcode_num_events=110)
        bad = {"status": 0, "fpt": 0, "dwell": 0}
        for first in range(0, total, 4096):
            n = min(4096, total - first)
            mb, a_s, a_e = make(first, n)
            fb = sig_proc.fingerprint_batch(mb, a_s, a_e, ph_)
            fpt, dwell, stats, status = oracle_fp(mb, a_s, a_e, po_, cores)
            ok = status == 0
            bad["status"] += int((fb.status != status).sum())
            bad["fpt"] += int((fb.fpt[ok].view(np.uint64) != fpt[ok].view(np.uint64)).any(axis=1).sum())
            bad["dwell"] += int((fb.dwell[ok] != dwell[ok]).any(axis=1).sum())
        return {"reads": total, "mismatching_reads": bad, "ok": not any(bad.values())}
    out["long_6400_8000"] = gate_long(long_reads(6400, 8000, 8000), args.long)
    out["long_8200_11200"] = gate_long(long_reads(8200, 11200, 11200), args.long * 2 // 5)
    out["long_11201_15200"] = gate_long(long_reads(11201, 15200, 15200), args.long // 5)   # score curve in HBM
    # ---- the other shipped parameter triples (fast kernels of widths 18 / 30) and the tRNA refinement flow -------------
    def gate_triple(E, d, W, total, quant=None):
        kw = dict(num_events=E, min_obs_per_base=d, running_stat_width=W, barcode_num_events=25)
        ph_, po_ = sig_proc.SegParams(**kw), orc.SegParams(**kw)
        bad = {"status": 0, "fpt": 0, "dwell": 0, "stats": 0}
        for first in range(0, total, 16384):
            n = min(16384, total - first)
            mb, a_s, a_e, _ = synth.generate_minibatch(spec, 7_000_000 + first, n, 9000)
            if quant:
                mb = np.round(mb / np.float32(quant)).astype(np.float32) * np.float32(quant)
            fb = sig_proc.fingerprint_batch(mb, a_s, a_e, ph_)
            fpt, dwell, stats, status = oracle_fp(mb, a_s, a_e, po_, cores)
            ok = status == 0
            bad["status"] += int((fb.status != status).sum())
            bad["fpt"] += int((fb.fpt[ok].view(np.uint64) != fpt[ok].view(np.uint64)).any(axis=1).sum())
            bad["dwell"] += int((fb.dwell[ok] != dwell[ok]).any(axis=1).sum())
            bad["stats"] += int((fb.stats[ok].view(np.uint64) != stats[ok].view(np.uint64)).any(axis=1).sum())
        return {"reads": total, "mismatching_reads": bad, "ok": not any(bad.values())}
    out["triple_rna002_110_15_30"] = gate_triple(110, 15, 30, args.quant // 2)
    out["triple_trna_120_9_18"] = gate_triple(120, 9, 18, args.quant // 2)
    out["triple_rna002_quantised"] = gate_triple(110, 15, 30, args.quant // 8, quant=0.1755)

    def gate_refine(total):
        cons = np.load(os.path.join(ROOT, "tests", "golden", "g8_refine.npz"))["consensus"]
        kw = dict(min_obs_per_base=9, running_stat_width=18, num_events=120, barcode_num_events=25)
        hp, hr = sig_proc.SegParams(**kw), sig_proc.RefineParams(query=cons, barcode_segm_events=25, barcode_keep_events=25)
        op, orr = orc.SegParams(**kw), orc.RefineParams(query=cons, barcode_segm_events=25, barcode_keep_events=25)
        bad = {"status": 0, "fpt": 0, "dwell": 0, "stats": 0, "idx": 0}
        hist = np.zeros(8, dtype=np.int64)
        for first in range(0, total, 4096):
            n = min(4096, total - first)
            r = np.random.default_rng(4000 + first)
            rows = []
            for i in range(n):
                emb = r.random() > 0.1
                lv = np.concatenate([r.normal(0, 1, int(r.integers(2, 34))), cons if emb else r.normal(0, 1, cons.size),
                                     r.normal(0, 1, int(r.integers(26, 70)))]) * 12.0 + 85.0
                dwl = r.integers(12, 60, lv.size)
                rows.append((np.repeat(lv, dwl) + r.normal(0, r.uniform(0.8, 2.5), int(dwl.sum()))).astype(np.float32)[:8100])
            stride = max(x.size for x in rows)
            mb = np.full((n, stride), np.nan, dtype=np.float32)
            for i, x in enumerate(rows):
                mb[i, :x.size] = x
            a_s = np.full(n, 100, dtype=np.int32)
            a_e = np.array([x.size - 100 for x in rows], dtype=np.int32)
            fb = sig_proc.fingerprint_refine_batch(mb, a_s, a_e, hp, hr)
            step = max(64, -(-n // (cores * 4)))
            with ThreadPoolExecutor(cores) as ex:
                parts = list(ex.map(lambda a: orc.fingerprint_refine_batch(mb[a:a + step], a_s[a:a + step], a_e[a:a + step], op, orr),
                                    range(0, n, step)))
            fpt, dwell, stats, idx, status = [np.concatenate([q[i] for q in parts]) for i in range(5)]
            good, rep = status == 0, (status == 0) | (status == 6)
            bad["status"] += int((fb.status != status).sum())
            bad["fpt"] += int((fb.fpt[good].view(np.uint64) != fpt[good].view(np.uint64)).any(axis=1).sum())
            bad["dwell"] += int((fb.dwell[good] != dwell[good]).any(axis=1).sum())
            bad["stats"] += int((fb.stats[rep].view(np.uint64) != stats[rep].view(np.uint64)).any(axis=1).sum())
            bad["idx"] += int((fb.refine_idx[rep] != idx[rep]).any(axis=1).sum())
            hist += np.bincount(status, minlength=8)[:8]
        return {"reads": total, "mismatching_reads": bad, "status_histogram": hist.tolist(), "ok": not any(bad.values())}
    out["trna_refinement_flow"] = gate_refine(args.quant // 4)

    out["all_ok"] = bool(out["r1"]["parity"]["ok"] and out["r2"]["bitwise_equal"] and out["r2"]["argmin_equal"]
                         and all(v["ok"] for k, v in out.items() if isinstance(v, dict) and "ok" in v))
    print(json.dumps(out, indent=1))
    sys.exit(0 if out["all_ok"] else 2)


if __name__ == "__main__":
    main()

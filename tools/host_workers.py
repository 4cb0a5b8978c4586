#!/usr/bin/env python3
"""The reference's real calling pattern, measured: P forked worker processes (file_proc.py:1197-1243,
ProcessPoolExecutor with the fork start method) sharing ONE GPU, each driving 1000-read x 10 000-sample float32
minibatches (file_proc.py:244-260, 380-454) through the engine: fingerprint -> DTW against 10 x 110-pt references ->
nearest-reference call, host buffers in and out (PCIe included).

    python tools/host_workers.py --workers 8 --mode pipe [--seconds 3] [--refill]

modes   sync   sig_proc.demux_batch on a pageable minibatch (what an unmodified worker loop would call)
        pipe   pipeline.MinibatchPipeline: two page-locked minibatch buffers, submit / wait on two streams
--refill       every iteration first copies the minibatch from a pageable array into the buffer it submits (the
               worker's own fill, which the reference does into its pageable array too)

The parent never touches the GPU; every child creates its context after the fork.  Each worker checks its results
against the CPU oracle once (outside the timed loop).  Prints one JSON line.
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

N_READS, STRIDE, K, N_REFS, WINDOW, PENALTY = 1000, 10000, 110, 10, 15, 0.1


def worker(wid, args, barrier, q):
    import numpy as np

    from oracle import wdx_oracle as orc
    from warpdemux_amd import pipeline, sig_proc, synth

    try:
        spec = synth.SynthSpec(n_barcodes=N_REFS)
        mb, a_s, a_e, _ = synth.generate_minibatch(spec, 1000 * wid, N_READS, STRIDE)
        refs = np.random.default_rng(0).normal(size=(N_REFS, K))
        params = sig_proc.SegParams(barcode_num_events=K)
        if args.mode == "sync":
            sig_proc.set_references(refs, WINDOW, PENALTY)
            src = mb.copy() if args.refill else None

            def step():
                if src is not None:
                    np.copyto(mb, src)
                return sig_proc.demux_batch(mb, a_s, a_e, params, want_dist=True)

            for _ in range(3):
                res = step()
            barrier.wait()
            t0 = time.perf_counter()
            n = 0
            while time.perf_counter() - t0 < args.seconds:
                res = step()
                n += 1
            dt = time.perf_counter() - t0
        else:
            pipe = pipeline.MinibatchPipeline(refs, WINDOW, PENALTY, params)
            bufs = [pipeline.pinned_empty((N_READS, STRIDE), np.float32) for _ in range(2)]
            for b in bufs:
                np.copyto(b, mb)
            for _ in range(2):
                for s in (0, 1):
                    pipe.submit(s, bufs[s], a_s, a_e)
                for s in (0, 1):
                    res = pipe.wait(s)
            barrier.wait()
            t0 = time.perf_counter()
            n = 0
            pipe.submit(0, bufs[0], a_s, a_e)
            k = 1
            while time.perf_counter() - t0 < args.seconds:
                s = k & 1
                if args.refill:
                    np.copyto(bufs[s], mb)       # the worker's fill of the next minibatch, overlapping the one in flight
                pipe.submit(s, bufs[s], a_s, a_e)
                res = pipe.wait(s ^ 1)
                n += 1
                k += 1
            res = pipe.wait((k - 1) & 1)
            n += 1
            dt = time.perf_counter() - t0
            pipe.close()
        fpt, dwell, stats, status = orc.fingerprint_batch(mb, a_s, a_e, orc.SegParams(barcode_num_events=K))
        ok = status == 0
        D = orc.dtw_matrix(fpt[ok], refs, WINDOW, PENALTY)
        parity = bool(np.array_equal(res.status, status) and np.array_equal(res.dist[ok].view(np.uint32), D.view(np.uint32))
                      and np.array_equal(res.call[ok], orc.argmin_rows(D)) and (res.call[~ok] == -1).all())
        q.put({"worker": wid, "minibatches": n, "seconds": dt, "parity": parity})
    except Exception as e:  # noqa: BLE001
        try:
            barrier.abort()
        except Exception:  # noqa: BLE001
            pass
        q.put({"worker": wid, "error": f"{type(e).__name__}: {e}"})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workers", type=int, default=4)
    ap.add_argument("--mode", choices=["sync", "pipe"], default="sync")
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--refill", action="store_true")
    args = ap.parse_args()
    ctx = mp.get_context("fork")      # the reference's start method (file_proc.py:1197)
    barrier = ctx.Barrier(args.workers)
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(w, args, barrier, q)) for w in range(args.workers)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(60)
    errs = [r for r in res if "error" in r]
    if errs:
        print(json.dumps({"error": errs}))
        sys.exit(1)
    reads = sum(r["minibatches"] for r in res) * N_READS
    wall = max(r["seconds"] for r in res)
    out = {"workers": args.workers, "mode": args.mode, "refill": bool(args.refill), "reads_per_s": reads / wall,
           "minibatches": sum(r["minibatches"] for r in res), "seconds": wall,
           "ms_per_minibatch_per_worker": 1e3 * wall / (sum(r["minibatches"] for r in res) / args.workers),
           "parity": all(r["parity"] for r in res)}
    print(json.dumps(out))
    sys.exit(0 if out["parity"] else 2)


if __name__ == "__main__":
    main()

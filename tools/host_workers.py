#!/usr/bin/env python3
"""The reference's real calling pattern, measured: P forked worker processes (file_proc.py:1197-1243,
ProcessPoolExecutor with the fork start method) sharing ONE GPU, each driving 1000-read x 10 000-sample float32
minibatches (file_proc.py:244-260, 380-454) through the engine: fingerprint -> DTW against 10 x 110-pt references ->
nearest-reference call, host buffers in and out (PCIe included).

    python tools/host_workers.py --workers 8 --mode pipe [--seconds 3] [--refill]

modes   sync   sig_proc.demux_batch on a pageable minibatch (what an unmodified worker loop would call)
        pipe   pipeline.MinibatchPipeline: two page-locked minibatch buffers, submit / wait on two streams
        feeder warpdemux_amd.feeder.Feeder: ONE GPU-facing process owns the context and serves a shared-memory ring of --slots
               minibatch slots (wdx_feeder_serve); the --workers processes call feeder.demux_batch (no context, no HIP
               call) -- the answer to "16 workers on 16 CPUs collapse": the device time-slices the processes' queues,
               so many small contexts take turns while one context's streams overlap
--jitter N     adapter_start ~ U{100 .. 100 + N} per read: rows carry whole reads (page-locked minibatches then go through
               the packed staging, only the windows cross the bus)
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
        mb, a_s, a_e, _ = synth.generate_minibatch(spec, 1000 * wid, N_READS, STRIDE, start_jitter=args.jitter)
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


def feeder_mode(args):
    """--mode feeder: warpdemux_amd.feeder.Feeder -- the parent creates the ring and the GPU-facing process (before it
    forks the workers, and without touching the GPU itself); every worker calls feeder.demux_batch on its own minibatch
    exactly as the sync mode calls sig_proc.demux_batch."""
    import numpy as np

    from warpdemux_amd import sig_proc
    from warpdemux_amd.feeder import Feeder

    ctx = mp.get_context("fork")
    P = args.workers
    full = args.mode == "feeder_full"
    if full:
        # the reference worker's whole minibatch (file_proc.py:380-454): ReadResults' arrays AND DTW_SVM.predict, on the
        # reference's own WDX10_rna004_v1_0 model (fixture g6b: 2 601 x 25-pt training fingerprints, 11 classes)
        from warpdemux_amd.models import DTW_SVM

        with np.load(os.path.join(ROOT, "tests", "golden", "g6b_dtw_svm_wdx10.npz")) as gz:
            g = {k: gz[k] for k in gz.files}     # (read here: the forked workers would share the lazy archive's file offset)
        lm = {int(k): int(v) for k, v in zip(g["label_keys"], g["label_vals"])}
        model = DTW_SVM(g["X_train"], g["n_support"], g["support"], g["dual_coef"], -g["intercept"], g["probA"], g["probB"], lm,
                        g["thresholds"], window=int(g["window"]), penalty=float(g["penalty"]), gamma=float(g["gamma"]),
                        pwr_dist=int(g["pwr_dist"]), block_size=int(g["block_size"]))
        KF = 25
        params = sig_proc.SegParams(barcode_num_events=KF)
        feeder = Feeder(model=model, params=params, max_reads=N_READS, stride=STRIDE, n_slots=args.slots)
    else:
        refs = np.random.default_rng(0).normal(size=(N_REFS, K))
        params = sig_proc.SegParams(barcode_num_events=K)
        feeder = Feeder(refs, WINDOW, PENALTY, params, max_reads=N_READS, stride=STRIDE, n_slots=args.slots)
    res_q = ctx.Queue()
    start = ctx.Barrier(P)

    def producer_full(pid):
        from oracle import wdx_oracle as orc
        from warpdemux_amd import synth
        try:
            spec = synth.SynthSpec(n_barcodes=N_REFS)
            mb, a_s, a_e, _ = synth.generate_minibatch(spec, 1000 * pid, N_READS, STRIDE, start_jitter=args.jitter)
            src = mb.copy() if args.refill else None
            for _ in range(2):
                fb, (y_pred, y_prob) = feeder.detect_and_predict(mb, a_s, a_e)
            start.wait()
            t0 = time.perf_counter()
            n = 0
            while time.perf_counter() - t0 < args.seconds:
                if src is not None:
                    np.copyto(mb, src)                # the worker's own fill of its minibatch
                fb, (y_pred, y_prob) = feeder.detect_and_predict(mb, a_s, a_e)
                n += 1
            dt = time.perf_counter() - t0
            m = 64       # (the oracle's DTW against 2 601 references: a sample of the minibatch)
            fpt, dwell, stats, status = orc.fingerprint_batch(mb[:m], a_s[:m], a_e[:m], orc.SegParams(barcode_num_events=KF))
            okk = status == 0
            D = orc.dtw_matrix(fpt[okk], np.ascontiguousarray(g["X_train"], dtype=np.float64), int(g["window"]), float(g["penalty"]))
            Kq = np.exp(-float(g["gamma"]) * np.power(D, int(g["pwr_dist"])))
            pr = orc.svm_predict_proba(Kq, g["n_support"].astype(np.int32), g["support"].astype(np.int32), g["dual_coef"],
                                       -g["intercept"], g["probA"], g["probB"])
            okf = fb.status == 0
            rows = np.cumsum(okf) - 1       # row of read i among the successful ones
            parity = bool(np.array_equal(fb.status[:m], status) and np.array_equal(fb.fpt[:m][okk].view(np.uint64), fpt[okk].view(np.uint64))
                          and np.array_equal(fb.dwell[:m][okk], dwell[okk]) and np.array_equal(fb.stats[:m][okk].view(np.uint64), stats[okk].view(np.uint64))
                          and np.abs(y_prob[rows[:m][okk]] - pr).max() <= 1e-5 and y_pred.shape[0] == int(okf.sum()))
            res_q.put({"worker": pid, "minibatches": n, "seconds": dt, "parity": parity})
        except Exception as e:  # noqa: BLE001
            try:
                start.abort()
            except Exception:  # noqa: BLE001
                pass
            res_q.put({"error": f"producer {pid} {type(e).__name__}: {e}"})

    def producer(pid):
        if full:
            return producer_full(pid)
        from oracle import wdx_oracle as orc
        from warpdemux_amd import synth
        try:
            spec = synth.SynthSpec(n_barcodes=N_REFS)
            mb, a_s, a_e, _ = synth.generate_minibatch(spec, 1000 * pid, N_READS, STRIDE, start_jitter=args.jitter)
            src = mb.copy() if args.refill else None
            for _ in range(2):
                res = feeder.demux_batch(mb, a_s, a_e)
            start.wait()
            t0 = time.perf_counter()
            n = 0
            while time.perf_counter() - t0 < args.seconds:
                if src is not None:
                    np.copyto(mb, src)                # the worker's own fill of its minibatch
                res = feeder.demux_batch(mb, a_s, a_e)
                n += 1
            dt = time.perf_counter() - t0
            fpt, dwell, stats, status = orc.fingerprint_batch(mb, a_s, a_e, orc.SegParams(barcode_num_events=K))
            okk = status == 0
            D = orc.dtw_matrix(fpt[okk], refs, WINDOW, PENALTY)
            parity = bool(np.array_equal(res.status, status) and np.array_equal(res.dist[okk].view(np.uint32), D.view(np.uint32)) and
                          np.array_equal(res.call[okk], orc.argmin_rows(D)) and (res.call[~okk] == -1).all())
            res_q.put({"worker": pid, "minibatches": n, "seconds": dt, "parity": parity})
        except Exception as e:  # noqa: BLE001
            try:
                start.abort()
            except Exception:  # noqa: BLE001
                pass
            res_q.put({"error": f"producer {pid} {type(e).__name__}: {e}"})

    procs = [ctx.Process(target=producer, args=(i,)) for i in range(P)]
    try:
        for p_ in procs:
            p_.start()
        res = []
        try:
            for _ in procs:
                res.append(res_q.get(timeout=600))
        except Exception as e:  # noqa: BLE001  (queue.Empty: a child hangs)
            res.append({"error": f"parent {type(e).__name__}: {e}"})
        errs = [r for r in res if "error" in r]
        for p_ in procs:
            p_.join(0.0 if errs else 60)
        if errs:
            print(json.dumps({"error": errs}))
            return 1
        reads = sum(r["minibatches"] for r in res) * N_READS
        wall = max(r["seconds"] for r in res)
        out = {"workers": P, "mode": args.mode, "slots": args.slots, "gpu_facing_processes": 1, "refill": bool(args.refill),
               "start_jitter": args.jitter, "reads_per_s": reads / wall, "minibatches": sum(r["minibatches"] for r in res),
               "seconds": wall, "ms_per_minibatch_per_worker": 1e3 * wall / (sum(r["minibatches"] for r in res) / P),
               "served_by_the_feeder": feeder.served(), "parity": all(r["parity"] for r in res)}
        print(json.dumps(out))
        return 0 if out["parity"] else 2
    finally:
        # no child outlives the ring: whatever still runs is ended first, then the feeder process and the shared memory
        for p_ in procs:
            if p_.is_alive():
                p_.terminate()
                p_.join(10)
        feeder.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workers", type=int, default=4)
    ap.add_argument("--mode", choices=["sync", "pipe", "feeder", "feeder_full"], default="sync",
                    help="feeder_full: feeder.detect_and_predict (fingerprints + dwell + statistics + DTW_SVM on the reference's WDX10 model)")
    ap.add_argument("--slots", type=int, default=16, help="feeder mode: ring slots (<= 32; at most 8 of them are in flight on the device)")
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--refill", action="store_true")
    ap.add_argument("--jitter", type=int, default=0, help="adapter_start ~ U{100 .. 100 + JITTER} per read (rows carry whole "
                    "reads, file_proc.py:244-260); 0 = every adapter starts at sample 100")
    args = ap.parse_args()
    if args.mode in ("feeder", "feeder_full"):
        sys.exit(feeder_mode(args))
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
    out = {"workers": args.workers, "mode": args.mode, "refill": bool(args.refill), "start_jitter": args.jitter,
           "reads_per_s": reads / wall,
           "minibatches": sum(r["minibatches"] for r in res), "seconds": wall,
           "ms_per_minibatch_per_worker": 1e3 * wall / (sum(r["minibatches"] for r in res) / args.workers),
           "parity": all(r["parity"] for r in res)}
    print(json.dumps(out))
    sys.exit(0 if out["parity"] else 2)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""The reference's real calling pattern, measured: P forked worker processes (file_proc.py:1197-1243,
ProcessPoolExecutor with the fork start method) sharing ONE GPU, each driving 1000-read x 10 000-sample float32
minibatches (file_proc.py:244-260, 380-454) through the engine: fingerprint -> DTW against 10 x 110-pt references ->
nearest-reference call, host buffers in and out (PCIe included).

    python tools/host_workers.py --workers 8 --mode pipe [--seconds 3] [--refill]

modes   sync   sig_proc.demux_batch on a pageable minibatch (what an unmodified worker loop would call)
        pipe   pipeline.MinibatchPipeline: two page-locked minibatch buffers, submit / wait on two streams
        feeder ONE GPU-facing process owns the context (MinibatchPipeline with --slots slots); --workers PRODUCER processes
               fill minibatches into a shared-memory ring that the feeder has page-locked (wdx_host_register) and get
               their results back through shared arrays -- the answer to "16 workers on 16 CPUs collapse": the GPU runs
               one process's kernels at a time, so many small contexts take turns while one context's streams overlap
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
    """--mode feeder: the parent creates the shared ring and the queues, forks the feeder and the producers."""
    import numpy as np
    from multiprocessing import shared_memory

    ctx = mp.get_context("fork")
    S, P = args.slots, args.workers
    mb_bytes = N_READS * STRIDE * 4
    shm = shared_memory.SharedMemory(create=True, size=S * mb_bytes)
    meta = shared_memory.SharedMemory(create=True, size=S * (N_READS * 8 + N_READS * 8 + N_READS * N_REFS * 4))
    try:
        ring = np.ndarray((S, N_READS, STRIDE), dtype=np.float32, buffer=shm.buf)
        mv = np.ndarray((S, N_READS * (4 + N_REFS)), dtype=np.int32, buffer=meta.buf)   # a_s | a_e | status | call | dist(f32 bits)
        free_q, ready_q = ctx.Queue(), ctx.Queue()
        done_q = [ctx.Queue() for _ in range(P)]
        res_q = ctx.Queue()
        for sl in range(S):
            free_q.put(sl)
        start = ctx.Barrier(P + 1)

        def feeder():
            from warpdemux_amd import pipeline, sig_proc
            try:
                refs = np.random.default_rng(0).normal(size=(N_REFS, K))
                params = sig_proc.SegParams(barcode_num_events=K)
                pipe = pipeline.MinibatchPipeline(refs, WINDOW, PENALTY, params, n_slots=S)
                unreg = pipeline.register_host(ring)
                import queue as _queue
                import threading

                start.wait()
                # two threads: one takes ready minibatches and submits them, one waits for the oldest in flight and hands
                # the results back (ctypes drops the GIL inside both calls; the slot index IS the ring index, and a ring
                # slot only comes back to the free list after its results were taken, so a slot is never submitted twice)
                inflight = _queue.Queue()
                counts = [0]
                errors = []

                def waiter():
                    try:
                        while True:
                            it = inflight.get()
                            if it is None:
                                return
                            sl, pid = it
                            r = pipe.wait(sl)
                            mv[sl, 2 * N_READS:3 * N_READS] = r.status
                            mv[sl, 3 * N_READS:4 * N_READS] = r.call
                            mv[sl, 4 * N_READS:] = r.dist.view(np.int32).ravel()
                            done_q[pid].put(sl)
                            counts[0] += 1
                    except Exception as e:  # noqa: BLE001
                        # abort: the submit loop stops (ready_q "stop"), every producer is told (done_q sentinel -1) so none
                        # of them spins on a minibatch that will never come back, the parent gets the error at once
                        errors.append(f"{type(e).__name__}: {e}")
                        res_q.put({"error": f"feeder waiter {errors[0]}"})
                        for dq in done_q:
                            dq.put(-1)
                        ready_q.put("stop")

                wt = threading.Thread(target=waiter)
                wt.start()
                while True:
                    item = ready_q.get()
                    if item == "stop":
                        break
                    if errors:
                        continue
                    sl, pid = item
                    pipe.submit(sl, ring[sl], mv[sl, :N_READS], mv[sl, N_READS:2 * N_READS])
                    inflight.put((sl, pid))
                inflight.put(None)
                wt.join()
                if errors:
                    raise RuntimeError(errors[0])
                n = counts[0]
                unreg()
                pipe.close()
                res_q.put({"feeder": n})
            except Exception as e:  # noqa: BLE001
                res_q.put({"error": f"feeder {type(e).__name__}: {e}"})
                try:
                    start.abort()
                except Exception:  # noqa: BLE001
                    pass

        def producer(pid):
            from oracle import wdx_oracle as orc
            from warpdemux_amd import synth
            try:
                spec = synth.SynthSpec(n_barcodes=N_REFS)
                mb, a_s, a_e, _ = synth.generate_minibatch(spec, 1000 * pid, N_READS, STRIDE, start_jitter=args.jitter)
                refs = np.random.default_rng(0).normal(size=(N_REFS, K))
                start.wait()
                t0 = time.perf_counter()
                import queue as _queue

                n, pending, last = 0, [], None

                deadline = t0 + args.seconds + 120.0      # hard stop: a producer never outlives a dead feeder

                def finish(sl):
                    nonlocal n, last
                    if sl < 0:
                        raise RuntimeError("the feeder aborted")
                    pending.remove(sl)
                    last = (mv[sl, 2 * N_READS:3 * N_READS].copy(), mv[sl, 3 * N_READS:4 * N_READS].copy(),
                            mv[sl, 4 * N_READS:].copy().view(np.float32).reshape(N_READS, N_REFS))
                    free_q.put(sl)
                    n += 1

                # (never block on the free list while a finished minibatch waits to be taken: with more producers than
                # ring slots that is a deadlock)
                while True:
                    active = time.perf_counter() - t0 < args.seconds
                    if not active and not pending:
                        break
                    if time.perf_counter() > deadline:
                        raise RuntimeError("no result from the feeder for 120 s")
                    progressed = False
                    if pending:
                        try:
                            finish(done_q[pid].get_nowait())
                            progressed = True
                        except _queue.Empty:
                            pass
                    if active and len(pending) < 2:
                        try:
                            sl = free_q.get_nowait()
                            np.copyto(ring[sl], mb)                   # the worker's own fill of its minibatch
                            mv[sl, :N_READS] = a_s
                            mv[sl, N_READS:2 * N_READS] = a_e
                            ready_q.put((sl, pid))
                            pending.append(sl)
                            progressed = True
                        except _queue.Empty:
                            pass
                    if not progressed:
                        if pending:
                            try:
                                finish(done_q[pid].get(timeout=0.005))
                            except _queue.Empty:
                                pass
                        else:
                            time.sleep(0.0005)
                dt = time.perf_counter() - t0
                fpt, dwell, stats, status = orc.fingerprint_batch(mb, a_s, a_e, orc.SegParams(barcode_num_events=K))
                okk = status == 0
                D = orc.dtw_matrix(fpt[okk], refs, WINDOW, PENALTY)
                parity = bool(last is not None and np.array_equal(last[0], status) and
                              np.array_equal(last[2][okk].view(np.uint32), D.view(np.uint32)) and
                              np.array_equal(last[1][okk], orc.argmin_rows(D)) and (last[1][~okk] == -1).all())
                res_q.put({"worker": pid, "minibatches": n, "seconds": dt, "parity": parity})
            except Exception as e:  # noqa: BLE001
                res_q.put({"error": f"producer {pid} {type(e).__name__}: {e}"})

        fp = ctx.Process(target=feeder)
        fp.start()
        procs = [ctx.Process(target=producer, args=(i,)) for i in range(P)]
        for p_ in procs:
            p_.start()
        res = []
        try:
            for _ in procs:
                res.append(res_q.get(timeout=600))
                if "error" in res[-1]:
                    break
            ready_q.put("stop")
            if not any("error" in r for r in res):
                res.append(res_q.get(timeout=120))
        except Exception as e:  # noqa: BLE001  (queue.Empty: a child hangs)
            res.append({"error": f"parent {type(e).__name__}: {e}"})
        for p_ in procs + [fp]:
            p_.join(0.0 if any("error" in r for r in res) else 60)
        errs = [r for r in res if "error" in r]
        if errs:
            print(json.dumps({"error": errs}))
            return 1
        w = [r for r in res if "worker" in r]
        reads = sum(r["minibatches"] for r in w) * N_READS
        wall = max(r["seconds"] for r in w)
        out = {"workers": P, "mode": "feeder", "slots": S, "gpu_facing_processes": 1, "refill": True, "start_jitter": args.jitter,
               "reads_per_s": reads / wall, "minibatches": sum(r["minibatches"] for r in w), "seconds": wall,
               "parity": all(r["parity"] for r in w)}
        print(json.dumps(out))
        return 0 if out["parity"] else 2
    finally:
        # no child outlives the shared memory: whatever still runs (a hung feeder holds the GPU) is ended first
        for p_ in list(locals().get("procs", [])) + [locals().get("fp")]:
            if p_ is not None and p_.is_alive():
                p_.terminate()
                p_.join(10)
        del ring, mv
        shm.close()
        shm.unlink()
        meta.close()
        meta.unlink()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workers", type=int, default=4)
    ap.add_argument("--mode", choices=["sync", "pipe", "feeder"], default="sync")
    ap.add_argument("--slots", type=int, default=8, help="feeder mode: ring slots = minibatches in flight (<= 8)")
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--refill", action="store_true")
    ap.add_argument("--jitter", type=int, default=0, help="adapter_start ~ U{100 .. 100 + JITTER} per read (rows carry whole "
                    "reads, file_proc.py:244-260); 0 = every adapter starts at sample 100")
    args = ap.parse_args()
    if args.mode == "feeder":
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

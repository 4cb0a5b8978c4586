"""Run by tests/test_gpu_live.py in a FRESH interpreter: warpdemux_amd.feeder.Feeder the way a WarpDemuX maintainer would
use it -- the parent (which never touches the GPU) creates the feeder, then a ProcessPoolExecutor forks the workers
(file_proc.py:1197-1243), which inherit it and call feeder.demux_batch / feeder.fingerprint_batch on their own minibatches.
Checks: every worker's results against the oracle (several minibatch shapes, success flags, jittered adapter starts: the
packed rows), the argument errors of the worker-side call, that the ring gets back the slot of a worker that was SIGKILLed
with it in its hands, and that a worker is TOLD when the feeder process has died -- while it is still an unreaped zombie --
instead of hanging.  Prints one JSON line."""
import json
import os
import signal
import sys
from concurrent.futures import ProcessPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import multiprocessing as mp  # noqa: E402

import numpy as np  # noqa: E402

from oracle import wdx_oracle as orc  # noqa: E402
from warpdemux_amd import _lib, sig_proc, synth  # noqa: E402
from warpdemux_amd.feeder import Feeder  # noqa: E402

REFS = np.random.default_rng(5).normal(size=(6, 110))
FEEDER = None


def work(widx):
    spec = synth.SynthSpec(n_barcodes=6)
    out = []
    for rep, (n, stride) in enumerate(((64, 9000), (200, 7000), (1, 9000), (37, 6000))):
        mb, a_s, a_e, _ = synth.generate_minibatch(spec, 1000 * widx + 100 * rep, n, stride, start_jitter=900 if rep == 0 else 0)
        ok_in = None
        if rep == 1:
            ok_in = np.ones(n, dtype=np.uint8)
            ok_in[::5] = 0
        res = FEEDER.demux_batch(mb, a_s, a_e, success=ok_in)
        fb = FEEDER.fingerprint_batch(mb, a_s, a_e, success=ok_in)
        fpt, dwell, stats, status = orc.fingerprint_batch(mb, a_s, a_e, orc.SegParams(barcode_num_events=110), ok=ok_in)
        ok = status == 0
        D = orc.dtw_matrix(fpt[ok], REFS, 15, 0.1)
        good = (np.array_equal(res.status, status) and np.array_equal(res.dist[ok].view(np.uint32), D.view(np.uint32))
                and np.array_equal(res.call[ok], orc.argmin_rows(D)) and (res.call[~ok] == -1).all())
        # the ReadResults' arrays: fingerprints, dwell times and the six statistics, bit for bit
        good = (good and np.array_equal(fb.status, status) and np.array_equal(fb.fpt[ok].view(np.uint64), fpt[ok].view(np.uint64))
                and np.array_equal(fb.dwell[ok], dwell[ok]) and np.array_equal(fb.stats[ok].view(np.uint64), stats[ok].view(np.uint64))
                and np.isnan(fb.fpt[~ok]).all())
        out.append(bool(good))
    errs = []
    try:
        FEEDER.demux_batch(np.zeros((300, 9000), np.float32), np.zeros(300, np.int32), np.zeros(300, np.int32))
    except ValueError as e:
        errs.append("do not fit" in str(e))
    try:
        FEEDER.demux_batch(np.zeros((4, 100), np.float32), np.zeros(3, np.int32), np.zeros(4, np.int32))
    except ValueError:
        errs.append(True)
    return os.getpid(), out, errs


def loop_until_stopped(widx):
    """ADVICE r5 (shutdown race): a worker that keeps calling while the parent stops the feeder.  Every call must either
    return a CORRECT result (a minibatch that was in flight when the stop came is finished and handed over -- never the
    previous occupant's numbers) or raise WdxNoDevice; nothing may hang."""
    spec = synth.SynthSpec(n_barcodes=6)
    batches = []
    for rep in range(3):
        mb, a_s, a_e, _ = synth.generate_minibatch(spec, 50_000 + 1000 * widx + 100 * rep, 48 + 16 * rep, 7000)
        fpt, dwell, stats, status = orc.fingerprint_batch(mb, a_s, a_e, orc.SegParams(barcode_num_events=110))
        batches.append((mb, a_s, a_e, status, fpt))
    served, told, wrong = 0, 0, 0
    for it in range(100000):
        mb, a_s, a_e, status, fpt = batches[it % 3]
        try:
            fb = FEEDER.fingerprint_batch(mb, a_s, a_e)
        except _lib.WdxNoDevice:
            told += 1
            break
        ok = status == 0
        if not (np.array_equal(fb.status, status) and np.array_equal(fb.fpt[ok].view(np.uint64), fpt[ok].view(np.uint64))):
            wrong += 1
        served += 1
    return served, told, wrong


def die_with_a_slot(_):
    """A worker that claims a ring slot (the test hook claims exactly like wdx_feeder_run does) and is killed before it
    gives it back -- an OOM kill, pool.terminate()."""
    import ctypes as C

    s = _lib.load().wdx_feeder_selftest(C.c_void_p(FEEDER._base), 1)
    os.kill(os.getpid(), signal.SIGKILL if s >= 0 else signal.SIGTERM)


def after_death(_):
    mb, a_s, a_e, _b = synth.generate_minibatch(synth.SynthSpec(n_barcodes=6), 7, 16, 9000)
    try:
        FEEDER.demux_batch(mb, a_s, a_e)
    except _lib.WdxNoDevice as e:
        return "told: " + str(e)[:60]
    return "served?!"


if __name__ == "__main__":
    import time

    FEEDER = Feeder(REFS, 15, 0.1, sig_proc.SegParams(barcode_num_events=110), max_reads=256, stride=9000, n_slots=6)
    ctx = mp.get_context("fork")
    with ProcessPoolExecutor(max_workers=4, mp_context=ctx) as ex:
        res = list(ex.map(work, range(4)))
    served = FEEDER.served()
    # a worker dies with a slot in its hands: the serving feeder's idle loop takes the slot back (the dead worker is not
    # reaped here: it stays a zombie, which is the hard case)
    killer = ctx.Process(target=die_with_a_slot, args=(0,))
    killer.start()
    t0 = time.monotonic()
    low = FEEDER.n_slots
    while time.monotonic() - t0 < 20:
        st = FEEDER.stats()
        low = min(low, st["free_slots"])
        if st["reclaimed"] >= 1 and st["free_slots"] == FEEDER.n_slots:
            break
        time.sleep(0.005)
    reclaim = dict(FEEDER.stats(), lowest_free_seen=low, seconds=round(time.monotonic() - t0, 3))
    killer.join(10)
    with ProcessPoolExecutor(max_workers=2, mp_context=ctx) as ex:      # the ring still serves
        again = list(ex.map(work, range(2)))
    os.kill(FEEDER._proc.pid, signal.SIGKILL)        # the feeder dies without a word -- and is NOT reaped before the worker calls
    with ProcessPoolExecutor(max_workers=1, mp_context=ctx) as ex:
        told = list(ex.map(after_death, range(1)))[0]
    zombie = open(f"/proc/{FEEDER._proc.pid}/stat").read().rsplit(") ", 1)[1][0]
    FEEDER.close()
    # a second feeder, stopped (close) while six workers hammer it: correct results or WdxNoDevice, nothing else, no hang
    FEEDER = Feeder(REFS, 15, 0.1, sig_proc.SegParams(barcode_num_events=110), max_reads=256, stride=9000, n_slots=4)
    with ProcessPoolExecutor(max_workers=6, mp_context=ctx) as ex:
        futs = [ex.submit(loop_until_stopped, w) for w in range(6)]
        t0 = time.monotonic()
        while FEEDER.served() < 60 and time.monotonic() - t0 < 60:
            time.sleep(0.002)
        FEEDER.close()
        stopped = [f.result(timeout=120) for f in futs]
    print(json.dumps({"pids": sorted({r[0] for r in res}), "ok": [r[1] for r in res] + [r[1] for r in again],
                      "errs": [r[2] for r in res], "parent": os.getpid(), "served": served, "reclaim": reclaim,
                      "feeder_state_when_the_worker_was_told": zombie, "after_death": told,
                      "stopped_while_busy": {"served": [s_[0] for s_ in stopped], "told": [s_[1] for s_ in stopped],
                                             "wrong": [s_[2] for s_ in stopped]}}))

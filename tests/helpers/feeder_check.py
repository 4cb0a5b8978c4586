"""Run by tests/test_gpu_live.py in a FRESH interpreter: warpdemux_amd.feeder.Feeder the way a WarpDemuX maintainer would
use it -- the parent (which never touches the GPU) creates the feeder, then a ProcessPoolExecutor forks the workers
(file_proc.py:1197-1243), which inherit it and call feeder.demux_batch on their own minibatches.  Checks: every worker's
results against the oracle (several minibatch shapes, success flags), the argument errors of the worker-side call, and
that a worker is TOLD when the feeder process has died instead of hanging.  Prints one JSON line."""
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
        mb, a_s, a_e, _ = synth.generate_minibatch(spec, 1000 * widx + 100 * rep, n, stride)
        ok_in = None
        if rep == 1:
            ok_in = np.ones(n, dtype=np.uint8)
            ok_in[::5] = 0
        res = FEEDER.demux_batch(mb, a_s, a_e, success=ok_in)
        fpt, dwell, stats, status = orc.fingerprint_batch(mb, a_s, a_e, orc.SegParams(barcode_num_events=110), ok=ok_in)
        ok = status == 0
        D = orc.dtw_matrix(fpt[ok], REFS, 15, 0.1)
        good = (np.array_equal(res.status, status) and np.array_equal(res.dist[ok].view(np.uint32), D.view(np.uint32))
                and np.array_equal(res.call[ok], orc.argmin_rows(D)) and (res.call[~ok] == -1).all())
        out.append(bool(good))
    errs = []
    try:
        FEEDER.demux_batch(np.zeros((300, 9000), np.float32), np.zeros(300, np.int32), np.zeros(300, np.int32))
    except ValueError as e:
        errs.append("does not fit" in str(e))
    try:
        FEEDER.demux_batch(np.zeros((4, 100), np.float32), np.zeros(3, np.int32), np.zeros(4, np.int32))
    except ValueError:
        errs.append(True)
    return os.getpid(), out, errs


def after_death(_):
    mb, a_s, a_e, _b = synth.generate_minibatch(synth.SynthSpec(n_barcodes=6), 7, 16, 9000)
    try:
        FEEDER.demux_batch(mb, a_s, a_e)
    except _lib.WdxNoDevice as e:
        return "told: " + str(e)[:60]
    return "served?!"


if __name__ == "__main__":
    FEEDER = Feeder(REFS, 15, 0.1, sig_proc.SegParams(barcode_num_events=110), max_reads=256, stride=9000, n_slots=6)
    ctx = mp.get_context("fork")
    with ProcessPoolExecutor(max_workers=4, mp_context=ctx) as ex:
        res = list(ex.map(work, range(4)))
    served = FEEDER.served()
    os.kill(FEEDER._proc.pid, signal.SIGKILL)        # the feeder dies without a word
    FEEDER._proc.join(10)
    with ProcessPoolExecutor(max_workers=1, mp_context=ctx) as ex:
        told = list(ex.map(after_death, range(1)))[0]
    FEEDER.close()
    print(json.dumps({"pids": sorted({r[0] for r in res}), "ok": [r[1] for r in res], "errs": [r[2] for r in res],
                      "parent": os.getpid(), "served": served, "after_death": told}))

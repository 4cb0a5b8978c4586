"""Run by tests/test_gpu_live.py in a FRESH interpreter: the reference's offline orchestrator forks its workers
from a parent that has imported everything (file_proc.py:1197-1243, ProcessPoolExecutor with the default fork
start method).  Here the parent imports the engine (without creating a context -- HIP must not exist before the
fork), forks 4 workers, and each worker creates its own context on the one GPU, fingerprints + demuxes its own
minibatch and checks it against the oracle.  Prints one JSON line."""
import json
import os
import sys
from concurrent.futures import ProcessPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from oracle import wdx_oracle as orc  # noqa: E402
from warpdemux_amd import _lib, parallel_distances as pdist, sig_proc, synth  # noqa: E402

_lib.load()          # dlopen in the parent is fine; no context, no HIP call
REFS = np.random.default_rng(5).normal(size=(6, 110))


def work(widx):
    spec = synth.SynthSpec(n_barcodes=6)
    out = []
    for rep in range(3):     # a worker handles several minibatches, like file_proc's pool
        mb, a_s, a_e, _ = synth.generate_minibatch(spec, 1000 * widx + 100 * rep, 64, 9000)
        p = sig_proc.SegParams(barcode_num_events=110)
        fb = sig_proc.fingerprint_batch(mb, a_s, a_e, p)
        fpt, dwell, stats, status = orc.fingerprint_batch(mb, a_s, a_e, orc.SegParams(barcode_num_events=110))
        ok = status == 0
        D = pdist.distance_matrix_to(fb.fpt[ok], REFS, window=15, penalty=0.1, n_jobs=1)
        good = (np.array_equal(fb.status, status) and np.array_equal(fb.fpt[ok], fpt[ok])
                and np.array_equal(fb.dwell[ok], dwell[ok])
                and np.array_equal(D, orc.dtw_matrix(fpt[ok], REFS, 15, 0.1)))
        out.append(bool(good and ok.sum() > 50))
    return os.getpid(), out


if __name__ == "__main__":
    with ProcessPoolExecutor(max_workers=4) as ex:   # default start method on Linux / py3.10: fork
        res = list(ex.map(work, range(4)))
    print(json.dumps({"pids": sorted({r[0] for r in res}), "ok": [r[1] for r in res], "parent": os.getpid()}))

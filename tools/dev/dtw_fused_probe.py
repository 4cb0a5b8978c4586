"""Development probe: how often does the fused-cell float32 differ from the reference's (WDX_OPT_DTW_UNFUSED = 3 vs 1),
and does the default mode (0) always return the reference's bits?  Prints one line per seed."""
import sys, time
import numpy as np
sys.path.insert(0, '/root/repo')
from warpdemux_amd import _lib, parallel_distances as pdist

ctx = _lib.default_context()
def run(X, Y, w, p, mode):
    ctx.set_option(_lib.OPT_DTW_UNFUSED, mode)
    try:
        return pdist.distance_matrix_to(X, Y, window=w, penalty=p, n_jobs=1)
    finally:
        ctx.set_option(_lib.OPT_DTW_UNFUSED, 0)

tot = diff3 = diff0 = diff2 = 0
for shape in ((20000, 2601, 25), (400000, 10, 110)):
    nX, nY, L = shape
    for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
        rng = np.random.default_rng(1000 * L + seed)
        Y = rng.normal(size=(nY, L))
        X = Y[rng.integers(0, nY, nX)] + 0.7 * rng.normal(size=(nX, L)) if seed % 2 else rng.normal(size=(nX, L))
        t = time.time()
        d1 = run(X, Y, 15, 0.1, 1)
        d0 = run(X, Y, 15, 0.1, 0)
        d3 = run(X, Y, 15, 0.1, 3)
        n3 = int(np.count_nonzero(d3 != d1)); n0 = int(np.count_nonzero(d0 != d1))
        w3 = np.argwhere(d3 != d1)[:4].tolist()
        print(shape, 'seed', seed, 'pairs', d1.size, 'fused!=ref', n3, w3, 'default!=ref', n0, f'{time.time()-t:.1f}s', flush=True)
        tot += d1.size; diff3 += n3; diff0 += n0
print('total pairs', tot, 'fused float32 differs', diff3, 'default differs', diff0)

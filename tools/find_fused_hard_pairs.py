"""Finds (query, reference) pairs whose DTW distance differs as float32 between fused cells (one v_fma_f64 per cell,
WDX_OPT_DTW_UNFUSED = 3) and the reference's six operations (= 1): about two pairs in 10^9.  The default mode (0) must
return the reference's float32 on them -- they are what wdx_dtw.hip's dtw_unsettled() exists for.  Needs an MI355X; writes
gpurun_out/g10_dtw_fused_hard_pairs.npz {x25, y25, x110, y110, window, penalty, pairs_searched} (committed as
tests/golden/g10_dtw_fused_hard_pairs.npz) -- inputs only: the expected values are computed by the oracle when the test
runs (tests/test_gpu_parity.py::test_dtw_fused_cells_settle_to_the_reference_bits).  Round 6, 160 rounds: 224 pairs of
8.3e11 at L = 25, 24 of 8.2e10 at L = 110 (profiles/r06d_fused_pair_search.txt; the fixture keeps the first 40 and all
24); the default mode returned the reference's float32 on every pair searched.
Usage: python3 tools/find_fused_hard_pairs.py [rounds]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from warpdemux_amd import _lib, sig_proc  # noqa: E402
from warpdemux_amd.engine import DemuxEngine  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
found = {}
total = {}
for L, nY, n in ((25, 2601, 2_000_000), (110, 512, 1_000_000)):
    rng = np.random.default_rng(L)
    Y = rng.normal(size=(nY, L))
    eng = DemuxEngine(Y, 15, 0.1, sig_proc.SegParams(barcode_num_events=L))
    Yd = torch.from_numpy(Y).to(eng.tdev)
    xs, ys = [], []
    pairs = 0
    for it in range(rounds):
        g = torch.Generator(device=eng.tdev)
        g.manual_seed(1000 * L + it)
        X = torch.randn((n, L), dtype=torch.float64, device=eng.tdev, generator=g)
        if it % 2:   # reads near a reference, like demultiplexed fingerprints
            lab = torch.randint(0, nY, (n,), device=eng.tdev, generator=g)
            X = Yd[lab] + 0.7 * X
        out = {}
        for mode in (1, 3, 0):
            eng.ctx.set_option(_lib.OPT_DTW_UNFUSED, mode)
            out[mode] = eng.dtw(X, want_argmin=False)[0]
        eng.ctx.set_option(_lib.OPT_DTW_UNFUSED, 0)
        torch.cuda.synchronize()
        bad0 = int((out[0] != out[1]).sum())
        # (row-wise first: torch.nonzero on > 2^31 elements fails)
        rows = torch.nonzero((out[3] != out[1]).any(dim=1)).flatten()
        idx = torch.tensor([(int(i), int(j)) for i in rows.tolist()
                            for j in torch.nonzero(out[3][i] != out[1][i]).flatten().tolist()], dtype=torch.int64).reshape(-1, 2)
        pairs += n * nY
        print(f"L={L} round {it}: {n * nY} pairs, fused float32 differs on {idx.shape[0]}, default mode differs on {bad0}", flush=True)
        assert bad0 == 0
        for i, j in idx.tolist():
            xs.append(X[i].cpu().numpy())
            ys.append(Y[j])
        del out, X
    found[L] = (np.array(xs).reshape(-1, L), np.array(ys).reshape(-1, L))
    total[L] = pairs
    del eng
    torch.cuda.empty_cache()
dst = os.path.join(ROOT, "gpurun_out", "g10_dtw_fused_hard_pairs.npz")
os.makedirs(os.path.dirname(dst), exist_ok=True)
np.savez_compressed(dst, x25=found[25][0], y25=found[25][1], x110=found[110][0], y110=found[110][1],
                    window=np.int32(15), penalty=np.float64(0.1), pairs_searched=np.array([total[25], total[110]]))
print("wrote", dst, {L: found[L][0].shape[0] for L in found}, "of", total)

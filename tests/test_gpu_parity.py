"""Parity of the HIP engine (through the C ABI) against the CPU oracle and the reference-derived
golden fixtures.  Needs a real MI355X: run with `pytest -m gpu`.

Bar: integer outputs (status, change-point-derived dwell, argmin) bit-exact; float64 fingerprints
and stats bit-exact (same operation order, no FMA); float32 DTW distances bit-exact against the
oracle (tolerance the north star allows: 1e-5 relative -- asserted separately so a rounding
difference in sqrt would be reported as such, not hidden).  The DTW kernels run fused cells first and
settle every pair whose float32 could differ on the reference's operations (wdx_dtw.hip: dtw_unsettled;
test_dtw_fused_cells_settle_to_the_reference_bits).
"""
import os

import numpy as np
import pytest

from oracle import wdx_oracle as orc
from warpdemux_amd import _lib, parallel_distances as pdist, sig_proc, synth

pytestmark = pytest.mark.gpu

RTOL = 1e-5  # north-star tolerance on the float distances


import contextlib


@contextlib.contextmanager
def _exact_path(device=None):
    """Fingerprint on the exact general kernel only (diagnostic context option; the product path leaves it off)."""
    ctx = _lib.default_context(device)
    ctx.set_option(_lib.OPT_EXACT_PATH, 1)
    try:
        yield
    finally:
        ctx.set_option(_lib.OPT_EXACT_PATH, 0)


def _same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and np.array_equal(a, b, equal_nan=True)


def _check_dist(got, ref):
    assert got.dtype == np.float32 and got.shape == ref.shape
    fin = np.isfinite(ref)
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    assert np.array_equal(np.isinf(got), np.isinf(ref))
    g64, r64 = got[fin].astype(np.float64), ref[fin].astype(np.float64)
    rel = np.abs(g64 - r64) / np.maximum(np.abs(r64), 1e-300)
    assert rel.size == 0 or rel.max() <= RTOL, f"max rel err {rel.max()}"
    assert _same(got, ref), f"not bit-exact: {np.count_nonzero(got != ref)} of {got.size} differ (max rel {rel.max() if rel.size else 0})"


# ---------------------------------------------------------------------------------------- DTW ----

@pytest.mark.parametrize("nX,nY,L,w,p", [
    (1000, 10, 110, 15, 0.1),     # R1 benchmark regime
    (1000, 4, 110, 15, 0.1),
    (300, 851, 25, 15, 0.1),      # R2 shipped-model regime (WDX4)
    (64, 6, 110, 15, 0.0),
    (5000, 10, 110, 15, 0.1),     # fused-argmin path (grid.x >= 2048 needs more; see below)
])
def test_dtw_matrix_regimes(nX, nY, L, w, p):
    rng = np.random.default_rng(nX * 7 + nY)
    X, Y = rng.normal(size=(nX, L)), rng.normal(size=(nY, L))
    ref = orc.dtw_matrix(X, Y, w, p)
    got, am = pdist.nearest_reference(X, Y, w, p)
    _check_dist(got, ref)
    assert np.array_equal(am, np.argmin(ref, axis=1))
    assert _same(pdist.distance_matrix_to(X, Y, window=w, penalty=p, n_jobs=1), ref)


def test_dtw_large_batch_fused_argmin():
    rng = np.random.default_rng(11)
    nX, nY, L = 140_000, 10, 110   # > 2048 waves -> in-kernel argmin
    Y = rng.normal(size=(nY, L))
    lab = rng.integers(0, nY, nX)
    X = Y[lab] + 0.7 * rng.normal(size=(nX, L))
    got, am = pdist.nearest_reference(X, Y, 15, 0.1)
    sub = rng.choice(nX, 3000, replace=False)
    ref = orc.dtw_matrix(X[sub], Y, 15, 0.1)
    _check_dist(got[sub], ref)
    assert np.array_equal(am, np.argmin(got, axis=1))
    assert np.array_equal(am[sub], np.argmin(ref, axis=1))
    assert (am == lab).mean() > 0.95


@pytest.mark.parametrize("w", [None, 0, 1, 2, 5, 8, 9, 15, 16, 17, 25, 32, 33, 60, 110, 500])
@pytest.mark.parametrize("L", [25, 110])
def test_dtw_windows(w, L):
    rng = np.random.default_rng(3)
    X, Y = rng.normal(size=(70, L)), rng.normal(size=(9, L))
    for p in (0.1, None, 1.5):
        ref = orc.dtw_matrix(X, Y, w, p)
        _check_dist(pdist.distance_matrix_to(X, Y, window=w, penalty=p, n_jobs=1), ref)


@pytest.mark.parametrize("L", [1, 2, 3, 14, 15, 16, 28, 29, 30, 31, 57, 200])
def test_dtw_lengths(L):
    rng = np.random.default_rng(L)
    X, Y = rng.normal(size=(65, L)), rng.normal(size=(5, L))
    _check_dist(pdist.distance_matrix_to(X, Y, window=15, penalty=0.1, n_jobs=1), orc.dtw_matrix(X, Y, 15, 0.1))


@pytest.mark.parametrize("nX", [0, 1, 2, 3, 63, 64, 65])
def test_dtw_few_reads_many_refs(nX):
    """live mode / per-read calls: lanes run over the references"""
    rng = np.random.default_rng(5)
    Y = rng.normal(size=(1368, 25))
    X = rng.normal(size=(nX, 25))
    got, am = pdist.nearest_reference(X, Y, 15, 0.1)
    ref = orc.dtw_matrix(X, Y, 15, 0.1)
    _check_dist(got, ref)
    if nX:
        assert np.array_equal(am, np.argmin(ref, axis=1))
    assert pdist.distance_matrix_to(X[:0], Y, 15, 0.1, n_jobs=1).shape == (0, 1368)
    assert pdist.distance_matrix_to(X, Y[:0], 15, 0.1, n_jobs=1).shape == (nX, 0)


def test_dtw_nan_inf_and_ties():
    rng = np.random.default_rng(6)
    X, Y = rng.normal(size=(130, 110)), rng.normal(size=(10, 110))
    X[3, 17] = np.nan
    X[64, 0] = np.nan
    Y[4, 109] = np.nan
    Y[7] = Y[2]            # duplicate reference -> exact tie, argmin takes the lower index
    X[10] = Y[2]           # distance exactly 0 to refs 2 and 7
    got, am = pdist.nearest_reference(X, Y, 15, 0.1)
    ref = orc.dtw_matrix(X, Y, 15, 0.1)
    _check_dist(got, ref)
    assert np.isnan(got[3]).all() and np.isnan(got[:, 4]).all()
    assert np.array_equal(am, orc.argmin_rows(ref))
    assert np.array_equal(am, np.argmin(ref, axis=1))   # a NaN column wins every row, like np.argmin
    Y[4, 109] = 0.25
    got, am = pdist.nearest_reference(X, Y, 15, 0.1)
    ref = orc.dtw_matrix(X, Y, 15, 0.1)
    _check_dist(got, ref)
    assert np.array_equal(am, np.argmin(ref, axis=1))
    assert got[10, 2] == 0.0 and got[10, 7] == 0.0 and am[10] == 2


@contextlib.contextmanager
def _dtw_mode(mode):
    """WDX_OPT_DTW_UNFUSED: 0 product | 1 the reference's six operations only | 2 fused + every pair settled | 3 fused, never settled"""
    ctx = _lib.default_context()
    ctx.set_option(_lib.OPT_DTW_UNFUSED, mode)
    try:
        yield
    finally:
        ctx.set_option(_lib.OPT_DTW_UNFUSED, 0)


def test_dtw_fused_cells_settle_to_the_reference_bits():
    """The DTW kernels compute a cell as one v_fma_f64 (five float64 operations, not six) and run a pair again on the
    reference's operations whenever its float32 could differ (wdx_dtw.hip: dtw_unsettled).  Fixture g10: 64 of the 248
    pairs among 9.1e11 searched on an MI355X whose fused float32 DOES differ (tools/find_fused_hard_pairs.py,
    profiles/r06d_fused_pair_search.txt; inputs only, the expected distances are the oracle's, computed here).  On them, and on everything around them: the product mode, the
    six-operations mode and the settle-everything mode return the oracle's bits; the never-settle diagnostic does not
    (the positive control: these pairs are what the check exists for).  Scaled inputs reach the underflow / overflow
    guards of the check."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g10_dtw_fused_hard_pairs.npz"))
    w, p = int(g["window"]), float(g["penalty"])
    for L, npad in ((25, 3000), (110, 20000)):   # (enough pairs for the band / short kernels: > 16 384)
        x, y = g[f"x{L}"], g[f"y{L}"]
        m = x.shape[0]
        assert m >= 1
        rng = np.random.default_rng(L + 77)
        X = rng.normal(size=(npad, L))
        at = rng.choice(npad, m, replace=False)
        X[at] = x
        ref = orc.dtw_matrix(X, y, w, p)
        for mode in (0, 1, 2):
            with _dtw_mode(mode):
                got, am = pdist.nearest_reference(X, y, w, p)
            _check_dist(got, ref)
            assert np.array_equal(am, np.argmin(ref, axis=1))
        with _dtw_mode(3):
            raw = pdist.distance_matrix_to(X, y, window=w, penalty=p, n_jobs=1)
        assert all(raw[at[i], i] != ref[at[i], i] for i in range(m)), "the fixture's pairs no longer tell fused from unfused cells"
        assert m <= np.count_nonzero(raw != ref) <= 2 * m   # (a query that is hard against two references sits in the fixture twice)
        assert np.allclose(raw, ref, rtol=2e-7, atol=0.0)   # (one float32 ulp)
    # the guards: sums below 1e-280 / above 1e280 are settled on the reference's operations (underflow makes errors absolute)
    rng = np.random.default_rng(2)
    X, Y = rng.normal(size=(2000, 110)), rng.normal(size=(10, 110))
    for scale in (1e-160, 1e-150, 1e-139, 1e139, 1e150, 1e160):
        ref = orc.dtw_matrix(X * scale, Y * scale, w, p)
        for mode in (0, 2):
            with _dtw_mode(mode):
                _check_dist(pdist.distance_matrix_to(X * scale, Y * scale, window=w, penalty=p, n_jobs=1), ref)
    # windows served by the masked band kernels and short rows, in the settle-everything mode
    for wv, Lv in ((3, 40), (8, 57), (12, 110), (16, 31), (20, 110), (32, 64)):
        X, Y = rng.normal(size=(700, Lv)), rng.normal(size=(30, Lv))
        ref = orc.dtw_matrix(X, Y, wv, 0.1)
        for mode in (0, 2):
            with _dtw_mode(mode):
                _check_dist(pdist.distance_matrix_to(X, Y, window=wv, penalty=0.1, n_jobs=1), ref)


def test_dtw_fused_and_six_operation_forms_agree_at_the_headline_size():
    """BASELINE's C3 shape at full size without the oracle: 5 M device-resident 110-point fingerprints x 10 references
    (one launch of the headline run) and 100 000 x 2 601 x 25 points (the shipped models' shape) -- the product mode must
    return the six-operation form's float32 and argmin everywhere (the six-operation form IS the oracle's arithmetic:
    every other DTW test), on random and on near-reference queries."""
    import torch
    from warpdemux_amd.engine import DemuxEngine
    for L, nY, n in ((110, 10, 5_000_000), (25, 2601, 100_000)):
        rng = np.random.default_rng(L)
        Y = rng.normal(size=(nY, L))
        eng = DemuxEngine(Y, 15, 0.1, sig_proc.SegParams(barcode_num_events=L))
        Yd = torch.from_numpy(Y).to(eng.tdev)
        for near in (False, True):
            g = torch.Generator(device=eng.tdev)
            g.manual_seed(7 * L + int(near))
            X = torch.randn((n, L), dtype=torch.float64, device=eng.tdev, generator=g)
            if near:
                X = Yd[torch.randint(0, nY, (n,), device=eng.tdev, generator=g)] + 0.7 * X
            out = {}
            for mode in (1, 0):
                eng.ctx.set_option(_lib.OPT_DTW_UNFUSED, mode)
                try:
                    out[mode] = eng.dtw(X, want_argmin=True)
                finally:
                    eng.ctx.set_option(_lib.OPT_DTW_UNFUSED, 0)
            assert bool(torch.equal(out[0][0], out[1][0])), int((out[0][0] != out[1][0]).sum())
            assert bool(torch.equal(out[0][1], out[1][1]))
            assert bool(torch.isfinite(out[0][0]).all())
            del out, X
        del eng
        torch.cuda.empty_cache()


def test_dtw_device_rows_with_nan_and_inf_take_the_lazy_sweep():
    """wdx_dtw_matrix_dev on row-major device fingerprints carries no NaN flags: the band kernel sweeps a wave's rows only
    when a result is not finite (a NaN sample leaves no finite cell behind it; wdx_dtw.hip).  Rows with a NaN at the first,
    a middle and the last sample, rows with +inf, -inf and both, in waves with and without clean rows, against the oracle
    in the product mode, the six-operation mode and the settle-everything mode."""
    import torch
    from warpdemux_amd.engine import DemuxEngine
    rng = np.random.default_rng(41)
    L, nY, n = 110, 10, 40_000   # (> 16 384 pairs: the band kernel; row-major device input)
    Y = rng.normal(size=(nY, L))
    X = rng.normal(size=(n, L))
    X[5, 0] = np.nan
    X[64 + 7, 55] = np.nan
    X[128 + 63, L - 1] = np.nan
    X[300, 17] = np.inf
    X[301, 18] = -np.inf
    X[302, 3], X[302, 90] = np.inf, -np.inf
    X[303, 3], X[303, 4] = np.inf, np.nan
    X[1000:1064, 40] = np.nan               # a whole wave of failed reads
    X[rng.choice(n, 500, replace=False), rng.integers(0, L, 500)] = np.nan
    eng = DemuxEngine(Y, 15, 0.1, sig_proc.SegParams(barcode_num_events=L))
    Xd = torch.from_numpy(X).to(eng.tdev)
    sub = np.unique(np.concatenate([np.arange(0, 2048), rng.choice(n, 3000, replace=False)]))
    with np.errstate(invalid="ignore", over="ignore"):
        ref = orc.dtw_matrix(X[sub], Y, 15, 0.1)
    for mode in (0, 1, 2):
        eng.ctx.set_option(_lib.OPT_DTW_UNFUSED, mode)
        try:
            d, am = eng.dtw(Xd, want_argmin=True)
        finally:
            eng.ctx.set_option(_lib.OPT_DTW_UNFUSED, 0)
        d, am = d.cpu().numpy(), am.cpu().numpy()
        _check_dist(d[sub], ref)
        assert np.array_equal(am[sub], orc.argmin_rows(ref))
        assert np.isnan(d[5]).all() and np.isnan(d[1000:1064]).all() and np.isinf(d[300]).all() and np.isinf(d[301]).all()
        nanrow = np.isnan(X).any(axis=1)
        assert np.array_equal(np.isnan(d).any(axis=1), nanrow) and np.array_equal(np.isnan(d).all(axis=1), nanrow)


def test_dtw_symmetry_and_block_api():
    rng = np.random.default_rng(8)
    X = rng.normal(size=(150, 25))
    D = pdist.parallel_distance_matrix(X, block_size=64, n_jobs=2, window=15, penalty=0.1)
    assert D.shape == (150, 150) and D.dtype == np.float32
    assert np.array_equal(D, D.T) and np.all(np.diag(D) == 0)
    _check_dist(D, orc.dtw_matrix(X, X, 15, 0.1))
    sub = pdist.parallel_distance_matrix(X, block_size=50, n_jobs=2, subset=((10, 40), (100, 150)), window=15, penalty=0.1)
    assert _same(sub, D[10:40, 100:150])
    i, j, blk = pdist.compute_block_distance((np.arange(5, 20), np.arange(30, 33)), X, 15, 0.1)
    assert _same(blk, D[5:20, 30:33])
    to = pdist.distance_matrix_to(X[:20], X[20:], window=15, penalty=0.1, block_size=7, n_jobs=4)
    assert _same(to, D[:20, 20:])


def test_refs_cache_distinguishes_content():
    rng = np.random.default_rng(9)
    X, Y1 = rng.normal(size=(80, 25)), rng.normal(size=(30, 25))
    Y2 = Y1.copy()
    Y2[5, 5] += 1e-9
    a = pdist.distance_matrix_to(X, Y1, 15, 0.1, n_jobs=1)
    b = pdist.distance_matrix_to(X, Y2, 15, 0.1, n_jobs=1)
    c = pdist.distance_matrix_to(X, Y1, 15, 0.2, n_jobs=1)
    _check_dist(a, orc.dtw_matrix(X, Y1, 15, 0.1))
    _check_dist(b, orc.dtw_matrix(X, Y2, 15, 0.1))
    _check_dist(c, orc.dtw_matrix(X, Y1, 15, 0.2))


# --------------------------------------------------------------------------------- fingerprint ----

def _params_from(g, k):
    pad, sig_norm, d, w, E, acc, seg_norm, K = (int(v) for v in g[f"params_{k}"])
    inv = {0: "none", 1: "mean", 2: "median"}
    kw = dict(padding=pad, sig_norm=inv[sig_norm], outlier_thresh=float(g[f"thresh_{k}"]),
              min_obs_per_base=d, running_stat_width=w, num_events=E, accept_less_cpts=bool(acc),
              seg_norm=inv[seg_norm], barcode_num_events=K)
    c64 = bool(int(g[f"clip64_{k}"]))   # fixture made with the NumPy-1.x evaluation of the clip bounds
    return (sig_proc.SegParams(clip_bounds="float64" if c64 else "float32", **kw),
            orc.SegParams(clip_bounds_f64=c64, **kw))


def test_fingerprint_golden_vectors(golden_dir):
    """Reference-derived fixtures (tests/golden/make_golden.py) through wdx_fingerprint_batch."""
    g = np.load(os.path.join(golden_dir, "g4_fingerprint.npz"))
    n = int(g["n"])
    seen = set()
    for k in range(n):
        tag = str(g[f"tag_{k}"])
        a_start, a_end, ok = (int(v) for v in g[f"args_{k}"])
        p_hip, p_orc = _params_from(g, k)
        row = g[f"row_{k}"]
        fb = sig_proc.fingerprint_batch(row.reshape(1, -1), [a_start], [a_end], p_hip, success=[ok])
        st_ref = int(g[f"status_{k}"])
        assert int(fb.status[0]) == st_ref, f"case {k} ({tag}): status {fb.status[0]} != {st_ref}"
        seen.add(st_ref)
        if st_ref != 0:
            assert np.isnan(fb.fpt).all() and (fb.dwell == 0).all() and np.isnan(fb.stats).all()
            continue
        if tag == "noise_free_steps":
            # exact score ties: the reference's own answer depends on np.argsort's unstable order;
            # the engine follows the oracle's documented tie rule instead
            o = orc.fingerprint_one(row, a_start, a_end, p_orc)
            assert _same(fb.fpt[0], o["fpt"]) and _same(fb.dwell[0], o["dwell"]) and _same(fb.stats[0], o["stats"])
            continue
        assert _same(fb.fpt[0], g[f"fpt_{k}"]), f"case {k} ({tag}) fpt"
        assert _same(fb.dwell[0], g[f"dwell_{k}"]), f"case {k} ({tag}) dwell"
        assert _same(fb.stats[0], g[f"stats_{k}"]), f"case {k} ({tag}) stats"
    assert {0, 1, 3, 4, 5} <= seen


@pytest.mark.parametrize("K", [25, 110])
def test_fingerprint_minibatch_vs_oracle(K):
    spec = synth.SynthSpec(n_barcodes=10)
    mb, a_s, a_e, _ = synth.generate_minibatch(spec, 5000, 700, 10000)
    ok = np.ones(700, dtype=np.uint8)
    ok[[5, 77]] = 0
    mb[9, 3000:] = np.nan            # NaN tail inside the adapter window
    a_e[11] = a_s[11] + 50           # too few peaks -> "event segmentation failed"
    a_e[13] = a_s[13] - 100          # 100-sample window: round(100/110/2) == 0 -> "unknown"
    a_e[12] = a_s[12] + 900          # short adapter -> parameter shrink
    ph = sig_proc.SegParams(barcode_num_events=K)
    po = orc.SegParams(barcode_num_events=K)
    fb = sig_proc.fingerprint_batch(mb, a_s, a_e, ph, success=ok)
    fpt, dwell, stats, status = orc.fingerprint_batch(mb, a_s, a_e, po, ok=ok)
    assert np.array_equal(fb.status, status)
    assert (status == 0).sum() > 650 and {1, 3, 4, 5} <= set(status.tolist())
    good = status == 0
    assert _same(fb.dwell[good], dwell[good])
    assert _same(fb.fpt[good], fpt[good])
    assert _same(fb.stats[good], stats[good])
    assert np.isnan(fb.fpt[~good]).all()


def test_fast_and_slow_paths_agree(monkeypatch):
    """The 256-thread fast kernel and the exact slow path (forced with the WDX_OPT_EXACT_PATH context option) must give
    identical bits, with and without the optional stats, for every segmentation normalisation."""
    spec = synth.SynthSpec(n_barcodes=10)
    mb, a_s, a_e, _ = synth.generate_minibatch(spec, 77_000, 1200, 9000)
    rng = np.random.default_rng(1)
    mb[3, 1500:1510] = np.nan                     # NaN -> slow path inside the fast launch
    mb[4, 200:4000] = np.round(mb[4, 200:4000])   # integer-valued samples: exact score ties/plateaus
    mb[5, :] = np.float32(80.0)                   # constant
    a_e[6] = a_s[6] + 1100                        # shrunk window width -> slow path
    for seg_norm in ("mean", "median", "none"):
        for K in (25, 110):
            ph = sig_proc.SegParams(barcode_num_events=K, seg_norm=seg_norm)
            po = orc.SegParams(barcode_num_events=K, seg_norm=seg_norm)
            fast = sig_proc.fingerprint_batch(mb, a_s, a_e, ph)
            with _exact_path():
                slow = sig_proc.fingerprint_batch(mb, a_s, a_e, ph)
            assert np.array_equal(fast.status, slow.status)
            assert _same(fast.fpt, slow.fpt) and _same(fast.dwell, slow.dwell) and _same(fast.stats, slow.stats)
            fpt, dwell, stats, status = orc.fingerprint_batch(mb, a_s, a_e, po)
            assert np.array_equal(fast.status, status)
            good = status == 0
            assert good.sum() > 1150
            assert _same(fast.fpt[good], fpt[good]) and _same(fast.dwell[good], dwell[good])
            assert _same(fast.stats[good], stats[good])


@contextlib.contextmanager
def _option(opt, value, device=None):
    ctx = _lib.default_context(device)
    ctx.set_option(opt, value)
    try:
        yield
    finally:
        ctx.set_option(opt, 0)


def _chain(device=None):
    """The launch chain of large batches (5120-sample main kernel on approximate score keys -> 6144-sample list
    kernel -> 8192-sample list kernel -> exact-scores retry -> exact general kernel) for a batch of any size
    (by default batches below 2048 reads take one 6144-sample launch with exact scores)."""
    return _option(_lib.OPT_FAST_CHAIN_MIN_READS, 1, device)


def test_banded_score_keys_match_exact_scores_and_oracle():
    """The fast kernel takes its order decisions (local maxima, suppression, top-E cut) on approximate score keys and
    only outside an error band; anything inside the band is redone from the reference's exact scores.  The result
    must be the bits of the exact-scores kernel (WDX_OPT_FAST_EXACT_SCORES) and of the oracle on data that stresses
    the band: a large offset with little noise (cancellation in S2 - S1^2/12 -> variance floor), flat clipped
    stretches and near-constant windows (ties, plateaus), tiny and huge scales."""
    spec = synth.SynthSpec(n_barcodes=10)
    mb, a_s, a_e, _ = synth.generate_minibatch(spec, 123_000, 500, 9000)
    rng = np.random.default_rng(5)
    cases = {"plain": mb.copy()}
    off = mb.copy()
    off[:, :] = (mb - np.float32(80.0)) * np.float32(0.002) + np.float32(900.0)  # sigma ~ 0.004 on a level of 900
    cases["offset"] = off
    cases["tiny"] = mb * np.float32(1e-12)
    cases["huge"] = mb * np.float32(3e9)
    flat = mb.copy()
    for r in range(0, 500, 3):  # stretches of equal samples inside the window: zero-variance windows, ties
        p0 = int(rng.integers(300, 3000))
        flat[r, p0:p0 + int(rng.integers(10, 90))] = flat[r, p0]
    cases["flat"] = flat
    rep = mb.copy()
    rep[:, 1000:1600] = rep[:, 400:1000]  # repeated stretch: equal scores 600 positions apart (top-E ties)
    cases["repeat"] = rep
    for name, data in cases.items():
        for K in (25, 110):
            ph = sig_proc.SegParams(barcode_num_events=K)
            with _chain():
                a = sig_proc.fingerprint_batch(data, a_s, a_e, ph)
                with _option(_lib.OPT_FAST_EXACT_SCORES, 1):
                    b = sig_proc.fingerprint_batch(data, a_s, a_e, ph)
            assert np.array_equal(a.status, b.status), name
            assert _same(a.fpt, b.fpt) and _same(a.dwell, b.dwell) and _same(a.stats, b.stats), name
            fpt, dwell, stats, status = orc.fingerprint_batch(data, a_s, a_e, orc.SegParams(barcode_num_events=K))
            assert np.array_equal(a.status, status), name
            good = status == 0
            assert good.sum() > 400, name
            assert _same(a.fpt[good], fpt[good]) and _same(a.dwell[good], dwell[good]), name
            assert _same(a.stats[good], stats[good]), name


def test_adc_quantised_signals_match_oracle():
    """Real pA signals are (int16 + offset) * scale: a few hundred distinct values, hence many exact
    duplicates (radix bins that never shrink), equal scores, plateaus and ties at the top-E cut.  The
    engine must still agree with the oracle bit for bit (the oracle's stable tie rule)."""
    spec = synth.SynthSpec(n_barcodes=10)
    mb, a_s, a_e, _ = synth.generate_minibatch(spec, 31_000, 600, 9000)
    scale = np.float32(0.1755)
    q = np.round(mb / scale).astype(np.float32) * scale          # NaN tail stays NaN
    coarse = np.round(mb / np.float32(2.0)).astype(np.float32) * np.float32(2.0)
    for data in (q, coarse):
        for K in (25, 110):
            with _chain():  # approximate keys: every tie is a doubt (tile fall-back / exact-scores retry)
                fc = sig_proc.fingerprint_batch(data, a_s, a_e, sig_proc.SegParams(barcode_num_events=K))
            fb = sig_proc.fingerprint_batch(data, a_s, a_e, sig_proc.SegParams(barcode_num_events=K))
            assert np.array_equal(fc.status, fb.status) and _same(fc.fpt, fb.fpt) and _same(fc.dwell, fb.dwell)
            assert _same(fc.stats, fb.stats)
            fpt, dwell, stats, status = orc.fingerprint_batch(data, a_s, a_e, orc.SegParams(barcode_num_events=K))
            assert np.array_equal(fb.status, status)
            good = status == 0
            assert good.sum() > 0
            assert _same(fb.dwell[good], dwell[good])
            assert _same(fb.fpt[good], fpt[good])
            assert _same(fb.stats[good], stats[good])


def test_fingerprint_long_rows_and_capacity():
    """adapter windows up to the exact kernel's LDS capacity (11 200: 1024-thread / one-workgroup-per-CU carve-up),
    beyond it up to WDX_MAX_ADAPTER_SAMPLES = 16 384 (score curve in the context's HBM block), and past that"""
    rng = np.random.default_rng(12)
    n, stride = 9, 17000
    mb = np.full((n, stride), np.nan, dtype=np.float32)
    lens = [11200, 11000, 9000, 11201, 14000, 7000, 16384, 16385, 17000]
    for i, ln in enumerate(lens):
        mb[i, :ln] = (np.repeat(rng.normal(80, 15, ln // 40 + 1), 40)[:ln] + rng.normal(0, 2, ln)).astype(np.float32)
    a_s = np.zeros(n, dtype=np.int32)
    a_e = np.array(lens, dtype=np.int32)
    ph, po = sig_proc.SegParams(padding=0), orc.SegParams(padding=0)
    fb = sig_proc.fingerprint_batch(mb, a_s, a_e, ph)
    fpt, dwell, stats, status = orc.fingerprint_batch(mb, a_s, a_e, po)
    for i, ln in enumerate(lens):
        if ln > 16384:
            assert fb.status[i] == 5   # documented engine limit (WDX_MAX_ADAPTER_SAMPLES); the reference's configs stop at 15 200
            assert np.isnan(fb.fpt[i]).all() and (fb.dwell[i] == 0).all()
        else:
            assert fb.status[i] == status[i] == 0
            assert _same(fb.fpt[i], fpt[i]) and _same(fb.dwell[i], dwell[i]) and _same(fb.stats[i], stats[i])


def test_long_window_golden_vectors(golden_dir):
    """g4b: the reference's own detect_results_to_fpt on windows of 9 000 .. 15 200 samples, three shipped
    parameter triples (tests/golden/make_golden_long.py) -- one read per call, all in one batch, and on the
    exact general kernel."""
    g = np.load(os.path.join(golden_dir, "g4b_long_windows.npz"))
    n = int(g["n"])
    for exact in (False, True):
        with (_exact_path() if exact else contextlib.nullcontext()):
            for k in range(n):
                tag = str(g[f"tag_{k}"])
                a_start, a_end, ok = (int(v) for v in g[f"args_{k}"])
                p_hip, _ = _params_from(g, k)
                fb = sig_proc.fingerprint_batch(g[f"row_{k}"].reshape(1, -1), [a_start], [a_end], p_hip, success=[ok])
                st_ref = int(g[f"status_{k}"])
                assert int(fb.status[0]) == st_ref, f"case {k} ({tag}): status {fb.status[0]} != {st_ref}"
                if st_ref == 0:
                    assert _same(fb.fpt[0], g[f"fpt_{k}"]), f"case {k} ({tag}) fpt"
                    assert _same(fb.dwell[0], g[f"dwell_{k}"]), f"case {k} ({tag}) dwell"
                    assert _same(fb.stats[0], g[f"stats_{k}"]), f"case {k} ({tag}) stats"
    # the nine plain cases share their parameters per triple: one mixed-length batch per triple
    for triple in ("rna004", "rna002", "trna"):
        ks = [k for k in range(n) if str(g[f"tag_{k}"]) in {f"{triple}_{m}" for m in (11201, 13000, 15200)}]
        assert len(ks) == 3
        stride = max(g[f"row_{k}"].size for k in ks)
        mb = np.full((len(ks), stride), np.nan, dtype=np.float32)
        for i, k in enumerate(ks):
            mb[i, :g[f"row_{k}"].size] = g[f"row_{k}"]
        a_s = np.array([int(g[f"args_{k}"][0]) for k in ks], dtype=np.int32)
        a_e = np.array([int(g[f"args_{k}"][1]) for k in ks], dtype=np.int32)
        fb = sig_proc.fingerprint_batch(mb, a_s, a_e, _params_from(g, ks[0])[0])
        for i, k in enumerate(ks):
            assert fb.status[i] == 0 and _same(fb.fpt[i], g[f"fpt_{k}"]) and _same(fb.dwell[i], g[f"dwell_{k}"])


@pytest.mark.parametrize("triple", [(110, 15, 30), (120, 9, 18), (110, 9, 30), (60, 17, 30), (110, 2, 18),
                                    # round 5: the other multiples of six (NBT = 2 instantiations like the RNA002 triple's)
                                    (110, 3, 6), (110, 6, 6), (110, 12, 24), (90, 17, 24), (100, 17, 36), (110, 8, 36),
                                    (110, 12, 12), (120, 15, 18)])      # the shipped widths with a reach beyond 9
def test_fast_kernels_of_the_other_shipped_triples(triple):
    """The fast kernels' instantiations for window widths 18 and 30 (tRNA and RNA002 triples; suppression reach up
    to d = 17): synthetic RNA004-like reads, ADC-quantised reads (ties, plateaus -> exact-score tiles, retries) and
    short reads whose parameters shrink (sig_proc.py:526-533), through the launch chain (approximate keys) and
    through the small-batch form (exact scores), against the oracle."""
    E, d, w = triple
    spec = synth.SynthSpec(n_barcodes=10)
    n = 2600
    mb, a_s, a_e, _ = synth.generate_minibatch(spec, 777, n, 9000)
    rng = np.random.default_rng(E * d + w)
    q = slice(2000, 2300)                                   # quantised block
    mb[q] = np.round(mb[q] / 0.1755) * np.float32(0.1755)
    for i in range(2300, 2400):                              # short windows: effective width / distance shrink
        a_e[i] = a_s[i] + int(rng.integers(300, 3300))
    kw = dict(num_events=E, min_obs_per_base=d, running_stat_width=w, barcode_num_events=25)
    fpt, dwell, stats, status = orc.fingerprint_batch(mb, a_s, a_e, orc.SegParams(**kw))
    assert (status == 0).sum() > 0.9 * n
    fb = sig_proc.fingerprint_batch(mb, a_s, a_e, sig_proc.SegParams(**kw))               # n >= 2048: launch chain
    assert np.array_equal(fb.status, status)
    good = status == 0
    assert _same(fb.fpt[good], fpt[good]) and _same(fb.dwell[good], dwell[good]) and _same(fb.stats[good], stats[good])
    sm = sig_proc.fingerprint_batch(mb[:700], a_s[:700], a_e[:700], sig_proc.SegParams(**kw))   # exact scores from the start
    assert np.array_equal(sm.status, status[:700]) and _same(sm.fpt[good[:700]], fpt[:700][good[:700]])


@pytest.mark.parametrize("w", range(6, 37))
def test_every_window_width_and_reach_against_the_oracle(w):
    """running_stat_width 6 .. 36 x min_obs_per_base 2 .. 17 (any value `--export segmentation.…` can set,
    config/sig_proc.py:16-70): multiples of six run on the fast kernels' approximate keys (reach <= 17), every other
    width -- odd ones included -- on their exact-scores pass (round 6: the partner window's statistics come from slot
    (rr + W) % 6 of the lane (rr + W) / 6 further) -- the same fingerprints as the oracle every way, through the launch
    chain of large batches."""
    spec = synth.SynthSpec(n_barcodes=10)
    n = 96
    mb, a_s, a_e, _ = synth.generate_minibatch(spec, 4242 + w, n, 9000)
    mb[80:] = np.round(mb[80:] / 0.1755) * np.float32(0.1755)         # ADC-quantised rows: ties and plateaus
    for d in (2, 5, 9, 13, 17):
        kw = dict(num_events=110, min_obs_per_base=d, running_stat_width=w, barcode_num_events=25)
        fpt, dwell, stats, status = orc.fingerprint_batch(mb, a_s, a_e, orc.SegParams(**kw))
        with _chain():
            fb = sig_proc.fingerprint_batch(mb, a_s, a_e, sig_proc.SegParams(**kw))
        assert np.array_equal(fb.status, status), (w, d, np.flatnonzero(fb.status != status))
        good = status == 0
        assert _same(fb.fpt[good], fpt[good]) and _same(fb.dwell[good], dwell[good]) and _same(fb.stats[good], stats[good]), (w, d)
        if d <= 9:
            assert good.sum() > 0.8 * n, (w, d, int(good.sum()))


@pytest.mark.parametrize("max_slice", [0, 700])     # 700: every launch of the chain cut into slices (block_base != 0)
@pytest.mark.parametrize("triple", [(110, 6, 12), (110, 15, 30), (120, 9, 18), (110, 12, 24), (110, 17, 36), (110, 4, 6)])
def test_long_windows_in_large_batches_vs_oracle(triple, max_slice):
    """3 000 reads whose windows straddle every capacity edge (5 120 / 6 144 / 8 192 / 11 200 / 16 384) through the
    launch chain: windows of 8 193 .. 16 384 samples take the streaming fast kernel, and what it declines beyond 11 200
    samples reaches fingerprint_big_kernel via the slow list."""
    E, d, w = triple
    rng = np.random.default_rng(E + d + w)
    n = 3000
    lens = rng.integers(2500, 9000, n)
    edge = [5119, 5120, 5121, 6144, 6145, 8192, 8193, 11199, 11200, 11201, 11264, 13000, 15200, 16383, 16384, 16385]
    lens[:len(edge)] = edge
    lens[100:140] = rng.integers(11201, 16385, 40)
    lens[140:440] = rng.integers(8193, 16385, 300)     # the streaming fast kernel's range (fingerprint_fast_stream_kernel)
    stride = int(lens.max())
    mb = np.full((n, stride), np.nan, dtype=np.float32)
    for i, ln in enumerate(lens):
        dw = max(12, ln // 135)
        mb[i, :ln] = (np.repeat(rng.normal(80, 15, ln // dw + 1), dw)[:ln] + rng.normal(0, 2, ln)).astype(np.float32)
    # long windows with flicker spikes below zero (the one-wave clip kernel clamps them away, or leaves the window to the
    # workgroup kernel behind it), and a few with mostly negative samples (always left to it)
    for i in range(140, 200):
        mb[i, rng.integers(0, lens[i], 1 + i % 7)] = -rng.uniform(1, 60, 1 + i % 7).astype(np.float32)
    for i in range(200, 206):
        mb[i, :lens[i]] -= np.float32(85.0)
    a_s = np.zeros(n, dtype=np.int32)
    a_e = lens.astype(np.int32)
    kw = dict(padding=0, num_events=E, min_obs_per_base=d, running_stat_width=w, barcode_num_events=25)
    with _option(_lib.OPT_MAX_LAUNCH_SLICE, max_slice):
        fb = sig_proc.fingerprint_batch(mb, a_s, a_e, sig_proc.SegParams(**kw))
        if w == 30:      # the same with the long windows' clip bounds from the workgroup kernel alone
            with _option(_lib.OPT_NO_WAVE_CLIP_LONG, 1):
                fb2 = sig_proc.fingerprint_batch(mb, a_s, a_e, sig_proc.SegParams(**kw))
            assert np.array_equal(fb2.status, fb.status) and _same(fb2.fpt, fb.fpt) and _same(fb2.dwell, fb.dwell)
    fpt, dwell, stats, status = orc.fingerprint_batch(mb, a_s, a_e, orc.SegParams(**kw))
    big = lens > 16384
    assert (fb.status[big] == 5).all() and big.sum() == 1
    assert np.array_equal(fb.status[~big], status[~big])
    good = (status == 0) & ~big
    assert good.sum() > 0.95 * n
    assert _same(fb.fpt[good], fpt[good]) and _same(fb.dwell[good], dwell[good]) and _same(fb.stats[good], stats[good])


def test_fingerprint_empty_and_ragged_inputs():
    """zero reads, zero-width rows, rows shorter than the adapter window, ragged packed batch"""
    p = sig_proc.SegParams()
    fb = sig_proc.fingerprint_batch(np.zeros((0, 100), np.float32), [], [], p)
    assert fb.fpt.shape == (0, 25) and fb.status.shape == (0,)
    fb = sig_proc.fingerprint_batch(np.zeros((3, 0), np.float32), [0, 0, 0], [0, 0, 0], p)
    assert fb.status.tolist() == [5, 5, 5]            # empty window -> "unknown" like the reference (tiny_0)
    spec = synth.SynthSpec(n_barcodes=4)
    sig, off, a_s, a_e, _ = synth.generate_packed(spec, 900, 40)
    # ragged minibatch: every row truncated at a different length (NaN tail), some inside the adapter
    stride = 5200
    mb = np.full((40, stride), np.nan, dtype=np.float32)
    for i in range(40):
        row = sig[off[i]:off[i + 1]][: stride - 37 * i]
        mb[i, : row.size] = row
    fb = sig_proc.fingerprint_batch(mb, a_s, a_e, p)
    fpt, dwell, stats, status = orc.fingerprint_batch(mb, a_s, a_e, orc.SegParams())
    assert np.array_equal(fb.status, status) and len(set(status.tolist())) >= 2
    good = status == 0
    assert _same(fb.fpt[good], fpt[good]) and _same(fb.dwell[good], dwell[good]) and _same(fb.stats[good], stats[good])


def test_fused_host_call_matches_separate_calls():
    spec = synth.SynthSpec(n_barcodes=6)
    rng = np.random.default_rng(4)
    for n, K, nY in ((1, 110, 6), (7, 25, 1368), (300, 110, 6)):
        mb, a_s, a_e, _ = synth.generate_minibatch(spec, 60_000, n, 8000)
        ok = np.ones(n, dtype=np.uint8)
        if n > 3:
            ok[2] = 0
            a_e[3] = a_s[3] - 100
        Y = rng.normal(size=(nY, K))
        p = sig_proc.SegParams(barcode_num_events=K)
        sig_proc.set_references(Y, 15, 0.1)
        res = sig_proc.demux_batch(mb, a_s, a_e, p, success=ok, want_fpt=True, n_refs=nY)
        fb = sig_proc.fingerprint_batch(mb, a_s, a_e, p, success=ok)
        assert np.array_equal(res.status, fb.status) and _same(res.fpt, fb.fpt)
        good = fb.status == 0
        D, am = pdist.nearest_reference(fb.fpt[good], Y, 15, 0.1)
        assert _same(res.dist[good], D) and np.isnan(res.dist[~good]).all()
        assert np.array_equal(res.call[good], am) and (res.call[~good] == -1).all()
        assert _same(D, orc.dtw_matrix(fb.fpt[good], Y, 15, 0.1))
    with pytest.raises(ValueError):
        sig_proc.demux_batch(mb, a_s, a_e, sig_proc.SegParams(barcode_num_events=24), n_refs=nY)
    # another user of the context replaces the resident references (same shape, other content): the tick
    # loop's references come back by themselves
    before = sig_proc.demux_batch(mb, a_s, a_e, p, success=ok, n_refs=nY)
    pdist.distance_matrix_to(fb.fpt[good][:3], Y[::-1].copy(), window=15, penalty=0.1, n_jobs=1)
    after = sig_proc.demux_batch(mb, a_s, a_e, p, success=ok, n_refs=nY)
    assert _same(before.dist, after.dist) and np.array_equal(before.call, after.call)


def test_detect_results_to_fpt_shim():
    from types import SimpleNamespace as NS

    spec = synth.SynthSpec(n_barcodes=4)
    mb, a_s, a_e, _ = synth.generate_minibatch(spec, 0, 6, 8000)
    spc = NS(sig_extract=NS(padding=100, normalization="none"), core=NS(sig_norm_outlier_thresh=5.0),
             segmentation=NS(num_events=110, min_obs_per_base=6, running_stat_width=12, accept_less_cpts=False,
                             consensus_refinement=False, normalization="mean", barcode_num_events=25))
    drs = [sig_proc.DetectResults(True, "", int(a_s[i]), int(a_e[i])) for i in range(6)]
    drs[2] = sig_proc.DetectResults(False, "no adapter found", None, None)
    drs[3] = sig_proc.DetectResults(True, "", 100, 0)   # 100-sample window -> find_peaks(distance=0) raises
    res = sig_proc.detect_results_to_fpt_batch(mb, spc, drs, read_ids=[f"r{i}" for i in range(6)])
    assert [r.success for r in res] == [True, True, False, False, True, True]
    assert res[2].fail_reason == "no adapter found" and res[2].barcode_fpt.size == 0
    assert res[3].fail_reason == "unknown" and res[3].detect_results is None
    o = orc.fingerprint_one(mb[0], a_s[0], a_e[0], orc.SegParams())
    assert _same(res[0].barcode_fpt, o["fpt"]) and res[0].read_id == "r0"
    assert res[0].adapter_event_mean == o["stats"][2]
    one = sig_proc.detect_results_to_fpt(mb[1], spc, drs[1])
    assert _same(one.barcode_fpt, res[1].barcode_fpt)
    spc.sig_extract.normalization = "mean"    # A2 "mean": float32 np.mean / np.std on the clipped signal
    resm = sig_proc.detect_results_to_fpt_batch(mb, spc, drs)
    om = orc.fingerprint_one(mb[0], a_s[0], a_e[0], orc.SegParams(sig_norm="mean"))
    assert resm[0].success and _same(resm[0].barcode_fpt, om["fpt"]) and not _same(om["fpt"], o["fpt"])
    with pytest.raises(ValueError):
        spc.sig_extract.normalization = "bogus"
        sig_proc.detect_results_to_fpt_batch(mb, spc, drs)


def test_signal_normalisation_mean_matches_oracle():
    """sig_extract.normalization = "mean" (sig_proc.py:99-111 on the float32 adapter signal): NumPy's float32
    pairwise sums incl. the 8192-element chunking of add.reduce (windows longer than 8192 samples), and the
    nanmean / nanstd branch for windows that hold NaNs.  The golden G4 cases pin the oracle to the reference;
    here the engine is compared with the oracle on many more shapes."""
    rng = np.random.default_rng(12)
    lens = [700, 1320, 2048, 4097, 4801, 6000, 8191, 8192, 8193, 8200, 9973, 11000, 11200, 5000, 5000, 300]
    n, stride = len(lens), 11200
    mb = np.full((n, stride), np.nan, dtype=np.float32)
    for i, ln in enumerate(lens):
        ev = rng.integers(20, 60)
        lvl = np.repeat(rng.normal(80, 15, ln // ev + 1), ev)[:ln]
        mb[i, :ln] = (lvl + rng.normal(0, 2, ln)).astype(np.float32)
    mb[13, 1000:1003] = np.nan          # NaNs inside the window -> nanmean / nanstd
    mb[14, 4990:5000] = np.nan
    a_s = np.zeros(n, dtype=np.int32)
    a_e = np.array(lens, dtype=np.int32)
    for K, thr in ((25, 5.0), (110, 3.0)):
        kw = dict(padding=0, sig_norm="mean", barcode_num_events=K, outlier_thresh=thr)
        fb = sig_proc.fingerprint_batch(mb, a_s, a_e, sig_proc.SegParams(**kw))
        fpt, dwell, stats, status = orc.fingerprint_batch(mb, a_s, a_e, orc.SegParams(**kw))
        assert np.array_equal(fb.status, status)
        good = status == 0
        assert good.sum() >= 12
        assert _same(fb.fpt[good], fpt[good]) and _same(fb.dwell[good], dwell[good]) and _same(fb.stats[good], stats[good])


def test_clip_bound_rules_float32_and_float64():
    """The clip bounds under NumPy >= 2 (float32) and under NumPy 1.x / np.float64 thresholds (float64, rounded
    once): both rules through the engine, each against the oracle; with a threshold like 2.7 they differ."""
    spec = synth.SynthSpec(n_barcodes=10)
    mb, a_s, a_e, _ = synth.generate_minibatch(spec, 31_000, 300, 9000)
    differ = 0
    for thr in (2.7, 3.3, 5.0):
        out = {}
        for rule in ("float32", "float64"):
            fb = sig_proc.fingerprint_batch(mb, a_s, a_e, sig_proc.SegParams(barcode_num_events=110, outlier_thresh=thr,
                                                                             clip_bounds=rule))
            fpt, dwell, stats, status = orc.fingerprint_batch(
                mb, a_s, a_e, orc.SegParams(barcode_num_events=110, outlier_thresh=thr, clip_bounds_f64=rule == "float64"))
            good = status == 0
            assert np.array_equal(fb.status, status) and good.sum() > 290
            assert _same(fb.fpt[good], fpt[good]) and _same(fb.dwell[good], dwell[good])
            out[rule] = fb.fpt
        differ += int(not _same(out["float32"], out["float64"]))
    assert differ >= 1
    # "auto" follows the NumPy of this process; an np.float64 threshold always means float64 bounds
    auto = sig_proc.SegParams(outlier_thresh=2.7).to_c()
    assert auto.clip_bounds_f64 == int(int(np.__version__.split(".")[0]) < 2)
    assert sig_proc.SegParams(outlier_thresh=np.float64(2.7)).to_c().clip_bounds_f64 == 1


def test_fast_score_sequences_match_the_compilers_sqrt_and_division():
    """The fast kernel's t-score uses the sqrt / quotient iterations without their range scaling
    (wdx_fingerprint_fast.inc: fast_sqrt_mid, fast_div_mid).  Bit-identity with the general float64 sqrt() and '/'
    over the documented value range is asserted here, so a toolchain that changes its expansions fails loudly."""
    import ctypes as C

    import torch

    rng = np.random.default_rng(99)
    n = 1 << 20
    # variance sums: squares of float64 differences of float32 values -> [2^-402, 2^261]; mean differences
    # {0} U [2^-201, 2^129]; plus realistic pA-scale values and exact powers of two / neighbours of them
    vs = np.concatenate([
        2.0 ** rng.uniform(-402, 261, n // 2), rng.uniform(1e-3, 1e5, n // 4), 2.0 ** rng.integers(-402, 262, n // 8).astype(np.float64),
        np.nextafter(2.0 ** rng.integers(-400, 260, n // 8).astype(np.float64), np.inf)])
    dm = np.concatenate([
        2.0 ** rng.uniform(-201, 129, n // 2), rng.uniform(0, 200, n // 4), np.zeros(n // 8),
        2.0 ** rng.integers(-201, 130, n // 8).astype(np.float64)])
    rng.shuffle(dm)
    t_dm, t_vs = torch.from_numpy(dm).cuda(), torch.from_numpy(vs).cuda()
    fast, ref = torch.empty_like(t_dm), torch.empty_like(t_dm)
    ctx = _lib.default_context()
    _lib.check(_lib.load().wdx_selftest_score_dev(ctx.handle, C.c_void_p(t_dm.data_ptr()), C.c_void_p(t_vs.data_ptr()), dm.size,
                                                  C.c_void_p(fast.data_ptr()), C.c_void_p(ref.data_ptr()),
                                                  C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    f, r = fast.cpu().numpy(), ref.cpu().numpy()
    assert np.array_equal(f.view(np.uint64), r.view(np.uint64)), int((f.view(np.uint64) != r.view(np.uint64)).sum())
    assert np.array_equal(r, dm / np.sqrt(vs))   # and the device's general expansions are the correctly rounded ones


def test_dtaidistance_cross_check_if_available():
    """dtaidistance is not in the reference tree (the DTW seam is pinned through the shipped models'
    KKT conditions instead, tests/test_gpu_kkt.py).  If the
    box this runs on happens to have the library, compare the genuine call of parallel_distances.py:59-67 with the
    engine on >= 1e4 pairs of both shapes; otherwise skip (never fail for its absence)."""
    dtai = pytest.importorskip("dtaidistance")
    from dtaidistance import dtw

    rng = np.random.default_rng(21)
    for nX, nY, L in ((1200, 10, 110), (16, 851, 25)):
        X, Y = rng.normal(size=(nX, L)), rng.normal(size=(nY, L))
        stack = np.vstack([X, Y])
        ref = dtw.distance_matrix(stack, block=((0, nX), (nX, nX + nY)), parallel=False, use_c=True, only_triu=True,
                                  window=15, penalty=0.1)[:nX, nX:].astype(np.float32)
        mine = pdist.distance_matrix_to(X, Y, window=15, penalty=0.1, n_jobs=1)
        assert mine.size >= 10_000
        assert np.array_equal(mine.argmin(1), ref.argmin(1)), dtai.__version__
        np.testing.assert_allclose(mine, ref, rtol=RTOL, atol=0)


# ------------------------------------------------------------------------ fused device pipeline ----

def test_device_synth_matches_numpy_and_fused_pipeline():
    import torch

    from warpdemux_amd.engine import DemuxEngine

    spec = synth.SynthSpec(n_barcodes=10)
    K = 110
    # references: fingerprints of low-noise template reads through the oracle
    clean = synth.SynthSpec(n_barcodes=10, noise_sigma=0.25, spikes=False)
    refs = np.zeros((10, K))
    rid, found = 0, set()
    while len(found) < 10:
        s, b = synth.generate_read(clean, rid)
        if b not in found:
            o = orc.fingerprint_one(s, synth.PAD, s.size - synth.PAD, orc.SegParams(barcode_num_events=K))
            assert o["status"] == 0
            refs[b] = o["fpt"]
            found.add(b)
        rid += 1
    eng = DemuxEngine(refs, 15, 0.1, sig_proc.SegParams(barcode_num_events=K))
    n = 3000
    sig, off, a_s, a_e, bc, max_len = eng.synth_packed(spec, 1_000_000, n)
    hs, ho, hs_s, hs_e, hb = synth.generate_packed(spec, 1_000_000, 64)
    torch.cuda.synchronize()
    assert np.array_equal(off[:65].cpu().numpy(), ho)
    assert np.array_equal(sig[: ho[-1]].cpu().numpy(), hs)   # device generator == NumPy generator, bitwise
    assert np.array_equal(bc[:64].cpu().numpy(), hb)
    res = eng.demux(sig, a_s, a_e, offsets=off, max_len=max_len + 0, want_fpt=True)
    torch.cuda.synchronize()
    sig_h, off_h = sig.cpu().numpy(), off.cpu().numpy()
    fpt, dwell, stats, status = orc.fingerprint_packed(sig_h, off_h, a_s.cpu().numpy(), a_e.cpu().numpy(),
                                                       orc.SegParams(barcode_num_events=K))
    assert np.array_equal(res.status.cpu().numpy(), status)
    good = status == 0
    assert good.mean() > 0.99
    assert _same(res.fpt.cpu().numpy()[good], fpt[good])
    D = orc.dtw_matrix(fpt[good], refs, 15, 0.1)
    _check_dist(res.dist.cpu().numpy()[good], D)
    call = res.call.cpu().numpy()
    assert np.array_equal(call[good], np.argmin(D, axis=1))
    assert (call[~good] == -1).all()
    counts = res.counts.cpu().numpy()
    exp = np.bincount(np.where(good, call, 10), minlength=11)
    assert np.array_equal(counts, exp) and counts.sum() == n
    acc = (call[good] == bc.cpu().numpy()[good]).mean()
    assert acc > 0.7, acc   # sanity only: nearest-template accuracy on the synthetic barcodes
    # second call accumulates into the same histogram; with timing on, the main fast kernel's event pair (the
    # roofline's denominator) lies inside the one around the whole fingerprint chain (n >= 2048: main kernel, list
    # kernels, exact-scores retry, exact general kernel)
    eng.kernel_time_reset()
    eng.kernel_timing(True)
    res2 = eng.demux(sig, a_s, a_e, offsets=off, max_len=max_len, counts=res.counts)
    torch.cuda.synchronize()
    eng.kernel_timing(False)
    assert np.array_equal(res2.counts.cpu().numpy(), 2 * exp)
    chain_ms, chain_n = eng.kernel_time(_lib.K_FINGERPRINT)
    main_ms, main_n = eng.kernel_time(_lib.K_FINGERPRINT_MAIN)
    assert chain_n == 1 and main_n == 1 and 0.0 < main_ms <= chain_ms
    eng.close()


# ----------------------------------------------------------------------------------------------
# N1: DTW_SVM.predict -- DTW -> exp kernel -> libsvm predict_probability -> process_probs
# ----------------------------------------------------------------------------------------------
def _svm_case(n_classes, n_train, n_query, seed, thresholds=False):
    from test_oracle_svm import make_model
    from warpdemux_amd.models import DTW_SVM

    svc, Xtr, centers, rng = make_model(n_classes, n_train, seed=seed)
    yq = rng.integers(0, n_classes, n_query)
    Xq = centers[yq] + 0.9 * rng.normal(size=(n_query, Xtr.shape[1]))
    label_mapper = {i: 3 * i + 1 for i in range(n_classes)}
    thr = rng.uniform(0.05, 0.6, n_classes) if thresholds else None
    n_support, support, dual_coef, rho, probA, probB, k = orc.svm_params(svc)
    m = DTW_SVM(Xtr, n_support, support, dual_coef, rho, probA, probB, label_mapper, thr, window=15, penalty=0.1,
                gamma=1.0, pwr_dist=1, block_size=2000)
    return svc, Xtr, Xq, label_mapper, thr, m


def _reference_tail(svc, Xtr, Xq, label_mapper, thr):
    """models/dtw_svm.py:85-98 with the oracle's DTW and scikit-learn's own predict_proba."""
    Kq = np.exp(-1.0 * np.power(orc.dtw_matrix(Xq, Xtr, 15, 0.1), 1))
    prob = svc.predict_proba(Kq)
    idx = np.argmax(prob, axis=1)
    pred = np.array([label_mapper[i] for i in idx])
    srt = np.sort(prob, axis=1)
    conf = srt[:, -1] - srt[:, -2]
    if thr is not None:
        pred[conf < thr[idx]] = -1
    return pred, prob, conf


@pytest.mark.gpu
@pytest.mark.parametrize("n_classes,n_train,thresholds", [(2, 80, False), (3, 150, True), (5, 300, True),
                                                          (11, 500, False), (12, 700, True),
                                                          (16, 800, True)])    # largest supported model
def test_dtw_svm_predict_matches_sklearn(n_classes, n_train, thresholds):
    sklearn = pytest.importorskip("sklearn")
    svc, Xtr, Xq, label_mapper, thr, m = _svm_case(n_classes, n_train, 333, seed=n_classes, thresholds=thresholds)
    pred_ref, prob_ref, conf_ref = _reference_tail(svc, Xtr, Xq, label_mapper, thr)
    pred, prob = m.predict(Xq)
    assert prob.shape == prob_ref.shape and pred.shape == pred_ref.shape
    # float32 exp on the device may differ from NumPy's by one ulp in the kernel value: 1e-5 on probabilities
    np.testing.assert_allclose(prob, prob_ref, rtol=0, atol=1e-5)
    np.testing.assert_allclose(prob.sum(axis=1), 1.0, atol=1e-12)
    # labels must agree wherever the reference's own decision is not within the tolerance of a tie / threshold
    srt = np.sort(prob_ref, axis=1)
    safe = (srt[:, -1] - srt[:, -2]) > 1e-4
    if thr is not None:
        safe &= np.abs(conf_ref - thr[np.argmax(prob_ref, axis=1)]) > 1e-4
    assert safe.mean() > 0.95
    assert np.array_equal(pred[safe], pred_ref[safe])
    if thr is not None:
        assert (pred_ref == -1).any() and (pred == -1).any()
    df = m.predict(Xq, return_df=True)
    assert list(df.columns[:2]) == ["predicted_barcode", "confidence_score"]
    assert list(df.columns[2:]) == [f"p{label_mapper[i]:02d}" for i in range(n_classes)]
    np.testing.assert_allclose(df["confidence_score"].to_numpy(), conf_ref, atol=1.5e-3)
    assert np.array_equal(df["predicted_barcode"].to_numpy(), pred)


@pytest.mark.gpu
def test_dtw_svm_predict_matches_oracle_on_device_kernel_values():
    """With the kernel matrix the device itself produced the oracle's libsvm restatement must agree to
    double rounding: isolates the classifier tail from the float32 exp."""
    pytest.importorskip("sklearn")
    svc, Xtr, Xq, label_mapper, thr, m = _svm_case(5, 300, 200, seed=9)
    D = pdist.distance_matrix_to(Xq, Xtr, window=15, penalty=0.1, n_jobs=1)
    assert np.array_equal(D, orc.dtw_matrix(Xq, Xtr, 15, 0.1))
    pred, prob = m.predict(Xq)
    pr = svc.predict_proba(np.exp(-1.0 * D))
    np.testing.assert_allclose(prob, pr, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("block_rows", [0, 2048])
def test_demux_svm_dev_whole_path_matches_the_chained_calls_and_sklearn(block_rows):
    """wdx_demux_svm_dev (raw rows -> fingerprint -> DTW vs the training set -> SVM tail, one device-resident call, the
    distance matrix in row blocks) against (i) the same engine's three separate calls, bit for bit, and (ii) the oracle's
    fingerprints + DTW with scikit-learn's own predict_proba / process_probs, 1e-5 on probabilities."""
    pytest.importorskip("sklearn")
    import torch

    from warpdemux_amd.engine import DemuxEngine

    svc, Xtr, Xq, label_mapper, thr, m = _svm_case(5, 300, 8, seed=21, thresholds=True)
    K = Xtr.shape[1]
    spec = synth.SynthSpec(n_barcodes=4)
    n = 5000
    params = sig_proc.SegParams(barcode_num_events=K)
    eng = DemuxEngine(Xtr, 15, 0.1, params)
    eng.set_svm(m)
    sig, off, a_s, a_e, bc, max_len = eng.synth_packed(spec, 100, n)
    a_e = a_e.clone()
    a_e[::97] = a_s[::97] + 40                      # a few reads that fail (window too short for 110 events)
    prob, pred, conf, status, dist, fpt = eng.demux_svm(sig, a_s, a_e, offsets=off, max_len=max_len, want_dist=True,
                                                        want_fpt=True, block_rows=block_rows)
    torch.cuda.synchronize()
    f2, _, _, st2 = eng.fingerprint(sig, a_s, a_e, offsets=off, max_len=max_len)
    d2, _ = eng.dtw(f2, want_argmin=False)
    p2, q2, c2 = eng.svm_predict(d2)
    torch.cuda.synchronize()
    status, st2 = status.cpu().numpy(), st2.cpu().numpy()
    ok = status == 0
    assert np.array_equal(status, st2) and (~ok).sum() >= 50 and ok.sum() > 0.95 * n
    assert _same(fpt.cpu().numpy()[ok], f2.cpu().numpy()[ok]) and _same(dist.cpu().numpy()[ok], d2.cpu().numpy()[ok])
    assert _same(prob.cpu().numpy()[ok], p2.cpu().numpy()[ok]) and np.array_equal(pred.cpu().numpy()[ok], q2.cpu().numpy()[ok])
    assert _same(conf.cpu().numpy()[ok], c2.cpu().numpy()[ok])
    assert (pred.cpu().numpy()[~ok] == -1).all() and np.isnan(prob.cpu().numpy()[~ok]).all() and np.isnan(conf.cpu().numpy()[~ok]).all()
    # without the distance output the shipped shape (25 points, window 15) takes the FUSED form: decision sums in the DTW
    # kernel's epilogue, no distance matrix -- the same model, another summation order (1e-12 on probabilities)
    pb, qb, cb, sb, _, _ = eng.demux_svm(sig, a_s, a_e, offsets=off, max_len=max_len, block_rows=block_rows)
    torch.cuda.synchronize()
    pbn, probn = pb.cpu().numpy(), prob.cpu().numpy()
    assert np.array_equal(sb.cpu().numpy(), status) and np.isnan(pbn[~ok]).all() and (qb.cpu().numpy()[~ok] == -1).all()
    np.testing.assert_allclose(pbn[ok], probn[ok], rtol=0, atol=1e-12)
    np.testing.assert_allclose(cb.cpu().numpy()[ok], conf.cpu().numpy()[ok], rtol=0, atol=1e-12)
    srtf = np.sort(probn[ok], axis=1)
    far = ((srtf[:, -1] - srtf[:, -2]) > 1e-9) & (np.abs(conf.cpu().numpy()[ok] - thr[np.argmax(probn[ok], axis=1)]) > 1e-9)
    assert far.mean() > 0.99 and np.array_equal(qb.cpu().numpy()[ok][far], pred.cpu().numpy()[ok][far])
    # the unfused form on request (diagnostic option): the chained calls' bits again
    eng.ctx.set_option(_lib.OPT_NO_SHORT_DTW, 1)
    pu, qu, cu, su, _, _ = eng.demux_svm(sig, a_s, a_e, offsets=off, max_len=max_len, block_rows=block_rows)
    torch.cuda.synchronize()
    eng.ctx.set_option(_lib.OPT_NO_SHORT_DTW, 0)
    np.testing.assert_allclose(pu.cpu().numpy()[ok], probn[ok], rtol=0, atol=1e-12)
    # against the oracle + scikit-learn on a sample
    ns = 600
    o = off[: ns + 1].cpu().numpy()
    ofp, _, _, ost = orc.fingerprint_packed(sig[: int(o[-1])].cpu().numpy(), o, a_s[:ns].cpu().numpy(), a_e[:ns].cpu().numpy(),
                                            orc.SegParams(barcode_num_events=K))
    assert np.array_equal(ost, status[:ns])
    oko = ost == 0
    pred_ref, prob_ref, conf_ref = _reference_tail(svc, Xtr, ofp[oko], label_mapper, thr)
    np.testing.assert_allclose(prob.cpu().numpy()[:ns][oko], prob_ref, rtol=0, atol=1e-5)
    np.testing.assert_allclose(pbn[:ns][oko], prob_ref, rtol=0, atol=1e-5)          # the fused form too
    srt = np.sort(prob_ref, axis=1)
    safe = ((srt[:, -1] - srt[:, -2]) > 1e-4) & (np.abs(conf_ref - thr[np.argmax(prob_ref, axis=1)]) > 1e-4)
    assert np.array_equal(pred.cpu().numpy()[:ns][oko][safe], pred_ref[safe])
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n_classes,n_train", [(2, 80), (3, 150), (11, 500), (16, 800)])
def test_demux_svm_dev_fused_form_over_class_counts(n_classes, n_train):
    """The fused form of wdx_demux_svm_dev (decision sums in the DTW kernel's epilogue: two chunks per class, k - 1 sums
    per lane in LDS) from the smallest model (2 classes: one sum) to the largest supported one (16: fifteen), classes
    with few support vectors included: against the row-block form (1e-12 on probabilities: another summation order)
    and against the oracle's fingerprints + scikit-learn (1e-5)."""
    pytest.importorskip("sklearn")
    import torch

    from warpdemux_amd.engine import DemuxEngine

    svc, Xtr, Xq, label_mapper, thr, m = _svm_case(n_classes, n_train, 8, seed=40 + n_classes, thresholds=True)
    K = Xtr.shape[1]
    spec = synth.SynthSpec(n_barcodes=4)
    n = 1500
    eng = DemuxEngine(Xtr, 15, 0.1, sig_proc.SegParams(barcode_num_events=K))
    eng.set_svm(m)
    sig, off, a_s, a_e, bc, max_len = eng.synth_packed(spec, 7000, n)
    pf, qf, cf, sf, _, _ = eng.demux_svm(sig, a_s, a_e, offsets=off, max_len=max_len)                      # fused
    pu, qu, cu, su, du, _ = eng.demux_svm(sig, a_s, a_e, offsets=off, max_len=max_len, want_dist=True)     # row blocks
    torch.cuda.synchronize()
    status = sf.cpu().numpy()
    ok = status == 0
    assert np.array_equal(status, su.cpu().numpy()) and ok.sum() > 0.95 * n
    pfn, pun = pf.cpu().numpy(), pu.cpu().numpy()
    np.testing.assert_allclose(pfn[ok], pun[ok], rtol=0, atol=1e-12)
    np.testing.assert_allclose(pfn[ok].sum(axis=1), 1.0, atol=1e-12)
    np.testing.assert_allclose(cf.cpu().numpy()[ok], cu.cpu().numpy()[ok], rtol=0, atol=1e-12)
    srt = np.sort(pun[ok], axis=1)
    far = ((srt[:, -1] - srt[:, -2]) > 1e-9) & (np.abs(cu.cpu().numpy()[ok] - thr[np.argmax(pun[ok], axis=1)]) > 1e-9)
    assert np.array_equal(qf.cpu().numpy()[ok][far], qu.cpu().numpy()[ok][far])
    ns = 300
    o = off[: ns + 1].cpu().numpy()
    ofp, _, _, ost = orc.fingerprint_packed(sig[: int(o[-1])].cpu().numpy(), o, a_s[:ns].cpu().numpy(), a_e[:ns].cpu().numpy(),
                                            orc.SegParams(barcode_num_events=K))
    oko = ost == 0
    _, prob_ref, _ = _reference_tail(svc, Xtr, ofp[oko], label_mapper, thr)
    np.testing.assert_allclose(pfn[:ns][oko], prob_ref, rtol=0, atol=1e-5)
    eng.close()


@pytest.mark.gpu
def test_dtw_svm_predict_errors_and_single_row():
    pytest.importorskip("sklearn")
    svc, Xtr, Xq, label_mapper, thr, m = _svm_case(3, 150, 8, seed=4)
    p1, q1 = m.predict(Xq[0])
    pa, qa = m.predict(Xq)
    assert p1.shape == (1,) and q1.shape == (1, 3)
    assert p1[0] == pa[0] and np.array_equal(q1[0], qa[0])
    with pytest.raises(ValueError):
        m.predict(Xq[:, :-1])
    m.block_size = None
    with pytest.raises(ValueError):
        m.predict(Xq)             # nproc=-1 without block_size, like the reference
    # engine limit: at most 16 classes (the coupling runs on 16-lane groups)
    from warpdemux_amd.models import DTW_SVM
    k = 17
    big = DTW_SVM(np.zeros((k, 25)), np.ones(k, dtype=np.int32), np.arange(k, dtype=np.int32), np.zeros((k - 1, k)),
                  np.zeros(k * (k - 1) // 2), np.zeros(k * (k - 1) // 2), np.zeros(k * (k - 1) // 2),
                  {i: i for i in range(k)}, None, window=15, penalty=0.1, block_size=100)
    with pytest.raises(ValueError):
        big.predict(np.zeros((2, 25)))
    # the context holds one reference set / one model at a time: interleaved users must not see each other's
    svc2, Xtr2, Xq2, lm2, thr2, m2 = _svm_case(3, 150, 8, seed=5)
    ref1 = m.predict(Xq, nproc=1)
    ref2 = m2.predict(Xq2, nproc=1)
    pdist.distance_matrix_to(Xq, np.flipud(Xtr2), window=15, penalty=0.1, n_jobs=1)   # same shape, other content
    again1, again2 = m.predict(Xq, nproc=1), m2.predict(Xq2, nproc=1)
    assert np.array_equal(ref1[0], again1[0]) and np.array_equal(ref1[1], again1[1])
    assert np.array_equal(ref2[0], again2[0]) and np.array_equal(ref2[1], again2[1])
    assert m.predict(Xq, nproc=1)[0].shape == (8,)


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["WDX4_rna004", "WDX10_rna004", "WDX12_rna002"])
def test_dtw_svm_predict_reference_model_golden(which):
    """g6 / g6b / g6c: outputs of the reference's DTW_SVM.predict on its shipped WDX4 / WDX10 rna004 models and on
    DEPRECATED WDX12_rna002 (3 617 x 25, 13 classes, gamma 1.2)."""
    from test_oracle_svm import load_g6
    from warpdemux_amd.models import DTW_SVM

    g, label_mapper = load_g6(which)
    m = DTW_SVM(g["X_train"], g["n_support"], g["support"], g["dual_coef"], -g["intercept"], g["probA"], g["probB"],
                label_mapper, g["thresholds"], window=int(g["window"]), penalty=float(g["penalty"]),
                gamma=float(g["gamma"]), pwr_dist=int(g["pwr_dist"]), block_size=int(g["block_size"]))
    pred, prob = m.predict(g["Xq"])
    np.testing.assert_allclose(prob, g["y_prob"], rtol=0, atol=1e-5)
    srt = np.sort(g["y_prob"], axis=1)
    conf = srt[:, -1] - srt[:, -2]
    safe = (conf > 1e-4) & (np.abs(conf - g["thresholds"][np.argmax(g["y_prob"], axis=1)]) > 1e-4)
    assert safe.mean() > 0.98
    assert np.array_equal(pred[safe], g["y_pred"][safe])
    df = m.predict(g["Xq"], return_df=True)
    assert list(df.columns) == list(g["df_cols"])
    np.testing.assert_allclose(df["confidence_score"].to_numpy(), g["df_conf"], atol=1.5e-3)
    np.testing.assert_allclose(df[[c for c in df.columns if c.startswith("p")]].to_numpy(), g["df_probs"], atol=1.5e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("n_bc", [4, 10])     # 4 = BASELINE config 2 exactly (1 M reads, WDX4), 10 = config 3's shape
def test_full_size_properties_one_million_reads(n_bc):
    """BASELINE config 2 scale (1 M synthetic reads, fused path), through properties that do not need the
    oracle at that size: run-to-run determinism, shard invariance (any split of the reads gives the same
    per-read results and the same count histogram -- what the multi-GPU sharding relies on), a checksum of
    the histogram, agreement of the calls with the generator's true barcodes, and oracle parity on a
    random sample of the 1 M."""
    import torch

    from bench import make_refs
    from warpdemux_amd.engine import DemuxEngine

    spec = synth.SynthSpec(n_barcodes=n_bc)
    K, n = 110, 1_000_000
    refs = make_refs(synth.SynthSpec(n_barcodes=n_bc, noise_sigma=0.25, spikes=False), synth, sig_proc, n_barcodes=n_bc)
    assert refs.shape == (n_bc, K)
    eng = DemuxEngine(refs, 15, 0.1, sig_proc.SegParams(barcode_num_events=K))
    sig, off, a_s, a_e, bc, max_len = eng.synth_packed(spec, 0, n)
    r1 = eng.demux(sig, a_s, a_e, offsets=off, max_len=max_len, want_fpt=True)
    r2 = eng.demux(sig, a_s, a_e, offsets=off, max_len=max_len, want_fpt=True)
    torch.cuda.synchronize()
    for a, b in ((r1.call, r2.call), (r1.status, r2.status), (r1.counts, r2.counts)):
        assert torch.equal(a, b)
    assert torch.equal(r1.dist.view(torch.int32), r2.dist.view(torch.int32))
    assert torch.equal(r1.fpt.view(torch.int64), r2.fpt.view(torch.int64))
    counts = r1.counts.cpu().numpy()
    assert counts.sum() == n and counts[-1] == int((r1.status != 0).sum())
    # shards: three uneven contiguous pieces, accumulated into one histogram
    acc = torch.zeros_like(r1.counts)
    cuts = [0, 333_333, 900_001, n]
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        o = off[lo:hi + 1] - off[lo]
        s = sig[int(off[lo]):int(off[hi])]
        rs = eng.demux(s, a_s[lo:hi], a_e[lo:hi], offsets=o.contiguous(), max_len=max_len, counts=acc)
        assert torch.equal(rs.call, r1.call[lo:hi]) and torch.equal(rs.status, r1.status[lo:hi])
        assert torch.equal(rs.dist.view(torch.int32), r1.dist[lo:hi].view(torch.int32))
    assert torch.equal(acc, r1.counts)
    # the calls mostly recover the generator's barcodes (sanity only)
    ok = r1.status == 0
    assert ok.float().mean().item() > 0.999
    assert (r1.call[ok] == bc[ok]).float().mean().item() > 0.75   # nearest single template, noisy synthetic barcodes
    # oracle parity on 20 000 reads: 40 blocks of 500 drawn from all over the batch, everything bitwise
    rng = np.random.default_rng(7)
    po = orc.SegParams(barcode_num_events=K)
    off_h = off.cpu().numpy()
    checked = 0
    for lo in np.sort(rng.choice(n - 500, 40, replace=False)):
        hi = lo + 500
        o = (off_h[lo:hi + 1] - off_h[lo]).astype(np.int64)
        s = sig[int(off_h[lo]):int(off_h[hi])].cpu().numpy()
        fpt, dwell, stats, status = orc.fingerprint_packed(s, o, a_s[lo:hi].cpu().numpy(), a_e[lo:hi].cpu().numpy(), po)
        assert np.array_equal(status, r1.status[lo:hi].cpu().numpy())
        okb = status == 0
        assert np.array_equal(r1.fpt[lo:hi].cpu().numpy()[okb].view(np.uint64), fpt[okb].view(np.uint64))
        D = orc.dtw_matrix(fpt[okb], refs, 15, 0.1)
        assert np.array_equal(r1.dist[lo:hi].cpu().numpy()[okb].view(np.uint32), D.view(np.uint32))
        assert np.array_equal(r1.call[lo:hi].cpu().numpy()[okb], orc.argmin_rows(D))
        checked += int(okb.sum())
    assert checked > 19_900
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("exact", [False, True])
def test_sliced_launches_of_a_30k_read_batch_vs_oracle(exact):
    """A pass over more reads than one grid admits is cut into equal launch slices with block_base != 0
    (wdx_fingerprint.hip `launch_sliced`, `launch_fp_chunks`, the clip launches): C3 runs 2 x 5 M.  Here the slice is
    forced down (WDX_OPT_MAX_LAUNCH_SLICE) so that 30 000 reads run as five slices of the main kernel, eight of the
    clip kernel, and the list kernels' entries straddle slices -- every read of every slice against the oracle, bitwise,
    through the fused call (fingerprint -> DTW -> call -> histogram) and through the fingerprint entry point."""
    import torch

    from bench import make_refs
    from warpdemux_amd.engine import DemuxEngine

    spec = synth.SynthSpec(n_barcodes=10)
    K, n = 110, 30_000
    refs = make_refs(synth.SynthSpec(n_barcodes=10, noise_sigma=0.25, spikes=False), synth, sig_proc, n_barcodes=10)
    eng = DemuxEngine(refs, 15, 0.1, sig_proc.SegParams(barcode_num_events=K))
    sig, off, a_s, a_e, bc, max_len = eng.synth_packed(spec, 7_000_000, n)
    whole = eng.demux(sig, a_s, a_e, offsets=off, max_len=max_len, want_fpt=True)
    eng.ctx.set_option(_lib.OPT_MAX_LAUNCH_SLICE, 6_000 if not exact else 7_001)   # (7 001: a short last slice)
    eng.ctx.set_option(_lib.OPT_EXACT_PATH, int(exact))
    try:
        sl = eng.demux(sig, a_s, a_e, offsets=off, max_len=max_len, want_fpt=True)
        f_fpt, f_dwell, f_stats, f_status = eng.fingerprint(sig, a_s, a_e, offsets=off, max_len=max_len)
    finally:
        eng.ctx.set_option(_lib.OPT_MAX_LAUNCH_SLICE, 0)
        eng.ctx.set_option(_lib.OPT_EXACT_PATH, 0)
    torch.cuda.synchronize()
    # sliced == unsliced, everything
    assert torch.equal(sl.call, whole.call) and torch.equal(sl.status, whole.status) and torch.equal(sl.counts, whole.counts)
    assert torch.equal(sl.dist.view(torch.int32), whole.dist.view(torch.int32))
    assert torch.equal(sl.fpt.view(torch.int64), whole.fpt.view(torch.int64))
    assert torch.equal(f_fpt.view(torch.int64), whole.fpt.view(torch.int64)) and torch.equal(f_status, whole.status)
    # ... == the oracle, every read
    fpt, dwell, stats, status = orc.fingerprint_packed(sig.cpu().numpy(), off.cpu().numpy().astype(np.int64),
                                                       a_s.cpu().numpy(), a_e.cpu().numpy(), orc.SegParams(barcode_num_events=K))
    assert np.array_equal(status, sl.status.cpu().numpy())
    ok = status == 0
    assert ok.mean() > 0.999
    assert np.array_equal(sl.fpt.cpu().numpy()[ok].view(np.uint64), fpt[ok].view(np.uint64))
    assert np.array_equal(f_dwell.cpu().numpy()[ok], dwell[ok]) and _same(f_stats.cpu().numpy()[ok], stats[ok])
    D = orc.dtw_matrix(fpt[ok], refs, 15, 0.1)
    assert np.array_equal(sl.dist.cpu().numpy()[ok].view(np.uint32), D.view(np.uint32))
    assert np.array_equal(sl.call.cpu().numpy()[ok], orc.argmin_rows(D))
    counts = sl.counts.cpu().numpy()
    assert counts.sum() == n and counts[-1] == int((~ok).sum())
    assert np.array_equal(counts[:-1], np.bincount(orc.argmin_rows(D), minlength=10))
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["rna004", "no_norm", "reach9", "few_events", "quantised", "accept_less", "numpy1_clip", "mixed_failures",
                                  "with_stats", "median_norm", "rna002_triple", "trna_triple", "w12_reach17", "w24_stats"])
def test_split_main_kernel_equals_the_one_piece_kernel_and_the_oracle(case):
    """Round 6: large RNA004 batches run the main fingerprint kernel as a PAIR -- the workgroup-per-read tile kernel exports
    the <= 256 peaks that can matter with the prefix sums of the clipped samples at their boundaries, a wave-per-read tail
    kernel does suppression / top-E / boundaries / event means / normalisation (wdx_fingerprint_split.inc).  Same decisions,
    same arithmetic: fingerprints, dwell times and status must equal the one-piece kernel's (WDX_OPT_NO_SPLIT_TAIL) and the
    oracle's bit for bit -- default parameters, no segment normalisation, the longest suppression reach of the
    instantiation, few events on long windows (more than 256 peaks above the cut: the list kernel's turn), ADC-quantised
    samples (ties and doubts: the retry and exact lists behind the tail kernel), accept_less_cpts, the NumPy-1.x clip rule --
    in one launch pair and in slices of 1 000 reads."""
    import torch

    from warpdemux_amd.engine import DemuxEngine

    kw = dict(barcode_num_events=110)
    spec = synth.SynthSpec(n_barcodes=10)
    n = 6_000
    if case == "no_norm":
        kw.update(seg_norm="none", barcode_num_events=25)
    elif case == "reach9":
        kw.update(min_obs_per_base=9, num_events=80, barcode_num_events=40)
    elif case == "few_events":
        kw.update(num_events=60, min_obs_per_base=3, barcode_num_events=25)
    elif case == "quantised":
        spec = synth.SynthSpec(n_barcodes=10, adc_quantum=2.0) if hasattr(synth.SynthSpec(), "adc_quantum") else spec
    elif case == "accept_less":
        kw.update(accept_less_cpts=True, num_events=125, barcode_num_events=25)
    elif case == "numpy1_clip":
        kw.update(clip_bounds="float64", outlier_thresh=2.7)
    elif case == "median_norm":
        kw.update(seg_norm="median", barcode_num_events=25)
    elif case == "rna002_triple":       # NBT = 2: the tail kernel's 20-peak windows (lanes +-1 by DPP, +-2 by ds_bpermute)
        kw.update(num_events=110, min_obs_per_base=15, running_stat_width=30, barcode_num_events=25)
    elif case == "trna_triple":
        kw.update(num_events=120, min_obs_per_base=9, running_stat_width=18, barcode_num_events=25)
    elif case == "w12_reach17":
        kw.update(num_events=110, min_obs_per_base=12, running_stat_width=12, barcode_num_events=25)
    elif case == "w24_stats":
        kw.update(num_events=110, min_obs_per_base=12, running_stat_width=24, barcode_num_events=40)
    want_stats = case in ("with_stats", "median_norm", "w24_stats", "rna002_triple")
    pt = sig_proc.SegParams(**kw)
    K = pt.barcode_num_events
    eng = DemuxEngine(np.zeros((2, K)), 15, 0.1, pt)
    sig, off, a_s, a_e, bc, max_len = eng.synth_packed(spec, 3_000_000, n)
    if case == "quantised" and not hasattr(synth.SynthSpec(), "adc_quantum"):
        sig = torch.round(sig * 0.5) * 2.0      # a coarse 2 pA grid: plateaus and exact score ties at scale
    okm = None
    if case == "mixed_failures":
        # every way a read leaves the split pair early: failed detection, a window too short to segment, a NaN and a negative
        # sample in the window (the clip record refuses: exact kernel), windows cut short -- the tail kernel must not pick up
        # the previous occupant of such a read's list slot (slices of 1 000 reads reuse the slots six times)
        okm = torch.ones(n, dtype=torch.uint8, device=sig.device)
        okm[5::97] = 0
        a_e = a_e.clone()
        a_e[11::101] = a_s[11::101] + 40             # window = 240 samples: below every kernel's minimum
        a_e[17::103] = a_s[17::103] + 900            # short window: parameters shrink (sig_proc.py:526-533) -> exact kernel
        sig = sig.clone()
        offh = off.cpu().numpy()
        for r in range(23, n, 107):
            sig[int(offh[r]) + 700] = float("nan")
        for r in range(29, n, 109):
            sig[int(offh[r]) + 900] = -3.0
    got = {}
    for name, opts in (("split", {}), ("split_sliced", {_lib.OPT_MAX_LAUNCH_SLICE: 1000}), ("one_piece", {_lib.OPT_NO_SPLIT_TAIL: 1})):
        for o, v in opts.items():
            eng.ctx.set_option(o, v)
        try:
            got[name] = eng.fingerprint(sig, a_s, a_e, offsets=off, max_len=max_len, want_stats=want_stats, ok=okm)
            torch.cuda.synchronize()
        finally:
            for o in opts:
                eng.ctx.set_option(o, 0)
    for name in ("split_sliced", "one_piece"):
        assert torch.equal(got[name][3], got["split"][3]), name
        okd = got["split"][3] == 0
        assert torch.equal(got[name][0][okd].view(torch.int64), got["split"][0][okd].view(torch.int64)), name
        assert torch.equal(got[name][1][okd], got["split"][1][okd]), name
        if want_stats:
            assert torch.equal(got[name][2][okd].view(torch.int64), got["split"][2][okd].view(torch.int64)), name
    m = 1_500     # ... and the oracle on a quarter of them
    fpt, dwell, stats, status = orc.fingerprint_packed(sig[: int(off[m].item())].cpu().numpy(), off[: m + 1].cpu().numpy().astype(np.int64),
                                                       a_s[:m].cpu().numpy(), a_e[:m].cpu().numpy(), orc.SegParams(**{k: v for k, v in kw.items() if k != "clip_bounds"},
                                                                                                                      **({"clip_bounds_f64": True} if case == "numpy1_clip" else {})))
    if okm is not None:     # (the packed oracle entry point has no success flags: a failed detection is status 1, sig_proc.py:400-407)
        status[okm[:m].cpu().numpy() == 0] = 1
    st = got["split"][3][:m].cpu().numpy()
    assert np.array_equal(st, status)
    ok = status == 0
    assert ok.mean() > (0.5 if case in ("few_events", "quantised") else 0.9)
    if case == "mixed_failures":
        assert set(np.unique(status).tolist()) >= {0, 1}, np.unique(status)
    assert np.array_equal(got["split"][0][:m].cpu().numpy()[ok].view(np.uint64), fpt[ok].view(np.uint64))
    assert np.array_equal(got["split"][1][:m].cpu().numpy()[ok], dwell[ok])
    if want_stats:
        assert np.array_equal(got["split"][2][:m].cpu().numpy()[ok].view(np.uint64), stats[ok].view(np.uint64))
    eng.close()


@pytest.mark.gpu
def test_fingerprint_mid_length_windows_take_the_8192_instantiation(monkeypatch):
    """Adapter windows of 6145..8192 samples are listed on the device by the 6144-sample fast kernel and
    retried by the 8192-sample one; longer ones (<= 11200) take the exact path.  All three must agree with
    the oracle bit for bit, in one mixed batch."""
    rng = np.random.default_rng(21)
    lens = [5000, 6144, 6145, 6500, 7000, 7777, 8191, 8192, 8193, 9000, 4000, 6300] * 8
    n, stride = len(lens), 9200
    mb = np.full((n, stride), np.nan, dtype=np.float32)
    for i, ln in enumerate(lens):
        ev = rng.integers(20, 60)
        lvl = np.repeat(rng.normal(80, 15, ln // ev + 1), ev)[:ln]
        mb[i, :ln] = np.round((lvl + rng.normal(0, 2, ln)) * 8) / 8   # quantised like ADC counts
    a_s = np.zeros(n, dtype=np.int32)
    a_e = np.array(lens, dtype=np.int32)
    for K, seg_norm in ((25, "mean"), (110, "median")):
        ph = sig_proc.SegParams(padding=0, barcode_num_events=K, seg_norm=seg_norm)
        po = orc.SegParams(padding=0, barcode_num_events=K, seg_norm=seg_norm)
        fb = sig_proc.fingerprint_batch(mb, a_s, a_e, ph)
        with _exact_path():
            sl = sig_proc.fingerprint_batch(mb, a_s, a_e, ph)
        fpt, dwell, stats, status = orc.fingerprint_batch(mb, a_s, a_e, po)
        assert np.array_equal(fb.status, status) and np.array_equal(sl.status, status) and (status == 0).all()
        assert _same(fb.fpt, fpt) and _same(fb.dwell, dwell) and _same(fb.stats, stats)
        assert _same(sl.fpt, fpt) and _same(sl.dwell, dwell) and _same(sl.stats, stats)
        # the launch chain of large batches: 5120-sample main kernel -> 6144 list kernel -> 8192 list kernel -> retry
        # -> exact kernel, also with peak lists too small for most reads (everything moves up the chain) and with
        # a one-entry grid of the per-entry list kernels (everything beyond it goes to the striding kernel)
        for peak_cap in (0, 600):
            with _chain(), _option(_lib.OPT_FAST_PEAK_CAP, peak_cap):
                ch = sig_proc.fingerprint_batch(mb, a_s, a_e, ph)
            assert np.array_equal(ch.status, status)
            assert _same(ch.fpt, fpt) and _same(ch.dwell, dwell) and _same(ch.stats, stats)


def test_exact_kernel_behind_the_chain_takes_the_clip_records():
    """The exact general kernel behind the launch chain reads the bounds of a CLIP_OK record instead of redoing the two
    medians (windows with NaN / mostly negative samples have no such record and keep the medians): every read forced
    onto it (peak lists too small for any read), all signal styles, with the records and without -- the oracle's bits."""
    rng = np.random.default_rng(4711)
    styles = ["gauss", "quantised", "integers", "heavy", "negative", "spiky", "flat_runs", "clipped_low"]
    n, stride = 192, 9000
    mb = np.full((n, stride), np.nan, dtype=np.float32)
    a_s = np.zeros(n, dtype=np.int32)
    a_e = np.zeros(n, dtype=np.int32)
    for i in range(n):
        ln = int(rng.choice([300, 1400, 4000, 5000, 5200, 6100, 6200, 7000, 8250, 8800])) + int(rng.integers(-40, 40))
        mb[i, :ln + 50] = _styled_signal(rng, ln + 50, styles[i % len(styles)])
        a_s[i], a_e[i] = 20, ln
    mb[7, 1000] = np.nan
    mb[15, 2000] = np.inf
    kw = dict(padding=20, barcode_num_events=25)
    fpt, dwell, stats, status = orc.fingerprint_batch(mb, a_s, a_e, orc.SegParams(**kw))
    assert (status == 0).sum() > 0.5 * n
    for reuse_off in (0, 1):
        with _chain(), _option(_lib.OPT_FAST_PEAK_CAP, 64), _option(_lib.OPT_NO_CLIP_REUSE, reuse_off):
            fb = sig_proc.fingerprint_batch(mb, a_s, a_e, sig_proc.SegParams(**kw))
        assert np.array_equal(fb.status, status), (reuse_off, np.flatnonzero(fb.status != status))
        assert _same(fb.fpt, fpt) and _same(fb.dwell, dwell) and _same(fb.stats, stats), reuse_off
        # ... and the exact kernel for the whole batch (WDX_OPT_EXACT_PATH; a width without a fast instantiation: 10), where
        # large batches get their records from a clip launch of their own, with both signal normalisations
        for kw2 in (dict(kw), dict(kw, running_stat_width=10), dict(kw, sig_norm="mean"), dict(kw, sig_norm="median")):
            o2 = orc.fingerprint_batch(mb, a_s, a_e, orc.SegParams(**kw2))
            with _chain(), _exact_path(), _option(_lib.OPT_NO_CLIP_REUSE, reuse_off):
                ex = sig_proc.fingerprint_batch(mb, a_s, a_e, sig_proc.SegParams(**kw2))
            assert np.array_equal(ex.status, o2[3]), (reuse_off, kw2)
            assert _same(ex.fpt, o2[0]) and _same(ex.dwell, o2[1]) and _same(ex.stats, o2[2]), (reuse_off, kw2)


@pytest.mark.parametrize("triple", [(110, 6, 12), (120, 9, 18)])
def test_peak_lists_that_outgrow_the_last_list_kernel_go_to_the_exact_scores_launch(triple, golden_dir):
    """A batch whose windows stay below 6 145 samples has no 8192-sample list kernel behind the 6144-sample one: a read
    whose threshold-filtered peak list outgrows 512 entries there (clip-heavy reads with many level changes: ~740 local
    maxima per 4.7 k samples, plateaus of equal scores where clipped samples slide in and out) is redone by the exact-scores
    launch with its 1376 entries instead of the exact general kernel.  Same bits as the oracle either way."""
    E, d, w = triple
    rng = np.random.default_rng(97 + w)
    n = 2304
    consensus = np.load(os.path.join(golden_dir, "g8_refine.npz"))["consensus"]     # (84 levels of a tRNA adapter: wide spread)
    rows = []
    for i in range(n):
        lv = np.concatenate([rng.normal(0, 1, int(rng.integers(2, 34))), consensus, rng.normal(0, 1, 30)]) * 12.0 + 85.0
        dw = rng.integers(12, 60, lv.size)
        x = np.repeat(lv, dw) + rng.normal(0, 1.5, int(dw.sum()))
        rows.append(x[:6100].astype(np.float32))
    stride = 6144
    mb = np.full((n, stride), np.nan, dtype=np.float32)
    for i, r in enumerate(rows):
        mb[i, :r.size] = r
    a_s = np.zeros(n, dtype=np.int32)
    a_e = np.array([r.size for r in rows], dtype=np.int32)
    kw = dict(padding=0, num_events=E, min_obs_per_base=d, running_stat_width=w, barcode_num_events=25)
    fb = sig_proc.fingerprint_batch(mb, a_s, a_e, sig_proc.SegParams(**kw))      # n >= 2048: the launch chain
    sub = np.arange(0, n, 3)
    fpt, dwell, stats, status = orc.fingerprint_batch(mb[sub], a_s[sub], a_e[sub], orc.SegParams(**kw))
    assert np.array_equal(fb.status[sub], status) and (status == 0).mean() > 0.5
    assert _same(fb.fpt[sub], fpt) and _same(fb.dwell[sub], dwell) and _same(fb.stats[sub], stats)
    with _exact_path():
        sl = sig_proc.fingerprint_batch(mb, a_s, a_e, sig_proc.SegParams(**kw))
    assert np.array_equal(fb.status, sl.status) and _same(fb.fpt, sl.fpt) and _same(fb.dwell, sl.dwell)


@pytest.mark.parametrize("triple", [(110, 6, 12), (120, 9, 18), (110, 15, 30), (110, 3, 6), (110, 12, 24), (110, 17, 36)])
def test_long_plateaus_of_equal_scores(triple):
    """One deviating sample inside a long stretch of equal samples (what the outlier clip leaves of a stalled read) gives
    2 W exactly equal scores between zeros -- scipy's plateau rule puts ONE peak at their midpoint.  Integer-valued samples
    and a deviation that is a multiple of W make every operation exact, so the ties are exact in the reference as well.
    The fast kernels follow such a plateau lane by lane to the end of a wave's scores (registers hold the next two lanes');
    plateaus across a wave's or a tile's end, and at the very end of the score curve, take the exact kernel.  Same
    change-points as the oracle in every case."""
    E, d, w = triple
    rng = np.random.default_rng(1000 * w + d)
    n, stride = 96, 9000
    mb = np.full((n, stride), np.nan, dtype=np.float32)
    a_e = np.zeros(n, dtype=np.int32)
    for i in range(n):
        ln = int(rng.integers(4200, 8800))
        dw = int(rng.integers(14, 40))
        x = np.round(np.repeat(rng.normal(80, 15, ln // dw + 1), dw)[:ln] + rng.normal(0, 2, ln))
        run = 4 * w + int(rng.integers(4, 40))
        starts = list(rng.integers(100, ln - run - 100, 10)) + [ln - run]          # (the last one: up to the window's end)
        for s0 in starts:
            x[s0:s0 + run] = 400.0
            x[s0 + int(rng.integers(2 * w, run - 2 * w + 1))] = 400.0 - 360.0 * int(rng.integers(1, 2))   # 360 = lcm of the widths
        mb[i, :ln] = x
        a_e[i] = ln
    a_s = np.zeros(n, dtype=np.int32)
    kw = dict(padding=0, num_events=E, min_obs_per_base=d, running_stat_width=w, barcode_num_events=25, outlier_thresh=1.0e6)
    fpt, dwell, stats, status = orc.fingerprint_batch(mb, a_s, a_e, orc.SegParams(**kw))
    assert (status == 0).sum() > 0.6 * n
    for chain in (False, True):
        if chain:
            with _chain():
                fb = sig_proc.fingerprint_batch(mb, a_s, a_e, sig_proc.SegParams(**kw))
        else:
            fb = sig_proc.fingerprint_batch(mb, a_s, a_e, sig_proc.SegParams(**kw))
        assert np.array_equal(fb.status, status), (chain, np.flatnonzero(fb.status != status))
        assert _same(fb.fpt, fpt) and _same(fb.dwell, dwell) and _same(fb.stats, stats), chain


def test_launch_chain_lists_longer_than_their_grids():
    """The per-entry list kernels are launched with grids sized for the expected share of a batch (a quarter for
    windows beyond the main instantiation, 1/64 -- at least 1024 -- for exact-score retries); entries beyond the grid
    belong to the striding 8192-sample kernel.  A batch in which EVERY window is longer than 5120 samples, and one in
    which every read is redone with exact scores (samples on a coarse 2 pA grid: ties everywhere), must still match
    the oracle read for read."""
    rng = np.random.default_rng(77)
    n, stride = 4608, 6000
    mb = np.full((n, stride), np.nan, dtype=np.float32)
    lens = rng.integers(5200, 5900, n)
    for i, ln in enumerate(lens):
        ev = int(rng.integers(25, 55))
        lvl = np.repeat(rng.normal(85, 14, ln // ev + 1), ev)[:ln]
        mb[i, :ln] = (lvl + rng.normal(0, 2, ln)).astype(np.float32)
    a_s = np.zeros(n, dtype=np.int32)
    a_e = lens.astype(np.int32)
    ph, po = sig_proc.SegParams(padding=0, barcode_num_events=25), orc.SegParams(padding=0, barcode_num_events=25)
    fb = sig_proc.fingerprint_batch(mb, a_s, a_e, ph)
    sub = np.arange(0, n, 9)  # the oracle on every ninth read (all grid / beyond-grid positions of the list occur)
    fpt, dwell, stats, status = orc.fingerprint_batch(mb[sub], a_s[sub], a_e[sub], po)
    assert np.array_equal(fb.status[sub], status) and (status == 0).mean() > 0.98
    assert _same(fb.fpt[sub], fpt) and _same(fb.dwell[sub], dwell) and _same(fb.stats[sub], stats)
    with _exact_path():
        sl = sig_proc.fingerprint_batch(mb, a_s, a_e, ph)
    assert np.array_equal(fb.status, sl.status) and _same(fb.fpt, sl.fpt) and _same(fb.dwell, sl.dwell)

    spec = synth.SynthSpec(n_barcodes=10)
    mb2, a_s2, a_e2, _ = synth.generate_minibatch(spec, 55_000, 2560, 9000)
    coarse = np.round(mb2 / np.float32(2.0)).astype(np.float32) * np.float32(2.0)
    fc = sig_proc.fingerprint_batch(coarse, a_s2, a_e2, sig_proc.SegParams(barcode_num_events=110))
    with _option(_lib.OPT_FAST_EXACT_SCORES, 1):
        fe = sig_proc.fingerprint_batch(coarse, a_s2, a_e2, sig_proc.SegParams(barcode_num_events=110))
    assert np.array_equal(fc.status, fe.status) and _same(fc.fpt, fe.fpt) and _same(fc.dwell, fe.dwell)
    sub = np.arange(0, 2560, 7)
    fpt, dwell, stats, status = orc.fingerprint_batch(coarse[sub], a_s2[sub], a_e2[sub], orc.SegParams(barcode_num_events=110))
    assert np.array_equal(fc.status[sub], status)
    good = status == 0
    assert good.mean() > 0.95
    assert _same(fc.fpt[sub][good], fpt[good]) and _same(fc.dwell[sub][good], dwell[good])


def test_accept_less_cpts_runs_on_the_fast_kernels():
    """segmentation.accept_less_cpts = True (sig_proc.py:185-188: fewer than num_events peaks -> fewer events instead of
    a failure) used to send the whole batch to the exact kernel.  Now the fast kernels take it and hand over only the
    reads it concerns.  Windows of 1 270 .. 1 700 samples (the shortest that keep the configured width 12) have ~100 .. 140
    kept peaks: some below num_events = 110, so both outcomes occur; everything against the oracle, both settings."""
    rng = np.random.default_rng(79)
    n, stride = 3072, 4800
    mb = np.full((n, stride), np.nan, dtype=np.float32)
    lens = np.concatenate([rng.integers(1270, 1701, 2048), rng.integers(3000, 4700, n - 2048)])
    for i, ln in enumerate(lens):
        ev = int(rng.integers(10, 30))
        lvl = np.repeat(rng.normal(85, 14, ln // ev + 1), ev)[:ln]
        mb[i, :ln] = (lvl + rng.normal(0, 2, ln)).astype(np.float32)
    a_s = np.zeros(n, dtype=np.int32)
    a_e = lens.astype(np.int32)
    seen = {}
    for acc in (False, True):
        kw = dict(padding=0, barcode_num_events=25, accept_less_cpts=acc)
        fb = sig_proc.fingerprint_batch(mb, a_s, a_e, sig_proc.SegParams(**kw))
        fpt, dwell, stats, status = orc.fingerprint_batch(mb, a_s, a_e, orc.SegParams(**kw))
        assert np.array_equal(fb.status, status)
        ok = status == 0
        assert _same(fb.fpt[ok], fpt[ok]) and _same(fb.dwell[ok], dwell[ok]) and _same(fb.stats[ok], stats[ok])
        seen[acc] = status
    gave_up = (seen[False] != 0) & (seen[True] == 0)      # too few peaks: a failure without the option, a fingerprint with it
    assert gave_up.sum() > 20 and (seen[False] == 0).sum() > 2000


def test_streaming_list_longer_than_its_grid():
    """In batches of more than 2 M reads whose windows stay below 8192 samples the streaming kernels' grids cover a
    sixteenth of the batch and the striding 8192-sample kernel takes the list entries beyond (WDX_OPT_MAX_LAUNCH_SLICE
    selects that form for a batch of any size): a batch in which EVERY window has 6200 .. 8000 samples (4 608 list
    entries, 288 of them inside the grid) still matches the oracle, on either side of the grid's end."""
    rng = np.random.default_rng(78)
    n, stride = 4608, 8064
    mb = np.full((n, stride), np.nan, dtype=np.float32)
    lens = rng.integers(6200, 8001, n)
    for i, ln in enumerate(lens):
        ev = int(rng.integers(25, 55))
        lvl = np.repeat(rng.normal(85, 14, ln // ev + 1), ev)[:ln]
        mb[i, :ln] = (lvl + rng.normal(0, 2, ln)).astype(np.float32)
    a_s = np.zeros(n, dtype=np.int32)
    a_e = lens.astype(np.int32)
    ph, po = sig_proc.SegParams(padding=0, barcode_num_events=25), orc.SegParams(padding=0, barcode_num_events=25)
    with _option(_lib.OPT_MAX_LAUNCH_SLICE, 1_000_000):
        fb = sig_proc.fingerprint_batch(mb, a_s, a_e, ph)
    plain = sig_proc.fingerprint_batch(mb, a_s, a_e, ph)         # (every entry inside the grid)
    assert np.array_equal(fb.status, plain.status) and _same(fb.fpt, plain.fpt) and _same(fb.dwell, plain.dwell)
    sub = np.concatenate([np.arange(0, 4000, 16), np.arange(4000, n, 3)])   # (list order is arbitrary: sample everywhere)
    fpt, dwell, stats, status = orc.fingerprint_batch(mb[sub], a_s[sub], a_e[sub], po)
    assert np.array_equal(fb.status[sub], status) and (status == 0).mean() > 0.98
    assert _same(fb.fpt[sub], fpt) and _same(fb.dwell[sub], dwell) and _same(fb.stats[sub], stats)
    with _exact_path():
        sl = sig_proc.fingerprint_batch(mb, a_s, a_e, ph)
    assert np.array_equal(fb.status, sl.status) and _same(fb.fpt, sl.fpt) and _same(fb.dwell, sl.dwell)


def test_window_lengths_at_every_capacity_and_chunk_edge():
    """Windows exactly at, one below and one above every instantiation's capacity, at multiples of the 1024-sample
    step that sets a thread's chunk of the event-mean prefix sums (where the last thread owns a full chunk or the
    thread after the end owns nothing), and at the fast path's minimum -- single launch and launch chain vs oracle."""
    rng = np.random.default_rng(4242)
    lens = [256, 257, 260, 1023, 1024, 1025, 2047, 2048, 2049, 3072, 4095, 4096, 4097, 5116, 5119, 5120, 5121, 5124,
            6143, 6144, 6145, 7168, 8191, 8192, 8193]
    lens = lens * 3
    n, stride = len(lens), 8400
    mb = np.full((n, stride), np.nan, dtype=np.float32)
    for i, ln in enumerate(lens):
        ev = int(rng.integers(14, 40))
        lvl = np.repeat(rng.normal(90, 16, ln // ev + 1), ev)[:ln]
        mb[i, :ln] = (lvl + rng.normal(0, 2, ln)).astype(np.float32)
    a_s = np.zeros(n, dtype=np.int32)
    a_e = np.array(lens, dtype=np.int32)
    for num_events in (6, 24):   # (a 256-sample window must still hold num_events events)
        kw = dict(padding=0, num_events=num_events, barcode_num_events=num_events, min_obs_per_base=3)
        ph, po = sig_proc.SegParams(**kw), orc.SegParams(**kw)
        fpt, dwell, stats, status = orc.fingerprint_batch(mb, a_s, a_e, po)
        assert (status == 0).mean() > 0.9
        for ctx in (contextlib.nullcontext(), _chain()):
            with ctx:
                fb = sig_proc.fingerprint_batch(mb, a_s, a_e, ph)
            assert np.array_equal(fb.status, status)
            assert _same(fb.fpt, fpt) and _same(fb.dwell, dwell) and _same(fb.stats, stats)


_RANDOMISED_OK = []


def _styled_signal(rng, ln, style):
    ev = int(rng.integers(12, 70))
    lvl = np.repeat(rng.normal(90, 18, ln // ev + 2), ev)[:ln]
    x = lvl + rng.normal(0, rng.uniform(0.5, 4.0), ln)
    if style == "quantised":
        x = np.round(x * 5.8) / 5.8 + 0.37
    elif style == "integers":
        x = np.round(x)
    elif style == "heavy":
        x = x + rng.standard_t(2, ln) * 3
    elif style == "negative":
        x = x - 95.0
    elif style == "spiky":
        k = rng.integers(0, ln, 25)
        x[k] += rng.choice([-1, 1], 25) * rng.uniform(60, 400, 25)
    elif style == "flat_runs":
        for _ in range(6):
            a = int(rng.integers(0, ln - 200))
            x[a:a + int(rng.integers(15, 180))] = x[a]
    elif style == "clipped_low":
        x = np.maximum(x, np.percentile(x, 30))
    return x.astype(np.float32)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(int(os.environ.get("WDX_SOAK_SEEDS", "16"))))   # soak: WDX_SOAK_SEEDS=400
def test_fingerprint_randomised_configs_and_signal_styles(seed):
    """Random segmentation parameters x signal styles (quantised, integer-valued, heavy-tailed, negative,
    spiky, flat runs, low-clipped) x window lengths on both sides of every fast-path capacity: GPU == oracle,
    bit for bit, whichever kernel a read ends up in."""
    rng = np.random.default_rng(1000 + seed)
    styles = ["gauss", "quantised", "integers", "heavy", "negative", "spiky", "flat_runs", "clipped_low"]
    n, stride = 96, 9100
    mb = np.full((n, stride), np.nan, dtype=np.float32)
    a_s = np.zeros(n, dtype=np.int32)
    a_e = np.zeros(n, dtype=np.int32)
    for i in range(n):
        ln = int(rng.choice([300, 900, 1400, 2500, 4000, 4100, 5000, 6100, 6200, 7000, 8150, 8250, 8900]))
        ln += int(rng.integers(-40, 40))
        st = int(rng.integers(0, 60))
        mb[i, :st + ln + 50] = _styled_signal(rng, st + ln + 50, styles[i % len(styles)])
        a_s[i], a_e[i] = st, st + ln
    ok = (rng.uniform(size=n) > 0.04).astype(np.uint8)
    kw = dict(padding=int(rng.choice([0, 30, 100])), outlier_thresh=float(rng.choice([3.0, 5.0, 8.0])),
              num_events=int(rng.choice([60, 110, 120])), min_obs_per_base=int(rng.choice([2, 3, 6, 9, 15])),
              running_stat_width=int(rng.choice([12, 12, 12, 10, 18, 30, 6, 24, 36])),
              seg_norm=str(rng.choice(["mean", "median", "none"])))
    kw["barcode_num_events"] = int(rng.choice([10, 25, kw["num_events"]]))
    ph, po = sig_proc.SegParams(**kw), orc.SegParams(**kw)
    fb = sig_proc.fingerprint_batch(mb, a_s, a_e, ph, success=ok)
    fpt, dwell, stats, status = orc.fingerprint_batch(mb, a_s, a_e, po, ok=ok)
    assert np.array_equal(fb.status, status), (kw, np.flatnonzero(fb.status != status))
    assert _same(fb.fpt, fpt) and _same(fb.dwell, dwell) and _same(fb.stats, stats), kw
    with _chain():  # the same batch through the launch chain of large batches (approximate score keys, lists, retry)
        ch = sig_proc.fingerprint_batch(mb, a_s, a_e, ph, success=ok)
    assert np.array_equal(ch.status, status), (kw, np.flatnonzero(ch.status != status))
    assert _same(ch.fpt, fpt) and _same(ch.dwell, dwell) and _same(ch.stats, stats), kw
    # the exact general kernel alone, with its suppression / top-E on the compact peak list and in position space
    for no_list in (0, 1):
        ctx = _lib.default_context(None)
        ctx.set_option(_lib.OPT_EXACT_NO_PEAK_LIST, no_list)
        try:
            with _exact_path():
                ex = sig_proc.fingerprint_batch(mb, a_s, a_e, ph, success=ok)
        finally:
            ctx.set_option(_lib.OPT_EXACT_NO_PEAK_LIST, 0)
        assert np.array_equal(ex.status, status), (kw, no_list, np.flatnonzero(ex.status != status))
        assert _same(ex.fpt, fpt) and _same(ex.dwell, dwell) and _same(ex.stats, stats), (kw, no_list)
    # (some parameter draws fail every read in the reference too -- statuses are compared above; the draws
    # as a whole must exercise the success path)
    _RANDOMISED_OK.append(int((status == 0).sum()))
    if seed == 15:
        assert sum(_RANDOMISED_OK) > 4 * n


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(int(os.environ.get("WDX_SOAK_SEEDS", "24"))))
def test_dtw_randomised_shapes_windows_penalties(seed):
    """Random series lengths, windows, penalties and batch shapes: every DTW kernel the dispatcher can pick
    (rolling band 8/15/16/32, unrolled 25x15, wavefront rows for few pairs, scratch rows for wide windows,
    fused argmin or separate) against the oracle, float32 distances bit for bit."""
    rng = np.random.default_rng(5000 + seed)
    L = int(rng.choice([1, 2, 7, 16, 24, 25, 26, 28, 29, 40, 64, 110, 111, 150]))
    w = rng.choice([None, 0, 1, 2, 3, 8, 9, 15, 16, 17, 31, 32, 33, 40, L, L + 3])
    w = None if w is None else int(w)
    p = rng.choice([None, 0.0, 0.1, 0.75, 1.5])
    p = None if p is None else float(p)
    nX = int(rng.choice([1, 2, 5, 63, 64, 65, 300, 2500]))
    nY = int(rng.choice([1, 2, 4, 10, 33, 200, 851]))
    if nX * nY * L > 4e7:
        nX = 300
    X, Y = rng.normal(size=(nX, L)), rng.normal(size=(nY, L))
    if rng.uniform() < 0.2 and nX > 2:
        X[1, rng.integers(0, L)] = np.nan
    ref = orc.dtw_matrix(X, Y, w, p)
    got, am = pdist.nearest_reference(X, Y, w, p)
    _check_dist(got, ref)
    ok_rows = ~np.isnan(ref).any(axis=1)
    assert np.array_equal(am[ok_rows], np.argmin(ref[ok_rows], axis=1))
    _check_dist(pdist.distance_matrix_to(X, Y, window=w, penalty=p, n_jobs=1), ref)

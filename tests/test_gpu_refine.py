"""N3 consensus-guided barcode refinement on the device (wdx_fingerprint_refine_batch) against the reference-derived
fixture g8 and, on many more reads, against the oracle.  Bit-exact: status, query start/end, barcode start, dwell,
float64 fingerprints and stats."""
import os
from types import SimpleNamespace as NS

import numpy as np
import pytest

from oracle import wdx_oracle as orc
from test_oracle_refine import params_from
from warpdemux_amd import sig_proc

pytestmark = pytest.mark.gpu


def _same(a, b):
    return np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)


def test_refinement_golden_vectors(golden_dir):
    g = np.load(os.path.join(golden_dir, "g8_refine.npz"))
    seen = set()
    for k in range(int(g["n"])):
        seg, ref = params_from(g, k)
        row = g[f"row_{k}"]
        a_s, a_e = (int(v) for v in g[f"args_{k}"])
        fb = sig_proc.fingerprint_refine_batch(row.reshape(1, -1), [a_s], [a_e], sig_proc.SegParams(**seg),
                                               sig_proc.RefineParams(query=g["consensus"], **ref))
        st, tag = int(g[f"status_{k}"]), str(g[f"tag_{k}"])
        assert int(fb.status[0]) == st, f"case {k} ({tag}): status {fb.status[0]} != {st}"
        seen.add(st)
        if st in (0, 6):
            assert _same(fb.stats[0], g[f"stats_{k}"]), f"case {k} ({tag}) stats"
            assert _same(fb.refine_idx[0], g[f"idx_{k}"]), f"case {k} ({tag}) query start/end, barcode start"
        if st == 0:
            assert _same(fb.fpt[0], g[f"fpt_{k}"]), f"case {k} ({tag}) fpt"
            assert _same(fb.dwell[0], g[f"dwell_{k}"]), f"case {k} ({tag}) dwell"
        else:
            assert np.isnan(fb.fpt).all() and (fb.dwell == 0).all()
    assert {0, 3, 5, 6} <= seen


def _reads(consensus, n, seed):
    rng = np.random.default_rng(seed)
    rows = []
    for i in range(n):
        n_lead = int(rng.integers(2, 34))
        emb = rng.random() > 0.15
        lv = list(rng.normal(0, 1, n_lead)) + list(consensus if emb else rng.normal(0, 1, consensus.size)) + list(rng.normal(0, 1, 30))
        lv = np.array(lv) * 12.0 + 85.0
        dw = rng.integers(12, 60, lv.size)
        x = np.repeat(lv, dw) + rng.normal(0, rng.uniform(0.8, 3.0), int(dw.sum()))
        rows.append(x.astype(np.float32))
    stride = max(r.size for r in rows)
    mb = np.full((n, stride), np.nan, dtype=np.float32)
    for i, r in enumerate(rows):
        mb[i, : r.size] = r
    a_s = np.full(n, 100, dtype=np.int32)
    a_e = np.array([r.size - 100 for r in rows], dtype=np.int32)
    return mb, a_s, a_e


@pytest.mark.parametrize("norms", [("mean", "mean"), ("median", "median"), ("mean", "none")])
def test_refinement_minibatch_vs_oracle(golden_dir, norms):
    consensus = np.load(os.path.join(golden_dir, "g8_refine.npz"))["consensus"]
    mb, a_s, a_e = _reads(consensus, 96, 11)
    ok = np.ones(96, dtype=np.uint8)
    ok[5] = 0
    mb[7, 2000:2004] = np.nan          # NaN in the window
    a_e[9] = a_s[9] + 900              # far too short
    seg = dict(min_obs_per_base=9, running_stat_width=18, num_events=120, seg_norm=norms[0])
    ref = dict(subseq_norm=norms[1], barcode_segm_events=25, barcode_keep_events=25)
    fb = sig_proc.fingerprint_refine_batch(mb, a_s, a_e, sig_proc.SegParams(barcode_num_events=25, **seg),
                                           sig_proc.RefineParams(query=consensus, **ref), success=ok)
    fpt, dwell, stats, idx, status = orc.fingerprint_refine_batch(mb, a_s, a_e, orc.SegParams(barcode_num_events=25, **seg),
                                                                  orc.RefineParams(query=consensus, **ref), ok=ok)
    assert np.array_equal(fb.status, status), (fb.status, status)
    assert (status == 6).sum() >= 3 and status[5] == 1
    if norms[1] != "none":   # un-normalised event means (pA) do not match the z-scored consensus: all outliers
        assert (status == 0).sum() >= 25
    good, rep = status == 0, (status == 0) | (status == 6)
    assert _same(fb.fpt[good], fpt[good]) and _same(fb.dwell[good], dwell[good])
    assert _same(fb.stats[rep], stats[rep]) and _same(fb.refine_idx[rep], idx[rep])
    assert np.isnan(fb.fpt[~good]).all()


def test_refinement_through_the_reference_shaped_api(golden_dir):
    """detect_results_to_fpt(_batch) with segmentation.consensus_refinement = True: the fields the reference's
    ReadResult carries in this mode (sig_proc.py:590-605, 497-512)."""
    consensus = np.load(os.path.join(golden_dir, "g8_refine.npz"))["consensus"]
    mb, a_s, a_e = _reads(consensus, 12, 5)
    spc = NS(sig_extract=NS(padding=100, normalization="none"), core=NS(sig_norm_outlier_thresh=5.0),
             segmentation=NS(min_obs_per_base=9, running_stat_width=18, num_events=120, accept_less_cpts=False,
                             normalization="mean", barcode_num_events=[25, 25], consensus_refinement=True,
                             consensus_model="rna004_130bps_v1_0", consensus_subseq_match_normalization="mean",
                             consensus_subseq_match_penalty=1.5, consensus_subseq_match_psi=[5, 0, 40, 0],
                             consensus_subseq_match_ub_start=18, consensus_subseq_match_lb_end=69,
                             consensus_subseq_match_ub_end=97, refinement_optimal_cpts=False))
    drs = [sig_proc.DetectResults(True, "", int(a_s[i]), int(a_e[i])) for i in range(12)]
    res = sig_proc.detect_results_to_fpt_batch(mb, spc, drs, consensus_query=consensus)
    fpt, dwell, stats, idx, status = orc.fingerprint_refine_batch(
        mb, a_s, a_e, orc.SegParams(min_obs_per_base=9, running_stat_width=18, num_events=120),
        orc.RefineParams(query=consensus))
    assert {0, 6} <= set(status.tolist())
    for i, r in enumerate(res):
        assert r.success == (status[i] == 0)
        if status[i] in (0, 6):
            assert (r.seg_cons_query_start, r.seg_cons_query_end, r.sig_barcode_start) == tuple(int(v) for v in idx[i])
            assert r.adapter_event_mean == stats[i, 2]
        if status[i] == 0:
            assert _same(r.barcode_fpt, fpt[i]) and _same(r.dwell_times, dwell[i])
        if status[i] == 6:
            assert r.fail_reason == "consensus query outlier" and r.barcode_fpt.size == 0
    one = sig_proc.detect_results_to_fpt(mb[0], spc, drs[0], consensus)
    assert _same(one.barcode_fpt, res[0].barcode_fpt)
    with pytest.raises(ValueError, match="consensus_model must be specified"):
        sig_proc.detect_results_to_fpt_batch(mb, spc, drs)
    spc.segmentation.barcode_num_events = 25
    with pytest.raises(ValueError, match="use a tuple instead"):
        sig_proc.detect_results_to_fpt_batch(mb, spc, drs, consensus_query=consensus)


@pytest.mark.parametrize("seed", range(int(os.environ.get("WDX_SOAK_SEEDS", "12"))))   # soak: WDX_SOAK_SEEDS=200
def test_refinement_randomised_parameters_vs_oracle(seed):
    """Random queries, event counts, window widths, penalties, relaxations and normalisations: engine == oracle on
    every output of the refinement branch, whatever the status."""
    rng = np.random.default_rng(1000 + seed)
    nq = int(rng.integers(8, 97))
    query = rng.normal(0, 1, nq)
    E = int(rng.integers(max(nq + 10, 40), 128))
    E2 = int(rng.integers(5, 40))
    keep = int(rng.integers(1, E2 + 2))
    w = int(rng.choice([6, 12, 18, 24]))
    d = int(rng.integers(2, w // 2 + 2))
    n = 40
    rows = []
    for i in range(n):
        n_lead = int(rng.integers(0, 30))
        lv = list(rng.normal(0, 1, n_lead)) + list(query + rng.normal(0, 0.05, nq)) + list(rng.normal(0, 1, E2 + 12))
        lv = np.array(lv) * 12.0 + 85.0
        dw = rng.integers(2 * d + 1, 5 * d + 12, lv.size)
        x = np.repeat(lv, dw) + rng.normal(0, rng.uniform(0.5, 2.5), int(dw.sum()))
        rows.append(x.astype(np.float32)[:11000])
    stride = max(r.size for r in rows)
    mb = np.full((n, stride), np.nan, dtype=np.float32)
    for i, r in enumerate(rows):
        mb[i, : r.size] = r
    pad = int(rng.choice([0, 20, 100]))
    a_s = np.full(n, pad, dtype=np.int32)
    a_e = np.array([r.size - pad for r in rows], dtype=np.int32)
    seg = dict(padding=pad, min_obs_per_base=d, running_stat_width=w, num_events=E,
               seg_norm=str(rng.choice(["mean", "median"])), outlier_thresh=float(rng.choice([3.0, 5.0])))
    ref = dict(subseq_norm=str(rng.choice(["mean", "median", "none"])), penalty=float(rng.choice([0.0, 0.5, 1.5, 3.0])),
               psi=(int(rng.integers(0, 8)), 0, int(rng.integers(0, 60)), 0), ub_start=int(rng.integers(5, 60)),
               lb_end=int(rng.integers(0, nq)), ub_end=int(rng.integers(nq, 160)), barcode_segm_events=E2,
               barcode_keep_events=keep)
    fb = sig_proc.fingerprint_refine_batch(mb, a_s, a_e, sig_proc.SegParams(barcode_num_events=keep, **seg),
                                           sig_proc.RefineParams(query=query, **ref))
    fpt, dwell, stats, idx, status = orc.fingerprint_refine_batch(mb, a_s, a_e, orc.SegParams(barcode_num_events=keep, **seg),
                                                                  orc.RefineParams(query=query, **ref))
    assert np.array_equal(fb.status, status), (seed, fb.status.tolist(), status.tolist())
    good, rep = status == 0, (status == 0) | (status == 6)
    assert _same(fb.fpt[good], fpt[good]) and _same(fb.dwell[good], dwell[good]), seed
    assert _same(fb.stats[rep], stats[rep]) and _same(fb.refine_idx[rep], idx[rep]), seed


def test_refinement_behind_the_fast_kernels_matches_the_exact_kernel_and_the_oracle(golden_dir):
    """The tRNA flow at batch size: a fast kernel segments the adapter, fingerprint_refine_tail_kernel does the rest
    (statistics, subsequence match, the barcode's segmentation from the re-clipped tail).  Reads with a barcode tail
    beyond the tail kernel's capacity, NaN windows and failed detections travel on to the exact kernel.  Everything
    against the exact kernel alone (WDX_OPT_EXACT_PATH) and against the oracle, bit for bit."""
    from warpdemux_amd import _lib

    consensus = np.load(os.path.join(golden_dir, "g8_refine.npz"))["consensus"]
    rng = np.random.default_rng(77)
    n = 700
    rows = []
    for i in range(n):
        n_lead = int(rng.integers(2, 34))
        n_tail = 30 if i % 9 else int(rng.integers(70, 110))          # every ninth read: a tail of > 2048 samples
        emb = rng.random() > 0.1
        lv = list(rng.normal(0, 1, n_lead)) + list(consensus if emb else rng.normal(0, 1, consensus.size)) + list(rng.normal(0, 1, n_tail))
        lv = np.array(lv) * 12.0 + 85.0
        dw = rng.integers(12, 50, lv.size)
        x = (np.repeat(lv, dw) + rng.normal(0, rng.uniform(0.8, 2.5), int(dw.sum()))).astype(np.float32)[:8100]
        if i % 50 == 7:
            x = x[:int(rng.integers(1900, 2150))]     # a window too short for the configured width: the parameters shrink
                                                      # (sig_proc.py:526-533), the exact kernel keeps the read to itself
        rows.append(x)
    stride = max(r.size for r in rows)
    mb = np.full((n, stride), np.nan, dtype=np.float32)
    for i, r in enumerate(rows):
        mb[i, : r.size] = r
    a_s = np.full(n, 100, dtype=np.int32)
    a_e = np.array([r.size - 100 for r in rows], dtype=np.int32)
    ok = np.ones(n, dtype=np.uint8)
    ok[3] = 0
    mb[5, 1500:1503] = np.nan
    mb[6, 3000] = np.inf                              # (the clip bounds become NaN: every sample NaN, no hand-over)
    seg = dict(min_obs_per_base=9, running_stat_width=18, num_events=120)
    ref = dict(barcode_segm_events=25, barcode_keep_events=25)
    hp, hr = sig_proc.SegParams(barcode_num_events=25, **seg), sig_proc.RefineParams(query=consensus, **ref)
    fb = sig_proc.fingerprint_refine_batch(mb, a_s, a_e, hp, hr, success=ok)
    ctx = _lib.default_context(None)
    ctx.set_option(_lib.OPT_EXACT_PATH, 1)
    try:
        ex = sig_proc.fingerprint_refine_batch(mb, a_s, a_e, hp, hr, success=ok)
    finally:
        ctx.set_option(_lib.OPT_EXACT_PATH, 0)
    fpt, dwell, stats, idx, status = orc.fingerprint_refine_batch(
        mb, a_s, a_e, orc.SegParams(barcode_num_events=25, **seg), orc.RefineParams(query=consensus, **ref), ok=ok)
    for got in (fb, ex):
        assert np.array_equal(got.status, status), np.flatnonzero(got.status != status)
        good, rep = status == 0, (status == 0) | (status == 6)
        assert _same(got.fpt[good], fpt[good]) and _same(got.dwell[good], dwell[good])
        assert _same(got.stats[rep], stats[rep]) and _same(got.refine_idx[rep], idx[rep])
        assert np.isnan(got.fpt[~good]).all() and (got.dwell[~good] == 0).all()
        assert (got.refine_idx[~rep] == -1).all() and np.isnan(got.stats[~rep]).all()
    assert (status == 0).sum() > 250 and (status == 6).sum() > 50 and status[3] == 1


@pytest.mark.parametrize("nq,E,psi", [(5, 40, (0, 0, 0, 0)), (33, 64, (3, 0, 10, 0)), (84, 120, (5, 0, 40, 0)), (96, 125, (7, 0, 59, 0)), (96, 127, (7, 0, 59, 0)),
                                       (50, 126, (0, 0, 120, 0))])
def test_refinement_wave_match_kernel_on_ties_and_odd_shapes(nq, E, psi):
    """fingerprint_refine_match_wave_kernel away from the tRNA shape: query lengths that are no multiple of its four rows
    per lane (two or three reads per wave), series shorter and longer than a wave, relaxations from none to nearly the
    whole series -- on signals whose levels and noise sit on a coarse grid, so that event means repeat, the DP's costs tie
    exactly and the direction codes / the arg-min of the last row are decided by the first-minimum rules.  A batch large
    enough for the launch chain (fast kernels + refinement kernels) against the oracle, bit for bit."""
    rng = np.random.default_rng(nq * 1000 + E)
    query = np.round(rng.normal(0, 1, nq) * 2) / 2
    n = 2304
    rows = []
    for i in range(n):
        n_lead = int(rng.integers(0, max(1, E - nq - 8)))
        n_tail = 28
        body = query if i % 3 else np.round(rng.normal(0, 1, nq) * 2) / 2
        lv = np.concatenate([np.round(rng.normal(0, 1, n_lead) * 2) / 2, body, np.round(rng.normal(0, 1, n_tail) * 2) / 2]) * 12.0 + 85.0
        dw = rng.integers(20, 44, lv.size)
        noise = np.round(rng.normal(0, 1.5, int(dw.sum())) * 4) / 4
        rows.append((np.repeat(lv, dw) + noise).astype(np.float32)[:6000])
    stride = max(r.size for r in rows)
    mb = np.full((n, stride), np.nan, dtype=np.float32)
    for i, r in enumerate(rows):
        mb[i, : r.size] = r
    a_s = np.full(n, 50, dtype=np.int32)
    a_e = np.array([r.size - 50 for r in rows], dtype=np.int32)
    seg = dict(padding=50, min_obs_per_base=9, running_stat_width=18, num_events=E)
    ref = dict(subseq_norm="median" if nq % 2 else "mean", penalty=1.5, psi=psi, ub_start=E, lb_end=0, ub_end=E + 1,
               barcode_segm_events=20, barcode_keep_events=20)
    fb = sig_proc.fingerprint_refine_batch(mb, a_s, a_e, sig_proc.SegParams(barcode_num_events=20, **seg),
                                           sig_proc.RefineParams(query=query, **ref))
    fpt, dwell, stats, idx, status = orc.fingerprint_refine_batch(mb, a_s, a_e, orc.SegParams(barcode_num_events=20, **seg),
                                                                  orc.RefineParams(query=query, **ref))
    assert np.array_equal(fb.status, status), np.flatnonzero(fb.status != status)[:10]
    good, rep = status == 0, (status == 0) | (status == 6)
    assert good.sum() > n // 4, np.bincount(status, minlength=8)
    assert _same(fb.fpt[good], fpt[good]) and _same(fb.dwell[good], dwell[good])
    assert _same(fb.stats[rep], stats[rep]) and _same(fb.refine_idx[rep], idx[rep])


@pytest.mark.parametrize("w,d", [(12, 6), (18, 9), (30, 15)])
def test_refinement_tail_wave_kernel_long_noisy_tails_and_hand_backs(golden_dir, w, d):
    """fingerprint_refine_tail_wave_kernel at the edges of what it takes: barcode tails of 1 700 .. 2 048 samples (its
    capacity; a sample more and the exact kernel refines the read), tails of nearly white noise whose score curves have
    about as many local maxima as its peak list holds (400 .. 450 of 512), coarse samples whose clipped stretches make
    runs of equal scores across tile ends, and the three shipped window widths (12 and 18 unrolled, 30 the
    general loop).  Against the oracle, bit for bit."""
    consensus = np.load(os.path.join(golden_dir, "g8_refine.npz"))["consensus"]
    rng = np.random.default_rng(500 + w)
    n = 360
    rows = []
    for i in range(n):
        n_lead = int(rng.integers(2, 20))
        head = np.concatenate([rng.normal(0, 1, n_lead), consensus]) * 12.0 + 85.0
        dwh = rng.integers(2 * d + 2, 4 * d + 8, head.size)
        x = np.repeat(head, dwh) + rng.normal(0, 1.5, int(dwh.sum()))
        tail_len = int(rng.integers(1700, 2060))                      # around kTailCap
        kind = i % 3
        if kind == 0:      # ordinary events
            lv = rng.normal(0, 1, tail_len // (3 * d) + 2) * 12.0 + 85.0
            t = np.repeat(lv, 3 * d)[:tail_len] + rng.normal(0, 1.5, tail_len)
        elif kind == 1:    # nearly white noise: a local maximum every ~5 positions
            t = 85.0 + rng.normal(0, 6.0, tail_len)
        else:              # coarse, heavily clipped
            t = np.round((85.0 + rng.normal(0, 9.0, tail_len)) / 4.0) * 4.0
        rows.append(np.concatenate([x, t]).astype(np.float32))
    stride = max(r.size for r in rows)
    mb = np.full((n, stride), np.nan, dtype=np.float32)
    for i, r in enumerate(rows):
        mb[i, : r.size] = r
    a_s = np.zeros(n, dtype=np.int32)
    a_e = np.array([r.size for r in rows], dtype=np.int32)
    E = 120
    seg = dict(padding=0, min_obs_per_base=d, running_stat_width=w, num_events=E, outlier_thresh=3.0 if w == 18 else 5.0)
    ref = dict(barcode_segm_events=60, barcode_keep_events=40, ub_start=E, lb_end=0, ub_end=E + 1, psi=(5, 0, 60, 0))
    fb = sig_proc.fingerprint_refine_batch(mb, a_s, a_e, sig_proc.SegParams(barcode_num_events=40, **seg),
                                           sig_proc.RefineParams(query=consensus, **ref))
    fpt, dwell, stats, idx, status = orc.fingerprint_refine_batch(mb, a_s, a_e, orc.SegParams(barcode_num_events=40, **seg),
                                                                  orc.RefineParams(query=consensus, **ref))
    assert np.array_equal(fb.status, status), (np.flatnonzero(fb.status != status)[:10], np.bincount(status, minlength=8))
    good, rep = status == 0, (status == 0) | (status == 6)
    assert good.sum() > n // 5, np.bincount(status, minlength=8)
    assert _same(fb.fpt[good], fpt[good]) and _same(fb.dwell[good], dwell[good])
    assert _same(fb.stats[rep], stats[rep]) and _same(fb.refine_idx[rep], idx[rep])


def test_refinement_device_resident_entry_point(golden_dir):
    """wdx_fingerprint_refine_dev (DemuxEngine.fingerprint_refine): device tensors in and out, same bits as the host
    batch call, minibatch and packed layouts."""
    import torch

    from warpdemux_amd.engine import DemuxEngine

    consensus = np.load(os.path.join(golden_dir, "g8_refine.npz"))["consensus"]
    mb, a_s, a_e = _reads(consensus, 300, 23)
    seg = dict(min_obs_per_base=9, running_stat_width=18, num_events=120)
    hp = sig_proc.SegParams(barcode_num_events=25, **seg)
    hr = sig_proc.RefineParams(query=consensus, barcode_segm_events=25, barcode_keep_events=25)
    ref = sig_proc.fingerprint_refine_batch(mb, a_s, a_e, hp, hr)
    eng = DemuxEngine(np.zeros((4, 25)), 15, 0.1, hp)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    lens = (a_e + 100).astype(np.int64)
    for layout in ("minibatch", "packed"):
        if layout == "minibatch":
            out = eng.fingerprint_refine(d(mb), d(a_s), d(a_e), hr, stride=mb.shape[1], max_len=int(lens.max()))
        else:
            off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
            packed = np.concatenate([mb[i, :lens[i]] for i in range(mb.shape[0])])
            out = eng.fingerprint_refine(d(packed), d(a_s), d(a_e), hr, offsets=d(off), max_len=int(lens.max()))
        torch.cuda.synchronize()
        fpt, dwell, stats, idx, status = (t.cpu().numpy() for t in out)
        assert np.array_equal(status, ref.status) and _same(fpt, ref.fpt) and _same(dwell, ref.dwell)
        assert _same(stats, ref.stats) and np.array_equal(idx, ref.refine_idx)
    assert (ref.status == 0).sum() > 100
    eng.close()

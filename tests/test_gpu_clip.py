"""clip_bounds_kernel (A1 ahead of the fast fingerprint kernels, wdx_clip.hip) against the oracle's float32
median / MAD (pinned to the reference's own functions by fixture G5) and sig_proc.py:421-431's bounds, read by read,
on the edge cases of its wave-level radix select: window lengths around every group boundary, ties, constant and
two-valued windows, negative samples (the clamp shortcut and its two refusals), NaN / infinities, denormals.
Needs a real MI355X: run with `pytest -m gpu`."""
import ctypes as C

import numpy as np
import pytest

from oracle import wdx_oracle as orc
from warpdemux_amd import _lib, sig_proc

pytestmark = pytest.mark.gpu

REC = np.dtype([("lo", "<f4"), ("hi", "<f4"), ("cmax", "<f4"), ("flag", "<i4")])


def _run(rows, cap, params):
    import torch

    lens = np.array([r.size for r in rows], dtype=np.int64)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    packed = np.concatenate(rows).astype(np.float32) if rows else np.zeros(0, np.float32)
    n = len(rows)
    d_sig = torch.from_numpy(packed).cuda()
    d_off = torch.from_numpy(off).cuda()
    d_as = torch.zeros(n, dtype=torch.int32, device="cuda")
    d_ae = torch.from_numpy(lens.astype(np.int32)).cuda()
    d_rec = torch.full((n, 4), -1, dtype=torch.int32, device="cuda")
    ctx = _lib.default_context()
    pc = params.to_c()
    _lib.check(_lib.load().wdx_selftest_clip_dev(ctx.handle, C.c_void_p(d_sig.data_ptr()), C.c_void_p(d_off.data_ptr()), 0, n,
                                                 C.c_void_p(d_as.data_ptr()), C.c_void_p(d_ae.data_ptr()), C.byref(pc), cap,
                                                 C.c_void_p(d_rec.data_ptr()), None))
    torch.cuda.synchronize()
    return d_rec.cpu().numpy().view(REC).reshape(n)


def _expected(x, params):
    """(lo, hi, cmax, gate) of sig_proc.py:421-431 in the arithmetic the engine is configured for."""
    med, mad = orc.nanmedian_mad_f32(x)
    if params.clip_bounds == "float64":
        tm = np.float64(params.outlier_thresh) * np.float64(mad)
        lo, hi = np.float32(np.float64(med) - tm), np.float32(np.float64(med) + tm)
    else:
        tm = np.float32(params.outlier_thresh) * mad
        lo, hi = np.float32(med - tm), np.float32(med + tm)
    cmin, cmax = np.clip(x.min(), lo, hi), np.clip(x.max(), lo, hi)
    gate = bool(lo <= hi and cmin > 0 and cmax < 3.0e38)
    if gate:
        fa = max(int(np.float32(cmin).view(np.uint32)) >> 23, 1)
        fb = int(np.float32(cmax).view(np.uint32)) >> 23
        gate = (fb - fa) + int(x.size).bit_length() <= 28
    return lo, hi, np.float32(cmax), gate


def _rows(rng, cap=6144):
    rows, tags = [], []

    def add(x, tag):
        rows.append(np.asarray(x, dtype=np.float32))
        tags.append(tag)

    for ln in [256, 257, 258, 259, 260, 263, 511, 512, 513, 767, 768, 1000, 1023, 1024, 1025, 2047, 2048, 2049, 3000,
               4095, 4096, 4097, 4100, 4863, 4864, 4865, 5000, 5117, 5118, 5119, 5120, 5121, 5633, 6143, 6144, 6145, 255, 100] + (
               [] if cap <= 6144 else [7000, 8191, 8192, 8193, 9999, 12287, 12288, 12289, 13056, 13057, 13311, 13312, 13313, 16384]):
        for rep in range(2):
            add(rng.normal(80, 15, ln) + rng.normal(0, 2, ln), "normal")
    for ln in [300, 1024, 4097, 5120] + ([] if cap <= 6144 else [8192, cap - 3]):
        add(np.full(ln, 77.25), "constant")
        add(np.where(rng.random(ln) < 0.5, 70.0, 90.0), "two-valued")
        add(np.round(rng.normal(80, 15, ln) * 4) / 4, "quantised")           # heavy ties: bins with > 64 members
        add(np.round(rng.normal(80, 3, ln)), "coarse")                      # a dozen distinct values
        x = np.full(ln, 81.5)
        x[ln // 3] = 12.0
        add(x, "all-but-one")
        x = rng.normal(80, 15, ln)
        x[rng.integers(0, ln, 5)] = -rng.uniform(1, 60, 5)                   # flicker spikes below zero: clamp applies
        add(x, "few-negative")
        x = rng.normal(80, 15, ln)
        x[rng.random(ln) < 0.6] *= -1                                        # most samples negative: the shortcut refuses
        add(x, "mostly-negative")
        add(-np.abs(rng.normal(80, 15, ln)), "all-negative")
        x = rng.normal(80, 15, ln)
        x[::7] = -0.0
        add(x, "minus-zero")
        x = rng.normal(30, 25, ln)                                           # a third negative, MAD comparable to med - min+
        add(x, "wide")
        for bad in (np.nan, np.inf, -np.inf, -np.nan):
            x = rng.normal(80, 15, ln)
            x[ln // 2] = bad
            add(x, "nonfinite")
        add(rng.normal(80, 15, ln) * 1e-42, "denormal")
        add(np.abs(rng.normal(80, 15, ln)) * 1e33, "huge")
        add(np.abs(rng.normal(0, 1, ln)) * 1e-6, "tiny")
        add(np.zeros(ln), "zeros")
    return rows, tags


@pytest.mark.parametrize("cap", [4096, 5120, 6144, 8192, 13312])   # (the last two: the long-window lists' instantiations)
@pytest.mark.parametrize("bounds", ["float32", "float64"])
def test_clip_bounds_kernel_matches_median_mad_bounds_read_by_read(cap, bounds):
    rng = np.random.default_rng(20240 + cap)
    rows, tags = _rows(rng, cap)
    params = sig_proc.SegParams(padding=0, outlier_thresh=(np.float64(3.3) if bounds == "float64" else 5.0), clip_bounds=bounds)
    rec = _run(rows, cap, params)
    seen = {}
    for x, tag, r in zip(rows, tags, rec):
        taken = 256 <= x.size <= cap
        if not taken:
            assert r["flag"] == 0, (tag, x.size, r)
            continue
        if not np.isfinite(x).all():
            assert r["flag"] == 2, (tag, x.size, r)
            continue
        if not (x >= 0).any():                           # no non-negative sample to clamp to: left to the exact kernel
            assert r["flag"] in (2, 3), (tag, x.size, r)
            continue
        assert r["flag"] in (1, 3), (tag, x.size, r)
        seen[(tag, int(r["flag"]))] = seen.get((tag, int(r["flag"])), 0) + 1
        if r["flag"] != 1:
            continue
        lo, hi, cmax, gate = _expected(x, params)
        assert gate, (tag, x.size)                       # the kernel never passes a window the gate refuses
        assert r["lo"].view(np.uint32) == lo.view(np.uint32) and r["hi"].view(np.uint32) == hi.view(np.uint32), (tag, x.size, r, lo, hi)
        assert r["cmax"].view(np.uint32) == cmax.view(np.uint32), (tag, x.size)
    # the cases are what they claim to be: plain windows and windows with a few negative spikes are taken (flag 1),
    # windows the shortcut for negative samples cannot serve are refused (flag 3), never answered wrongly
    assert seen.get(("normal", 3), 0) == 0 and seen.get(("few-negative", 3), 0) == 0
    assert seen.get(("quantised", 3), 0) == 0 and seen.get(("coarse", 3), 0) == 0 and seen.get(("two-valued", 3), 0) == 0
    assert seen.get(("mostly-negative", 1), 0) == 0
    assert seen.get(("zeros", 1), 0) == 0


@pytest.mark.parametrize("cap", [6144, 8192, 13312])
def test_clip_bounds_kernel_many_random_reads_and_gate_agreement(cap):
    """2 000 random windows (lengths 256..cap, random offsets / scales / spike rates): wherever the kernel answers, the
    bounds are the reference's bit for bit; wherever the gate holds and no sample is negative, it answers."""
    rng = np.random.default_rng(77 + cap)
    rows = []
    for i in range(2000):
        ln = int(rng.integers(256 if cap == 6144 else 5000, cap + 1))
        x = rng.normal(rng.uniform(20, 200), rng.uniform(0.5, 30), ln) + rng.normal(0, rng.uniform(0.01, 3), ln)
        k = rng.poisson(ln * rng.choice([0, 0.001, 0.01]))
        if k:
            x[rng.integers(0, ln, k)] += rng.choice([-1, 1], k) * rng.uniform(20, 120, k)
        if i % 5 == 0:
            x = np.round(x / 0.1755) * 0.1755     # ADC quantum
        rows.append(x.astype(np.float32))
    params = sig_proc.SegParams(padding=0)
    rec = _run(rows, cap, params)
    answered = refused_gate_ok = gate_ok = 0
    for x, r in zip(rows, rec):
        assert r["flag"] in (1, 3)
        lo, hi, cmax, gate = _expected(x, params)
        gate_ok += gate
        if r["flag"] == 1:
            answered += 1
            assert gate
            assert r["lo"].view(np.uint32) == lo.view(np.uint32) and r["hi"].view(np.uint32) == hi.view(np.uint32)
            assert r["cmax"].view(np.uint32) == cmax.view(np.uint32)
        elif (x >= 0).all():
            assert not gate
        elif gate:
            refused_gate_ok += 1      # negative samples and the clamp shortcut's conditions do not hold
    print("answered", answered, "gate holds", gate_ok, "refused although the gate holds", refused_gate_ok)
    assert answered > 1000 and refused_gate_ok <= 0.02 * gate_ok

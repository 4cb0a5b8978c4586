"""Deterministic synthetic RNA004-like adapter signals.

Spec "wdx-synth v1" -- integer-hash driven so that the NumPy implementation here and the HIP
generator kernel (csrc/wdx_synth.hip) produce bit-identical float32 samples:

* ``h(seed, read, stream, ctr)`` = splitmix64 finaliser over a 64-bit combination of its inputs.
* per read ``r`` (GLOBAL read index, so any sharding regenerates the same reads):
    barcode  b      = h(seed, r, 0, 0) mod n_barcodes
    n_events n_ev   = 125 + h(seed, r, 0, 1) mod 16
    dwell[e]        = DWELL_TABLE[h(seed, r, 1, e) & 1023]      (6 + geometric-like, mean ~34.7)
    level[e]        = barcode template for the last ``n_bc_events`` events, shared leader template
                      before that (templates are float32 tables, ``80 + 15*N(0,1)`` pA)
    sample[t]       = fl32(level + fl32(isum * NOISE_SCALE)),  isum = sum of four 16-bit chunks of
                      h(seed, r, 2, t) minus 131070 (Irwin-Hall ~ N(0, sigma=2 pA))
    flicker         : if h(seed, r, 3, t) mod 1000 == 0: sample += +-60 pA (sign = bit 32)
    layout          : [100 samples pad @95 pA][events...][100 samples pad @105 pA]
    adapter_start   = 100, adapter_end = len - 100  (so sig_extract.padding = 100 recovers the row)

This module is host-side plumbing for tests and the bench; it contains no reference code.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

MASK64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_C1 = np.uint64(0x9E3779B97F4A7C15)
_C2 = np.uint64(0xBF58476D1CE4E5B9)
_C3 = np.uint64(0x94D049BB133111EB)

BASE_SEED = 0x57445800
PAD = 100
PRE_LEVEL = np.float32(95.0)
POST_LEVEL = np.float32(105.0)
NOISE_SIGMA = 2.0
# sd of the sum of four independent U{0..65535}: sqrt(4 * (65536^2 - 1) / 12)
_IH_SD = float(np.sqrt(4.0 * (65536.0**2 - 1.0) / 12.0))
SPIKE = np.float32(60.0)
MAX_EVENTS = 140
MIN_EVENTS = 125
N_LEAD = 160  # leader template length (>= MAX_EVENTS)
N_BC_TEMPLATE = 64


def hash64(seed, read, stream, ctr):
    """Vectorised splitmix64-style hash; all arguments broadcast as uint64."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + _C1 * (np.asarray(read, dtype=np.uint64) + np.uint64(1))
        z = z ^ (np.asarray(stream, dtype=np.uint64) * _C2)
        z = z + np.asarray(ctr, dtype=np.uint64) * _C3
        z = (z ^ (z >> np.uint64(30))) * _C2
        z = (z ^ (z >> np.uint64(27))) * _C3
        z = z ^ (z >> np.uint64(31))
    return z


def dwell_table(scale: float = 1.0) -> np.ndarray:
    """1024-entry inverse-CDF table: 6 + geometric(mean 28.7), capped at 400 (int32) -- RNA004 at 130 bases/s.
    ``scale`` stretches all three numbers: 2.5 gives the RNA002-like table 15 + geometric(mean 71.75), capped at 1000
    (ASSUMPTION: the older chemistry's adapter windows are ~2.5x longer in samples -- its config's max_obs_trace,
    min_obs_per_base and running_stat_width are 2.5x RNA004's, DEPRECATED/config_files/rna002_70bps@v0.4.4.toml:2-12)."""
    q = (np.arange(1024, dtype=np.float64) + 0.5) / 1024.0
    p = 1.0 / (28.7 * scale)
    g = np.floor(np.log1p(-q) / np.log1p(-p))
    return (int(round(6 * scale)) + np.minimum(g, int(round(394 * scale)))).astype(np.int32)


@dataclass
class SynthSpec:
    n_barcodes: int = 10
    seed: int = BASE_SEED
    n_bc_events: int = 36
    noise_sigma: float = NOISE_SIGMA
    spikes: bool = True
    dwell_scale: float = 1.0   # 2.5: RNA002-like dwell times (mean window ~11.6 k samples; see dwell_table)

    def tables(self):
        """(lead[N_LEAD], bc[n_barcodes, N_BC_TEMPLATE], dwell_table[1024]) -- float32/int32."""
        rng = np.random.Generator(np.random.PCG64(self.seed ^ 0x7E3A))
        lead = (80.0 + 15.0 * rng.standard_normal(N_LEAD)).astype(np.float32)
        bc = (80.0 + 15.0 * rng.standard_normal((self.n_barcodes, N_BC_TEMPLATE))).astype(np.float32)
        return lead, bc, dwell_table(self.dwell_scale)

    @property
    def noise_scale(self) -> np.float32:
        return np.float32(self.noise_sigma / _IH_SD)


def read_layout(spec: SynthSpec, read_ids: np.ndarray):
    """barcode[n], n_ev[n], dwell[n, MAX_EVENTS] (0 beyond n_ev), length[n] (incl. both pads)."""
    read_ids = np.asarray(read_ids, dtype=np.uint64)
    _, _, dt = spec.tables()
    bc = (hash64(spec.seed, read_ids, 0, 0) % np.uint64(spec.n_barcodes)).astype(np.int32)
    n_ev = (MIN_EVENTS + (hash64(spec.seed, read_ids, 0, 1) % np.uint64(16))).astype(np.int32)
    e = np.arange(MAX_EVENTS, dtype=np.uint64)[None, :]
    dw = dt[(hash64(spec.seed, read_ids[:, None], 1, e) & np.uint64(1023)).astype(np.int64)]
    dw = np.where(e.astype(np.int64) < n_ev[:, None], dw, 0).astype(np.int32)
    length = dw.sum(axis=1).astype(np.int64) + 2 * PAD
    return bc, n_ev, dw, length


def generate_read(spec: SynthSpec, read_id: int) -> tuple[np.ndarray, int]:
    """One read: (float32 signal incl. pads, barcode id)."""
    lead, bct, _ = spec.tables()
    bc, n_ev, dw, length = read_layout(spec, np.array([read_id]))
    bc, n_ev, dw, length = int(bc[0]), int(n_ev[0]), dw[0], int(length[0])
    level = np.empty(length, dtype=np.float32)
    level[:PAD] = PRE_LEVEL
    level[length - PAD :] = POST_LEVEL
    pos = PAD
    for e in range(n_ev):
        k = n_ev - 1 - e  # event index counted from the 3' end
        lv = bct[bc, k] if k < spec.n_bc_events else lead[k]
        level[pos : pos + dw[e]] = lv
        pos += dw[e]
    t = np.arange(length, dtype=np.uint64)
    hn = hash64(spec.seed, read_id, 2, t)
    isum = (
        (hn & np.uint64(0xFFFF)).astype(np.int64)
        + ((hn >> np.uint64(16)) & np.uint64(0xFFFF)).astype(np.int64)
        + ((hn >> np.uint64(32)) & np.uint64(0xFFFF)).astype(np.int64)
        + ((hn >> np.uint64(48)) & np.uint64(0xFFFF)).astype(np.int64)
        - 131070
    )
    noise = isum.astype(np.float32) * spec.noise_scale
    sig = (level + noise).astype(np.float32)
    if spec.spikes:
        hs = hash64(spec.seed, read_id, 3, t)
        is_spike = (hs % np.uint64(1000)) == 0
        sign = np.where(((hs >> np.uint64(32)) & np.uint64(1)) == 1, SPIKE, -SPIKE).astype(np.float32)
        sig = np.where(is_spike, (sig + sign).astype(np.float32), sig)
    return sig, bc


def generate_packed(spec: SynthSpec, first_read: int, n_reads: int):
    """Packed batch: (sig f32[total], offsets i64[n+1], a_start i32[n], a_end i32[n], barcode i32[n])."""
    ids = np.arange(first_read, first_read + n_reads, dtype=np.uint64)
    bc, _, _, length = read_layout(spec, ids)
    off = np.zeros(n_reads + 1, dtype=np.int64)
    np.cumsum(length, out=off[1:])
    sig = np.empty(int(off[-1]), dtype=np.float32)
    for i in range(n_reads):
        s, _ = generate_read(spec, first_read + i)
        sig[off[i] : off[i + 1]] = s
    a_start = np.full(n_reads, PAD, dtype=np.int32)
    a_end = (length - PAD).astype(np.int32)
    return sig, off, a_start, a_end, bc


def generate_minibatch(spec: SynthSpec, first_read: int, n_reads: int, stride: int, start_jitter: int = 0):
    """file_proc-style minibatch (file_proc.py:244-260): (n, stride) float32, NaN tail; rows longer
    than ``stride`` are truncated like sig_preload_size truncates real reads.

    ``start_jitter`` > 0: the rows carry WHOLE reads the way file_proc's do -- read r's adapter is preceded by
    J_r = h(seed, r, 4, 0) mod (start_jitter + 1) samples of pre-adapter signal (open-pore-like, 95 pA with the
    generator's noise statistics), so adapter_start = 100 + J_r varies per read (sig_proc.py:382-391 slices it
    out); start_jitter = 2900 gives adapter_start ~ U{100 .. 3000}.  The adapter window itself is unchanged."""
    sig, off, a_start, a_end, bc = generate_packed(spec, first_read, n_reads)
    mb = np.full((n_reads, stride), np.nan, dtype=np.float32)
    if start_jitter <= 0:
        for i in range(n_reads):
            row = sig[off[i] : off[i + 1]][:stride]
            mb[i, : row.size] = row
        return mb, a_start, a_end, bc
    rid = np.arange(first_read, first_read + n_reads, dtype=np.uint64)
    J = (hash64(spec.seed, rid, 4, 0) % np.uint64(start_jitter + 1)).astype(np.int64)
    a_start, a_end = a_start.copy(), a_end.copy()
    for i in range(n_reads):
        j = int(J[i])
        t = np.arange(j, dtype=np.uint64)
        h = hash64(spec.seed, rid[i], 5, t)
        isum = ((h & np.uint64(0xFFFF)) + ((h >> np.uint64(16)) & np.uint64(0xFFFF)) + ((h >> np.uint64(32)) & np.uint64(0xFFFF)) +
                (h >> np.uint64(48))).astype(np.float32) - np.float32(131070.0)
        pre = (PRE_LEVEL + isum * spec.noise_scale).astype(np.float32)
        row = np.concatenate([pre, sig[off[i] : off[i + 1]]])[:stride]
        mb[i, : row.size] = row
        a_start[i] += j
        a_end[i] += j
    return mb, a_start, a_end, bc

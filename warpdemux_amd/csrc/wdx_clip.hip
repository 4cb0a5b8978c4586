// Clip bounds ahead of the fast fingerprint kernels: one WAVE per read (gfx950).
//
// clip_bounds_kernel: A1 of the hot path (sig_proc.py:421-431: med = nanmedian(x), mad = nanmedian(|x - med|),
// bounds med -/+ thresh * mad) for the reads of a large batch, BEFORE the launch chain of the fast kernels, which then
// start at the clip (fast_body<..., EXT = true>): their two workgroup-wide radix selects were 26 % of the main
// kernel's wave-cycles for 2.9 k of its 10.5 k VALU instructions per read, with ~18 of its ~50 barriers.
//
// One wave owns one read and keeps ALL its samples in registers (NPL per lane: the 18 KB are read from HBM once more
// -- the chain used 9 % of the HBM rate -- and never touch LDS); LDS holds only the wave's 2048-bin histogram (8.5 KB
// per wave; one wave = one read = one workgroup, see kClipWaves).  There is no barrier anywhere: what one lane writes
// to LDS the others read in program order (DS operations of a wave execute in issue order).
//   * extremes of the raw bit patterns (non-negative samples: the pattern orders like the value and is the key;
//     negative samples -- outliers by construction of the statistic -- are clamped to the smallest non-negative one,
//     which leaves both medians unchanged under two conditions that are checked); an infinity or a NaN ->
//     CLIP_NAN_NEG, the read goes to the exact general kernel
//   * exact order statistics by MSD radix select with 11-bit digits (first level of |x - med|: the VALUE binned
//     linearly over [0, dmax], as in fast_select); the bin holding rank k is found from the histogram by a transposing
//     reduction (v_permlane32_swap / v_permlane16_swap / row rotations: lane l ends up with the total of the 128-bin
//     chunk l >> 2) + two wave scans; its <= 64 members are gathered and ranked through v_readlane
//   * lanes past the window hold copies of the largest key (ranks below n are unaffected)
// Output per read: ClipRec {lo, hi, clipped maximum, flag} -- the same float32 bounds, bit for bit, as the in-kernel
// path (fast_median_lds) and the exact kernel compute; the exactness gate of the event-mean sums rides along.

#include "wdx_fp_types.h"
#include "wdx_wave.h"

#include <algorithm>
#include <type_traits>

namespace wdx {

typedef float v2f __attribute__((ext_vector_type(2)));

// (one read = one wave = one workgroup: with four reads per workgroup a finished wave's registers and LDS waited for the slowest
// of the four -- reads differ by their > 64-member levels -- and the kernel ran 45.1 instead of 38.9 ms per 10 M reads)
constexpr int kClipWaves = 1;                       // reads per workgroup
constexpr int kClipWaveWords = kHB + 64;            // histogram | member list

struct ClipArgs {
    FpArgs a;
    ClipRec *rec;
    int cap;  // windows of 256 .. cap samples are taken (cap <= 64 * NPL)
    const unsigned *in_count;  // list form: the first `grid` entries of a device-side read list (else null)
    const int32_t *in_list;
    int defer;  // 1: a window this kernel cannot decide (negative samples it may not clamp, NaN, infinities) keeps CLIP_NONE --
                // clip_bounds_block_kernel, which takes any sign, comes after it (the long-window lists)
};

// xor-exchange inside a row of 16 / a quad (VALU only)
__device__ __forceinline__ unsigned clip_xor8(unsigned x) { return WDX_DPP(0u, x, 0x128, 0xf); }   // row_ror:8
__device__ __forceinline__ unsigned clip_xor2(unsigned x) { return WDX_DPP(0u, x, 0x4e, 0xf); }    // quad_perm [2,3,0,1]
__device__ __forceinline__ unsigned clip_xor1(unsigned x) { return WDX_DPP(0u, x, 0xb1, 0xf); }    // quad_perm [1,0,3,2]
__device__ __forceinline__ unsigned clip_xor4(unsigned x) {
    return (unsigned)__builtin_amdgcn_ds_swizzle((int)x, 0x101f);  // bit mode: and 0x1f, xor 4
}
__device__ __forceinline__ void clip_wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <int NPL>
__device__ __forceinline__ void clip_wave(const FpArgs &A, ClipRec *recs, const int64_t r, const int cap, unsigned *lds,
                                          const bool defer = false) {
    static_assert(NPL % 4 == 0 && NPL * 64 <= 16384, "samples per lane");
    const int lane = threadIdx.x & 63;
    const wdx_seg_params &P = A.p;
    unsigned *hist = lds;          // kHB words
    unsigned *wl = lds + kHB;      // 64 words

    // A0 extract_adapter (as fast_body)
    const int64_t row_off = A.row_off ? A.row_off[r] : r * A.stride;
    const int64_t row_len = A.row_len ? (int64_t)A.row_len[r] : (A.row_off ? A.row_off[r + 1] - A.row_off[r] : A.stride);
    int64_t start = (int64_t)A.a_start[r] - P.padding;
    if (start < 0) start = 0;
    int64_t stop = (int64_t)A.a_end[r] + P.padding;
    if (stop > row_len) stop = row_len;
    const int64_t n64 = stop - start;
    ClipRec out{0.0f, 0.0f, 0.0f, CLIP_NONE};
    if ((A.ok && !A.ok[r]) || n64 < 256 || n64 > (int64_t)cap) {
        if (lane == 0) recs[r] = out;
        return;
    }
    const int n = (int)n64;
    constexpr int NG = NPL / 4;
    const int ng = (n + 255) >> 8;  // groups of 256 samples that hold data (wave-uniform, >= 1)
    // diagnostic builds of the chain (wdx_fingerprint_profile_dev): shader-clock stamps in slots 26..31 of the read's row
    // (26 start, 27 loaded, 28 median: histogram, 29 bin found, 30 median done, 31 keys of |x - med|, 23 MAD: histogram,
    // 24 MAD done)
    long long *pp = (A.prof && r < A.prof_reads) ? A.prof + r * 32 : nullptr;
    auto stamp = [&](int k) __attribute__((always_inline)) {
        if (pp) {
            const long long t = (long long)__builtin_amdgcn_s_memtime();
            if (lane == 0) pp[k] = t;
        }
    };
    stamp(26);

    // HBM -> registers: four consecutive samples per lane and load (dword-aligned 16-byte loads, clamped to the last
    // whole group: a clamped lane holds samples of the window again, which cannot move the extremes)
    unsigned u[NPL];
    {
        struct __attribute__((packed, aligned(4))) F4U { unsigned x, y, z, w; };
        const unsigned *__restrict__ src = reinterpret_cast<const unsigned *>(A.sig + row_off + start);
        const unsigned last4 = (unsigned)(n - 4);
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            u[4 * j] = u[4 * j + 1] = u[4 * j + 2] = u[4 * j + 3] = 0u;
            if (j < ng) {
                const unsigned i = (unsigned)(j * 64 + lane) * 4u;
                const F4U v = *reinterpret_cast<const F4U *>(src + min(i, last4));
                u[4 * j] = v.x; u[4 * j + 1] = v.y; u[4 * j + 2] = v.z; u[4 * j + 3] = v.w;
            }
        }
    }
    unsigned umn = 0xffffffffu, umx = 0u;
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        if (j < ng) {
            umn = min(min(umn, u[4 * j]), u[4 * j + 1]);
            umn = min(min(umn, u[4 * j + 2]), u[4 * j + 3]);
            umx = max(max(umx, u[4 * j]), u[4 * j + 1]);
            umx = max(max(umx, u[4 * j + 2]), u[4 * j + 3]);
        }
    }
    umn = wave_min_u32(umn);
    umx = wave_max_u32(umx);
    stamp(27);
    // Non-negative samples (the usual case): the raw pattern orders like the value and IS the key; umn / umx are the
    // extremes.  A pattern beyond +inf's is a negative sample (flicker spikes below zero: ~15 % of the synthetic reads
    // have one), -0.0 or a NaN.  Negative samples are CLAMPED to the smallest non-negative sample -- one v_max_i32 each,
    // no re-keying: they are the smallest elements before and after, so every order statistic above them is unchanged
    // (checked below: the median must lie above the clamped value), and their keys |x - med| stay at or above
    // med - min+ (checked: the MAD must lie below that), so both medians are those of the original samples.
    const bool neg = umx > 0x7f800000u;  // wave-uniform
    float xmin_true = __uint_as_float(umn);
    if (neg) {
        // the most negative sample (or a NaN with the sign bit set) is the largest pattern; the largest non-negative
        // sample (or a NaN without it) the largest pattern as a signed integer
        int smx = 0;
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            if (j < ng) {
#pragma unroll
                for (int q = 0; q < 4; ++q) smx = max(smx, (int)u[4 * j + q]);
            }
        }
        smx = (int)wave_max_u32((unsigned)smx);  // (non-negative integers)
        // -inf / NaN, +inf / NaN, or not one finite non-negative sample -> the exact general kernel
        if (umx >= 0xff800000u || (unsigned)smx >= 0x7f800000u || umn >= 0x7f800000u) {
            out.flag = defer ? CLIP_NONE : CLIP_NAN_NEG;
            if (lane == 0) recs[r] = out;
            return;
        }
        xmin_true = __uint_as_float(umx);
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            if (j < ng) {
#pragma unroll
                for (int q = 0; q < 4; ++q) u[4 * j + q] = (unsigned)max((int)u[4 * j + q], (int)umn);
            }
        }
        umx = (unsigned)smx;
    } else if (umx >= 0x7f800000u) {  // +inf
        out.flag = defer ? CLIP_NONE : CLIP_NAN_NEG;
        if (lane == 0) recs[r] = out;
        return;
    }
    // lanes of the last group that lie past the window: valid[q] <=> the sample they hold is theirs
    const unsigned last4 = (unsigned)(n - 4);
    const unsigned i_last = (unsigned)((ng - 1) * 64 + lane) * 4u;  // this lane's first index in the last group
    auto pad_last_group = [&](unsigned padkey) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            if (j == ng - 1) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const bool valid = min(i_last, last4) + (unsigned)q >= i_last;
                    u[4 * j + q] = valid ? u[4 * j + q] : padkey;
                }
            }
        }
    };
    pad_last_group(umx);

    // every element of the groups that hold data, highest group first: ONE dispatch on ng (a compare tree) and then
    // straight fall-through, instead of a test per group and pass
    // every group that holds data (a uniform test per group; a switch on ng with fall-through was no faster)
    // (an opaque copy of ng per pass: the group tests are one scalar compare each -- as common subexpressions of all
    // passes the compiler keeps 20 lane masks alive for the whole kernel and spills them through v_writelane / v_readlane)
    auto each4 = [&](auto &&f4) __attribute__((always_inline)) {
        int ngp = ng;
        asm volatile("" : "+s"(ngp));
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            if (j < ngp) f4(u[4 * j], u[4 * j + 1], u[4 * j + 2], u[4 * j + 3]);
        }
    };
    auto each = [&](auto &&f) __attribute__((always_inline)) {
        each4([&](unsigned x0, unsigned x1, unsigned x2, unsigned x3) __attribute__((always_inline)) {
            f(x0);
            f(x1);
            f(x2);
            f(x3);
        });
    };

    // rank k (0-based) of the keys, all inside [a, b]; want_hi: also rank k + 1 (khi).  m = number of keys.
    // One LEVEL = histogram of a digit (MODE 0: the first level of the median, every register is a key and a member;
    // MODE 1: the first level of the MAD -- the registers still hold the samples, the key |x - med| is made on the fly
    // and its VALUE is binned linearly; MODE 2: a later level, the registers hold keys and the members are those
    // inside [a, b]), the bin that holds rank k, and -- with at most 64 members -- their gather and ranking.  The first
    // level is straight-line code and only the rare later levels sit in a loop: inside a loop every expression of the
    // registers alone (the linear bins, 64 * NPL of them) is loop-invariant, gets hoisted and spills.
    unsigned sk, sa, sb;           // state of the select: rank, key range
    unsigned klo = 0, khi = 0;
    bool in_hi = false, done = false;
    float med = 0.0f, lin_scale = 0.0f;  // MAD
    // MAD: the registers hold the samples throughout (never rewritten inside a branch: in-place changes on one side of
    // a branch made the register allocator shuffle and spill all 64 * NPL of them at the join); a key is made where
    // it is used.  (The empty asm keeps the SLP vectoriser from pairing the subtractions into v_pk_add_f32.)
    auto level = [&](auto mode_t, auto mad_t) __attribute__((always_inline)) {
        constexpr int MODE = decltype(mode_t)::value;
        constexpr bool MAD = decltype(mad_t)::value;
        const unsigned a = sa, b = sb, k = sk;
        const unsigned range = b - a;
        // MODE 0 / 2: bin = (key >> shift) - (a >> shift), at most 2048 of them (the subtraction lives in the
        // histogram's base address: two VALU operations per key in front of the ds_add)
        int shift = 32 - __clz((int)range) - kHBits;  // (range > 0)
        if (shift < 0) shift = 0;
        if (((b >> shift) - (a >> shift)) >= (unsigned)kHB) ++shift;
        const unsigned abin = a >> shift;
        // (opaque copies of the MAD operands per pass: the compiler otherwise keeps the 64 * NPL keys / bins of one
        // pass alive for the next, one more register per sample)
        auto opaque = [](float v) __attribute__((always_inline)) -> float {
            asm volatile("" : "+v"(v));
            return v;
        };
        // (wave priorities: the histogram passes are throughput work -- 80 independent keys --, finding the bin, the gather and
        // the ranking are dependent chains and go first among the SIMD's waves: 38.9 -> 38.5 ms per 10 M reads)
        __builtin_amdgcn_s_setprio(0);
        {
            uint4 *h4 = reinterpret_cast<uint4 *>(hist);
#pragma unroll
            for (int q = 0; q < kHB / 256; ++q) h4[q * 64 + lane] = make_uint4(0, 0, 0, 0);
        }
        clip_wave_fence();
        if constexpr (MODE == 1) {
            // (x - med) * scale two samples at a time: v_pk_add_f32 / v_pk_mul_f32 on the register pairs the 16-byte
            // loads filled (same roundings as the scalar operations: -ffp-contract=off)
            const float m1 = opaque(med), s1 = opaque(lin_scale);
            const v2f m1v = {m1, m1}, s1v = {s1, s1};
            each4([&](unsigned x0, unsigned x1, unsigned x2, unsigned x3) __attribute__((always_inline)) {
                const v2f pa = (v2f{__uint_as_float(x0), __uint_as_float(x1)} - m1v) * s1v;
                const v2f pb = (v2f{__uint_as_float(x2), __uint_as_float(x3)} - m1v) * s1v;
                atomicAdd(&hist[(unsigned)fabsf(pa.x)], 1u);
                atomicAdd(&hist[(unsigned)fabsf(pa.y)], 1u);
                atomicAdd(&hist[(unsigned)fabsf(pb.x)], 1u);
                atomicAdd(&hist[(unsigned)fabsf(pb.y)], 1u);
            });
        } else if constexpr (MODE == 0) {
            unsigned *hb = hist - abin;
            each([&](unsigned key) { atomicAdd(&hb[key >> shift], 1u); });
        } else {
            unsigned *hb = hist - abin;
            const float m1 = opaque(med);
            each([&](unsigned x) {
                unsigned key = x;
                if constexpr (MAD) {
                    float d = __uint_as_float(x) - m1;
                    asm("" : "+v"(d));
                    key = __float_as_uint(fabsf(d));
                }
                if (key - a <= range) atomicAdd(&hb[key >> shift], 1u);
            });
        }
        clip_wave_fence();
        __builtin_amdgcn_s_setprio(1);
        if constexpr (MODE == 0) stamp(28);
        if constexpr (MODE == 1) stamp(23);
        // the bin that holds rank k: lane l reads the bin pairs (q * 128 + 2 l, + 1), q = 0 .. 15 (conflict-free
        // 8-byte reads); a transposing reduction leaves the total of chunk l >> 2 (128 bins) in lane l
        unsigned B, cnt, kin;
        {
            const uint2 *h2 = reinterpret_cast<const uint2 *>(hist);
            unsigned s[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const uint2 v = h2[q * 64 + lane];
                s[q] = v.x + v.y;
            }
            // lanes 0..31 keep chunk q, lanes 32..63 chunk q + 8
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const auto sw = __builtin_amdgcn_permlane32_swap(s[q], s[q + 8], false, false);
                s[q] = sw[0] + sw[1];
            }
            // rows of 16: row 0 chunk q, row 1 chunk q + 4, row 2 chunk q + 8, row 3 chunk q + 12
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const auto sw = __builtin_amdgcn_permlane16_swap(s[q], s[q + 4], false, false);
                s[q] = sw[0] + sw[1];
            }
            const bool b3 = (lane & 8) != 0, b2 = (lane & 4) != 0;
#pragma unroll
            for (int q = 0; q < 2; ++q) {  // lane bit 3: + 2
                const unsigned keep = b3 ? s[q + 2] : s[q], give = b3 ? s[q] : s[q + 2];
                s[q] = keep + clip_xor8(give);
            }
            {  // lane bit 2: + 1
                const unsigned keep = b2 ? s[1] : s[0], give = b2 ? s[0] : s[1];
                s[0] = keep + clip_xor4(give);
            }
            unsigned T = s[0];
            T += clip_xor2(T);
            T += clip_xor1(T);  // every lane of quad c holds the total of chunk c
            const unsigned incl = wave_incl_scan_u32((lane & 3) == 0 ? T : 0u);
            const unsigned long long mk = __ballot(incl > k);
            const int fl = (int)__builtin_ctzll(mk);  // (mk != 0: the histogram holds m > k keys)
            const unsigned qs = (unsigned)fl >> 2;
            const unsigned k2 = k - (unsigned)__builtin_amdgcn_readlane((int)(incl - T), fl);
            const uint2 c = h2[qs * 64 + lane];
            const unsigned ps = c.x + c.y;
            const unsigned incl2 = wave_incl_scan_u32(ps);
            const unsigned long long mk2 = __ballot(incl2 > k2);
            const int l2 = (int)__builtin_ctzll(mk2);
            unsigned k3 = k2 - (unsigned)__builtin_amdgcn_readlane((int)(incl2 - ps), l2);
            const unsigned cx = (unsigned)__builtin_amdgcn_readlane((int)c.x, l2);
            const unsigned cy = (unsigned)__builtin_amdgcn_readlane((int)c.y, l2);
            B = qs * 128u + 2u * (unsigned)l2;
            cnt = cx;
            if (k3 >= cx) {
                k3 -= cx;
                B += 1u;
                cnt = cy;
            }
            kin = k3;
        }
        if constexpr (MODE == 0) stamp(29);
        // members of bin B: MODE 0 / 2 the keys of [ma, ma + mspan]; MODE 1 the samples whose key falls into bin B
        unsigned ma = a, mspan = 0u;
        if constexpr (MODE != 1) {
            const unsigned lo_b = (B + abin) << shift, hi_b = lo_b + ((1u << shift) - 1u);
            ma = lo_b > a ? lo_b : a;
            mspan = (hi_b < b ? hi_b : b) - ma;
        }
        // element -> (member?, key); every pass with its own opaque copies of med / scale
        auto classify = [&](unsigned x, float m2, float s2, unsigned &key) __attribute__((always_inline)) -> bool {
            key = x;
            if constexpr (MAD) {
                float d = __uint_as_float(x) - m2;
                asm("" : "+v"(d));
                key = __float_as_uint(fabsf(d));
                if constexpr (MODE == 1) return (unsigned)fabsf(d * s2) == B;
            }
            return (key - ma) <= mspan;
        };
        if (cnt <= 64u) {
            // gather without LDS atomics: the wave's count is a scalar, a member's slot comes from the ballot
            unsigned wc = 0;
            const float m2 = opaque(med), s2 = opaque(lin_scale);
            auto put = [&](bool mbr, unsigned key) __attribute__((always_inline)) {
                const unsigned long long mask = __ballot(mbr);
                if (mask) {
                    const unsigned at = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                                                  __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                    if (mbr) wl[wc + at] = key;
                    wc += (unsigned)__popcll(mask);
                }
            };
            if constexpr (MODE == 1) {
                const v2f m2v = {m2, m2}, s2v = {s2, s2};
                each4([&](unsigned x0, unsigned x1, unsigned x2, unsigned x3) __attribute__((always_inline)) {
                    const v2f da = v2f{__uint_as_float(x0), __uint_as_float(x1)} - m2v, pa = da * s2v;
                    const v2f db = v2f{__uint_as_float(x2), __uint_as_float(x3)} - m2v, pb = db * s2v;
                    put((unsigned)fabsf(pa.x) == B, __float_as_uint(fabsf(da.x)));
                    put((unsigned)fabsf(pa.y) == B, __float_as_uint(fabsf(da.y)));
                    put((unsigned)fabsf(pb.x) == B, __float_as_uint(fabsf(db.x)));
                    put((unsigned)fabsf(pb.y) == B, __float_as_uint(fabsf(db.y)));
                });
            } else {
                each([&](unsigned x) {
                    unsigned key;
                    const bool mbr = classify(x, m2, s2, key);
                    put(mbr, key);
                });
            }
            clip_wave_fence();
            const unsigned mine = (unsigned)lane < cnt ? wl[lane] : 0xffffffffu;
            unsigned rank = 0;
            for (unsigned j = 0; j < cnt; ++j) {
                const unsigned o = (unsigned)__builtin_amdgcn_readlane((int)mine, (int)j);
                rank += (o < mine || (o == mine && j < (unsigned)lane)) ? 1u : 0u;
            }
            const bool mem = (unsigned)lane < cnt;
            klo = (unsigned)__builtin_amdgcn_readlane((int)mine, (int)__builtin_ctzll(__ballot(mem && rank == kin)));
            if (kin + 1u < cnt) {
                khi = (unsigned)__builtin_amdgcn_readlane((int)mine, (int)__builtin_ctzll(__ballot(mem && rank == kin + 1u)));
                in_hi = true;
            }
            done = true;
            return;
        }
        // more than 64 members (copies of one value, mostly): their exact key range, then digits again
        unsigned mmn = 0xffffffffu, mmx = 0u;
        {
            const float m2 = opaque(med), s2 = opaque(lin_scale);
            each([&](unsigned x) {
                unsigned key;
                const bool mbr = classify(x, m2, s2, key);
                mmn = min(mmn, mbr ? key : 0xffffffffu);
                mmx = max(mmx, mbr ? key : 0u);
            });
        }
        sa = wave_min_u32(mmn);
        sb = wave_max_u32(mmx);
        sk = kin;
        if (sa == sb) {  // every member equals sa
            klo = khi = sa;
            in_hi = kin + 1u < cnt;
            done = true;
        }
    };
    auto select = [&](auto mad_t, unsigned k, unsigned a, unsigned b, unsigned m, const bool want_hi) __attribute__((always_inline)) {
        constexpr bool MAD = decltype(mad_t)::value;
        sk = k; sa = a; sb = b;
        in_hi = false;
        done = false;
        if (a == b) {
            klo = khi = a;
            in_hi = k + 1u < m;
        } else {
            if constexpr (MAD) level(std::integral_constant<int, 1>{}, mad_t);
            else level(std::integral_constant<int, 0>{}, mad_t);
#pragma clang loop unroll(disable)
            while (!done) level(std::integral_constant<int, 2>{}, mad_t);
        }
        if (want_hi && !in_hi) {  // rank k + 1 lies beyond klo's bin: the smallest key above klo
            unsigned nxt = 0xffffffffu;
            const unsigned kl = klo;
            float m2 = med;
            asm volatile("" : "+v"(m2));
            each([&](unsigned x) {
                unsigned key = x;
                if constexpr (MAD) {
                    float d = __uint_as_float(x) - m2;
                    asm("" : "+v"(d));
                    key = __float_as_uint(fabsf(d));
                }
                nxt = min(nxt, key > kl ? key : 0xffffffffu);
            });
            khi = wave_min_u32(nxt);
        }
    };

    const unsigned h = (unsigned)n / 2u;
    const bool odd = (n & 1) != 0;
    const unsigned m_all = (unsigned)ng * 256u;  // keys incl. the copies of the largest one
    select(std::false_type{}, odd ? h : h - 1u, umn, umx, m_all, !odd);
    stamp(30);
    if (neg && klo == umn) {  // the median does not lie above the clamped samples: not provably the original's
        out.flag = defer ? CLIP_NONE : CLIP_INEXACT;
        if (lane == 0) recs[r] = out;
        return;
    }
    med = odd ? __uint_as_float(klo) : (__uint_as_float(klo) + __uint_as_float(khi)) / 2.0f;
    // |x - med| is monotone on either side of med: its largest value is attained at one of the data extremes
    const float xmin = __uint_as_float(umn), xmax = __uint_as_float(umx);  // (xmin: of the clamped samples)
    const float dmin_ = fabsf(xmin - med), dmax_ = fabsf(xmax - med);
    const float dmax = fmaxf(dmin_, dmax_);
    const unsigned dmaxk = __float_as_uint(dmax);
    // lanes past the window: the sample whose key is the largest
    pad_last_group(__float_as_uint(dmin_ > dmax_ ? xmin : xmax));
    // (dmax == 0: every sample equals med and the select returns at once; else 0 < dmax < inf -- the extremes are
    // finite.  A dmax so small that the scale overflows (< 6e-36) is left to the exact kernel.)
    lin_scale = dmax > 0.0f ? (float)(kHB - 1) / dmax : 1.0f;
    if (!(lin_scale > 0.0f && lin_scale < 3.0e38f)) {
        out.flag = defer ? CLIP_NONE : CLIP_INEXACT;
        if (lane == 0) recs[r] = out;
        return;
    }
    stamp(31);
    select(std::true_type{}, odd ? h : h - 1u, 0u, dmaxk, m_all, !odd);
    stamp(24);
    if (neg && (odd ? klo : khi) >= __float_as_uint(dmin_)) {  // the MAD does not lie below the clamped samples' key
        out.flag = defer ? CLIP_NONE : CLIP_INEXACT;
        if (lane == 0) recs[r] = out;
        return;
    }
    const float mad = odd ? __uint_as_float(klo) : (__uint_as_float(klo) + __uint_as_float(khi)) / 2.0f;
    float lo, hi;
    clip_bounds(P, med, mad, lo, hi);
    // exactness gate of the event-mean sums (see fast_body P1)
    const float cmin = __builtin_amdgcn_fmed3f(xmin_true, lo, hi);
    const float cmax = __builtin_amdgcn_fmed3f(xmax, lo, hi);
    bool exact = (lo == lo) && (hi == hi) && lo <= hi && cmin > 0.0f && cmax < 3.0e38f;
    if (exact) {
        int fa = (int)(__float_as_uint(cmin) >> 23), fb = (int)(__float_as_uint(cmax) >> 23);
        if (fa == 0) fa = 1;
        exact = (fb - fa) + (32 - __clz(n)) <= 28;
    }
    out.lo = lo;
    out.hi = hi;
    out.cmax = cmax;
    out.flag = exact ? CLIP_OK : CLIP_INEXACT;
    if (lane == 0) recs[r] = out;
}

// workgroups per CU the register budget admits (512 VGPRs per SIMD lane, NPL of them are the samples)
constexpr int clip_wgs_per_cu(int npl) { return (npl <= 80 ? 4 : (npl <= 128 ? 3 : 2)) * (4 / kClipWaves); }

template <int NPL>
__global__ __launch_bounds__(kClipWaves * 64, clip_wgs_per_cu(NPL)) void clip_bounds_kernel(ClipArgs C) {
    __shared__ __attribute__((aligned(16))) unsigned clip_lds[kClipWaves][kClipWaveWords];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t r = C.a.block_base + (int64_t)blockIdx.x * kClipWaves + wave;
    if (r >= C.a.n_reads) return;
    clip_wave<NPL>(C.a, C.rec, r, C.cap, clip_lds[wave]);
}

// the reads of a device-side list (the main kernel's hand-overs to the 6144-sample list kernel): one wave per entry,
// waves past the end of the list and reads that already have their record leave at once
template <int NPL>
__global__ __launch_bounds__(kClipWaves * 64, clip_wgs_per_cu(NPL)) void clip_bounds_list_kernel(ClipArgs C) {
    __shared__ __attribute__((aligned(16))) unsigned clip_lds[kClipWaves][kClipWaveWords];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t k = C.a.block_base + (int64_t)blockIdx.x * kClipWaves + wave;
    if (k >= (int64_t)*C.in_count) return;
    const int64_t r = C.in_list[k];
    if (C.rec[r].flag != CLIP_NONE) return;
    clip_wave<NPL>(C.a, C.rec, r, C.cap, clip_lds[wave], C.defer != 0);
}

// The reads whose window is longer than the main instantiation takes, listed for the next instantiation AHEAD of the main
// kernel: one thread per read, one atomic per workgroup of 1024.  Listing them from the main kernel costs one returning atomic per
// workgroup on a single counter -- 11 ns each once every workgroup of the launch does nothing else (a batch of
// RNA002-length windows: 0.75 ms per 65 536 reads).  Same predicate as fast_body's (successful detection, window > cap).
__global__ __launch_bounds__(1024) void route_long_windows_kernel(FpArgs A, int cap, unsigned *big_count, int32_t *big_list) {
    __shared__ unsigned wcnt[16], wbase;
    const int64_t r = A.block_base + (int64_t)blockIdx.x * 1024 + threadIdx.x;
    const int wave = (int)(threadIdx.x >> 6);
    bool take = false;
    if (r < A.n_reads && !(A.ok && !A.ok[r])) {
        const int64_t row_len = A.row_len ? (int64_t)A.row_len[r] : (A.row_off ? A.row_off[r + 1] - A.row_off[r] : A.stride);
        int64_t start = (int64_t)A.a_start[r] - A.p.padding;
        if (start < 0) start = 0;
        int64_t stop = (int64_t)A.a_end[r] + A.p.padding;
        if (stop > row_len) stop = row_len;
        take = stop - start > (int64_t)cap;
    }
    // one atomic per WORKGROUP (1024 reads): the counter is one address, and returning atomics on it serialise at ~10 ns
    const unsigned long long mask = __ballot(take);
    if ((threadIdx.x & 63) == 0) wcnt[wave] = (unsigned)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned tot = 0;
        for (int w = 0; w < 16; ++w) tot += wcnt[w];
        wbase = tot ? atomicAdd(big_count, tot) : 0u;
    }
    __syncthreads();
    if (!take) return;
    unsigned base = wbase;
    for (int w = 0; w < wave; ++w) base += wcnt[w];
    const unsigned at = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
    big_list[base + at] = (int32_t)r;
}

int launch_route_long_windows(const FpArgs &A, int cap, unsigned *d_count, int32_t *d_list, hipStream_t stream) {
    FpArgs R = A;
    const int64_t n_wg = (A.n_reads + 1023) / 1024, max_slice = launch_slice_limit(1ll << 21);
    for (int64_t base = 0; base < n_wg; base += max_slice) {
        R.block_base = base * 1024;
        hipLaunchKernelGGL(route_long_windows_kernel, dim3((unsigned)std::min<int64_t>(max_slice, n_wg - base)), dim3(1024), 0, stream,
                           R, cap, d_count, d_list);
    }
    WDX_HIP_TRY(hipGetLastError());
    return WDX_SUCCESS;
}

// cap 6144 (the hand-overs of the main kernel), or the long-window lists ahead of clip_bounds_block_kernel: 8192 samples at 128
// registers per lane (three workgroups per CU), 13 312 at 208 (two; the register file ends at 256 per lane)
int launch_clip_bounds_list(const FpArgs &A, ClipRec *d_rec, const unsigned *d_count, const int32_t *d_list, int64_t n_entries,
                            hipStream_t stream, int cap, bool defer) {
    if (n_entries <= 0) return WDX_SUCCESS;
    if (cap != 6144 && cap != 8192 && cap != kClipWaveLongCap) {
        set_error("clip_bounds_list_kernel: capacities are 6144, 8192 and 13312 samples");
        return WDX_ERR_INVALID;
    }
    void (*kclip)(ClipArgs) = cap == 6144 ? clip_bounds_list_kernel<96> : (cap == 8192 ? clip_bounds_list_kernel<128> : clip_bounds_list_kernel<kClipWaveLongCap / 64>);
    ClipArgs CA{A, d_rec, cap, d_count, d_list, defer ? 1 : 0};
    const int64_t n_wg = (n_entries + kClipWaves - 1) / kClipWaves, max_slice = launch_slice_limit(1ll << 22);
    for (int64_t base = 0; base < n_wg; base += max_slice) {
        CA.a.block_base = base * kClipWaves;
        hipLaunchKernelGGL(kclip, dim3((unsigned)std::min<int64_t>(max_slice, n_wg - base)), dim3(kClipWaves * 64), 0, stream, CA);
    }
    WDX_HIP_TRY(hipGetLastError());
    return WDX_SUCCESS;
}

int launch_clip_bounds(const FpArgs &A, ClipRec *d_rec, int cap, hipStream_t stream) {
    if (cap > kClipWaveLongCap) {
        set_error("clip_bounds_kernel takes windows of at most 13312 samples");
        return WDX_ERR_INVALID;
    }
    // (the chain launches this form up to 6144 samples; beyond, the long-window lists take the list form -- the two long
    // instantiations are reachable here for the kernel's own tests)
    void (*kclip)(ClipArgs) = cap <= 4096 ? clip_bounds_kernel<64> : (cap <= 5120 ? clip_bounds_kernel<80> : (cap <= 6144 ? clip_bounds_kernel<96>
                              : (cap <= 8192 ? clip_bounds_kernel<128> : clip_bounds_kernel<kClipWaveLongCap / 64>)));
    ClipArgs CA{A, d_rec, cap, nullptr, nullptr, 0};
    const int64_t n_wg = (A.n_reads + kClipWaves - 1) / kClipWaves, max_slice = launch_slice_limit(1ll << 22);
    for (int64_t base = 0; base < n_wg; base += max_slice) {
        CA.a.block_base = base * kClipWaves;
        hipLaunchKernelGGL(kclip, dim3((unsigned)std::min<int64_t>(max_slice, n_wg - base)), dim3(kClipWaves * 64), 0, stream, CA);
    }
    WDX_HIP_TRY(hipGetLastError());
    return WDX_SUCCESS;
}

int launch_clip_bounds_selftest(const float *d_sig, const int64_t *d_row_off, int64_t stride, int64_t n_reads,
                                const int32_t *d_a_start, const int32_t *d_a_end, const wdx_seg_params &p, int cap,
                                void *d_rec, hipStream_t stream) {
    FpArgs A{};
    A.sig = d_sig;
    A.row_off = d_row_off;
    A.stride = stride;
    A.n_reads = n_reads;
    A.a_start = d_a_start;
    A.a_end = d_a_end;
    A.p = p;
    return launch_clip_bounds(A, reinterpret_cast<ClipRec *>(d_rec), cap, stream);
}

}  // namespace wdx

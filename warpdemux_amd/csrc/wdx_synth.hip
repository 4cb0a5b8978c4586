// Synthetic RNA004-like adapter signals generated on the device (bench / tests only).
// Spec "wdx-synth v1" -- see warpdemux_amd/synth.py, which is the bit-identical NumPy statement
// of the same integer-hash construction (SURVEY.md §8(d) describes the signal model).
#include "wdx_common.h"

namespace wdx {

constexpr int kPad = 100;
constexpr int kMaxEv = 140;
constexpr int kMinEv = 125;

__device__ __forceinline__ uint64_t synth_hash(uint64_t seed, uint64_t read, uint64_t stream,
                                               uint64_t ctr) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (read + 1ull);
    z ^= stream * 0xBF58476D1CE4E5B9ull;
    z += ctr * 0x94D049BB133111EBull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ void synth_lengths_kernel(uint64_t seed, int64_t first_read, int64_t n,
                                     const int32_t *__restrict__ dwell_table,
                                     int64_t *__restrict__ len) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t rid = (uint64_t)(first_read + i);
    const int n_ev = kMinEv + (int)(synth_hash(seed, rid, 0, 1) % 16ull);
    int64_t total = 2 * kPad;
    for (int e = 0; e < n_ev; ++e) total += dwell_table[synth_hash(seed, rid, 1, (uint64_t)e) & 1023ull];
    len[i] = total;
}

__global__ __launch_bounds__(256) void synth_fill_kernel(
    uint64_t seed, int64_t first_read, int64_t n, int32_t n_barcodes, int32_t n_bc_events,
    float noise_scale, int32_t spikes, const int32_t *__restrict__ dwell_table,
    const float *__restrict__ lead, const float *__restrict__ bc, const int64_t *__restrict__ off,
    float *__restrict__ sig, int32_t *__restrict__ barcode, int64_t block_base) {
    __shared__ int bound[kMaxEv + 1];  // bound[e] = first sample of event e (relative to row)
    __shared__ float level[kMaxEv];
    const int64_t r = block_base + blockIdx.x;
    const uint64_t rid = (uint64_t)(first_read + r);
    const int b = (int)(synth_hash(seed, rid, 0, 0) % (uint64_t)n_barcodes);
    const int n_ev = kMinEv + (int)(synth_hash(seed, rid, 0, 1) % 16ull);
    if (threadIdx.x < n_ev) {
        const int e = threadIdx.x;
        bound[e + 1] = dwell_table[synth_hash(seed, rid, 1, (uint64_t)e) & 1023ull];
        const int k = n_ev - 1 - e;
        level[e] = k < n_bc_events ? bc[b * 64 + k] : lead[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int acc = kPad;
        for (int e = 0; e < n_ev; ++e) {
            int d = bound[e + 1];
            bound[e] = acc;
            acc += d;
        }
        bound[n_ev] = acc;
        if (barcode) barcode[r] = b;
    }
    __syncthreads();
    const int64_t base = off[r];
    const int len = (int)(off[r + 1] - base);
    const int ev_end = bound[n_ev];
    for (int t = threadIdx.x; t < len; t += blockDim.x) {
        float lv;
        if (t < kPad) lv = 95.0f;
        else if (t >= ev_end) lv = 105.0f;
        else {
            int lo = 0, hi = n_ev;  // largest e with bound[e] <= t
            while (hi - lo > 1) {
                int mid = (lo + hi) >> 1;
                if (bound[mid] <= t) lo = mid;
                else hi = mid;
            }
            lv = level[lo];
        }
        const uint64_t hn = synth_hash(seed, rid, 2, (uint64_t)t);
        const int isum = (int)(hn & 0xFFFFull) + (int)((hn >> 16) & 0xFFFFull) +
                         (int)((hn >> 32) & 0xFFFFull) + (int)((hn >> 48) & 0xFFFFull) - 131070;
        const float noise = (float)isum * noise_scale;
        float s = lv + noise;
        if (spikes) {
            const uint64_t hs = synth_hash(seed, rid, 3, (uint64_t)t);
            if (hs % 1000ull == 0ull) s = s + (((hs >> 32) & 1ull) ? 60.0f : -60.0f);
        }
        sig[base + t] = s;
    }
}

// Diagnostic: streams n floats with one coalesced dword load per lane (the fingerprint kernel's
// global access pattern) so the FETCH_SIZE counter can be calibrated against a known byte count.
__global__ void calib_read_dword_kernel(const float *__restrict__ p, int64_t n, float *__restrict__ out) {
    float acc = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        acc += p[i];
    if (acc == 123456.789f) out[0] = acc;  // keep the loads alive without a measurable write
}

int launch_calib_read(const float *p, int64_t n, float *out, hipStream_t stream) {
    hipLaunchKernelGGL(calib_read_dword_kernel, dim3(256 * 8), dim3(256), 0, stream, p, n, out);
    WDX_HIP_TRY(hipGetLastError());
    return WDX_SUCCESS;
}

int launch_synth_lengths(uint64_t seed, int64_t first_read, int64_t n, int32_t n_barcodes,
                         const int32_t *dwell_table, int64_t *d_len, hipStream_t stream) {
    (void)n_barcodes;
    if (n == 0) return WDX_SUCCESS;
    hipLaunchKernelGGL(synth_lengths_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                       seed, first_read, n, dwell_table, d_len);
    WDX_HIP_TRY(hipGetLastError());
    return WDX_SUCCESS;
}

int launch_synth_fill(uint64_t seed, int64_t first_read, int64_t n, int32_t n_barcodes,
                      int32_t n_bc_events, float noise_scale, int32_t spikes,
                      const int32_t *dwell_table, const float *lead, const float *bc,
                      const int64_t *off, float *sig, int32_t *barcode, hipStream_t stream) {
    if (n == 0) return WDX_SUCCESS;
    if (n_bc_events < 0 || n_bc_events > 64 || n_barcodes < 1) {
        set_error("synth: n_bc_events must be in [0,64] and n_barcodes >= 1");
        return WDX_ERR_INVALID;
    }
    const int64_t slice = 1 << 22;  // grid.x * block.x must stay below 2^32
    for (int64_t base = 0; base < n; base += slice) {
        const int64_t m = n - base < slice ? n - base : slice;
        hipLaunchKernelGGL(synth_fill_kernel, dim3((unsigned)m), dim3(256), 0, stream, seed,
                           first_read, n, n_barcodes, n_bc_events, noise_scale, spikes, dwell_table,
                           lead, bc, off, sig, barcode, base);
    }
    WDX_HIP_TRY(hipGetLastError());
    return WDX_SUCCESS;
}

// ---- packed staging of a page-locked minibatch (wdx_api.hip: demux_batch_enqueue) ------------------------------------
// One workgroup per read: dword loads from the mapped host row (16 in flight per thread: the bus is 49 GB/s and ~1.5 us
// away), coalesced dword stores to the packed device row.
__global__ __launch_bounds__(256) void pack_windows_kernel(const float *__restrict__ src, int64_t stride, const int64_t *__restrict__ off,
                                                           const int32_t *__restrict__ st, const int32_t *__restrict__ len,
                                                           float *__restrict__ dst) {
    const int64_t r = blockIdx.x;
    const int n = len[r];
    const float *__restrict__ s = src + r * stride + st[r];
    float *__restrict__ d = dst + off[r];
    if (((uintptr_t)s & 15) == 0) {
        // 16-byte loads over the bus (the host passes 16-byte aligned window starts whenever the row stride allows): a
        // wave asks for 1 KB of consecutive bytes per instruction, four instructions in flight per thread
        const int n4 = n >> 2;
        const float4 *__restrict__ s4 = reinterpret_cast<const float4 *>(s);
        float4 *__restrict__ d4 = reinterpret_cast<float4 *>(d);
        for (int i0 = threadIdx.x; i0 < n4; i0 += 256 * 4) {
            float4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = i0 + 256 * k;
                v[k] = i < n4 ? s4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = i0 + 256 * k;
                if (i < n4) d4[i] = v[k];
            }
        }
        const int i = 4 * n4 + threadIdx.x;
        if (i < n) d[i] = s[i];
        return;
    }
    for (int i0 = threadIdx.x; i0 < n; i0 += 256 * 16) {
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int i = i0 + 256 * k;
            v[k] = i < n ? s[i] : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int i = i0 + 256 * k;
            if (i < n) d[i] = v[k];
        }
    }
}
int launch_pack_windows(const float *src_dev, int64_t stride, int64_t n_reads, const int64_t *d_off, const int32_t *d_st,
                        const int32_t *d_len, float *dst, hipStream_t stream) {
    if (n_reads == 0) return WDX_SUCCESS;
    hipLaunchKernelGGL(pack_windows_kernel, dim3((unsigned)n_reads), dim3(256), 0, stream, src_dev, stride, d_off, d_st, d_len, dst);
    WDX_HIP_TRY(hipGetLastError());
    return WDX_SUCCESS;
}

}  // namespace wdx

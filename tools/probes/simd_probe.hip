// Which SIMD of its CU does wave w of a 256-thread workgroup run on?  Answer (MI355X, round 5): the dispatcher rotates the
// first SIMD from workgroup to workgroup -- wave 0 lands on each SIMD a quarter of the time (50109 / 49964 / 49912 / 50015 of
// 200 000), the four waves always on four different SIMDs.  So the stretches one wave works alone (below) do NOT pile up
// on one SIMD, and rotating them over the waves by read index measured +0.2 % (noise).
// one wave of the workgroup works alone -- fast_mean_sd, the boundary threads of the event means: if wave 0 always
// lands on the same SIMD, that SIMD carries all of them.)  Workgroups shaped like the main kernel's: 256 threads, 32 KB
// of LDS, five per CU, a grid many times the chip, a little work per workgroup so that dispatch is in steady state.
// Build: hipcc --offload-arch=gfx950 -O2 tools/probes/simd_probe.hip -o tools/probes/simd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256, 5) void k(unsigned *out, int spin) {
    extern __shared__ unsigned char smem[];
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    float x = (float)threadIdx.x;
    for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f;   // keep the CU busy: steady-state dispatch
    smem[threadIdx.x] = (unsigned char)x;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) out[(size_t)blockIdx.x * 4 + (threadIdx.x >> 6)] = hw | ((unsigned)smem[(threadIdx.x + 1) & 255] << 31);
}
int main() {
    const int G = 200000;
    unsigned *out;
    hipMalloc(&out, (size_t)G * 16);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 31984);
    hipLaunchKernelGGL(k, dim3(G), dim3(256), 31984, 0, out, 2000);
    hipDeviceSynchronize();
    std::vector<unsigned> h((size_t)G * 4);
    hipMemcpy(h.data(), out, (size_t)G * 16, hipMemcpyDeviceToHost);
    long hist[4][4] = {};
    long same_wg_distinct = 0;
    for (int b = 0; b < G; ++b) {
        unsigned seen = 0;
        for (int w = 0; w < 4; ++w) {
            const unsigned simd = (h[(size_t)b * 4 + w] >> 4) & 3u;   // HW_ID[5:4] = SIMD_ID (gfx9)
            hist[w][simd]++;
            seen |= 1u << simd;
        }
        same_wg_distinct += seen == 0xf;
    }
    printf("workgroups whose four waves sit on four different SIMDs: %ld of %d\n", same_wg_distinct, G);
    for (int w = 0; w < 4; ++w)
        printf("wave %d of its workgroup -> SIMD 0..3: %7ld %7ld %7ld %7ld\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    return 0;
}

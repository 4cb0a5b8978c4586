// Cost of launching workgroups that read one counter and leave (the over-provisioned grids of the list kernels).
// Build: hipcc --offload-arch=gfx950 -O2 tools/probes/empty_wg_probe.hip -o tools/probes/empty_wg_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(const unsigned *cnt, int *out) {
    extern __shared__ unsigned char smem[];
    if (blockIdx.x >= *cnt) return;
    smem[threadIdx.x] = 1;
    __syncthreads();
    out[blockIdx.x] = smem[(threadIdx.x + 1) & 255];
}
int main() {
    unsigned *cnt; int *out;
    hipMalloc(&cnt, 4); hipMalloc(&out, 4 << 20);
    hipMemset(cnt, 0, 4);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 40960);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (long g : {1024L, 100000L, 1000000L, 2500000L, 5000000L}) {
        for (size_t lds : {0ul, 40432ul}) {
            float best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(k, dim3((unsigned)g), dim3(256), lds, 0, cnt, out);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("grid %8ld  lds %6zu B: %.3f ms\n", g, lds, best);
        }
    }
    return 0;
}

// Accuracy of v_rcp_f64 (+ one Newton step) and issue cost of v_rcp_f64 / v_rsq_f64 relative to v_fma_f64 on gfx950.
// Build: hipcc --offload-arch=gfx950 -O2 tools/probes/rcp_probe.hip -o tools/probes/rcp_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include <random>

__global__ void acc_kernel(const double *x, double *o0, double *o1, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double r = __builtin_amdgcn_rcp(x[i]);
    o0[i] = r;
    double e = __builtin_fma(-x[i], r, 1.0);
    o1[i] = __builtin_fma(r, e, r);
}

template <int MODE>
__global__ void rate_kernel(double *out, int iters) {
    double a[8];
    for (int k = 0; k < 8; ++k) a[k] = 1.0 + threadIdx.x * 1e-3 + k;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (MODE == 0) a[k] = __builtin_fma(a[k], 0.999, 0.5);
            if (MODE == 1) a[k] = __builtin_amdgcn_rcp(a[k]);
            if (MODE == 2) a[k] = __builtin_amdgcn_rsq(a[k]);
            if (MODE == 3) a[k] = (double)__builtin_amdgcn_rcpf((float)a[k]);
        }
    }
    double s = 0;
    for (int k = 0; k < 8; ++k) s += a[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    const int n = 1 << 22;
    std::vector<double> h(n), r0(n), r1(n);
    std::mt19937_64 g(1);
    std::uniform_real_distribution<double> U(-20.0, 40.0);
    for (auto &v : h) v = std::exp2(U(g));
    double *d, *o0, *o1;
    hipMalloc(&d, n * 8); hipMalloc(&o0, n * 8); hipMalloc(&o1, n * 8);
    hipMemcpy(d, h.data(), n * 8, hipMemcpyHostToDevice);
    acc_kernel<<<n / 256, 256>>>(d, o0, o1, n);
    hipMemcpy(r0.data(), o0, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(r1.data(), o1, n * 8, hipMemcpyDeviceToHost);
    double m0 = 0, m1 = 0;
    for (int i = 0; i < n; ++i) {
        long double t = 1.0L / (long double)h[i];
        m0 = std::fmax(m0, (double)fabsl(((long double)r0[i] - t) / t));
        m1 = std::fmax(m1, (double)fabsl(((long double)r1[i] - t) / t));
    }
    printf("v_rcp_f64 max rel err 2^%.2f ; after one Newton step 2^%.2f\n", std::log2(m0), std::log2(m1));
    double *out; hipMalloc(&out, 1024 * 256 * 4 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    float ms[4];
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) rate_kernel<0><<<4096, 256>>>(out, iters);
            if (mode == 1) rate_kernel<1><<<4096, 256>>>(out, iters);
            if (mode == 2) rate_kernel<2><<<4096, 256>>>(out, iters);
            if (mode == 3) rate_kernel<3><<<4096, 256>>>(out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms[mode], e0, e1);
        }
    }
    printf("issue cost relative to v_fma_f64: v_rcp_f64 %.2f  v_rsq_f64 %.2f  cvt+v_rcp_f32+cvt %.2f (fma %.3f ms)\n",
           ms[1] / ms[0], ms[2] / ms[0], ms[3] / ms[0], ms[0]);
    return 0;
}

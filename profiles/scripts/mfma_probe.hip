#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int KIND>
__global__ __launch_bounds__(256) void probe(double* sink, long long* cyc, int iters, double seed) {
    v4d a0 = {0,0,0,0}, a1 = a0, a2 = a0, a3 = a0;
    double b0 = 0, b1 = 0, b2 = 0, b3 = 0;
    double x = seed + threadIdx.x * 1e-3, y = seed - threadIdx.x * 1e-3;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, a3, 0, 0, 0);
        } else if (KIND == 1) {
            b0 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, b0, 0, 0, 0);
            b1 = __builtin_amdgcn_mfma_f64_4x4x4f64(y, x, b1, 0, 0, 0);
            b2 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, x, b2, 0, 0, 0);
            b3 = __builtin_amdgcn_mfma_f64_4x4x4f64(y, y, b3, 0, 0, 0);
        } else {
            b0 = fma(x, y, b0); b1 = fma(y, x, b1); b2 = fma(x, x, b2); b3 = fma(y, y, b3);
            b0 = fma(x, b1, b0); b1 = fma(y, b2, b1); b2 = fma(x, b3, b2); b3 = fma(y, b0, b3);
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    v4d t = a0 + a1 + a2 + a3;
    double s = t[0] + t[1] + t[2] + t[3] + b0 + b1 + b2 + b3;
    if (s == 12345.678) sink[blockIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int KIND> void run(const char* name, int blocks, double flops_per_iter_per_wave) {
    double* sink; long long* cyc; hipMalloc(&sink, 1 << 20); hipMalloc(&cyc, 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    int iters = 4000;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a); hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, sink, cyc, iters, 1.0 + rep);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        double tf = (double)blocks * 4 * iters * flops_per_iter_per_wave / (ms * 1e-3) * 1e-12;
        if (rep == 2) printf("%-28s blocks %5d: %.3f ms  %.1f TF/s  block0 cycles/iter %.1f (memtime ticks @100MHz?)\n", name, blocks, ms, tf, (double)c / iters);
    }
}
int main() {
    for (int blocks : {256, 512, 1024, 2048}) {
        run<0>("mfma_f64_16x16x4 x4", blocks, 4 * 2048.0);
        run<1>("mfma_f64_4x4x4_4b x4", blocks, 4 * 512.0);
        run<2>("v_fma_f64 x8", blocks, 8 * 2.0 * 64);
    }
    return 0;
}

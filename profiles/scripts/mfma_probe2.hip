// fp64 MFMA issue-rate probe, round 2: how many independent accumulators / waves per SIMD the
// 16x16x4 and the 4x4x4 (4 blocks) forms need to reach their pipe rate, and the clock they run at.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_probe2 mfma_probe2.hip && ./mfma_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int KIND, int NACC, int THREADS>
__global__ __launch_bounds__(THREADS) void probe(double *sink, long long *cyc, int iters, double seed)
{
    v4d a[NACC];
    double b[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) { a[i] = {0, 0, 0, 0}; b[i] = 0; }
    double x = seed + threadIdx.x * 1e-3, y = seed - threadIdx.x * 1e-3;
    long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if (KIND == 0) a[i] = __builtin_amdgcn_mfma_f64_16x16x4f64((i & 1) ? x : y, (i & 2) ? x : y, a[i], 0, 0, 0);
            else b[i] = __builtin_amdgcn_mfma_f64_4x4x4f64((i & 1) ? x : y, (i & 2) ? x : y, b[i], 0, 0, 0);
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += a[i][0] + a[i][1] + a[i][2] + a[i][3] + b[i];
    if (s == 12345.678) sink[blockIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = r1 - r0; }
}

template <int KIND, int NACC, int THREADS>
void run(int blocks)
{
    static double *sink = nullptr;
    static long long *cyc = nullptr;
    if (!sink) { hipMalloc(&sink, 1 << 20); hipMalloc(&cyc, 16); }
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int iters = 64000 / NACC;
    const double fl = KIND == 0 ? 2048.0 : 512.0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL((probe<KIND, NACC, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, sink, cyc, iters, 1.0 + rep);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        long long c[2];
        hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
        const double tf = (double)blocks * (THREADS / 64) * iters * NACC * fl / (ms * 1e-3) * 1e-12;
        if (rep == 2)
            printf("%-10s acc %2d  waves/SIMD %.1f (threads %4d x blocks %4d): %7.3f ms %6.1f TF/s  cyc/mfma/wave %6.1f  clock %.2f GHz\n",
                   KIND == 0 ? "16x16x4" : "4x4x4_4b", NACC, (double)blocks * THREADS / 64 / 1024, THREADS, blocks, ms, tf,
                   (double)c[0] / ((double)iters * NACC), (double)c[0] / (double)c[1] * 0.1);
    }
}

int main()
{
    run<0, 4, 256>(256);  run<0, 8, 256>(256);  run<0, 16, 256>(256); run<0, 28, 256>(256);
    run<0, 4, 512>(256);  run<0, 8, 512>(256);  run<0, 16, 512>(256); run<0, 28, 512>(256);
    run<0, 4, 256>(1024); run<0, 8, 256>(1024); run<0, 16, 256>(1024);
    run<1, 4, 256>(256);  run<1, 8, 256>(256);  run<1, 16, 256>(256); run<1, 32, 256>(256);
    run<1, 4, 512>(256);  run<1, 8, 512>(256);  run<1, 16, 512>(256); run<1, 32, 512>(256);
    run<1, 8, 256>(1024); run<1, 16, 256>(1024);
    return 0;
}

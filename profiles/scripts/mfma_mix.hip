// How v_mfma_f64_4x4x4 (4 blocks) behaves inside a stream of v_mfma_f64_16x16x4: cycles per group of
// N16 16x16x4 + N4 4x4x4 instructions, one wave per SIMD (256 threads) and two (512), all CUs busy.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_mix mfma_mix.hip && ./mfma_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int N16, int N4, int THREADS, int SAMEACC>
__global__ __launch_bounds__(THREADS) void probe(double *sink, long long *cyc, int iters, double seed)
{
    v4d a[N16 ? N16 : 1];
    double b[N4 ? N4 : 1];
#pragma unroll
    for (int i = 0; i < N16; ++i) a[i] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < N4; ++i) b[i] = 0;
    double x = seed + threadIdx.x * 1e-3, y = seed - threadIdx.x * 1e-3;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < N16; ++i) a[i] = __builtin_amdgcn_mfma_f64_16x16x4f64((i & 1) ? x : y, (i & 2) ? x : y, a[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < N4; ++i) b[SAMEACC ? 0 : i] = __builtin_amdgcn_mfma_f64_4x4x4f64((i & 1) ? x : y, (i & 2) ? x : y, b[SAMEACC ? 0 : i], 0, 0, 0);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < N16; ++i) s += a[i][0] + a[i][1] + a[i][2] + a[i][3];
#pragma unroll
    for (int i = 0; i < N4; ++i) s += b[i];
    if (s == 12345.678) sink[blockIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int N16, int N4, int THREADS, int SAMEACC = 0>
void run()
{
    static double *sink = nullptr;
    static long long *cyc = nullptr;
    if (!sink) { (void)hipMalloc(&sink, 1 << 20); (void)hipMalloc(&cyc, 16); }
    const int iters = 20000;
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL((probe<N16, N4, THREADS, SAMEACC>), dim3(256), dim3(THREADS), 0, 0, sink, cyc, iters, 1.0 + rep);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        (void)hipEventElapsedTime(&ms, a, b);
    }
    long long c = 0;
    (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double per_group = (double)c / iters;
    printf("N16=%2d N4=%2d waves/SIMD %d sameacc %d: %7.1f cycles per group (wave 0), ideal %4d, kernel %.3f ms = %.1f cycles per group per SIMD\n", N16,
           N4, THREADS / 256, SAMEACC, per_group, 64 * N16 + 16 * N4, ms, ms * 1e-3 * 2.4e9 / iters / (THREADS / 256));
}

int main()
{
    run<6, 0, 256>(); run<0, 4, 256>(); run<6, 1, 256>(); run<6, 1, 256, 1>(); run<12, 2, 256>(); run<6, 4, 256>(); run<30, 5, 256>(); run<3, 1, 256>();
    run<6, 0, 512>(); run<6, 1, 512>(); run<12, 2, 512>(); run<6, 4, 512>(); run<3, 1, 512>();
    return 0;
}

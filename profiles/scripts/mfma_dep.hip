// How long does a DEPENDENT v_mfma_f64_16x16x4 take (one accumulator chain per wave), with register operands and with
// operands read from LDS per instruction?  One workgroup; waves = 1, 4, 16; chains per wave = 1, 4.
// Build and run on the GPU box: hipcc --offload-arch=gfx950 -O3 profiles/scripts/mfma_dep.hip -o /tmp/mfma_dep && /tmp/mfma_dep
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int CH, bool LDS>
__global__ __launch_bounds__(1024) void k(double *out, long long *cyc, int n)
{
    __shared__ double sm[64 * 66];
    const int lane = threadIdx.x & 63;
    for (int e = threadIdx.x; e < 64 * 66; e += blockDim.x) sm[e] = 1e-3 * (e % 7);
    __syncthreads();
    v4d acc[CH];
    for (int c = 0; c < CH; ++c) acc[c] = v4d{0, 0, 0, 0};
    double a = 1.0 + lane * 1e-6, b = 1.0 - lane * 1e-6;
    const double *pa = sm + (lane >> 4) * 66 + (lane & 15);
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (LDS) { a = pa[((4 * i + c) & 15) * 264]; b = pa[((4 * i + c + 1) & 15) * 264 + 16]; }
            acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int CH, bool LDS>
void run(int threads, int n)
{
    double *out; long long *cyc, h = 0;
    hipMalloc(&out, 1024 * 8); hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<CH, LDS>), dim3(1), dim3(threads), 0, 0, out, cyc, n);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("waves %2d chains %d %s: %6.1f us for %d x %d MFMA per wave -> %.1f ns per MFMA of a chain step, counter %lld (%.1f per step)\n", threads / 64, CH,
           LDS ? "LDS operands" : "reg operands", ms * 1e3, n, CH, ms * 1e6 / n, h, (double)h / n);
    hipFree(out); hipFree(cyc);
}
int main()
{
    const int n = 2000;
    for (int th : {64, 256, 1024}) {
        run<1, false>(th, n); run<4, false>(th, n); run<1, true>(th, n); run<4, true>(th, n);
    }
    return 0;
}

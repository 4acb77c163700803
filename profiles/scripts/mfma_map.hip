#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int CBSZ, int ABID>
__device__ double m4(double a, double b) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, CBSZ, ABID, 0); }
__global__ void probe(double* out) {  // out[variant 5][la 64][lb 64][lane 64]
    int la = blockIdx.x / 64, lb = blockIdx.x % 64, lane = threadIdx.x;
    double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
    double r[5] = {m4<0, 0>(a, b), m4<2, 0>(a, b), m4<2, 1>(a, b), m4<2, 2>(a, b), m4<2, 3>(a, b)};
    for (int v = 0; v < 5; ++v) out[((size_t)(v * 64 + la) * 64 + lb) * 64 + lane] = r[v];
}
int main() {
    size_t n = 5ull * 64 * 64 * 64; double* d; hipMalloc(&d, n * 8);
    hipLaunchKernelGGL(probe, dim3(4096), dim3(64), 0, 0, d);
    std::vector<double> h(n); hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
    const char* names[5] = {"cbsz0", "cbsz2 abid0", "cbsz2 abid1", "cbsz2 abid2", "cbsz2 abid3"};
    for (int v = 0; v < 5; ++v) {
        printf("== %s: for output lanes 0,1,4,5,16,17,21,63: contributing (A lane, B lane) pairs\n", names[v]);
        for (int lane : {0, 1, 4, 5, 16, 17, 21, 63}) {
            printf("  D[lane %2d] = ", lane);
            for (int la = 0; la < 64; ++la) for (int lb = 0; lb < 64; ++lb)
                if (h[((size_t)(v * 64 + la) * 64 + lb) * 64 + lane] != 0.0) printf("a%d*b%d ", la, lb);
            printf("\n");
        }
    }
    // check the conjecture: block = lane/16; within block A[i][k] at i+4k, B[k][j] at j+4k, D[i][j] at j+4i
    int bad = 0;
    for (int v = 0; v < 5; ++v) for (int la = 0; la < 64; ++la) for (int lb = 0; lb < 64; ++lb) for (int lane = 0; lane < 64; ++lane) {
        int blk = lane / 16, i = (lane % 16) / 4, j = lane % 4;
        int ablk = v == 0 ? blk : v - 1;
        double want = 0;
        for (int k = 0; k < 4; ++k) if (la == ablk * 16 + i + 4 * k && lb == blk * 16 + j + 4 * k) want = 1;
        if (h[((size_t)(v * 64 + la) * 64 + lb) * 64 + lane] != want) ++bad;
    }
    printf("conjecture mismatches: %d\n", bad);
    return 0;
}

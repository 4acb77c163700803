// Tall-skinny products of the orthogonalising sketches (tt_orth.hip) with the Gram matrix of the result from the
// accumulator registers: Y = A B (A: m x K with m ~ 10^4, B: K x n, n <= 64) and, in the same launch, the partial
// sums of Y^T Y -- CholeskyQR2's "product, then Gram matrix of the product" pairs (M = T W with M^T M; Q1 = M R1^-1
// with Q1^T Q1) read A once and never read Y back (sketch_dispatch.py:160-174: the QR of the (r1 n) x r2 unfolding).
//
// One wavefront owns a 16-row tile: the A operands come straight from global memory in the MFMA layout (16 rows x 4
// consecutive k per instruction), B sits zero-padded in LDS, the tile of Y leaves the accumulators for memory and
// -- register j of lane l holding Y[(l >> 4) + 4 j][l & 15] -- is at the same time both operands of Y^T Y's k-block j.
// The four waves of a workgroup add their Gram partials in LDS; gram_reduce_kernel sums the workgroups in a fixed
// order (bit-reproducible, no atomics).
#include <algorithm>
#include "common.h"
#include "skinny.h"
#include "linalg_int.h"

namespace ttsk {

struct TallMul {
    const double *A; int64_t lda;
    const double *B; int64_t ldb;
    double *Y; int64_t ldy;
    double *slab;               // GRAM: gridDim.x x (pairs x 256) partial Gram tiles in accumulator order
    int64_t m;
    int K, n, tiles;
};

constexpr int TM_KC = 32;       // k-blocks (of 4) whose A operands a wave holds at once: all of K <= 128 in one go

__host__ __device__ constexpr int tm_sb(int nt) { return nt == 1 ? 16 : (nt <= 3 ? 48 : 80); }   // row stride of B in LDS: = 16 mod 32

template <int NT, bool GRAM>
__global__ __launch_bounds__(512) void tall_mul_kernel(TallMul p)   // launched with 256 threads: the bound keeps the compiler off the AGPR file
{
    extern __shared__ double sm[];
    constexpr int SB = tm_sb(NT), NP = NT * (NT + 1) / 2;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, x16 = lane & 15, g = lane >> 4;
    const int K4 = (p.K + 3) >> 2;
    int tile = blockIdx.x * 4 + wv;
    // the A operands of the first tile are on their way while B is staged
    double a[TM_KC];
    // (unconditional loads from clamped addresses: rows of B beyond K are zero in LDS, rows beyond m are masked after
    // the product -- a select behind every load makes the compiler wait for each one)
    auto load_a = [&](int t, int kb0) {
        const int64_t row = (int64_t)t * 16 + x16;
        const double *ar = p.A + (row < p.m ? row : p.m - 1) * p.lda;
#pragma unroll
        for (int i = 0; i < TM_KC; ++i) {
            const int k = 4 * (kb0 + i) + g;
            a[i] = ar[k < p.K ? k : p.K - 1];
        }
    };
    load_a(tile < p.tiles ? tile : 0, 0);
    {
        // B (K x n) -> LDS rows of SB doubles, zero padded; eight loads in flight per thread
        const int total = K4 * 4 * SB;
        for (int e0 = tid; e0 < total; e0 += 256 * 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = e0 + 256 * u, k = e / SB, c = e - k * SB;
                const bool ok = e < total && k < p.K && c < p.n;
                v[u] = ok ? p.B[(int64_t)k * p.ldb + c] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (e0 + 256 * u < total) sm[e0 + 256 * u] = v[u];
        }
    }
    __syncthreads();
    v4d gacc[GRAM ? NP : 1];
#pragma unroll
    for (int q = 0; q < (GRAM ? NP : 1); ++q) gacc[q] = v4d{0.0, 0.0, 0.0, 0.0};
    for (; tile < p.tiles; tile += gridDim.x * 4) {
        v4d acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = v4d{0.0, 0.0, 0.0, 0.0};
        for (int kb0 = 0; kb0 < K4; kb0 += TM_KC) {
            if (kb0) load_a(tile, kb0);
#pragma unroll
            for (int i = 0; i < TM_KC; ++i) {
                if (kb0 + i < K4) {
                    const double *br = sm + (4 * (kb0 + i) + g) * SB + x16;
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[t] = mfma16(a[i], br[16 * t], acc[t]);
                }
                if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);       // keeps the LDS reads of later k-blocks out of the registers
            }
        }
        if ((int64_t)tile * 16 + 16 > p.m) {                                         // the last tile: rows beyond m are not there
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if ((int64_t)tile * 16 + g + 4 * j >= p.m) acc[t][j] = 0.0;
        }
        if (tile + gridDim.x * 4 < p.tiles) load_a(tile + gridDim.x * 4, 0);       // the next tile's operands behind the stores
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t r = (int64_t)tile * 16 + g + 4 * j;
                const int c = 16 * t + x16;
                if (r < p.m && c < p.n) p.Y[r * p.ldy + c] = acc[t][j];
            }
        if (GRAM) {
            int q = 0;
#pragma unroll
            for (int t1 = 0; t1 < NT; ++t1)
#pragma unroll
                for (int t2 = t1; t2 < NT; ++t2, ++q)
#pragma unroll
                    for (int j = 0; j < 4; ++j) gacc[q] = mfma16(acc[t1][j], acc[t2][j], gacc[q]);
        }
    }
    if (GRAM) {
        __syncthreads();                                   // every wave is done with B
        double *gs = sm + (size_t)wv * NP * 256;
#pragma unroll
        for (int q = 0; q < NP; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) gs[q * 256 + j * 64 + lane] = gacc[q][j];
        __syncthreads();
        double *out = p.slab + (size_t)blockIdx.x * NP * 256;
        for (int e = tid; e < NP * 256; e += 256)
            out[e] = (sm[e] + sm[NP * 256 + e]) + (sm[2 * NP * 256 + e] + sm[3 * NP * 256 + e]);
    }
}

// G (n x n, symmetric, full) = sum over the workgroups of their partial tiles; one block per tile pair, four groups
// of 256 threads share the sum
__global__ __launch_bounds__(1024) void gram_reduce_kernel(const double *__restrict__ slab, int nwg, int nt, int n, double *__restrict__ G)
{
    __shared__ double part[4][256];
    const int np = nt * (nt + 1) / 2, q = blockIdx.x, e = threadIdx.x & 255, grp = threadIdx.x >> 8;
    double acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = 0.0;
    for (int w0 = grp; w0 < nwg; w0 += 32) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int w = w0 + 4 * u;
            if (w < nwg) acc[u] += slab[((size_t)w * np + q) * 256 + e];
        }
    }
    const double s0 = (acc[0] + acc[1]) + (acc[2] + acc[3]), s1 = (acc[4] + acc[5]) + (acc[6] + acc[7]);
    part[grp][e] = s0 + s1;
    __syncthreads();
    if (grp) return;
    const double v = (part[0][e] + part[1][e]) + (part[2][e] + part[3][e]);
    int t1 = 0, rest = q;
    while (rest >= nt - t1) { rest -= nt - t1; ++t1; }
    const int t2 = t1 + rest, lane = e & 63, j = e >> 6;
    const int r = 16 * t1 + (lane >> 4) + 4 * j, c = 16 * t2 + (lane & 15);
    if (r < n && c < n) {
        G[(size_t)r * n + c] = v;
        if (t1 != t2) G[(size_t)c * n + r] = v;
    }
}

size_t tall_mul_ws_elems(int64_t m, int n)
{
    const int nt = (n + 15) >> 4;
    const int64_t tiles = (m + 15) >> 4;
    const int64_t nwg = std::min<int64_t>((tiles + 3) / 4, 256);
    return (size_t)nwg * (nt * (nt + 1) / 2) * 256;
}

template <int NT>
static int launch_tm(const TallMul &p, bool gram, int nwg, size_t lds, hipStream_t st)
{
    static bool attr = false;
    if (!attr) {
        TTSK_HIP(hipFuncSetAttribute((const void *)tall_mul_kernel<NT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        TTSK_HIP(hipFuncSetAttribute((const void *)tall_mul_kernel<NT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        attr = true;
    }
    if (gram) hipLaunchKernelGGL((tall_mul_kernel<NT, true>), dim3(nwg), dim3(256), lds, st, p);
    else hipLaunchKernelGGL((tall_mul_kernel<NT, false>), dim3(nwg), dim3(256), lds, st, p);
    TTSK_LAUNCH_CHECK();
    return 1;
}

// Y (m x n, row stride ldy) = A (m x K, row stride lda) B (K x n, row stride ldb); G != nullptr: also G (n x n) = Y^T Y
// (slab: tall_mul_ws_elems(m, n) doubles).  1 = queued, 0 = shape outside the kernel (n > 64 or B beyond the LDS).
int tall_mul(const double *A, int64_t lda, int K, const double *B, int64_t ldb, double *Y, int64_t ldy, int64_t m, int n,
             double *G, double *slab, hipStream_t st)
{
    const int nt = (n + 15) >> 4;
    if (nt < 1 || nt > 4 || K < 1 || m < 1) return 0;
    const int K4 = (K + 3) >> 2, np = nt * (nt + 1) / 2;
    const size_t lds = std::max((size_t)K4 * 4 * tm_sb(nt), G ? (size_t)4 * np * 256 : (size_t)0) * 8;
    if (lds > 150 * 1024) return 0;
    const int64_t tiles = (m + 15) >> 4;
    if (tiles > (1 << 30)) return 0;
    const int nwg = (int)std::min<int64_t>((tiles + 3) / 4, 256);
    TallMul p{A, lda, B, ldb, Y, ldy, slab, m, K, n, (int)tiles};
    int rc;
    switch (nt) {
    case 1: rc = launch_tm<1>(p, G != nullptr, nwg, lds, st); break;
    case 2: rc = launch_tm<2>(p, G != nullptr, nwg, lds, st); break;
    case 3: rc = launch_tm<3>(p, G != nullptr, nwg, lds, st); break;
    default: rc = launch_tm<4>(p, G != nullptr, nwg, lds, st); break;
    }
    if (rc < 0) return rc;
    if (G) {
        hipLaunchKernelGGL(gram_reduce_kernel, dim3(np), dim3(1024), 0, st, slab, nwg, nt, n, G);
        TTSK_LAUNCH_CHECK();
    }
    return 1;
}

}  // namespace ttsk

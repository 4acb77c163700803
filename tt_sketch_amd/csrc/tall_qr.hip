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

__host__ __device__ constexpr int tm_sb(int nt) { return nt == 1 ? 16 : (nt <= 3 ? 48 : 80); }   // row stride of B in LDS: = 16 mod 32
__host__ __device__ inline int tm_sa(int K4) { return 4 * K4 + 2; }                                // row stride of an A tile: half of it odd

template <int NT, bool GRAM>
__global__ __launch_bounds__(512) void tall_mul_kernel(TallMul p)   // launched with 256 threads: the bound keeps the compiler off the AGPR file
{
    extern __shared__ double sm[];
    constexpr int SB = tm_sb(NT), NP = NT * (NT + 1) / 2;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, x16 = lane & 15, g = lane >> 4;
    const int K4 = (p.K + 3) >> 2, SA = tm_sa(K4);
    double *As = sm + (size_t)K4 * 4 * SB + (size_t)wv * 16 * SA;          // this wave's 16 x K tile of A
    int tile = blockIdx.x * 4 + wv;
    // A tile -> LDS, row by row (coalesced: 64 consecutive doubles per instruction); the columns K .. 4 K4 are zeroed
    auto stage_a = [&](int t) {
        double v[16][2];
        if (p.K <= 128) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = (int64_t)t * 16 + r;
                const double *ar = p.A + (row < p.m ? row : p.m - 1) * p.lda;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int k = lane + 64 * h;
                    v[r][h] = ar[k < p.K ? k : p.K - 1];
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int k = lane + 64 * h;
                    if (k < 4 * K4) As[r * SA + k] = k < p.K ? v[r][h] : 0.0;
                }
        } else {
#pragma nounroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = (int64_t)t * 16 + r;
                const double *ar = p.A + (row < p.m ? row : p.m - 1) * p.lda;
#pragma nounroll
                for (int k = lane; k < 4 * K4; k += 64) As[r * SA + k] = k < p.K ? ar[k] : 0.0;
            }
        }
    };
    if (tile < p.tiles) stage_a(tile);
    {
        // B (K x n) -> LDS rows of SB doubles, zero padded; sixteen loads in flight per thread
        const int total = K4 * 4 * SB;
        for (int e0 = tid; e0 < total; e0 += 256 * 16) {
            double v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int e = e0 + 256 * u, k = e / SB, c = e - k * SB;
                const bool ok = e < total && k < p.K && c < p.n;
                v[u] = ok ? p.B[(int64_t)k * p.ldb + c] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u)
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
        const double *ap = As + x16 * SA + g, *bp = sm + g * SB + x16;

        for (int kb = 0; kb < K4; ++kb) {
            const double a = ap[4 * kb];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = mfma16(a, bp[4 * kb * SB + 16 * t], acc[t]);
        }
        if ((int64_t)tile * 16 + 16 > p.m) {                                         // the last tile: rows beyond m are not there
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if ((int64_t)tile * 16 + g + 4 * j >= p.m) acc[t][j] = 0.0;
        }
        if (tile + gridDim.x * 4 < p.tiles) stage_a(tile + gridDim.x * 4);          // (this wave alone reads and writes its tile: program order suffices)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t r = (int64_t)tile * 16 + g + 4 * j;
                const int c = 16 * t + x16;
                if (r < p.m && c < p.n) p.Y[r * p.ldy + c] = acc[t][j];
            }
        if (GRAM) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int q = 0;
#pragma unroll
                for (int t1 = 0; t1 < NT; ++t1)
#pragma unroll
                    for (int t2 = t1; t2 < NT; ++t2, ++q) gacc[q] = mfma16(acc[t1][j], acc[t2][j], gacc[q]);
            }
        }
    }
    if (GRAM) {
        __syncthreads();                                   // every wave is done with B and its A tile
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

// G (n x n, symmetric, full) = sum over the workgroups of their partial tiles.  Block (x, pair): 32 consecutive
// elements of the pair's tile, 32 thread groups each summing every 32nd partial (all its loads in flight at once),
// then a fixed-order sum of the groups in LDS.
__global__ __launch_bounds__(1024) void gram_reduce_kernel(const double *__restrict__ slab, int nwg, int nt, int n, double *__restrict__ G)
{
    __shared__ double part[32][33];
    const int np = nt * (nt + 1) / 2, q = blockIdx.y, x = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int e = blockIdx.x * 32 + x;
    double acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = 0.0;
    for (int w0 = grp; w0 < nwg; w0 += 256) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int w = w0 + 32 * u;
            if (w < nwg) acc[u] += slab[((size_t)w * np + q) * 256 + e];
        }
    }
    part[grp][x] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    __syncthreads();
    if (grp) return;
    double v = 0.0;
#pragma unroll
    for (int k = 0; k < 32; ++k) v += part[k][x];
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
    const size_t lds = std::max((size_t)K4 * 4 * tm_sb(nt) + (size_t)4 * 16 * tm_sa(K4), G ? (size_t)4 * np * 256 : (size_t)0) * 8;
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
        hipLaunchKernelGGL(gram_reduce_kernel, dim3(8, np), dim3(1024), 0, st, slab, nwg, nt, n, G);
        TTSK_LAUNCH_CHECK();
    }
    return 1;
}

}  // namespace ttsk

// Y (m x n) = A (m x K) B (K x n), contiguous row-major, and (G != NULL) G (n x n) = Y^T Y: the fused kernel behind
// ttsk_tt_orth_sketch's CholeskyQR2 steps, exported for the parity tests and the profiling scripts.
extern "C" int ttsk_tall_mul(const double *A, int64_t m, int64_t K, const double *B, int64_t n, double *Y, double *G, int stream)
{
    using namespace ttsk;
    TTSK_STREAM(st, stream);
    TTSK_ARG(A && B && Y && m >= 1 && K >= 1 && n >= 1, "ttsk_tall_mul: bad argument");
    double *slab = nullptr;
    if (G) {
        slab = (double *)scratch(stream, SCRATCH_MISC, tall_mul_ws_elems(m, (int)n) * 8);
        if (!slab) return TTSK_ERR_HIP;
    }
    const int rc = (n <= 64 && K <= (1 << 20)) ? tall_mul(A, K, (int)K, B, n, Y, n, m, (int)n, G, slab, st) : 0;
    if (rc < 0) return rc;
    if (rc == 0) { set_error("ttsk_tall_mul: (%lld x %lld) (%lld x %lld) is outside the kernel", (long long)m, (long long)K, (long long)K, (long long)n); return TTSK_ERR_UNSUPPORTED; }
    return TTSK_OK;
}


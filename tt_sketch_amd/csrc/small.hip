// Small products (Omega_mu = L^T R, the first / last mode of the chains, Psi_0): a few MFLOP each,
// pure latency.  One wave per 16 x 32 output tile, operands straight from memory into MFMA
// registers for the whole K (ring of 8 k-blocks in flight), no LDS and no barrier; up to
// SK_MAXB problems of one shape (the tensors of a batch, or a uniformly strided batch) per launch.
// Long contractions (K >= 128: Omega of a sum of 32 terms has K = 640) are cut over 2, 4 or 8 waves of a
// workgroup, the partial tiles summed in wave order through LDS.
// tensor_train_sketch.py:8-19 (Omega), tensor_train_drm.py:79-88 (first mode) in the reference.
#include <cstdlib>
#include "skinny.h"

#ifndef SMALL_DEPTH
#define SMALL_DEPTH 16       // k-blocks in flight per wave (operands written by the previous kernel come from another XCD's L2: ~2 us a round trip)
#endif

namespace ttsk {

struct SmallG {
    const double *A[SK_MAXB], *B[SK_MAXB];
    double *C[SK_MAXB];
    int nb, M, N, K, tiles_m, tiles_n;   // tiles_n counts 32-column tiles
    int64_t a_m, a_k, b_k, b_n, c_m, c_n;
    int64_t a_extent, b_extent, c_extent;
    double alpha;
    int accumulate;
};

constexpr int SG_D = SMALL_DEPTH;

__global__ __launch_bounds__(512) void small_gemm_kernel(SmallG a)
{
    __shared__ double part[7][64][9];                 // partial tiles of waves 1..7 (padded rows)
    const int lane = threadIdx.x & 63, x16 = lane & 15, kq = lane >> 4;
    const int ws = threadIdx.x >> 6, nws = blockDim.x >> 6;
    int w = blockIdx.x;
    const int nt = w % a.tiles_n;
    w /= a.tiles_n;
    const int mt = w % a.tiles_m, prob = w / a.tiles_m;
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(uniform_ptr(a.A[prob]), a.a_extent * 8);
    const __amdgpu_buffer_rsrc_t rb = make_rsrc(uniform_ptr(a.B[prob]), a.b_extent * 8);
    const __amdgpu_buffer_rsrc_t rc = make_rsrc(uniform_ptr(a.C[prob]), a.c_extent * 8);
    const int row = 16 * mt + x16, col0 = 32 * nt + x16, col1 = col0 + 16;
    const uint32_t oa = row < a.M ? (uint32_t)(((int64_t)row * a.a_m + (int64_t)kq * a.a_k) * 8) : OOB_OFF;
    const uint32_t ob0 = col0 < a.N ? (uint32_t)(((int64_t)col0 * a.b_n + (int64_t)kq * a.b_k) * 8) : OOB_OFF;
    const uint32_t ob1 = col1 < a.N ? (uint32_t)(((int64_t)col1 * a.b_n + (int64_t)kq * a.b_k) * 8) : OOB_OFF;
    const uint32_t sa = (uint32_t)(4 * a.a_k * 8), sb = (uint32_t)(4 * a.b_k * 8);
    const int nkb_lane = (a.K - kq + 3) >> 2, KB = (a.K + 3) >> 2;
    // this wave's stretch of the k-blocks
    const int per = (KB + nws - 1) / nws, kb_beg = ws * per, kb_end = kb_beg + per < KB ? kb_beg + per : KB;
    const int ITER = kb_end > kb_beg ? (kb_end - kb_beg + SG_D - 1) / SG_D : 0;

    double ra_[SG_D], rb0[SG_D], rb1[SG_D];
    int kb_load = kb_beg;
    auto issue = [&](int d) {
        const bool ok = kb_load < nkb_lane && kb_load < kb_end;
        ra_[d] = ld8(ra, (ok && oa != OOB_OFF) ? oa + (uint32_t)kb_load * sa : OOB_OFF, 0);
        rb0[d] = ld8(rb, (ok && ob0 != OOB_OFF) ? ob0 + (uint32_t)kb_load * sb : OOB_OFF, 0);
        rb1[d] = ld8(rb, (ok && ob1 != OOB_OFF) ? ob1 + (uint32_t)kb_load * sb : OOB_OFF, 0);
        ++kb_load;
    };
#pragma unroll
    for (int d = 0; d < SG_D; ++d) issue(d);
    double acc[2][4];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[q][t] = 0.0;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int d = 0; d < SG_D; ++d) {
            double rA[4];
            rot4(ra_[d], rA);
            const double b0 = rb0[d], b1 = rb1[d];
            issue(d);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc[0][t] = mfma4(rA[t], b0, acc[0][t]);
                acc[1][t] = mfma4(rA[t], b1, acc[1][t]);
            }
        }
    }
    if (nws > 1) {                                     // partial tiles meet in wave 0, summed in wave order
        if (ws > 0) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int t = 0; t < 4; ++t) part[ws - 1][lane][4 * q + t] = acc[q][t];
        }
        __syncthreads();
        if (ws > 0) return;
        for (int o = 1; o < nws; ++o)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[q][t] += part[o - 1][lane][4 * q + t];
    }
    // rotated A: acc[q][t] at lane (i = l>>4, beta = (l>>2)&3, j4 = l&3) is D[4((beta+t)&3) + i][4 beta + j4]
    const int i = lane >> 4, beta = (lane >> 2) & 3, j4 = lane & 3;
    uint32_t off[2][4];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int m = 16 * mt + 4 * ((beta + t) & 3) + i, n = 32 * nt + 16 * q + 4 * beta + j4;
            off[q][t] = (m < a.M && n < a.N) ? (uint32_t)(((int64_t)m * a.c_m + (int64_t)n * a.c_n) * 8) : OOB_OFF;
            acc[q][t] *= a.alpha;
        }
    if (a.accumulate) {
        double old[2][4];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int t = 0; t < 4; ++t) old[q][t] = ld8(rc, off[q][t], 0);
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[q][t] += old[q][t];
    }
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int t = 0; t < 4; ++t) st8(rc, off[q][t], acc[q][t]);
}

static int small_mode()
{
    static int m = [] {
        const char *e = getenv("TTSK_SMALL");
        return e ? atoi(e) : 1;
    }();
    return m;
}

// d: single contracted index (Ko == 1); nb pointer triples, or one triple with a uniformly strided d.batch
int small_try_batch(const ttsk_gemm_desc &d, int nb, const double *const *A, const double *const *B, double *const *C,
                    int stream, hipStream_t st)
{
    (void)stream;
    if (!small_mode() || d.Ko != 1) return 0;
    const int64_t K = d.Ki;
    int64_t count = nb;
    if (d.batch > 1) {
        if (nb != 1) return 0;
        count = d.batch;
    }
    if (count < 1 || count > SK_MAXB) return 0;
    if (K < 1 || K > 1024 || d.M > 512 || d.N > 512) return 0;
    if (2.0 * d.M * d.N * K > 48e6) return 0;            // beyond a few MFLOP the tiled kernels win
    if (d.a_m < 0 || d.a_ki < 0 || d.b_ki < 0 || d.b_n < 0 || d.c_m < 0 || d.c_n < 0) return 0;
    if (d.batch > 1 && (d.a_b < 0 || d.b_b < 0 || d.c_b < 0)) return 0;
    SmallG g{};
    g.nb = (int)count;
    for (int b = 0; b < count; ++b) {
        g.A[b] = d.batch > 1 ? A[0] + b * d.a_b : A[b];
        g.B[b] = d.batch > 1 ? B[0] + b * d.b_b : B[b];
        g.C[b] = d.batch > 1 ? C[0] + b * d.c_b : C[b];
    }
    g.M = (int)d.M; g.N = (int)d.N; g.K = (int)K;
    g.tiles_m = (int)cdiv(d.M, 16);
    g.tiles_n = (int)cdiv(d.N, 32);
    g.a_m = d.a_m; g.a_k = d.a_ki; g.b_k = d.b_ki; g.b_n = d.b_n; g.c_m = d.c_m; g.c_n = d.c_n;
    g.a_extent = (d.M - 1) * d.a_m + (K - 1) * d.a_ki + 1;
    g.b_extent = (d.N - 1) * d.b_n + (K - 1) * d.b_ki + 1;
    g.c_extent = (d.M - 1) * d.c_m + (d.N - 1) * d.c_n + 1;
    if ((g.a_extent + 64 * d.a_ki) * 8 >= (1ll << 32) - 64 || (g.b_extent + 64 * d.b_ki) * 8 >= (1ll << 32) - 64 ||
        g.c_extent * 8 >= (1ll << 32) - 64)
        return 0;
    g.alpha = d.alpha;
    g.accumulate = d.accumulate;
    const bool prof = prof_on();
    if (prof) prof_open(st, 2.0 * count * (double)d.M * (double)d.N * (double)K, 5, 0, false, false);
    static const int ks_max = [] { const char *e = getenv("TTSK_SMALL_KSPLIT"); return e ? atoi(e) : 8; }();
    int ksplit = K >= 512 ? 8 : (K >= 256 ? 4 : (K >= 128 ? 2 : 1));
    if (ksplit > ks_max) ksplit = ks_max;
    hipLaunchKernelGGL(small_gemm_kernel, dim3((unsigned)(count * g.tiles_m * g.tiles_n)), dim3(64 * ksplit), 0, st, g);
    if (prof) prof_close(st);
    TTSK_LAUNCH_CHECK();
    return 1;
}

}  // namespace ttsk

// Psi_mu = T_mu R_mu (tensor_train_sketch.py:28-34) as a streamed x small product in the style of
// chain_fused.h:
//   C[j][a] (+)= sum_c S[j][c] W[c][a]       j ~ 10^4 rows streamed once; K1 as the LDS allows (rank-150 TTs),
//                                            A beyond 112 in column chunks dealt over workgroups
// W sits in LDS in the pair-interleaved image of chain_fused.h (conflict-free fragment reads), every one of
// the 8 waves of a workgroup walks its own 16-row tiles of S: fragments straight from memory through a
// register ring, the W fragments of the next k-block requested before this one's matrix instructions,
// 25 k-blocks per straight-line run, partial tiles of A as 4-wide strips (v_mfma_f64_4x4x4).  With
// mfma(A = S fragment, B = W fragment) a result register holds 16 consecutive a of one row: 128-byte stores.
// No barrier after W is staged.  The older skinny_s_kernel (skinny.h) covers the same product for every
// stride pattern; this one takes the contiguous case the TT sketch produces.
#pragma once
#include "skinny.h"

namespace ttsk {

struct StreamSmall {
    const double *S[SK_MAXB];
    const double *W[SK_MAXB];
    double *C[SK_MAXB];
    int nb, wpp;                 // problems, row-tile groups per problem
    int nac, ac;                 // column chunks of A (one workgroup each: disjoint outputs), columns per chunk
    int J, K1, A;
    int64_t s_j, w_c, c_j;       // row strides (elements): S rows, W rows, C rows; columns contiguous
    int64_t s_extent, c_extent;
    int AP;
    int accumulate;
};

template <int NF, int STR, int D, int UNR = 5 * D>
__global__ __launch_bounds__(512) void stream_small_kernel(StreamSmall a)
{
    static_assert(UNR % D == 0, "the unrolled body must keep the ring slots static");
    extern __shared__ double ss_lds[];
    double *Wl = ss_lds;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int x16 = lane & 15, kq = lane >> 4;
    const int prob = blockIdx.x / (a.wpp * a.nac), unit = blockIdx.x - prob * (a.wpp * a.nac);
    const int g = unit / a.nac, a0 = (unit - g * a.nac) * a.ac;
    const int cnt = a.A - a0 < a.ac ? a.A - a0 : a.ac;       // columns of this chunk
    const int KB1 = ((a.K1 + 3) / 4 + UNR - 1) / UNR * UNR, AP = a.AP;      // padded to whole runs: zero rows of the W image
    {
        const double *Wp = uniform_ptr(a.W[prob]);
        const __amdgpu_buffer_rsrc_t rw = make_rsrc(Wp, ((int64_t)(a.K1 - 1) * a.w_c + a.A) * 8);
        const int total = 4 * (KB1 + 1) * AP;          // one k-block of zeros behind the image: the last look-ahead
        constexpr int BATCH = 10;
        for (int e0 = tid; e0 < total; e0 += 512 * BATCH) {
            double v[BATCH];
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int e = e0 + 512 * u;
                const int c = e / AP, col = e - c * AP;
                v[u] = ld8(rw, (e < total && c < a.K1 && col < cnt) ? (uint32_t)(((int64_t)c * a.w_c + a0 + col) * 8) : OOB_OFF, 0);
            }
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int e = e0 + 512 * u;
                const int c = e / AP, col = e - c * AP;
                if (e < total) Wl[(c >> 1) * 2 * AP + 2 * col + (c & 1)] = v[u];
            }
        }
    }
    const int ntiles = (a.J + 15) >> 4;
    const int tstep = a.wpp * 8;                       // tiles between two visits of this wave
    int tile = g * 8 + w;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(uniform_ptr(a.S[prob]), a.s_extent * 8);
    const __amdgpu_buffer_rsrc_t rc = make_rsrc(uniform_ptr(a.C[prob]), a.c_extent * 8);
    // this lane's (row, k) offset inside a tile; the (tile, k-block) part is a scalar offset.  No masks: k
    // beyond K1 meets zero rows of the W image, rows beyond J are never stored, past the end reads 0.
    const uint32_t slane = (uint32_t)(((int64_t)x16 * a.s_j + kq) * 8);
    const uint32_t tstride = __builtin_amdgcn_readfirstlane((uint32_t)(16 * a.s_j * 8));
    auto sload = [&](uint32_t so) -> double { return ld8(rs, slane, __builtin_amdgcn_readfirstlane(so)); };
    const int ITER = KB1 / UNR;
    double ring[D];
#pragma unroll
    for (int d = 0; d < D; ++d) ring[d] = sload((uint32_t)tile * tstride + (uint32_t)d * 32u);
    const int wl_lane = (kq >> 1) * 2 * AP + 2 * x16 + (kq & 1);
    const int ws_lane = (kq >> 1) * 2 * AP + 2 * (16 * NF + (x16 & 3)) + (kq & 1);
    __syncthreads();                                   // W staged; from here on the waves run free

    for (; tile < ntiles; tile += tstep) {
        v4d acc[NF ? NF : 1];
        double accs[STR ? STR : 1];
#pragma unroll
        for (int p = 0; p < NF; ++p)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[p][t] = 0.0;
#pragma unroll
        for (int q = 0; q < STR; ++q) accs[q] = 0.0;
        double bf[NF ? NF : 1], sf[STR ? STR : 1];
        auto wfetch = [&](const double *wrun, int u, double (&f)[NF ? NF : 1], double (&gq)[STR ? STR : 1]) {
#pragma unroll
            for (int p = 0; p < NF; ++p) f[p] = wrun[wl_lane + u * 4 * AP + 32 * p];
#pragma unroll
            for (int q = 0; q < STR; ++q) gq[q] = wrun[ws_lane + u * 4 * AP + 8 * q];
        };
        wfetch(Wl, 0, bf, sf);
        uint32_t so = (uint32_t)tile * tstride + (uint32_t)D * 32u;
        auto kblock = [&](const double *wrun, int u, bool wrap) {
            const int d = u % D;
            double bfn[NF ? NF : 1], sfn[STR ? STR : 1];
            wfetch(wrun, u + 1, bfn, sfn);
            const double af = ring[d];
#pragma unroll
            for (int p = 0; p < NF; ++p) acc[p] = mfma16(af, bf[p], acc[p]);
#pragma unroll
            for (int q = 0; q < STR; ++q) accs[q] = mfma4(af, sf[q], accs[q]);
            if (wrap) so = (uint32_t)(tile + tstep) * tstride;       // the first fragments of this wave's next tile
            ring[d] = sload(so);
            so += 32u;
#pragma unroll
            for (int p = 0; p < NF; ++p) bf[p] = bfn[p];
#pragma unroll
            for (int q = 0; q < STR; ++q) sf[q] = sfn[q];
        };
        for (int it = 0; it < ITER; ++it) {
            const double *wrun = Wl + it * UNR * 4 * AP;
            const bool last = it == ITER - 1;
#pragma unroll
            for (int u = 0; u < UNR; ++u) kblock(wrun, u, last && u + D == UNR);
        }
        // register t of tile p: row 16 tile + 4 t + kq, columns 16 p .. 16 p + 15
        const int j0 = 16 * tile;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int j = j0 + 4 * t + kq;
            const uint32_t ro = (uint32_t)(((int64_t)j * a.c_j + a0 + x16) * 8);
#pragma unroll
            for (int p = 0; p < NF; ++p) {
                const uint32_t off = (j < a.J && 16 * p + x16 < cnt) ? ro + 128u * p : OOB_OFF;
                double v = acc[p][t];
                if (a.accumulate) v += ld8(rc, off, 0);
                st8(rc, off, v);
            }
        }
#pragma unroll
        for (int q = 0; q < STR; ++q) {
            // 4x4x4 result: lane (i = l >> 4, beta = (l >> 2) & 3, c = l & 3) holds row 4 beta + i, column c of the strip
            const int j = j0 + 4 * ((lane >> 2) & 3) + kq, col = 16 * NF + 4 * q + (lane & 3);
            const uint32_t off = (j < a.J && col < cnt) ? (uint32_t)(((int64_t)j * a.c_j + a0 + col) * 8) : OOB_OFF;
            double v = accs[q];
            if (a.accumulate) v += ld8(rc, off, 0);
            st8(rc, off, v);
        }
    }
}

// The same product SUMMED over nb terms into ONE output: C[j][a] (+)= sum_b sum_c S_b[j][c] W_b[c][a] -- Psi_mu of a
// sum of TTs (sketch_dispatch.py:111-139: the TensorSum loop) without the nb per-term Psi arrays and their sum.
// A wave owns ONE 16-row tile and one column chunk for the whole launch and keeps its accumulators across the terms;
// the (term, k-block) pairs form one long contraction: the ring of S fragments runs on from term to term (the terms'
// operands are equally spaced in the driver's workspace: one descriptor, a term stride), the W image of term b + 1
// is fetched into registers before the k-loop of term b and written to the other LDS buffer after it (one barrier
// per term).  Column chunks (>= 2: two W images must fit) of a row group share an XCD, so S is read from HBM once.
struct StreamSmallSum {
    const double *S, *W;         // term 0; term b at + b * s_b / + b * w_b
    double *C;
    int nb, groups;              // terms, row-tile groups (8 tiles each)
    int nac, ac;
    int J, K1, A;
    int64_t s_j, w_c, c_j, s_b, w_b;
    int64_t s_extent, w_extent, c_extent;
    int AP, xcd_map;
    int accumulate;
};

template <int NF, int STR, int D, int UNR = 5 * D>
__global__ __launch_bounds__(512) void stream_small_sum_kernel(StreamSmallSum a)
{
    static_assert(UNR % D == 0, "the unrolled body must keep the ring slots static");
    extern __shared__ double ss_lds[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int x16 = lane & 15, kq = lane >> 4;
    const int units = a.groups * a.nac;
    int unit = blockIdx.x;
    if (a.xcd_map) {             // blocks b and b + 8 share an XCD: consecutive units (the chunks of a row group) on one XCD
        const int upx = units >> 3, xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        unit = xcd * upx + j;
    }
    const int g = unit / a.nac, a0 = (unit - g * a.nac) * a.ac;
    const int cnt = a.A - a0 < a.ac ? a.A - a0 : a.ac;
    const int KB1 = ((a.K1 + 3) / 4 + UNR - 1) / UNR * UNR, AP = a.AP;
    const int img = 4 * (KB1 + 1) * AP;                // doubles per W image (one k-block of zeros behind it)
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.W, a.w_extent * 8);
    constexpr int BATCH = 12;                          // W elements per thread and image: img <= 512 * BATCH
    double wv[BATCH];
    auto w_fetch = [&](int b) {
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int e = tid + 512 * u;
            const int c = e / AP, col = e - c * AP;
            wv[u] = ld8(rw, (e < img && c < a.K1 && col < cnt) ? (uint32_t)(((int64_t)b * a.w_b + (int64_t)c * a.w_c + a0 + col) * 8) : OOB_OFF, 0);
        }
    };
    auto w_store = [&](double *Wl) {
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int e = tid + 512 * u;
            const int c = e / AP, col = e - c * AP;
            if (e < img) Wl[(c >> 1) * 2 * AP + 2 * col + (c & 1)] = wv[u];
        }
    };
    w_fetch(0);
    w_store(ss_lds);
    const int ntiles = (a.J + 15) >> 4;
    const int tile = g * 8 + w;
    const bool live = tile < ntiles;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.S, a.s_extent * 8);
    const __amdgpu_buffer_rsrc_t rc = make_rsrc(a.C, a.c_extent * 8);
    const uint32_t slane = live ? (uint32_t)((((int64_t)tile * 16 + x16) * a.s_j + kq) * 8) : OOB_OFF;
    const uint32_t bstride = __builtin_amdgcn_readfirstlane((uint32_t)(a.s_b * 8));
    auto sload = [&](uint32_t so) -> double { return ld8(rs, slane, __builtin_amdgcn_readfirstlane(so)); };
    const int ITER = KB1 / UNR;
    double ring[D];
#pragma unroll
    for (int d = 0; d < D; ++d) ring[d] = sload((uint32_t)d * 32u);
    const int wl_lane = (kq >> 1) * 2 * AP + 2 * x16 + (kq & 1);
    const int ws_lane = (kq >> 1) * 2 * AP + 2 * (16 * NF + (x16 & 3)) + (kq & 1);
    v4d acc[NF ? NF : 1];
    double accs[STR ? STR : 1];
#pragma unroll
    for (int p = 0; p < NF; ++p)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[p][t] = 0.0;
#pragma unroll
    for (int q = 0; q < STR; ++q) accs[q] = 0.0;
    __syncthreads();
    for (int b = 0; b < a.nb; ++b) {
        const double *Wl = ss_lds + (size_t)(b & 1) * img;
        if (b + 1 < a.nb) w_fetch(b + 1);              // in flight under this term's matrix instructions
        double bf[NF ? NF : 1], sf[STR ? STR : 1];
        auto wfetch = [&](const double *wrun, int u, double (&f)[NF ? NF : 1], double (&gq)[STR ? STR : 1]) {
#pragma unroll
            for (int p = 0; p < NF; ++p) f[p] = wrun[wl_lane + u * 4 * AP + 32 * p];
#pragma unroll
            for (int q = 0; q < STR; ++q) gq[q] = wrun[ws_lane + u * 4 * AP + 8 * q];
        };
        wfetch(Wl, 0, bf, sf);
        uint32_t so = (uint32_t)b * bstride + (uint32_t)D * 32u;
        auto kblock = [&](const double *wrun, int u, bool wrap) {
            const int d = u % D;
            double bfn[NF ? NF : 1], sfn[STR ? STR : 1];
            wfetch(wrun, u + 1, bfn, sfn);
            const double af = ring[d];
#pragma unroll
            for (int p = 0; p < NF; ++p) acc[p] = mfma16(af, bf[p], acc[p]);
#pragma unroll
            for (int q = 0; q < STR; ++q) accs[q] = mfma4(af, sf[q], accs[q]);
            if (wrap) so = (uint32_t)(b + 1) * bstride;              // the first fragments of the next term
            ring[d] = sload(so);
            so += 32u;
#pragma unroll
            for (int p = 0; p < NF; ++p) bf[p] = bfn[p];
#pragma unroll
            for (int q = 0; q < STR; ++q) sf[q] = sfn[q];
        };
        for (int it = 0; it < ITER; ++it) {
            const double *wrun = Wl + it * UNR * 4 * AP;
            const bool last = it == ITER - 1;
#pragma unroll
            for (int u = 0; u < UNR; ++u) kblock(wrun, u, last && u + D == UNR);
        }
        if (b + 1 < a.nb) w_store(ss_lds + (size_t)((b + 1) & 1) * img);   // image (b + 1) & 1 was last read in term b - 1
        __syncthreads();
    }
    if (!live) return;
    const int j0 = 16 * tile;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int j = j0 + 4 * t + kq;
        const uint32_t ro = (uint32_t)(((int64_t)j * a.c_j + a0 + x16) * 8);
#pragma unroll
        for (int p = 0; p < NF; ++p) {
            const uint32_t off = (j < a.J && 16 * p + x16 < cnt) ? ro + 128u * p : OOB_OFF;
            double v = acc[p][t];
            if (a.accumulate) v += ld8(rc, off, 0);
            st8(rc, off, v);
        }
    }
#pragma unroll
    for (int q = 0; q < STR; ++q) {
        const int j = j0 + 4 * ((lane >> 2) & 3) + kq, col = 16 * NF + 4 * q + (lane & 3);
        const uint32_t off = (j < a.J && col < cnt) ? (uint32_t)(((int64_t)j * a.c_j + a0 + col) * 8) : OOB_OFF;
        double v = accs[q];
        if (a.accumulate) v += ld8(rc, off, 0);
        st8(rc, off, v);
    }
}

// C (J x A) (+)= sum_b S_b W_b, the nb terms equally spaced: 1 = launched, 0 = shape not covered, < 0 = error
struct StreamSmallSumArgs {
    int nb, J, K1, A;
    const double *S;             // term 0 (J x K1, row stride s_j); term b at S + b * s_b
    int64_t s_j, s_b;
    const double *W;             // term 0 (K1 x A, row stride w_c); term b at W + b * w_b
    int64_t w_c, w_b;
    double *C;
    int64_t c_j;
    int accumulate;
};
int stream_small_sum_try(const StreamSmallSumArgs &c, int stream, hipStream_t st);

// 1 = launched, 0 = shape not covered, < 0 = error
struct StreamSmallArgs {
    int nb, J, K1, A;
    const double *const *S;      // nb operands (J x K1), row stride s_j, columns contiguous
    int64_t s_j;
    const double *const *W;      // nb small operands (K1 x A), row stride w_c
    int64_t w_c;
    double *const *C;            // nb results (J x A), row stride c_j
    int64_t c_j;
    int accumulate;
};
int stream_small_try(const StreamSmallArgs &c, int stream, hipStream_t st);

}  // namespace ttsk

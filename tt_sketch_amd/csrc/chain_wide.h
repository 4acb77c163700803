// The fused chain step (chain_fused.h) for every shape the first kernel's single-image plan does not hold:
//
//   reference: tensor_train_drm.py:81-87   L_mu[l,m] = sum_{i,j,k} L_{mu-1}[i,j] X_mu[i,k,l] D_mu[j,k,m]
//
// Same two phases per slice k of the mode, T_k never leaving the registers,
//   phase A   T_k^T[a][j] = sum_c W[c][a] G_k[j][c]          phase B   Out[j][a'] += sum_a T_k[j][a] E_k[a][a']
// but the contracted DRM rank a is cut into CHUNKS dealt over workgroups: a workgroup holds the columns
// [a0, a0 + ac) of W and the rows [a0, a0 + ac) of every E_k, computes that part of T_k and adds its
// contribution to ALL of Out -- nothing is computed twice, the partial results meet in the slab reduce that the
// slice ranges need anyway.  With that
//   * the W image is K1 x ac instead of K1 x A: TT ranks beyond 128 and DRM ranks beyond 112 fit the LDS
//     (rank-150 TT against rank-110 DRM: 72 KB + 49 KB),
//   * the tile structures of A (chunk: NQF tiles + STRQ strips) and A2 (output: NNF + STRN) are independent,
//   * a single tensor offers n x chunks workgroups instead of n.
// Rows: a wave owns one or TWO 16-row tiles of the output (both use the same W / E fragments: half the LDS
// reads per matrix instruction); the tiles are dealt so that the four SIMDs of the CU carry equal shares --
// 150 rows = 10 tiles as 3 + 3 + 2 + 2 -- with the loader wave on the lightest SIMD (table built on the host).
// Few rows per tensor (TT rank 20 against DRM rank 100: two row tiles): one tensor per workgroup would leave six of
// eight waves without rows; a workgroup then serves up to seven tensors of the batch at once (one W image per tensor,
// a wave per tensor with both its tiles) and every E_k it brings into LDS is used by all of them.
// Odd DRM ranks: the E image is filled in 16-byte units from 8-byte-aligned rows; the unit behind the last
// column of a row takes the first element of the next slice along (it meets an output column that is never
// stored), and the one unit that would reach past the end of the core is patched with an 8-byte load.
#pragma once
#include "chain_fused.h"

namespace ttsk {

struct ChainWide {
    const double *W[SK_MAXB];
    const double *X[SK_MAXB];
    double *T[SK_MAXB];          // WT: T[a][k][j] (A x n x J contiguous)
    const double *E;
    double *slab;                // [problem][unit = slice range x chunk][J][A2]
    int nb, wpp, nac, n;         // problems, slice ranges per problem, chunks of A, slices (mode size)
    int K1, A, A2, J;
    int64_t w_c;                 // row stride of W (elements); columns contiguous
    int64_t x_j, x_k, x_c;       // element strides of X
    int64_t x_extent, t_extent;  // elements addressable from the bases
    int ac;                      // columns of W / rows of E per chunk (<= 16 NQF + 4 STRQ)
    int A2P;                     // A2 rounded up to even: 16-byte units per section of the E image
    int ebase;                   // offset (doubles) of the E image in LDS
    int eunits;                  // 16-byte units of one E image (multiple of 64)
    int ebuf2;                   // 1: two E images (slice k in image k & 1)
    int xcd_map;                 // 1: workgroups of one (slice range, chunk) share an XCD (E_k from one L2)
    int loader;                  // the wave that only feeds E
    signed char tile0[8], tile1[8];   // row tiles of each wave, -1 = none
    // Few rows per tensor (J <= 64: the rank-20 terms of a sum): a workgroup serves `tpw` tensors at once -- one W image
    // each (`wimg` doubles apart), wave w works for tensor slot[w] of the group -- and they share every E_k it loads.
    int tpw, wimg;
    signed char slot[8];
};

// One compute wave: MT row tiles (t[0], t[1]).
template <int NQF, int STRQ, int NNF, int STRN, bool WT, int UNR, int MT>
__device__ __forceinline__ void cw_compute(const ChainWide &a, const double *Wl, const double *El, const int prob, const int unit,
                                           const int a0, const int cnt, const int k_beg, const int k_end, const int t0, const int t1)
{
    constexpr int D = 5;
    constexpr int AP = 16 * NQF + 4 * STRQ;
    static_assert(UNR % D == 0, "the unrolled body must keep the ring slots static");
    const int lane = threadIdx.x & 63, x16 = lane & 15, kq = lane >> 4;
    const int KB1 = ((a.K1 + 3) / 4 + UNR - 1) / UNR * UNR;
    const int A2P = a.A2P;
    int j0[MT];
    bool jok[MT];
    uint32_t xlane[MT];
    j0[0] = 16 * t0;
    if constexpr (MT == 2) j0[1] = 16 * t1;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        jok[m] = j0[m] + x16 < a.J;
        xlane[m] = (uint32_t)(((int64_t)(j0[m] + x16) * a.x_j + (int64_t)kq * a.x_c) * 8);
    }
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(uniform_ptr(a.X[prob]), a.x_extent * 8);
    const uint32_t xstep = __builtin_amdgcn_readfirstlane((uint32_t)(4 * a.x_c * 8));
    const uint32_t kstep = __builtin_amdgcn_readfirstlane((uint32_t)(a.x_k * 8));
    const int ITER = KB1 / UNR;
    // no masks on the ring loads: k beyond K1 meets zero rows of the W image, rows beyond J are never stored, a
    // prefetch past the last slice is unused, whatever lies beyond the core reads as 0 (descriptor range)
    double ring[MT][D];
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int m = 0; m < MT; ++m)
            ring[m][d] = ld8(rx, xlane[m], __builtin_amdgcn_readfirstlane((uint32_t)k_beg * kstep + (uint32_t)d * xstep));

    v4d acc2[MT][NNF ? NNF : 1];
    double acc2s[MT][STRN ? STRN : 1];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int p = 0; p < NNF; ++p)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc2[m][p][t] = 0.0;
#pragma unroll
        for (int q = 0; q < STRN; ++q) acc2s[m][q] = 0.0;
    }

    const int wl_lane = (kq >> 1) * 2 * AP + 2 * x16 + (kq & 1);
    const int ws_lane = (kq >> 1) * 2 * AP + 2 * (16 * NQF + (x16 & 3)) + (kq & 1);
    const int el_lane = (kq >> 1) * 2 * A2P + 4 * (x16 >> 1) + 2 * (kq & 1) + (x16 & 1);
    const int es_col = 16 * NNF + (x16 & 3);
    const int es_lane = (kq >> 1) * 2 * A2P + 2 * (kq & 1);

    __amdgpu_buffer_rsrc_t rt;
    if constexpr (WT) rt = make_rsrc(uniform_ptr(a.T[prob]), a.t_extent * 8);

    for (int k = k_beg; k < k_end; ++k) {
        v4d acc1[MT][NQF ? NQF : 1];
        double acc1s[MT][STRQ ? STRQ : 1];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
            for (int p = 0; p < NQF; ++p)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc1[m][p][t] = 0.0;
#pragma unroll
            for (int q = 0; q < STRQ; ++q) acc1s[m][q] = 0.0;
        }
        // ---- phase A (look-ahead of the W fragments, ring slot reloaded after use: see chain_fused.h)
        double af[NQF ? NQF : 1], sf[STRQ ? STRQ : 1];
        auto wfetch = [&](const double *wrun, int u, double (&f)[NQF ? NQF : 1], double (&g)[STRQ ? STRQ : 1]) {
#pragma unroll
            for (int p = 0; p < NQF; ++p) f[p] = LDS_UNPAIRED(wrun[wl_lane + u * 4 * AP + 32 * p]);
#pragma unroll
            for (int q = 0; q < STRQ; ++q) g[q] = wrun[ws_lane + u * 4 * AP + 8 * q];
        };
        wfetch(Wl, 0, af, sf);
        uint32_t so = (uint32_t)k * kstep + (uint32_t)D * xstep;
        auto kblock = [&](const double *wrun, int u, bool wrap) {
            const int d = u % D;
            double afn[NQF ? NQF : 1], sfn[STRQ ? STRQ : 1];
            wfetch(wrun, u + 1, afn, sfn);
            double bf[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) bf[m] = ring[m][d];
#pragma unroll
            for (int p = 0; p < NQF; ++p)
#pragma unroll
                for (int m = 0; m < MT; ++m) acc1[m][p] = mfma16(af[p], bf[m], acc1[m][p]);
#pragma unroll
            for (int q = 0; q < STRQ; ++q)
#pragma unroll
                for (int m = 0; m < MT; ++m) acc1s[m][q] = mfma4(sf[q], bf[m], acc1s[m][q]);
            if (wrap) so = (uint32_t)(k + 1) * kstep;
            const uint32_t sou = __builtin_amdgcn_readfirstlane(so);
#pragma unroll
            for (int m = 0; m < MT; ++m) ring[m][d] = ld8(rx, xlane[m], sou);
            so += xstep;
#pragma unroll
            for (int p = 0; p < NQF; ++p) af[p] = afn[p];
#pragma unroll
            for (int q = 0; q < STRQ; ++q) sf[q] = sfn[q];
        };
        for (int it = 0; it < ITER; ++it) {
            const double *wrun = Wl + it * UNR * 4 * AP;
            const bool last = it == ITER - 1;
#pragma unroll
            for (int u = 0; u < UNR; ++u) kblock(wrun, u, last && u + D == UNR);
        }
        if constexpr (WT) {
            // T[a][k][j]: register t of tile p is row a = a0 + 16 p + 4 t + kq, 16 consecutive j per row
            const int64_t nJ = (int64_t)a.n * a.J;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const uint32_t tl = jok[m] ? (uint32_t)((((int64_t)a0 + kq) * nJ + (int64_t)k * a.J + j0[m] + x16) * 8) : OOB_OFF;
#pragma unroll
                for (int p = 0; p < NQF; ++p)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        st8(rt, (jok[m] && 16 * p + 4 * t + kq < cnt) ? tl + (uint32_t)((16 * p + 4 * t) * nJ * 8) : OOB_OFF, acc1[m][p][t]);
#pragma unroll
                for (int q = 0; q < STRQ; ++q)
                    st8(rt, (jok[m] && 16 * NQF + 4 * q + kq < cnt) ? tl + (uint32_t)((16 * NQF + 4 * q) * nJ * 8) : OOB_OFF, acc1s[m][q]);
            }
        }
        cf_barrier();                                  // B1: E_k is in LDS
        // ---- phase B: k-block kap = 4 p + t of T is register t of tile p
        {
            const double *eb = El + ((a.ebuf2 && (k & 1)) ? a.eunits * 2 : 0);
            double bf[NNF ? NNF : 1], bs[STRN ? STRN : 1];
            int el0 = el_lane, es0 = es_lane + 4 * (es_col >> 1) + (es_col & 1);
            asm volatile("" : "+v"(el0), "+v"(es0));
            auto efetch = [&](int kap, double (&f)[NNF ? NNF : 1], double (&g)[STRN ? STRN : 1]) {
#pragma unroll
                for (int nn = 0; nn < NNF; ++nn) f[nn] = LDS_UNPAIRED(eb[el0 + kap * 4 * A2P + 32 * nn]);
#pragma unroll
                for (int q = 0; q < STRN; ++q) g[q] = eb[es0 + kap * 4 * A2P + 8 * q];
            };
            double bf2[NNF ? NNF : 1], bs2[STRN ? STRN : 1];
            efetch(0, bf, bs);
            constexpr int KB2C = 4 * NQF + STRQ;
#pragma unroll
            for (int kap = 0; kap < KB2C; ++kap) {
                const int p = kap >> 2, t = kap & 3;
                double af2[MT];
#pragma unroll
                for (int m = 0; m < MT; ++m) af2[m] = p < NQF ? acc1[m][p < NQF ? p : 0][t] : acc1s[m][t < STRQ ? t : 0];
                if (kap & 1) {
                    if (kap + 1 < KB2C) efetch(kap + 1, bf, bs);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int nn = 0; nn < NNF; ++nn)
#pragma unroll
                        for (int m = 0; m < MT; ++m) acc2[m][nn] = mfma16(af2[m], bf2[nn], acc2[m][nn]);
#pragma unroll
                    for (int q = 0; q < STRN; ++q)
#pragma unroll
                        for (int m = 0; m < MT; ++m) acc2s[m][q] = mfma4(af2[m], bs2[q], acc2s[m][q]);
                } else {
                    if (kap + 1 < KB2C) efetch(kap + 1, bf2, bs2);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int nn = 0; nn < NNF; ++nn)
#pragma unroll
                        for (int m = 0; m < MT; ++m) acc2[m][nn] = mfma16(af2[m], bf[nn], acc2[m][nn]);
#pragma unroll
                    for (int q = 0; q < STRN; ++q)
#pragma unroll
                        for (int m = 0; m < MT; ++m) acc2s[m][q] = mfma4(af2[m], bs[q], acc2s[m][q]);
                }
            }
        }
        cf_barrier();                                  // B2: El may be overwritten
    }

    // ---- partial result of this workgroup: slab[problem][unit][j][a']
    double *slab = a.slab + ((int64_t)prob * a.wpp * a.nac + unit) * a.J * a.A2;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int nn = 0; nn < NNF; ++nn)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int j = j0[m] + 4 * t + kq, col = 16 * nn + x16;
                if (j < a.J && col < a.A2) slab[(int64_t)j * a.A2 + col] = acc2[m][nn][t];
            }
#pragma unroll
        for (int q = 0; q < STRN; ++q) {
            const int j = j0[m] + 4 * ((lane >> 2) & 3) + kq, col = 16 * NNF + 4 * q + (lane & 3);
            if (j < a.J && col < a.A2) slab[(int64_t)j * a.A2 + col] = acc2s[m][q];
        }
    }
}

template <int NQF, int STRQ, int NNF, int STRN, bool WT, int UNR, bool MT2>
__global__ __launch_bounds__(512, 2) void chain_wide_kernel(ChainWide a)
{
    extern __shared__ double cf_lds[];
    constexpr int AP = 16 * NQF + 4 * STRQ;
    double *Wl = cf_lds;
    double *El = cf_lds + a.ebase;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int units = a.wpp * a.nac;
    int prob, unit;
    if (a.xcd_map) {
        // blocks b and b + 8 share an XCD: deal the units over the XCDs, all problems of a unit to the same one
        const int upx = units >> 3, xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        unit = xcd * upx + j % upx;
        prob = j / upx;
    } else {
        prob = blockIdx.x / units;
        unit = blockIdx.x - prob * units;
    }
    const int prob0 = prob * a.tpw;                    // first tensor of this workgroup's group
    const int g = unit / a.nac, ac = unit - g * a.nac;
    const int a0 = ac * a.ac;
    const int cnt = a.A - a0 < a.ac ? a.A - a0 : a.ac;          // columns of W / rows of E this chunk really has
    const int k_beg = (int)((int64_t)g * a.n / a.wpp), k_end = (int)((int64_t)(g + 1) * a.n / a.wpp);
    const int KB1 = ((a.K1 + 3) / 4 + UNR - 1) / UNR * UNR;
    const int A2P = a.A2P;

    // ---- stage the chunk of W (of every tensor of the group): Wl[(c >> 1) * 2 AP + 2 col + (c & 1)] = W[c][a0 + col], zero beyond (K1, cnt)
    for (int sl = 0; sl < a.tpw; ++sl) {
        if (prob0 + sl >= a.nb) break;
        const double *Wp = uniform_ptr(a.W[prob0 + sl]);
        double *Wd = Wl + (size_t)sl * a.wimg;
        const __amdgpu_buffer_rsrc_t rw = make_rsrc(Wp, ((int64_t)(a.K1 - 1) * a.w_c + a.A) * 8);
        const int total = 4 * KB1 * AP;
        constexpr int BATCH = 8;
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
                if (e < total) Wd[(c >> 1) * 2 * AP + 2 * col + (c & 1)] = v[u];
            }
        }
    }

    if (w == a.loader) {
        // ---- loader: rows a0 .. of E_k -> El in 16-byte units (layout: chain_fused.h).  Rows beyond the chunk
        // (and beyond A: clamped) meet exact zeros of T.
        const int NI = a.eunits >> 6;
        const uint32_t inv = (uint32_t)(((1ull << 32) + (uint32_t)A2P - 1) / (uint32_t)A2P);   // U / A2P for U < 2^16
        const int64_t rowstride = (int64_t)a.n * a.A2;
        const int ebuf = a.eunits * 2;                 // doubles per E image
        const bool odd = a.A2 & 1;
        auto fill = [&](int k) {
            const double *Ek = a.E + (int64_t)k * a.A2;
            double *dst = El + ((a.ebuf2 && (k & 1)) ? ebuf : 0);
            const bool tail = odd && k == a.n - 1;     // the unit behind the last column would reach past the core
#pragma unroll 2
            for (int m = 0; m < NI; ++m) {
                const uint32_t U = 64u * (uint32_t)m + (uint32_t)lane;
                const uint32_t sec = (uint32_t)(((uint64_t)U * inv) >> 32), u = U - sec * (uint32_t)A2P;
                int row = a0 + (int)(2 * sec + (u & 1));
                row = row < a.A ? row : a.A - 1;
                int col = (int)(2 * (u >> 1));
                col = (col + 1 < a.A2 || (odd && !tail && col < a.A2)) ? col : 0;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(Ek + row * rowstride + col),
                                                 (__attribute__((address_space(3))) void *)(dst + m * 128), 16, 0, 0);
            }
            if (tail) {
                // last slice, odd A2: column A2 - 1 of every row by an 8-byte load (after the units have landed)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane < AP) {
                    int row = a0 + lane;
                    row = row < a.A ? row : a.A - 1;
                    const double v = Ek[row * rowstride + a.A2 - 1];
                    dst[(lane >> 1) * 2 * A2P + 4 * ((a.A2 - 1) >> 1) + 2 * (lane & 1)] = v;
                }
            }
        };
        __syncthreads();                               // W staged (all waves)
        if (a.ebuf2) {
            if (k_beg < k_end) fill(k_beg);
            for (int k = k_beg; k < k_end; ++k) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                cf_barrier();                          // B1: E_k is in LDS
                if (k + 1 < k_end) fill(k + 1);        // image (k + 1) & 1 was last read in phase B of slice k - 1
                cf_barrier();                          // B2
            }
        } else {
            for (int k = k_beg; k < k_end; ++k) {
                fill(k);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                cf_barrier();                          // B1
                cf_barrier();                          // B2: phase B of slice k is done, El may be overwritten
            }
        }
        return;
    }
    __syncthreads();                                   // W staged
    const int t0 = a.tile0[w], t1 = a.tile1[w];
    const int myprob = prob0 + a.slot[w];
    if (t0 < 0 || myprob >= a.nb) return;              // no rows for this wave (a finished wave leaves the barrier count)
    const double *Wmine = Wl + (size_t)a.slot[w] * a.wimg;
    if constexpr (MT2) {
        if (t1 >= 0) {
            cw_compute<NQF, STRQ, NNF, STRN, WT, UNR, 2>(a, Wmine, El, myprob, unit, a0, cnt, k_beg, k_end, t0, t1);
            return;
        }
    }
    cw_compute<NQF, STRQ, NNF, STRN, WT, UNR, 1>(a, Wmine, El, myprob, unit, a0, cnt, k_beg, k_end, t0, -1);
}

int chain_wide_try(const ChainStepArgs &c, int stream, hipStream_t st, bool force = false);

}  // namespace ttsk

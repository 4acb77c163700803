// One step of a TensorTrainDRM chain with the intermediate kept on chip.
//
//   reference: tensor_train_drm.py:81-87   L_mu[l,m] = sum_{i,j,k} L_{mu-1}[i,j] X_mu[i,k,l] D_mu[j,k,m]
//   (right sketches run the same line on the transposed tensor, drm_base.py:122-145)
//
// In the names used here, for one tensor of the batch and one mode:
//   W [c][a]      the carried matrix L_{mu-1} / R  (K1 x A), c = TT bond contracted with X, a = DRM rank in
//   X             the TT core, addressed by strides as G_k[j][c] (k = mode index, j = the other TT bond)
//   E [a][k][a']  the DRM core (A x n x A2), shared by all tensors of the batch
//   Out[j][a']    = sum_k ( G_k W ) E_k          (J x A2)
// The two-launch form (skinny.h) writes T[a][k][j] = sum_c W[c][a] G_k[j][c] to HBM and reads it back
// for the second product: ~55 % of the bytes a chain step moves.  Here a workgroup owns a range of
// slices k of one tensor; per slice
//   phase A   T_k^T[a][j] = sum_c W[c][a] G_k[j][c]      W from LDS, G_k fragments straight from memory
//   phase B   Out[j][a'] += sum_a T_k[j][a] E_k[a][a']   E_k from LDS
// and T_k never leaves the registers: with mfma(A = W^T fragment, B = G fragment) register t of the
// 16x16 accumulator tile Q holds T^T[16 Q + 4 t + (lane >> 4)][lane & 15], which IS the A-operand
// fragment of k-block 4 Q + t for phase B (lane (x, kq) holds T[j0 + x][4 (4 Q + t) + kq]).
// Wave w owns the 16 rows j0 = 16 w of the output; a further wave does nothing but bring E_k into LDS
// (global_load_lds, 1 KB per instruction) while the others are in phase A.  The left chain also needs
// T in memory for Psi_mu = T_mu R_mu (tensor_train_sketch.py:28-34): it is stored from the phase-A
// registers on the way (WT).
//
// Partial tiles (rank 100 = 6 x 16 + 4, rank 50 = 3 x 16 + 2) are 4-wide strips computed with
// v_mfma_f64_4x4x4 (4 blocks): 16 instead of 64 cycles of the matrix pipe.
#pragma once
#include "skinny.h"

namespace ttsk {

struct ChainStep {
    const double *W[SK_MAXB];
    const double *X[SK_MAXB];
    double *T[SK_MAXB];          // WT: T[a][k][j] (A x n x J contiguous)
    const double *E;
    double *slab;                // [problem][workgroup of the problem][J][A2]
    int nb, wpp, n;              // problems, workgroups per problem, slices (mode size)
    int K1, A, A2, J;
    int64_t w_c;                 // row stride of W (elements); columns contiguous
    int64_t x_j, x_k, x_c;       // element strides of X
    int64_t x_extent, t_extent;  // elements addressable from the bases
    int AP, A2P;                 // padded extents of the two LDS images
    int ebase;                   // offset (doubles) of the E image in LDS
    int eunits;                  // 16-byte units of the E image the loader fills (multiple of 64)
    int xcd_map;                 // 1: workgroups of the same slice range share an XCD (E_k from one L2)
    int diag;                    // timing experiments (TTSK_CF_DIAG): 1 = no X loads, 2 = no E loads, 4 = no barriers; results are then wrong
    long long *stamps;           // diagnostics (TTSK_CF_STAMPS): s_memtime of workgroup 0, [slice][wave][8]
};

constexpr int CF_MAX_DMA = 160;  // loader instructions per slice (1 KB each)

__device__ __forceinline__ void cf_barrier()
{
    // LDS traffic of this wave is complete, nothing moves across; vector-memory loads stay in flight
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// Timing experiments (TTSK_CF_DIAG / TTSK_CF_STAMPS) exist in a lab build only (-DTTSK_LAB): the shipped code object carries
// neither the run-time switches nor the stamp stores in its hot loop.
#ifdef TTSK_LAB
#define CF_DIAG(bit) (a.diag & (bit))
#define CF_STAMP(i) do { if (a.stamps && blockIdx.x == 0 && lane == 0 && k - k_beg < 8) a.stamps[((k - k_beg) * 8 + w) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define CF_DIAG(bit) 0
#define CF_STAMP(i) do { } while (0)
#endif
#define CF_BARRIER() do { if (!CF_DIAG(4)) cf_barrier(); } while (0)

// NQF / NNF full 16-wide tiles of a / a', STRQ / STRN 4-wide strips behind them; D ring depth of the
// G fragments; WT: T is also written to memory; OCC workgroups per CU the register budget allows.
template <int NQF, int STRQ, int NNF, int STRN, int D, bool WT, int OCC, int EBUF, int UNR = 5 * D>
__global__ __launch_bounds__(512, 2 * OCC) void chain_step_kernel(ChainStep a)
{
    extern __shared__ double cf_lds[];
    double *Wl = cf_lds;
    double *El = cf_lds + a.ebase;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int x16 = lane & 15, kq = lane >> 4;
    int prob, g;
    if (a.xcd_map) {
        // blocks b and b + 8 share an XCD: deal the slice ranges g over the XCDs, all problems of a
        // range to the same one (they read the same E_k)
        const int gpx = a.wpp >> 3, xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        g = xcd * gpx + j % gpx;
        prob = j / gpx;
    } else {
        prob = blockIdx.x / a.wpp;
        g = blockIdx.x - prob * a.wpp;
    }
    const int k_beg = (int)((int64_t)g * a.n / a.wpp), k_end = (int)((int64_t)(g + 1) * a.n / a.wpp);
    const int NW = (a.J + 15) >> 4;
    // k-blocks of phase A, padded to whole runs of UNR (the padded ones meet zero rows of the W image)
    const int KB1 = ((a.K1 + 3) / 4 + UNR - 1) / UNR * UNR;
    const int AP = a.AP, A2P = a.A2P;

    // ---- stage W: Wl[(c >> 1) * 2 AP + 2 a + (c & 1)], zero beyond (K1, A).  A pair of k rows interleaved:
    // the 32 lanes (kq in {0, 1}, x) of a fragment read 32 consecutive doubles = all 64 banks once.
    {
        const double *Wp = uniform_ptr(a.W[prob]);
        const __amdgpu_buffer_rsrc_t rw = make_rsrc(Wp, ((int64_t)(a.K1 - 1) * a.w_c + a.A) * 8);
        const int total = 4 * KB1 * AP;
        constexpr int BATCH = 10;
        for (int e0 = tid; e0 < total; e0 += 512 * BATCH) {
            double v[BATCH];
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int e = e0 + 512 * u;
                const int c = e / AP, col = e - c * AP;
                v[u] = ld8(rw, (e < total && c < a.K1 && col < a.A) ? (uint32_t)(((int64_t)c * a.w_c + col) * 8) : OOB_OFF, 0);
            }
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int e = e0 + 512 * u;
                const int c = e / AP, col = e - c * AP;
                if (e < total) Wl[(c >> 1) * 2 * AP + 2 * col + (c & 1)] = v[u];
            }
        }
    }

    if (w == 7) {
        // ---- loader: E_k -> El in 16-byte units.  Unit u of section sec (rows 2 sec, 2 sec + 1) is
        // columns (2 (u >> 1), + 1) of row 2 sec + (u & 1); lanes (kq in {0, 1}, x) of a fragment read 16
        // consecutive units.  Rows beyond A repeat row A - 1: they meet exact zeros of T.
        // (A rolled loop: unrolled, the 80 LDS destinations are hoisted into 80 scalar registers and the
        // whole kernel starts spilling them.)
        const int NI = CF_DIAG(2) ? 0 : a.eunits >> 6;
        const uint32_t inv = (uint32_t)(((1ull << 32) + (uint32_t)A2P - 1) / (uint32_t)A2P);   // U / A2P for U < 2^16
        const int64_t rowstride = (int64_t)a.n * a.A2;
        const int ebuf = a.eunits * 2;                 // doubles per E image
        auto fill = [&](int k) {
            const double *Ek = a.E + (int64_t)k * a.A2;
            double *dst = El + (EBUF == 2 ? (k & 1) * ebuf : 0);
#pragma unroll 2
            for (int m = 0; m < NI; ++m) {
                const uint32_t U = 64u * (uint32_t)m + (uint32_t)lane;
                const uint32_t sec = (uint32_t)(((uint64_t)U * inv) >> 32), u = U - sec * (uint32_t)A2P;
                int row = (int)(2 * sec + (u & 1));
                row = row < a.A ? row : a.A - 1;
                int col = (int)(2 * (u >> 1));
                col = col + 1 < a.A2 ? col : 0;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(Ek + row * rowstride + col),
                                                 (__attribute__((address_space(3))) void *)(dst + m * 128), 16, 0, 0);
            }
        };
        __syncthreads();                               // W staged (all waves)
        if constexpr (EBUF == 2) {
            // two images: E_{k+1} travels while the others are in phase B of slice k and phase A of k + 1
            // (short phases -- small ranks -- would otherwise wait for the load at every B1)
            if (k_beg < k_end) fill(k_beg);
            for (int k = k_beg; k < k_end; ++k) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                CF_STAMP(1);
                CF_BARRIER();                          // B1: E_k is in LDS
                CF_STAMP(2);
                if (k + 1 < k_end) fill(k + 1);        // image (k + 1) & 1 was last read in phase B of slice k - 1
                CF_STAMP(3);
                CF_BARRIER();                          // B2
                CF_STAMP(4);
            }
        } else {
            for (int k = k_beg; k < k_end; ++k) {
                CF_STAMP(0);
                fill(k);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                CF_STAMP(1);
                CF_BARRIER();                          // B1: E_k is in LDS, phase A of slice k is done
                CF_STAMP(2);
                CF_STAMP(3);
                CF_BARRIER();                          // B2: phase B of slice k is done, El may be overwritten
                CF_STAMP(4);
            }
        }
        return;
    }
    __syncthreads();                                   // W staged
    if (w >= NW) return;                               // no rows for this wave (a finished wave leaves the barrier count)

    // ---- compute wave: rows j0 .. j0 + 15 of the output
    const int j0 = 16 * w;
    const bool jok = j0 + x16 < a.J;
    const uint32_t xlane = !CF_DIAG(1) ? (uint32_t)(((int64_t)(j0 + x16) * a.x_j + (int64_t)kq * a.x_c) * 8) : OOB_OFF;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(uniform_ptr(a.X[prob]), a.x_extent * 8);
    // wave-uniform by construction; said so explicitly, or a value the register allocator parks in a
    // vector register turns every load below into a waterfall loop
    const uint32_t xstep = __builtin_amdgcn_readfirstlane((uint32_t)(4 * a.x_c * 8));
    const uint32_t kstep = __builtin_amdgcn_readfirstlane((uint32_t)(a.x_k * 8));
    static_assert(UNR % D == 0, "the unrolled body must keep the ring slots static");
    // k-blocks are issued in straight-line runs of UNR (the compiler's wait counts are exact inside a run;
    // at a loop edge it waits for every load in flight); a slice is padded to whole runs
    const int ITER = KB1 / UNR;
    // fragment kb of slice k: per-lane (row, k) offset fixed, the (slice, k-block) part a scalar offset.  No
    // masks: k beyond K1 meets the zero rows of the W image, rows beyond J give output rows that are never
    // stored, a prefetch past the workgroup's last slice is unused, and whatever lies beyond the core reads
    // as 0 through the descriptor's range check.
    auto xload = [&](uint32_t so) -> double { return ld8(rx, xlane, __builtin_amdgcn_readfirstlane(so)); };
    double ring[D];
#pragma unroll
    for (int d = 0; d < D; ++d) ring[d] = xload((uint32_t)k_beg * kstep + (uint32_t)d * xstep);

    v4d acc2[NNF ? NNF : 1];
    double acc2s[STRN ? STRN : 1];
#pragma unroll
    for (int p = 0; p < NNF; ++p)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc2[p][t] = 0.0;
#pragma unroll
    for (int q = 0; q < STRN; ++q) acc2s[q] = 0.0;

    // LDS addresses of this lane's fragment elements inside a k-block's two sections
    const int wl_lane = (kq >> 1) * 2 * AP + 2 * x16 + (kq & 1);
    const int ws_lane = (kq >> 1) * 2 * AP + 2 * (16 * NQF + (x16 & 3)) + (kq & 1);
    const int el_lane = (kq >> 1) * 2 * A2P + 4 * (x16 >> 1) + 2 * (kq & 1) + (x16 & 1);
    const int es_col = 16 * NNF + (x16 & 3);
    const int es_lane = (kq >> 1) * 2 * A2P + 2 * (kq & 1);

    __amdgpu_buffer_rsrc_t rt;
    if constexpr (WT) rt = make_rsrc(uniform_ptr(a.T[prob]), a.t_extent * 8);

    for (int k = k_beg; k < k_end; ++k) {
        v4d acc1[NQF ? NQF : 1];
        double acc1s[STRQ ? STRQ : 1];
#pragma unroll
        for (int p = 0; p < NQF; ++p)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc1[p][t] = 0.0;
#pragma unroll
        for (int q = 0; q < STRQ; ++q) acc1s[q] = 0.0;

        CF_STAMP(0);
        // ---- phase A.  Each k-block: the W fragments of the NEXT k-block are requested from LDS before this
        // one's matrix instructions are issued (a wave then covers its own LDS latency -- the matrix pipe
        // serves the older wave of a SIMD first, so the younger one cannot be counted on to fill the gaps),
        // and the ring slot is reloaded AFTER its fragment has been consumed, so that the load can land in the
        // same register (reloading first forces a copy at the loop edge behind an s_waitcnt vmcnt(0)).
        double af[NQF ? NQF : 1], sf[STRQ ? STRQ : 1];
        // fragments of k-block (run base) + u: compile-time offsets from the run's base address (u = UNR, the
        // look-ahead of a run's last k-block, reads the rows behind the W image: finite, unused)
        auto wfetch = [&](const double *wrun, int u, double (&f)[NQF ? NQF : 1], double (&g)[STRQ ? STRQ : 1]) {
#pragma unroll
            for (int p = 0; p < NQF; ++p) f[p] = LDS_UNPAIRED(wrun[wl_lane + u * 4 * AP + 32 * p]);
#pragma unroll
            for (int q = 0; q < STRQ; ++q) g[q] = wrun[ws_lane + u * 4 * AP + 8 * q];
        };
        wfetch(Wl, 0, af, sf);
        uint32_t so = (uint32_t)k * kstep + (uint32_t)D * xstep;      // scalar offset of the next fragment to request
        // k-block (run base) + u with u < UNR static; the reloads of the slice's last D k-blocks are the first
        // fragments of the next slice
        auto kblock = [&](const double *wrun, int u, bool wrap) {
            const int d = u % D;
            double afn[NQF ? NQF : 1], sfn[STRQ ? STRQ : 1];
            wfetch(wrun, u + 1, afn, sfn);
            const double bf = ring[d];
#pragma unroll
            for (int p = 0; p < NQF; ++p) acc1[p] = mfma16(af[p], bf, acc1[p]);
#pragma unroll
            for (int q = 0; q < STRQ; ++q) acc1s[q] = mfma4(sf[q], bf, acc1s[q]);
            if (wrap) so = (uint32_t)(k + 1) * kstep;
            ring[d] = xload(so);
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
            // T[a][k][j]: register t of tile p is row a = 16 p + 4 t + kq, 16 consecutive j per row
            const int64_t nJ = (int64_t)a.n * a.J;
            const uint32_t tl = jok ? (uint32_t)(((int64_t)kq * nJ + (int64_t)k * a.J + j0 + x16) * 8) : OOB_OFF;
#pragma unroll
            for (int p = 0; p < NQF; ++p)
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    st8(rt, (jok && 16 * p + 4 * t + kq < a.A) ? tl + (uint32_t)((16 * p + 4 * t) * nJ * 8) : OOB_OFF, acc1[p][t]);
#pragma unroll
            for (int q = 0; q < STRQ; ++q)
                st8(rt, (jok && 16 * NQF + 4 * q + kq < a.A) ? tl + (uint32_t)((16 * NQF + 4 * q) * nJ * 8) : OOB_OFF, acc1s[q]);
        }
        CF_STAMP(1);
        CF_BARRIER();                                  // B1
        CF_STAMP(2);

        // ---- phase B: k-block kap = 4 p + t of T is register t of tile p.  Same look-ahead: the E fragments of
        // k-block kap + 1 are requested before the matrix instructions of kap (left to itself the compiler
        // requests them right in front of their first use and every k-block waits out the LDS latency).
        {
            const double *eb = El + (EBUF == 2 ? (k & 1) * a.eunits * 2 : 0);
            double bf[NNF ? NNF : 1], bs[STRN ? STRN : 1];
            // per-slice copies of the lane offsets the compiler cannot see through: it then forms each k-block's
            // address with one add where it is used, instead of keeping 2 x 25 precomputed addresses alive across
            // the slice loop (and spilling them into the phase)
            int el0 = el_lane, es0 = es_lane + 4 * (es_col >> 1) + (es_col & 1);
            asm volatile("" : "+v"(el0), "+v"(es0));
            auto efetch = [&](int kap, double (&f)[NNF ? NNF : 1], double (&g)[STRN ? STRN : 1]) {
#pragma unroll
                for (int nn = 0; nn < NNF; ++nn) f[nn] = LDS_UNPAIRED(eb[el0 + kap * 4 * A2P + 32 * nn]);
#pragma unroll
                for (int q = 0; q < STRN; ++q) g[q] = eb[es0 + kap * 4 * A2P + 8 * q];
            };
            // two fragment sets used in turn (even / odd k-blocks): with one set and a copy the compiler merges the
            // look-ahead registers with the current ones and the requests slide back behind the matrix instructions
            double bf2[NNF ? NNF : 1], bs2[STRN ? STRN : 1];
            efetch(0, bf, bs);
            constexpr int KB2C = 4 * NQF + STRQ;
#pragma unroll
            for (int kap = 0; kap < KB2C; ++kap) {
                const int p = kap >> 2, t = kap & 3;
                const double af2 = p < NQF ? acc1[p < NQF ? p : 0][t] : acc1s[t < STRQ ? t : 0];
                if (kap & 1) {
                    if (kap + 1 < KB2C) efetch(kap + 1, bf, bs);
                    __builtin_amdgcn_sched_barrier(0);      // the requests stay in front of this k-block's matrix instructions
#pragma unroll
                    for (int nn = 0; nn < NNF; ++nn) acc2[nn] = mfma16(af2, bf2[nn], acc2[nn]);
#pragma unroll
                    for (int q = 0; q < STRN; ++q) acc2s[q] = mfma4(af2, bs2[q], acc2s[q]);
                } else {
                    if (kap + 1 < KB2C) efetch(kap + 1, bf2, bs2);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int nn = 0; nn < NNF; ++nn) acc2[nn] = mfma16(af2, bf[nn], acc2[nn]);
#pragma unroll
                    for (int q = 0; q < STRN; ++q) acc2s[q] = mfma4(af2, bs[q], acc2s[q]);
                }
            }
        }
        CF_STAMP(3);
        CF_BARRIER();                                  // B2
        CF_STAMP(4);
    }

    // ---- partial result of this workgroup: slab[problem][g][j][a']
    double *slab = a.slab + ((int64_t)prob * a.wpp + g) * a.J * a.A2;
#pragma unroll
    for (int nn = 0; nn < NNF; ++nn)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int j = j0 + 4 * t + kq, col = 16 * nn + x16;
            if (j < a.J && col < a.A2) slab[(int64_t)j * a.A2 + col] = acc2[nn][t];
        }
#pragma unroll
    for (int q = 0; q < STRN; ++q) {
        // 4x4x4 result: lane (i = l >> 4, beta = (l >> 2) & 3, c = l & 3) holds row 4 beta + i, column c of the strip
        const int j = j0 + 4 * ((lane >> 2) & 3) + kq, col = 16 * NNF + 4 * q + (lane & 3);
        if (j < a.J && col < a.A2) slab[(int64_t)j * a.A2 + col] = acc2s[q];
    }
}

// 1 = launched (the slab reduce included), 0 = shape not covered, < 0 = error
struct ChainStepArgs {
    int nb, n, K1, A, A2, J;
    const double *const *W;      // nb carried matrices (K1 x A), row stride w_c
    int64_t w_c;
    const double *const *X;      // nb cores
    int64_t x_j, x_k, x_c, x_extent;
    const double *E;             // (A, n, A2) contiguous
    double *const *T;            // nullptr, or nb buffers (A, n, J) contiguous
    double *const *Out;          // nb results (J x A2) contiguous
};
int chain_fused_try(const ChainStepArgs &c, int stream, hipStream_t st, bool force = false);

}  // namespace ttsk

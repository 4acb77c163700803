// One step of a TensorTrainDRM chain for MANY low-rank tensor trains at once (the terms of a TensorSum, a batch).
//
//   reference: tensor_train_drm.py:81-87   L_mu[l,m] = sum_{i,j,k} L_{mu-1}[i,j] X_mu[i,k,l] D_mu[j,k,m]
//   per summand of sketch_dispatch.py:85-147 (the DRM is shared by the summands), right sketches through
//   drm_base.py:122-145 by strides.
//
// chain_fused.h gives a wave 16 output rows of ONE tensor: with TT rank 20 two of eight waves have rows, and
// several tensors per workgroup pad 20 rows to 32.  Here the rows of TPW consecutive terms are STACKED (20 rows
// each: 4 terms = 80 rows = five full 16-row tiles), the DRM slice E_k is shared by all of them, and the work of
// the second product is cut by output COLUMNS as well as rows, so that the four SIMDs carry equal shares:
//
//   phase A   T_k[(b, j)][a] = sum_c G_{b,k}[j][c] W_b[c][a]       per term, as 4-row strips (v_mfma_f64_4x4x4):
//             a strip never straddles two terms.  W_b lives in REGISTERS (a wave computes its own columns a of one
//             term, every slice again), G_k comes through a small LDS image, T_k goes to LDS.
//   phase B   Out[(b, j)][a'] += sum_a T_k[(b, j)][a] E_k[a][a']   16x16x4 tiles that may straddle terms (E_k is
//             the same for all); every wave owns a rectangle of (row tile, column tile) pairs -- the table comes
//             from the host, which deals the rectangles so that the SIMDs are level (chain_sum.hip).
//
// LDS: T image (pair-interleaved, conflict-free fragment reads) + E image (16-byte units, filled by
// global_load_lds from all eight waves while phase A runs) + G image.  Two barriers per slice.
// Partial results per (term, slice range) go to slabs; skinny_r_reduce sums them (fixed order, no atomics).
#pragma once
#include <cstddef>
#include "chain_fused.h"

namespace ttsk {

__device__ __forceinline__ void st8s(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, double v)
{
    __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<v2i_t *>(&v), r, (int)voff, (int)soff, 0);
}

constexpr int CS_NAMAX = 4;       // a-tiles of T a wave computes in phase A (W fragments in registers)
constexpr int CS_SRMAX = 6;       // row tiles of the strip column (one wave owns all of them)
constexpr int CS_NSMAX = 2;       // 4-wide strips behind the full column tiles
constexpr int CS_DMAMAX = 13;     // E loader instructions per wave and slice (1 KB each)

struct ChainSumRole {
    unsigned char term, at0, na;  // phase A: local term, first a-tile, a-tiles (0 = none)
    unsigned char body;           // phase-B body: 16 * RT + CT for a rectangle of full tiles, 128 + 16 * SR + NS for the strip column
    unsigned char rt0, ct0;       // origin of the rectangle of full tiles
    unsigned char pad0, pad1;
};

// everything but the pointer tables: what a wave's body reads (no dynamically indexed member -- the kernel picks the
// wave's pointers and role from the tables itself, straight from the kernel-argument segment; handed to the body as one
// struct WITH the tables the compiler copies all of it to scratch)
struct ChainSumS {
    const double *E;
    double *T;                    // nullptr, or T[b * t_b + (a * n + k) * t_ld + j]
    double *slab;                 // [term][slice range][J][A2]
    int nb, n, K1, A, A2, J;
    int tpw, ngroups, nranges;
    int64_t w_c, x_j, x_k, x_c, x_extent;
    int64_t t_b, t_ld, t_extent;
    int RP;                       // row pitch of the T image (>= 16 row tiles; 2 mod 4: the stores of phase A then meet 2-way instead of 4-way bank conflicts)
    int KB2;                      // k-blocks of phase B = ceil(A / 4)
    int A2P;                      // row length of the E image (even, >= 16 NNF + 4 NS)
    int NNF, NS;                  // full column tiles of Out, 4-wide strips behind them
    int ebase, gbase;             // LDS offsets (doubles) of the E and G images; the T image sits at 0
    int eunits;                   // 16-byte units of the E image (multiple of 64)
    int xcd_map;                  // 1: the term groups of a slice range share an XCD (E_k from one L2)
    int c_fast;                   // G loader: 1 = c is the contiguous index of X (right chain), 0 = j
    // What the set-up would otherwise derive with divisions and 64-bit products: it runs once per workgroup, on a cold
    // instruction cache, with nothing else resident on the CU -- its length in INSTRUCTIONS is what it costs (1640 of
    // them took 13 k cycles of a 110 k-cycle launch).
    int kbase, krem;              // slice range rr = [rr kbase + min(rr, krem), + kbase + (rr < krem))
    int inv_ng;                   // (1 << 20) / ngroups + 1: i / ngroups == (i * inv_ng) >> 20 for i * ngroups < 2^20
    int wpt, per, gu;             // G loader: waves per term (8 / tpw), elements per wave, loads per lane and slice
    uint32_t w_c8, x_j8, x_c8;    // byte strides as 32-bit numbers (the host checks that every offset fits)
    uint32_t t_b8, t_a8;          // T: bytes per term, bytes per row a (n t_ld 8)
    uint32_t slab_t8, slab_r8;    // slab: bytes per term (nranges J A2 8) and per slice range (J A2 8)
    uint32_t e_inv;               // ceil(2^32 / A2P): unit U of the E image is in section (U e_inv) >> 32
#ifdef TTSK_LAB                    // timing experiments: only in a lab build (-DTTSK_LAB), never in the shipped code object
    long long *stamps;            // s_memtime of workgroup 0, [slice][wave][8]
    int diag;                     // (results wrong) 1 no E DMA, 2 no phase A, 4 no phase B, 8 no G loads, 16 no barriers, 32 no fragment reads in
                                  // phase B, 64 no priorities, 128 no small rectangles
#endif
};

#ifdef TTSK_LAB
#define CS_DIAG(bit) (a.diag & (bit))
#else
#define CS_DIAG(bit) 0
#endif

struct ChainSum {
    ChainSumS s;
    const double *W[SK_MAXB];
    const double *X[SK_MAXB];
    ChainSumRole role[8];
};

// What a wave sets up once, before the slice loop -- the same code for every role, so it sits in the kernel in front of
// the switch over the phase-B bodies: eight waves share one pass through the instruction cache (inside the bodies the
// straight-line set-up was fetched cold once per role: 17 us of a 59 us launch).
template <int JS, int KB1, int NA>
struct CsPre {
    static constexpr int JP = 4 * JS, KP = 4 * KB1;
    static constexpr int GMAX = (KP * JP + 127) / 128;   // G elements per lane: at least two waves load a term (tpw <= 4)
    double Wf[NA][KB1];          // W fragments: lane (m = x16, k = kq) holds W[4 kb + k][16 (at0 + p) + m]
    uint32_t goff[GMAX];         // G loader: byte offset in X (without the slice term), OOB_OFF = zero
    int glds[GMAX];              //           element offset in the G image, -1 = nothing to store
    int esrc[CS_DMAMAX];         // E loader: byte offset of this lane's 16-byte unit from E_k
    int tw[NA], twt[NA];         // T image offsets of the phase-A results (strips / 16-row tiles), -1 = not stored
    uint32_t tg[NA], tgt[NA];    // the same in memory (WT), without the slice term
    __amdgpu_buffer_rsrc_t rx, rt;
    uint32_t kstep, tkstep, tastep;
    int gl_lane, gl_tile, tl_lane, el_lane, es_lane;
    int k_beg, k_end, tb;
    double greg0[GMAX];          // the first slice's G elements, on their way when the set-up ends
};

template <int JS, int KB1, int NA, bool WT>
__device__ __forceinline__ void cs_prologue(const ChainSumS &a, const ChainSumRole ro, const double *Wterm, const double *Xterm,
                                            double *lds, const int g, const int rr, const int w, const int lane, CsPre<JS, KB1, NA> &P)
{
    constexpr int JP = 4 * JS, KP = 4 * KB1, GMAX = CsPre<JS, KB1, NA>::GMAX;
    const int x16 = lane & 15, kq = lane >> 4;
    const int RP = a.RP, A2P = a.A2P, KB2 = a.KB2;
#ifdef TTSK_LAB
#define CS_PSTAMP(i) do { if (a.stamps && blockIdx.x == 0 && lane == 0) a.stamps[((4 * 8) + w) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define CS_PSTAMP(i) do { } while (0)
#endif
    P.k_beg = rr * a.kbase + (rr < a.krem ? rr : a.krem);
    P.k_end = P.k_beg + a.kbase + (rr < a.krem ? 1 : 0);
    const int tb = g * a.tpw + ro.term;             // the term whose T columns this wave computes
    const bool tv = tb < a.nb;
    P.tb = tb;
    // ---- G loader: the 8 / tpw waves of a term share its K1 x J slice (zero beyond)
    const int wpt = a.wpt;                           // waves per term: 8, 4 or 2
    const int lt = wpt == 8 ? 0 : (wpt == 4 ? w >> 2 : w >> 1), part = w - lt * wpt;
    const int gt = g * a.tpw + lt;
    const int per = a.per;                           // ceil(KP JP / wpt)
    const int gu = a.gu;                             // loads per lane and slice (<= GMAX)
#pragma unroll
    for (int u = 0; u < GMAX; ++u) {
        const uint32_t e = (uint32_t)(part * per + u * 64 + lane);
        const bool in = u < gu && u * 64 + lane < per && e < (uint32_t)(KP * JP);
        uint32_t c, j;
        if constexpr (KP == JP) {                    // one division by a constant, the roles picked afterwards
            const uint32_t q = e / (uint32_t)JP, r = e - q * (uint32_t)JP;
            c = a.c_fast ? r : q;
            j = a.c_fast ? q : r;
        } else {
            c = a.c_fast ? e % (uint32_t)KP : e / (uint32_t)JP;
            j = a.c_fast ? e / (uint32_t)KP : e % (uint32_t)JP;
        }
        uint32_t off = j * a.x_j8 + c * a.x_c8;
        asm volatile("" : "+v"(off));                // (computed in front of the select: no branch around it)
        P.goff[u] = (in && gt < a.nb && c < (uint32_t)a.K1 && j < (uint32_t)a.J) ? off : OOB_OFF;
        P.glds[u] = in ? lt * (KP * JP) + (int)(c * JP + j) : -1;
    }
    P.rx = make_rsrc(uniform_ptr(Xterm), a.x_extent * 8);
    P.kstep = (uint32_t)(a.x_k * 8);
    // first slice's G: requested in front of the W fragments (160 load instructions per workgroup stand in the queue
    // for 2-3 k cycles; the barrier below needs only G)
    double greg[GMAX];
#pragma unroll
    for (int u = 0; u < GMAX; ++u) greg[u] = ld8(P.rx, P.goff[u], __builtin_amdgcn_readfirstlane((uint32_t)P.k_beg * P.kstep));
    asm volatile("" ::: "memory");
    CS_PSTAMP(0);
    // ---- W fragments of this wave's phase-A unit; zero beyond (K1, A) and for a-tiles the wave does not own (they give
    // zero columns of T that are not stored)
    {
        const double *Wp = uniform_ptr(Wterm);
        const __amdgpu_buffer_rsrc_t rw = make_rsrc(Wp, ((int64_t)(a.K1 - 1) * a.w_c + a.A) * 8);
        // lane offset of (row kq, this lane's column) or out of range, the k-block as the scalar offset: two instructions per load
        // (with the conditions inside the offset expression the compiler built a branch around every load)
        uint32_t wrow = (uint32_t)kq * a.w_c8;
        const uint32_t wstep = 4u * a.w_c8;
        uint32_t rmask[KB1];
#pragma unroll
        for (int kb = 0; kb < KB1; ++kb) rmask[kb] = 4 * kb + kq < a.K1 ? 0u : OOB_OFF;
#pragma unroll
        for (int p = 0; p < NA; ++p) {
            const int col = 16 * (ro.at0 + p) + x16;
            uint32_t w0 = (wrow + 8u * (uint32_t)col) | ((tv && p < ro.na && col < a.A) ? 0u : OOB_OFF);
            asm volatile("" : "+v"(w0));
#pragma unroll
            for (int kb = 0; kb < KB1; ++kb) P.Wf[p][kb] = ld8(rw, w0 | rmask[kb], (uint32_t)kb * wstep);
        }
    }
    CS_PSTAMP(1);
    // ---- E loader: unit U of the image = columns (2 (u >> 1), + 1) of row 2 sec + (u & 1), sec = U / A2P, u = U % A2P;
    // wave w issues instructions w, w + 8, ... (1 KB each); the per-lane source offsets are the same for every slice.
    // Rows beyond A repeat row A - 1 (they meet exact zeros of T).
    {
        const uint32_t inv = a.e_inv;                // ceil(2^32 / A2P)
        const int rowstride = a.n * a.A2;
#pragma unroll
        for (int i = 0; i < CS_DMAMAX; ++i) {
            const int m = w + 8 * i;
            const uint32_t U = 64u * (uint32_t)m + (uint32_t)lane;
            const uint32_t sec = (uint32_t)(((uint64_t)U * inv) >> 32), u = U - sec * (uint32_t)A2P;
            int row = (int)(2 * sec + (u & 1));
            row = row < a.A ? row : a.A - 1;
            int col = (int)(2 * (u >> 1));
            col = col + 1 < a.A2 ? col : 0;
            P.esrc[i] = (row * rowstride + col) * 8;   // bytes (< 2^32: checked by the host)
        }
    }
    CS_PSTAMP(2);
    // lane offsets of the fragment reads
    P.gl_lane = ro.term * (KP * JP) + kq * JP + (x16 & 3);                   // + (4 kb) JP + 4 qs
    P.gl_tile = ro.term * (KP * JP) + kq * JP + x16;                         // G fragment of a 16-row tile: + (4 kb) JP + 16 q
    P.tl_lane = (kq >> 1) * 2 * RP + 2 * x16 + (kq & 1);                      // + (2 kb) 2 RP + 32 rt
    P.el_lane = (kq >> 1) * 2 * A2P + 4 * (x16 >> 1) + 2 * (kq & 1) + (x16 & 1);   // + (2 kb) 2 A2P + 32 ct
    P.es_lane = (kq >> 1) * 2 * A2P + 4 * ((16 * a.NNF + (x16 & 3)) >> 1) + 2 * (kq & 1) + (x16 & 1);   // + 8 s
    // where this lane's phase-A results go.  Strips: lane (i = kq, beta, j4) holds a = 16 (at0 + p) + 4 beta + kq, row = term JP
    // + j4 (+ the strip's rows); tiles: register t of lane (x16, kq) is a = 16 (at0 + p) + 4 t + kq, row = term JP + x16
    // (four rows a further = t * 4 RP elements of the image).  Memory (WT): byte offsets without the slice term.
    const int beta = (lane >> 2) & 3, j4 = lane & 3;
    if constexpr (WT) P.rt = make_rsrc(a.T, a.t_extent * 8);
#pragma unroll
    for (int p = 0; p < NA; ++p) {
        const int ai = 16 * (ro.at0 + p) + 4 * beta + kq;
        P.tw[p] = (p < ro.na && ai < 4 * KB2) ? (ai >> 1) * 2 * RP + 2 * (ro.term * JP + j4) + (ai & 1) : -1;
        if constexpr (WT) {
            uint32_t o = (uint32_t)tb * a.t_b8 + (uint32_t)ai * a.t_a8 + 8u * (uint32_t)j4;
            asm volatile("" : "+v"(o));
            P.tg[p] = (tv && p < ro.na && ai < a.A) ? o : OOB_OFF;
        } else P.tg[p] = OOB_OFF;
        const int at = 16 * (ro.at0 + p) + kq;
        P.twt[p] = p < ro.na ? (at >> 1) * 2 * RP + 2 * (ro.term * JP + x16) + (at & 1) : -1;
        if constexpr (WT) {
            uint32_t o = (uint32_t)tb * a.t_b8 + (uint32_t)at * a.t_a8 + 8u * (uint32_t)x16;
            asm volatile("" : "+v"(o));
            P.tgt[p] = (tv && p < ro.na) ? o : OOB_OFF;
        } else P.tgt[p] = OOB_OFF;
    }
    P.tkstep = (uint32_t)(a.t_ld * 8);
    P.tastep = 4u * a.t_a8;                                    // four rows a further in T
    CS_PSTAMP(3);
#pragma unroll
    for (int u = 0; u < GMAX; ++u) P.greg0[u] = greg[u];
#undef CS_PSTAMP
}

// the wave's share of the workgroup: the slice loop (accumulators live across slices) and the partial results.
// NA: a-tiles per wave in phase A; (RT, CT): the wave's rectangle of full tiles, or (SR, NS): the strip column
// (SR row tiles x NS 4-wide strips; RT = CT = 0 then)
template <int JS, int KB1, int NA, bool WT, int RT, int CT, int SR, int NS>
__device__ __forceinline__ void cs_wave(const ChainSumS &a, const ChainSumRole ro, const CsPre<JS, KB1, NA> &P, double *lds,
                                        const int g, const int rr, const int w, const int lane)
{
    constexpr int JP = 4 * JS, KP = 4 * KB1;
    constexpr int GMAX = CsPre<JS, KB1, NA>::GMAX;
    constexpr int NST = WT ? NA * (4 * (JS / 4) + JS % 4) : 0;   // T stores per slice (masked ones are issued out of range: static count)
    constexpr bool HS = SR > 0;
#ifdef TTSK_LAB
    if (a.stamps && blockIdx.x == 0 && lane == 0) a.stamps[((5 * 8) + w) * 8 + 0] = __builtin_amdgcn_s_memtime();      // body entered
#endif
    double *TL = lds, *EL = lds + a.ebase, *GL = lds + a.gbase;
    const int x16 = lane & 15, kq = lane >> 4;
    const int k_beg = P.k_beg, k_end = P.k_end;
    const int RP = a.RP, A2P = a.A2P, KB2 = a.KB2;
    const int NI = a.eunits >> 6;
    const int beta = (lane >> 2) & 3, j4 = lane & 3;
    const int tb = P.tb;
    const auto &Wf = P.Wf;
    const auto &tw = P.tw; const auto &twt = P.twt; const auto &tg = P.tg; const auto &tgt = P.tgt;
    const __amdgpu_buffer_rsrc_t rt = P.rt;
    const uint32_t tkstep = P.tkstep, tastep = P.tastep;
    const int gl_lane = P.gl_lane, gl_tile = P.gl_tile, tl_lane = P.tl_lane, el_lane = P.el_lane, es_lane = P.es_lane;
    double greg[GMAX];
    auto g_load = [&](int k) {
#pragma unroll
        for (int u = 0; u < GMAX; ++u)            // always GMAX loads (masked ones out of range): the wait counts below are static
            greg[u] = ld8(P.rx, P.goff[u], __builtin_amdgcn_readfirstlane((uint32_t)k * P.kstep));
    };
    auto g_store = [&]() {
#pragma unroll
        for (int u = 0; u < GMAX; ++u)
            if (P.glds[u] >= 0) GL[P.glds[u]] = greg[u];
    };
    auto e_fill = [&](int k) {
        // uniform base + 32-bit lane offset: one address instruction per load
        const char *Ek = (const char *)uniform_ptr(a.E + (int64_t)k * a.A2);
#pragma unroll
        for (int i = 0; i < CS_DMAMAX; ++i)
            if (w + 8 * i < NI)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(Ek + (uint32_t)P.esrc[i]),
                                                 (__attribute__((address_space(3))) void *)(EL + (w + 8 * i) * 128), 16, 0, 0);
    };

    // ---- accumulators of phase B
    v4d acc2[RT ? RT : 1][CT ? CT : 1];
    double acc2s[HS ? SR : 1][HS ? NS : 1];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc2[r][c][t] = 0.0;
    if constexpr (HS) {
#pragma unroll
        for (int r = 0; r < SR; ++r)
#pragma unroll
            for (int s = 0; s < NS; ++s) acc2s[r][s] = 0.0;
    }

    // The first slice's G into its image -- HERE, behind the switch over the bodies and this body's own set-up: the cold
    // instruction fetches of both (2-3 k cycles) run while the loads are on their way instead of behind the barrier.
    {
#ifdef TTSK_LAB
        if (a.stamps && blockIdx.x == 0 && lane == 0) a.stamps[((4 * 8) + w) * 8 + 4] = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
        for (int u = 0; u < GMAX; ++u)
            if (P.glds[u] >= 0) GL[P.glds[u]] = P.greg0[u];
#ifdef TTSK_LAB
        if (a.stamps && blockIdx.x == 0 && lane == 0) a.stamps[((4 * 8) + w) * 8 + 5] = __builtin_amdgcn_s_memtime();
#endif
        __syncthreads();
#ifdef TTSK_LAB
        if (a.stamps && blockIdx.x == 0 && lane == 0) a.stamps[((4 * 8) + w) * 8 + 7] = __builtin_amdgcn_s_memtime();
#endif
    }

#ifdef TTSK_LAB
#define CS_STAMP(i) do { if (a.stamps && blockIdx.x == 0 && lane == 0 && k - k_beg < 8) a.stamps[((k - k_beg) * 8 + w) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define CS_STAMP(i) do { } while (0)
#endif
    for (int k = k_beg; k < k_end; ++k) {
        CS_STAMP(0);
        if (!(CS_DIAG(1))) e_fill(k);                // lands while phase A runs
        if (k + 1 < k_end && !(CS_DIAG(8))) g_load(k + 1);
        // ---- phase A: all G fragments of the slice requested up front (one LDS round trip per slice)
        if (!(CS_DIAG(2))) {
            // Per term 16 rows as one 16x16x4 tile per (a-tile, k-block) and the rows beyond as 4-row strips (4x4x4): the same
            // matrix-pipe time as strips throughout, two instructions and two fragment reads per k-block instead of five.
            constexpr int NT = JS / 4, NR = JS % 4;      // 16-row tiles and 4-row strips of a term (J <= 4 JS)
            v4d acc1[NA][NT ? NT : 1];
            double acc1s[NA][NR ? NR : 1];
#pragma unroll
            for (int p = 0; p < NA; ++p) {
#pragma unroll
                for (int q = 0; q < NT; ++q)
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc1[p][q][t] = 0.0;
#pragma unroll
                for (int q = 0; q < NR; ++q) acc1s[p][q] = 0.0;
            }
            // G fragments: tile q: lane (n = x16, k = kq) holds G[16 q + n][4 kb + k]; strip q: lane (j4, k) holds G[16 NT + 4 q + j4][4 kb + k]
            // (replicated over the lane blocks); those of k-block kb + 1 are requested behind the first matrix instructions of kb
            double gt[2][NT ? NT : 1], gs[2][NR ? NR : 1];
            auto gfetch = [&](int kb, int set) {
#pragma unroll
                for (int q = 0; q < NT; ++q) gt[set][q] = GL[gl_tile + 4 * kb * JP + 16 * q];
#pragma unroll
                for (int q = 0; q < NR; ++q) gs[set][q] = GL[gl_lane + 4 * kb * JP + 16 * NT + 4 * q];
            };
            gfetch(0, 0);
#pragma unroll
            for (int kb = 0; kb < KB1; ++kb) {
                const int set = kb & 1;
                if constexpr (NT > 0) acc1[0][0] = mfma16(Wf[0][kb], gt[set][0], acc1[0][0]);
                else acc1s[0][0] = mfma4(Wf[0][kb], gs[set][0], acc1s[0][0]);
                __builtin_amdgcn_sched_barrier(0);
                if (kb + 1 < KB1) gfetch(kb + 1, set ^ 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int p = 0; p < NA; ++p) {
#pragma unroll
                    for (int q = 0; q < NT; ++q)
                        if (p + q > 0) acc1[p][q] = mfma16(Wf[p][kb], gt[set][q], acc1[p][q]);
#pragma unroll
                    for (int q = 0; q < NR; ++q)
                        if (NT > 0 || p + q > 0) acc1s[p][q] = mfma4(Wf[p][kb], gs[set][q], acc1s[p][q]);
                }
            }
            // tile (p, q), register t of lane (x16, kq): T_k^T[a = 16 (at0 + p) + 4 t + kq][row = term JP + 16 q + x16]
            // strip (p, q), lane (i = kq, beta, j4):      T_k^T[a = 16 (at0 + p) + 4 beta + kq][row = term JP + 16 NT + 4 q + j4]
            uint32_t so = 0;
            if constexpr (WT) so = __builtin_amdgcn_readfirstlane((uint32_t)k * tkstep);
#pragma unroll
            for (int p = 0; p < NA; ++p) {
                if (p < ro.na) {
#pragma unroll
                    for (int q = 0; q < NT; ++q)
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            if (twt[p] >= 0 && 16 * (ro.at0 + p) + 4 * t + kq < 4 * KB2) TL[twt[p] + t * 4 * RP + 32 * q] = acc1[p][q][t];
                    if (tw[p] >= 0) {
#pragma unroll
                        for (int q = 0; q < NR; ++q) TL[tw[p] + 2 * (16 * NT + 4 * q)] = acc1s[p][q];
                    }
                }
                if constexpr (WT) {
#pragma unroll
                    for (int q = 0; q < NT; ++q)
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            st8s(rt, (tgt[p] != OOB_OFF && 16 * (ro.at0 + p) + 4 * t + kq < a.A && 16 * q + x16 < a.J)
                                         ? tgt[p] + (uint32_t)t * tastep + 128u * q : OOB_OFF, so, acc1[p][q][t]);
#pragma unroll
                    for (int q = 0; q < NR; ++q)
                        st8s(rt, (tg[p] != OOB_OFF && 16 * NT + 4 * q + j4 < a.J) ? tg[p] + 8u * (16 * NT + 4 * q) : OOB_OFF, so, acc1s[p][q]);
                }
            }
        } else if constexpr (WT) {
#pragma unroll
            for (int s = 0; s < NST; ++s) st8(rt, OOB_OFF, 0.0);         // keeps the store count of the wait below static
        }
        CS_STAMP(1);
        // E_k has landed (this wave's share; the barrier collects the others'); the T stores may still be in flight
        if (k + 1 < k_end) {
            if constexpr (WT) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NST + GMAX) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(GMAX) : "memory");
        } else {
            if constexpr (WT) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NST) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        CS_STAMP(2);
        if (!(CS_DIAG(16))) cf_barrier();            // B1: T_k and E_k are in LDS, nobody reads the G image any more
        CS_STAMP(3);
        if (k + 1 < k_end) {
            if constexpr (WT) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NST) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            g_store();                               // G_{k+1}: read after B2
        }

        // ---- phase B.  The fragments of k-block kb + 1 are requested BEHIND the first matrix instruction of kb: the wait in
        // front of that instruction then covers only reads issued a whole k-block earlier (the compiler waits with
        // lgkmcnt(0), so reads issued in front of it would be waited for at once), and they have the rest of the k-block to
        // land -- one wave alone on its SIMD hides its own LDS latency that way.
        // A wave with little work per k-block (a small rectangle, the strip column) shares its SIMD with one that has six
        // matrix instructions per k-block and never stalls: by age the big one would win every arbitration, and the small one's
        // work would run afterwards, alone, with its LDS latency exposed (measured: 16.3 k cycles per slice instead of 12.8 k).
        // With priority the small wave takes the pipe whenever it is ready and the big one fills the rest.
        constexpr bool SMALL = RT * CT <= 3;
        if constexpr (SMALL) { if (!(CS_DIAG(64))) __builtin_amdgcn_s_setprio(2); }
        // (an fp64 matrix instruction does not hide vector-ALU instructions of its own wave the way the narrow ones do -- measured:
        // a lone wave with ten address / spill-reload instructions per k-block runs at 84 instead of 64 cycles per matrix
        // instruction -- so the loops below advance one LDS pointer per operand and nothing else)
        if constexpr (RT * CT > 0) {
            double af[2][RT], bf[2][CT];
            int tl0 = tl_lane + 32 * ro.rt0, el0 = el_lane + 32 * ro.ct0;
            asm volatile("" : "+v"(tl0), "+v"(el0));
            const double *tp = TL + tl0, *ep = EL + el0;
            const int tstep = 4 * RP, estep = 4 * A2P;
            auto fetch = [&](int set) {              // fragments of the k-block the pointers stand at; advances them
#pragma unroll
                for (int r = 0; r < RT; ++r) af[set][r] = LDS_UNPAIRED(tp[32 * r]);
#pragma unroll
                for (int c = 0; c < CT; ++c) bf[set][c] = LDS_UNPAIRED(ep[32 * c]);
                tp += tstep;
                ep += estep;
            };
            auto step = [&](int set, bool more) {    // matrix instructions of one k-block, the next one's fragments behind the first
                acc2[0][0] = mfma16(af[set][0], bf[set][0], acc2[0][0]);
                __builtin_amdgcn_sched_barrier(0);
                if (more) fetch(set ^ 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < RT; ++r)
#pragma unroll
                    for (int c = 0; c < CT; ++c)
                        if (r + c > 0) acc2[r][c] = mfma16(af[set][r], bf[set][c], acc2[r][c]);
            };
            if (!(CS_DIAG(4))) {
                fetch(0);
                const int pairs = (KB2 - 1) >> 1;    // k-blocks 0 .. 2 pairs - 1 in pairs, each followed by another one
                for (int i = 0; i < pairs; ++i) {
                    step(0, true);
                    step(1, true);
                }
                if (KB2 & 1) step(0, false);
                else { step(0, true); step(1, false); }
            }
        }
        if constexpr (HS) {
            double as[2][SR], bs[2][NS];
            int ts0 = tl_lane, es0 = es_lane;
            asm volatile("" : "+v"(ts0), "+v"(es0));
            const double *tp = TL + ts0, *ep = EL + es0;
            const int tstep = 4 * RP, estep = 4 * A2P;
            auto fetch = [&](int set) {
#pragma unroll
                for (int r = 0; r < SR; ++r) as[set][r] = LDS_UNPAIRED(tp[32 * r]);
#pragma unroll
                for (int s = 0; s < NS; ++s) bs[set][s] = ep[8 * s];
                tp += tstep;
                ep += estep;
            };
            auto step = [&](int set, bool more) {
                acc2s[0][0] = mfma4(as[set][0], bs[set][0], acc2s[0][0]);
                __builtin_amdgcn_sched_barrier(0);
                if (more) fetch(set ^ 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < SR; ++r)
#pragma unroll
                    for (int s = 0; s < NS; ++s)
                        if (r + s > 0) acc2s[r][s] = mfma4(as[set][r], bs[set][s], acc2s[r][s]);
            };
            if (!(CS_DIAG(4))) {
                fetch(0);
                const int pairs = (KB2 - 1) >> 1;
                for (int i = 0; i < pairs; ++i) {
                    step(0, true);
                    step(1, true);
                }
                if (KB2 & 1) step(0, false);
                else { step(0, true); step(1, false); }
            }
        }
        if constexpr (SMALL) __builtin_amdgcn_s_setprio(0);
        CS_STAMP(4);
        if (!(CS_DIAG(16))) cf_barrier();            // B2: the T and E images may be overwritten
        CS_STAMP(5);
    }
#undef CS_STAMP

    // ---- partial results: slab[term][range][j][a']
    // (byte offsets as 32-bit numbers into one buffer descriptor, the row part once per row: this code runs once, cold,
    // at the very end of the workgroup -- 24 stores with their own 64-bit index products took 6-7 k cycles)
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.slab, (int64_t)a.nb * a.slab_t8);
    const uint32_t slab0 = (uint32_t)rr * a.slab_r8;
    auto rowoff = [&](int row) -> uint32_t {         // byte offset of slab[term of the row][rr][j][0], OOB_OFF = no such row
        const int lt2 = row / JP, j = row - lt2 * JP, t2 = g * a.tpw + lt2;
        return (lt2 < a.tpw && t2 < a.nb && j < a.J) ? slab0 + (uint32_t)t2 * a.slab_t8 + (uint32_t)(j * a.A2) * 8u : OOB_OFF;
    };
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint32_t ro8 = rowoff(16 * (ro.rt0 + r) + 4 * t + kq);
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const int col = 16 * (ro.ct0 + c) + x16;
                st8(rs, (ro8 != OOB_OFF && col < a.A2) ? ro8 + 8u * (uint32_t)col : OOB_OFF, acc2[r][c][t]);
            }
        }
    if constexpr (HS) {
        // 4x4x4 result: lane (i = kq, beta, j4) holds row 4 beta + i, column j4 of the strip
#pragma unroll
        for (int r = 0; r < SR; ++r) {
            const uint32_t ro8 = rowoff(16 * r + 4 * beta + kq);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int col = 16 * a.NNF + 4 * s + j4;
                st8(rs, (ro8 != OOB_OFF && col < a.A2) ? ro8 + 8u * (uint32_t)col : OOB_OFF, acc2s[r][s]);
            }
        }
    }
}

// the phase-B bodies a kernel instantiates (the host's wave table deals only these)
#define CS_RECT_BODIES(X) X(1, 1) X(1, 2) X(1, 3) X(2, 1) X(2, 2) X(2, 3) X(3, 1) X(3, 2) X(4, 1) X(5, 1)
#define CS_STRIP_BODIES(X) X(1, 1) X(2, 1) X(3, 1) X(4, 1) X(5, 1) X(6, 1) X(1, 2) X(2, 2) X(3, 2) X(4, 2) X(5, 2) X(6, 2)

// JS: 4-row strips per term (J <= 4 JS), KB1: k-blocks of phase A (K1 <= 4 KB1), NA: a-tiles per wave in phase A,
// WT: T is also written to memory
template <int JS, int KB1, int NA, bool WT>
__global__ __launch_bounds__(512, 2) void chain_sum_kernel(ChainSum a)
{
    extern __shared__ double cs_lds[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
#ifdef TTSK_LAB
    if (a.s.stamps && blockIdx.x == 0 && lane == 0) a.s.stamps[w * 8 + 6] = __builtin_amdgcn_s_memtime();      // kernel entry
#endif
    int g, rr;
    if (a.s.xcd_map) {
        // blocks b and b + 8 share an XCD: all term groups of a slice range on the same one (they read the same E_k)
        const int x = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int q = (int)(((uint32_t)idx * (uint32_t)a.s.inv_ng) >> 20);
        g = idx - q * a.s.ngroups;
        rr = x + 8 * q;
    } else {
        rr = (int)(((uint32_t)blockIdx.x * (uint32_t)a.s.inv_ng) >> 20);
        g = (int)blockIdx.x - rr * a.s.ngroups;
    }
    // The wave's role and pointers, read from the kernel-argument segment at a computed (uniform) offset: one scalar load
    // each.  (a.role[w] / a.W[tb] with a dynamic index make the compiler copy the whole argument block to scratch;
    // compare-and-select chains over the tables were 270 scalar instructions and 128 SGPRs, spilled.)
    typedef const __attribute__((address_space(4))) char *karg_t;
    typedef const __attribute__((address_space(4))) unsigned long long *karg64_t;
    const karg_t kp = (karg_t)__builtin_amdgcn_kernarg_segment_ptr();
    const unsigned long long rbits = *(karg64_t)(kp + offsetof(ChainSum, role) + 8 * w);
    ChainSumRole ro;
    ro.term = (unsigned char)(rbits & 0xff); ro.at0 = (unsigned char)((rbits >> 8) & 0xff); ro.na = (unsigned char)((rbits >> 16) & 0xff);
    ro.body = (unsigned char)((rbits >> 24) & 0xff); ro.rt0 = (unsigned char)((rbits >> 32) & 0xff); ro.ct0 = (unsigned char)((rbits >> 40) & 0xff);
    ro.pad0 = ro.pad1 = 0;
    // the term whose T columns this wave computes, and the term whose slice of X it brings in
    const int wpt = a.s.wpt;
    const int tb = g * a.s.tpw + ro.term, gt = g * a.s.tpw + (wpt == 8 ? 0 : (wpt == 4 ? w >> 2 : w >> 1));
    const double *Wterm = (const double *)*(karg64_t)(kp + offsetof(ChainSum, W) + 8 * (tb < a.s.nb ? tb : 0));
    const double *Xterm = (const double *)*(karg64_t)(kp + offsetof(ChainSum, X) + 8 * (gt < a.s.nb ? gt : 0));
#ifdef TTSK_LAB
    if (a.s.stamps && blockIdx.x == 0 && lane == 0) a.s.stamps[(4 * 8 + w) * 8 + 6] = __builtin_amdgcn_s_memtime();      // role and pointers picked
#endif
    CsPre<JS, KB1, NA> pre;
    cs_prologue<JS, KB1, NA, WT>(a.s, ro, Wterm, Xterm, cs_lds, g, rr, w, lane, pre);
#define CS_CALL(RT_, CT_, SR_, NS_) cs_wave<JS, KB1, NA, WT, RT_, CT_, SR_, NS_>(a.s, ro, pre, cs_lds, g, rr, w, lane)
#define CS_RECT(RT_, CT_) case 16 * RT_ + CT_: CS_CALL(RT_, CT_, 0, 0); break;
#define CS_STRIP(SR_, NS_) case 128 + 16 * SR_ + NS_: CS_CALL(0, 0, SR_, NS_); break;
    switch (ro.body) {
    CS_RECT_BODIES(CS_RECT)
    CS_STRIP_BODIES(CS_STRIP)
    default: CS_CALL(0, 0, 0, 0); break;
    }
#undef CS_CALL
#undef CS_RECT
#undef CS_STRIP
#ifdef TTSK_LAB
    if (a.s.stamps && blockIdx.x == 0 && lane == 0) a.s.stamps[w * 8 + 7] = __builtin_amdgcn_s_memtime();      // behind the partial results
#endif
}

// 1 = launched (the slab reduce included), 0 = shape not covered, < 0 = error.  T of the base arguments is ignored:
// the intermediate goes to Tint[b * t_b + (a * n + k) * t_ld + j] when Tint is given.
struct ChainSumArgs {
    ChainStepArgs s;
    double *Tint;
    int64_t t_b, t_ld, t_extent;
};
int chain_sum_try(const ChainSumArgs &c, int stream, hipStream_t st, bool force = false);
int launch_chain_sum_2(const ChainSum &a, bool wt, size_t lds, int grid, hipStream_t st);
int launch_chain_sum_4(const ChainSum &a, bool wt, size_t lds, int grid, hipStream_t st);

}  // namespace ttsk

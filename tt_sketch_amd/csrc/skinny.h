// Barrier-free fp64 MFMA kernels for the two product shapes the TT chains are made of
// (tensor_train_drm.py:79-141, tensor_train_sketch.py:21-35 in the reference):
//
//   "streamed x small"   C[m, j] = sum_k W[k, m] S[j, k]     j ~ 10^4..10^7 rows, K, P <= 128
//        chain GEMM1  T = R^T X^T / T = L^T X,  Psi = T R
//   "long-K"             C[m, n] = sum_k A[m, k] B[k, n]     M, N <= 128, K ~ 10^4..10^7
//        chain GEMM2  R' = sum_{q,k} T E,  L' = sum_{q,k} T D
//
// The generic kernel (gemm_kernel.h) stages both operands through LDS behind workgroup
// barriers; at K ~ 100 that is 3-4 barrier round trips of ~2 us each per tile and the matrix
// pipe idles.  Here every wave runs on its own: the operand fragments of v_mfma_f64_4x4x4 are
// exactly "lane (x = l & 15, k = l >> 4) holds element [x][k]", so a wave loads them straight
// from HBM/L2 into the MFMA operand registers with one 8-byte buffer load per lane (rows of
// any stride, out-of-range lanes read 0), keeps a ring of D k-blocks in flight and never
// synchronises with its neighbours.  The small operand W is read from LDS, staged once.
#pragma once
#include "gemm_kernel.h"

namespace ttsk {

constexpr int SK_MAXB = 32;  // problems of one shape per launch (one tensor of a batch each)

#ifndef TTSK_S_M16
#define TTSK_S_M16 1
#endif

struct SkinnyS {
    const double *W[SK_MAXB], *S[SK_MAXB];
    double *C[SK_MAXB];
    int nb, wpp;        // problems, workgroups per problem: workgroup x serves problem x / wpp
    int64_t w_k, w_m;   // W[k][m]
    int64_t s_j, s_k;   // S[j][k]
    // Two-level streamed index j = u V + v (u < U at stride s_u, v < V at stride s_j; U = 1: plain).
    // A 16-row block is then a (16 >> tvl) (u) x (1 << tvl) (v) tile.  tvl = 4 (16 consecutive v of
    // one u: the right chain's GEMM1 reads X[p'', k, :] for 16 p'' of one k and writes whole 128-byte
    // lines of T[q][k][p'']) measured best; 4 x 4 and 2 x 8 tiles read longer runs but write
    // 32- / 64-byte pieces and were 5-10 % slower end to end.
    int64_t s_u;
    int U, V, nbv, tvl;
    int64_t c_m, c_j;   // C[m][j]
    int64_t c_u;        // two-level streamed index: C[m][u][v] at m c_m + u c_u + v c_j
    int64_t J;
    int P, K, groups;
    int64_t s_extent, w_extent, c_extent;   // elements addressable from each base (all < 2^29)
    double alpha;
    int accumulate;
    int big;            // 1: 64-bit origins per row block / k-block / output tile (operands beyond 4 GB)
    long long *stamps;   // diagnostics (TTSK_SK_STAMPS): s_memtime at phase boundaries, 8 per workgroup
};

#define SK_STAMP(i) do { if (a.stamps && threadIdx.x == 0) a.stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define SK_STAMP_RT(i) do { if (a.stamps && threadIdx.x == 0) a.stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)

// A pointer picked from a kernel-argument array by a workgroup-uniform index: tell the compiler it
// is wave-uniform, or every buffer load through its resource descriptor gets a waterfall loop.
template <typename T>
__device__ __forceinline__ T *uniform_ptr(T *p)
{
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (T *)(((uint64_t)hi << 32) | lo);
}

__device__ __forceinline__ int64_t uniform_i64(int64_t v)
{
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

constexpr uint32_t OOB_OFF = 0xFFFFFFF0u;   // beyond any num_records: loads return 0, stores are dropped

// plain write-back stores: non-temporal and write-through (sc0 sc1) variants measured 10-35 % slower
__device__ __forceinline__ void st8(__amdgpu_buffer_rsrc_t r, uint32_t voff, double v)
{
    __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<v2i_t *>(&v), r, (int)voff, 0, 0);
}

template <int NPT, int NT, bool SH, int D, int RB, bool BIG, int STR>
__device__ __forceinline__ void skinny_s_wave_impl(const SkinnyS &a, const double *Wl, const int v, const int tile0,
                                                   const int tshared);

template <int NPT, int NT, bool SH, int D, int RB, int STR = 0>
__device__ __forceinline__ void skinny_s_wave(const SkinnyS &a, const double *Wl, const int v, const int tile0,
                                              const int tshared)
{
    if (a.big) skinny_s_wave_impl<NPT, NT, SH, D, RB, true, STR>(a, Wl, v, tile0, tshared);
    else skinny_s_wave_impl<NPT, NT, SH, D, RB, false, STR>(a, Wl, v, tile0, tshared);
}

// Stage W once per workgroup: Wl[k][m], zero beyond (K, P).  All loads of a thread are issued
// before the first LDS store (one memory round trip); masked elements read out of range = 0.
template <int NPT, int D>
__device__ __forceinline__ void skinny_s_stage(const SkinnyS &a, double *Wl)
{
    constexpr int LDW = ldmf(16 * NPT);
    const int tid = threadIdx.x;
    const double *Wp = uniform_ptr(a.W[blockIdx.x / a.wpp]);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(Wp, a.w_extent * 8);
    const int KB = (a.K + 3) >> 2, krows = ((KB + D - 1) / D) * D * 4;
    constexpr int HW = 8 * NPT;                       // pairs per LDS row
    const int total = krows * HW;
    const bool vec = a.w_m == 1 && !(a.w_k & 1) && !(a.P & 1) && !((uintptr_t)Wp & 15);
    constexpr int BATCH = 12;
    for (int e0 = tid; e0 < total; e0 += 512 * BATCH) {
        double2 v[BATCH];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int e = e0 + 512 * u;
            const int k = e / HW, m = 2 * (e - k * HW);
            const bool ok = e < total && k < a.K;
            const uint32_t o0 = (ok && m < a.P) ? (uint32_t)((k * a.w_k + m * a.w_m) * 8) : OOB_OFF;
            if (vec) {
                v[u] = ld16(rw, o0, 0);
            } else {
                const uint32_t o1 = (ok && m + 1 < a.P) ? (uint32_t)((k * a.w_k + (m + 1) * a.w_m) * 8) : OOB_OFF;
                v[u].x = ld8(rw, o0, 0);
                v[u].y = ld8(rw, o1, 0);
            }
        }
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int e = e0 + 512 * u;
            const int k = e / HW, m = 2 * (e - k * HW);
            if (e < total) *reinterpret_cast<double2 *>(&Wl[k * LDW + m]) = v[u];
        }
    }
}

// One workgroup = 8 free-running waves (two per SIMD, so one wave's DPP / LDS / wait slots are
// filled by the other's MFMAs) around one LDS copy of W.  A group is 4 (+1) blocks of 16 streamed
// rows: waves v and v+4 share row block 4g+v and split W's NPT column tiles ceil/floor; if SH,
// the tiles of a fifth block are dealt one per wave.  For C3 (NPT = 7) that is 9,9,9,8 tile
// strips on the four SIMDs and 1250 row blocks -> 250 workgroups = one round over 256 CUs.
// BIG: operands or outputs beyond the reach of one buffer descriptor + 32-bit offset (dense
// unfoldings); see the comment at `Sp` below.  Two variants because rebuilding a descriptor per load
// costs the rank-100 chain shapes 4-9 %.
// STR > 0: the wave's last tile is W's partial tile with at most 4 STR valid columns (P mod 16 <= 8);
// it is computed as STR strips of 4 by v_mfma_f64_4x4x4_4b -- the strip's 4 x 4 block of W against the
// four 4-row blocks of the streamed register, which is that instruction's natural operand layout --
// instead of one 16x16x4 of which 3/4 or 1/2 would be padding (100 = 6 x 16 + 4: the chain products
// are bounded by the matrix pipe, so the padding was 1/7 of GEMM1's run time).  Strip s lands in
// register s of the tile's accumulator, exactly where the 16x16x4 form keeps rows 4 s .. 4 s + 3.
template <int NPT, int NT, bool SH, int D, int RB, bool BIG, int STR>
__device__ __forceinline__ void skinny_s_wave_impl(const SkinnyS &a, const double *Wl, const int v, const int tile0,
                                                   const int tshared)
{
    constexpr int LDW = ldmf(16 * NPT);
    constexpr int NTC = NT ? NT : 1;
    const int lane = threadIdx.x & 63;
    const int x16 = lane & 15, kq = lane >> 4;
    const int KB = (a.K + 3) >> 2, ITER = (KB + D - 1) / D;
    const int prob = blockIdx.x / a.wpp, wg = blockIdx.x - prob * a.wpp;
    // Every load goes through a descriptor whose base is the origin of its row block and k-block
    // (64-bit scalar arithmetic), the per-lane offset only spans 16 rows x 4 k: streamed operands and
    // outputs of any size (a dense unfolding is 8.6 GB).  All masking is explicit, so the descriptors'
    // own range is just the OOB_OFF sentinel.
    const double *Sp = uniform_ptr(a.S[prob]);
    const int64_t kstep = 4 * a.s_k;
    const __amdgpu_buffer_rsrc_t rs_all = make_rsrc(Sp, a.s_extent * 8);      // !BIG: one descriptor, 32-bit offsets
    const int nkb_lane = (a.K - kq + 3) >> 2;          // k-blocks in which this lane's k = 4 kb + kq is < K
    const bool sh_on = SH && tshared < NPT;

    // streamed row c16 of row block rb: the element offset of its output column, or -1 past the end
    auto row_of = [&](int64_t rb, int c16, int64_t &in_off) -> int64_t {
        if (a.U == 1) {
            const int64_t j = rb * 16 + c16;
            in_off = j * a.s_j;
            return j < a.J ? j * a.c_j : -1;
        }
        const int bu = (int)((uint32_t)rb / (uint32_t)a.nbv), bv = (int)rb - bu * a.nbv;
        const int u = (bu << (4 - a.tvl)) + (c16 >> a.tvl), vv = (bv << a.tvl) + (c16 & ((1 << a.tvl) - 1));
        in_off = (int64_t)u * a.s_u + (int64_t)vv * a.s_j;
        return (u < a.U && vv < a.V) ? (int64_t)u * a.c_u + (int64_t)vv * a.c_j : -1;
    };
    // element offset of the first row of block rb (wave-uniform) ...
    auto blk_org = [&](int64_t rb) -> int64_t {
        if constexpr (!BIG) return 0;
        int64_t in0;
        (void)row_of(rb, 0, in0);
        return uniform_i64(in0);
    };
    // ... and this lane's byte offset from it (OOB_OFF: row past the end)
    auto lane_off = [&](int64_t rb, int64_t org) -> uint32_t {
        int64_t in_off;
        const int64_t j = row_of(rb, x16, in_off);
        return j >= 0 ? (uint32_t)((in_off - org + (int64_t)kq * a.s_k) * 8) : OOB_OFF;
    };
    // masked lanes (k >= K, group past the end) read offset OOB_OFF = 0.0
    auto fetch = [&](int64_t org, uint32_t off, int nkb, int kb) -> double {
        if constexpr (BIG) {
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(Sp + org + (int64_t)kb * kstep, (int64_t)OOB_OFF);
            return ld8(rs, (kb < nkb && off != OOB_OFF) ? off : OOB_OFF, 0);
        } else {
            return ld8(rs_all, (kb < nkb && off != OOB_OFF) ? off : OOB_OFF, (uint32_t)kb * (uint32_t)(kstep * 8));
        }
    };

    int g = wg;
    int nkb = g < a.groups ? nkb_lane : 0;
    int64_t ogA = blk_org((int64_t)g * RB + v), ogB = blk_org((int64_t)g * RB + 4);
    uint32_t voA = lane_off((int64_t)g * RB + v, ogA), voB = lane_off((int64_t)g * RB + 4, ogB);
    double ringA[D], ringB[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        if (NT) ringA[d] = fetch(ogA, voA, nkb, d);
        if (SH) ringB[d] = fetch(ogB, voB, sh_on ? nkb : 0, d);
    }

    skinny_s_stage<NPT, D>(a, const_cast<double *>(Wl));   // after the ring loads: both round trips overlap
    SK_STAMP(1);
    __syncthreads();   // every wave reaches this exactly once
    SK_STAMP(2);

    const double *wl_lane = Wl + kq * LDW + x16 + 16 * tile0;
    const int soffS = 16 * (sh_on ? tshared : 0) - 16 * tile0;
    const int soff4 = x16 & ~3;                   // strip operand: column 4 q + (x16 & 3) of the partial tile

    v4d accA[NTC], accB;      // [t]: the four 4x4x4 accumulators of a tile, or the 4 result registers of 16x16x4
#pragma unroll
    for (int p = 0; p < NTC; ++p)
#pragma unroll
        for (int t = 0; t < 4; ++t) accA[p][t] = 0.0;
#pragma unroll
    for (int t = 0; t < 4; ++t) accB[t] = 0.0;

    double *Cp = uniform_ptr(a.C[prob]);
    // acc[t] at lane (i = l>>4, beta = (l>>2)&3, j4 = l&3) is D[4 beta + i][4((beta+t)&3) + j4]
    // (16x16x4 form: register t of lane l is D[4 t + (l >> 4)][l & 15])
    int e_m[4], e_j[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        e_m[t] = TTSK_S_M16 ? 4 * t + (lane >> 4) : 4 * ((lane >> 2) & 3) + (lane >> 4);
        e_j[t] = TTSK_S_M16 ? (lane & 15) : 4 * ((((lane >> 2) & 3) + t) & 3) + (lane & 3);
    }
    while (g < a.groups) {
        uint32_t nA = voA, nB = voB;
        int64_t ngA = ogA, ngB = ogB;
        int nkb_n = nkb;
        // K is padded to ITER * D k-blocks (zero rows of W, masked loads of S): one loop shape for
        // every iteration keeps the k-blocks overlapping across the whole group
        for (int it = 0; it < ITER; ++it) {
            int itn = it + 1;
            if (itn == ITER) {
                itn = 0;
                const int gn = g + a.wpp;
                ngA = blk_org((int64_t)gn * RB + v);
                ngB = blk_org((int64_t)gn * RB + 4);
                nA = lane_off((int64_t)gn * RB + v, ngA);
                nB = lane_off((int64_t)gn * RB + 4, ngB);
                nkb_n = gn < a.groups ? nkb_lane : 0;
            }
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int kb = it * D + d;
                const double *wk = wl_lane + kb * 4 * LDW;
                double af[NTC], as = 0.0, sf[STR ? STR : 1];
                constexpr int NFULL = STR ? NT - 1 : NT;
                constexpr int LASTT = NT > 0 ? NT - 1 : 0;       // the partial tile (strips)
#pragma unroll
                for (int p = 0; p < NFULL; ++p) af[p] = wk[16 * p];
#pragma unroll
                for (int q = 0; q < STR; ++q) sf[q] = wk[16 * LASTT + 4 * q - soff4];
                if (SH) as = wk[soffS];
                double rA[4], rB[4];
                const double sA = NT ? ringA[d] : 0.0, sB = SH ? ringB[d] : 0.0;
                if (NT) {
                    if (!TTSK_S_M16) rot4(ringA[d], rA);
                    ringA[d] = fetch(ngA, nA, nkb_n, itn * D + d);
                }
                if (SH) {
                    if (!TTSK_S_M16) rot4(ringB[d], rB);
                    ringB[d] = fetch(ngB, nB, sh_on ? nkb_n : 0, itn * D + d);
                }
                if (TTSK_S_M16) {
#pragma unroll
                    for (int p = 0; p < NFULL; ++p) accA[p] = mfma16(af[p], sA, accA[p]);
#pragma unroll
                    for (int q = 0; q < STR; ++q) accA[LASTT][q] = mfma4(sf[q], sA, accA[LASTT][q]);
                    if (SH) accB = mfma16(as, sB, accB);
                } else {
#pragma unroll
                    for (int p = 0; p < NT; ++p)
#pragma unroll
                        for (int t = 0; t < 4; ++t) accA[p][t] = mfma4(af[p], rA[t], accA[p][t]);
                    if (SH) {
#pragma unroll
                        for (int t = 0; t < 4; ++t) accB[t] = mfma4(as, rB[t], accB[t]);
                    }
                }
            }
        }
        SK_STAMP(3);
        // epilogue: branch-free buffer stores, masked lanes go out of range; one descriptor per
        // (row block, tile of W) whose base is the output element of the block's first row
        {
            int64_t dummy;
            const int64_t outA = BIG ? uniform_i64(row_of((int64_t)g * RB + v, 0, dummy)) : 0;
            const int64_t outB = BIG ? uniform_i64(row_of((int64_t)g * RB + 4, 0, dummy)) : 0;
            uint32_t offA[4], offB[4];       // this lane's byte offset inside a tile, per register
            bool okA[4], okB[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int64_t ja = row_of((int64_t)g * RB + v, e_j[t], dummy);
                const int64_t jb = row_of((int64_t)g * RB + 4, e_j[t], dummy);
                okA[t] = ja >= 0;
                okB[t] = sh_on && jb >= 0;
                offA[t] = (uint32_t)((ja - outA + (int64_t)e_m[t] * a.c_m) * 8);
                offB[t] = (uint32_t)((jb - outB + (int64_t)e_m[t] * a.c_m) * 8);
            }
            __amdgpu_buffer_rsrc_t rcA[NTC], rcB;
            const uint32_t tile_step = BIG ? 0u : (uint32_t)(16 * a.c_m * 8);      // !BIG: tiles by offset
#pragma unroll
            for (int p = 0; p < NT; ++p)
                rcA[p] = BIG ? make_rsrc(Cp + (outA < 0 ? 0 : outA) + (int64_t)16 * (tile0 + p) * a.c_m, (int64_t)OOB_OFF)
                             : make_rsrc(Cp, a.c_extent * 8);
            rcB = BIG ? make_rsrc(Cp + (outB < 0 ? 0 : outB) + (int64_t)16 * tshared * a.c_m, (int64_t)OOB_OFF)
                      : make_rsrc(Cp, a.c_extent * 8);
            uint32_t offs[NTC][4], offt[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int p = 0; p < NT; ++p) {
                    offs[p][t] = (okA[t] && 16 * (tile0 + p) + e_m[t] < a.P) ? offA[t] + (tile0 + p) * tile_step : OOB_OFF;
                    accA[p][t] *= a.alpha;
                }
                offt[t] = (okB[t] && 16 * tshared + e_m[t] < a.P) ? offB[t] + tshared * tile_step : OOB_OFF;
                accB[t] *= a.alpha;
            }
            if (a.accumulate) {
                double old[NTC][4], olds[4];
#pragma unroll
                for (int p = 0; p < NT; ++p)
#pragma unroll
                    for (int t = 0; t < 4; ++t) old[p][t] = ld8(rcA[p], offs[p][t], 0);
                if (SH) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) olds[t] = ld8(rcB, offt[t], 0);
                }
#pragma unroll
                for (int p = 0; p < NT; ++p)
#pragma unroll
                    for (int t = 0; t < 4; ++t) accA[p][t] += old[p][t];
                if (SH) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) accB[t] += olds[t];
                }
            }
#pragma unroll
            for (int p = 0; p < NT; ++p)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    st8(rcA[p], offs[p][t], accA[p][t]);
                    accA[p][t] = 0.0;
                }
            if (SH) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    st8(rcB, offt[t], accB[t]);
                    accB[t] = 0.0;
                }
            }
        }
        SK_STAMP(4);
        g += a.wpp;
        voA = nA;
        voB = nB;
        ogA = ngA;
        ogB = ngB;
        nkb = nkb_n;
    }
    if (a.stamps) {
        __builtin_amdgcn_s_waitcnt(0);
        SK_STAMP(5);
        SK_STAMP_RT(7);
    }
}

// MODE 0: groups of 4 row blocks, waves v and v+4 split the tiles of block v.
// MODE 1: the same plus a fifth block whose tiles are dealt one per wave (fills 256 CUs in one
//         round at the single-tensor C3 shapes).
// MODE 2: groups of 8 row blocks, every wave owns one block against all NPT tiles: no fragment is
//         loaded twice.  The vector memory pipe (one tag lookup per touched line, 16 lines per
//         k-fast load) is what saturates in these kernels, so this is the mode of choice whenever
//         the group count still fills the CUs.
template <int NPT, int MODE, int D, int STR = 0>
__global__ __launch_bounds__(512) void skinny_s_kernel(SkinnyS a)
{
    static_assert(TTSK_S_M16 || STR == 0, "strips need the 16x16x4 accumulator layout");
    extern __shared__ double Wl[];
    const int tid = threadIdx.x;
    SK_STAMP(0);
    SK_STAMP_RT(6);
    const int w = tid >> 6, v = w & 3;
    constexpr int H0 = (NPT + 1) / 2, H1 = NPT / 2;
    if constexpr (MODE == 2) {
        skinny_s_wave<NPT, NPT, false, D, 8, STR>(a, Wl, w, 0, 0);
    } else {
        constexpr bool SH = MODE == 1;
        // the partial tile is the last one: it belongs to the upper waves (to the lower ones if NPT = 1)
        if (w < 4) skinny_s_wave<NPT, H0, SH, D, SH ? 5 : 4, H1 == 0 ? STR : 0>(a, Wl, v, 0, w);
        else       skinny_s_wave<NPT, H1, SH, D, SH ? 5 : 4, H1 == 0 ? 0 : STR>(a, Wl, v, H0, w);
    }
}

// ---- long-K products -------------------------------------------------------------------
// C[m, n] = sum_kappa A[kappa][m] B[kappa][n] with m and n contiguous in memory (both chain
// GEMM2s after the driver lays T out that way), kappa = (ko, ki) with per-operand strides.
// One workgroup = one chunk of kappa against the whole (<= 128 x 128) output; its 8 waves tile
// the output as 2 row halves x 4 column strips and stream exactly the A rows and B columns of
// their tiles straight into MFMA operand registers (ring of D kappa-blocks in flight, no LDS, no
// barriers).  The vector memory pipe is the scarce unit here (every MFMA operand comes from
// memory), so tiles are fetched in pairs: one 16-byte load per lane brings rows 2x and 2x+1 of a
// 32-row block, i.e. the fragments of the "even rows" and the "odd rows" tile at once, 256
// contiguous bytes per kappa.  Partial outputs go to slab[chunk][m][n]; skinny_r_reduce sums them.
#ifndef TTSK_R_M16
#define TTSK_R_M16 1
#endif
#ifndef TTSK_R_DEPTH
#define TTSK_R_DEPTH 4     // kappa-blocks in flight per wave (8 measured no faster, 230 VGPRs)
#endif

struct SkinnyR {
    const double *A[SK_MAXB], *B[SK_MAXB];
    double *slab;       // [problem][chunk][m][n]
    int nb, chunks;     // workgroup x serves chunk x % chunks of (problem, row tile) x / chunks
    int64_t a_ko, a_ki, b_ko, b_ki;
    int64_t Ki, K, chunk;
    int64_t a_extent, b_extent;
    int M, N;           // rows of one row tile (<= 128), columns (<= 128)
    // A's row dimension may be longer than 128: m_tiles tiles of M rows, Mtot rows in all (the dense
    // unfoldings: 20 x K times K x 4096 ... ).  With a single-level kappa (`rebase`) every workgroup
    // addresses its operands from its own chunk / tile origin, so operands beyond 4 GB are fine.
    int m_tiles, rebase;
    int64_t Mtot;
    // An operand whose rows are not contiguous, or not pairable (odd extent / stride, unaligned base),
    // is "generic": plain 16-row tiles, one 8-byte load per tile and lane at row stride a_m / b_n,
    // each tile through its own descriptor (the rows of a k-contiguous unfolding are 128 MB apart).
    int a_gen, b_gen;
    int64_t a_m, b_n;
    long long *stamps;
};

__device__ __forceinline__ void st16(double *p, double x, double y)
{
    *reinterpret_cast<double2 *>(p) = make_double2(x, y);
}

// Wave tile: rows [row0, row0 + 32 PA + 16 SA) x columns [col0, col0 + 32 PB + 16 SB):
// PA / PB paired 32-blocks and an optional single 16-block per side.
// NSUB > 1: the wave works on the sub-th of NSUB equal pieces of the workgroup's chunk (small
// outputs: every wave owns the whole tile set and its own stretch of kappa, one slab per wave).
// GEN = 2: generic tiles whose kappa is contiguous on BOTH sides (the first-mode unfolding of a dense
// tensor against a k-contiguous sketching matrix): one 16-byte load per tile and lane brings kappa =
// 8 i + 2 kq and 8 i + 2 kq + 1, i.e. the fragments of two k-blocks -- the x halves are one k-block
// (kappa 0, 2, 4, 6 of the eight), the y halves the other; any split of the sum is fine as long as both
// operands use the same one.  Half the vector-memory instructions and 64 instead of 32 contiguous bytes
// per row and instruction.
template <int PA, int SA, int PB, int SB, int D, int GEN, int NSUB>
__device__ __forceinline__ void skinny_r_wave_impl(const SkinnyR &a, const int row0, const int col0, const int sub)
{
    constexpr bool GK = GEN == 2;
    static_assert(!GK || TTSK_R_M16, "16-byte generic tiles use the 16x16x4 form");
    constexpr int TM = 2 * PA + SA, TN = 2 * PB + SB;
    const int lane = threadIdx.x & 63;
    const int x16 = lane & 15, kq = lane >> 4;
    SK_STAMP(0);
    const int pt = blockIdx.x / a.chunks, ci = blockIdx.x - pt * a.chunks;
    const int prob = pt / a.m_tiles, mtile = pt - prob * a.m_tiles;
    const int64_t piece = NSUB > 1 ? (((a.chunk + NSUB - 1) / NSUB + 3) & ~(int64_t)3) : a.chunk;
    const int64_t k0 = (int64_t)ci * a.chunk + (NSUB > 1 ? sub * piece : 0);
    const int64_t kend = ((int64_t)ci + 1) * a.chunk < a.K ? ((int64_t)ci + 1) * a.chunk : a.K;
    const int64_t len = kend - k0 < 0 ? 0 : (kend - k0 < piece ? kend - k0 : piece);
    // GK: a "k-block" below is eight kappa (two MFMA k-blocks), this lane's are 8 i + 2 kq (+ 1); len is even
    const int nkb_lane = GK ? (int)(len > 2 * kq ? (len - 2 * kq + 7) >> 3 : 0) : (int)(len > kq ? (len - kq + 3) >> 2 : 0);
    const int KB = GK ? (int)((len + 7) >> 3) : (int)((len + 3) >> 2);
    const int64_t tile_row0 = (int64_t)mtile * a.M;                 // first row of this tile in A
    const int rows_here = (int)(a.Mtot - tile_row0 < a.M ? a.Mtot - tile_row0 : a.M);
    // operand origins of this workgroup (rebase: + the chunk's kappa offset; every out-of-range lane is
    // masked explicitly, the descriptor's own range check is then only the OOB_OFF sentinel)
    const int64_t a_org = tile_row0 * (GEN ? a.a_m : 1) + (a.rebase ? k0 * a.a_ki : 0), b_org = a.rebase ? k0 * a.b_ki : 0;
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(uniform_ptr(a.A[prob]) + a_org, a.rebase ? (int64_t)OOB_OFF : (a.a_extent - tile_row0) * 8);
    const __amdgpu_buffer_rsrc_t rb = make_rsrc(uniform_ptr(a.B[prob]) + b_org, a.rebase ? (int64_t)OOB_OFF : a.b_extent * 8);

    // per-lane walk over kappa = k0 + kq + 4 i: byte offsets of A's and B's kappa part
    const int64_t kap = (a.rebase ? 0 : k0) + (GK ? 2 * kq : kq);
    const int64_t ko0 = (a.rebase || a.Ki == a.K) ? 0 : kap / a.Ki;
    int ki = (int)(kap - ko0 * a.Ki);
    uint32_t offA = (uint32_t)((ko0 * a.a_ko + (int64_t)ki * a.a_ki) * 8);
    uint32_t offB = (uint32_t)((ko0 * a.b_ko + (int64_t)ki * a.b_ki) * 8);
    const uint32_t stepA = (uint32_t)((GK ? 8 : 4) * a.a_ki * 8), stepB = (uint32_t)((GK ? 8 : 4) * a.b_ki * 8);
    const uint32_t wrapA = (uint32_t)((a.a_ko - a.Ki * a.a_ki) * 8), wrapB = (uint32_t)((a.b_ko - a.Ki * a.b_ki) * 8);
    const int Ki = (int)a.Ki;
    // column (element) offsets inside a kappa row; elements past M / N are masked per lane
    uint32_t rowA[PA + SA], colB[PB + SB];
#pragma unroll
    for (int p = 0; p < PA; ++p) rowA[p] = row0 + 32 * p + 2 * x16 < rows_here ? (uint32_t)(row0 + 32 * p + 2 * x16) * 8u : OOB_OFF;
    if (SA) rowA[PA] = row0 + 32 * PA + x16 < rows_here ? (uint32_t)(row0 + 32 * PA + x16) * 8u : OOB_OFF;
#pragma unroll
    for (int q = 0; q < PB; ++q) colB[q] = col0 + 32 * q + 2 * x16 < a.N ? (uint32_t)(col0 + 32 * q + 2 * x16) * 8u : OOB_OFF;
    if (SB) colB[PB] = col0 + 32 * PB + x16 < a.N ? (uint32_t)(col0 + 32 * PB + x16) * 8u : OOB_OFF;

    // generic operands: per-tile descriptors and validity
    __amdgpu_buffer_rsrc_t raT[TM], rbT[TN];
    bool okA[TM], okB[TN];
    const uint32_t xA = (uint32_t)((int64_t)x16 * a.a_m * 8), xB = (uint32_t)((int64_t)x16 * a.b_n * 8);
    if constexpr (GEN) {
#pragma unroll
    for (int p = 0; p < TM; ++p) {
        raT[p] = make_rsrc(uniform_ptr(a.A[prob] + a_org + (int64_t)(row0 + 16 * p) * a.a_m),
                           a.rebase ? (int64_t)OOB_OFF : (a.a_extent - tile_row0 * a.a_m - (int64_t)(row0 + 16 * p) * a.a_m) * 8);
        okA[p] = row0 + 16 * p + x16 < rows_here;
    }
#pragma unroll
    for (int q = 0; q < TN; ++q) {
        rbT[q] = make_rsrc(uniform_ptr(a.B[prob] + b_org + (int64_t)(col0 + 16 * q) * a.b_n),
                           a.rebase ? (int64_t)OOB_OFF : (a.b_extent - (int64_t)(col0 + 16 * q) * a.b_n) * 8);
        okB[q] = col0 + 16 * q + x16 < a.N;
    }
    }

    double ringA[D][TM], ringB[D][TN];
    double ringA2[GK ? D : 1][TM], ringB2[GK ? D : 1][TN];      // GK: the odd kappa of each 16-byte load
    int kb_load = 0;
    auto issue = [&](int d) {
        const bool ok = kb_load < nkb_lane;
        if constexpr (GK) {
#pragma unroll
            for (int p = 0; p < TM; ++p) {
                const double2 v = ld16(raT[p], (ok && okA[p]) ? xA + offA : OOB_OFF, 0);
                ringA[d][p] = v.x;
                ringA2[d][p] = v.y;
            }
        } else if constexpr (GEN) {
#pragma unroll
            for (int p = 0; p < TM; ++p) ringA[d][p] = ld8(raT[p], (ok && okA[p]) ? xA + offA : OOB_OFF, 0);
        } else {
#pragma unroll
            for (int p = 0; p < PA; ++p) {
                const double2 v = ld16(ra, (ok && rowA[p] != OOB_OFF) ? rowA[p] + offA : OOB_OFF, 0);
                ringA[d][2 * p] = v.x;
                ringA[d][2 * p + 1] = v.y;
            }
            if (SA) ringA[d][2 * PA] = ld8(ra, (ok && rowA[PA] != OOB_OFF) ? rowA[PA] + offA : OOB_OFF, 0);
        }
        if constexpr (GK) {
#pragma unroll
            for (int q = 0; q < TN; ++q) {
                const double2 v = ld16(rbT[q], (ok && okB[q]) ? xB + offB : OOB_OFF, 0);
                ringB[d][q] = v.x;
                ringB2[d][q] = v.y;
            }
        } else if constexpr (GEN) {
#pragma unroll
            for (int q = 0; q < TN; ++q) ringB[d][q] = ld8(rbT[q], (ok && okB[q]) ? xB + offB : OOB_OFF, 0);
        } else {
#pragma unroll
            for (int q = 0; q < PB; ++q) {
                const double2 v = ld16(rb, (ok && colB[q] != OOB_OFF) ? colB[q] + offB : OOB_OFF, 0);
                ringB[d][2 * q] = v.x;
                ringB[d][2 * q + 1] = v.y;
            }
            if (SB) ringB[d][2 * PB] = ld8(rb, (ok && colB[PB] != OOB_OFF) ? colB[PB] + offB : OOB_OFF, 0);
        }
        ++kb_load;
        ki += GK ? 8 : 4;
        offA += stepA;
        offB += stepB;
        const bool wrap = ki >= Ki;
        ki = wrap ? ki - Ki : ki;
        offA += wrap ? wrapA : 0u;
        offB += wrap ? wrapB : 0u;
    };
#pragma unroll
    for (int d = 0; d < D; ++d) issue(d);

    v4d acc[TM][TN];
#pragma unroll
    for (int p = 0; p < TM; ++p)
#pragma unroll
        for (int q = 0; q < TN; ++q)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[p][q][t] = 0.0;

    SK_STAMP(1);
    auto kblock = [&](int d, bool more) {
        double af[TM], bf[TN], rB[TN][4], af2[TM], bf2[TN];
#pragma unroll
        for (int p = 0; p < TM; ++p) {
            af[p] = ringA[d][p];
            if (GK) af2[p] = ringA2[d][p];
        }
#pragma unroll
        for (int q = 0; q < TN; ++q) {
            bf[q] = ringB[d][q];
            if (GK) bf2[q] = ringB2[d][q];
            if (!TTSK_R_M16) rot4(ringB[d][q], rB[q]);
        }
        if (more) issue(d);
#pragma unroll
        for (int p = 0; p < TM; ++p)
#pragma unroll
            for (int q = 0; q < TN; ++q) {
                if (TTSK_R_M16) {
                    acc[p][q] = mfma16(af[p], bf[q], acc[p][q]);
                    if (GK) acc[p][q] = mfma16(af2[p], bf2[q], acc[p][q]);
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[p][q][t] = mfma4(af[p], rB[q][t], acc[p][q][t]);
                }
            }
    };
    const int FULLIT = KB / D, TAIL = KB - FULLIT * D;
    for (int it = 0; it < FULLIT; ++it) {
#pragma unroll
        for (int d = 0; d < D; ++d) kblock(d, true);
    }
#pragma unroll
    for (int d = 0; d < D - 1; ++d)
        if (d < TAIL) kblock(d, false);
    SK_STAMP(3);
    // acc[p][q][t] at lane (i = l>>4, beta = (l>>2)&3, j4 = l&3) is fragment element
    // (r = 4 beta + i, c = 4((beta+t)&3) + j4); pair fragments 2p / 2p+1 are rows 32p + 2r + {0,1}
    // (16x16x4 form: register t of lane l is fragment element (r = 4 t + (l >> 4), c = l & 15))
    double *slab = a.slab + ((int64_t)blockIdx.x * NSUB + sub) * a.M * a.N;
#pragma unroll
    for (int p = 0; p < TM; ++p) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int r16 = TTSK_R_M16 ? 4 * t + (lane >> 4) : 4 * ((lane >> 2) & 3) + (lane >> 4);
            const int c16 = TTSK_R_M16 ? (lane & 15) : 4 * ((((lane >> 2) & 3) + t) & 3) + (lane & 3);
            const int m = GEN ? row0 + 16 * p + r16
                                  : (p < 2 * PA ? row0 + 32 * (p >> 1) + 2 * r16 + (p & 1) : row0 + 32 * PA + r16);
            if (m < rows_here) {
                double *srow = slab + (int64_t)m * a.N;
                if constexpr (GEN) {
#pragma unroll
                    for (int q = 0; q < TN; ++q) {
                        const int n = col0 + 16 * q + c16;
                        if (n < a.N) srow[n] = acc[p][q][t];
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < PB; ++q) {
                        const int n = col0 + 32 * q + 2 * c16;     // even N (16-byte loads) => n + 1 < N too
                        if (n < a.N) st16(srow + n, acc[p][2 * q][t], acc[p][2 * q + 1][t]);
                    }
                    if (SB) {
                        const int n = col0 + 32 * PB + c16;
                        if (n < a.N) srow[n] = acc[p][2 * PB][t];
                    }
                }
            }
        }
    }
    SK_STAMP(4);
    if (a.stamps) {
        __builtin_amdgcn_s_waitcnt(0);
        SK_STAMP(5);
    }
}

template <int PA, int SA, int PB, int SB, int D, int NSUB = 1>
__device__ __forceinline__ void skinny_r_wave(const SkinnyR &a, const int row0, const int col0, const int sub = 0)
{
    // two straight-line variants (a branch inside the k-block loop would break its software pipeline)
    if (a.a_gen == 2) skinny_r_wave_impl<PA, SA, PB, SB, D, 2, NSUB>(a, row0, col0, sub);
    else if (a.a_gen) skinny_r_wave_impl<PA, SA, PB, SB, D, 1, NSUB>(a, row0, col0, sub);
    else skinny_r_wave_impl<PA, SA, PB, SB, D, 0, NSUB>(a, row0, col0, sub);
}

// outputs of at most SKR_KSPLIT_TILES tiles: kappa split over the 8 waves instead of the tiles
constexpr int SKR_KSPLIT_TILES = 12;

// 8 waves over the (<= 8) x (<= 8) tiles of the output: 2 row halves x 4 column strips, or -- when
// there are at most 4 column tiles -- 4 row quarters x 2 column strips, which keeps all waves busy
// and needs 2 instead of 3 loads per 4 tiles on the narrow shapes (left chain: 7 x 4 tiles).
template <int NMT, int NNT, int D>
__global__ __launch_bounds__(512) void skinny_r_kernel(SkinnyR a)
{
    const int w = threadIdx.x >> 6, s = w & 3, h = w >> 2;
    if constexpr (NMT * NNT <= SKR_KSPLIT_TILES) {
        // no operand is loaded twice and 7 loads feed 12 tiles instead of 2 feeding 1
        skinny_r_wave<NMT / 2, NMT % 2, NNT / 2, NNT % 2, D, 8>(a, 0, 0, w);
    } else if constexpr (NNT <= 4) {
        constexpr int TQ = (NMT + 3) / 4;                    // row tiles per quarter (1 or 2)
        constexpr int TS = (NNT + 1) / 2;                    // column tiles per strip (1 or 2)
        const int rows = NMT - s * TQ < TQ ? NMT - s * TQ : TQ;          // <= 0: nothing for this wave
        const int cols = NNT - h * TS < TS ? NNT - h * TS : TS;
        if (rows <= 0 || cols <= 0) return;
        const int row0 = 16 * s * TQ, col0 = 16 * h * TS;
        if constexpr (TQ == 2 && TS == 2) {
            if (rows == 2 && cols == 2) skinny_r_wave<1, 0, 1, 0, D>(a, row0, col0);
            else if (rows == 2) skinny_r_wave<1, 0, 0, 1, D>(a, row0, col0);
            else if (cols == 2) skinny_r_wave<0, 1, 1, 0, D>(a, row0, col0);
            else skinny_r_wave<0, 1, 0, 1, D>(a, row0, col0);
        } else if constexpr (TQ == 2) {
            if (rows == 2) skinny_r_wave<1, 0, 0, 1, D>(a, row0, col0);
            else skinny_r_wave<0, 1, 0, 1, D>(a, row0, col0);
        } else if constexpr (TS == 2) {
            if (cols == 2) skinny_r_wave<0, 1, 1, 0, D>(a, row0, col0);
            else skinny_r_wave<0, 1, 0, 1, D>(a, row0, col0);
        } else {
            skinny_r_wave<0, 1, 0, 1, D>(a, row0, col0);
        }
    } else {
        constexpr int H0 = (NMT + 1) / 2, H1 = NMT / 2;          // tiles of the two row halves
        constexpr int TN = (NNT + 3) / 4;                        // tiles per column strip (2 here)
        constexpr int FULL = NNT / TN, REST = NNT - FULL * TN;   // FULL strips of TN tiles, then one of REST
        if (h == 0) {
            if (s < FULL) skinny_r_wave<H0 / 2, H0 % 2, TN / 2, TN % 2, D>(a, 0, 16 * s * TN);
            if constexpr (REST > 0) {
                if (s == FULL) skinny_r_wave<H0 / 2, H0 % 2, REST / 2, REST % 2, D>(a, 0, 16 * s * TN);
            }
        } else {
            if constexpr (H1 > 0) {
                if (s < FULL) skinny_r_wave<H1 / 2, H1 % 2, TN / 2, TN % 2, D>(a, 16 * H0, 16 * s * TN);
                if constexpr (REST > 0) {
                    if (s == FULL) skinny_r_wave<H1 / 2, H1 % 2, REST / 2, REST % 2, D>(a, 16 * H0, 16 * s * TN);
                }
            }
        }
    }
}

// sum of the per-chunk partial results of the long-K kernel and of the fused chain step (skinny.hip)
struct ReduceOut { double *C[SK_MAXB]; };
// out[m, n] (+)= alpha sum_c slab[c][m][n] for nprob x m_tiles (problem, row tile) blocks of slabs behind one another
int launch_r_reduce(hipStream_t st, const double *slab, int chunks, int M, int N, int m_tiles, int64_t Mtot, const ReduceOut &ro,
                    int nprob, int64_t c_m, int64_t c_n, double alpha, int accumulate);

// 1 = launched, 0 = shape not covered (caller falls through to the generic kernel), < 0 = error
int skinny_try(const ttsk_gemm_desc &d, const double *A, const double *B, double *C, const double *k_scale,
               int stream, hipStream_t st);
// small products (small.hip): nb pointer triples, or one triple with a uniformly strided d.batch
int small_try_batch(const ttsk_gemm_desc &d, int nb, const double *const *A, const double *const *B, double *const *C,
                    int stream, hipStream_t st);
// the same product for nb <= SK_MAXB problems of one shape (the tensors of a batch) in one launch
int skinny_try_batch(const ttsk_gemm_desc &d, int nb, const double *const *A, const double *const *B,
                     double *const *C, int stream, hipStream_t st);

}  // namespace ttsk

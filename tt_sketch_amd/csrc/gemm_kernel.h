// fp64 MFMA contraction kernel template (see gemm.hip for the host side).
#pragma once
#include <cstdint>
#include "common.h"

namespace ttsk {

constexpr int BK = 32;        // K depth of one staged tile
constexpr int LDKF = BK + 2;  // "k-fast" LDS layout tile[x][k], ld == 2 (mod 32)

__host__ __device__ constexpr int ldmf(int bx) { return (bx % 32 == 16) ? bx : bx + 16; }  // == 16 (mod 32)

struct KMap {
    int64_t Ko, Ki, s_ko, s_ki;
    __device__ __forceinline__ int64_t off(int64_t kk) const
    {
        if (Ko == 1) return kk * s_ki;
        int64_t ko = kk / Ki;
        return ko * s_ko + (kk - ko * Ki) * s_ki;
    }
};

// --- fp64 matrix instruction ------------------------------------------------------------
// Measured on MI355X (scratch probes): v_mfma_f64_16x16x4_f64 peaks at 49 TF/s
// (32 TF/s with one wave per SIMD) while v_mfma_f64_4x4x4_4b_f64 reaches 65 TF/s (58 at one
// wave per SIMD).  The 4x4x4 form multiplies, for each of the four lane sub-groups beta
// (lanes 16k + 4 beta + {0..3}), the 4x4 blocks A[4beta+i][k] (lane 4beta+i+16k) and
// B[k][4beta+j] (lane 4beta+j+16k) into D[4beta+i][4beta+j] at lane 16i+4beta+j, i.e. the
// diagonal 4x4 blocks of the 16x16 product of the SAME operand registers the 16x16x4 form
// takes.  Rotating one operand by 4, 8, 12 lanes inside each row of 16 (DPP row_ror) and
// issuing the instruction four times yields the full 16x16x4 product in 4 accumulators:
//   acc[t] at lane L (i = L>>4, beta = (L>>2)&3, j = L&3)
//     ROTB:  D[4 beta + i][4 ((beta+t)&3) + j]      ROTA:  D[4 ((beta+t)&3) + i][4 beta + j]
template <int CTRL>
__device__ __forceinline__ double dpp_row(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
// value of lane (l + 4t) mod 16 of the same row of 16: row_ror:(16-4t)
__device__ __forceinline__ void rot4(double v, double (&r)[4])
{
    r[0] = v;
    r[1] = dpp_row<0x120 + 12>(v);
    r[2] = dpp_row<0x120 + 8>(v);
    r[3] = dpp_row<0x120 + 4>(v);
}
__device__ __forceinline__ double mfma4(double a, double b, double c)
{
    return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}

// One operand tile: global -> registers -> LDS.  X = the non-contracted index (m or n).
//
// Fast path (interior K-tiles, offsets below 2 GB): every pair of the tile is one
// `global_load_dwordx4 v, v_off, s[base]` -- a wave-uniform base pointer that the scalar unit
// advances per K-tile plus a per-thread 32-bit byte offset computed once in init().  No vector
// ALU work per load, which matters because the fp64 MFMA rate is so low on this part that a
// few VALU instructions per MFMA already set the pace (measured: 4.3 VALU per MFMA cost 60 %).
// Row / column overhang is handled by clamping the offset in init() and zeroing at the LDS
// store.  Slow path (K tail, tiles that straddle two `ko` slices, > 2 GB strides): branch-free
// clamped loads with full index arithmetic.
template <bool KF, int NE2>
struct Stager {
    double2 r[NE2];
    uint32_t goff[NE2];  // byte offset of the pair from the tile base (x clamped), K-tile relative
    int lds[NE2];        // LDS offset [15:0] | k [21:16] | pair-element-1 invalid [28] | element-0 invalid [29];
                         // negative: pair not in the tile
    __device__ __forceinline__ void init(int bx, int ld, int tid, int64_t x0, int64_t X, int64_t xs,
                                         int64_t s_ki)
    {
        const int half = KF ? BK / 2 : bx / 2;
#pragma unroll
        for (int e = 0; e < NE2; ++e) {
            const int idx = tid + 256 * e;
            int x, k;
            if (KF) { k = 2 * (idx % (BK / 2)); x = idx / (BK / 2); }
            else    { x = 2 * (idx % half); k = idx / half; }
            if (x < bx && k < BK) {
                const int64_t gx = x0 + x;
                const int x1 = KF ? x : x + 1;
                const bool in0 = gx < X, in1 = x0 + x1 < X;
                // clamp so that even a masked-out pair reads valid memory
                int64_t cx = gx;
                if (gx >= X) {
                    cx = KF ? X - 1 : X - 2;      // a whole pair that exists (16-byte loads need X even)
                    if (cx < x0) cx = x0;
                }
                goff[e] = (uint32_t)(((cx - x0) * xs + (int64_t)k * s_ki) * 8);
                lds[e] = (KF ? x * LDKF + k : k * ld + x) | (k << 16) | (in0 ? 0 : 0x20000000) | (in1 ? 0 : 0x10000000);
            } else {
                goff[e] = 0;
                lds[e] = -1;
            }
        }
    }
    // Fast path: base = address of tile element (x0, k0).  second_off: byte distance between the two
    // elements of a pair when 16-byte loads are not possible.
    __device__ __forceinline__ void load_fast(const char *__restrict__ base, bool vec, uint32_t second_off)
    {
        if (vec) {
#pragma unroll
            for (int e = 0; e < NE2; ++e) r[e] = *reinterpret_cast<const double2 *>(base + goff[e]);
        } else {
#pragma unroll
            for (int e = 0; e < NE2; ++e) {
                r[e].x = *reinterpret_cast<const double *>(base + goff[e]);
                r[e].y = *reinterpret_cast<const double *>(base + goff[e] + ((lds[e] & 0x10000000) ? 0u : second_off));
            }
        }
    }
    // K tail (kcount < BK valid k) and tiles that straddle two `ko` slices (ki0 + k wraps at Ki):
    // same loads with a per-element K adjustment; scalar 8-byte loads, the pair may split.
    __device__ __forceinline__ void load_adj(const char *__restrict__ base, uint32_t second_off, int kcount,
                                             int64_t ki0, int64_t Ki, int64_t s_ki, int64_t s_ko)
    {
        const int64_t wrap = (s_ko - Ki * s_ki) * 8;
#pragma unroll
        for (int e = 0; e < NE2; ++e) {
            const int k = (lds[e] >> 16) & 63;
            const int k1 = KF ? k + 1 : k;
            const int c0 = k < kcount ? k : kcount - 1, c1 = k1 < kcount ? k1 : kcount - 1;
            // goff holds x-part + k*s_ki*8; re-base the k part on the clamped / wrapped index
            const int64_t o0 = (int64_t)goff[e] + (int64_t)(c0 - k) * s_ki * 8 + ((ki0 + c0 >= Ki) ? wrap : 0);
            int64_t o1 = (int64_t)goff[e] + (int64_t)(c1 - k) * s_ki * 8 + ((ki0 + c1 >= Ki) ? wrap : 0);
            if (!KF) o1 += (lds[e] & 0x10000000) ? 0 : second_off;
            r[e].x = *reinterpret_cast<const double *>(base + o0);
            r[e].y = *reinterpret_cast<const double *>(base + o1);
        }
    }
    __device__ __forceinline__ void store(double *S, bool edge, int kcount) const
    {
#pragma unroll
        for (int e = 0; e < NE2; ++e) {
            if (lds[e] >= 0) {
                double2 v = r[e];
                if (edge) {
                    const int k = (lds[e] >> 16) & 63;
                    if ((lds[e] & 0x20000000) || k >= kcount) v.x = 0.0;
                    if ((lds[e] & 0x10000000) || (KF ? k + 1 : k) >= kcount) v.y = 0.0;
                }
                *reinterpret_cast<double2 *>(S + (lds[e] & 0xFFFF)) = v;
            }
        }
    }
};

// Slow path of the staging (K tail, tiles that straddle two `ko` slices, strides beyond the 32-bit
// offset range, k_scale): a rolled loop straight from global memory into the LDS tile with full
// index arithmetic.  Deliberately NOT unrolled: the hot loop has to stay inside the instruction
// cache (an earlier fully unrolled variant was 13k instructions and ran 3x slower).
template <bool KF>
__device__ __noinline__ void slow_fill(double *S, const double *__restrict__ P, int64_t xs, KMap km, int64_t x0,
                                       int64_t X, int64_t k0, int64_t kend, int bx, int ld,
                                       const double *__restrict__ kscale, int tid)
{
#pragma unroll 4
    for (int idx = tid; idx < bx * BK; idx += 256) {
        int x, k;
        if (KF) { k = idx % BK; x = idx / BK; }
        else    { x = idx % bx; k = idx / bx; }
        const int64_t gx = x0 + x, gk = k0 + k;
        double v = 0.0;
        if (gx < X && gk < kend) {
            v = P[gx * xs + km.off(gk)];
            if (kscale) v *= kscale[gk];
        }
        S[KF ? x * LDKF + k : k * ld + x] = v;
    }
}

// WM x WN waves, each owning TMX x TNX MFMA tiles of 16x16 (exact, compile time).
template <int WM, int WN, int TMX, int TNX, bool AKF, bool BKF>
__global__ __launch_bounds__(256) void gemm_f64_kernel(ttsk_gemm_desc d, const double *__restrict__ A,
                                                       const double *__restrict__ B, double *__restrict__ C,
                                                       const double *__restrict__ kscale, int splits,
                                                       int64_t kchunk, double *__restrict__ partial,
                                                       int avec, int bvec, int fast_ok)
{
    static_assert(WM * WN == 4, "256 threads");
    constexpr int tm = TMX, tn = TNX;
    constexpr bool ROTA = WM > WN;    // skinny-N family: one A tile per wave, rotate it
    constexpr int NEA = WM * TMX, NEB = WN * TNX;  // double2 per thread: 16 t W * 32 / 256 / 2
    extern __shared__ double smem[];
    constexpr int bm = WM * tm * 16, bn = WN * tn * 16;
    constexpr int lda = ldmf(bm), ldb = ldmf(bn);
    double *As = smem;
    double *Bs = smem + (AKF ? bm * LDKF : BK * lda);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WN, wc = wave % WN;
    const int64_t m0 = (int64_t)blockIdx.y * bm, n0 = (int64_t)blockIdx.x * bn;
    const int64_t bz = blockIdx.z;
    const int64_t b = bz / splits;
    const int z = (int)(bz - b * splits);
    const int64_t Ktot = d.Ko * d.Ki;
    const int64_t kbeg = (int64_t)z * kchunk;
    const int64_t kend = (kbeg + kchunk < Ktot) ? kbeg + kchunk : Ktot;
    const double *Ab = A + b * d.a_b;
    const double *Bb = B + b * d.b_b;
    const KMap ka{d.Ko, d.Ki, d.a_ko, d.a_ki};
    const KMap kb{d.Ko, d.Ki, d.b_ko, d.b_ki};
    const bool edge_a = m0 + bm > d.M, edge_b = n0 + bn > d.N;

    double acc[TMX][TNX][4];
#pragma unroll
    for (int i = 0; i < TMX; ++i)
#pragma unroll
        for (int j = 0; j < TNX; ++j)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[i][j][t] = 0.0;

    Stager<AKF, NEA> sa;
    Stager<BKF, NEB> sb;
    sa.init(bm, lda, tid, m0, d.M, d.a_m, d.a_ki);
    sb.init(bn, ldb, tid, n0, d.N, d.b_n, d.b_ki);
    const uint32_t a2 = (uint32_t)((AKF ? d.a_ki : d.a_m) * 8), b2 = (uint32_t)((BKF ? d.b_ki : d.b_n) * 8);
    const char *a_base = reinterpret_cast<const char *>(Ab + m0 * d.a_m);
    const char *b_base = reinterpret_cast<const char *>(Bb + n0 * d.b_n);

    // Staging modes per K-tile: 2 = pure fast path, 1 = fast path with K adjustment (tail /
    // ko-straddle), 0 = rolled generic fill (k_scale, tiny Ki, offsets beyond 32 bits).
    auto mode_of = [&](int64_t k0) -> int {
        if (!fast_ok || kscale) return 0;
        const bool full = k0 + BK <= kend;
        if (d.Ko == 1) return full ? 2 : 1;
        if (d.Ki < BK) return 0;
        const int64_t ki0 = k0 % d.Ki;
        return (full && ki0 + BK <= d.Ki) ? 2 : 1;
    };
    auto prefetch = [&](int64_t k0, int mode) {
        int64_t ko0 = 0, ki0 = k0;
        if (d.Ko != 1) { ko0 = k0 / d.Ki; ki0 = k0 - ko0 * d.Ki; }
        const char *pa = a_base + (ko0 * d.a_ko + ki0 * d.a_ki) * 8;
        const char *pb = b_base + (ko0 * d.b_ko + ki0 * d.b_ki) * 8;
        if (mode == 2) {
            sa.load_fast(pa, avec, a2);
            sb.load_fast(pb, bvec, b2);
        } else {
            const int kc = (int)((kend - k0 < BK) ? kend - k0 : BK);
            const int64_t Ki = d.Ko == 1 ? (int64_t)1 << 60 : d.Ki;
            sa.load_adj(pa, a2, kc, ki0, Ki, d.a_ki, d.a_ko);
            sb.load_adj(pb, b2, kc, ki0, Ki, d.b_ki, d.b_ko);
        }
    };

    const int fi = lane >> 4, fj = lane & 15;
    int cur_mode = kbeg < kend ? mode_of(kbeg) : 0;
    if (cur_mode) prefetch(kbeg, cur_mode);
    for (int64_t k0 = kbeg; k0 < kend; k0 += BK) {
        const int kcount = (int)((kend - k0 < BK) ? kend - k0 : BK);
        __syncthreads();
        if (cur_mode) {
            sa.store(As, edge_a || cur_mode == 1, kcount);
            sb.store(Bs, edge_b || cur_mode == 1, kcount);
        } else {
            slow_fill<AKF>(As, Ab, d.a_m, ka, m0, d.M, k0, kend, bm, lda, kscale, tid);
            slow_fill<BKF>(Bs, Bb, d.b_n, kb, n0, d.N, k0, kend, bn, ldb, nullptr, tid);
        }
        __syncthreads();
        cur_mode = k0 + BK < kend ? mode_of(k0 + BK) : 0;
        if (cur_mode) prefetch(k0 + BK, cur_mode);
#pragma unroll
        for (int ks = 0; ks < BK; ks += 4) {
            if (ks >= kcount) break;
            double af[TMX], bf[TNX];
#pragma unroll
            for (int i = 0; i < TMX; ++i) {
                const int x = (wr * tm + i) * 16 + fj;
                af[i] = As[AKF ? x * LDKF + ks + fi : (ks + fi) * lda + x];
            }
#pragma unroll
            for (int j = 0; j < TNX; ++j) {
                const int x = (wc * tn + j) * 16 + fj;
                bf[j] = Bs[BKF ? x * LDKF + ks + fi : (ks + fi) * ldb + x];
            }
            if (ROTA) {
#pragma unroll
                for (int i = 0; i < TMX; ++i) {
                    double ar[4];
                    rot4(af[i], ar);
#pragma unroll
                    for (int j = 0; j < TNX; ++j)
#pragma unroll
                        for (int t = 0; t < 4; ++t) acc[i][j][t] = mfma4(ar[t], bf[j], acc[i][j][t]);
                }
            } else {
#pragma unroll
                for (int j = 0; j < TNX; ++j) {
                    double br[4];
                    rot4(bf[j], br);
#pragma unroll
                    for (int i = 0; i < TMX; ++i)
#pragma unroll
                        for (int t = 0; t < 4; ++t) acc[i][j][t] = mfma4(af[i], br[t], acc[i][j][t]);
                }
            }
        }
    }

    if (partial) {
        // split-K: every wave dumps its accumulators as they sit in registers (64 contiguous
        // doubles per store instruction); splitk_reduce_kernel undoes the lane permutation.
        const int64_t tiles_m = (int64_t)gridDim.y * WM * tm, tiles_n = (int64_t)gridDim.x * WN * tn;
        double *pz = partial + ((int64_t)bz * tiles_m * tiles_n) * 256;
#pragma unroll
        for (int i = 0; i < TMX; ++i)
#pragma unroll
            for (int j = 0; j < TNX; ++j) {
                const int64_t ti = (int64_t)blockIdx.y * WM * tm + wr * tm + i;
                const int64_t tj = (int64_t)blockIdx.x * WN * tn + wc * tn + j;
                double *pt = pz + (ti * tiles_n + tj) * 256;
#pragma unroll
                for (int t = 0; t < 4; ++t) pt[t * 64 + lane] = acc[i][j][t];
            }
        return;
    }
    const int li = lane >> 4, beta = (lane >> 2) & 3, jj = lane & 3;
    // element (row, col) of accumulator t of tile (i, j): see the layout note at the top
    double *Cw = C + b * d.c_b + (m0 + wr * tm * 16 + li) * d.c_m + (n0 + wc * tn * 16 + jj) * d.c_n;
    const int64_t mrem = d.M - (m0 + wr * tm * 16 + li), nrem = d.N - (n0 + wc * tn * 16 + jj);
#pragma unroll
    for (int i = 0; i < TMX; ++i)
#pragma unroll
        for (int j = 0; j < TNX; ++j)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int rb = ROTA ? ((beta + t) & 3) : beta, cb = ROTA ? beta : ((beta + t) & 3);
                const int dm = i * 16 + 4 * rb, dn = j * 16 + 4 * cb;
                if (dm < mrem && dn < nrem) {
                    double v = d.alpha * acc[i][j][t];
                    double *p = Cw + dm * d.c_m + dn * d.c_n;
                    if (d.accumulate) *p += v;
                    else *p = v;
                }
            }
}

// launcher generated per staging-layout pair in gemm_inst_*.hip
struct GemmLaunch {
    ttsk_gemm_desc d;
    const double *A, *B, *ks;
    double *C, *partial;
    int family, tiles;      // family 0: 2x2 waves of 2x2 tiles; 1: 1x4 waves of tiles x 1; 2: 4x1 waves of 1 x tiles
    int splits, avec, bvec, fast_ok;
    int64_t kchunk;
    int bm, bn;
};
template <bool AKF, bool BKF> int launch_gemm_layout(const GemmLaunch &g, hipStream_t st);

template <int WM, int WN, int TMX, int TNX, bool AKF, bool BKF>
static int launch_one(const GemmLaunch &g, hipStream_t st)
{
    constexpr int bm = WM * TMX * 16, bn = WN * TNX * 16;
    dim3 grid((unsigned)cdiv(g.d.N, bn), (unsigned)cdiv(g.d.M, bm), (unsigned)(g.d.batch * g.splits));
    const size_t lds = 8 * (size_t)((AKF ? bm * LDKF : BK * ldmf(bm)) + (BKF ? bn * LDKF : BK * ldmf(bn)));
    hipLaunchKernelGGL((gemm_f64_kernel<WM, WN, TMX, TNX, AKF, BKF>), grid, dim3(256), lds, st, g.d, g.A, g.B, g.C,
                       g.ks, g.splits, g.kchunk, g.partial, g.avec, g.bvec, g.fast_ok);
    TTSK_LAUNCH_CHECK();
    return TTSK_OK;
}

template <bool AKF, bool BKF>
int launch_gemm_layout(const GemmLaunch &g, hipStream_t st)
{
    if (g.family == 0) return launch_one<2, 2, 2, 2, AKF, BKF>(g, st);
    if (g.family == 1) {
        switch (g.tiles) {
        case 1: return launch_one<1, 4, 1, 1, AKF, BKF>(g, st);
        case 2: return launch_one<1, 4, 2, 1, AKF, BKF>(g, st);
        case 3: return launch_one<1, 4, 3, 1, AKF, BKF>(g, st);
        case 4: return launch_one<1, 4, 4, 1, AKF, BKF>(g, st);
        case 5: return launch_one<1, 4, 5, 1, AKF, BKF>(g, st);
        case 6: return launch_one<1, 4, 6, 1, AKF, BKF>(g, st);
        case 7: return launch_one<1, 4, 7, 1, AKF, BKF>(g, st);
        default: return launch_one<1, 4, 8, 1, AKF, BKF>(g, st);
        }
    }
    switch (g.tiles) {
    case 1: return launch_one<4, 1, 1, 1, AKF, BKF>(g, st);
    case 2: return launch_one<4, 1, 1, 2, AKF, BKF>(g, st);
    case 3: return launch_one<4, 1, 1, 3, AKF, BKF>(g, st);
    case 4: return launch_one<4, 1, 1, 4, AKF, BKF>(g, st);
    case 5: return launch_one<4, 1, 1, 5, AKF, BKF>(g, st);
    case 6: return launch_one<4, 1, 1, 6, AKF, BKF>(g, st);
    case 7: return launch_one<4, 1, 1, 7, AKF, BKF>(g, st);
    default: return launch_one<4, 1, 1, 8, AKF, BKF>(g, st);
    }
}

}  // namespace ttsk

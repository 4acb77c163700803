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
// Loads are branch free: indices are clamped into the valid range and the value is zeroed by a
// select afterwards, so the compiler can issue all of a tile's loads back to back behind a
// single wait (a per-element `if` makes hipcc wait for every load separately).  All divisions
// happen once in init(); per K-tile the (ko, ki) split of the tile origin is wave-uniform.
template <bool KF, int NE2>
struct Stager {
    double2 r[NE2];
    int xk[NE2];  // x | k << 16 of the first element of pair e, -1 if the pair is outside the tile
    __device__ __forceinline__ void init(int bx, int tid)
    {
        const int half = KF ? BK / 2 : bx / 2;
#pragma unroll
        for (int e = 0; e < NE2; ++e) {
            const int idx = tid + 256 * e;
            int x, k;
            if (KF) { k = 2 * (idx % (BK / 2)); x = idx / (BK / 2); }
            else    { x = 2 * (idx % half); k = idx / half; }
            xk[e] = (x < bx && k < BK) ? (x | (k << 16)) : -1;
        }
    }
    __device__ __forceinline__ int64_t koff(const KMap &km, int64_t ko0, int64_t ki0, int k) const
    {
        if (km.Ko == 1) return (ki0 + k) * km.s_ki;
        int64_t ki = ki0 + k, ko = ko0;
        if (km.Ki >= BK) {          // at most one wrap inside a tile
            const bool w = ki >= km.Ki;
            ki -= w ? km.Ki : 0;
            ko += w ? 1 : 0;
        } else {
            const int64_t q = ki / km.Ki;
            ko += q;
            ki -= q * km.Ki;
        }
        return ko * km.s_ko + ki * km.s_ki;
    }
    // pair e covers tile elements (x, k),(x, k+1) [KF] or (x, k),(x+1, k) [m-fast]
    __device__ __forceinline__ void load(const double *__restrict__ P, int64_t xs, const KMap &km, int64_t x0,
                                         int64_t X, int64_t k0, int64_t kend, bool vec,
                                         const double *__restrict__ kscale)
    {
        int64_t ko0 = 0, ki0 = k0;
        if (km.Ko != 1) { ko0 = k0 / km.Ki; ki0 = k0 - ko0 * km.Ki; }
        const int64_t xlast = X - 1, klast = kend - 1 - k0;   // clamps (tile-relative for k)
        if (vec) {
#pragma unroll
            for (int e = 0; e < NE2; ++e) {
                const int x = xk[e] & 0xFFFF, k = (xk[e] >> 16) & 0x7FFF;
                // vec is only set when extents are even: a pair is inside or outside as a whole
                int64_t gx = x0 + x;
                int kc = k;
                const bool in = xk[e] >= 0 && gx < X && k <= klast;
                gx = gx < X ? gx : (KF ? xlast : xlast - 1);
                kc = kc <= klast ? kc : (int)(KF ? klast - 1 : klast);
                const double2 v = *reinterpret_cast<const double2 *>(P + gx * xs + koff(km, ko0, ki0, kc));
                r[e].x = in ? v.x : 0.0;
                r[e].y = in ? v.y : 0.0;
            }
        } else {
#pragma unroll
            for (int e = 0; e < NE2; ++e) {
                const int x = xk[e] & 0xFFFF, k = (xk[e] >> 16) & 0x7FFF;
                const int64_t gx = x0 + x, gx1 = KF ? gx : gx + 1;
                const int k1 = KF ? k + 1 : k;
                const bool in0 = xk[e] >= 0 && gx < X && k <= klast;
                const bool in1 = xk[e] >= 0 && gx1 < X && k1 <= klast;
                const int64_t cx0 = gx < X ? gx : xlast, cx1 = gx1 < X ? gx1 : xlast;
                const int ck0 = k <= klast ? k : (int)klast, ck1 = k1 <= klast ? k1 : (int)klast;
                const double v0 = P[cx0 * xs + koff(km, ko0, ki0, ck0)];
                const double v1 = P[cx1 * xs + koff(km, ko0, ki0, ck1)];
                r[e].x = in0 ? v0 : 0.0;
                r[e].y = in1 ? v1 : 0.0;
            }
        }
        if (kscale) {
#pragma unroll
            for (int e = 0; e < NE2; ++e) {
                const int k = (xk[e] >> 16) & 0x7FFF;
                const int k1 = KF ? k + 1 : k;
                const int ck0 = k <= klast ? k : (int)klast, ck1 = k1 <= klast ? k1 : (int)klast;
                r[e].x *= kscale[k0 + ck0];
                r[e].y *= kscale[k0 + ck1];
            }
        }
    }
    __device__ __forceinline__ void store(double *S, int ld) const
    {
#pragma unroll
        for (int e = 0; e < NE2; ++e) {
            if (xk[e] >= 0) {
                const int x = xk[e] & 0xFFFF, k = xk[e] >> 16;
                *reinterpret_cast<double2 *>(S + (KF ? x * LDKF + k : k * ld + x)) = r[e];
            }
        }
    }
};

// WM x WN waves, each owning TMX x TNX MFMA tiles of 16x16 (exact, compile time).
template <int WM, int WN, int TMX, int TNX, bool AKF, bool BKF>
__global__ __launch_bounds__(256) void gemm_f64_kernel(ttsk_gemm_desc d, const double *__restrict__ A,
                                                       const double *__restrict__ B, double *__restrict__ C,
                                                       const double *__restrict__ kscale, int splits,
                                                       int64_t kchunk, double *__restrict__ partial,
                                                       int avec, int bvec)
{
    constexpr int tm = TMX, tn = TNX;
    static_assert(WM * WN == 4, "256 threads");
    constexpr bool ROTA = TMX < TNX;  // rotate the operand with fewer tiles per wave
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

    double acc[TMX][TNX][4];
#pragma unroll
    for (int i = 0; i < TMX; ++i)
#pragma unroll
        for (int j = 0; j < TNX; ++j)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[i][j][t] = 0.0;

    Stager<AKF, NEA> sa;
    Stager<BKF, NEB> sb;
    const int fi = lane >> 4, fj = lane & 15;
    sa.init(bm, tid);
    sb.init(bn, tid);
    if (kbeg < kend) {
        sa.load(Ab, d.a_m, ka, m0, d.M, kbeg, kend, avec, kscale);
        sb.load(Bb, d.b_n, kb, n0, d.N, kbeg, kend, bvec, nullptr);
    }
    for (int64_t k0 = kbeg; k0 < kend; k0 += BK) {
        __syncthreads();
        sa.store(As, lda);
        sb.store(Bs, ldb);
        __syncthreads();
        if (k0 + BK < kend) {
            sa.load(Ab, d.a_m, ka, m0, d.M, k0 + BK, kend, avec, kscale);
            sb.load(Bb, d.b_n, kb, n0, d.N, k0 + BK, kend, bvec, nullptr);
        }
#pragma unroll
        for (int ks = 0; ks < BK; ks += 4) {
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
                    if (i < tm) {
                        double ar[4];
                        rot4(af[i], ar);
#pragma unroll
                        for (int j = 0; j < TNX; ++j)
                            if (j < tn) {
#pragma unroll
                                for (int t = 0; t < 4; ++t) acc[i][j][t] = mfma4(ar[t], bf[j], acc[i][j][t]);
                            }
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < TNX; ++j) {
                    if (j < tn) {
                        double br[4];
                        rot4(bf[j], br);
#pragma unroll
                        for (int i = 0; i < TMX; ++i)
                            if (i < tm) {
#pragma unroll
                                for (int t = 0; t < 4; ++t) acc[i][j][t] = mfma4(af[i], br[t], acc[i][j][t]);
                            }
                    }
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
            for (int j = 0; j < TNX; ++j)
                if (i < tm && j < tn) {
                    const int64_t ti = (int64_t)blockIdx.y * WM * tm + wr * tm + i;
                    const int64_t tj = (int64_t)blockIdx.x * WN * tn + wc * tn + j;
                    double *pt = pz + (ti * tiles_n + tj) * 256;
#pragma unroll
                    for (int t = 0; t < 4; ++t) pt[t * 64 + lane] = acc[i][j][t];
                }
        return;
    }
    double *Cb = C + b * d.c_b;
    const int li = lane >> 4, beta = (lane >> 2) & 3, jj = lane & 3;
#pragma unroll
    for (int i = 0; i < TMX; ++i)
#pragma unroll
        for (int j = 0; j < TNX; ++j)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (i < tm && j < tn) {
                    const int rb = ROTA ? ((beta + t) & 3) : beta, cb = ROTA ? beta : ((beta + t) & 3);
                    const int64_t m = m0 + (wr * tm + i) * 16 + 4 * rb + li;
                    const int64_t n = n0 + (wc * tn + j) * 16 + 4 * cb + jj;
                    if (m < d.M && n < d.N) {
                        double v = d.alpha * acc[i][j][t];
                        double *p = Cb + m * d.c_m + n * d.c_n;
                        if (d.accumulate) *p += v;
                        else *p = v;
                    }
                }
            }
}


// launcher generated per staging-layout pair in gemm_inst_*.hip
struct GemmLaunch {
    ttsk_gemm_desc d;
    const double *A, *B, *ks;
    double *C, *partial;
    int family, tiles;      // family 0: 2x2 waves of 2x2 tiles; 1: 1x4 waves of tiles x 1; 2: 4x1 waves of 1 x tiles
    int splits, avec, bvec;
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
                       g.ks, g.splits, g.kchunk, g.partial, g.avec, g.bvec);
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

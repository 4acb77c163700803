// fp64 MFMA contraction kernel template (see gemm.hip for the host side).
#pragma once
#include <cstdint>
#include <type_traits>
#include "common.h"

namespace ttsk {

// in-kernel cycle stamps for the diagnostic build in scratch/ (no-ops in the library)
#ifdef TTSK_STAMPS
__device__ long long g_stamps[256];
#define TTSK_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) g_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TTSK_STAMP(i) do { } while (0)
#endif

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
// Measured on MI355X (profiles/scripts/mfma_probe2.hip, mfma_mix.hip): both forms reach
// the pipe rate -- 16x16x4 one instruction per 64 cycles and SIMD (77.7 TF/s at 2.4 GHz), 4x4x4 (4 blocks) one per
// 16-17 cycles (75 TF/s), freely mixed.  (Round 1 read 49 vs 65 TF/s off a probe compiled with
// __launch_bounds__(256): there hipcc keeps the accumulators in AGPRs and copies them around every
// instruction.  The same happened to this kernel until it declared two waves per SIMD.)  The 4x4x4 form multiplies, for each of the four lane sub-groups beta
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
    // row_ror reads a valid lane everywhere, so there is no "old" value to preserve (one v_mov_dpp each)
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, false);
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
// Loads are `buffer_load_dwordx4 v, v_off, s[rsrc], s_off offen`: a wave-uniform buffer resource
// for the operand (base rebased to the workgroup's first row/column, num_records = what is left of
// the operand, so anything past its end reads as 0 instead of faulting), a per-thread 32-bit byte
// offset fixed at kernel start and a scalar offset that the SALU advances per K-tile.  No vector
// ALU work and no branches per load; with one wave per SIMD every VALU instruction costs 4-8
// cycles that the fp64 matrix pipe (16 cycles per MFMA) cannot hide.
// Rows / columns beyond M / N read valid-or-zero data and only pollute accumulator rows /
// columns that are never stored.  Only the contracted index needs exact zeros: K-tail tiles
// (and tiles straddling two `ko` slices) go through load_adj + a masked LDS store.
// The 256 threads x NE2 pairs cover the (16 T W) x 32 tile exactly: no per-element validity.
typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef int v2i_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const double *base, int64_t bytes)
{
    const uint32_t n = bytes <= 0 ? 0u : (bytes > 0xFFFFFFFFll ? 0xFFFFFFFFu : (uint32_t)bytes);
    return __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)n, 0x00020000);
}
__device__ __forceinline__ double2 ld16(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff)
{
    v4i_t v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
    return *reinterpret_cast<double2 *>(&v);
}
__device__ __forceinline__ double ld8(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff)
{
    v2i_t v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0);
    return *reinterpret_cast<double *>(&v);
}

template <bool KF, int NE2>
struct Stager {
    uint32_t goff[NE2];  // byte offset of pair e from the tile origin (x0, k-tile start)
    uint32_t lds[NE2];   // LDS byte offset of pair e
    template <int BX>
    __device__ __forceinline__ void init(int ld, int tid, uint32_t xs8, uint32_t ski8)
    {
        constexpr int half = KF ? BK / 2 : BX / 2;
#pragma unroll
        for (int e = 0; e < NE2; ++e) {
            const int idx = tid + 256 * e;
            int x, k;
            if (KF) { k = 2 * (idx % half); x = idx / half; }
            else    { x = 2 * (idx % half); k = idx / half; }
            goff[e] = (uint32_t)x * xs8 + (uint32_t)k * ski8;
            lds[e] = (uint32_t)(KF ? x * LDKF + k : k * ld + x) * 8u;
        }
    }
    template <int BX>
    __device__ __forceinline__ int kk(int e, int tid) const
    {
        constexpr int half = KF ? BK / 2 : BX / 2;
        const int idx = tid + 256 * e;
        return KF ? 2 * (idx % half) : idx / half;
    }
    __device__ __forceinline__ void load_fast(double2 (&r)[NE2], __amdgpu_buffer_rsrc_t rs, uint32_t soff, bool vec,
                                              uint32_t second_off) const
    {
        if (vec) {
#pragma unroll
            for (int e = 0; e < NE2; ++e) r[e] = ld16(rs, goff[e], soff);
        } else {
#pragma unroll
            for (int e = 0; e < NE2; ++e) {
                r[e].x = ld8(rs, goff[e], soff);
                r[e].y = ld8(rs, goff[e] + second_off, soff);
            }
        }
    }
    // tiles that straddle two `ko` slices (ki0 + k wraps at Ki): per-element K adjustment, 8-byte loads
    template <int BX>
    __device__ __forceinline__ void load_adj(double2 (&r)[NE2], __amdgpu_buffer_rsrc_t rs, uint32_t soff,
                                             uint32_t second_off, int64_t ki0, int64_t Ki, int64_t wrap_bytes,
                                             int tid) const
    {
#pragma unroll
        for (int e = 0; e < NE2; ++e) {
            const int k = kk<BX>(e, tid), k1 = KF ? k + 1 : k;
            const uint32_t w0 = (ki0 + k >= Ki) ? (uint32_t)wrap_bytes : 0u;
            const uint32_t w1 = (ki0 + k1 >= Ki) ? (uint32_t)wrap_bytes : 0u;
            r[e].x = ld8(rs, goff[e] + w0, soff);
            r[e].y = ld8(rs, goff[e] + second_off + w1, soff);
        }
    }
    template <int BX>
    __device__ __forceinline__ void store(const double2 (&r)[NE2], char *S, bool mask_k, int kcount, int tid) const
    {
        if (!mask_k) {
#pragma unroll
            for (int e = 0; e < NE2; ++e) *reinterpret_cast<double2 *>(S + lds[e]) = r[e];
        } else {
#pragma unroll
            for (int e = 0; e < NE2; ++e) {
                const int k = kk<BX>(e, tid), k1 = KF ? k + 1 : k;
                double2 v = r[e];
                if (k >= kcount) v.x = 0.0;
                if (k1 >= kcount) v.y = 0.0;
                *reinterpret_cast<double2 *>(S + lds[e]) = v;
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
__global__ __launch_bounds__(256, 2) void gemm_f64_kernel(ttsk_gemm_desc d, const double *__restrict__ A,
                                                       const double *__restrict__ B, double *__restrict__ C,
                                                       const double *__restrict__ kscale, int splits,
                                                       int64_t kchunk, double *__restrict__ partial,
                                                       int avec, int bvec, int fast_ok, int64_t a_extent,
                                                       int64_t b_extent)
{
    static_assert(WM * WN == 4, "256 threads");
    TTSK_STAMP(0);
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

    double acc[TMX][TNX][4];
#pragma unroll
    for (int i = 0; i < TMX; ++i)
#pragma unroll
        for (int j = 0; j < TNX; ++j)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[i][j][t] = 0.0;

    Stager<AKF, NEA> sa;
    Stager<BKF, NEB> sb;
    sa.template init<bm>(lda, tid, (uint32_t)(d.a_m * 8), (uint32_t)(d.a_ki * 8));
    sb.template init<bn>(ldb, tid, (uint32_t)(d.b_n * 8), (uint32_t)(d.b_ki * 8));
    TTSK_STAMP(1);
    const uint32_t a2 = (uint32_t)((AKF ? d.a_ki : d.a_m) * 8), b2 = (uint32_t)((BKF ? d.b_ki : d.b_n) * 8);
    // buffer resources rebased to this workgroup's first row / column; what lies past the operand reads 0
    const __amdgpu_buffer_rsrc_t rsa = make_rsrc(Ab + m0 * d.a_m, (a_extent - m0 * d.a_m) * 8);
    const __amdgpu_buffer_rsrc_t rsb = make_rsrc(Bb + n0 * d.b_n, (b_extent - n0 * d.b_n) * 8);
    char *Asb = reinterpret_cast<char *>(As), *Bsb = reinterpret_cast<char *>(Bs);

    // Staging modes per K-tile: 2 = plain, 1 = K adjustment (tail / ko-straddle), 0 = rolled generic
    // fill (k_scale, tiny Ki, offsets beyond 32 bits).
    auto mode_of = [&](int64_t k0) -> int {
        if (!fast_ok || kscale) return 0;
        const bool full = k0 + BK <= kend;
        if (d.Ko == 1) return full ? 2 : 1;
        if (d.Ki < BK) return 0;
        const int64_t ki0 = k0 % d.Ki;
        return (full && ki0 + BK <= d.Ki) ? 2 : 1;
    };
    // NST K-tiles in flight in registers: with K of order 100 a workgroup only has 3-4 tiles, and
    // issuing their loads back to back (instead of one tile per iteration) removes the
    // per-iteration memory round trip of these short-K products.
    constexpr int NST = 3;
    double2 ra[NST][NEA], rb[NST][NEB];
    auto prefetch = [&](double2 (&xa)[NEA], double2 (&xb)[NEB], int64_t k0, int mode) {
        int64_t ko0 = 0, ki0 = k0;
        if (d.Ko != 1) { ko0 = k0 / d.Ki; ki0 = k0 - ko0 * d.Ki; }
        const uint32_t sa_off = (uint32_t)((ko0 * d.a_ko + ki0 * d.a_ki) * 8);
        const uint32_t sb_off = (uint32_t)((ko0 * d.b_ko + ki0 * d.b_ki) * 8);
        if (mode == 2 || d.Ko == 1) {
            sa.load_fast(xa, rsa, sa_off, avec, a2);
            sb.load_fast(xb, rsb, sb_off, bvec, b2);
        } else {
            sa.template load_adj<bm>(xa, rsa, sa_off, a2, ki0, d.Ki, (d.a_ko - d.Ki * d.a_ki) * 8, tid);
            sb.template load_adj<bn>(xb, rsb, sb_off, b2, ki0, d.Ki, (d.b_ko - d.Ki * d.b_ki) * 8, tid);
        }
    };

    const int fi = lane >> 4, fj = lane & 15;
    // one K-tile: registers (slot P) -> LDS, refill the slot NST tiles ahead, MFMAs
    auto step = [&](auto SLOT, int64_t k0, int &mode) {
        constexpr int P = decltype(SLOT)::value;
        const int kcount = (int)((kend - k0 < BK) ? kend - k0 : BK);
        const int sb_ = 8 + 8 * (int)((k0 - kbeg) / BK);
        TTSK_STAMP(sb_);
        __syncthreads();
        TTSK_STAMP(sb_ + 1);
        if (mode) {
            sa.template store<bm>(ra[P], Asb, mode == 1, kcount, tid);
            sb.template store<bn>(rb[P], Bsb, mode == 1, kcount, tid);
        } else {
            slow_fill<AKF>(As, Ab, d.a_m, ka, m0, d.M, k0, kend, bm, lda, kscale, tid);
            slow_fill<BKF>(Bs, Bb, d.b_n, kb, n0, d.N, k0, kend, bn, ldb, nullptr, tid);
        }
        TTSK_STAMP(sb_ + 2);
        __syncthreads();
        TTSK_STAMP(sb_ + 3);
        mode = k0 + NST * BK < kend ? mode_of(k0 + NST * BK) : 0;
        if (mode) prefetch(ra[P], rb[P], k0 + NST * BK, mode);
        TTSK_STAMP(sb_ + 4);
        // software-pipelined k-steps: the LDS fragment reads (and the DPP rotation) of step ks+4
        // are issued before the MFMAs of step ks, so one wave per SIMD keeps its matrix pipe busy
        auto frag = [&](int ks, double (&af)[TMX], double (&bf)[TNX]) {
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
        };
        constexpr int NR = ROTA ? TMX : TNX;          // tiles of the rotated operand
        auto rotate = [&](const double (&af)[TMX], const double (&bf)[TNX], double (&rr)[NR][4]) {
#pragma unroll
            for (int q = 0; q < NR; ++q) rot4(ROTA ? af[q] : bf[q], rr[q]);
        };
        auto mma = [&](const double (&af)[TMX], const double (&bf)[TNX], const double (&rr)[NR][4]) {
#pragma unroll
            for (int i = 0; i < TMX; ++i)
#pragma unroll
                for (int j = 0; j < TNX; ++j)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        acc[i][j][t] = ROTA ? mfma4(rr[i][t], bf[j], acc[i][j][t])
                                            : mfma4(af[i], rr[j][t], acc[i][j][t]);
        };
        double af0[TMX], bf0[TNX], af1[TMX], bf1[TNX], r0[NR][4], r1[NR][4];
        frag(0, af0, bf0);
        rotate(af0, bf0, r0);
#pragma unroll
        for (int ks = 0; ks < BK; ks += 8) {
            if (ks >= kcount) break;
            const bool more1 = ks + 4 < kcount, more2 = ks + 8 < kcount;
            if (more1) frag(ks + 4, af1, bf1);
            mma(af0, bf0, r0);
            if (!more1) break;
            rotate(af1, bf1, r1);
            if (more2 && ks + 8 < BK) frag(ks + 8, af0, bf0);
            mma(af1, bf1, r1);
            if (more2 && ks + 8 < BK) rotate(af0, bf0, r0);
        }
        TTSK_STAMP(sb_ + 5);
    };
    int modes[NST];
#pragma unroll
    for (int q = 0; q < NST; ++q) modes[q] = kbeg + q * BK < kend ? mode_of(kbeg + q * BK) : 0;
    if (modes[0]) prefetch(ra[0], rb[0], kbeg, modes[0]);
    if (modes[1]) prefetch(ra[1], rb[1], kbeg + BK, modes[1]);
    if (modes[2]) prefetch(ra[2], rb[2], kbeg + 2 * BK, modes[2]);
    TTSK_STAMP(2);
    for (int64_t k0 = kbeg; k0 < kend; k0 += NST * BK) {
        step(std::integral_constant<int, 0>{}, k0, modes[0]);
        if (k0 + BK < kend) step(std::integral_constant<int, 1>{}, k0 + BK, modes[1]);
        if (k0 + 2 * BK < kend) step(std::integral_constant<int, 2>{}, k0 + 2 * BK, modes[2]);
    }

    TTSK_STAMP(3);
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
    int64_t a_extent, b_extent;   // elements from the (per-batch) operand base to its last element + 1
};
template <bool AKF, bool BKF> int launch_gemm_layout(const GemmLaunch &g, hipStream_t st);

template <int WM, int WN, int TMX, int TNX, bool AKF, bool BKF>
static int launch_one(const GemmLaunch &g, hipStream_t st)
{
    constexpr int bm = WM * TMX * 16, bn = WN * TNX * 16;
    dim3 grid((unsigned)cdiv(g.d.N, bn), (unsigned)cdiv(g.d.M, bm), (unsigned)(g.d.batch * g.splits));
    const size_t lds = 8 * (size_t)((AKF ? bm * LDKF : BK * ldmf(bm)) + (BKF ? bn * LDKF : BK * ldmf(bn)));
    hipLaunchKernelGGL((gemm_f64_kernel<WM, WN, TMX, TNX, AKF, BKF>), grid, dim3(256), lds, st, g.d, g.A, g.B, g.C,
                       g.ks, g.splits, g.kchunk, g.partial, g.avec, g.bvec, g.fast_ok, g.a_extent, g.b_extent);
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

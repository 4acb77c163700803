// The pass over a dense tensor that a sketch with tensor-train DRMs starts with (reference dense_sketch.py:7-52 with the
// matrices of tensor_train_drm.py:109-122): BOTH products that read the tensor, from ONE read of it.
//
//   X[b][q][t]    the tensor, b = first mode (NB = 32 or 64), t = last mode (T), q = everything in between (Q), C order
//   Z[p'][q][t] = sum_b C[b][p'] X[b][q][t]        the first left product, C the left DRM's first core (NB x ll), ll <= 20
//   U[b][p][t]  = sum_q P[q][p]  X[b][q][t]        the long-K half of Psi_0, P the right DRM's matrix of one core less (Q x r),
//                                                   r <= 40 (Psi_0 = U x last core follows as a small product)
//
// Z needs every b of a column in one workgroup, U sums over q: a workgroup owns 16 values of t (128-byte pieces of the tensor's
// rows: whole cache lines) and a range of q, for all b.  Its eight waves hold U for NB / 8 values of b each -- (2 tiles + 2
// four-wide strips) x NB / 8 accumulators of v_mfma_f64_16x16x4 / 4x4x4, 160 registers at NB = 64, which is what limits the
// t range to 16 -- and walk q in steps of 4: a (NB x 4 x 16) tile of X (32 KB) goes to LDS by global_load_lds into a ring of
// four images (three tiles travel while one is computed on; the barrier per tile carries no fence, so they keep travelling
// across it), and is read twice from there: once with q as the K index of the matrix instruction (U: lanes (t, q)), once with
// b as the K index (Z: lanes (t, b); the tile's four rows go to waves 0..3 on even tiles and to waves 4..7 on odd ones, each
// chaining over all b).  Row q of b's chunk sits at row q ^ (b & 1): both fragment shapes then read 32 consecutive doubles
// per half wave (no bank conflicts: SQ_LDS_BANK_CONFLICT = 0).  The tensor is read once (8.59 GB at C2 instead of 17.2), Z
// is written once, the partial U of the q ranges are summed by a second launch.
//
// Work per tile at NB = 64: U 64 x (2 x 64 + 2 x 16) + Z 4 x 16 x (64 + 16) = 15360 cycles of the matrix pipes per 32 KB
// = 129 GF per C2 sketch (no padded rows: 20 = 16 + 4, 40 = 32 + 2 x 4), 1.64 ms at the fp64 peak (SQ_VALU_MFMA_BUSY_CYCLES
// = 4.03e9 over 1024 SIMDs agrees); the HBM side is 8.59 + 2.68 GB = 2.1 ms at the 5.5 TB/s a mixed stream reaches.
// Measured: 2.9 - 3.1 ms = 0.53 - 0.57 of the matrix peak, 3.6 - 3.9 TB/s; without its loads and stores
// the same instruction stream takes 2.0 - 2.3 ms.  What the memory side loses: 64 + 20 rows 128 MB apart per workgroup -- one
// translation per row and instruction (TCP_UTCL1_TRANSLATION_MISS = 22 M per launch, about two per kilobyte moved).
#include "common.h"
#include <type_traits>

namespace ttsk {

namespace {

__device__ __forceinline__ int64_t uniform_i64(int64_t v)
{
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

__device__ __forceinline__ double mfma4s(double a, double b, double c)
{
    return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}

struct DensePass {
    const double *X;
    int64_t sb;            // elements between consecutive b
    int Q, T;
    const double *C;
    int ll;
    const double *P;
    int r;
    double *Z;
    double *slab;          // [workgroup][NB][10][64]: the accumulators as the lanes hold them
    int nt, nqc;           // t ranges (T / 16), q chunks (a multiple of 8)
    int dbg;               // diagnostics (TTSK_DP_DBG): 1 = no tile loads after the first, 2 = no Z stores, 4 = no rotated start
};

constexpr int DP_PROW = 48;        // doubles per row of the P image (40 used; 48 = 96 dwords: rows kq, kq + 1 on disjoint banks)
constexpr int DP_PBUF = 256;       // doubles per P image (4 rows of 48, loaded as two 1 KB instructions)
constexpr int DP_RING = 4;         // tile images: three tiles travel while one is computed on

template <int NBW>
__global__ __launch_bounds__(512) void dense_pass_kernel(DensePass a)
{
    constexpr int NB = 8 * NBW;
    extern __shared__ double lds[];
    double *Xl = lds;                              // [DP_RING][NB][4][16]
    double *Pl = Xl + DP_RING * NB * 64;           // [DP_RING][DP_PBUF]
    double *Cf = Pl + DP_RING * DP_PBUF;           // [NB][16]   columns 0..15 of C
    double *Cs = Cf + NB * 16;                     // [NB][4]    columns 16..19

    const int tid = threadIdx.x, lane = tid & 63, x16 = lane & 15, kq = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int id = blockIdx.x;
    // the nt workgroups that share the 512-byte rows of a q range sit on one XCD (ids 8 apart) next to each other in time
    const int tr = (id >> 3) % a.nt, qc = (id & 7) + 8 * (id / (8 * a.nt));
    const int t0 = 16 * tr;
    const int total = a.Q >> 2;
    const int it_beg = (int)((int64_t)qc * total / a.nqc), it_end = (int)((int64_t)(qc + 1) * total / a.nqc);

    for (int e = tid; e < NB * 20; e += 512) {
        const int b = e / 20, c = e - 20 * b;
        const double v = c < a.ll ? a.C[(int64_t)b * a.ll + c] : 0.0;
        if (c < 16) Cf[b * 16 + c] = v; else Cs[b * 4 + c - 16] = v;
    }

    // tile `it` (4 values of q) -> image `buf`: wave w brings the chunks of its own NBW values of b, two per 1 KB instruction
    // (lanes 0..31 / 32..63), row q of b's chunk to row q ^ (b & 1); waves 0 and 1 the 4 rows of P (24 16-byte units per row,
    // the last 4 of them and the second instruction's upper half padding: whatever they read is never used).  Addresses as
    // a wave-uniform 64-bit base plus a 32-bit lane offset (the saddr form: one register per lane, not two per chunk).
    const int bsel = lane >> 5, xrow = (lane >> 3) & 3, xcol = 2 * (lane & 7);
    const uint32_t xoff = (uint32_t)(((int64_t)bsel * a.sb + (xrow ^ bsel) * a.T + xcol) * 8);
    const int pU = (64 * w + lane) % 96, prow = pU / 24, ppair = pU - 24 * prow;
    const uint32_t poff = (uint32_t)((prow * a.r + (2 * ppair + 2 <= a.r ? 2 * ppair : 0)) * 8);
    const char *xbase = (const char *)a.X + ((int64_t)w * NBW * a.sb + t0) * 8;
    auto issue = [&](int it, int buf) {
        const int64_t tile = uniform_i64((int64_t)it * 4 * a.T * 8);
#pragma unroll
        for (int u = 0; u < NBW / 2; ++u) {
            const char *src = xbase + uniform_i64(tile + (int64_t)2 * u * a.sb * 8) + xoff;
            // (aux = 2: non-temporal -- every byte of the tensor is used exactly once; 3 % on the whole kernel)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(Xl + (buf * NB + w * NBW + 2 * u) * 64), 16, 0, 2);
        }
        if (w < 2) {
            const char *src = (const char *)a.P + uniform_i64((int64_t)it * 4 * a.r * 8) + poff;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(Pl + buf * DP_PBUF + w * 128), 16, 0, 0);
        }
    };

    v4d acc[NBW][2];
    double accs[NBW][2];
#pragma unroll
    for (int bi = 0; bi < NBW; ++bi)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            accs[bi][p] = 0.0;
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[bi][p][j] = 0.0;
        }

    // Z: the four rows of a tile belong to waves 0..3 on even tiles, to waves 4..7 on odd ones (waves w and w + 4 share a SIMD:
    // its matrix pipe sees the same work every tile).  A wave stores its row one tile later, when it has no row to compute.
    const int rw = w & 3, zrole = w >> 2;
    v4d z;
    double zs = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) z[j] = 0.0;
    const int64_t zs_row = (int64_t)a.Q * a.T;
    const uint32_t zoff = (uint32_t)((kq * zs_row + x16) * 8);          // 3 Q T doubles at most: the host checks the range
    auto store_z = [&](int it) {
        const int64_t at = uniform_i64((((int64_t)it * 4 + rw) * a.T + t0) * 8);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (kq + 4 * j < a.ll) *(double *)((char *)a.Z + uniform_i64(at + 4 * j * zs_row * 8) + zoff) = z[j];
        if (16 + kq < a.ll) *(double *)((char *)a.Z + uniform_i64(at + 16 * zs_row * 8) + zoff) = zs;
    };

    // NBW steps per tile; step s: U of b = w NBW + s (q the K index: 2 tiles + 2 strips of p) and, on the wave's own tiles, links
    // 2 s, 2 s + 1 of the Z chain (b the K index, 4 per link).  The fragments of step s + 1 are read while the matrix
    // instructions of step s run; the scheduling barrier per step keeps that order -- left alone, the compiler hoists every
    // read of the tile to the front and spills the accumulators.
    auto compute = [&](int buf, const bool ZA) {
        const double *xb = Xl + buf * NB * 64, *pb = Pl + buf * DP_PBUF;
        const double *xu = xb + w * NBW * 64 + x16;
        const double *xz = xb + kq * 64 + (rw ^ (kq & 1)) * 16 + x16;
        const double *cf = Cf + kq * 16 + x16, *cs = Cs + kq * 4 + (x16 & 3);
        const double *pr = pb + kq * DP_PROW + x16, *prs = pb + kq * DP_PROW + 32 + (x16 & 3);
        const double pf0 = pr[0], pf1 = pr[16], ps0 = prs[0], ps1 = prs[4];
        double xf = xu[kq * 16], xz0 = 0.0, xz1 = 0.0, cf0 = 0.0, cf1 = 0.0, cs0 = 0.0, cs1 = 0.0;
        if (ZA) {                              // (ZA is wave-uniform: scalar branches around the Z instructions only, so
            xz0 = xz[0]; xz1 = xz[256];        //  that the 160 accumulator registers of U never meet at a join)
            cf0 = cf[0]; cf1 = cf[64];
            cs0 = cs[0]; cs1 = cs[16];
#pragma unroll
            for (int j = 0; j < 4; ++j) z[j] = 0.0;
            zs = 0.0;
        }
#pragma unroll
        for (int s = 0; s < NBW; ++s) {
            double xf_n = 0.0;
            if (s + 1 < NBW) xf_n = xu[(s + 1) * 64 + (kq ^ ((s + 1) & 1)) * 16];
            acc[s][0] = mfma16(xf, pf0, acc[s][0]);
            acc[s][1] = mfma16(xf, pf1, acc[s][1]);
            accs[s][0] = mfma4s(xf, ps0, accs[s][0]);
            accs[s][1] = mfma4s(xf, ps1, accs[s][1]);
            xf = xf_n;
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (ZA) {
                double xz0_n = 0.0, xz1_n = 0.0, cf0_n = 0.0, cf1_n = 0.0, cs0_n = 0.0, cs1_n = 0.0;
                if (s + 1 < NBW) {
                    xz0_n = xz[(2 * s + 2) * 256]; xz1_n = xz[(2 * s + 3) * 256];
                    cf0_n = cf[(2 * s + 2) * 64]; cf1_n = cf[(2 * s + 3) * 64];
                    cs0_n = cs[(2 * s + 2) * 16]; cs1_n = cs[(2 * s + 3) * 16];
                }
                z = mfma16(cf0, xz0, z);
                zs = mfma4s(cs0, xz0, zs);
                z = mfma16(cf1, xz1, z);
                zs = mfma4s(cs1, xz1, zs);
                xz0 = xz0_n; xz1 = xz1_n; cf0 = cf0_n; cf1 = cf1_n; cs0 = cs0_n; cs1 = cs1_n;
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    // The range is walked from a start that differs from workgroup to workgroup (the same for the nt workgroups that share
    // rows): with power-of-two extents every b row and every q range begins at the same offset modulo 2 MB, and workgroups
    // marching in step would all ask the same few memory channels at any moment.
    const int len = it_end - it_beg;
    const int rot = len > 0 && !(a.dbg & 4) ? (int)(((uint32_t)qc * 2654435761u >> 8) % (uint32_t)len) : 0;
    auto tile_of = [&](int rel) { const int t = rel + rot; return it_beg + (t >= len ? t - len : t); };
#pragma unroll 1
    for (int k = 0; k < DP_RING - 1; ++k)
        if (k < len) issue(tile_of(k), k);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int pend = -1;                                  // the tile whose Z row this wave still has to store
#pragma unroll 1
    for (int rel = 0; rel < len; ++rel) {
        const int it = tile_of(rel), buf = rel & (DP_RING - 1);
        const bool mine = (it & 1) == zrole;
        if (pend >= 0 && !(a.dbg & 2)) store_z(pend);
        pend = mine ? it : -1;
        // the image of the previous tile is free: every wave has passed the barrier behind it
        if (rel + DP_RING - 1 < len && !(a.dbg & 1)) issue(tile_of(rel + DP_RING - 1), (rel + DP_RING - 1) & (DP_RING - 1));
        compute(buf, mine);
        // the next tile must have landed.  Loads complete in order among themselves, and the two tiles issued behind it are
        // NBW / 2 (+ 1 for P) loads per wave each: no more than that many operations outstanding means it is in (outstanding
        // stores only make the wait longer).  The last tiles have fewer behind them: wait for everything.
        if (rel + DP_RING - 1 < len) {
            if (w < 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBW + 2) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBW) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        // (not __syncthreads(): its fence waits for every outstanding load -- the two tiles that are meant to keep travelling)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (pend >= 0 && !(a.dbg & 2)) store_z(pend);

    // the accumulators as they are: [b][slot][lane], slot = 4 * tile + register, 8 + strip
    double *out = a.slab + ((int64_t)id * NB + w * NBW) * 640 + lane;
#pragma unroll
    for (int bi = 0; bi < NBW; ++bi) {
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int j = 0; j < 4; ++j) out[(bi * 10 + 4 * p + j) * 64] = acc[bi][p][j];
        out[(bi * 10 + 8) * 64] = accs[bi][0];
        out[(bi * 10 + 9) * 64] = accs[bi][1];
    }
}

// U[b][p][t] = sum over the q chunks; one thread per accumulator element of a t range
__global__ __launch_bounds__(256) void dense_pass_reduce(const double *slab, int NB, int nt, int nqc, int T, int r, double *U)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int per = NB * 640;
    if (e >= (int64_t)per * nt) return;
    const int tr = (int)(e / per), f = (int)(e - (int64_t)tr * per);
    const int b = f / 640, slot = (f - 640 * b) >> 6, l = f & 63;
    double s = 0.0;
    for (int qc = 0; qc < nqc; ++qc) {
        const int id = (qc & 7) + 8 * (tr + nt * (qc >> 3));
        s += slab[(int64_t)id * per + f];
    }
    int p, t;
    if (slot < 8) {                       // 16x16x4: register j of lane l is row (l >> 4) + 4 j, column l & 15
        p = 16 * (slot >> 2) + (l & 15);
        t = (l >> 4) + 4 * (slot & 3);
    } else {                              // 4x4x4: lane (i = l >> 4, beta = (l >> 2) & 3, c = l & 3) is row 4 beta + i, column c
        p = 32 + 4 * (slot - 8) + (l & 3);
        t = 4 * ((l >> 2) & 3) + (l >> 4);
    }
    if (p < r) U[((int64_t)b * r + p) * T + 16 * tr + t] = s;
}

}  // namespace

}  // namespace ttsk

extern "C" int ttsk_dense_first_pass(const double *X, int64_t n0, int64_t Q, int64_t T, const double *C, int64_t ll,
                                     const double *P, int64_t r, double *Z, double *U, int stream)
{
    using namespace ttsk;
    TTSK_ARG(X && C && P && Z && U, "ttsk_dense_first_pass: NULL operand");
    TTSK_ARG(n0 > 0 && Q > 0 && T > 0 && ll > 0 && r > 0, "ttsk_dense_first_pass: empty extent");
    static const int on = [] { const char *e = getenv("TTSK_DENSE_ONE_PASS"); return e ? atoi(e) : 1; }();
    if (!on || (n0 != 32 && n0 != 64) || (T & 15) || (Q & 3) || ll > 20 || r > 40 || (r & 1) || Q * T >= (1ll << 27) ||
        (((uintptr_t)X | (uintptr_t)P) & 15)) {
        set_error("ttsk_dense_first_pass: shape outside the kernel's cover (first mode 32 or 64, last mode a multiple of 16, "
                  "middle extent a multiple of 4, left rank <= 20, even right rank <= 40)");
        return TTSK_ERR_UNSUPPORTED;
    }
    TTSK_STREAM(st, stream);
    const int nt = (int)(T / 16);
    int64_t nqc = 8 * std::max<int64_t>(1, 32 / nt);
    nqc = std::min<int64_t>(nqc, 8 * cdiv(Q / 4, 8));          // at least one tile for most chunks
    const int NB = (int)n0;
    const int64_t grid = (int64_t)nt * nqc;
    double *slab = (double *)scratch(stream, SCRATCH_MISC, (size_t)grid * NB * 640 * 8);
    if (!slab) return TTSK_ERR_HIP;
    static const int dbg = [] { const char *e = getenv("TTSK_DP_DBG"); return e ? atoi(e) : 0; }();
    DensePass a{X, Q * T, (int)Q, (int)T, C, (int)ll, P, (int)r, Z, slab, nt, (int)nqc, dbg};
    const size_t lds = (size_t)(DP_RING * NB * 64 + DP_RING * DP_PBUF + NB * 20) * 8;
    static bool attr = false;
    if (!attr) {
        TTSK_HIP(hipFuncSetAttribute((const void *)dense_pass_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
        TTSK_HIP(hipFuncSetAttribute((const void *)dense_pass_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
        attr = true;
    }
    if (prof_on()) prof_open_named(st, PROF_SOLVE, 0.0, "dense_pass");
    if (NB == 64)
        hipLaunchKernelGGL(dense_pass_kernel<8>, dim3((unsigned)grid), dim3(512), lds, st, a);
    else
        hipLaunchKernelGGL(dense_pass_kernel<4>, dim3((unsigned)grid), dim3(512), lds, st, a);
    TTSK_LAUNCH_CHECK();
    const int64_t elems = (int64_t)NB * 640 * nt;
    hipLaunchKernelGGL(dense_pass_reduce, dim3((unsigned)cdiv(elems, 256)), dim3(256), 0, st, slab, NB, nt, (int)nqc, (int)T, (int)r, U);
    TTSK_LAUNCH_CHECK();
    if (prof_on()) prof_close(st);
    return TTSK_OK;
}

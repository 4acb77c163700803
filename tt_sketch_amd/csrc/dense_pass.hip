// The pass over a dense tensor that a sketch with tensor-train DRMs starts with (reference dense_sketch.py:7-52 with the
// matrices of tensor_train_drm.py:109-122): BOTH products that read the tensor, from ONE read of it.
//
//   X[b][q][t]    the tensor, b = first mode (32, or a multiple of 64: blocks of NB = 64), t = last mode (T), q = everything in
//                 between (Q), C order
//   Z[p'][q][t] = sum_b C[b][p'] X[b][q][t]        the first left product, C the left DRM's first core (NB x ll), ll <= 20
//   U[b][p][t]  = sum_q P[q][p]  X[b][q][t]        the long-K half of Psi_0, P the right DRM's matrix of one core less (Q x r),
//                                                   r <= 40 (Psi_0 = U x last core follows as a small product)
//
// Z needs every b of a column in one workgroup, U sums over q: a workgroup owns 16 values of t (128-byte pieces of the tensor's
// rows: whole cache lines) and a range of q, for all b.  Its eight waves hold U for NB / 8 values of b each -- (2 tiles + 2
// four-wide strips) x NB / 8 accumulators of v_mfma_f64_16x16x4 / 4x4x4, 160 registers at NB = 64, which is what limits the
// t range to 16 -- and walk q in steps of 8: a (NB x 8 x 16) tile of X (64 KB) goes to LDS by global_load_lds (one 1 KB
// instruction per b: one row of the tensor, one page), two images, and is read twice from there: once with q as the K index of
// the matrix instruction (U: lanes (t, q)), once with b as the K index (Z: lanes (t, b); wave w takes row q0 + w and chains
// over all b).  Row q of b's chunk sits at row q ^ (b & 1): both fragment shapes then read 32 consecutive doubles per half
// wave (SQ_LDS_BANK_CONFLICT = 0).  The next tile's loads are issued one instruction per step from the second step of a tile on, so
// that the matrix pipes start right behind the barrier -- which carries no fence -- and no wave's instruction stream waits behind a
// burst of loads.  The tensor is read once (8.59 GB at C2 instead of 17.2), Z
// is written once, the partial U of the q ranges are summed by a second launch.
//
// The second pair of a sketch -- Z_1 and Psi_1 from Z_0, whose rows are (p', i_1): 1280 at C2 -- is the same two sums one level down:
// a first mode beyond 64 runs as blocks of 64 values of b (blockIdx = block * (t ranges * q ranges) + ...), U per block as it is,
// Z as one partial sum per block (scratch) that a third launch adds up: 2.7 GB read once + 0.84 GB of partial Z instead of 2.7 GB
// read twice (C2: 1.7 -> 1.27 ms).
//
// Work per tile at NB = 64: U 2 x 64 x (2 x 64 + 2 x 16) + Z 8 x 16 x (64 + 16) = 30720 cycles of the matrix pipes per 64 KB
// = 129 GF per C2 sketch (no padded rows: 20 = 16 + 4, 40 = 32 + 2 x 4), 1.64 ms at the fp64 peak (SQ_VALU_MFMA_BUSY_CYCLES
// = 4.03e9 over 1024 SIMDs agrees); the HBM side is 8.59 + 2.68 GB = 2.1 ms at the 5.5 TB/s a mixed stream reaches.
// Measured: 2.58 - 2.66 ms = 0.62 - 0.64 of the matrix peak, 4.2 - 4.4 TB/s; the same instruction stream
// takes 2.05 ms without its loads and stores.
//
// Ranks beyond 20 / 40 (round 4): the kernel is a template over the shapes of its two accumulator sets -- U: TP tiles of 16
// columns + SP strips of 4 (r <= 16 TP + 4 SP), Z: ZT tiles of 16 rows + ZS strips (ll <= 16 ZT + 4 ZS).  (2 + 2, 1 + 1) is the
// C2 instantiation above; r <= 48 runs 3 + 0, r <= 64 4 + 0, ll <= 32 2 + 0, with NBW = 4 (blocks of 32 values of b: 128
// accumulator registers at 4 + 0) -- a first mode of 64 is then two blocks, their partial Z summed as for longer modes.  An odd
// r reads a copy of P padded to r + 1 columns (16-byte units need an even row).
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

// Timing experiments (TTSK_DP_DBG) exist in a lab build only (-DTTSK_LAB)
#ifdef TTSK_LAB
#define DP_DBG(bit) (a.dbg & (bit))
#else
#define DP_DBG(bit) 0
#endif

struct DensePass {
    const double *X;
    int64_t sb;            // elements between consecutive b
    int Q, T;
    const double *C;
    int ll;
    const double *P;
    int r, pr;             // columns of P in use, its row length (even)
    double *Z;
    double *slab;          // [workgroup][NB][4 TP + SP][64]: the accumulators as the lanes hold them
    int nt, nqc;           // t ranges (T / 16), q chunks (a multiple of 8)
    int64_t zblock;        // first mode beyond 64: blocks of 64 values of b, blockIdx = block * nt * nqc + ...; block k's partial Z at Z + k * zblock
    int split;             // 1: two kinds of workgroups per (t range, q range, block): kind k takes columns [k p_half, ..) of P / U and
    int p_half, z_half;    //    rows [k z_half, ..) of Z -- each reads the tile (the second read from L2 / MALL); 0: one kind takes all
    int dbg;               // diagnostics (TTSK_DP_DBG): 1 = no tile loads after the first, 2 = no Z stores, 4 = no rotated start, 8 = loads and stores of a tile in one burst
};

// doubles per row of the P image: the columns in use rounded up to 16 mod 32 (rows kq, kq + 1 on disjoint halves of the banks):
// 48 for 2 tiles + 2 strips (40 used) and for 3 tiles, 80 for 4 tiles
constexpr int dp_prow(int TP, int SP) { return (16 * TP + 4 * SP + 15) / 32 * 32 + 16; }

template <int NBW, int TP, int SP, int ZT, int ZS>
__global__ __launch_bounds__(512) void dense_pass_kernel(DensePass a)
{
    constexpr int NB = 8 * NBW;
    constexpr int DP_PROW = dp_prow(TP, SP), DP_PBUF = 8 * DP_PROW, DP_PU = DP_PROW / 2;
    constexpr int CW = 16 * ZT + 4 * ZS;           // columns of C held
    static_assert((8 * DP_PU) % 64 == 0 && 8 * DP_PU <= 512, "the rows of P are brought by whole waves");
    extern __shared__ double lds[];
    double *Xl = lds;                              // [2][NB][8][16]
    double *Pl = Xl + 2 * NB * 128;                // [2][8][DP_PROW]
    double *Cf = Pl + 2 * DP_PBUF;                 // [ZT][NB][16]   columns 16 zt .. of C
    double *Cs = Cf + ZT * NB * 16;                // [ZS][NB][4]    columns 16 ZT + 4 zs ..

    const int tid = threadIdx.x, lane = tid & 63, x16 = lane & 15, kq = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // kinds alternate in groups of 8 consecutive ids: both kinds of a (t range, q range) on one XCD, one dispatch round apart
    const int kind = a.split ? (int)((blockIdx.x >> 3) & 1) : 0;
    const int id = a.split ? (int)(((blockIdx.x >> 4) << 3) | (blockIdx.x & 7)) : (int)blockIdx.x;
    const int p_off = kind * a.p_half, z_off = kind * a.z_half;
    const int per_block = a.nt * a.nqc, bb = id / per_block, idl = id - bb * per_block;
    // the nt workgroups that share the 512-byte rows of a q range sit on one XCD (ids 8 apart) next to each other in time
    const int tr = (idl >> 3) % a.nt, qc = (idl & 7) + 8 * (idl / (8 * a.nt));
    const int t0 = 16 * tr;
    const int total = a.Q >> 3;
    const int it_beg = (int)((int64_t)qc * total / a.nqc), it_end = (int)((int64_t)(qc + 1) * total / a.nqc);

    for (int e = tid; e < NB * CW; e += 512) {
        const int b = e / CW, c = e - CW * b;
        const double v = z_off + c < a.ll ? a.C[((int64_t)bb * NB + b) * a.ll + z_off + c] : 0.0;
        if (c < 16 * ZT) Cf[((c >> 4) * NB + b) * 16 + (c & 15)] = v;
        else Cs[(((c - 16 * ZT) >> 2) * NB + b) * 4 + (c & 3)] = v;
    }

    // tile `it` (8 values of q) -> image `buf`: wave w brings the chunks of its own NBW values of b, one 1 KB instruction each
    // (one row of the tensor = one page per instruction), row q of b's chunk to row q ^ (b & 1); the first waves the 8 rows of P
    // (DP_PU 16-byte units per row, the last of them padding: whatever they read is never used).  Addresses as a wave-uniform
    // 64-bit base plus a 32-bit lane offset (the saddr form: one register per lane, not two per chunk).
    const int xrow = lane >> 3, xcol = 2 * (lane & 7);
    const uint32_t xoff0 = (uint32_t)((xrow * a.T + xcol) * 8), xoff1 = (uint32_t)(((xrow ^ 1) * a.T + xcol) * 8);
    const int pU = 64 * w + lane, prow = (pU / DP_PU) & 7, ppair = pU % DP_PU;
    const uint32_t poff = (uint32_t)((prow * a.pr + (p_off + 2 * ppair + 2 <= a.pr ? p_off + 2 * ppair : 0)) * 8);
    const char *xbase = (const char *)a.X + (((int64_t)bb * NB + w * NBW) * a.sb + t0) * 8;
    // (one chunk at a time: `issue_x(tile offset, image, u)` is spread over the steps of a tile -- eight load instructions in a
    // row from each of eight waves stall every wave's instruction stream, and with it the matrix pipes, behind the address unit:
    // 2.86 -> 2.58 ms at C2)
    auto issue_x = [&](int64_t tile, int buf, int u) {
        const char *src = xbase + uniform_i64(tile + (int64_t)u * a.sb * 8) + ((u & 1) ? xoff1 : xoff0);
        // (aux = 2: non-temporal -- every byte of the tensor is used exactly once; 3 % on the whole kernel)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(Xl + (buf * NB + w * NBW + u) * 128), 16, 0, 2);
    };
    auto issue_p = [&](int it, int buf) {
        if (w < 8 * DP_PU / 64) {
            const char *src = (const char *)a.P + uniform_i64((int64_t)it * 8 * a.pr * 8) + poff;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(Pl + buf * DP_PBUF + w * 128), 16, 0, 0);
        }
    };
    auto issue = [&](int it, int buf) {
        const int64_t tile = uniform_i64((int64_t)it * 8 * a.T * 8);
#pragma unroll
        for (int u = 0; u < NBW; ++u) issue_x(tile, buf, u);
        issue_p(it, buf);
    };

    v4d acc[NBW][TP > 0 ? TP : 1];
    double accs[NBW][SP > 0 ? SP : 1];
#pragma unroll
    for (int bi = 0; bi < NBW; ++bi) {
#pragma unroll
        for (int p = 0; p < TP; ++p)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[bi][p][j] = 0.0;
#pragma unroll
        for (int p = 0; p < (SP > 0 ? SP : 1); ++p) accs[bi][p] = 0.0;
    }

    // Z: wave w chains row q0 + w of the tile over all b; the row is stored while the next tile is computed on
    v4d zp[ZT > 0 ? ZT : 1];
    double zps[ZS > 0 ? ZS : 1];
#pragma unroll
    for (int zt = 0; zt < ZT; ++zt)
#pragma unroll
        for (int j = 0; j < 4; ++j) zp[zt][j] = 0.0;
#pragma unroll
    for (int zs = 0; zs < (ZS > 0 ? ZS : 1); ++zs) zps[zs] = 0.0;
    const int64_t zs_row = (int64_t)a.Q * a.T;
    const uint32_t zoff = (uint32_t)((kq * zs_row + x16) * 8);          // 3 Q T doubles at most: the host checks the range
    const int z_end = a.split && kind == 0 && a.z_half < a.ll ? a.z_half : a.ll;          // rows of Z this kind owns: [z_off, z_end)
    auto store_z = [&](int it) {
        const int64_t at = uniform_i64(((int64_t)bb * a.zblock + ((int64_t)it * 8 + w) * a.T + t0) * 8);
#pragma unroll
        for (int zt = 0; zt < ZT; ++zt)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (z_off + 16 * zt + kq + 4 * j < z_end)
                    *(double *)((char *)a.Z + uniform_i64(at + (z_off + 16 * zt + 4 * j) * zs_row * 8) + zoff) = zp[zt][j];
#pragma unroll
        for (int zs = 0; zs < ZS; ++zs)
            if (z_off + 16 * ZT + 4 * zs + kq < z_end)
                *(double *)((char *)a.Z + uniform_i64(at + (z_off + 16 * ZT + 4 * zs) * zs_row * 8) + zoff) = zps[zs];
    };

    // The range is walked from a start that differs from workgroup to workgroup (the same for the nt workgroups that share
    // rows): with power-of-two extents every b row and every q range begins at the same offset modulo 2 MB.
    const int len = it_end - it_beg;
    const int rot = len > 0 && !(DP_DBG(4)) ? (int)(((uint32_t)qc * 2654435761u >> 8) % (uint32_t)len) : 0;
    auto tile_of = [&](int rel) { const int t = rel + rot; return it_beg + (t >= len ? t - len : t); };

    if (len > 0) issue(tile_of(0), 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

#pragma unroll 1
    for (int rel = 0; rel < len; ++rel) {
        const int buf = rel & 1;
        const double *xb = Xl + buf * NB * 128, *pb = Pl + buf * DP_PBUF;
        // 2 NBW steps; step s: U of (4-q block j = s / NBW, b = w NBW + s % NBW) with q as the K index (2 tiles + 2 strips of
        // p), and link s of the Z chain (b = 4 s .. 4 s + 3 as the K index).  The fragments of step s + 1 are read behind the
        // first matrix instruction of step s: the compiler puts lgkmcnt(0) in front of a step's first use, so the reads must
        // be neither younger than that wait nor right behind it; the scheduling barrier per step keeps that order -- left
        // alone, the compiler hoists every read of the tile to the front and spills the accumulators.
        v4d z[ZT > 0 ? ZT : 1];
        double zs[ZS > 0 ? ZS : 1];
#pragma unroll
        for (int zt = 0; zt < ZT; ++zt)
#pragma unroll
            for (int j = 0; j < 4; ++j) z[zt][j] = 0.0;
#pragma unroll
        for (int q = 0; q < (ZS > 0 ? ZS : 1); ++q) zs[q] = 0.0;
        const int next_it = rel + 1 < len ? tile_of(rel + 1) : 0;
        const int64_t next_tile = uniform_i64((int64_t)next_it * 8 * a.T * 8);
        const double *xu = xb + w * NBW * 128 + x16;
        const double *xz = xb + kq * 128 + (w ^ (kq & 1)) * 16 + x16;
        const double *cf = Cf + kq * 16 + x16, *cs = Cs + kq * 4 + (x16 & 3);
        const double *pr = pb + kq * DP_PROW + x16, *prs = pb + kq * DP_PROW + 16 * TP + (x16 & 3);
        double pf[TP > 0 ? TP : 1], ps[SP > 0 ? SP : 1], cff[ZT > 0 ? ZT : 1], csf[ZS > 0 ? ZS : 1];
#pragma unroll
        for (int p = 0; p < TP; ++p) pf[p] = pr[16 * p];
#pragma unroll
        for (int p = 0; p < SP; ++p) ps[p] = prs[4 * p];
#pragma unroll
        for (int zt = 0; zt < ZT; ++zt) cff[zt] = cf[zt * NB * 16];
#pragma unroll
        for (int q = 0; q < ZS; ++q) csf[q] = cs[q * NB * 4];
        double xf = xu[kq * 16], xzf = xz[0];
#pragma unroll
        for (int s = 0; s < 2 * NBW; ++s) {
            const int bi = s % NBW;
            double xf_n = 0.0, xz_n = 0.0, pf_n[TP > 0 ? TP : 1], ps_n[SP > 0 ? SP : 1], cf_n[ZT > 0 ? ZT : 1], cs_n[ZS > 0 ? ZS : 1];
#pragma unroll
            for (int p = 0; p < TP; ++p) pf_n[p] = pf[p];
#pragma unroll
            for (int p = 0; p < SP; ++p) ps_n[p] = ps[p];
#pragma unroll
            for (int zt = 0; zt < ZT; ++zt) cf_n[zt] = 0.0;
#pragma unroll
            for (int q = 0; q < ZS; ++q) cs_n[q] = 0.0;
            if (s + 1 < 2 * NBW) {
                const int jn = (s + 1) / NBW, bn = (s + 1) % NBW;
                xf_n = xu[bn * 128 + ((4 * jn + kq) ^ (bn & 1)) * 16];
                xz_n = xz[(s + 1) * 512];
#pragma unroll
                for (int zt = 0; zt < ZT; ++zt) cf_n[zt] = cf[zt * NB * 16 + (s + 1) * 64];
#pragma unroll
                for (int q = 0; q < ZS; ++q) cs_n[q] = cs[q * NB * 4 + (s + 1) * 16];
                if (bn == 0) {
#pragma unroll
                    for (int p = 0; p < TP; ++p) pf_n[p] = pr[4 * jn * DP_PROW + 16 * p];
#pragma unroll
                    for (int p = 0; p < SP; ++p) ps_n[p] = prs[4 * jn * DP_PROW + 4 * p];
                }
            }
            if constexpr (TP > 0) acc[bi][0] = mfma16(xf, pf[0], acc[bi][0]);
            if constexpr (ZT > 0) z[0] = mfma16(cff[0], xzf, z[0]);
#pragma unroll
            for (int p = 1; p < TP; ++p) acc[bi][p] = mfma16(xf, pf[p], acc[bi][p]);
#pragma unroll
            for (int zt = 1; zt < ZT; ++zt) z[zt] = mfma16(cff[zt], xzf, z[zt]);
#pragma unroll
            for (int p = 0; p < SP; ++p) accs[bi][p] = mfma4s(xf, ps[p], accs[bi][p]);
#pragma unroll
            for (int q = 0; q < ZS; ++q) zs[q] = mfma4s(csf[q], xzf, zs[q]);
            xf = xf_n; xzf = xz_n;
#pragma unroll
            for (int p = 0; p < TP; ++p) pf[p] = pf_n[p];
#pragma unroll
            for (int p = 0; p < SP; ++p) ps[p] = ps_n[p];
#pragma unroll
            for (int zt = 0; zt < ZT; ++zt) cff[zt] = cf_n[zt];
#pragma unroll
            for (int q = 0; q < ZS; ++q) csf[q] = cs_n[q];
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2 + ZT + ZS + TP + SP, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, TP + SP + ZT + ZS - 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            // behind the first step, so that the matrix pipes start right after the barrier, and ONE instruction per step: the
            // next tile's loads (its image was last read before that barrier) at steps 1 .. NBW, the rows of P, then the
            // previous tile's row of Z.  (TTSK_DP_DBG=8: all of them behind step 1, as one burst.)
            if ((DP_DBG(8)) ? s == 1 : (s >= 1 && s <= NBW + 2)) {
                const bool more = rel + 1 < len && !(DP_DBG(1));
                if (DP_DBG(8)) {
                    if (more) issue(tile_of(rel + 1), buf ^ 1);
                    if (rel > 0 && !(DP_DBG(2))) store_z(tile_of(rel - 1));
                } else {
                    if (s <= NBW) { if (more) issue_x(next_tile, buf ^ 1, s - 1); }
                    else if (s == NBW + 1) { if (more) issue_p(next_it, buf ^ 1); }
                    else if (rel > 0 && !(DP_DBG(2))) store_z(tile_of(rel - 1));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int zt = 0; zt < ZT; ++zt) zp[zt] = z[zt];
#pragma unroll
        for (int q = 0; q < ZS; ++q) zps[q] = zs[q];
        // the next tile has landed (and the stores are done); no fence: LDS traffic of this wave complete, then the barrier
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (len > 0 && !(DP_DBG(2))) store_z(tile_of(len - 1));

    // the accumulators as they are: [b][slot][lane], slot = 4 * tile + register, 4 TP + strip
    constexpr int SL = 4 * TP + SP;
    double *out = a.slab + (((int64_t)kind * gridDim.x / (a.split ? 2 : 1) + id) * NB + w * NBW) * (SL * 64) + lane;
#pragma unroll
    for (int bi = 0; bi < NBW; ++bi) {
#pragma unroll
        for (int p = 0; p < TP; ++p)
#pragma unroll
            for (int j = 0; j < 4; ++j) out[(bi * SL + 4 * p + j) * 64] = acc[bi][p][j];
#pragma unroll
        for (int p = 0; p < SP; ++p) out[(bi * SL + 4 * TP + p) * 64] = accs[bi][p];
    }
}

// U[b][p][t] = sum over the q chunks; one thread per accumulator element of a t range
__global__ __launch_bounds__(256) void dense_pass_reduce(const double *slab, int NB, int TP, int SP, int nt, int nqc, int nbb, int T, int r, double *U,
                                                         int p_off, int pcount)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int sl64 = (4 * TP + SP) * 64, per = NB * sl64;
    if (e >= (int64_t)per * nt * nbb) return;
    const int bt = (int)(e / per), f = (int)(e - (int64_t)bt * per);
    const int bb = bt / nt, tr = bt - bb * nt;
    const int b = f / sl64, slot = (f - sl64 * b) >> 6, l = f & 63;
    double s = 0.0;
    for (int qc = 0; qc < nqc; ++qc) {
        const int64_t id = (int64_t)bb * nt * nqc + (qc & 7) + 8 * (tr + nt * (qc >> 3));
        s += slab[id * per + f];
    }
    int p, t;
    if (slot < 4 * TP) {                  // 16x16x4: register j of lane l is row (l >> 4) + 4 j, column l & 15
        p = 16 * (slot >> 2) + (l & 15);
        t = (l >> 4) + 4 * (slot & 3);
    } else {                              // 4x4x4: lane (i = l >> 4, beta = (l >> 2) & 3, c = l & 3) is row 4 beta + i, column c
        p = 16 * TP + 4 * (slot - 4 * TP) + (l & 3);
        t = 4 * ((l >> 2) & 3) + (l >> 4);
    }
    if (p < pcount) U[(((int64_t)bb * NB + b) * r + p_off + p) * T + 16 * tr + t] = s;
}

// Z = sum of the blocks' partial Z (first mode beyond 64)
__global__ __launch_bounds__(256) void dense_pass_zsum(const double2 *part, int nbb, int64_t pairs, double2 *Z)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= pairs) return;
    double2 s = part[e];
    for (int k = 1; k < nbb; ++k) {
        const double2 v = part[(int64_t)k * pairs + e];
        s.x += v.x;
        s.y += v.y;
    }
    Z[e] = s;
}

// P with one zero column appended (an odd r: the 16-byte units of the P loader need an even row)
__global__ __launch_bounds__(256) void dense_pass_pad_p(const double *P, int64_t Q, int r, double *out)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= Q * (r + 1)) return;
    const int64_t q = e / (r + 1);
    const int c = (int)(e - q * (r + 1));
    out[e] = c < r ? P[q * r + c] : 0.0;
}

template <int NBW, int TP, int SP, int ZT, int ZS>
int dense_pass_launch(const DensePass &a, int64_t grid, hipStream_t st)
{
    constexpr int NB = 8 * NBW;
    const size_t lds = (size_t)(2 * NB * 128 + 2 * 8 * dp_prow(TP, SP) + NB * (16 * ZT + 4 * ZS)) * 8;
    static PerInit attr;
    if (attr.first() && hipFuncSetAttribute((const void *)dense_pass_kernel<NBW, TP, SP, ZT, ZS>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            156 * 1024) != hipSuccess)
        return TTSK_ERR_HIP;
    hipLaunchKernelGGL((dense_pass_kernel<NBW, TP, SP, ZT, ZS>), dim3((unsigned)grid), dim3(512), lds, st, a);
    return hipGetLastError() == hipSuccess ? TTSK_OK : TTSK_ERR_HIP;
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
    if (!on || (n0 & 31) || n0 > 64 * 64 || (T & 15) || (Q & 7) || ll > 32 || r > 64 || Q * T >= (1ll << 27) ||
        (((uintptr_t)X | (uintptr_t)P | (uintptr_t)Z) & 15)) {
        set_error("ttsk_dense_first_pass: shape outside the kernel's cover (first mode a multiple of 32, last mode a multiple "
                  "of 16, middle extent a multiple of 8, left rank <= 32, right rank <= 64)");
        return TTSK_ERR_UNSUPPORTED;
    }
    TTSK_STREAM(st, stream);
    const int nt = (int)(T / 16);
    // accumulator shapes: U as TP tiles + SP strips of columns, Z as ZT tiles + ZS strips of rows.  Up to 40 / 20: one kind of
    // workgroup, blocks of 64 values of b (the C2 shape).  Beyond, with whole blocks of 64: TWO kinds of workgroups, each with
    // half the columns of U and half the rows of Z -- the accumulators of all 64 b fit, Z needs no partial sums over blocks
    // of b, the matrix work stays proportional to the ranks, and the tile is read twice, the kinds of a (t range, q range) one
    // dispatch round apart on one XCD.  Otherwise (first mode a multiple of 32 only): blocks of 32 with wider accumulators.
    // just beyond 20 / 40 (r <= 44, ll <= 24): one more strip of either still fits one kind of workgroup (11 spilled registers, outside the step loop)
    static const int one_kind_on = [] { const char *e = getenv("TTSK_DP_ONE_KIND"); return e ? atoi(e) : 1; }();
    const bool one_kind = one_kind_on && !(n0 & 63) && (r > 40 || ll > 20) && r <= 44 && ll <= 24;
    const bool split = !one_kind && !(n0 & 63) && (r > 40 || ll > 20);
    const int p_half = split ? (int)((r + 1) / 2 + 3) / 4 * 4 : 0, z_half = split ? (int)((ll + 1) / 2 + 3) / 4 * 4 : 0;
    const int ushape = split ? (p_half <= 24 ? 3 : 4) : r <= 40 ? 0 : r <= 48 ? 1 : 2;
    const int zshape = split ? (z_half <= 12 ? 2 : 3) : ll <= 20 ? 0 : 1;
    const int TP = one_kind ? 2 : ushape == 0 ? 2 : ushape == 1 ? 3 : ushape == 2 ? 4 : ushape == 3 ? 1 : 2;
    const int SP = one_kind ? 3 : ushape == 0 || ushape == 3 ? 2 : 0;
    const int NB = (one_kind || split || (ushape == 0 && zshape == 0 && !(n0 & 63))) ? 64 : 32, nbb = (int)(n0 / NB);
    const int kinds = split ? 2 : 1;
    // q ranges: about 256 workgroups for one block of b, about 1024 over all blocks otherwise (several rounds over the CUs even
    // out the ranges); a multiple of 8, and not more than there are tiles
    int64_t nqc = 8 * std::max<int64_t>(1, (nbb == 1 ? 32 : (128 + nbb * nt / 2) / (nbb * nt)) / (nbb == 1 ? nt : 1));
    static const int nqc_env = [] { const char *e = getenv("TTSK_DP_NQC"); return e ? atoi(e) : 0; }();   // A/B runs
    if (nqc_env > 0 && nbb > 1) nqc = 8 * cdiv(nqc_env, 8);
    nqc = std::min<int64_t>(nqc, 8 * cdiv(Q / 8, 8));          // at least one tile for most chunks
    const int64_t grid1 = (int64_t)nbb * nt * nqc, grid = grid1 * kinds;
    const int64_t sl64 = (4 * TP + SP) * 64;
    // partial results beyond 4 GB (many blocks of a large operand): not this kernel's case -- the caller forms the products separately
    if ((nbb > 1 && (int64_t)nbb * ll * Q * T * 8 > (4ll << 30)) || grid * NB * sl64 * 8 > (4ll << 30)) {
        set_error("ttsk_dense_first_pass: %d blocks of rows would need more than 4 GB of partial results", nbb);
        return TTSK_ERR_UNSUPPORTED;
    }
    const int64_t pr = r + (r & 1);
    const size_t pad_bytes = (r & 1) ? (size_t)Q * pr * 8 + 64 : 0;
    char *ws = (char *)scratch(stream, SCRATCH_MISC, (size_t)grid * NB * sl64 * 8 + pad_bytes);
    if (!ws) return TTSK_ERR_HIP;
    double *slab = (double *)ws;
    if (r & 1) {
        double *pp = (double *)(ws + (size_t)grid * NB * sl64 * 8);
        hipLaunchKernelGGL(dense_pass_pad_p, dim3((unsigned)cdiv(Q * pr, 256)), dim3(256), 0, st, P, Q, (int)r, pp);
        TTSK_LAUNCH_CHECK();
        P = pp;
    }
    // blocks of b: each block's Z is a partial sum over its values of b
    const int64_t zblock = ll * Q * T;
    double *zout = Z;
    if (nbb > 1) {
        zout = (double *)scratch(stream, SCRATCH_GEMM, (size_t)nbb * zblock * 8);
        if (!zout) return TTSK_ERR_HIP;
    }
#ifdef TTSK_LAB
    static const int dbg = [] { const char *e = getenv("TTSK_DP_DBG"); return e ? atoi(e) : 0; }();
#else
    constexpr int dbg = 0;
#endif
    DensePass a{X, Q * T, (int)Q, (int)T, C, (int)ll, P, (int)r, (int)pr, zout, slab, nt, (int)nqc, zblock, split ? 1 : 0, p_half, z_half, dbg};
    if (prof_on()) prof_open_named(st, PROF_SOLVE, 0.0, "dense_pass");
    int rc;
    if (one_kind) rc = dense_pass_launch<8, 2, 3, 1, 2>(a, grid, st);
    else if (split) {
        if (ushape == 3) rc = zshape == 2 ? dense_pass_launch<8, 1, 2, 0, 3>(a, grid, st) : dense_pass_launch<8, 1, 2, 1, 0>(a, grid, st);
        else rc = zshape == 2 ? dense_pass_launch<8, 2, 0, 0, 3>(a, grid, st) : dense_pass_launch<8, 2, 0, 1, 0>(a, grid, st);
    } else if (NB == 64) rc = dense_pass_launch<8, 2, 2, 1, 1>(a, grid, st);
    else if (ushape == 0) rc = zshape == 0 ? dense_pass_launch<4, 2, 2, 1, 1>(a, grid, st) : dense_pass_launch<4, 2, 2, 2, 0>(a, grid, st);
    else if (ushape == 1) rc = zshape == 0 ? dense_pass_launch<4, 3, 0, 1, 1>(a, grid, st) : dense_pass_launch<4, 3, 0, 2, 0>(a, grid, st);
    else rc = zshape == 0 ? dense_pass_launch<4, 4, 0, 1, 1>(a, grid, st) : dense_pass_launch<4, 4, 0, 2, 0>(a, grid, st);
    if (rc != TTSK_OK) { set_error("ttsk_dense_first_pass: launch failed"); return rc; }
    const int64_t elems = (int64_t)NB * sl64 * nt * nbb;
    for (int k = 0; k < kinds; ++k) {
        const int p_off = k * p_half, pcount = split ? (k == 0 ? std::min<int>(p_half, (int)r) : (int)r - p_half) : (int)r;
        if (pcount <= 0) continue;
        hipLaunchKernelGGL(dense_pass_reduce, dim3((unsigned)cdiv(elems, 256)), dim3(256), 0, st, slab + (size_t)k * grid1 * NB * sl64, NB, TP, SP, nt,
                           (int)nqc, nbb, (int)T, (int)r, U, p_off, pcount);
        TTSK_LAUNCH_CHECK();
    }
    if (nbb > 1) {
        const int64_t pairs = zblock / 2;               // Q T is a multiple of 128
        hipLaunchKernelGGL(dense_pass_zsum, dim3((unsigned)cdiv(pairs, 256)), dim3(256), 0, st, (const double2 *)zout, nbb, pairs, (double2 *)Z);
        TTSK_LAUNCH_CHECK();
    }
    if (prof_on()) prof_close(st);
    return TTSK_OK;
}

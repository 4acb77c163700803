// SparseTensor x SparseGaussianDRM, streaming sketch, without the (nnz x rank) panels.
//
//   reference: sparse_gaussian_drm.py:29-44 (G[e, k] = ndtri(u(hash(flat_e + hash(k) + seed)))),
//              sparse_sketch.py:8-69      (Psi[:, j, :] = sum_{e: idx_mu[e] = j} val_e L[:, e] R[:, e]^T,
//                                          Omega = (L o val) R^T)
//
// The generator path (sampler.hip + sparse.hip) materialises every L_mu / R_mu as an (nnz x rank) matrix and
// reads it back through the mode permutation: ~18 GB of traffic at C4 for 480 MB of input.  Here one PASS per
// mode mu walks the nonzeros in mode-mu order ONCE and produces Psi_mu -- and one Omega riding along -- with
// every DRM row made where it is consumed:
//   * the nonzeros of a mode are a resident STREAM of records (flat prefix index, flat suffix index, mode
//     index, value) in mode order, built once per tensor (ttsk_sparse_mode_stream): 28 bytes per nonzero and
//     pass, read sequentially; the flat index of the prefix one mode longer (for the Omega that shares the right
//     factor) follows from the record by one multiply-add;
//   * a DRM factor with few possible prefixes / suffixes (the shallow modes: 200 or 30 000 rows at C4) is a
//     TABLE sampled once per sketch and gathered from L2; a deep one is SAMPLED in the pass, bit-identically
//     (same hash, same Cephes ndtri): a wave stages 32 nonzeros, computes the central branch of ndtri in place
//     and queues the tail samples in LDS so that only full waves pay for the tail code (as sample_rows_kernel);
//   * the products run on the matrix cores from the staged tile: lane (x, q) of k-block b holds val A[e][x] and
//     B[e][x] of nonzero e = 4 b + q, the operand layout of v_mfma_f64_16x16x4 (ranks <= 16 per factor);
//   * NO atomics: a wave stores the slices that lie inside its stretch, its first and last (shared) slices go to
//     per-wave partial blocks that a second kernel adds in wave order -- the sketch is bit-reproducible.
#include <cstdlib>
#include <type_traits>
#include <hipcub/hipcub.hpp>
#include "sampler_dev.h"

namespace ttsk {

struct SgF {
    int kind;            // 0: ones (width 1), 1: table gathered by flat index, 2: normals sampled in the pass, 3: sign rows sampled in the pass
    int w;               // columns of the factor (<= 16 NT)
    int rank_min;
    int src;             // flat index: 0 = prefix, 1 = suffix, 2 = prefix + j * mul, 3 = suffix + j * mul
    uint64_t mul, seed;
    const double *table;
    int full, nnz;       // kind 3: length of the whole DRM row ([rank_min, rank_min + w) of it is used), its +-1 entries
    int units, rcp;      // kind 1: a row of the staged block is ceil(w / 2) units of 16 bytes; floor(65536 / units) + 1
};

struct SgPass {
    const uint64_t *fl, *fr;   // flat prefix / suffix index of every record (w32: arrays of uint32 behind these pointers)
    int w32;
    const int32_t *jj;
    const double *val;
    size_t N, chunk;     // nonzeros, nonzeros per wave (multiple of 32)
    int64_t n;           // slices of Psi
    SgF f[3];            // Psi = (val A) (x) B by slice; Omega = (val C) (x) B  (c_left)  or  (val A) (x) C
    int c_left, has_om;
    int off[3], tcols;   // sampled factor: column offset in the staged tile (of what the products read), the tile's row length;
    int tab;             // table factor: offset of its block [SG_T][2 units] behind the tile; doubles of all table blocks
    int qcols;           // columns of the sampled factors: the tail queue holds at most SG_T * qcols slots
    double *psi;         // [wA][n][wB]
    double *part_psi;    // [wave][2][wA * wB]
    int *part_j;         // [wave][3]: first slice, last slice (= first if none), 1 if the last partial exists
    double *part_om;     // [wave][wOl * wOr]
#ifdef TTSK_LAB
    int lab;             // TTSK_SG_LAB: 1 = no table DMA, 2 = no products, 4 = no sampling (timing experiments; results are wrong)
#endif
};

constexpr int SG_T = 32;         // nonzeros per staged tile (the stretches of the waves are multiples of it)

// ndtri as a CALL in this kernel: inlined at its two sites it takes the pass kernel to ~230 VGPRs (two waves per SIMD,
// or 49 spilled registers under a tighter cap); the call costs a few scalar instructions per ~100 of arithmetic.
__device__ __attribute__((noinline)) double sg_ndtri(double u) { return ndtri_dev(u); }

__device__ __forceinline__ uint64_t sg_flat(const SgF &f, uint64_t fl, uint64_t fr, int j)
{
    const uint64_t base = (f.src & 1) ? fr : fl;
    return (f.src & 2) ? base + (uint64_t)(int64_t)j * f.mul : base;
}

__host__ __device__ inline size_t sg_per_wave(int tcols, int tab, int qcols, int T)
{
    return (size_t)T * tcols + tab + 4 * T + T / 2 + ((size_t)T * qcols + 3) / 4;
}

// NT: 16-column matrix tiles per factor.  NT = 1 (every factor <= 16 columns: C4) is the round-3 kernel; NT = 2 takes
// factors of up to 32 columns with 2 x 2 accumulator tiles per product (two waves per SIMD: 64 accumulator registers).
__device__ __forceinline__ double sg_mfma4(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }

// accumulators of one product when the factors reach NS 4-wide strips beyond their first 16 columns (NS = 1: <= 20 columns, 2: <= 24):
// the 16 x 16 tile, NS column strips (rows 0..15 x columns 16 + 4 q ..: A as for the tile, B replicated over the four blocks of
// v_mfma_f64_4x4x4), NS row strips (rows 16 + 4 q .. x columns 0..15: A replicated, B as for the tile) and the corner as the same
// row strips against B's second 16 columns.  64 + 3 NS x 16 matrix cycles per four nonzeros instead of the 256 of 2 x 2 tiles
// (fp64 matrix instructions run on the vector ALU's FMA units: padding is paid for).
template <int NS> struct SgEdge { v4d t; double sb[NS], sa[NS], c[NS]; };
template <int NS> struct SgOps { double t, s[NS], b1; };

// T: nonzeros per staged tile.  32, or 16 where three wide factors would leave room for ONE workgroup per CU (a 32-nonzero tile of
// 72 staged columns is 24 KB of LDS per wave): the sampling stage then deals 4 columns of a nonzero over the wave instead of 2.
template <int NT, int NS, int T>
__global__ __launch_bounds__(256, NT == 1 ? 3 : 2) void sg_pass_kernel(SgPass a)
{
    static_assert(NS == 0 || NT == 2, "edge strips belong to the wide instantiation");
    static_assert(T == 32 || T == 16, "tile of 32 or 16 nonzeros");
    constexpr int CP = 64 / T;                         // columns of one nonzero evaluated side by side in the sampling stage
    constexpr int NSA = NS ? NS : 1;
    extern __shared__ double sg_lds[];
    __shared__ uint64_t salt[3][16 * NT];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int x16 = lane & 15, kq = lane >> 4;
    if (tid < 48 * NT) {
        const int f = tid / (16 * NT), c = tid % (16 * NT);
        salt[f][c] = mix64((uint64_t)((a.f[f].kind == 3 ? 0 : a.f[f].rank_min) + c)) + a.f[f].seed;
    }
    __syncthreads();                                   // the only workgroup barrier: waves run free from here
    const int tcols = a.tcols;
    // per-wave LDS: tile[T][tcols] (the sampled factors) | table blocks [T][2 units] | byte offsets of the table rows [3][T] | val[T] | j[T] (int) |
    // queue (ushort, T * qcols)
    const size_t per_wave = sg_per_wave(tcols, a.tab, a.qcols, T);
    double *tile = sg_lds + (size_t)wv * per_wave;
    double *tabs = tile + T * tcols;
    uint64_t *ro = (uint64_t *)(tabs + a.tab);
    double *rv = (double *)(ro + 3 * T);
    int *rj = (int *)(rv + T);
    unsigned short *q = (unsigned short *)(rj + T);

    // (finite values everywhere a product may read: the rows of nonzeros beyond the stretch are multiplied by val = 0)
    for (int i = lane; i < T * tcols + a.tab; i += 64) tile[i] = 0.0;
    const size_t w_id = (size_t)blockIdx.x * 4 + wv;
    const size_t beg = w_id * a.chunk;
    const int wA = a.f[0].w, wB = a.f[1].w;
    const int wOl = a.c_left ? a.f[2].w : wA, wOr = a.c_left ? wB : a.f[2].w;
    int *pj = a.part_j + w_id * 3;
    if (beg >= a.N) {                                  // a padding wave of the last workgroup: nothing but zeros for the sums
        if (lane == 0) { pj[0] = 0x7fffffff; pj[1] = 0x7fffffff; pj[2] = 0; }
        if (a.has_om)
            for (int t = lane; t < wOl * wOr; t += 64) a.part_om[w_id * (size_t)(wOl * wOr) + t] = 0.0;
        return;
    }
    const size_t end = beg + a.chunk < a.N ? beg + a.chunk : a.N;
    const int jfirst = a.jj ? a.jj[beg] : 0;
    v4d accP[NT][NT], accO[NT][NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            accP[i][k] = v4d{0.0, 0.0, 0.0, 0.0};
            accO[i][k] = v4d{0.0, 0.0, 0.0, 0.0};
        }
    SgEdge<NSA> eP, eO;
    auto edge_zero = [](SgEdge<NSA> &P) {
        P.t = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < NSA; ++q) { P.sb[q] = 0.0; P.sa[q] = 0.0; P.c[q] = 0.0; }
    };
    edge_zero(eP);
    edge_zero(eO);
    // the cells of an edge set to dst[row * stride + col] where row < wa and col < wb
    auto edge_store = [&](const SgEdge<NSA> &P, double *dst, int64_t stride, int wa, int wb) {
        const int beta = x16 >> 2, j4 = x16 & 3;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int aa = 4 * t + kq;
            if (aa < wa && x16 < wb) dst[aa * stride + x16] = P.t[t];
        }
#pragma unroll
        for (int q = 0; q < NSA; ++q) {
            { const int aa = 4 * beta + kq, cc = 16 + 4 * q + j4; if (aa < wa && cc < wb) dst[aa * stride + cc] = P.sb[q]; }
            { const int aa = 16 + 4 * q + kq; if (aa < wa && x16 < wb) dst[aa * stride + x16] = P.sa[q]; }
            { const int aa = 16 + 4 * q + kq, cc = 16 + x16; if (aa < wa && cc < wb) dst[aa * stride + cc] = P.c[q]; }
        }
    };
    int cur = jfirst;
    bool first_done = false;

    // store the finished slice k: the wave's first slice and (final) its last one go to the partial blocks
    auto flush = [&](int k, bool final) {
        double *dst;
        int64_t stride_a;
        if (!first_done && k == jfirst) {
            dst = a.part_psi + (w_id * 2) * (size_t)(wA * wB);
            stride_a = wB;
            first_done = true;
        } else if (final) {
            dst = a.part_psi + (w_id * 2 + 1) * (size_t)(wA * wB);
            stride_a = wB;
            if (lane == 0) { pj[1] = k; pj[2] = 1; }
        } else {
            dst = a.psi + (size_t)k * wB;
            stride_a = (int64_t)a.n * wB;
        }
        if constexpr (NS > 0) {
            edge_store(eP, dst, stride_a, wA, wB);
            edge_zero(eP);
            return;
        }
#pragma unroll
        for (int ta = 0; ta < NT; ++ta)
#pragma unroll
            for (int tb = 0; tb < NT; ++tb)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int aa = 16 * ta + 4 * t + kq, cc = 16 * tb + x16;
                    if (aa < wA && cc < wB) dst[aa * stride_a + cc] = accP[ta][tb][t];
                    accP[ta][tb][t] = 0.0;
                }
    };
    if (lane == 0) { pj[0] = jfirst; pj[1] = jfirst; pj[2] = 0; }

    // the records of the NEXT tile travel while this one is worked on
    const int t32 = lane & (T - 1), half = lane / T;      // (nonzero of the tile, which of its CP column slots)
    auto rec_load = [&](size_t t0, uint64_t &xfl, uint64_t &xfr, int &xj, double &xv) {
        const size_t pos = t0 + t32;
        const bool in = pos < end;
        if (a.w32) {
            xfl = (in && a.fl) ? ((const uint32_t *)a.fl)[pos] : 0;
            xfr = (in && a.fr) ? ((const uint32_t *)a.fr)[pos] : 0;
        } else {
            xfl = (in && a.fl) ? a.fl[pos] : 0;
            xfr = (in && a.fr) ? a.fr[pos] : 0;
        }
        xj = in ? (a.jj ? a.jj[pos] : 0) : -1;
        xv = in ? a.val[pos] : 0.0;
    };
    uint64_t nx_fl, nx_fr;
    int nx_j;
    double nx_v;
    rec_load(beg, nx_fl, nx_fr, nx_j, nx_v);
    for (size_t t0 = beg; t0 < end; t0 += T) {
        // ---- (1) the records of the tile
        const uint64_t my_fl = nx_fl, my_fr = nx_fr;
        const int my_j = nx_j;
        const bool valid = my_j >= 0;
        if (lane < T) {
#pragma unroll
            for (int f = 0; f < 3; ++f)                // where the nonzero's row of table factor f starts (a missing nonzero: row 0)
                if (a.f[f].kind == 1) ro[f * T + lane] = valid ? sg_flat(a.f[f], my_fl, my_fr, my_j) * (uint64_t)(8 * a.f[f].w) : 0;
            rv[lane] = nx_v;
            rj[lane] = my_j;
        }
        const bool tile_one_slice = __ballot(valid && my_j != cur) == 0ull;
        rec_load(t0 + T, nx_fl, nx_fr, nx_j, nx_v);
        __builtin_amdgcn_wave_barrier();
        // ---- (2a) the table factors: row flat[t] of the table into a block [t][2 units] by LDS-DMA, 16 bytes per lane (unit
        // i = 64 k + lane of the block in instruction k: nonzero i / units, unit i % units of its row), no registers held: the rows
        // of ALL table factors travel while the sampled factors are evaluated, and are waited for once, in front of the
        // products.  (The register gather of rounds 3-4 paid one round trip per table factor; dword DMAs, one per column,
        // were bound by the address unit: 40 instructions per tile at C4 against 7 now.)  An odd row's last unit reads 8 bytes
        // of the next row (the table has a spare row behind its last one); rows are 8-byte aligned only.
#pragma unroll 1
        for (int f = 0; f < 3; ++f) {
            const SgF &F = a.f[f];
            if (F.kind != 1) continue;
#ifdef TTSK_LAB
            if (a.lab & 1) continue;
#endif
            double *blk = tabs + a.off[f];
#pragma unroll 1
            for (int i0 = 0; i0 < T * F.units; i0 += 64) {
                const int i = i0 + lane;
                const int t = (i * F.rcp) >> 16, cu = i - t * F.units;
                if (t < T) {
                    const char *src = (const char *)F.table + ro[f * T + t] + 16 * cu;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                     (__attribute__((address_space(3))) void *)(blk + 2 * i0), 16, 0, 0);
                }
            }
        }
        // ---- (2b) the sampled factors into the tile
        int qn = 0;
#pragma unroll 1
        for (int f = 0; f < 3; ++f) {
            const SgF &F = a.f[f];
#ifdef TTSK_LAB
            if (a.lab & 4) continue;
#endif
            if (F.kind == 2) {
                const uint64_t flat = sg_flat(F, my_fl, my_fr, my_j);
                for (int ci = 0; CP * ci < F.w; ++ci) {         // the same trip count in every part: the ballots below are wave-wide
                    const int c = CP * ci + half;
                    const bool act = valid && c < F.w;
                    const uint64_t h = mix64(flat + salt[f][c & (16 * NT - 1)]);
                    const uint64_t bits = (h | 0x2000000000000000ULL) & 0x3FFFFFFFFFFFFFFFULL;
                    const double u = mant_unit(bits);
                    const double expm2 = 0.13533528323661269189;
                    const int slot = t32 * tcols + a.off[f] + c;
                    const bool central = u > expm2 && u <= 1.0 - expm2;
                    const bool tail = act && !central;
                    if (act && central) tile[slot] = ndtri_central_dev(u);      // inline: no call (and no wait for the DMA) in this stage
                    const unsigned long long m = __ballot(tail);
                    if (tail) {
                        tile[slot] = u;
                        q[qn + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)slot;
                    }
                    qn += __popcll(m);
                }
            } else if (F.kind == 3) {
                // a sparse-sign row (fast_lazy_gaussian.pyx:121-180, as sign_kernel of sampler.hip): +-1 at the first nnz
                // positions of a row of `full` zeros, then nnz swaps in order.  The whole row lives in the tile, at columns
                // off - rank_min ..; the products read [off, off + w).  Signs by both halves, the swaps by one lane per nonzero.
                const uint64_t flat = sg_flat(F, my_fl, my_fr, my_j);
                double *row = tile + t32 * tcols + (a.off[f] - F.rank_min);
                for (int c = half; c < F.full; c += CP) {
                    double s = 0.0;
                    if (c < F.nnz) {
                        const uint64_t h = mix64(flat + salt[f][c]);
                        const uint64_t bits = (h | 0x2000000000000000ULL) & 0x3FFFFFFFFFFFFFFFULL;
                        const int ex = (int)((bits >> 52) & 0x7FF) - 1022;
                        s = (double)((((ex % 2) + 2) % 2) * 2 - 1);
                    }
                    row[c] = s;
                }
                __builtin_amdgcn_wave_barrier();
                if (half == 0)
                    for (int c = 0; c < F.nnz; ++c) {
                        const uint64_t h = mix64(flat + salt[f][c]);
                        const double u = mant_unit((h | 0x2000000000000000ULL) & 0x3FFFFFFFFFFFFFFFULL);
                        const int pick = (int)(u * (double)(F.full - c) + (double)c);
                        const double x = row[c], y = row[pick];
                        row[c] = y;
                        row[pick] = x;
                    }
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ---- (3) the tail samples, a full wave at a time
        for (int i = 0; i < qn; i += 64) {
            if (i + lane < qn) {
                const int slot = q[i + lane];
                tile[slot] = sg_ndtri(tile[slot]);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the table blocks have landed
        __builtin_amdgcn_wave_barrier();
        // ---- (4) the products: k-block b = nonzeros 4 b .. 4 b + 3 of the tile
        // element (nonzero e, column c) of factor i: at [e][pitch of its table block, or of the tile][c]
        const int se0 = a.f[0].kind == 1 ? 2 * a.f[0].units : tcols, se1 = a.f[1].kind == 1 ? 2 * a.f[1].units : tcols,
                  se2 = a.f[2].kind == 1 ? 2 * a.f[2].units : tcols;
        const double *p0 = (a.f[0].kind == 1 ? tabs : tile) + a.off[0] + kq * se0 + x16;
        const double *p1 = (a.f[1].kind == 1 ? tabs : tile) + a.off[1] + kq * se1 + x16;
        const double *p2 = (a.f[2].kind == 1 ? tabs : tile) + a.off[2] + kq * se2 + x16;
#ifdef TTSK_LAB
        if (a.lab & 2) continue;
#endif
        // operands of k-block b: (val A)[e][c], B[e][c], C[e][c] for e = 4 b + kq, c = 16 t + x16.  No masks: a column beyond a
        // factor's width only reaches result cells that are never stored, and a nonzero beyond the stretch has val = 0 and finite
        // (stale or zero-initialised) factor rows
        auto operands = [&](int b, double v, double (&av)[NT], double (&bv)[NT], double (&cv)[NT]) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const double la = p0[4 * b * se0 + 16 * t], lb = p1[4 * b * se1 + 16 * t], lc = p2[4 * b * se2 + 16 * t];
                const double one = 16 * t + x16 == 0 ? 1.0 : 0.0;
                av[t] = (a.f[0].kind ? la : one) * v;
                bv[t] = a.f[1].kind ? lb : one;
                cv[t] = lc;
            }
        };
        if constexpr (NS > 0) {
            // ---- factors of 17 .. 16 + 4 NS columns: tile + edge strips.  Per factor and k-block: the tile operand (column x16), NS
            // strip operands (column 16 + 4 q + (x16 & 3)) and the second-tile operand (column 16 + x16); whatever lies beyond a
            // factor's width is garbage that only reaches cells never stored.
            const double one16 = x16 == 0 ? 1.0 : 0.0;
            const int j4 = x16 & 3;
            const double *q0 = p0 - x16, *q1 = p1 - x16, *q2 = p2 - x16;
            const bool one0 = a.f[0].kind == 0, one1 = a.f[1].kind == 0;
            auto ld = [&](const double *qf, int se, bool one, int b) {
                const double *r = qf + 4 * b * se;
                SgOps<NSA> o;
                o.t = one ? one16 : r[x16];
                o.b1 = one ? 0.0 : r[16 + x16];
#pragma unroll
                for (int q = 0; q < NSA; ++q) o.s[q] = one ? 0.0 : r[16 + 4 * q + j4];
                return o;
            };
            auto scaled = [&](SgOps<NSA> o, double v) {
                o.t *= v;
#pragma unroll
                for (int q = 0; q < NSA; ++q) o.s[q] *= v;
                return o;
            };
            auto mac = [&](SgEdge<NSA> &P, const SgOps<NSA> &A, const SgOps<NSA> &B) {
                P.t = mfma16(A.t, B.t, P.t);
#pragma unroll
                for (int q = 0; q < NSA; ++q) P.sb[q] = sg_mfma4(A.t, B.s[q], P.sb[q]);
#pragma unroll
                for (int q = 0; q < NSA; ++q) P.sa[q] = sg_mfma4(A.s[q], B.t, P.sa[q]);
#pragma unroll
                for (int q = 0; q < NSA; ++q) P.c[q] = sg_mfma4(A.s[q], B.b1, P.c[q]);
            };
            auto kblock = [&](int b, double v, double vpsi, auto with_om, auto cleft) {
                const SgOps<NSA> f0 = ld(q0, se0, one0, b), f1 = ld(q1, se1, one1, b);
                if constexpr (decltype(with_om)::value) {
                    const SgOps<NSA> f2 = ld(q2, se2, false, b);
                    if constexpr (decltype(cleft)::value) mac(eO, scaled(f2, v), f1);
                    else mac(eO, scaled(f0, v), f2);
                }
                mac(eP, scaled(f0, vpsi), f1);
            };
            auto run = [&](auto with_om, auto cleft) {
                if (tile_one_slice) {
#pragma unroll 2
                    for (int b = 0; b < T / 4; ++b) {
                        const double v = rv[4 * b + kq];
                        kblock(b, v, v, with_om, cleft);
                    }
                } else {
#pragma unroll 1
                    for (int b = 0; b < T / 4; ++b) {
                        const int e = 4 * b + kq;
                        const int je = rj[e];
                        const bool ok = je >= 0;
                        const double v = rv[e];
                        if (__ballot(ok && je != cur) == 0ull) {
                            kblock(b, v, v, with_om, cleft);
                        } else {
                            kblock(b, v, 0.0, with_om, cleft);            // Omega of the k-block (and nothing into Psi), then Psi nonzero by nonzero
                            for (int qq = 0; qq < 4; ++qq) {
                                const int okq = __shfl((int)ok, 16 * qq);
                                const int jq = __shfl(je, 16 * qq);
                                if (!okq) continue;
                                if (jq != cur) {
                                    flush(cur, false);
                                    cur = jq;
                                }
                                kblock(b, 0.0, kq == qq ? v : 0.0, std::false_type{}, cleft);
                            }
                        }
                    }
                }
            };
            if (!a.has_om) run(std::false_type{}, std::false_type{});
            else if (a.c_left) run(std::true_type{}, std::true_type{});
            else run(std::true_type{}, std::false_type{});
        } else if (tile_one_slice) {
            // every nonzero of the tile belongs to the running slice (all but one tile in ~10^3 at C4): no slice test per
            // k-block, the validity bits from the ballot of stage (1), nothing but loads and matrix instructions in the loop
            constexpr int SG_UNR = NT == 1 ? 8 : 2;
            auto run = [&](auto with_om, auto all_tiles) {
                constexpr bool every = NT == 1 || decltype(all_tiles)::value;
#pragma unroll SG_UNR
                for (int b = 0; b < T / 4; ++b) {
                    const int e = 4 * b + kq;
                    double av[NT], bv[NT], cv[NT];
                    operands(b, rv[e], av, bv, cv);
#pragma unroll
                    for (int ta = 0; ta < NT; ++ta)
#pragma unroll
                        for (int tb = 0; tb < NT; ++tb) {
                            // (a tile wholly beyond a factor's width -- a factor of <= 16 columns beside a wider one -- is skipped)
                            if constexpr (decltype(with_om)::value)
                                if (every || (16 * ta < wOl && 16 * tb < wOr))
                                    accO[ta][tb] = mfma16(a.c_left ? cv[ta] * rv[e] : av[ta], a.c_left ? bv[tb] : cv[tb], accO[ta][tb]);
                            if (every || (16 * ta < wA && 16 * tb < wB)) accP[ta][tb] = mfma16(av[ta], bv[tb], accP[ta][tb]);
                        }
                }
            };
            // (NT = 2 with every factor beyond 16 columns: no tile to skip, no tests in the loop)
            const bool full = NT == 1 || (wA > 16 && wB > 16 && (!a.has_om || (wOl > 16 && wOr > 16)));
            if (a.has_om) { if (full) run(std::true_type{}, std::true_type{}); else run(std::true_type{}, std::false_type{}); }
            else { if (full) run(std::false_type{}, std::true_type{}); else run(std::false_type{}, std::false_type{}); }
        } else {
#pragma unroll 1
            for (int b = 0; b < T / 4; ++b) {
                const int e = 4 * b + kq;
                const int je = rj[e];
                const bool ok = je >= 0;
                const double v = rv[e];
                double av[NT], bv[NT], cv[NT];
                operands(b, v, av, bv, cv);
                if (a.has_om) {
#pragma unroll
                    for (int ta = 0; ta < NT; ++ta)
#pragma unroll
                        for (int tb = 0; tb < NT; ++tb)
                            accO[ta][tb] = a.c_left ? mfma16(cv[ta] * v, bv[tb], accO[ta][tb]) : mfma16(av[ta], cv[tb], accO[ta][tb]);
                }
                if (__ballot(ok && je != cur) == 0ull) {
#pragma unroll
                    for (int ta = 0; ta < NT; ++ta)
#pragma unroll
                        for (int tb = 0; tb < NT; ++tb) accP[ta][tb] = mfma16(av[ta], bv[tb], accP[ta][tb]);
                } else {
                    for (int qq = 0; qq < 4; ++qq) {
                        const int okq = __shfl((int)ok, 16 * qq);
                        const int jq = __shfl(je, 16 * qq);
                        if (!okq) continue;
                        if (jq != cur) {
                            flush(cur, false);
                            cur = jq;
                        }
#pragma unroll
                        for (int ta = 0; ta < NT; ++ta)
#pragma unroll
                            for (int tb = 0; tb < NT; ++tb) accP[ta][tb] = mfma16(kq == qq ? av[ta] : 0.0, bv[tb], accP[ta][tb]);
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    flush(cur, true);
    if (a.has_om) {
        double *dst = a.part_om + w_id * (size_t)(wOl * wOr);
        if constexpr (NS > 0) {
            edge_store(eO, dst, wOr, wOl, wOr);
            return;
        }
#pragma unroll
        for (int ta = 0; ta < NT; ++ta)
#pragma unroll
            for (int tb = 0; tb < NT; ++tb)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int aa = 16 * ta + 4 * t + kq, cc = 16 * tb + x16;
                    if (aa < wOl && cc < wOr) dst[aa * wOr + cc] = accO[ta][tb][t];
                }
    }
}

// Psi[:, j, :] += the partial blocks of slice j, waves in ascending order (first-slice partials, then last-slice
// partials): one workgroup per slice, one thread per (a, c); the waves of a slice by bisection (part_j is sorted).
__global__ __launch_bounds__(256) void sg_psi_reduce_kernel(const double *__restrict__ part, const int *__restrict__ pj, int waves,
                                                            int wA, int wB, int64_t n, double *__restrict__ psi)
{
    const int cells = wA * wB;
    for (int64_t j = blockIdx.x; j < n; j += gridDim.x) {
        auto lower = [&](int col, int64_t key) {           // first wave with pj[w][col] >= key
            int lo = 0, hi = waves;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (pj[mid * 3 + col] < key) lo = mid + 1; else hi = mid; }
            return lo;
        };
        const int f0 = lower(0, j), f1 = lower(0, j + 1), l0 = lower(1, j), l1 = lower(1, j + 1);
        if (f0 == f1 && l0 == l1) continue;
        for (int cell = threadIdx.x; cell < cells; cell += 256) {
            double acc = 0.0;
            for (int w = f0; w < f1; ++w) acc += part[((size_t)w * 2) * cells + cell];
            for (int w = l0; w < l1; ++w)
                if (pj[w * 3 + 2]) acc += part[((size_t)w * 2 + 1) * cells + cell];
            const int aa = cell / wB, c = cell - aa * wB;
            psi[((size_t)aa * n + j) * wB + c] += acc;
        }
    }
}

// out[t] += sum_w part[w][t] in wave order
__global__ __launch_bounds__(256) void sg_om_reduce_kernel(const double *__restrict__ part, int waves, int cells, double *__restrict__ out)
{
    const int t = blockIdx.x;
    double acc = 0.0;
    for (int w = threadIdx.x; w < waves; w += 256) acc += part[(size_t)w * cells + t];
    __shared__ double red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[t] += red[0];
}

// the resident stream of one mode: record pos = nonzero perm[pos]
template <typename FLAT>
__global__ void sg_stream_kernel(const int64_t *__restrict__ idx, IndexMap lm, IndexMap rm, int64_t mode_off,
                                 const int64_t *__restrict__ perm, const double *__restrict__ val, size_t N, FLAT *__restrict__ fl,
                                 FLAT *__restrict__ fr, int32_t *__restrict__ jj, double *__restrict__ vv)
{
    for (size_t pos = (size_t)blockIdx.x * blockDim.x + threadIdx.x; pos < N; pos += (size_t)gridDim.x * blockDim.x) {
        const size_t e = perm ? (size_t)perm[pos] : pos;
        fl[pos] = (FLAT)(lm.m ? flat_index(idx, lm, e) : 0);
        fr[pos] = (FLAT)(rm.m ? flat_index(idx, rm, e) : 0);
        jj[pos] = (int32_t)idx[mode_off + (int64_t)e];
        vv[pos] = val[e];
    }
}

// sort key of the mode order: (mode index, low 40 bits of the suffix flat index) -- slices in ascending order and,
// inside a slice, the nonzeros in the order of the suffix they share: the rows of a right-hand DRM table are then
// visited in ascending order within every slice (each line fetched once per slice instead of once per nonzero)
__global__ void sg_key_kernel(const int64_t *__restrict__ idx, IndexMap rm, int64_t mode_off, size_t N, uint64_t *__restrict__ keys,
                              int64_t *__restrict__ iota)
{
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < N; e += (size_t)gridDim.x * blockDim.x) {
        const uint64_t suffix = rm.m ? flat_index(idx, rm, e) : 0;
        keys[e] = ((uint64_t)idx[mode_off + (int64_t)e] << 40) | (suffix & ((1ull << 40) - 1));
        iota[e] = (int64_t)e;
    }
}

}  // namespace ttsk

using namespace ttsk;

extern "C" {

int ttsk_sparse_flat_mult(const uint64_t *shape, int m, uint64_t *mult_out)
{
    TTSK_ARG(shape && mult_out, "ttsk_sparse_flat_mult: NULL argument");
    IndexMap im;
    int rc = make_index_map(shape, m, 0, nullptr, &im);
    if (rc) return rc;
    for (int i = 0; i < m; ++i) mult_out[i] = im.mult[i];
    return TTSK_OK;
}

// perm: the nonzeros in ascending (mode index, suffix flat index) order.  The radix sort is the library's
// (hipcub::DeviceRadixSort, stable): once per tensor and mode, off the per-sketch path.
int ttsk_sparse_mode_order(const int64_t *dev_idx, int64_t row_stride, size_t N, const int *r_rows, const uint64_t *r_shape, int r_m,
                           int mode_row, int64_t n, int64_t *dev_perm, int stream)
{
    TTSK_STREAM(st, stream);
    TTSK_ARG(dev_idx && dev_perm && r_m >= 0 && mode_row >= 0 && n >= 1, "ttsk_sparse_mode_order: bad argument");
    TTSK_ARG(N < (1ull << 31), "ttsk_sparse_mode_order: more than 2^31 nonzeros");
    // the key holds the mode index in its upper 24 bits (sg_key_kernel): a longer mode would leave index bits out of the
    // sort and the stream out of slice order, which the passes rely on -- refuse it, the panel path takes over
    if (n > (1ll << 24)) {
        set_error("ttsk_sparse_mode_order: mode of %lld entries, the sort key holds 2^24", (long long)n);
        return TTSK_ERR_UNSUPPORTED;
    }
    if (N == 0) return TTSK_OK;
    IndexMap rm{};
    int rc;
    if (r_m && (rc = make_index_map(r_shape, r_m, row_stride, r_rows, &rm))) return rc;
    int bits = 1;
    while ((1ll << bits) < n && bits < 24) ++bits;
    size_t temp_bytes = 0;
    TTSK_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, (const uint64_t *)nullptr, (uint64_t *)nullptr,
                                                (const int64_t *)nullptr, (int64_t *)nullptr, (int)N, 0, 40 + bits, st));
    char *ws = (char *)scratch(stream, SCRATCH_MISC, 3 * N * 8 + temp_bytes + 256);
    if (!ws) return TTSK_ERR_HIP;
    uint64_t *keys = (uint64_t *)ws, *keys_out = keys + N;
    int64_t *iota = (int64_t *)(keys_out + N);
    void *temp = ws + 3 * N * 8;
    size_t blocks = (N + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(sg_key_kernel, dim3((unsigned)blocks), dim3(256), 0, st, dev_idx, rm, (int64_t)mode_row * row_stride, N, keys, iota);
    TTSK_LAUNCH_CHECK();
    TTSK_HIP(hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys, keys_out, iota, dev_perm, (int)N, 0, 40 + bits, st));
    return TTSK_OK;
}

static int sg_mode_stream(const int64_t *dev_idx, int64_t row_stride, const int64_t *dev_perm, size_t N, const int *l_rows,
                          const uint64_t *l_shape, int l_m, const int *r_rows, const uint64_t *r_shape, int r_m, int mode_row,
                          const double *dev_val, void *dev_fl, void *dev_fr, int w32, int32_t *dev_j, double *dev_v, int stream)
{
    TTSK_STREAM(st, stream);
    TTSK_ARG(dev_idx && dev_val && dev_fl && dev_fr && dev_j && dev_v, "ttsk_sparse_mode_stream: NULL argument");
    TTSK_ARG(l_m >= 0 && r_m >= 0 && mode_row >= 0, "ttsk_sparse_mode_stream: bad argument");
    IndexMap lm{}, rm{};
    int rc;
    if (l_m && (rc = make_index_map(l_shape, l_m, row_stride, l_rows, &lm))) return rc;
    if (r_m && (rc = make_index_map(r_shape, r_m, row_stride, r_rows, &rm))) return rc;
    if (N == 0) return TTSK_OK;
    size_t blocks = (N + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    if (w32) {
        // 32-bit records only where every flat index fits (no wrap of the reference's 32-bit running product either)
        double pl = 1.0, pr = 1.0;
        for (int i = 0; i < l_m; ++i) pl *= (double)l_shape[i];
        for (int i = 0; i < r_m; ++i) pr *= (double)r_shape[i];
        if (!(pl < 2147483648.0) || !(pr < 2147483648.0)) {
            set_error("ttsk_sparse_mode_stream_u32: %g prefixes / %g suffixes do not fit 32-bit records", pl, pr);
            return TTSK_ERR_UNSUPPORTED;
        }
        hipLaunchKernelGGL(sg_stream_kernel<uint32_t>, dim3((unsigned)blocks), dim3(256), 0, st, dev_idx, lm, rm, (int64_t)mode_row * row_stride,
                           dev_perm, dev_val, N, (uint32_t *)dev_fl, (uint32_t *)dev_fr, dev_j, dev_v);
    } else {
        hipLaunchKernelGGL(sg_stream_kernel<uint64_t>, dim3((unsigned)blocks), dim3(256), 0, st, dev_idx, lm, rm, (int64_t)mode_row * row_stride,
                           dev_perm, dev_val, N, (uint64_t *)dev_fl, (uint64_t *)dev_fr, dev_j, dev_v);
    }
    TTSK_LAUNCH_CHECK();
    return TTSK_OK;
}

int ttsk_sparse_mode_stream(const int64_t *dev_idx, int64_t row_stride, const int64_t *dev_perm, size_t N, const int *l_rows,
                            const uint64_t *l_shape, int l_m, const int *r_rows, const uint64_t *r_shape, int r_m, int mode_row,
                            const double *dev_val, uint64_t *dev_fl, uint64_t *dev_fr, int32_t *dev_j, double *dev_v, int stream)
{
    return sg_mode_stream(dev_idx, row_stride, dev_perm, N, l_rows, l_shape, l_m, r_rows, r_shape, r_m, mode_row, dev_val, dev_fl, dev_fr, 0,
                          dev_j, dev_v, stream);
}

int ttsk_sparse_mode_stream_u32(const int64_t *dev_idx, int64_t row_stride, const int64_t *dev_perm, size_t N, const int *l_rows,
                                const uint64_t *l_shape, int l_m, const int *r_rows, const uint64_t *r_shape, int r_m, int mode_row,
                                const double *dev_val, uint32_t *dev_fl, uint32_t *dev_fr, int32_t *dev_j, double *dev_v, int stream)
{
    return sg_mode_stream(dev_idx, row_stride, dev_perm, N, l_rows, l_shape, l_m, r_rows, r_shape, r_m, mode_row, dev_val, dev_fl, dev_fr, 1,
                          dev_j, dev_v, stream);
}

static int sg_gauss_pass(const void *dev_fl, const void *dev_fr, int w32, const int32_t *dev_j, const double *dev_val, size_t N,
                         int64_t n, const ttsk_sg_factor *A, const ttsk_sg_factor *B, const ttsk_sg_factor *C, int c_left,
                         double *dev_psi, double *dev_omega, int stream);

int ttsk_sparse_gauss_pass(const uint64_t *dev_fl, const uint64_t *dev_fr, const int32_t *dev_j, const double *dev_val, size_t N,
                           int64_t n, const ttsk_sg_factor *A, const ttsk_sg_factor *B, const ttsk_sg_factor *C, int c_left,
                           double *dev_psi, double *dev_omega, int stream)
{
    return sg_gauss_pass(dev_fl, dev_fr, 0, dev_j, dev_val, N, n, A, B, C, c_left, dev_psi, dev_omega, stream);
}

int ttsk_sparse_gauss_pass_u32(const uint32_t *dev_fl, const uint32_t *dev_fr, const int32_t *dev_j, const double *dev_val, size_t N,
                               int64_t n, const ttsk_sg_factor *A, const ttsk_sg_factor *B, const ttsk_sg_factor *C, int c_left,
                               double *dev_psi, double *dev_omega, int stream)
{
    return sg_gauss_pass(dev_fl, dev_fr, 1, dev_j, dev_val, N, n, A, B, C, c_left, dev_psi, dev_omega, stream);
}

static int sg_gauss_pass(const void *dev_fl, const void *dev_fr, int w32, const int32_t *dev_j, const double *dev_val, size_t N,
                         int64_t n, const ttsk_sg_factor *A, const ttsk_sg_factor *B, const ttsk_sg_factor *C, int c_left,
                         double *dev_psi, double *dev_omega, int stream)
{
    TTSK_STREAM(st, stream);
    TTSK_ARG(dev_val && dev_psi && n >= 1, "ttsk_sparse_gauss_pass: NULL argument");
    TTSK_ARG(dev_j || n == 1, "ttsk_sparse_gauss_pass: a NULL mode index means a single slice");
    TTSK_ARG(!C || dev_omega, "ttsk_sparse_gauss_pass: an Omega factor needs an output");
    if (N == 0) return TTSK_OK;
    SgPass a{};
    a.fl = (const uint64_t *)dev_fl; a.fr = (const uint64_t *)dev_fr; a.w32 = w32; a.jj = dev_j; a.val = dev_val; a.N = N; a.n = n;
    const ttsk_sg_factor *fs[3] = {A, B, C};
    int cols = 0, widest = 1;
    for (int i = 0; i < 3; ++i) {
        SgF &F = a.f[i];
        if (!fs[i]) { F.kind = 0; F.w = 1; continue; }
        F.kind = fs[i]->kind; F.w = fs[i]->kind ? fs[i]->w : 1; F.rank_min = fs[i]->rank_min; F.src = fs[i]->src;
        F.mul = fs[i]->mul; F.seed = fs[i]->seed; F.table = fs[i]->table;
        F.full = fs[i]->full; F.nnz = fs[i]->nnz;
        TTSK_ARG(F.kind >= 0 && F.kind <= 3 && F.w >= 1 && F.w <= 32, "ttsk_sparse_gauss_pass: factor %d: kind %d, width %d", i, F.kind, F.w);
        TTSK_ARG(F.kind != 1 || F.table, "ttsk_sparse_gauss_pass: table factor without a table");
        TTSK_ARG(F.kind != 3 || (F.full >= 1 && F.full <= 32 && F.nnz >= 0 && F.nnz <= F.full && F.rank_min >= 0 && F.rank_min + F.w <= F.full),
                 "ttsk_sparse_gauss_pass: factor %d: sign row of %d entries, %d non-zero, columns [%d, %d)", i, F.full, F.nnz, F.rank_min,
                 F.rank_min + F.w);
        TTSK_ARG(!((F.src & 1) ? !dev_fr : !dev_fl) || F.kind == 0, "ttsk_sparse_gauss_pass: factor %d needs a flat index stream", i);
        if (F.kind == 3) {                             // the whole row is staged; the products read its slice
            a.off[i] = cols + F.rank_min;
            cols += F.full;
            if (F.full > widest) widest = F.full;
        } else if (F.kind == 2) {
            a.off[i] = cols;
            cols += F.w;
            a.qcols += F.w;
        } else if (F.kind == 1) {                      // a block of its own behind the tile
            F.units = (F.w + 1) / 2;
            F.rcp = 65536 / F.units + 1;              // i / units == (i * rcp) >> 16 for i < 1024, units <= 16
            a.off[i] = a.tab;                         // in doubles per nonzero of the tile here; times the tile size below
            a.tab += 2 * F.units;
        }
        if (F.w > widest) widest = F.w;
    }
    const int NT = widest > 16 ? 2 : 1;
    int wmax = 1;
    for (int i = 0; i < 3; ++i) wmax = a.f[i].w > wmax ? a.f[i].w : wmax;
    static const int edges_on = [] { const char *e = getenv("TTSK_SG_EDGES"); return e ? atoi(e) : 1; }();
    const int NS = (NT == 2 && edges_on && wmax > 16 && wmax <= 24) ? (wmax <= 20 ? 1 : 2) : 0;   // strips beyond the first 16 columns
#ifdef TTSK_LAB
    { const char *e = getenv("TTSK_SG_LAB"); a.lab = e ? atoi(e) : 0; }
#endif
    a.has_om = C != nullptr;
    a.c_left = c_left;
    a.tcols = cols > 0 ? cols : 1;
    a.psi = dev_psi;
    // waves: what is resident at once (NT = 1: 3 workgroups of 4 waves per CU -- 141 VGPRs, <= 53 KB of LDS each; NT = 2: two
    // by the registers, fewer by the LDS of wide tiles), so that the grid is one even round; stretches of whole tiles
    static const size_t n_cu = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        return (size_t)v;
    }();
    // tile of 32 nonzeros; of 16 where 32 would leave LDS for one wide workgroup per CU only
    static const int t16_on = [] { const char *e = getenv("TTSK_SG_T16"); return e ? atoi(e) : 1; }();
    const size_t fixed = 16 * NT * 24 + 64;
    int T = SG_T;
    if (NT == 2 && t16_on && (size_t)(156 * 1024) / (sg_per_wave(a.tcols, SG_T * a.tab, a.qcols, SG_T) * 32 + fixed) < 2) T = 16;
    for (int i = 0; i < 3; ++i)
        if (a.f[i].kind == 1) a.off[i] *= T;
    a.tab *= T;
    const size_t per_wave = sg_per_wave(a.tcols, a.tab, a.qcols, T);
    const size_t lds = per_wave * 4 * 8;
    size_t wg_per_cu = (size_t)(156 * 1024) / (lds + fixed);
    if (wg_per_cu > (NT == 1 ? 3u : 2u)) wg_per_cu = NT == 1 ? 3 : 2;
    if (wg_per_cu < 1) {
        set_error("ttsk_sparse_gauss_pass: a staged tile of %d columns does not fit the LDS", a.tcols);
        return TTSK_ERR_UNSUPPORTED;
    }
    const size_t resident = n_cu * 4 * wg_per_cu;
    size_t waves = resident;
    size_t chunk = ((N + waves - 1) / waves + SG_T - 1) / SG_T * SG_T;
    if (chunk < 8 * SG_T) chunk = 8 * SG_T;
    waves = (N + chunk - 1) / chunk;
    const size_t blocks = (waves + 3) / 4;
    const size_t wtot = blocks * 4;
    a.chunk = chunk;
    const int wA = a.f[0].w, wB = a.f[1].w;
    const int cellsP = wA * wB, cellsO = a.has_om ? (c_left ? a.f[2].w * wB : wA * a.f[2].w) : 0;
    char *ws = (char *)scratch(stream, SCRATCH_MISC, wtot * ((size_t)(2 * cellsP + cellsO) * 8 + 16) + 256);
    if (!ws) return TTSK_ERR_HIP;
    a.part_psi = (double *)ws;
    a.part_om = a.part_psi + wtot * 2 * cellsP;
    a.part_j = (int *)(a.part_om + wtot * (size_t)cellsO);
    static PerInit attr;
    if (attr.first()) {
        TTSK_HIP(hipFuncSetAttribute((const void *)sg_pass_kernel<1, 0, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        TTSK_HIP(hipFuncSetAttribute((const void *)sg_pass_kernel<2, 0, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
        TTSK_HIP(hipFuncSetAttribute((const void *)sg_pass_kernel<2, 1, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
        TTSK_HIP(hipFuncSetAttribute((const void *)sg_pass_kernel<2, 2, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
        TTSK_HIP(hipFuncSetAttribute((const void *)sg_pass_kernel<2, 0, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
        TTSK_HIP(hipFuncSetAttribute((const void *)sg_pass_kernel<2, 1, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
        TTSK_HIP(hipFuncSetAttribute((const void *)sg_pass_kernel<2, 2, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
    }
    const bool prof = prof_on();
    if (prof) prof_open_named(st, PROF_SPARSE, (w32 ? 20.0 : 28.0) * (double)N, "sg_pass_kernel");
    const dim3 grid((unsigned)blocks), wg(256);
    if (NT == 1) hipLaunchKernelGGL((sg_pass_kernel<1, 0, 32>), grid, wg, lds, st, a);
    else if (T == 32) {
        if (NS == 1) hipLaunchKernelGGL((sg_pass_kernel<2, 1, 32>), grid, wg, lds, st, a);
        else if (NS == 2) hipLaunchKernelGGL((sg_pass_kernel<2, 2, 32>), grid, wg, lds, st, a);
        else hipLaunchKernelGGL((sg_pass_kernel<2, 0, 32>), grid, wg, lds, st, a);
    } else {
        if (NS == 1) hipLaunchKernelGGL((sg_pass_kernel<2, 1, 16>), grid, wg, lds, st, a);
        else if (NS == 2) hipLaunchKernelGGL((sg_pass_kernel<2, 2, 16>), grid, wg, lds, st, a);
        else hipLaunchKernelGGL((sg_pass_kernel<2, 0, 16>), grid, wg, lds, st, a);
    }
    TTSK_LAUNCH_CHECK();
    const int64_t rb = n < 4096 ? n : 4096;
    hipLaunchKernelGGL(sg_psi_reduce_kernel, dim3((unsigned)rb), dim3(256), 0, st, a.part_psi, a.part_j, (int)wtot, wA, wB, n, dev_psi);
    TTSK_LAUNCH_CHECK();
    if (a.has_om) {
        hipLaunchKernelGGL(sg_om_reduce_kernel, dim3((unsigned)cellsO), dim3(256), 0, st, a.part_om, (int)wtot, cellsO, dev_omega);
        TTSK_LAUNCH_CHECK();
    }
    if (prof) prof_close(st);
    return TTSK_OK;
}

}  // extern "C"

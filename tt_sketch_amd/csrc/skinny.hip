// Host side of the barrier-free chain kernels (skinny.h): shape tests and launch geometry.
#include <cstdlib>
#include "skinny.h"

namespace ttsk {

template <int D, int STR>
int launch_skinny_s_depth(const SkinnyS &a, int npt, int spt, size_t lds_bytes, int grid, hipStream_t st);
#define TTSK_S_EXT(D, STR) extern template int launch_skinny_s_depth<D, STR>(const SkinnyS &, int, int, size_t, int, hipStream_t)
TTSK_S_EXT(4, 0); TTSK_S_EXT(4, 1); TTSK_S_EXT(4, 2); TTSK_S_EXT(5, 0); TTSK_S_EXT(5, 1); TTSK_S_EXT(5, 2);
#undef TTSK_S_EXT

int launch_skinny_r_0(const SkinnyR &a, int nmt, int nnt, int grid, hipStream_t st);
int launch_skinny_r_1(const SkinnyR &a, int nmt, int nnt, int grid, hipStream_t st);
int launch_skinny_r_2(const SkinnyR &a, int nmt, int nnt, int grid, hipStream_t st);
int launch_skinny_r_3(const SkinnyR &a, int nmt, int nnt, int grid, hipStream_t st);

// out[m, n] (+)= alpha sum_c slab[c][m][n] per (problem, row tile) y; 16 chunk lanes x 16 consecutive n
__global__ __launch_bounds__(256) void skinny_r_reduce(const double *__restrict__ slab_all, int chunks, int M, int N,
                                                       int m_tiles, int64_t Mtot, ReduceOut outs, int64_t c_m,
                                                       int64_t c_n, double alpha, int accumulate)
{
    __shared__ double part[16][17];
    const int prob = blockIdx.y / m_tiles, tile = blockIdx.y - prob * m_tiles;
    const double *__restrict__ slab = slab_all + (int64_t)blockIdx.y * chunks * M * N;
    double *__restrict__ C = outs.C[prob];
    const int x = threadIdx.x & 15, z = threadIdx.x >> 4;
    const int64_t e = (int64_t)blockIdx.x * 16 + x, MN = (int64_t)M * N;
    double sum = 0.0;
    if (e < MN) {
        double v[8];
        for (int c0 = z; c0 < chunks; c0 += 16 * 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = c0 + 16 * u < chunks ? slab[(int64_t)(c0 + 16 * u) * MN + e] : 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u) sum += v[u];
        }
    }
    part[z][x] = sum;
    __syncthreads();
    if (z == 0 && e < MN) {
        double tot = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) tot += part[k][x];
        const int m = (int)(e / N), n = (int)(e - (int64_t)m * N);
        const int64_t row = (int64_t)tile * M + m;
        if (row < Mtot) {
            double *c = C + row * c_m + n * c_n;
            *c = accumulate ? *c + alpha * tot : alpha * tot;
        }
    }
}

// the same, two consecutive n per lane (16-byte loads and stores): N, c_m even, c_n = 1, 16-byte aligned bases
__global__ __launch_bounds__(256) void skinny_r_reduce2(const double2 *__restrict__ slab_all, int chunks, int M, int N,
                                                        int m_tiles, int64_t Mtot, ReduceOut outs, int64_t c_m, double alpha,
                                                        int accumulate)
{
    __shared__ double2 part[16][17];
    const int prob = blockIdx.y / m_tiles, tile = blockIdx.y - prob * m_tiles;
    const int64_t MN2 = (int64_t)M * N / 2;
    const double2 *__restrict__ slab = slab_all + (int64_t)blockIdx.y * chunks * MN2;
    double *__restrict__ C = outs.C[prob];
    const int x = threadIdx.x & 15, z = threadIdx.x >> 4;
    const int64_t e = (int64_t)blockIdx.x * 16 + x;
    double2 sum = make_double2(0.0, 0.0);
    if (e < MN2) {
        double2 v[8];
        for (int c0 = z; c0 < chunks; c0 += 16 * 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = c0 + 16 * u < chunks ? slab[(int64_t)(c0 + 16 * u) * MN2 + e] : make_double2(0.0, 0.0);
#pragma unroll
            for (int u = 0; u < 8; ++u) { sum.x += v[u].x; sum.y += v[u].y; }
        }
    }
    part[z][x] = sum;
    __syncthreads();
    if (z == 0 && e < MN2) {
        double2 tot = make_double2(0.0, 0.0);
#pragma unroll
        for (int k = 0; k < 16; ++k) { tot.x += part[k][x].x; tot.y += part[k][x].y; }
        const int m = (int)(2 * e / N), n = (int)(2 * e - (int64_t)m * N);
        const int64_t row = (int64_t)tile * M + m;
        if (row < Mtot) {
            double2 *c = reinterpret_cast<double2 *>(C + row * c_m + n);
            double2 o = make_double2(alpha * tot.x, alpha * tot.y);
            if (accumulate) { const double2 old = *c; o.x += old.x; o.y += old.y; }
            *c = o;
        }
    }
}

// many chunks of one small result (the fused chain step of a single tensor: one partial per mode index): 64 element
// pairs x 16 chunk groups per block -- a wave reads 1 KB of one chunk per instruction instead of four 256-byte pieces
__global__ __launch_bounds__(1024) void skinny_r_reduce2w(const double2 *__restrict__ slab_all, int chunks, int M, int N,
                                                          int m_tiles, int64_t Mtot, ReduceOut outs, int64_t c_m, double alpha,
                                                          int accumulate)
{
    __shared__ double2 part[16][65];
    const int prob = blockIdx.y / m_tiles, tile = blockIdx.y - prob * m_tiles;
    const int64_t MN2 = (int64_t)M * N / 2;
    const double2 *__restrict__ slab = slab_all + (int64_t)blockIdx.y * chunks * MN2;
    double *__restrict__ C = outs.C[prob];
    const int x = threadIdx.x & 63, z = threadIdx.x >> 6;
    const int64_t e = (int64_t)blockIdx.x * 64 + x;
    double2 sum = make_double2(0.0, 0.0);
    if (e < MN2) {
        double2 v[8];
        for (int c0 = z; c0 < chunks; c0 += 16 * 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = c0 + 16 * u < chunks ? slab[(int64_t)(c0 + 16 * u) * MN2 + e] : make_double2(0.0, 0.0);
#pragma unroll
            for (int u = 0; u < 8; ++u) { sum.x += v[u].x; sum.y += v[u].y; }
        }
    }
    part[z][x] = sum;
    __syncthreads();
    if (z == 0 && e < MN2) {
        double2 tot = make_double2(0.0, 0.0);
#pragma unroll
        for (int k = 0; k < 16; ++k) { tot.x += part[k][x].x; tot.y += part[k][x].y; }
        const int m = (int)(2 * e / N), n = (int)(2 * e - (int64_t)m * N);
        const int64_t row = (int64_t)tile * M + m;
        if (row < Mtot) {
            double2 *c = reinterpret_cast<double2 *>(C + row * c_m + n);
            double2 o = make_double2(alpha * tot.x, alpha * tot.y);
            if (accumulate) { const double2 old = *c; o.x += old.x; o.y += old.y; }
            *c = o;
        }
    }
}

int launch_r_reduce(hipStream_t st, const double *slab, int chunks, int M, int N, int m_tiles, int64_t Mtot, const ReduceOut &ro,
                    int nprob, int64_t c_m, int64_t c_n, double alpha, int accumulate)
{
    const int64_t mn = (int64_t)M * N;
    bool vec = c_n == 1 && !(N & 1) && !(c_m & 1) && !((uintptr_t)slab & 15);
    for (int b = 0; b < nprob && vec; ++b) vec = !((uintptr_t)ro.C[b] & 15);
    const unsigned gy = (unsigned)(nprob * m_tiles);
    static const int wide = [] { const char *e = getenv("TTSK_REDUCE_WIDE"); return e ? atoi(e) : 1; }();
    if (vec && wide && chunks >= 64 && gy * cdiv(mn / 2, 64) >= 64)
        hipLaunchKernelGGL(skinny_r_reduce2w, dim3((unsigned)cdiv(mn / 2, 64), gy), dim3(1024), 0, st, (const double2 *)slab, chunks, M,
                           N, m_tiles, Mtot, ro, c_m, alpha, accumulate);
    else if (vec)
        hipLaunchKernelGGL(skinny_r_reduce2, dim3((unsigned)cdiv(mn / 2, 16), gy), dim3(256), 0, st, (const double2 *)slab, chunks, M, N,
                           m_tiles, Mtot, ro, c_m, alpha, accumulate);
    else
        hipLaunchKernelGGL(skinny_r_reduce, dim3((unsigned)cdiv(mn, 16), gy), dim3(256), 0, st, slab, chunks, M, N, m_tiles, Mtot, ro,
                           c_m, c_n, alpha, accumulate);
    return hipGetLastError() == hipSuccess ? TTSK_OK : TTSK_ERR_HIP;
}

static int num_cu()
{
    static int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 256;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) return 256;
        return v;
    }();
    return n;
}

static int skinny_mode()
{
    static int m = [] {
        const char *e = getenv("TTSK_SKINNY");
        return e ? atoi(e) : 1;
    }();
    return m;
}

// C[m, j] = alpha sum_k W[k, m] S[j, k]
static int run_s(SkinnyS a, hipStream_t st, int *prof, double flops)
{
    const int npt = (int)cdiv(a.P, 16);
    const int kb = (a.K + 3) / 4;
    const int64_t nrb = a.U == 1 ? cdiv(a.J, 16) : cdiv(a.U, 16 >> a.tvl) * cdiv(a.V, 1 << a.tvl);
    static int wgdiv = [] { const char *e = getenv("TTSK_S_SPLIT"); return e ? atoi(e) : 1; }();
    // CUs one problem of the batch can count on (TTSK_S_SPLIT=0: every problem spreads over all CUs)
    const int cus = (wgdiv && num_cu() / a.nb > 0) ? num_cu() / a.nb : num_cu();
    // mode (see skinny_s_kernel): rounds of workgroups over the CUs x tile strips per SIMD and round.
    // Mode 2 loads no fragment twice and measured ~10 % faster at equal strip counts, so the others
    // have to beat it by more than that; ties between them go to the one without the shared block.
    static int mode_force = [] { const char *e = getenv("TTSK_S_MODE"); return e ? atoi(e) : -1; }();
    const int64_t g4 = cdiv(nrb, 4), g5 = cdiv(nrb, 5), g8 = cdiv(nrb, 8);
    const int64_t t4 = cdiv(g4, cus) * npt, t5 = cdiv(g5, cus) * (npt + (npt > 4 ? 2 : 1)), t8 = cdiv(g8, cus) * 2 * npt;
    int spt = 2;
    int64_t best = t8;
    if (t4 * 112 < best * 100) { best = t4 * 112 / 100; spt = 0; }
    if (t5 * 112 < best * 100 && t5 < t4) { best = t5 * 112 / 100; spt = 1; }
    if (mode_force >= 0 && mode_force <= 2) spt = mode_force;
    const int64_t groups = spt == 2 ? g8 : (spt == 1 ? g5 : g4);
    a.groups = (int)groups;
    const int ldw = ldmf(16 * npt);
    // ring depth 5 unless 4 pads K less (k-blocks are processed in multiples of the depth)
    // (a second, register-free prefetch level through `buffer_load ... lds`, a ring as deep as the
    // whole K and opposite k orders for the two waves of a SIMD were all measured slower)
    const int dring = cdiv(kb, 5) * 5 <= cdiv(kb, 4) * 4 ? 5 : 4;
    const size_t lds = (size_t)cdiv(kb, dring) * dring * 4 * ldw * 8;
    a.wpp = (int)(groups < cus ? groups : cus);
    const int grid = a.wpp * a.nb;
    // partial last tile of W with <= 8 valid columns: 4-wide strips instead of a padded 16x16x4 tile
    static int strips_on = [] { const char *e = getenv("TTSK_S_STRIPS"); return e ? atoi(e) : 1; }();
    const int rem = (int)(a.P % 16);
    const int str = (strips_on && rem > 0 && rem <= 8) ? (rem + 3) / 4 : 0;
    if (prof) prof_open(st, flops, 3, str * 1000 + npt * 100 + spt, str > 0, dring == 4);
#define TTSK_S_GO(D) (str == 2 ? launch_skinny_s_depth<D, 2>(a, npt, spt, lds, grid, st) \
                      : str == 1 ? launch_skinny_s_depth<D, 1>(a, npt, spt, lds, grid, st) \
                                 : launch_skinny_s_depth<D, 0>(a, npt, spt, lds, grid, st))
    if (dring == 4) return TTSK_S_GO(4);
    return TTSK_S_GO(5);
#undef TTSK_S_GO
}

// long-K: one side <= 128, the other <= 128 or cut into row tiles of 128; both operands contiguous
// along their non-contracted index, 16-byte loads possible (even extents / strides, aligned bases);
// desc not collapsed (two-level kappa ok when the operands stay below 4 GB)
static int try_r(const ttsk_gemm_desc &d, int nb, const double *const *A, const double *const *B, double *const *C,
                 int stream, hipStream_t st)
{
    const int64_t K = d.Ko * d.Ki;
    if ((d.M > 128 && d.N > 128) || K < 4096) return 0;
    if (d.Ko > 1 && d.Ki < 4) return 0;
    if (d.a_m < 0 || d.b_n < 0 || d.a_ko < 0 || d.a_ki < 0 || d.b_ko < 0 || d.b_ki < 0) return 0;
    // pairable = contiguous rows, even extent / strides, 16-byte aligned base; otherwise generic tiles
    bool a_pair = d.a_m == 1 && !((d.M | d.a_ko | d.a_ki) & 1), b_pair = d.b_n == 1 && !((d.N | d.b_ko | d.b_ki) & 1);
    for (int b = 0; b < nb; ++b) {
        if ((uintptr_t)A[b] & 15) a_pair = false;
        if ((uintptr_t)B[b] & 15) b_pair = false;
    }
    SkinnyR r{};
    r.nb = nb;
    // the operand with more 16-row tiles plays "A" (rows, tiled by 128 when longer), the other "B"
    const bool swap = cdiv(d.N, 16) > cdiv(d.M, 16);
    int64_t big, small_;
    if (!swap) {
        r.a_ko = d.a_ko; r.a_ki = d.a_ki; big = d.M; r.a_m = d.a_m; r.a_gen = !a_pair;
        r.b_ko = d.b_ko; r.b_ki = d.b_ki; small_ = d.N; r.b_n = d.b_n; r.b_gen = !b_pair;
    } else {
        r.a_ko = d.b_ko; r.a_ki = d.b_ki; big = d.N; r.a_m = d.b_n; r.a_gen = !b_pair;
        r.b_ko = d.a_ko; r.b_ki = d.a_ki; small_ = d.M; r.b_n = d.a_m; r.b_gen = !a_pair;
    }
    if (r.a_gen || r.b_gen) r.a_gen = r.b_gen = 1;     // one generic variant: both sides through plain tiles
    // ... or its 16-byte form when kappa is contiguous on both sides (single-level kappa, even strides / K,
    // 16-byte aligned bases)
    static int gk_on = [] { const char *e = getenv("TTSK_R_GK"); return e ? atoi(e) : 1; }();
    bool gk = gk_on && r.a_gen && d.Ko == 1 && r.a_ki == 1 && r.b_ki == 1 && !(K & 1) && !(r.a_m & 1) && !(r.b_n & 1);
    for (int b = 0; b < nb && gk; ++b)
        if (((uintptr_t)A[b] | (uintptr_t)B[b]) & 15) gk = false;
    // a lane's row offset inside a tile (15 rows) has to fit 32 bits next to the kappa walk
    if (15 * r.a_m * 8 >= (1ll << 31) || 15 * r.b_n * 8 >= (1ll << 31)) return 0;
    r.Mtot = big;
    r.M = (int)(big < 128 ? big : 128);
    r.m_tiles = (int)cdiv(big, 128);
    r.N = (int)small_;
    if ((int64_t)nb * r.m_tiles > 60000) return 0;
    for (int b = 0; b < nb; ++b) {
        r.A[b] = swap ? B[b] : A[b];
        r.B[b] = swap ? A[b] : B[b];
    }
    if (d.Ko == 1) { r.a_ko = 0; r.b_ko = 0; }
    r.Ki = d.Ki;
    r.K = K;
    if (d.Ko > 1 && r.a_ko == d.Ki * r.a_ki && r.b_ko == d.Ki * r.b_ki) r.Ki = K;   // uniform walk
    r.rebase = r.Ki == K ? 1 : 0;
    if (gk && r.rebase) r.a_gen = r.b_gen = 2;
    { const char *e = getenv("TTSK_SK_STAMPS"); r.stamps = e ? (long long *)strtoull(e, nullptr, 0) : nullptr; }
    r.a_extent = (big - 1) * r.a_m + (d.Ko - 1) * r.a_ko + (d.Ki - 1) * r.a_ki + 1;
    r.b_extent = (r.N - 1) * r.b_n + (d.Ko - 1) * r.b_ko + (d.Ki - 1) * r.b_ki + 1;
    const int cus = num_cu() / nb > 0 ? num_cu() / nb : 1;
    const int64_t want_chunks = cus / r.m_tiles > 0 ? cus / r.m_tiles : 1;
    r.chunk = cdiv(cdiv(K, want_chunks), 8) * 8;
    if (r.chunk < 64) r.chunk = 64;
    if (r.rebase) {
        // per-workgroup origins: 32-bit offsets only have to span one chunk and one row tile; shorter
        // chunks when a kappa row is so long (an unfolding with 2 MB rows) that a chunk would not fit
        auto span = [&](int64_t chunk) {
            const int64_t sa = ((r.a_gen ? 16 * r.a_m : 144) + (chunk + 64) * r.a_ki) * 8;
            const int64_t sb = ((r.b_gen ? 16 * r.b_n : 144) + (chunk + 64) * r.b_ki) * 8;
            return sa > sb ? sa : sb;
        };
        while (span(r.chunk) >= (1ll << 32) - 64 && r.chunk > 64) r.chunk = cdiv(r.chunk / 2, 8) * 8;
        if (span(r.chunk) >= (1ll << 32) - 64) return 0;
    }
    const int chunks = (int)cdiv(K, r.chunk);
    r.chunks = chunks;
    if ((int64_t)nb * r.m_tiles * chunks > (1 << 30) / 8) return 0;
    if (!r.rebase) {
        // 32-bit byte offsets including the kappa walk past the end of the last chunk
        const int64_t reach_a = ((r.a_gen ? 16 * r.a_m : big + 144) + (d.Ko + 1) * r.a_ko + (d.Ki + 64) * r.a_ki) * 8;
        const int64_t reach_b = ((r.b_gen ? 16 * r.b_n : 144) + (d.Ko + 1) * r.b_ko + (d.Ki + 64) * r.b_ki) * 8;
        if (reach_a >= (1ll << 32) - 64 || reach_b >= (1ll << 32) - 64) return 0;
    }
    // (one slab per XCD filled with L2-local fp64 atomics was measured 3x slower than slab + reduce)
    const int nmt = (int)cdiv(r.M, 16), nnt = (int)cdiv(r.N, 16);
    const int nsub = nmt * nnt <= SKR_KSPLIT_TILES ? 8 : 1;      // small outputs: one slab per wave
    const int64_t nslab = (int64_t)nb * r.m_tiles * chunks;
    r.slab = (double *)scratch(stream, SCRATCH_GEMM, (size_t)nslab * nsub * r.M * r.N * 8 + 64);
    if (!r.slab) return TTSK_ERR_HIP;
    const bool prof = prof_on();
    int rc;
    // (a chunk -> XCD mapping that makes all problems of a batch fetch the shared operand into one L2
    // halved the L2-fabric traffic of the batched GEMM2 but not its time)
    if (prof) prof_open(st, 2.0 * nb * (double)d.M * (double)d.N * (double)K, 4, nmt * 10 + nnt, false, false);
    if (nmt <= 4) rc = launch_skinny_r_0(r, nmt, nnt, (int)nslab, st);
    else if (nmt <= 6) rc = launch_skinny_r_1(r, nmt, nnt, (int)nslab, st);
    else if (nmt == 7) rc = launch_skinny_r_2(r, nmt, nnt, (int)nslab, st);
    else rc = launch_skinny_r_3(r, nmt, nnt, (int)nslab, st);
    if (prof) prof_close(st);
    if (rc != TTSK_OK) return rc;
    const int64_t mn = (int64_t)r.M * r.N;
    ReduceOut ro{};
    for (int b = 0; b < nb; ++b) ro.C[b] = C[b];
    (void)mn;
    if (launch_r_reduce(st, r.slab, chunks * nsub, r.M, r.N, r.m_tiles, r.Mtot, ro, nb, swap ? d.c_n : d.c_m, swap ? d.c_m : d.c_n,
                        d.alpha, d.accumulate) != TTSK_OK) {
        set_error("skinny_r_reduce launch failed");
        return TTSK_ERR_HIP;
    }
    return 1;
}

int skinny_try(const ttsk_gemm_desc &d, const double *A, const double *B, double *C, const double *k_scale,
               int stream, hipStream_t st)
{
    if (k_scale) return 0;
    return skinny_try_batch(d, 1, &A, &B, &C, stream, st);
}

int skinny_try_batch(const ttsk_gemm_desc &d, int nb, const double *const *A, const double *const *B,
                     double *const *C, int stream, hipStream_t st)
{
    if (!skinny_mode() || nb < 1 || nb > SK_MAXB) return 0;
    if (d.batch == 1) {
        const int rr = try_r(d, nb, A, B, C, stream, st);
        if (rr != 0) return rr;
    }
    if (d.Ko != 1) return 0;                       // (ko, ki) already collapsed by the caller when uniform
    const int64_t K = d.Ki;
    if (K > 128 || K < 1) return 0;
    // which side streams?  A batch index joins the streamed index when the small operand is shared
    // by the batch and the output is contiguous across it (right chain GEMM1: b = k, n = p'').
    // (when both sides could play the small operand the smaller one does; a batched product whose larger side
    // is the shared one -- T[q,k,b,p''] of tt_fused.hip with few p'' -- still goes here, with that side small)
    const bool a_ok = d.M <= 128 && d.batch * d.N >= 2048 && (d.batch == 1 || (d.a_b == 0 && d.c_b >= 0));
    const bool b_ok = d.N <= 128 && d.batch * d.M >= 2048 && (d.batch == 1 || (d.b_b == 0 && d.c_b >= 0));
    const bool a_small = a_ok && (d.M <= d.N || !b_ok);
    const bool b_small = !a_small && b_ok;
    if (!a_small && !b_small) return 0;
    SkinnyS s{};
    s.nb = nb;
    for (int b = 0; b < nb; ++b) {
        s.W[b] = a_small ? A[b] : B[b];
        s.S[b] = a_small ? B[b] : A[b];
        s.C[b] = C[b];
    }
    if (a_small) {
        s.w_k = d.a_ki; s.w_m = d.a_m; s.P = (int)d.M;
        s.s_j = d.b_n; s.s_k = d.b_ki; s.s_u = d.b_b; s.V = (int)d.N;
        s.c_m = d.c_m; s.c_j = d.c_n;
    } else {
        s.w_k = d.b_ki; s.w_m = d.b_n; s.P = (int)d.N;
        s.s_j = d.a_m; s.s_k = d.a_ki; s.s_u = d.a_b; s.V = (int)d.M;
        s.c_m = d.c_n; s.c_j = d.c_m;
    }
    s.J = d.batch * (int64_t)s.V;
    s.U = (int)d.batch;
    if (d.batch == 1) s.s_u = 0;
    s.tvl = 4;
    s.nbv = (int)cdiv(s.V, 1 << s.tvl);
    if (s.s_j < 0 || s.s_k < 0 || s.s_u < 0 || s.w_k < 0 || s.w_m < 0 || s.c_m < 0 || s.c_j < 0) return 0;
    if (s.J >= (1ll << 31) - 256) return 0;
    s.K = (int)K;
    s.alpha = d.alpha;
    s.accumulate = d.accumulate;
    { const char *e = getenv("TTSK_SK_STAMPS"); s.stamps = e ? (long long *)strtoull(e, nullptr, 0) : nullptr; }
    s.s_extent = (d.batch - 1) * s.s_u + ((int64_t)s.V - 1) * s.s_j + (K - 1) * s.s_k + 1;
    s.w_extent = (K - 1) * s.w_k + (s.P - 1) * s.w_m + 1;
    s.c_u = d.batch == 1 ? 0 : d.c_b;
    s.c_extent = (s.P - 1) * s.c_m + (d.batch - 1) * s.c_u + ((int64_t)(d.batch == 1 ? s.J : s.V) - 1) * s.c_j + 1;
    // per-lane 32-bit byte offsets only span one row block (16 rows, 4 k) and one output tile (16 x 16)
    const int64_t span_s = (16 * s.s_j + 4 * s.s_k) * 8, span_c = (16 * s.c_j + 16 * s.c_m) * 8;
    if (span_s >= (1ll << 32) - 64 || span_c >= (1ll << 32) - 64 || s.w_extent * 8 >= (1ll << 31)) return 0;
    // everything within reach of one descriptor + 32-bit offsets (incl. the last partial group and the
    // padded k-blocks)?  Otherwise the variant with 64-bit origins.
    const int64_t reach = ((d.batch + 4) * s.s_u + ((int64_t)s.V + 96) * s.s_j + (K + 24) * s.s_k) * 8;
    const int64_t reach_c = ((d.batch + 4) * s.c_u + ((int64_t)(d.batch == 1 ? s.J : s.V) + 96) * s.c_j + 144 * s.c_m) * 8;
    s.big = (reach >= (1ll << 32) - 64 || reach_c >= (1ll << 32) - 64) ? 1 : 0;
    const int npt = (int)cdiv(s.P, 16);
    if ((size_t)((cdiv(K, 4) + 4) * 4 * ldmf(16 * npt) + 16) * 8 + 2048 > 160 * 1024) return 0;
    const bool prof = prof_on();
    int sh = 0;
    int rc = run_s(s, st, prof ? &sh : nullptr, 2.0 * nb * (double)d.batch * (double)d.M * (double)d.N * (double)K);
    if (prof) prof_close(st);
    return rc == TTSK_OK ? 1 : rc;
}

}  // namespace ttsk

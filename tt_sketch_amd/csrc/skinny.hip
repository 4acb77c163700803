// Host side of the barrier-free chain kernels (skinny.h): shape tests and launch geometry.
#include <cstdlib>
#include "skinny.h"

namespace ttsk {

template <int D>
int launch_skinny_s_depth(const SkinnyS &a, int npt, int spt, size_t lds_bytes, int grid, hipStream_t st);
extern template int launch_skinny_s_depth<5>(const SkinnyS &, int, int, size_t, int, hipStream_t);

int launch_skinny_r_0(const SkinnyR &a, int nmt, int nnt, int grid, hipStream_t st);
int launch_skinny_r_1(const SkinnyR &a, int nmt, int nnt, int grid, hipStream_t st);
int launch_skinny_r_2(const SkinnyR &a, int nmt, int nnt, int grid, hipStream_t st);
int launch_skinny_r_3(const SkinnyR &a, int nmt, int nnt, int grid, hipStream_t st);

// out[m, n] (+)= alpha sum_c slab[c][m][n]; 16 chunk lanes x 16 consecutive n per workgroup
__global__ __launch_bounds__(256) void skinny_r_reduce(const double *__restrict__ slab, int chunks, int M, int N,
                                                       double *__restrict__ C, int64_t c_m, int64_t c_n, double alpha,
                                                       int accumulate)
{
    __shared__ double part[16][17];
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
        double *c = C + m * c_m + n * c_n;
        *c = accumulate ? *c + alpha * tot : alpha * tot;
    }
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
    const int64_t nrb = cdiv(a.J, 16);
    const int cus = num_cu();
    // fifth, shared row block or not: fewer rounds of workgroups over the CUs wins
    // SIMD s runs waves s and s+4: ceil(npt/2) + floor(npt/2) own tile strips, + 2 shared ones if dealt
    int spt = 0;
    {
        const int64_t g4 = cdiv(nrb, 4), g5 = cdiv(nrb, 5);
        const int sh_cost = npt > 4 ? 2 : 1;
        const int64_t t4 = cdiv(g4, cus) * npt, t5 = cdiv(g5, cus) * (npt + sh_cost);
        if (t5 < t4) spt = 1;
    }
    const int64_t groups = cdiv(nrb, spt ? 5 : 4);
    a.groups = (int)groups;
    const int ldw = ldmf(16 * npt);
    const size_t lds = (size_t)kb * 4 * ldw * 8;
    const int grid = (int)(groups < cus ? groups : cus);
    if (prof) prof_open(st, flops, 3, npt, spt != 0, false);
    return launch_skinny_s_depth<5>(a, npt, spt, lds, grid, st);
}

// long-K: both M, N <= 128, both operands contiguous along their non-contracted index, 16-byte
// loads possible (even extents / strides, aligned bases); desc not collapsed (two-level kappa ok)
static int try_r(const ttsk_gemm_desc &d, const double *A, const double *B, double *C, int stream, hipStream_t st)
{
    const int64_t K = d.Ko * d.Ki;
    if (d.M > 128 || d.N > 128 || K < 4096) return 0;
    if (d.Ko > 1 && d.Ki < 4) return 0;
    if (d.a_m != 1 || d.b_n != 1 || d.a_ko < 0 || d.a_ki < 0 || d.b_ko < 0 || d.b_ki < 0) return 0;
    if ((d.M | d.N | d.a_ko | d.a_ki | d.b_ko | d.b_ki) & 1) return 0;
    if (((uintptr_t)A | (uintptr_t)B) & 15) return 0;
    SkinnyR r{};
    // the operand with more 16-row tiles plays "A" (row halves), the other "B" (column strips)
    const bool swap = cdiv(d.N, 16) > cdiv(d.M, 16);
    if (!swap) {
        r.A = A; r.a_ko = d.a_ko; r.a_ki = d.a_ki; r.M = (int)d.M;
        r.B = B; r.b_ko = d.b_ko; r.b_ki = d.b_ki; r.N = (int)d.N;
    } else {
        r.A = B; r.a_ko = d.b_ko; r.a_ki = d.b_ki; r.M = (int)d.N;
        r.B = A; r.b_ko = d.a_ko; r.b_ki = d.a_ki; r.N = (int)d.M;
    }
    if (d.Ko == 1) { r.a_ko = 0; r.b_ko = 0; }
    r.Ki = d.Ki;
    r.K = K;
    if (d.Ko > 1 && r.a_ko == d.Ki * r.a_ki && r.b_ko == d.Ki * r.b_ki) r.Ki = K;   // uniform walk
    { const char *e = getenv("TTSK_SK_STAMPS"); r.stamps = e ? (long long *)strtoull(e, nullptr, 0) : nullptr; }
    r.a_extent = (r.M - 1) + (d.Ko - 1) * r.a_ko + (d.Ki - 1) * r.a_ki + 1;
    r.b_extent = (r.N - 1) + (d.Ko - 1) * r.b_ko + (d.Ki - 1) * r.b_ki + 1;
    // 32-bit byte offsets including the kappa walk past the end of the last chunk
    const int64_t reach_a = (144 + (d.Ko + 1) * r.a_ko + (d.Ki + 64) * r.a_ki) * 8;
    const int64_t reach_b = (144 + (d.Ko + 1) * r.b_ko + (d.Ki + 64) * r.b_ki) * 8;
    if (reach_a >= (1ll << 32) - 64 || reach_b >= (1ll << 32) - 64) return 0;
    const int cus = num_cu();
    r.chunk = cdiv(cdiv(K, cus), 4) * 4;
    const int chunks = (int)cdiv(K, r.chunk);
    // (one slab per XCD filled with L2-local fp64 atomics was measured 3x slower than slab + reduce)
    r.slab = (double *)scratch(stream, SCRATCH_GEMM, (size_t)chunks * r.M * r.N * 8 + 64);
    if (!r.slab) return TTSK_ERR_HIP;
    const int nmt = (int)cdiv(r.M, 16), nnt = (int)cdiv(r.N, 16);
    const bool prof = prof_on();
    if (prof) prof_open(st, 2.0 * (double)d.M * (double)d.N * (double)K, 4, nmt * 10 + nnt, false, false);
    int rc;
    if (nmt <= 4) rc = launch_skinny_r_0(r, nmt, nnt, chunks, st);
    else if (nmt <= 6) rc = launch_skinny_r_1(r, nmt, nnt, chunks, st);
    else if (nmt == 7) rc = launch_skinny_r_2(r, nmt, nnt, chunks, st);
    else rc = launch_skinny_r_3(r, nmt, nnt, chunks, st);
    if (prof) prof_close(st);
    if (rc != TTSK_OK) return rc;
    const int64_t mn = (int64_t)r.M * r.N;
    hipLaunchKernelGGL(skinny_r_reduce, dim3((unsigned)cdiv(mn, 16)), dim3(256), 0, st, r.slab, chunks, r.M, r.N, C,
                       swap ? d.c_n : d.c_m, swap ? d.c_m : d.c_n, d.alpha, d.accumulate);
    TTSK_LAUNCH_CHECK();
    return 1;
}

int skinny_try(const ttsk_gemm_desc &d, const double *A, const double *B, double *C, const double *k_scale,
               int stream, hipStream_t st)
{
    if (!skinny_mode() || k_scale) return 0;
    if (d.batch == 1) {
        const int rr = try_r(d, A, B, C, stream, st);
        if (rr != 0) return rr;
    }
    if (d.Ko != 1) return 0;                       // (ko, ki) already collapsed by the caller when uniform
    const int64_t K = d.Ki;
    if (K > 128 || K < 1) return 0;
    // which side streams?  A batch index joins the streamed index when the small operand is shared
    // by the batch and the output is contiguous across it (right chain GEMM1: b = k, n = p'').
    const bool a_small = d.M <= 128 && d.batch * d.N >= 2048 && d.M <= d.N &&
                         (d.batch == 1 || (d.a_b == 0 && d.c_b == d.N * d.c_n));
    const bool b_small = !a_small && d.N <= 128 && d.batch * d.M >= 2048 &&
                         (d.batch == 1 || (d.b_b == 0 && d.c_b == d.M * d.c_m));
    if (!a_small && !b_small) return 0;
    SkinnyS s{};
    if (a_small) {
        s.W = A; s.w_k = d.a_ki; s.w_m = d.a_m; s.P = (int)d.M;
        s.S = B; s.s_j = d.b_n; s.s_k = d.b_ki; s.s_jo = d.b_b; s.Ji = d.N;
        s.c_m = d.c_m; s.c_j = d.c_n;
    } else {
        s.W = B; s.w_k = d.b_ki; s.w_m = d.b_n; s.P = (int)d.N;
        s.S = A; s.s_j = d.a_m; s.s_k = d.a_ki; s.s_jo = d.a_b; s.Ji = d.M;
        s.c_m = d.c_n; s.c_j = d.c_m;
    }
    s.J = d.batch * s.Ji;
    if (d.batch == 1) s.s_jo = 0;
    if (s.s_j < 0 || s.s_k < 0 || s.s_jo < 0 || s.w_k < 0 || s.w_m < 0 || s.c_m < 0 || s.c_j < 0) return 0;
    if (s.J >= (1ll << 31) - 256) return 0;
    s.K = (int)K;
    s.C = C;
    s.alpha = d.alpha;
    s.accumulate = d.accumulate;
    { const char *e = getenv("TTSK_SK_STAMPS"); s.stamps = e ? (long long *)strtoull(e, nullptr, 0) : nullptr; }
    s.s_extent = (d.batch - 1) * s.s_jo + (s.Ji - 1) * s.s_j + (K - 1) * s.s_k + 1;
    s.w_extent = (K - 1) * s.w_k + (s.P - 1) * s.w_m + 1;
    s.c_extent = (s.P - 1) * s.c_m + (s.J - 1) * s.c_j + 1;
    // 32-bit byte offsets, including the rows of the last (partial) group and the padded k-blocks
    const int64_t reach = ((d.batch + 1) * s.s_jo + (s.Ji + 96) * s.s_j + (K + 24) * s.s_k) * 8;
    const int64_t reach_c = ((s.J + 96) * s.c_j + 144 * s.c_m) * 8;
    if (reach >= (1ll << 32) - 64 || reach_c >= (1ll << 32) - 64 || s.w_extent * 8 >= (1ll << 31)) return 0;
    const int npt = (int)cdiv(s.P, 16);
    if ((size_t)((cdiv(K, 4) + 4) * 4 * ldmf(16 * npt) + 16) * 8 > 160 * 1024) return 0;
    const bool prof = prof_on();
    int sh = 0;
    int rc = run_s(s, st, prof ? &sh : nullptr, 2.0 * (double)d.batch * (double)d.M * (double)d.N * (double)K);
    if (prof) prof_close(st);
    return rc == TTSK_OK ? 1 : rc;
}

}  // namespace ttsk

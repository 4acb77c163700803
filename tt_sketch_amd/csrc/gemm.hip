// Generic strided / batched fp64 contraction on MFMA 16x16x4 (gfx950).
//
//   C[b,m,n] (+)= alpha * sum_{ko,ki} kscale[ko,ki] * A[b,m,ko,ki] * B[b,ko,ki,n]
//
// This is the workhorse behind every einsum of the sketch path that is not served by
// a fused kernel (CP / Tucker / dense inputs, DenseGaussianDRM, Omega products, the
// pinv / QR applications).  Arbitrary element strides make Tensor.T, rank_min:rank_max
// slices and unfoldings zero-copy views.
//
// Tiling: one 256-thread workgroup (4 wave64, 2x2) per BM x BN tile of C, BK = 16.
// A and B tiles are staged in LDS in one of two layouts chosen by which index is
// contiguous in global memory, so that both the global read and the LDS write are
// conflict free and the MFMA fragment reads (ds_read_b64) hit 64 distinct banks:
//   "m-fast": tile[k][m], ld = BM+16 (== 16 mod 32)   "k-fast": tile[m][k], ld = 18 (== 2 mod 32)
// Split-K over (ko,ki) through gridDim.z with fp64 global atomics.
#include <cstdlib>
#include "gemm_kernel.h"
#include "skinny.h"

namespace ttsk {

// C[b,m,n] (+)= alpha * sum_z partial[b*splits+z][tile(m), tile(n)][t][lane]
// 256 threads = 64 consecutive partial elements x 4 interleaved z-lanes, 4 independent loads in
// flight per thread, LDS combine: the sum over ~100 slabs costs ~8 dependent round trips, not 100.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(ttsk_gemm_desc d, const double *__restrict__ partial,
                                                            double *__restrict__ C, int splits, int64_t tiles_m,
                                                            int64_t tiles_n, int rota)
{
    __shared__ double red[4][64];
    const int64_t per_b = tiles_m * tiles_n * 256;
    const int64_t total = d.batch * per_b;
    const int e = threadIdx.x & 63, g = threadIdx.x >> 6;
    for (int64_t g0 = (int64_t)blockIdx.x * 64; g0 < total; g0 += (int64_t)gridDim.x * 64) {
        const int64_t idx = g0 + e;                 // per_b is a multiple of 256: a group never straddles b
        const int64_t b = idx / per_b, rem = idx - b * per_b;
        const double *p = partial + (b * splits) * per_b + rem;
        double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        int z = g;
        for (; z + 12 < splits; z += 16) {
            s0 += p[(int64_t)z * per_b];
            s1 += p[(int64_t)(z + 4) * per_b];
            s2 += p[(int64_t)(z + 8) * per_b];
            s3 += p[(int64_t)(z + 12) * per_b];
        }
        for (; z < splits; z += 4) s0 += p[(int64_t)z * per_b];
        red[g][e] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        if (g == 0) {
            const double s = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
            const int64_t tile = rem >> 8;
            const int t = (int)((rem >> 6) & 3), lane = (int)(rem & 63);
            const int64_t ti = tile / tiles_n, tj = tile - ti * tiles_n;
            const int li = lane >> 4, beta = (lane >> 2) & 3, jj = lane & 3;
            const int rb = rota ? ((beta + t) & 3) : beta, cb = rota ? beta : ((beta + t) & 3);
            const int64_t m = ti * 16 + 4 * rb + li, n = tj * 16 + 4 * cb + jj;
            if (m < d.M && n < d.N) {
                double *c = C + b * d.c_b + m * d.c_m + n * d.c_n;
                *c = d.alpha * s + (d.accumulate ? *c : 0.0);
            }
        }
        __syncthreads();
    }
}

__global__ void fill3_kernel(double *C, int64_t batch, int64_t M, int64_t N, int64_t c_b, int64_t c_m,
                             int64_t c_n, double value)
{
    int64_t tot = batch * M * N;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < tot;
         i += (int64_t)gridDim.x * blockDim.x) {
        int64_t n = i % N, t = i / N;
        int64_t m = t % M, b = t / M;
        C[b * c_b + m * c_m + n * c_n] = value;
    }
}

struct CopyDesc {
    int64_t shape[5], ds[5], ss[5];
    int64_t total;
};
__global__ void copy_strided_kernel(double *dst, const double *src, CopyDesc c)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < c.total;
         i += (int64_t)gridDim.x * blockDim.x) {
        int64_t t = i, so = 0, dof = 0;
#pragma unroll
        for (int a = 4; a >= 0; --a) {
            int64_t q = t / c.shape[a], r = t - q * c.shape[a];
            so += r * c.ss[a];
            dof += r * c.ds[a];
            t = q;
        }
        dst[dof] = src[so];
    }
}

__global__ void axpby_kernel(double *y, const double *x, double a, double b, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x)
        y[i] = a * x[i] + (b == 0.0 ? 0.0 : b * y[i]);
}

// dst[i] (+)= sum_b src[b * stride + i]: the sketches of a batch summed into one (TensorSum)
// (the slices are requested eight at a time and added in order: one load after the other, each waiting for the sum,
// was 15 us for 32 slices of 6400 numbers -- a handful of workgroups, nothing to hide a round trip behind)
__global__ __launch_bounds__(256) void sum_slices_kernel(double2 *dst, const double2 *src, int nb, size_t stride2,
                                                         size_t n2, int accumulate)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) {
        double2 acc = accumulate ? dst[i] : make_double2(0.0, 0.0);
        const double2 *p = src + i;
        int b = 0;
        for (; b + 8 <= nb; b += 8, p += 8 * stride2) {
            double2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(size_t)u * stride2];
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; }
        }
        for (; b < nb; ++b, p += stride2) {
            const double2 v = *p;
            acc.x += v.x;
            acc.y += v.y;
        }
        dst[i] = acc;
    }
}

// the same for odd lengths / strides or bases that are not 16-byte aligned
__global__ __launch_bounds__(256) void sum_slices_scalar_kernel(double *dst, const double *src, int nb, size_t stride, size_t n,
                                                                int accumulate)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        double acc = accumulate ? dst[i] : 0.0;
        const double *p = src + i;
        int b = 0;
        for (; b + 8 <= nb; b += 8, p += 8 * stride) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(size_t)u * stride];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (; b < nb; ++b, p += stride) acc += *p;
        dst[i] = acc;
    }
}

// 512 threads = two waves per SIMD and at most 256 registers each: with __launch_bounds__(256) hipcc parks the
// accumulators in AGPRs and copies them in and out around every instruction (32 v_accvgpr moves + s_nop per
// 4 matrix instructions) -- round 1's "49 TF/s ceiling of the 16x16x4 form" was that code, not the pipe.
__global__ __launch_bounds__(512) void mfma_probe_kernel(double *sink, int iters, double seed)
{
    v4d a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    double x = seed + threadIdx.x * 1e-3, y = seed - threadIdx.x * 1e-3;
    for (int i = 0; i < iters; ++i) {
        a0 = mfma16(x, y, a0);
        a1 = mfma16(y, x, a1);
        a2 = mfma16(x, x, a2);
        a3 = mfma16(y, y, a3);
    }
    v4d t = a0 + a1 + a2 + a3;
    if (t[0] + t[1] + t[2] + t[3] == 12345.678) sink[blockIdx.x] = t[0];
}

struct GemmPlan {
    int family, tiles, bm, bn;
};

static GemmPlan plan_gemm(int64_t M, int64_t N)
{
    // Tiles per wave in the skinny families are capped (default 4): smaller workgroups leave room
    // for 3-4 of them per CU, and co-resident workgroups are what hides the load latency of these
    // short-K products (one wave per SIMD cannot overlap its own loads with its MFMAs).
    static int cap = [] {
        const char *e = getenv("TTSK_MAX_TILES");
        int v = e ? atoi(e) : 4;
        return v < 1 ? 1 : (v > 8 ? 8 : v);
    }();
    GemmPlan p;
    auto split = [&](int64_t X) {            // tiles per wave so that rows*tiles covers ceil(X/16) evenly
        const int need = (int)cdiv(X, 16);
        const int rows = (int)cdiv(need, cap);
        return (int)cdiv(need, rows);
    };
    if (M <= 128 && M <= N) { p.family = 1; p.tiles = split(M); p.bm = 16 * p.tiles; p.bn = 64; }
    else if (N <= 128) { p.family = 2; p.tiles = split(N); p.bm = 64; p.bn = 16 * p.tiles; }
    else { p.family = 0; p.tiles = 2; p.bm = 64; p.bn = 64; }
    return p;
}

int launch_splitk_reduce(const ttsk_gemm_desc &d, const double *partial, double *C, int splits, int64_t tiles_m,
                         int64_t tiles_n, int rota, hipStream_t st)
{
    const int64_t total = d.batch * tiles_m * tiles_n * 256;
    int64_t blocks = cdiv(total, 64);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, d, partial, C, splits,
                       tiles_m, tiles_n, rota);
    TTSK_LAUNCH_CHECK();
    return TTSK_OK;
}

static bool even(int64_t v) { return (v & 1) == 0; }

}  // namespace ttsk

using namespace ttsk;

extern "C" {

int ttsk_gemm(const ttsk_gemm_desc *dp, const double *A, const double *B, double *C,
              const double *k_scale, int stream)
{
    TTSK_STREAM(st, stream);
    TTSK_ARG(dp && A && B && C, "ttsk_gemm: NULL argument");
    ttsk_gemm_desc d = *dp;
    TTSK_ARG(d.batch >= 0 && d.M >= 0 && d.N >= 0 && d.Ko >= 0 && d.Ki >= 0, "ttsk_gemm: negative size");
    if (d.batch == 0 || d.M == 0 || d.N == 0) return TTSK_OK;
    const int64_t K = d.Ko * d.Ki;
    static int trace = [] { const char *e = getenv("TTSK_GEMM_TRACE"); return e ? atoi(e) : 0; }();
    if (trace)
        fprintf(stderr, "ttsk_gemm b=%lld M=%lld N=%lld Ko=%lld Ki=%lld | a: b%lld m%lld ko%lld ki%lld | b: b%lld ko%lld ki%lld n%lld | c: b%lld m%lld n%lld acc=%d\n",
                (long long)d.batch, (long long)d.M, (long long)d.N, (long long)d.Ko, (long long)d.Ki, (long long)d.a_b,
                (long long)d.a_m, (long long)d.a_ko, (long long)d.a_ki, (long long)d.b_b, (long long)d.b_ko,
                (long long)d.b_ki, (long long)d.b_n, (long long)d.c_b, (long long)d.c_m, (long long)d.c_n, d.accumulate);
    if (K == 0) {
        if (!d.accumulate) {
            hipLaunchKernelGGL(fill3_kernel, dim3(256), dim3(256), 0, st, C, d.batch, d.M, d.N, d.c_b,
                               d.c_m, d.c_n, 0.0);
            TTSK_LAUNCH_CHECK();
        }
        return TTSK_OK;
    }
    if (d.batch == 1 && !k_scale && d.c_n == 1) {
        // rows of an unfolding against a DRM matrix, both contiguous along a long contracted index (the right-hand products
        // of a dense sketch with Gaussian matrices): both operands through LDS by LDS-DMA (dense_right_pass.hip)
        ttsk_gemm_desc n = d;
        if (n.Ki == 1) { n.Ki = n.Ko; n.Ko = 1; n.a_ki = n.a_ko; n.b_ki = n.b_ko; }
        else if (n.Ko > 1 && n.a_ko == n.Ki * n.a_ki && n.b_ko == n.Ki * n.b_ki) { n.Ki *= n.Ko; n.Ko = 1; }
        if (n.Ko == 1 && n.a_ki == 1 && n.b_ki == 1 && n.N <= 48 && n.a_m >= n.Ki && n.b_n >= n.Ki) {
            const int rs = rows_longk_try(A, n.M, n.a_m, B, (int)n.N, n.b_n, n.Ki, C, n.c_m, n.alpha, n.accumulate, stream, st);
            if (rs < 0) return rs;
            if (rs == 1) return TTSK_OK;
        }
    }
    {
        // the chain shapes (tall-skinny, K <= 128) have their own barrier-free kernels
        ttsk_gemm_desc n = d;
        if (n.Ki == 1) { n.Ki = n.Ko; n.Ko = 1; n.a_ki = n.a_ko; n.b_ki = n.b_ko; }
        else if (n.Ko > 1 && n.a_ko == n.Ki * n.a_ki && n.b_ko == n.Ki * n.b_ki) { n.Ki *= n.Ko; n.Ko = 1; }
        int rs = skinny_try(n, A, B, C, k_scale, stream, st);
        if (rs == 0 && !k_scale) rs = small_try_batch(n, 1, &A, &B, &C, stream, st);
        if (rs < 0) return rs;
        if (rs == 1) return TTSK_OK;
    }
    if (d.batch >= 2 && K >= 32768 && !k_scale && d.c_b != 0) {
        // a long contraction batched over an index that only one operand carries (the per-slice right products of
        // dense_sketch.py: P^T X_b for every slice b of the tensor): up to SK_MAXB problems per launch of the long-K
        // chain kernel instead of the generic tiles (C2's Psi_0 pass: 2.76 -> 2.59 ms)
        static int on = [] { const char *e = getenv("TTSK_GEMM_BATCH_LONGK"); return e ? atoi(e) : 1; }();
        ttsk_gemm_desc n = d;
        n.batch = 1; n.a_b = n.b_b = n.c_b = 0;
        if (n.Ki == 1) { n.Ki = n.Ko; n.Ko = 1; n.a_ki = n.a_ko; n.b_ki = n.b_ko; }
        else if (n.Ko > 1 && n.a_ko == n.Ki * n.a_ki && n.b_ko == n.Ki * n.b_ki) { n.Ki *= n.Ko; n.Ko = 1; }
        bool done = on != 0;
        for (int64_t b0 = 0; b0 < d.batch && done; b0 += SK_MAXB) {
            const int cnt = (int)(d.batch - b0 < SK_MAXB ? d.batch - b0 : SK_MAXB);
            const double *Ap[SK_MAXB], *Bp[SK_MAXB];
            double *Cp[SK_MAXB];
            for (int b = 0; b < cnt; ++b) {
                Ap[b] = A + (b0 + b) * d.a_b; Bp[b] = B + (b0 + b) * d.b_b; Cp[b] = C + (b0 + b) * d.c_b;
            }
            const int rs = skinny_try_batch(n, cnt, Ap, Bp, Cp, stream, st);
            if (trace) fprintf(stderr, "  long-K batch slice %lld: rs = %d\n", (long long)b0, rs);
            if (rs < 0) return rs;
            if (rs == 0) {
                // rejected (a slice's alignment / extent differs from the first ones'): the slices done so far stand, the
                // rest go through the generic tiles below -- same result in both accumulate modes
                A += b0 * d.a_b; B += b0 * d.b_b; C += b0 * d.c_b;
                d.batch -= b0;
                done = false;
            }
        }
        if (done) return TTSK_OK;
    }
    const GemmPlan p = plan_gemm(d.M, d.N);
    const int64_t tiles = d.batch * cdiv(d.M, p.bm) * cdiv(d.N, p.bn);
    int splits = d.split_k;
    if (splits <= 0) {
        splits = 1;
        if (tiles < 192 && K >= 4 * BK) {
            int64_t want = cdiv(512, tiles);
            int64_t maxs = cdiv(K, K >= 32 * BK ? 4 * BK : BK);
            splits = (int)(want < maxs ? want : maxs);
            if (splits > 1024) splits = 1024;
            if (splits < 1) splits = 1;
        }
    }
    TTSK_ARG(d.batch * (int64_t)splits <= 65535, "ttsk_gemm: batch*split_k too large (%lld)",
             (long long)(d.batch * splits));
    int64_t kchunk = cdiv(cdiv(K, splits), BK) * BK;
    splits = (int)cdiv(K, kchunk);
    double *partial = nullptr;
    const int64_t tiles_m = cdiv(d.M, p.bm) * (p.bm / 16), tiles_n = cdiv(d.N, p.bn) * (p.bn / 16);
    if (splits > 1) {
        partial = (double *)scratch(stream, SCRATCH_GEMM, (size_t)(d.batch * splits * tiles_m * tiles_n * 256) * 8 + 64);
        if (!partial) return TTSK_ERR_HIP;
    }
    // normalise a single contracted index into the inner slot
    if (d.Ki == 1) { d.Ki = d.Ko; d.Ko = 1; d.a_ki = d.a_ko; d.b_ki = d.b_ko; }
    if (d.Ko == 1) { d.a_ko = 0; d.b_ko = 0; }
    if (d.batch == 1) { d.a_b = 0; d.b_b = 0; }
    // which index is contiguous in memory decides the staging layout; 16-byte loads when the
    // contiguous index pairs up cleanly (even strides, pairs never straddle ko, aligned base)
    const bool ak = d.a_m != 1 && d.a_ki == 1;
    const bool bk = d.b_n != 1 && d.b_ki == 1;
    const bool k_pairs = d.Ko == 1 || even(d.Ki);
    int avec = ak ? (k_pairs && even(d.a_m) && even(d.a_ko) && even(d.a_b))
                  : (d.a_m == 1 && even(d.a_ki) && even(d.a_ko) && even(d.a_b));
    int bvec = bk ? (k_pairs && even(d.b_n) && even(d.b_ko) && even(d.b_b))
                  : (d.b_n == 1 && even(d.b_ki) && even(d.b_ko) && even(d.b_b));
    if (((uintptr_t)A & 15) != 0) avec = 0;
    if (((uintptr_t)B & 15) != 0) bvec = 0;
    // 16-byte loads additionally need whole pairs inside the extents
    if (ak ? (K & 1) : (d.M & 1)) avec = 0;
    if (bk ? (K & 1) : (d.N & 1)) bvec = 0;
    // The fast staging path addresses a tile through a buffer resource: 32-bit byte offsets from the
    // workgroup's first row / column, K offset in a scalar register.
    const int64_t a_extent = (d.M - 1) * d.a_m + (d.Ko - 1) * d.a_ko + (d.Ki - 1) * d.a_ki + 1;
    const int64_t b_extent = (d.N - 1) * d.b_n + (d.Ko - 1) * d.b_ko + (d.Ki - 1) * d.b_ki + 1;
    const int64_t span_a = (int64_t)p.bm * d.a_m + d.Ko * d.a_ko + (d.Ki + BK) * d.a_ki;
    const int64_t span_b = (int64_t)p.bn * d.b_n + d.Ko * d.b_ko + (d.Ki + BK) * d.b_ki;
    const int fast_ok = d.a_m >= 0 && d.b_n >= 0 && d.a_ki >= 0 && d.b_ki >= 0 && d.a_ko >= 0 && d.b_ko >= 0 &&
                        span_a * 8 < (1ll << 31) && span_b * 8 < (1ll << 31);
    GemmLaunch g{d, A, B, k_scale, C, partial, p.family, p.tiles, splits, avec, bvec, fast_ok, kchunk, p.bm, p.bn,
                 a_extent, b_extent};
    int rc;
    const bool prof = prof_on();
    if (prof) prof_open(st, 2.0 * (double)d.batch * (double)d.M * (double)d.N * (double)K, p.family, p.tiles, ak, bk);
    if (ak && bk) rc = launch_gemm_layout<true, true>(g, st);
    else if (ak) rc = launch_gemm_layout<true, false>(g, st);
    else if (bk) rc = launch_gemm_layout<false, true>(g, st);
    else rc = launch_gemm_layout<false, false>(g, st);
    if (prof) prof_close(st);
    if (rc || !partial) return rc;
    return launch_splitk_reduce(d, partial, C, splits, tiles_m, tiles_n, p.family == 2 ? 1 : 0, st);
}

int ttsk_mfma_f64_peak_probe(double *tflops)
{
    TTSK_STREAM(st, 0);
    TTSK_ARG(tflops, "ttsk_mfma_f64_peak_probe: NULL");
    double *sink = (double *)scratch(0, SCRATCH_GEMM, 1 << 16);
    if (!sink) return TTSK_ERR_HIP;
    const int blocks = 256 * 4, iters = 2000;
    hipEvent_t a, b;
    TTSK_HIP(hipEventCreate(&a));
    TTSK_HIP(hipEventCreate(&b));
    double best = 0;
    for (int rep = 0; rep < 4; ++rep) {
        TTSK_HIP(hipEventRecord(a, st));
        hipLaunchKernelGGL(mfma_probe_kernel, dim3(blocks), dim3(512), 0, st, sink, iters, 1.0 + rep);
        TTSK_HIP(hipEventRecord(b, st));
        TTSK_HIP(hipEventSynchronize(b));
        float ms = 0;
        TTSK_HIP(hipEventElapsedTime(&ms, a, b));
        double fl = (double)blocks * 8 /*waves*/ * iters * 4 /*mfma*/ * 2048.0;
        double tf = fl / (ms * 1e-3) * 1e-12;
        if (rep > 0 && tf > best) best = tf;
    }
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    *tflops = best;
    return TTSK_OK;
}

int ttsk_copy_strided(double *dst, const double *src, int ndim, const int64_t *shape,
                      const int64_t *dst_strides, const int64_t *src_strides, int stream)
{
    TTSK_STREAM(st, stream);
    TTSK_ARG(ndim >= 0 && ndim <= 5, "ttsk_copy_strided: ndim %d > 5", ndim);
    CopyDesc c;
    c.total = 1;
    for (int a = 0; a < 5; ++a) {
        int src_a = a - (5 - ndim);
        if (src_a >= 0) {
            c.shape[a] = shape[src_a];
            c.ds[a] = dst_strides[src_a];
            c.ss[a] = src_strides[src_a];
        } else {
            c.shape[a] = 1;
            c.ds[a] = 0;
            c.ss[a] = 0;
        }
        TTSK_ARG(c.shape[a] >= 0, "ttsk_copy_strided: negative extent");
        c.total *= c.shape[a];
    }
    if (c.total == 0) return TTSK_OK;
    int64_t blocks = cdiv(c.total, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(copy_strided_kernel, dim3((unsigned)blocks), dim3(256), 0, st, dst, src, c);
    TTSK_LAUNCH_CHECK();
    return TTSK_OK;
}

int ttsk_sum_slices(double *dst, const double *src, int nb, size_t stride, size_t n, int accumulate, int stream)
{
    TTSK_STREAM(st, stream);
    TTSK_ARG(dst && src && nb >= 1, "ttsk_sum_slices: bad argument");
    if (n == 0) return TTSK_OK;
    if ((n & 1) || (stride & 1) || (((uintptr_t)dst | (uintptr_t)src) & 15)) {
        size_t blocks = (n + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(sum_slices_scalar_kernel, dim3((unsigned)blocks), dim3(256), 0, st, dst, src, nb, stride, n, accumulate);
        TTSK_LAUNCH_CHECK();
        return TTSK_OK;
    }
    size_t blocks = (n / 2 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(sum_slices_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (double2 *)dst, (const double2 *)src,
                       nb, stride / 2, n / 2, accumulate);
    TTSK_LAUNCH_CHECK();
    return TTSK_OK;
}

int ttsk_axpby(double *y, const double *x, double a, double b, size_t n, int stream)
{
    TTSK_STREAM(st, stream);
    if (n == 0) return TTSK_OK;
    size_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(axpby_kernel, dim3((unsigned)blocks), dim3(256), 0, st, y, x, a, b, n);
    TTSK_LAUNCH_CHECK();
    return TTSK_OK;
}

}  // extern "C"

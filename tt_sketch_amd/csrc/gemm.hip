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
#include "common.h"

namespace ttsk {

constexpr int BK = 16;
constexpr int LDK = 18;  // k-fast layout leading dimension

template <int BT, bool KFAST>
struct TileLayout {
    static constexpr int LD = KFAST ? LDK : BT + 16;
    static constexpr int SIZE = KFAST ? BT * LDK : BK * (BT + 16);
    __device__ static __forceinline__ int at(int x, int k) { return KFAST ? x * LDK + k : k * LD + x; }
};

struct KMap {
    int64_t Ko, Ki, s_ko, s_ki;
    __device__ __forceinline__ int64_t off(int64_t kk) const
    {
        if (Ko == 1) return kk * s_ki;
        int64_t ko = kk / Ki;
        return ko * s_ko + (kk - ko * Ki) * s_ki;
    }
};

template <int BM, int BN, bool AK, bool BKF>
__global__ __launch_bounds__(256) void gemm_f64_kernel(ttsk_gemm_desc d, const double *__restrict__ A,
                                                       const double *__restrict__ B,
                                                       double *__restrict__ C,
                                                       const double *__restrict__ kscale, int splits,
                                                       int64_t kchunk, int use_atomic)
{
    using LA = TileLayout<BM, AK>;
    using LB = TileLayout<BN, BKF>;
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int TM = WM / 16, TN = WN / 16;
    constexpr int EA = BM * BK / 256, EB = BN * BK / 256;  // elements per thread per tile
    __shared__ double As[LA::SIZE];
    __shared__ double Bs[LB::SIZE];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int64_t m0 = (int64_t)blockIdx.y * BM, n0 = (int64_t)blockIdx.x * BN;
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

    v4d acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = v4d{0.0, 0.0, 0.0, 0.0};

    double ra[EA], rb[EB];

    auto load_tiles = [&](int64_t k0) {
#pragma unroll
        for (int e = 0; e < EA; ++e) {
            int x, k;
            if (AK) { k = tid & 15; x = (tid >> 4) + 16 * e; }
            else    { x = tid % BM; k = tid / BM + (256 / BM) * e; }
            int64_t kk = k0 + k, m = m0 + x;
            double v = 0.0;
            if (kk < kend && m < d.M) {
                v = Ab[m * d.a_m + ka.off(kk)];
                if (kscale) v *= kscale[kk];
            }
            ra[e] = v;
        }
#pragma unroll
        for (int e = 0; e < EB; ++e) {
            int x, k;
            if (BKF) { k = tid & 15; x = (tid >> 4) + 16 * e; }
            else     { x = tid % BN; k = tid / BN + (256 / BN) * e; }
            int64_t kk = k0 + k, n = n0 + x;
            double v = 0.0;
            if (kk < kend && n < d.N) v = Bb[n * d.b_n + kb.off(kk)];
            rb[e] = v;
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int e = 0; e < EA; ++e) {
            int x, k;
            if (AK) { k = tid & 15; x = (tid >> 4) + 16 * e; }
            else    { x = tid % BM; k = tid / BM + (256 / BM) * e; }
            As[LA::at(x, k)] = ra[e];
        }
#pragma unroll
        for (int e = 0; e < EB; ++e) {
            int x, k;
            if (BKF) { k = tid & 15; x = (tid >> 4) + 16 * e; }
            else     { x = tid % BN; k = tid / BN + (256 / BN) * e; }
            Bs[LB::at(x, k)] = rb[e];
        }
    };

    const int fi = lane >> 4, fj = lane & 15;
    if (kbeg < kend) load_tiles(kbeg);
    for (int64_t k0 = kbeg; k0 < kend; k0 += BK) {
        __syncthreads();
        store_tiles();
        __syncthreads();
        if (k0 + BK < kend) load_tiles(k0 + BK);
#pragma unroll
        for (int ks = 0; ks < BK; ks += 4) {
            double af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = As[LA::at(wr * WM + i * 16 + fj, ks + fi)];
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = Bs[LB::at(wc * WN + j * 16 + fj, ks + fi)];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = mfma16(af[i], bf[j], acc[i][j]);
        }
    }

    double *Cb = C + b * d.c_b;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int64_t m = m0 + wr * WM + i * 16 + fi + 4 * r;
                int64_t n = n0 + wc * WN + j * 16 + fj;
                if (m < d.M && n < d.N) {
                    double v = d.alpha * acc[i][j][r];
                    double *p = Cb + m * d.c_m + n * d.c_n;
                    if (use_atomic) unsafeAtomicAdd(p, v);
                    else if (d.accumulate) *p += v;
                    else *p = v;
                }
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

__global__ __launch_bounds__(256) void mfma_probe_kernel(double *sink, int iters, double seed)
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

template <int BM, int BN>
static int launch_gemm(const ttsk_gemm_desc &d, const double *A, const double *B, double *C,
                       const double *ks, int splits, int64_t kchunk, int use_atomic, bool ak, bool bk,
                       hipStream_t st)
{
    dim3 grid((unsigned)cdiv(d.N, BN), (unsigned)cdiv(d.M, BM), (unsigned)(d.batch * splits));
    dim3 block(256);
#define L(AKV, BKV)                                                                               \
    hipLaunchKernelGGL((gemm_f64_kernel<BM, BN, AKV, BKV>), grid, block, 0, st, d, A, B, C, ks,   \
                       splits, kchunk, use_atomic)
    if (ak && bk) L(true, true);
    else if (ak) L(true, false);
    else if (bk) L(false, true);
    else L(false, false);
#undef L
    TTSK_LAUNCH_CHECK();
    return TTSK_OK;
}

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
    if (K == 0) {
        if (!d.accumulate) {
            hipLaunchKernelGGL(fill3_kernel, dim3(256), dim3(256), 0, st, C, d.batch, d.M, d.N, d.c_b,
                               d.c_m, d.c_n, 0.0);
            TTSK_LAUNCH_CHECK();
        }
        return TTSK_OK;
    }
    // tile shape by aspect ratio
    int bm = 64, bn = 64;
    if (d.M <= 32 && d.N > 64) { bm = 32; bn = 128; }
    else if (d.N <= 32 && d.M > 64) { bm = 128; bn = 32; }
    const int64_t tiles = d.batch * cdiv(d.M, bm) * cdiv(d.N, bn);
    int splits = d.split_k;
    if (splits <= 0) {
        splits = 1;
        if (tiles < 256 && K >= 256) {
            int64_t want = cdiv(768, tiles);
            int64_t maxs = cdiv(K, 64);
            splits = (int)(want < maxs ? want : maxs);
            if (splits > 512) splits = 512;
            if (splits < 1) splits = 1;
        }
    }
    TTSK_ARG(d.batch * (int64_t)splits <= 65535, "ttsk_gemm: batch*split_k too large (%lld)",
             (long long)(d.batch * splits));
    int64_t kchunk = cdiv(cdiv(K, splits), BK) * BK;
    splits = (int)cdiv(K, kchunk);
    const int use_atomic = splits > 1;
    if (use_atomic && !d.accumulate) {
        hipLaunchKernelGGL(fill3_kernel, dim3(256), dim3(256), 0, st, C, d.batch, d.M, d.N, d.c_b, d.c_m,
                           d.c_n, 0.0);
        TTSK_LAUNCH_CHECK();
    }
    // which index is contiguous in memory decides the staging layout
    const bool ak = (d.a_m != 1) && (d.Ki == 1 ? d.a_ko == 1 || d.a_ki == 1 : d.a_ki == 1);
    const bool bk = (d.b_n != 1) && (d.Ki == 1 ? d.b_ko == 1 || d.b_ki == 1 : d.b_ki == 1);
    if (bm == 64) return launch_gemm<64, 64>(d, A, B, C, k_scale, splits, kchunk, use_atomic, ak, bk, st);
    if (bm == 32) return launch_gemm<32, 128>(d, A, B, C, k_scale, splits, kchunk, use_atomic, ak, bk, st);
    return launch_gemm<128, 32>(d, A, B, C, k_scale, splits, kchunk, use_atomic, ak, bk, st);
}

int ttsk_mfma_f64_peak_probe(double *tflops)
{
    TTSK_STREAM(st, 0);
    TTSK_ARG(tflops, "ttsk_mfma_f64_peak_probe: NULL");
    double *sink = (double *)scratch(0, 1 << 16);
    if (!sink) return TTSK_ERR_HIP;
    const int blocks = 256 * 8, iters = 2000;
    hipEvent_t a, b;
    TTSK_HIP(hipEventCreate(&a));
    TTSK_HIP(hipEventCreate(&b));
    double best = 0;
    for (int rep = 0; rep < 4; ++rep) {
        TTSK_HIP(hipEventRecord(a, st));
        hipLaunchKernelGGL(mfma_probe_kernel, dim3(blocks), dim3(256), 0, st, sink, iters, 1.0 + rep);
        TTSK_HIP(hipEventRecord(b, st));
        TTSK_HIP(hipEventSynchronize(b));
        float ms = 0;
        TTSK_HIP(hipEventElapsedTime(&ms, a, b));
        double fl = (double)blocks * 4 /*waves*/ * iters * 4 /*mfma*/ * 2048.0;
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

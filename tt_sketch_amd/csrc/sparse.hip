// Sparse (COO) input kernels: gather-chain for a TT DRM, column gather for a dense
// Gaussian DRM, and the Psi scatter.  All HBM-bound streaming of (nnz x rank) panels.
#include <hipcub/hipcub.hpp>
#include "common.h"

namespace ttsk {

// TensorTrainDRM.sketch_sparse, tensor_train_drm.py:60-69.
// vout[e,b] = sum_a vin[e,a] * D[a, idx[e], b]; one thread per (e,b), b fastest so that
// the core slice row D[a,k,:] and the output row are read/written contiguously.
__global__ void sparse_ttdrm_kernel(const double *__restrict__ vin, int64_t rho,
                                    const double *__restrict__ core, int64_t n, int64_t rhop,
                                    const int64_t *__restrict__ idx, size_t N, double *__restrict__ vout)
{
    const size_t tot = N * (size_t)rhop;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < tot;
         g += (size_t)gridDim.x * blockDim.x) {
        size_t e = g / rhop;
        int64_t b = (int64_t)(g - e * rhop);
        int64_t k = idx[e];
        const double *c = core + k * rhop + b;
        double acc = 0.0;
        if (vin) {
            const double *v = vin + e * rho;
            for (int64_t a = 0; a < rho; ++a) acc = fma(v[a], c[a * n * rhop], acc);
        } else {
            acc = c[0];
        }
        vout[g] = acc;
    }
}

constexpr int MAX_MODES = 32;
struct RavelMap {
    int m;
    int row_order[MAX_MODES];
    int64_t mult[MAX_MODES];  // C-order multipliers (np.ravel_multi_index)
    int64_t row_stride;
};

// DenseGaussianDRM.sketch_sparse, dense_gaussian_drm.py:59-66.
__global__ void sparse_dense_gather_kernel(const double *__restrict__ mat, int64_t rank, int64_t cols,
                                           const int64_t *__restrict__ idx, RavelMap rm, size_t N,
                                           double *__restrict__ out)
{
    const size_t tot = N * (size_t)rank;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < tot;
         g += (size_t)gridDim.x * blockDim.x) {
        size_t e = g / rank;
        int64_t j = (int64_t)(g - e * rank);
        int64_t flat = 0;
        for (int i = 0; i < rm.m; ++i)
            flat += idx[(int64_t)rm.row_order[i] * rm.row_stride + (int64_t)e] * rm.mult[i];
        out[g] = mat[j * cols + flat];
    }
}

// sketch_psi_sparse, sparse_sketch.py:8-36,49-69 (scatter form).
// Psi[a, idx[e], c] += val[e] * Lv[e,a] * Rv[e,c].
// One workgroup walks a contiguous chunk of nonzeros; thread t owns output pair
// (a,c) = (t / r, t % r) (looping if l*r > blockDim) and keeps a running partial sum
// while consecutive nonzeros fall in the same slice k, flushing with one fp64 atomic
// per slice change.  For mode-sorted input (ttsk sorts once per tensor and mode) this
// is a segmented reduction with O(#slices) atomics per chunk; for unsorted input it
// degrades gracefully to one atomic per nonzero per pair.
__global__ __launch_bounds__(256) void sparse_psi_kernel(const double *__restrict__ val,
                                                         const int64_t *__restrict__ idx,
                                                         const int64_t *__restrict__ perm, size_t N,
                                                         const double *__restrict__ Lv, int64_t l,
                                                         const double *__restrict__ Rv, int64_t r, int64_t n,
                                                         double *__restrict__ psi, size_t chunk)
{
    // The inner loop is LDS-bandwidth bound (one read per operand and output pair), so the staged
    // panel holds val * L (one read instead of two) and the slice boundaries of a batch are a 64-bit
    // mask in scalar registers instead of a per-element index read.
    constexpr int TB = 64;  // nonzeros staged per LDS batch
    extern __shared__ double sm[];
    double *sL = sm;                 // TB * l, already scaled by the entry
    double *sR = sL + TB * l;        // TB * r
    int64_t *sK = (int64_t *)(sR + TB * r);
    __shared__ unsigned long long s_change;   // bit t: nonzero t of the batch starts a new slice
    const size_t beg = (size_t)blockIdx.x * chunk;
    const size_t end = beg + chunk < N ? beg + chunk : N;
    const int64_t pairs = l * r;
    for (int64_t p0 = 0; p0 < pairs; p0 += blockDim.x) {
        const int64_t p = p0 + threadIdx.x;
        const bool live = p < pairs;
        const int64_t a = live ? p / r : 0, c = live ? p - (p / r) * r : 0;
        double acc = 0.0;
        int64_t cur = -1;
        for (size_t b0 = beg; b0 < end; b0 += TB) {
            const int cnt = (int)((end - b0) < TB ? (end - b0) : TB);
            __syncthreads();
            if (threadIdx.x < TB) {
                const int t = threadIdx.x;
                int64_t k = -2;
                if (t < cnt) {
                    const size_t e = perm ? (size_t)perm[b0 + t] : b0 + t;
                    k = idx ? idx[e] : 0;
                }
                sK[t] = k;
            }
            for (int64_t t = threadIdx.x; t < (int64_t)cnt * l; t += blockDim.x) {
                int64_t w = t / l, a2 = t - w * l;
                size_t e = perm ? (size_t)perm[b0 + w] : b0 + w;
                sL[t] = val[e] * (Lv ? Lv[e * l + a2] : 1.0);
            }
            for (int64_t t = threadIdx.x; t < (int64_t)cnt * r; t += blockDim.x) {
                int64_t w = t / r, c2 = t - w * r;
                size_t e = perm ? (size_t)perm[b0 + w] : b0 + w;
                sR[t] = Rv ? Rv[e * r + c2] : 1.0;
            }
            __syncthreads();
            if (threadIdx.x < 64) {     // first wave: where does the slice index change inside the batch?
                const int t = threadIdx.x;
                const bool chg = t < cnt && t > 0 && sK[t] != sK[t - 1];
                const unsigned long long m = __ballot(chg);
                if (t == 0) s_change = m;
            }
            __syncthreads();
            if (live) {
                const unsigned long long change = s_change;
                const int64_t k0 = sK[0];
                if (k0 != cur) {
                    if (cur >= 0 && acc != 0.0) unsafeAtomicAdd(&psi[(a * n + cur) * r + c], acc);
                    acc = 0.0;
                    cur = k0;
                }
                if (change == 0) {                    // the usual case: the whole batch is one slice
                    for (int t = 0; t < cnt; ++t) acc = fma(sL[t * l + a], sR[t * r + c], acc);
                } else {
                    for (int t = 0; t < cnt; ++t) {
                        if ((change >> t) & 1) {
                            if (acc != 0.0) unsafeAtomicAdd(&psi[(a * n + cur) * r + c], acc);
                            acc = 0.0;
                            cur = sK[t];
                        }
                        acc = fma(sL[t * l + a], sR[t * r + c], acc);
                    }
                }
            }
        }
        if (live && cur >= 0 && acc != 0.0) unsafeAtomicAdd(&psi[(a * n + cur) * r + c], acc);
    }
}


// MFMA form of the same sum for mode-sorted input (or a single slice): one slice of Psi is the long-K
// product (val o L_slice)^T R_slice over the nonzeros of the slice, so a wave walks its stretch of the
// sorted nonzero list four at a time -- lane (x = l & 15, q = l >> 4) holds val[e_q] L[e_q][x] and
// R[e_q][x], the operand layout of v_mfma_f64_16x16x4 -- and accumulates TL x TR tiles.  The rows are
// gathered through the mode permutation: perm is read 2 D k-blocks ahead, the rows it names D k-blocks
// ahead (two rings, so neither level of the dependent gather stalls the wave).  A k-block that contains a
// slice boundary (rare: slices are long when this kernel is chosen) is replayed one nonzero at a time.
// Slices that straddle two waves meet in Psi through fp64 atomics, as in the scatter kernel; with ONE
// slice (Omega, first / last mode) every wave would hit the same l x r addresses, so there the partial
// sums go to `part` (one l x r block per wave) and sparse_part_reduce_kernel adds them.
template <int TL, int TR, bool PERM>
__global__ __launch_bounds__(256, 2) void sparse_psi_mfma_kernel(const double *__restrict__ val, const int64_t *__restrict__ idx,
                                                              const int64_t *__restrict__ perm, size_t N,
                                                              const double *__restrict__ Lv, int l,
                                                              const double *__restrict__ Rv, int r, int64_t n,
                                                              double *__restrict__ psi, size_t chunk,
                                                              double *__restrict__ part)
{
    constexpr int D = 6;
    const int lane = threadIdx.x & 63, x16 = lane & 15, kq = lane >> 4;
    const size_t wv = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t beg = wv * chunk;
    if (beg >= N) return;
    const size_t end = beg + chunk < N ? beg + chunk : N;
    const int nblk = (int)((end - beg + 3) >> 2);
    v4d acc[TL][TR];
#pragma unroll
    for (int i = 0; i < TL; ++i)
#pragma unroll
        for (int j = 0; j < TR; ++j)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[i][j][t] = 0.0;
    long long cur = idx ? -1 : 0;

    auto flush = [&](long long k) {
#pragma unroll
        for (int i = 0; i < TL; ++i)
#pragma unroll
            for (int j = 0; j < TR; ++j)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int a = 16 * i + 4 * t + (lane >> 4), c = 16 * j + (lane & 15);
                    if (a < l && c < r) {
                        if (part) part[wv * (size_t)(l * r) + (size_t)a * r + c] = acc[i][j][t];
                        else if (acc[i][j][t] != 0.0) unsafeAtomicAdd(&psi[((size_t)a * n + k) * r + c], acc[i][j][t]);
                    }
                    acc[i][j][t] = 0.0;
                }
    };

    long long pe[D];                       // nonzero number of this lane's position, k-blocks b + D .. b + 2 D - 1
    double ra[D][TL], rb[D][TR], rv[D];    // rows of L and R, entries, of k-blocks b .. b + D - 1
    long long rk[D];
    bool rok[D];
    auto load_perm = [&](int b, int d) {
        const size_t pos = beg + 4 * (size_t)b + kq;
        pe[d] = pos < end ? (PERM ? perm[pos] : (long long)pos) : -1;
    };
    auto load_rows = [&](int d) {
        const long long e = pe[d];
        const bool ok = e >= 0;
        const size_t ee = ok ? (size_t)e : 0;
        rok[d] = ok;
        rv[d] = ok ? val[ee] : 0.0;
        rk[d] = (ok && idx) ? idx[ee] : 0;
#pragma unroll
        for (int i = 0; i < TL; ++i) {
            const int a = 16 * i + x16;
            ra[d][i] = (ok && a < l) ? (Lv ? Lv[ee * l + a] : 1.0) : 0.0;
        }
#pragma unroll
        for (int j = 0; j < TR; ++j) {
            const int c = 16 * j + x16;
            rb[d][j] = (ok && c < r) ? (Rv ? Rv[ee * r + c] : 1.0) : 0.0;
        }
    };
#pragma unroll
    for (int d = 0; d < D; ++d) load_perm(d, d);
#pragma unroll
    for (int d = 0; d < D; ++d) { load_rows(d); load_perm(D + d, d); }

    const int rounds = (nblk + D - 1) / D;
    for (int it = 0; it < rounds; ++it) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            double a[TL], b[TR];
#pragma unroll
            for (int i = 0; i < TL; ++i) a[i] = ra[d][i] * rv[d];
#pragma unroll
            for (int j = 0; j < TR; ++j) b[j] = rb[d][j];
            const long long kk = rk[d];
            const bool ok = rok[d];
            // refill: rows of k-block (it + 1) D + d through the perm entry fetched a round ago, then its successor
            load_rows(d);
            load_perm((it + 2) * D + d, d);
            if (__ballot(ok && kk != cur) == 0ull) {
#pragma unroll
                for (int i = 0; i < TL; ++i)
#pragma unroll
                    for (int j = 0; j < TR; ++j) acc[i][j] = mfma16(a[i], b[j], acc[i][j]);
            } else {
                for (int q = 0; q < 4; ++q) {
                    const int okq = __shfl((int)ok, 16 * q);
                    const long long kq_slice = __shfl(kk, 16 * q);
                    if (!okq) continue;
                    if (kq_slice != cur) {
                        if (cur >= 0) flush(cur);
                        cur = kq_slice;
                    }
#pragma unroll
                    for (int i = 0; i < TL; ++i) {
                        const double am = kq == q ? a[i] : 0.0;
#pragma unroll
                        for (int j = 0; j < TR; ++j) acc[i][j] = mfma16(am, b[j], acc[i][j]);
                    }
                }
            }
        }
    }
    if (cur >= 0) flush(cur);
}

// out[t] += sum_w part[w][t]: one workgroup per output element, fixed summation order
__global__ __launch_bounds__(256) void sparse_part_reduce_kernel(const double *__restrict__ part, size_t nparts, int lr,
                                                                 double *__restrict__ out)
{
    const int t = blockIdx.x;
    double acc = 0.0;
    for (size_t w = threadIdx.x; w < nparts; w += 256) acc += part[w * (size_t)lr + t];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    __shared__ double ws[4];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[t] += (ws[0] + ws[1]) + (ws[2] + ws[3]);
}

__global__ void iota_kernel(int64_t *p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = (int64_t)i;
}

}  // namespace ttsk

using namespace ttsk;

static unsigned grid_for(size_t n, unsigned cap)
{
    size_t b = (n + 255) / 256;
    if (b < 1) b = 1;
    return (unsigned)(b > cap ? cap : b);
}

extern "C" {

int ttsk_sparse_ttdrm_step(const double *dev_vin, int64_t rho, const double *dev_core, int64_t n,
                           int64_t rhop, const int64_t *dev_idx_row, size_t N, double *dev_vout,
                           int stream)
{
    TTSK_STREAM(st, stream);
    TTSK_ARG(dev_core && dev_idx_row && dev_vout, "ttsk_sparse_ttdrm_step: NULL argument");
    TTSK_ARG(rho >= 1 && rhop >= 1 && n >= 1, "ttsk_sparse_ttdrm_step: bad core shape");
    TTSK_ARG(dev_vin || rho == 1, "ttsk_sparse_ttdrm_step: first mode needs rho == 1");
    if (N == 0) return TTSK_OK;
    hipLaunchKernelGGL(sparse_ttdrm_kernel, dim3(grid_for(N * (size_t)rhop, 1u << 20)), dim3(256), 0, st,
                       dev_vin, rho, dev_core, n, rhop, dev_idx_row, N, dev_vout);
    TTSK_LAUNCH_CHECK();
    return TTSK_OK;
}

int ttsk_sparse_densedrm_gather(const double *dev_mat, int64_t rank, int64_t cols, const int64_t *dev_idx,
                                int64_t row_stride, const int *row_order, const int64_t *shape, int m,
                                size_t N, double *dev_out, int stream)
{
    TTSK_STREAM(st, stream);
    TTSK_ARG(m >= 1 && m <= MAX_MODES, "ttsk_sparse_densedrm_gather: %d index rows unsupported", m);
    RavelMap rm;
    rm.m = m;
    rm.row_stride = row_stride;
    int64_t mult = 1;
    for (int i = m - 1; i >= 0; --i) {
        rm.mult[i] = mult;
        mult *= shape[i];
        rm.row_order[i] = row_order ? row_order[i] : i;
    }
    TTSK_ARG(mult == cols, "ttsk_sparse_densedrm_gather: matrix has %lld columns, index space %lld",
             (long long)cols, (long long)mult);
    if (N == 0 || rank == 0) return TTSK_OK;
    hipLaunchKernelGGL(sparse_dense_gather_kernel, dim3(grid_for(N * (size_t)rank, 1u << 20)), dim3(256), 0,
                       st, dev_mat, rank, cols, dev_idx, rm, N, dev_out);
    TTSK_LAUNCH_CHECK();
    return TTSK_OK;
}

int ttsk_sparse_sort_mode(const int64_t *dev_idx_row, size_t N, int64_t n, int64_t *dev_perm, int stream)
{
    TTSK_STREAM(st, stream);
    TTSK_ARG(dev_idx_row && dev_perm, "ttsk_sparse_sort_mode: NULL argument");
    if (N == 0) return TTSK_OK;
    int bits = 1;
    while ((1ll << bits) < n && bits < 62) ++bits;
    // scratch: sorted keys (discarded), identity values, cub temp storage
    size_t temp_bytes = 0;
    TTSK_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, (const int64_t *)nullptr, (int64_t *)nullptr,
                                                (const int64_t *)nullptr, (int64_t *)nullptr, (int)N, 0, bits, st));
    char *ws = (char *)scratch(stream, SCRATCH_MISC, 2 * N * 8 + temp_bytes + 256);
    if (!ws) return TTSK_ERR_HIP;
    int64_t *keys_out = (int64_t *)ws, *vals_in = (int64_t *)(ws + N * 8);
    void *temp = ws + 2 * N * 8;
    TTSK_ARG(N < (1ull << 31), "ttsk_sparse_sort_mode: more than 2^31 nonzeros");
    hipLaunchKernelGGL(iota_kernel, dim3(grid_for(N, 1u << 16)), dim3(256), 0, st, vals_in, N);
    TTSK_LAUNCH_CHECK();
    TTSK_HIP(hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, dev_idx_row, keys_out, vals_in, dev_perm, (int)N, 0,
                                                bits, st));
    return TTSK_OK;
}

int ttsk_sparse_psi(const double *dev_val, const int64_t *dev_idx_row, const int64_t *dev_perm, size_t N,
                    const double *dev_Lv, int64_t l, const double *dev_Rv, int64_t r, int64_t n, double *dev_psi,
                    int stream)
{
    TTSK_STREAM(st, stream);
    TTSK_ARG(dev_val && dev_psi, "ttsk_sparse_psi: NULL argument");
    TTSK_ARG(l >= 1 && r >= 1 && n >= 1, "ttsk_sparse_psi: bad shape");
    TTSK_ARG(dev_idx_row || n == 1, "ttsk_sparse_psi: a NULL index row means a single slice (n = 1)");
    if (N == 0) return TTSK_OK;
    struct ProfScope {      // device time of the segmented sum (work unit: algorithmic bytes = panels + values + indices)
        hipStream_t st; bool on;
        ProfScope(hipStream_t s, double bytes) : st(s), on(prof_on()) { if (on) prof_open_named(st, PROF_SPARSE, bytes, "sparse_psi_mfma_kernel"); }
        ~ProfScope() { if (on) prof_close(st); }
    } prof_scope(st, 8.0 * (double)N * (double)((dev_Lv ? l : 0) + (dev_Rv ? r : 0) + 1 + (dev_idx_row ? 1 : 0) + (dev_perm ? 1 : 0)));
    // MFMA kernel: small ranks, and either one slice or mode-sorted input with long slices
    static int mfma_on = [] { const char *e = getenv("TTSK_SPARSE_MFMA"); return e ? atoi(e) : 1; }();
    const bool single = dev_idx_row == nullptr || n == 1;
    if (mfma_on && l <= 32 && r <= 32 && N < (1ull << 40) && (single || (dev_perm && N / (size_t)n >= 32))) {
        const int tl = l <= 16 ? 1 : 2, tr = r <= 16 ? 1 : 2;
        // waves: ~8 per SIMD for the scatter case, one per SIMD slot pair when every wave leaves a partial block
        const size_t want = single ? 2048 : 16384;
        size_t chunk = ((N + want - 1) / want + 3) & ~(size_t)3;
        if (chunk < 256) chunk = 256;
        const size_t waves = (N + chunk - 1) / chunk, blocks = (waves + 3) / 4;
        double *part = nullptr;
        if (single) {
            part = (double *)scratch(stream, SCRATCH_MISC, waves * (size_t)(l * r) * 8);
            if (!part) return TTSK_ERR_HIP;
        }
        const int64_t *idxp = single ? nullptr : dev_idx_row;
#define TTSK_PSI_GO(TL, TR)                                                                                              \
        do {                                                                                                           \
            if (dev_perm) hipLaunchKernelGGL((sparse_psi_mfma_kernel<TL, TR, true>), dim3((unsigned)blocks), dim3(256), 0, st, \
                                             dev_val, idxp, dev_perm, N, dev_Lv, (int)l, dev_Rv, (int)r, n, dev_psi, chunk, part); \
            else hipLaunchKernelGGL((sparse_psi_mfma_kernel<TL, TR, false>), dim3((unsigned)blocks), dim3(256), 0, st,  \
                                    dev_val, idxp, dev_perm, N, dev_Lv, (int)l, dev_Rv, (int)r, n, dev_psi, chunk, part); \
        } while (0)
        if (tl == 1 && tr == 1) TTSK_PSI_GO(1, 1);
        else if (tl == 1) TTSK_PSI_GO(1, 2);
        else if (tr == 1) TTSK_PSI_GO(2, 1);
        else TTSK_PSI_GO(2, 2);
#undef TTSK_PSI_GO
        TTSK_LAUNCH_CHECK();
        if (single) {
            const int lr = (int)(l * r);
            hipLaunchKernelGGL(sparse_part_reduce_kernel, dim3((unsigned)lr), dim3(256), 0, st, part, waves, lr, dev_psi);
            TTSK_LAUNCH_CHECK();
        }
        return TTSK_OK;
    }
    const size_t lds = (size_t)64 * (l + r + 1) * 8;
    TTSK_ARG(lds <= 64 * 1024, "ttsk_sparse_psi: l + r = %lld too large for the staging buffer",
             (long long)(l + r));
    size_t chunk = 4096;
    size_t blocks = (N + chunk - 1) / chunk;
    hipLaunchKernelGGL(sparse_psi_kernel, dim3((unsigned)blocks), dim3(256), lds, st, dev_val, dev_idx_row, dev_perm,
                       N, dev_Lv, l, dev_Rv, r, n, dev_psi, chunk);
    TTSK_LAUNCH_CHECK();
    return TTSK_OK;
}

}  // extern "C"

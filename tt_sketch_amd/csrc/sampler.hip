// Hash-based lazy sampler on the device: 64-bit mix -> forced-exponent double ->
// inverse normal CDF, and sparse-sign rows.  Replaces the reference's only native
// module, tt_sketch/drm/fast_lazy_gaussian.pyx (hash :13-37, rand double :52-105,
// normal :39-50/:183-202, sparse sign :121-180).  Integer stages are bit-exact;
// the Gaussian stage follows the Cephes ndtri algorithm that the reference calls
// through SciPy (pyx:49) and agrees with it to a few ulp (device log/sqrt).
#include <cmath>
#include <vector>
#include "common.h"
#include "sampler_dev.h"

namespace ttsk {

// ndtri over a workgroup's 256 samples.  27 % of uniform samples fall in the tails of ndtri (two
// software logs, a square root, three divisions), so practically every wave would execute both
// branches for all its lanes.  Instead the central branch is evaluated in place and the tail samples
// are compacted into an LDS queue and evaluated by the first threads of the workgroup: only full
// waves of tail samples pay for the tail code.  Arithmetic per sample is unchanged (bit-identical).
// All 256 threads must call this together; `live` = this thread has a sample.
__device__ __forceinline__ void ndtri_block(bool live, double u, size_t g, double scale, double *__restrict__ out)
{
    __shared__ double q_u[256];
    __shared__ size_t q_g[256];
    __shared__ int q_n;
    const double expm2 = 0.13533528323661269189;
    if (threadIdx.x == 0) q_n = 0;
    __syncthreads();
    const bool central = live && u > expm2 && u <= 1.0 - expm2;
    if (central) {
        out[g] = scale * ndtri_dev(u);
    } else if (live) {
        const int slot = atomicAdd(&q_n, 1);
        q_u[slot] = u;
        q_g[slot] = g;
    }
    __syncthreads();
    const int n = q_n;
    if ((int)threadIdx.x < n) out[q_g[threadIdx.x]] = scale * ndtri_dev(q_u[threadIdx.x]);
    __syncthreads();
}

// The same split for a whole LDS tile of samples: the producer loop stores scale * ndtri(u) for central
// samples and parks the tail samples' u in their slot, queueing the slot number; after ONE barrier the
// queue is drained by all threads, full waves of tail code.  (ndtri_block above pays three barriers per
// 256 samples, i.e. per column of a row tile.)
__device__ __forceinline__ void tile_sample(double u, int slot, double scale, double *tile, unsigned short *q, int *qn)
{
    const double expm2 = 0.13533528323661269189;
    if (u > expm2 && u <= 1.0 - expm2) {
        tile[slot] = scale * ndtri_dev(u);
    } else {
        tile[slot] = u;
        q[atomicAdd(qn, 1)] = (unsigned short)slot;
    }
}
__device__ __forceinline__ void tile_drain(double scale, double *tile, const unsigned short *q, const int *qn)
{
    __syncthreads();
    const int n = *qn;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int slot = q[i];
        tile[slot] = scale * ndtri_dev(tile[slot]);
    }
    __syncthreads();
}

__global__ void hash_kernel(uint64_t *v, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x)
        v[i] = mix64(v[i]);
}

// mode 0: raw forced-exponent bits as double, mode 1: N(0,1)
template <int MODE>
__global__ void sample_kernel(const int64_t *__restrict__ idx, IndexMap im, size_t N, int rank_min,
                              int rank, uint64_t seed, double *__restrict__ out)
{
    const size_t tot = N * (size_t)rank;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    // uniform trip count per workgroup (ndtri_block has barriers)
    for (size_t g0 = (size_t)blockIdx.x * blockDim.x; g0 < tot; g0 += stride) {
        const size_t g = g0 + threadIdx.x;
        const bool live = g < tot;
        uint64_t bits = 0;
        if (live) {
            size_t e = g / rank;
            int j = (int)(g - e * rank);
            bits = rand_bits(flat_index(idx, im, e), rank_min + j, seed);
        }
        if (MODE == 0) {
            if (live) out[g] = __longlong_as_double(bits);
        } else {
            ndtri_block(live, live ? mant_unit(bits) : 0.5, g, 1.0, out);
        }
    }
}

// Same samples, one thread per index row (rank <= 32): the flat index and the column salts are
// computed once per row / per workgroup instead of once per sample, and there is no 64-bit division;
// the 256 x rank tile goes through LDS so that the store is contiguous.  ~2.5x fewer integer
// instructions per sample than sample_kernel, which remains for wider ranks.
template <int MODE>
__global__ __launch_bounds__(256) void sample_rows_kernel(const int64_t *__restrict__ idx, IndexMap im, size_t N,
                                                          int rank_min, int rank, uint64_t seed,
                                                          double *__restrict__ out)
{
    extern __shared__ double tile[];                         // [256][rank] + rank salts + tail queue
    uint64_t *salt = (uint64_t *)(tile + 256 * rank);
    unsigned short *tq = (unsigned short *)(salt + rank);    // [256 * rank] slots of tail samples
    __shared__ int tq_n;
    const int tid = threadIdx.x;
    if (tid < rank) salt[tid] = mix64((uint64_t)(rank_min + tid)) + seed;
    __syncthreads();
    for (size_t e0 = (size_t)blockIdx.x * 256; e0 < N; e0 += (size_t)gridDim.x * 256) {
        const size_t e = e0 + tid;
        const bool live = e < N;
        const uint64_t flat = live ? (idx ? flat_index(idx, im, e) : (uint64_t)e) : 0;   // no index rows: row e IS the flat index
        if (tid == 0) tq_n = 0;
        __syncthreads();
        for (int j = 0; j < rank; ++j) {
            const uint64_t h = mix64(flat + salt[j]);
            const uint64_t bits = (h | 0x2000000000000000ULL) & 0x3FFFFFFFFFFFFFFFULL;
            if (MODE == 0) tile[tid * rank + j] = __longlong_as_double(bits);
            else if (live) tile_sample(mant_unit(bits), tid * rank + j, 1.0, tile, tq, &tq_n);
        }
        if (MODE == 0) __syncthreads();
        else tile_drain(1.0, tile, tq, &tq_n);
        const size_t cnt = (N - e0 < 256 ? N - e0 : 256) * (size_t)rank;
        for (size_t t = tid; t < cnt; t += 256) out[e0 * rank + t] = tile[t];
        __syncthreads();
    }
}

// One thread per index row.  The full-length row lives in a scratch plane laid out
// [position][row] so that neighbouring threads touch neighbouring bytes.
template <typename OUT>
__global__ void sign_kernel(const int64_t *__restrict__ idx, IndexMap im, size_t N, int rank, int nnz,
                            int rank_min, int rank_max, uint64_t seed, int8_t *__restrict__ ws,
                            OUT *__restrict__ out)
{
    size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N) return;
    const uint64_t flat = idx ? flat_index(idx, im, e) : (uint64_t)e;   // no index rows: row e IS the flat index
    for (int j = 0; j < rank; ++j) ws[(size_t)j * N + e] = 0;
    for (int j = 0; j < nnz; ++j) {
        uint64_t bits = rand_bits(flat, j, seed);
        int ex = (int)((bits >> 52) & 0x7FF) - 1022;  // frexp exponent
        int par = ((ex % 2) + 2) % 2;                 // Python-style modulo (pyx:145)
        ws[(size_t)j * N + e] = (int8_t)(par * 2 - 1);
    }
    for (int j = 0; j < nnz; ++j) {
        double u = mant_unit(rand_bits(flat, j, seed));
        int pick = (int)(u * (double)(rank - j) + (double)j);
        int8_t a = ws[(size_t)j * N + e], b = ws[(size_t)pick * N + e];
        ws[(size_t)j * N + e] = b;
        ws[(size_t)pick * N + e] = a;
    }
    const int w = rank_max - rank_min;
    for (int j = 0; j < w; ++j) out[e * (size_t)w + j] = (OUT)ws[(size_t)(rank_min + j) * N + e];
}

__device__ __forceinline__ void fill_normal_body(double *out, size_t n, uint64_t key, double scale)
{
    constexpr int PER = 8;                                   // samples per thread and tile
    __shared__ double tile[256 * PER];
    __shared__ unsigned short tq[256 * PER];
    __shared__ int tq_n;
    const size_t stride = (size_t)gridDim.x * 256 * PER;
    for (size_t i0 = (size_t)blockIdx.x * 256 * PER; i0 < n; i0 += stride) {
        if (threadIdx.x == 0) tq_n = 0;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int slot = threadIdx.x + 256 * k;
            const size_t i = i0 + slot;
            if (i < n) {
                uint64_t h = mix64((uint64_t)i + key);
                h = (h | 0x2000000000000000ULL) & 0x3FFFFFFFFFFFFFFFULL;
                double u = mant_unit(h);
                if (u == 0.0) u = 0x1p-53;
                tile_sample(u, slot, scale, tile, tq, &tq_n);
            }
        }
        tile_drain(scale, tile, tq, &tq_n);
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const size_t i = i0 + threadIdx.x + 256 * k;
            if (i < n) out[i] = tile[threadIdx.x + 256 * k];
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void fill_normal_kernel(double *out, size_t n, uint64_t key, double scale)
{
    fill_normal_body(out, n, key, scale);
}

// several independent fills in one launch (the d - 1 cores of a TensorTrainDRM): blockIdx.y picks the array
constexpr int FILL_MANY_MAX = 32;
struct FillMany {
    double *out[FILL_MANY_MAX];
    size_t n[FILL_MANY_MAX];
    uint64_t key[FILL_MANY_MAX];
    double scale[FILL_MANY_MAX];
};
__global__ __launch_bounds__(256) void fill_normal_many_kernel(FillMany a)
{
    const int seg = blockIdx.y;
    fill_normal_body(a.out[seg], a.n[seg], a.key[seg], a.scale[seg]);
}

// What the chip gives for hash + ndtri and nothing else (the ceiling the sampling passes of the sparse
// sketch are priced against, instead of an instruction-count estimate).  Same device code as the kernels (mix64, mant_unit,
// ndtri_dev), the same split they use -- central branch in place, tail samples queued per wave by ballot and evaluated a full
// wave at a time -- operands in registers / LDS, no global traffic but one checksum per lane; the 73 / 27 central / tail mix is
// the uniform distribution's.  MODE 1: every lane evaluates ndtri_dev as it comes (both branches under divergence).
constexpr int PROBE_COLS = 16;
template <int MODE>
__global__ __launch_bounds__(256) void ndtri_probe_kernel(double *sink, int reps, uint64_t seed)
{
    __shared__ double tile[4][64 * PROBE_COLS];
    __shared__ unsigned short tq[4][64 * PROBE_COLS];
    __shared__ uint64_t salt[PROBE_COLS];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid < PROBE_COLS) salt[tid] = mix64((uint64_t)tid) + seed;
    __syncthreads();
    double *T = tile[wv];
    unsigned short *q = tq[wv];
    const uint64_t row0 = ((uint64_t)blockIdx.x * 256 + tid) * (uint64_t)reps;
    const double expm2 = 0.13533528323661269189;
    double acc = 0.0;
    for (int r = 0; r < reps; ++r) {
        const uint64_t flat = row0 + r;
        int qn = 0;
#pragma unroll 4
        for (int c = 0; c < PROBE_COLS; ++c) {
            const uint64_t h = mix64(flat + salt[c]);
            const double u = mant_unit((h | 0x2000000000000000ULL) & 0x3FFFFFFFFFFFFFFFULL);
            if (MODE == 1) {
                acc += ndtri_dev(u);
            } else {
                const int slot = lane * PROBE_COLS + c;
                const bool central = u > expm2 && u <= 1.0 - expm2;
                if (central) T[slot] = ndtri_dev(u);
                const unsigned long long m = __ballot(!central);
                if (!central) {
                    T[slot] = u;
                    q[qn + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)slot;
                }
                qn += __popcll(m);
            }
        }
        if (MODE == 0) {
            __builtin_amdgcn_wave_barrier();
            for (int i = 0; i < qn; i += 64)
                if (i + lane < qn) {
                    const int slot = q[i + lane];
                    T[slot] = ndtri_dev(T[slot]);
                }
            __builtin_amdgcn_wave_barrier();
#pragma unroll 4
            for (int c = 0; c < PROBE_COLS; ++c) acc += T[lane * PROBE_COLS + c];
        }
    }
    sink[(size_t)blockIdx.x * 256 + tid] = acc;
}

static unsigned grid_for(size_t n, unsigned block = 256, unsigned cap = 16384)
{
    size_t b = (n + block - 1) / block;
    if (b < 1) b = 1;
    return (unsigned)(b > cap ? cap : b);
}

template <int MODE>
static void launch_sample(const int64_t *idx, const IndexMap &im, size_t N, int rank_min, int w, uint64_t seed,
                          double *out, hipStream_t st)
{
    if (w <= 32) {
        static PerInit attr_done;       // 256 x w tile + salts + tail queue: up to 82 KB at w = 32
        if (attr_done.first()) {
            (void)hipFuncSetAttribute((const void *)sample_rows_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        }
        size_t blocks = (N + 255) / 256;
        if (blocks > (1u << 16)) blocks = 1u << 16;
        hipLaunchKernelGGL((sample_rows_kernel<MODE>), dim3((unsigned)blocks), dim3(256), (size_t)(256 * w + w) * 8 + (size_t)256 * w * 2, st,
                           idx, im, N, rank_min, w, seed, out);
    } else {
        hipLaunchKernelGGL((sample_kernel<MODE>), dim3(grid_for(N * (size_t)w, 256, 1u << 20)), dim3(256), 0, st, idx,
                           im, N, rank_min, w, seed, out);
    }
}

// out[e, :] = table[flat index of row e, :]; a workgroup takes 256 rows at a time: their flat indices once into
// LDS, then the 256 x w tile is written contiguously
__global__ __launch_bounds__(256) void expand_rows_kernel(const int64_t *__restrict__ idx, IndexMap im, size_t N, int w,
                                                          const double *__restrict__ table, double *__restrict__ out)
{
    __shared__ unsigned flat_s[256];
    const int tid = threadIdx.x;
    const float inv_w = 1.0f / (float)w;
    for (size_t e0 = (size_t)blockIdx.x * 256; e0 < N; e0 += (size_t)gridDim.x * 256) {
        __syncthreads();
        flat_s[tid] = e0 + tid < N ? (unsigned)flat_index(idx, im, e0 + tid) : 0u;
        __syncthreads();
        const int cnt = (int)(N - e0 < 256 ? N - e0 : 256) * w;
        double *o = out + e0 * (size_t)w;
        for (int t = tid; t < cnt; t += 256) {
            const int r = (int)(((float)t + 0.5f) * inv_w);          // t / w for t < 2^13 (w <= 32)
            o[t] = table[(size_t)flat_s[r] * (unsigned)w + (unsigned)(t - r * w)];
        }
    }
}

template <typename OUT>
static int sign_dev(const int64_t *dev_idx, const IndexMap &im, size_t N, int true_rank, int rank_min,
                    int rank_max, int nnz, uint64_t seed, OUT *dev_out, hipStream_t st, int stream)
{
    TTSK_ARG(nnz >= 0 && nnz <= true_rank, "sparse sign: nnz_per_row %d not in [0, %d]", nnz, true_rank);
    TTSK_ARG(0 <= rank_min && rank_min <= rank_max && rank_max <= true_rank,
             "sparse sign: bad rank slice [%d,%d) of %d", rank_min, rank_max, true_rank);
    if (N == 0 || rank_max == rank_min) return TTSK_OK;
    int8_t *ws = (int8_t *)scratch(stream, SCRATCH_MISC, N * (size_t)true_rank);
    if (!ws) return TTSK_ERR_HIP;
    hipLaunchKernelGGL((sign_kernel<OUT>), dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, dev_idx, im,
                       N, true_rank, nnz, rank_min, rank_max, seed, ws, dev_out);
    TTSK_LAUNCH_CHECK();
    return TTSK_OK;
}

}  // namespace ttsk

using namespace ttsk;

extern "C" {

// Gaussian samples per second of the hash + ndtri code on this device: [0] with the kernels' central / tail split, [1] every
// lane through both branches (what a plain per-lane ndtri costs under divergence).  G samples / s.
int ttsk_ndtri_rate_probe(double *gsamples)
{
    TTSK_STREAM(st, 0);
    TTSK_ARG(gsamples, "ttsk_ndtri_rate_probe: NULL");
    const int blocks = 256 * 8, reps = 256;
    double *sink = (double *)scratch(0, SCRATCH_MISC, (size_t)blocks * 256 * 8);
    if (!sink) return TTSK_ERR_HIP;
    hipEvent_t a, b;
    TTSK_HIP(hipEventCreate(&a));
    TTSK_HIP(hipEventCreate(&b));
    for (int mode = 0; mode < 2; ++mode) {
        double best = 0;
        for (int rep = 0; rep < 4; ++rep) {
            TTSK_HIP(hipEventRecord(a, st));
            if (mode == 0) hipLaunchKernelGGL((ndtri_probe_kernel<0>), dim3(blocks), dim3(256), 0, st, sink, reps, (uint64_t)(17 + rep));
            else hipLaunchKernelGGL((ndtri_probe_kernel<1>), dim3(blocks), dim3(256), 0, st, sink, reps, (uint64_t)(17 + rep));
            TTSK_HIP(hipEventRecord(b, st));
            TTSK_HIP(hipEventSynchronize(b));
            float ms = 0;
            TTSK_HIP(hipEventElapsedTime(&ms, a, b));
            const double rate = (double)blocks * 256 * reps * PROBE_COLS / (ms * 1e-3) * 1e-9;
            if (rep > 0 && rate > best) best = rate;
        }
        gsamples[mode] = best;
    }
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    return TTSK_OK;
}

int ttsk_hash_u64(uint64_t *host_vals, size_t n)
{
    TTSK_STREAM(st, 0);
    if (n == 0) return TTSK_OK;
    TTSK_ARG(host_vals, "ttsk_hash_u64: NULL");
    uint64_t *d = nullptr;
    TTSK_HIP(hipMalloc((void **)&d, n * 8));
    hipError_t e = hipMemcpyAsync(d, host_vals, n * 8, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(hash_kernel, dim3(grid_for(n)), dim3(256), 0, st, d, n);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(host_vals, d, n * 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(d);
    TTSK_HIP(e);
    return TTSK_OK;
}

int ttsk_sparse_normal_dev(const int64_t *dev_idx, int64_t row_stride, const int *row_order,
                           const uint64_t *shape, int m, size_t N, int rank_min, int rank_max,
                           uint64_t seed, double *dev_out, int stream)
{
    TTSK_STREAM(st, stream);
    TTSK_ARG(rank_max >= rank_min && rank_min >= 0, "bad rank slice [%d,%d)", rank_min, rank_max);
    IndexMap im;
    int rc = make_index_map(shape, m, row_stride, row_order, &im);
    if (rc) return rc;
    size_t tot = N * (size_t)(rank_max - rank_min);
    if (tot == 0) return TTSK_OK;
    // Few distinct prefixes (the first modes of a long COO list: 200 or 30 000 values under 10^7 nonzeros): a sample is a
    // function of (flat prefix index, column, seed) only, so every possible prefix is sampled once and the rows are
    // copied out -- 0.25 instead of 0.78 ms per mode at C4.  Only without the reference's 32-bit wrap of the
    // multipliers (fast_lazy_gaussian.pyx:60-71), i.e. when the flat index really is < prod(shape).
    static const int dedupe = [] { const char *e = getenv("TTSK_SPARSE_DEDUPE"); return e ? atoi(e) : 1; }();
    const int w = rank_max - rank_min;
    struct ProfScope {      // device time of the whole sampling pass (work unit: Gaussian samples delivered)
        hipStream_t st; bool on;
        ProfScope(hipStream_t s, double samples) : st(s), on(prof_on()) { if (on) prof_open_named(st, PROF_SAMPLER, samples, "sample_rows_kernel / expand_rows_kernel"); }
        ~ProfScope() { if (on) prof_close(st); }
    } prof_scope(st, (double)tot);
    double prod = 1.0;
    for (int i = 0; i < m; ++i) prod *= (double)shape[i];
    if (dedupe && w <= 32 && prod * 4.0 <= (double)N && prod < 16777216.0) {
        const size_t P = (size_t)prod;
        double *table = (double *)scratch(stream, SCRATCH_MISC, P * (size_t)w * 8);
        if (!table) return TTSK_ERR_HIP;
        launch_sample<1>(nullptr, im, P, rank_min, w, seed, table, st);
        TTSK_LAUNCH_CHECK();
        hipLaunchKernelGGL(expand_rows_kernel, dim3(grid_for(N, 256, 1u << 15)), dim3(256), 0, st, dev_idx, im, N, w, table,
                           dev_out);
        TTSK_LAUNCH_CHECK();
        return TTSK_OK;
    }
    launch_sample<1>(dev_idx, im, N, rank_min, w, seed, dev_out, st);
    TTSK_LAUNCH_CHECK();
    return TTSK_OK;
}

// The samples of EVERY possible index prefix: row f of the table is what ttsk_sparse_normal_dev returns for an index
// row whose flat index (fast_lazy_gaussian.pyx:60-71) is f.  Only where that flat index is a bijection onto
// [0, prod(shape)): no 32-bit wrap of the running product (TTSK_ERR_UNSUPPORTED otherwise).
int ttsk_sparse_normal_table(const uint64_t *shape, int m, int rank_min, int rank_max, uint64_t seed, double *dev_out, int stream)
{
    TTSK_STREAM(st, stream);
    TTSK_ARG(shape && dev_out && rank_max > rank_min && rank_min >= 0, "ttsk_sparse_normal_table: bad argument");
    IndexMap im;
    int rc = make_index_map(shape, m, 0, nullptr, &im);
    if (rc) return rc;
    double prod = 1.0;
    for (int i = 0; i < m; ++i) prod *= (double)shape[i];
    if (!(prod < 2147483648.0)) {
        set_error("ttsk_sparse_normal_table: %g prefixes: the reference's 32-bit running product wraps", prod);
        return TTSK_ERR_UNSUPPORTED;
    }
    const bool prof = prof_on();
    if (prof) prof_open_named(st, PROF_SAMPLER, prod * (rank_max - rank_min), "sample_rows_kernel (prefix table)");
    launch_sample<1>(nullptr, im, (size_t)prod, rank_min, rank_max - rank_min, seed, dev_out, st);
    if (prof) prof_close(st);
    TTSK_LAUNCH_CHECK();
    return TTSK_OK;
}

int ttsk_sparse_sign_dev(const int64_t *dev_idx, int64_t row_stride, const int *row_order,
                         const uint64_t *shape, int m, size_t N, int true_rank, int rank_min,
                         int rank_max, int nnz_per_row, uint64_t seed, double *dev_out, int stream)
{
    TTSK_STREAM(st, stream);
    IndexMap im;
    int rc = make_index_map(shape, m, row_stride, row_order, &im);
    if (rc) return rc;
    return sign_dev<double>(dev_idx, im, N, true_rank, rank_min, rank_max, nnz_per_row, seed, dev_out, st, stream);
}

int ttsk_sparse_sign_table(const uint64_t *shape, int m, int true_rank, int rank_min, int rank_max, int nnz_per_row, uint64_t seed,
                           double *dev_out, int stream)
{
    TTSK_STREAM(st, stream);
    TTSK_ARG(shape && dev_out, "ttsk_sparse_sign_table: NULL argument");
    IndexMap im;
    int rc = make_index_map(shape, m, 0, nullptr, &im);
    if (rc) return rc;
    double prod = 1.0;
    for (int i = 0; i < m; ++i) prod *= (double)shape[i];
    if (!(prod < 2147483648.0)) {
        set_error("ttsk_sparse_sign_table: %g prefixes: the reference's 32-bit running product wraps", prod);
        return TTSK_ERR_UNSUPPORTED;
    }
    return sign_dev<double>(nullptr, im, (size_t)prod, true_rank, rank_min, rank_max, nnz_per_row, seed, dev_out, st, stream);
}

static int host_sample(const void *host_idx, const uint64_t *shape, int m, size_t N, int rank_min,
                       int rank_max, uint64_t seed, void *host_out, int mode, int true_rank, int nnz)
{
    TTSK_STREAM(st, 0);
    TTSK_ARG(rank_max >= rank_min && rank_min >= 0, "bad rank slice [%d,%d)", rank_min, rank_max);
    IndexMap im;
    int rc = make_index_map(shape, m, (int64_t)N, nullptr, &im);
    if (rc) return rc;
    const int w = rank_max - rank_min;
    const size_t tot = N * (size_t)w;
    if (tot == 0) return TTSK_OK;
    TTSK_ARG(host_idx && host_out, "hash sampler: NULL buffer");
    int64_t *didx = nullptr;
    void *dout = nullptr;
    const size_t out_bytes = tot * (mode == 2 ? sizeof(int16_t) : sizeof(double));
    TTSK_HIP(hipMalloc((void **)&didx, (size_t)m * N * 8));
    hipError_t e = hipMalloc(&dout, out_bytes);
    if (e == hipSuccess) e = hipMemcpyAsync(didx, host_idx, (size_t)m * N * 8, hipMemcpyHostToDevice, st);
    int status = TTSK_OK;
    if (e == hipSuccess) {
        if (mode == 0)
            launch_sample<0>(didx, im, N, rank_min, w, seed, (double *)dout, st);
        else if (mode == 1)
            launch_sample<1>(didx, im, N, rank_min, w, seed, (double *)dout, st);
        else
            status = sign_dev<int16_t>(didx, im, N, true_rank, rank_min, rank_max, nnz, seed, (int16_t *)dout, st, 0);
        e = hipGetLastError();
    }
    if (e == hipSuccess && status == TTSK_OK)
        e = hipMemcpyAsync(host_out, dout, out_bytes, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(didx);
    (void)hipFree(dout);
    if (status != TTSK_OK) return status;
    TTSK_HIP(e);
    return TTSK_OK;
}

int ttsk_inds_to_rand_double(const uint64_t *host_idx, const uint64_t *shape, int m, size_t N,
                             int rank_min, int rank_max, uint64_t seed, double *host_out)
{
    return host_sample(host_idx, shape, m, N, rank_min, rank_max, seed, host_out, 0, 0, 0);
}

int ttsk_inds_to_normal(const int64_t *host_idx, const uint64_t *shape, int m, size_t N, int rank_min,
                        int rank_max, uint64_t seed, double *host_out)
{
    return host_sample(host_idx, shape, m, N, rank_min, rank_max, seed, host_out, 1, 0, 0);
}

int ttsk_inds_to_sparse_sign(const int64_t *host_idx, const uint64_t *shape, int m, size_t N,
                             int true_rank, int rank_min, int rank_max, int nnz_per_row, uint64_t seed,
                             int16_t *host_out)
{
    return host_sample(host_idx, shape, m, N, rank_min, rank_max, seed, host_out, 2, true_rank,
                       nnz_per_row);
}

int ttsk_fill_normal(double *dev_out, size_t n, uint64_t seed, double scale, int stream)
{
    TTSK_STREAM(st, stream);
    if (n == 0) return TTSK_OK;
    uint64_t key = mix64(seed ^ 0x9E3779B97F4A7C15ULL);
    hipLaunchKernelGGL(fill_normal_kernel, dim3(grid_for(n, 2048, 1u << 16)), dim3(256), 0, st, dev_out, n,
                       key, scale);
    TTSK_LAUNCH_CHECK();
    return TTSK_OK;
}

int ttsk_fill_normal_many(int count, double *const *dev_outs, const size_t *ns, const uint64_t *seeds,
                          const double *scales, int stream)
{
    TTSK_STREAM(st, stream);
    TTSK_ARG(count >= 0 && (count == 0 || (dev_outs && ns && seeds && scales)), "ttsk_fill_normal_many: bad argument");
    for (int c0 = 0; c0 < count; c0 += FILL_MANY_MAX) {
        const int cnt = count - c0 < FILL_MANY_MAX ? count - c0 : FILL_MANY_MAX;
        FillMany a{};
        size_t nmax = 0;
        for (int i = 0; i < cnt; ++i) {
            a.out[i] = dev_outs[c0 + i];
            a.n[i] = ns[c0 + i];
            a.key[i] = mix64(seeds[c0 + i] ^ 0x9E3779B97F4A7C15ULL);      // as ttsk_fill_normal: the same samples
            a.scale[i] = scales[c0 + i];
            nmax = a.n[i] > nmax ? a.n[i] : nmax;
        }
        if (nmax == 0) continue;
        hipLaunchKernelGGL(fill_normal_many_kernel, dim3(grid_for(nmax, 2048, 1u << 14), (unsigned)cnt), dim3(256), 0, st, a);
        TTSK_LAUNCH_CHECK();
    }
    return TTSK_OK;
}

}  // extern "C"

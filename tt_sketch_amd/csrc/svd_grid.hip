// One-sided Jacobi SVD on the whole chip, for factors beyond the one-workgroup kernel of linalg.hip
// (n > 1024 columns): the classical TT-SVD of a dense tensor (reference tt_svd.py:10-49 takes LAPACK's
// SVD of every unfolding) meets (r n) x (r n) triangular factors with r n in the thousands.
//
// Same algorithm as jacobi_pinv_kernel: round-robin ordering, n - 1 rounds of n / 2 disjoint column pairs
// per sweep, dgesvj's stopping rule, noise-level pairs skipped.  Here a pair belongs to one wavefront, the
// pairs of a round are spread over all workgroups, W (m x n, column-major) and V (n x n) live in global
// memory, and rounds are separated by a grid-wide barrier (atomic counter + generation; agent-scope fences
// write the XCD-local L2 back and drop stale lines, so a column rotated on one XCD is seen by the wavefront
// that pairs it next on another).  All workgroups are resident (grid <= what the occupancy query allows) and
// every wait is bounded: a barrier that does not complete within ~1 s raises `abort` and every workgroup leaves.
#include <cfloat>
#include <algorithm>
#include <numeric>
#include <vector>
#include "common.h"

namespace ttsk {

struct GridCtl {
    unsigned count, gen;
    int abort_, sweeps;
    int rot[3];
    int pad_;
    unsigned long long smax_bits[3];
};

#define TTSK_AGENT __HIP_MEMORY_SCOPE_AGENT

__device__ __forceinline__ bool grid_sync(GridCtl *c, unsigned nblocks, int *s_ok)
{
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        int ok = 1;
        const unsigned g = __hip_atomic_load(&c->gen, __ATOMIC_RELAXED, TTSK_AGENT);
        if (__hip_atomic_fetch_add(&c->count, 1u, __ATOMIC_ACQ_REL, TTSK_AGENT) == nblocks - 1) {
            __hip_atomic_store(&c->count, 0u, __ATOMIC_RELAXED, TTSK_AGENT);
            __hip_atomic_fetch_add(&c->gen, 1u, __ATOMIC_RELEASE, TTSK_AGENT);
        } else {
            long spins = 0;
            while (__hip_atomic_load(&c->gen, __ATOMIC_ACQUIRE, TTSK_AGENT) == g) {
                __builtin_amdgcn_s_sleep(8);
                if (++spins > 2000000 || __hip_atomic_load(&c->abort_, __ATOMIC_RELAXED, TTSK_AGENT)) {
                    __hip_atomic_store(&c->abort_, 1, __ATOMIC_RELAXED, TTSK_AGENT);
                    ok = 0;
                    break;
                }
            }
        }
        if (__hip_atomic_load(&c->abort_, __ATOMIC_RELAXED, TTSK_AGENT)) ok = 0;
        *s_ok = ok;
    }
    __syncthreads();
    __threadfence();
    return *s_ok != 0;
}

__device__ __forceinline__ double wave_sum(double x)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    return x;
}

// One pair step by a wavefront.  IT > 0: columns of <= 64 IT rows stay in registers between the inner
// products and the rotation.
template <int IT>
__device__ __forceinline__ void wave_pair(double *wp, double *wq, double *vp, double *vq, const int m, const int n,
                                          const int lane, const double tol2, const double tiny2, int *rot)
{
    constexpr int ITC = IT ? IT : 1;
    double x[ITC], y[ITC];
    double a = 0, b = 0, g = 0;
    if constexpr (IT > 0) {
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int i = lane + 64 * it;
            x[it] = i < m ? wp[i] : 0.0;
            y[it] = i < m ? wq[i] : 0.0;
        }
#pragma unroll
        for (int it = 0; it < IT; ++it) { a = fma(x[it], x[it], a); b = fma(y[it], y[it], b); g = fma(x[it], y[it], g); }
    } else {
        for (int i = lane; i < m; i += 64) {
            const double xx = wp[i], yy = wq[i];
            a = fma(xx, xx, a); b = fma(yy, yy, b); g = fma(xx, yy, g);
        }
    }
    a = wave_sum(a); b = wave_sum(b); g = wave_sum(g);
    if (g * g <= tol2 * (a * b) || g == 0.0 || (a <= tiny2 && b <= tiny2)) return;
    if (lane == 0) __hip_atomic_store(rot, 1, __ATOMIC_RELAXED, TTSK_AGENT);
    const double zeta = (b - a) / (2.0 * g);
    const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
    const double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
    if constexpr (IT > 0) {
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int i = lane + 64 * it;
            if (i < m) { wp[i] = cs * x[it] - sn * y[it]; wq[i] = sn * x[it] + cs * y[it]; }
        }
    } else {
        for (int i = lane; i < m; i += 64) {
            const double xx = wp[i], yy = wq[i];
            wp[i] = cs * xx - sn * yy; wq[i] = sn * xx + cs * yy;
        }
    }
    for (int i = lane; i < n; i += 64) {
        const double xx = vp[i], yy = vq[i];
        vp[i] = cs * xx - sn * yy; vq[i] = sn * xx + cs * yy;
    }
}

template <int IT>
__global__ __launch_bounds__(256) void svd_grid_kernel(double *W, double *V, const int m, const int n, GridCtl *c,
                                                       double *sig2, const int max_sweeps)
{
    __shared__ int s_ok;
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
    const int np = n + (n & 1), nm1 = np - 1;
    const double tol = fmax(4.0, sqrt((double)m)) * DBL_EPSILON, tol2 = tol * tol;
    int sweep = 0;
    for (; sweep < max_sweeps; ++sweep) {
        const int s0 = sweep % 3, s1 = (sweep + 1) % 3;
        if (blockIdx.x == 0 && threadIdx.x == 0) {          // the next sweep's slots; nobody touches them during this one
            __hip_atomic_store(&c->rot[s1], 0, __ATOMIC_RELAXED, TTSK_AGENT);
            __hip_atomic_store(&c->smax_bits[s1], 0ull, __ATOMIC_RELAXED, TTSK_AGENT);
        }
        double mx = 0;
        for (int j = gw; j < n; j += nw) {
            const double *wj = W + (size_t)j * m;
            double a = 0;
            for (int i = lane; i < m; i += 64) a = fma(wj[i], wj[i], a);
            a = wave_sum(a);
            mx = fmax(mx, a);
        }
        if (lane == 0 && mx > 0)     // non-negative doubles order like their bit patterns
            __hip_atomic_fetch_max(&c->smax_bits[s0], (unsigned long long)__double_as_longlong(mx), __ATOMIC_RELAXED, TTSK_AGENT);
        if (!grid_sync(c, gridDim.x, &s_ok)) return;
        const double smax = __longlong_as_double((long long)__hip_atomic_load(&c->smax_bits[s0], __ATOMIC_RELAXED, TTSK_AGENT));
        const double tiny = 4.0 * m * DBL_EPSILON, tiny2 = tiny * tiny * smax;
        for (int round = 0; round < nm1; ++round) {
            for (int pi = gw; pi < np / 2; pi += nw) {
                int p, q;
                if (pi == 0) { p = nm1; q = round; }
                else {
                    p = round + pi; p -= p >= nm1 ? nm1 : 0;
                    q = round + nm1 - pi; q -= q >= nm1 ? nm1 : 0;
                }
                if (p >= n || q >= n) continue;
                if (p > q) { const int t = p; p = q; q = t; }
                wave_pair<IT>(W + (size_t)p * m, W + (size_t)q * m, V + (size_t)p * n, V + (size_t)q * n, m, n, lane,
                              tol2, tiny2, &c->rot[s0]);
            }
            if (!grid_sync(c, gridDim.x, &s_ok)) return;
        }
        if (!__hip_atomic_load(&c->rot[s0], __ATOMIC_RELAXED, TTSK_AGENT)) { ++sweep; break; }
    }
    for (int j = gw; j < n; j += nw) {
        const double *wj = W + (size_t)j * m;
        double a = 0;
        for (int i = lane; i < m; i += 64) a = fma(wj[i], wj[i], a);
        a = wave_sum(a);
        if (lane == 0) sig2[j] = a;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) c->sweeps = sweep;
}

// W[j m + i] = A[i n + j] (32 x 32 tiles through LDS), V = I
__global__ __launch_bounds__(256) void svd_grid_load_kernel(const double *__restrict__ A, int m, int n, double *W, double *V)
{
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
    for (int r = ty; r < 32; r += 8) {
        const int i = i0 + r, j = j0 + tx;
        tile[r][tx] = (i < m && j < n) ? A[(size_t)i * n + j] : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int j = j0 + r, i = i0 + tx;
        if (i < m && j < n) W[(size_t)j * m + i] = tile[tx][r];
    }
    if (blockIdx.y * 32 < n)
        for (int r = ty; r < 32; r += 8) {
            const int i = i0 + r, j = j0 + tx;
            if (i < n && j < n) V[(size_t)j * n + i] = i == j ? 1.0 : 0.0;
        }
}

// US[i n + k] = W[ord[k] m + i], Vt[k n + i] = V[ord[k] n + i], S[k] = sqrt(sig2[ord[k]])
__global__ __launch_bounds__(256) void svd_grid_store_kernel(const double *__restrict__ W, const double *__restrict__ V,
                                                             const double *__restrict__ sig2, const int *__restrict__ ord,
                                                             int m, int n, double *US, double *S, double *Vt)
{
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int i0 = blockIdx.y * 32, k0 = blockIdx.x * 32;
    for (int r = ty; r < 32; r += 8) {
        const int k = k0 + r, i = i0 + tx;
        tile[r][tx] = (k < n && i < m) ? W[(size_t)ord[k] * m + i] : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int i = i0 + r, k = k0 + tx;
        if (i < m && k < n) US[(size_t)i * n + k] = tile[tx][r];
    }
    if (i0 < n)
        for (int r = ty; r < 32; r += 8) {
            const int k = k0 + r, i = i0 + tx;
            if (k < n && i < n) Vt[(size_t)k * n + i] = V[(size_t)ord[k] * n + i];
        }
    if (blockIdx.y == 0 && threadIdx.x < 32) {
        const int k = k0 + threadIdx.x;
        if (k < n) S[k] = sqrt(sig2[ord[k]]);
    }
}

// A (m, n) row-major, m >= n: US (m, n) = U diag(S), S (n) descending, Vt (n, n).  Blocks the host once (the
// column order is an argsort of n numbers on the host).
int svd_jacobi_grid(const double *A, int64_t m, int64_t n, double *US, double *S, double *Vt, int stream, hipStream_t st)
{
    const size_t wv = (size_t)(m * n + n * n);
    const size_t bytes = (wv + (size_t)n) * 8 + (size_t)n * 4 + 256;
    double *ws = (double *)scratch(stream, SCRATCH_MISC, bytes);
    if (!ws) return TTSK_ERR_HIP;
    double *W = ws, *V = ws + m * n, *sig2 = ws + wv;
    int *ord = (int *)(sig2 + n);
    GridCtl *ctl = (GridCtl *)(((uintptr_t)(ord + n) + 63) & ~(uintptr_t)63);
    const bool regs = m <= 2048;
    auto kern = regs ? svd_grid_kernel<32> : svd_grid_kernel<0>;
    static int cus = 0, occ_r = 0, occ_g = 0;
    if (!cus) {
        int dev = 0;
        TTSK_HIP(hipGetDevice(&dev));
        TTSK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        TTSK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_r, svd_grid_kernel<32>, 256, 0));
        TTSK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_g, svd_grid_kernel<0>, 256, 0));
    }
    const int occ = regs ? occ_r : occ_g;
    TTSK_ARG(occ >= 1, "svd: the grid kernel does not fit a compute unit");
    const int64_t want = cdiv(cdiv(n, 2), 4);                      // one wavefront per pair of a round
    const int64_t cap = (int64_t)cus * (occ < 4 ? occ : 4);
    const int grid = (int)(want < cap ? want : cap);
    // two barrier kernels in flight on different streams could each hold part of the chip and wait for the rest:
    // one at a time (the second starts when the first has drained)
    static hipEvent_t last = nullptr;
    if (!last) TTSK_HIP(hipEventCreateWithFlags(&last, hipEventDisableTiming));
    else TTSK_HIP(hipStreamWaitEvent(st, last, 0));
    // ... and whatever the OTHER library streams have queued so far is drained first: the barrier needs every
    // workgroup resident, compute units held by a neighbour's kernel would make it wait
    static hipEvent_t busy[TTSK_NUM_STREAMS] = {};
    for (int i = 0; i < TTSK_NUM_STREAMS; ++i) {
        hipStream_t other = stream_of(i);
        if (!other || other == st) continue;
        if (!busy[i]) TTSK_HIP(hipEventCreateWithFlags(&busy[i], hipEventDisableTiming));
        TTSK_HIP(hipEventRecord(busy[i], other));
        TTSK_HIP(hipStreamWaitEvent(st, busy[i], 0));
    }
    TTSK_HIP(hipMemsetAsync(ctl, 0, sizeof(GridCtl), st));
    hipLaunchKernelGGL(svd_grid_load_kernel, dim3((unsigned)cdiv(n, 32), (unsigned)cdiv(m, 32)), dim3(256), 0, st, A, (int)m,
                       (int)n, W, V);
    TTSK_LAUNCH_CHECK();
    {
        // cooperative launch: the runtime checks that the whole grid can be resident at once and refuses otherwise
        int mi = (int)m, ni = (int)n, sweeps = 60;
        void *kargs[] = {(void *)&W, (void *)&V, (void *)&mi, (void *)&ni, (void *)&ctl, (void *)&sig2, (void *)&sweeps};
        const hipError_t ce = hipLaunchCooperativeKernel((const void *)kern, dim3(grid), dim3(256), kargs, 0, st);
        if (ce != hipSuccess) {
            (void)hipGetLastError();
            set_error("svd: cooperative launch of the whole-chip Jacobi kernel refused (%s; grid %d): it needs every workgroup "
                      "resident, i.e. the GPU to itself", hipGetErrorString(ce), grid);
            return TTSK_ERR_HIP;
        }
    }
    TTSK_HIP(hipEventRecord(last, st));
    std::vector<double> h((size_t)n);
    GridCtl hc;
    TTSK_HIP(hipMemcpyAsync(h.data(), sig2, (size_t)n * 8, hipMemcpyDeviceToHost, st));
    TTSK_HIP(hipMemcpyAsync(&hc, ctl, sizeof(GridCtl), hipMemcpyDeviceToHost, st));
    TTSK_HIP(hipStreamSynchronize(st));
    if (hc.abort_) {
        set_error("svd: the grid barrier of the Jacobi kernel timed out (grid %d, %lld x %lld): the kernel needs all its workgroups "
                  "resident at once -- another process on this GPU, or kernels queued on other streams after this one, held "
                  "compute units (single-tenant requirement)", grid, (long long)m, (long long)n);
        return TTSK_ERR_HIP;
    }
    std::vector<int> order((size_t)n);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return h[(size_t)x] > h[(size_t)y]; });
    TTSK_HIP(hipMemcpyAsync(ord, order.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(svd_grid_store_kernel, dim3((unsigned)cdiv(n, 32), (unsigned)cdiv(m, 32)), dim3(256), 0, st, W, V, sig2,
                       ord, (int)m, (int)n, US, S, Vt);
    TTSK_LAUNCH_CHECK();
    TTSK_HIP(hipStreamSynchronize(st));            // `order` is pageable host memory of this frame
    return TTSK_OK;
}

}  // namespace ttsk

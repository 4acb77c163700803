// Small dense solves of the sketch path, on the device.
//
//  ttsk_pinv     pinv(Omega) with gelsd's truncation rule (utils.py:98-109): one-sided
//                Jacobi SVD of the tall orientation of Omega inside ONE workgroup
//                (Omega is l x r with l, r of order 10..300: a few tens of KB).
//  ttsk_qr_thin  Householder thin QR of the tall-skinny Psi unfolding
//                (sketch_dispatch.py:172), row blocks spread over the whole chip, two
//                launches per column, LAPACK dlarfg sign convention.
#include <cfloat>
#include "common.h"

namespace ttsk {

// ---------------------------------------------------------------- Jacobi SVD pinv
// W: mW x nW (mW >= nW) column-major in Wc (column j at Wc + j*mW), V: nW x nW column-major.
// On exit P[i*ldp_i + k*ldp_k] = sum_{j kept} Wc_j[i] * V_j[k] / sigma_j^2.
__global__ __launch_bounds__(1024) void jacobi_pinv_kernel(const double *__restrict__ omega, int64_t l,
                                                           int64_t r, int transposed, double *Wc,
                                                           double *V, double rcond, double *P,
                                                           int *rank_out)
{
    const int mW = (int)(transposed ? r : l), nW = (int)(transposed ? l : r);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwave = blockDim.x >> 6;
    __shared__ int s_rot;
    __shared__ double s_smax;
    // load: W = Omega^T (transposed) or Omega
    for (int t = tid; t < mW * nW; t += blockDim.x) {
        int j = t / mW, i = t - j * mW;
        Wc[t] = transposed ? omega[(int64_t)j * r + i] : omega[(int64_t)i * r + j];
    }
    for (int t = tid; t < nW * nW; t += blockDim.x) V[t] = (t / nW == t % nW) ? 1.0 : 0.0;
    __syncthreads();
    const int np = nW + (nW & 1);  // players (one dummy if odd)
    const double tol = 4.0 * DBL_EPSILON;
    for (int sweep = 0; sweep < 60; ++sweep) {
        if (tid == 0) s_rot = 0;
        __syncthreads();
        for (int round = 0; round < np - 1; ++round) {
            for (int pi = wave; pi < np / 2; pi += nwave) {
                int p, q;
                if (pi == 0) { p = np - 1; q = round; }
                else { p = (round + pi) % (np - 1); q = (round + np - 1 - pi) % (np - 1); }
                if (p >= nW || q >= nW) continue;
                if (p > q) { int t = p; p = q; q = t; }
                double *wp = Wc + (size_t)p * mW, *wq = Wc + (size_t)q * mW;
                double a = 0, b = 0, g = 0;
                for (int i = lane; i < mW; i += 64) {
                    double x = wp[i], y = wq[i];
                    a = fma(x, x, a); b = fma(y, y, b); g = fma(x, y, g);
                }
                for (int o = 32; o > 0; o >>= 1) {
                    a += __shfl_xor(a, o); b += __shfl_xor(b, o); g += __shfl_xor(g, o);
                }
                if (fabs(g) <= tol * sqrt(a * b) || g == 0.0) continue;
                if (lane == 0) s_rot = 1;
                double zeta = (b - a) / (2.0 * g);
                double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                for (int i = lane; i < mW; i += 64) {
                    double x = wp[i], y = wq[i];
                    wp[i] = c * x - s * y; wq[i] = s * x + c * y;
                }
                double *vp = V + (size_t)p * nW, *vq = V + (size_t)q * nW;
                for (int i = lane; i < nW; i += 64) {
                    double x = vp[i], y = vq[i];
                    vp[i] = c * x - s * y; vq[i] = s * x + c * y;
                }
            }
            __syncthreads();
        }
        const int rot = s_rot;
        __syncthreads();
        if (!rot) break;
    }
    // singular values -> reuse the first nW entries of a shared array
    __shared__ double s_inv2[1024];
    if (tid == 0) s_smax = 0.0;
    __syncthreads();
    for (int j = wave; j < nW; j += nwave) {
        double a = 0;
        const double *wj = Wc + (size_t)j * mW;
        for (int i = lane; i < mW; i += 64) a = fma(wj[i], wj[i], a);
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
        if (lane == 0) s_inv2[j] = a;  // sigma^2
    }
    __syncthreads();
    if (tid == 0) {
        double mx = 0;
        for (int j = 0; j < nW; ++j) mx = fmax(mx, s_inv2[j]);
        s_smax = sqrt(mx);
        int rk = 0;
        const double thr = rcond * s_smax;
        for (int j = 0; j < nW; ++j) {
            double sg = sqrt(s_inv2[j]);
            if (sg > thr && sg > 0.0) { s_inv2[j] = 1.0 / s_inv2[j]; ++rk; }
            else s_inv2[j] = 0.0;
        }
        if (rank_out) *rank_out = rk;
    }
    __syncthreads();
    // P (r x l row-major): transposed -> P[i][k] (i<mW=r, k<nW=l); else P[k][i] (k<nW=r, i<mW=l)
    for (int t = tid; t < mW * nW; t += blockDim.x) {
        int i = t / nW, k = t - i * nW;
        double acc = 0;
        for (int j = 0; j < nW; ++j) acc = fma(Wc[(size_t)j * mW + i] * s_inv2[j], V[(size_t)j * nW + k], acc);
        if (transposed) P[(int64_t)i * l + k] = acc;
        else P[(int64_t)k * l + i] = acc;
    }
}

// ---------------------------------------------------------------- Householder QR
struct Refl { double tau, scale, beta; };
__device__ __forceinline__ Refl make_refl(double alpha, double xnorm2)
{
    // LAPACK dlarfg: x = (alpha, tail), xnorm2 = |tail|^2
    Refl h;
    if (xnorm2 == 0.0) { h.tau = 0.0; h.scale = 0.0; h.beta = alpha; return h; }
    double nrm = sqrt(alpha * alpha + xnorm2);
    h.beta = alpha >= 0 ? -nrm : nrm;
    h.tau = (h.beta - alpha) / h.beta;
    h.scale = 1.0 / (alpha - h.beta);
    return h;
}

constexpr int QR_ROWS = 128;  // rows per workgroup

// tail2[j] = sum_{i>j} A[i][j]^2
__global__ void qr_tail_norm_kernel(const double *A, int64_t m, int64_t n, int64_t j, double *tail2)
{
    double acc = 0;
    for (int64_t i = j + 1 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m;
         i += (int64_t)gridDim.x * blockDim.x) {
        double x = A[i * n + j];
        acc = fma(x, x, acc);
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0 && acc != 0.0) unsafeAtomicAdd(&tail2[j], acc);
}

// w[k] += sum_{i in block, i>=j} v_i * M[i][k], k in [k0, n); v from column j of A.
__global__ __launch_bounds__(256) void qr_w_kernel(const double *__restrict__ A, const double *__restrict__ M,
                                                   int64_t m, int64_t n, int64_t j, int64_t k0,
                                                   const double *__restrict__ tail2, double *__restrict__ w)
{
    const Refl h = make_refl(A[j * n + j], tail2[j]);
    if (h.tau == 0.0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t r0 = j + (int64_t)blockIdx.x * QR_ROWS;
    const int64_t r1 = r0 + QR_ROWS < m ? r0 + QR_ROWS : m;
    __shared__ double red[4][64];
    for (int64_t kb = k0; kb < n; kb += 64) {
        const int64_t k = kb + lane;
        double acc = 0;
        if (k < n)
            for (int64_t i = r0 + wave; i < r1; i += 4) {
                double v = (i == j) ? 1.0 : A[i * n + j] * h.scale;
                acc = fma(v, M[i * n + k], acc);
            }
        red[wave][lane] = acc;
        __syncthreads();
        if (wave == 0 && k < n) {
            double s = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
            if (s != 0.0) unsafeAtomicAdd(&w[k], s);
        }
        __syncthreads();
    }
}

// M[i][k] -= tau * v_i * w[k] for i>=j, k in [k0,n); optionally accumulate the tail norm of
// column j+1 of M (rows > j+1) for the next reflector.
__global__ __launch_bounds__(256) void qr_update_kernel(const double *__restrict__ A, double *__restrict__ M,
                                                        int64_t m, int64_t n, int64_t j, int64_t k0,
                                                        const double *__restrict__ tail2,
                                                        const double *__restrict__ w, double *next_tail2)
{
    const Refl h = make_refl(A[j * n + j], tail2[j]);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t r0 = j + (int64_t)blockIdx.x * QR_ROWS;
    const int64_t r1 = r0 + QR_ROWS < m ? r0 + QR_ROWS : m;
    double nacc = 0;
    for (int64_t kb = k0; kb < n; kb += 64) {
        const int64_t k = kb + lane;
        if (k >= n) continue;
        const double tw = h.tau * w[k];
        for (int64_t i = r0 + wave; i < r1; i += 4) {
            double v = (i == j) ? 1.0 : A[i * n + j] * h.scale;
            double x = M[i * n + k];
            if (h.tau != 0.0) { x = fma(-tw, v, x); M[i * n + k] = x; }
            if (next_tail2 && k == j + 1 && i > j + 1) nacc = fma(x, x, nacc);
        }
    }
    if (next_tail2) {
        for (int o = 32; o > 0; o >>= 1) nacc += __shfl_xor(nacc, o);
        if (lane == 0 && nacc != 0.0) unsafeAtomicAdd(&next_tail2[j + 1], nacc);
    }
}

__global__ void eye_kernel(double *Q, int64_t m, int64_t n)
{
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < m * n;
         t += (int64_t)gridDim.x * blockDim.x)
        Q[t] = (t / n == t % n) ? 1.0 : 0.0;
}

}  // namespace ttsk

using namespace ttsk;

extern "C" {

int ttsk_pinv(const double *dev_omega, int64_t l, int64_t r, double rcond, double *dev_pinv,
              int *host_rank, int stream)
{
    TTSK_STREAM(st, stream);
    TTSK_ARG(dev_omega && dev_pinv, "ttsk_pinv: NULL argument");
    TTSK_ARG(l >= 1 && r >= 1, "ttsk_pinv: bad shape (%lld, %lld)", (long long)l, (long long)r);
    const int transposed = r >= l;
    const int64_t mW = transposed ? r : l, nW = transposed ? l : r;
    TTSK_ARG(nW <= 1024, "ttsk_pinv: min(l, r) = %lld > 1024 unsupported", (long long)nW);
    if (rcond < 0) rcond = DBL_EPSILON;
    double *ws = nullptr;
    const size_t ws_elems = (size_t)(mW * nW + nW * nW) + 1;
    TTSK_HIP(hipMallocAsync((void **)&ws, ws_elems * 8, st));
    int *drank = (int *)(ws + mW * nW + nW * nW);
    hipLaunchKernelGGL(jacobi_pinv_kernel, dim3(1), dim3(1024), 0, st, dev_omega, l, r, transposed, ws,
                       ws + mW * nW, rcond, dev_pinv, host_rank ? drank : (int *)nullptr);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && host_rank) {
        e = hipMemcpyAsync(host_rank, drank, sizeof(int), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    (void)hipFreeAsync(ws, st);
    TTSK_HIP(e);
    return TTSK_OK;
}

int ttsk_qr_thin(double *A, int64_t m, int64_t n, int stream)
{
    TTSK_STREAM(st, stream);
    TTSK_ARG(A, "ttsk_qr_thin: NULL argument");
    TTSK_ARG(m >= n && n >= 1, "ttsk_qr_thin: need m >= n >= 1, got (%lld, %lld)", (long long)m,
             (long long)n);
    // scratch: tail2[n+1], w_fact[n*n], w_q[n*n], Q[m*n]
    double *ws = nullptr;
    const size_t small = (size_t)(n + 1) + 2 * (size_t)n * n;
    TTSK_HIP(hipMallocAsync((void **)&ws, (small + (size_t)m * n) * 8, st));
    hipError_t e = hipMemsetAsync(ws, 0, small * 8, st);
    double *tail2 = ws, *wf = ws + n + 1, *wq = wf + n * n, *Q = ws + small;
    if (e == hipSuccess) {
        unsigned nb = (unsigned)cdiv(m, 256 * 8);
        hipLaunchKernelGGL(qr_tail_norm_kernel, dim3(nb ? nb : 1), dim3(256), 0, st, A, m, n, (int64_t)0, tail2);
        // factorisation: reflector j from column j, applied to columns j+1..n-1
        for (int64_t j = 0; j < n && j < m; ++j) {
            if (j + 1 >= n) break;  // last column: reflector only (its tail norm is already known)
            unsigned blocks = (unsigned)cdiv(m - j, QR_ROWS);
            hipLaunchKernelGGL(qr_w_kernel, dim3(blocks), dim3(256), 0, st, A, A, m, n, j, j + 1, tail2,
                               wf + j * n);
            hipLaunchKernelGGL(qr_update_kernel, dim3(blocks), dim3(256), 0, st, A, A, m, n, j, j + 1, tail2,
                               wf + j * n, tail2);
        }
        // Q = H_0 H_1 ... H_{n-1} [I; 0]
        hipLaunchKernelGGL(eye_kernel, dim3(1024), dim3(256), 0, st, Q, m, n);
        for (int64_t j = n - 1; j >= 0; --j) {
            unsigned blocks = (unsigned)cdiv(m - j, QR_ROWS);
            hipLaunchKernelGGL(qr_w_kernel, dim3(blocks), dim3(256), 0, st, A, Q, m, n, j, j, tail2, wq + j * n);
            hipLaunchKernelGGL(qr_update_kernel, dim3(blocks), dim3(256), 0, st, A, Q, m, n, j, j, tail2,
                               wq + j * n, (double *)nullptr);
        }
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(A, Q, (size_t)m * n * 8, hipMemcpyDeviceToDevice, st);
    }
    (void)hipFreeAsync(ws, st);
    TTSK_HIP(e);
    return TTSK_OK;
}

}  // extern "C"

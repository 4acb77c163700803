// Small dense solves of the sketch path, on the device.
//
//  ttsk_pinv     pinv(Omega) with gelsd's truncation rule (utils.py:98-109): one-sided
//                Jacobi SVD of the tall orientation of Omega inside ONE workgroup
//                (Omega is l x r with l, r of order 10..300: a few tens of KB).
//  ttsk_qr_thin  Householder thin QR of the tall-skinny Psi unfolding
//                (sketch_dispatch.py:172), row blocks spread over the whole chip, two
//                launches per column, LAPACK dlarfg sign convention.
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include "common.h"
#include "skinny.h"
#include "linalg_int.h"

namespace ttsk {

constexpr int CHOL_MAX_N = 256;
constexpr int CHOL_SIGN_MAX = 72;             // second factorisation + sign reconstruction in one kernel: three n x (n + 1) images in LDS
constexpr size_t SMALL_QR_MAX = 19000;      // doubles of LDS the one-workgroup Householder QR may take (152 KB)

// max that keeps a NaN (fmax returns the other operand)
__device__ __forceinline__ double nan_max(double a, double b) { return (b > a || b != b) ? b : a; }

// sum over the 16 lanes of a DPP row, result in every lane: x += ror(x, 8), 4, 2, 1 (v_mov_dpp row_ror)
template <int CTRL>
__device__ __forceinline__ double jac_dpp(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row_sum16(double x)
{
    x += jac_dpp<0x120 + 8>(x);
    x += jac_dpp<0x120 + 4>(x);
    x += jac_dpp<0x120 + 2>(x);
    x += jac_dpp<0x120 + 1>(x);
    return x;
}

// One Jacobi pair step by a 16-lane group: columns wp, wq (length mW) of W and vp, vq (length nW) of V.
// IT > 0: mW, nW <= 16 IT -- the columns stay in registers between the inner products and the rotation
// (one LDS read instead of two) and the loops are straight-line code; IT = 0: any length.
template <int IT, bool WITH_V = true>
__device__ __forceinline__ void jac_pair(double *wp, double *wq, double *vp, double *vq, const int mW, const int nW,
                                         const int gl, const double tol2, const double tiny2, int *s_rot)
{
    constexpr int ITC = IT ? IT : 1;
    double x[ITC], y[ITC];
    double a = 0, b = 0, g = 0, a1 = 0, b1 = 0, g1 = 0;
    if constexpr (IT > 0) {
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int i = gl + 16 * it;
            x[it] = i < mW ? wp[i] : 0.0;
            y[it] = i < mW ? wq[i] : 0.0;
        }
#pragma unroll
        for (int it = 0; it < IT; it += 2) {
            a = fma(x[it], x[it], a); b = fma(y[it], y[it], b); g = fma(x[it], y[it], g);
            if (it + 1 < IT) {
                a1 = fma(x[it + 1], x[it + 1], a1); b1 = fma(y[it + 1], y[it + 1], b1); g1 = fma(x[it + 1], y[it + 1], g1);
            }
        }
        a += a1; b += b1; g += g1;
    } else {
        for (int i = gl; i < mW; i += 16) {
            const double xx = wp[i], yy = wq[i];
            a = fma(xx, xx, a); b = fma(yy, yy, b); g = fma(xx, yy, g);
        }
    }
    a = row_sum16(a); b = row_sum16(b); g = row_sum16(g);
    // both columns at rounding-noise level (norm^2 <= tiny2 = (4 m eps)^2 x the largest column norm^2 of this
    // sweep): directions the rank rule drops anyway; rotating noise against noise never converges in the
    // relative sense and kept rank-deficient sketches sweeping until the limit.  W = A V holds regardless.
    if (g * g <= tol2 * (a * b) || g == 0.0 || (a <= tiny2 && b <= tiny2)) return;
    if (gl == 0) *s_rot = 1;
    const double zeta = (b - a) / (2.0 * g);
    const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
    const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
    if constexpr (IT > 0) {
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int i = gl + 16 * it;
            if (i < mW) { wp[i] = c * x[it] - s * y[it]; wq[i] = s * x[it] + c * y[it]; }
        }
        if constexpr (WITH_V) {
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int i = gl + 16 * it;
                x[it] = i < nW ? vp[i] : 0.0;
                y[it] = i < nW ? vq[i] : 0.0;
            }
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int i = gl + 16 * it;
                if (i < nW) { vp[i] = c * x[it] - s * y[it]; vq[i] = s * x[it] + c * y[it]; }
            }
        }
    } else {
        for (int i = gl; i < mW; i += 16) {
            const double xx = wp[i], yy = wq[i];
            wp[i] = c * xx - s * yy; wq[i] = s * xx + c * yy;
        }
        if constexpr (WITH_V)
            for (int i = gl; i < nW; i += 16) {
                const double xx = vp[i], yy = vq[i];
                vp[i] = c * xx - s * yy; vq[i] = s * xx + c * yy;
            }
    }
}

// Householder QR of Wc (column-major mW x nW, mW >= nW) in place, R only: afterwards the leading nW x nW block
// holds R^T (lower triangular, i.e. column j = row j of R) and nothing else of Wc is meaningful.  One barrier pair
// per column; every 16-lane group recomputes the reflector of column j itself and applies it to its own columns.
__device__ void jac_qr_rt(double *Wc, const int mW, const int nW, const int tid, const int nthreads, double *s_beta)
{
    const int grp = tid >> 4, gl = tid & 15, ngrp = nthreads >> 4;
    for (int j = 0; j < nW; ++j) {
        const double *cj = Wc + (size_t)j * mW;
        double sig = 0.0;
        for (int i = j + 1 + gl; i < mW; i += 16) sig = fma(cj[i], cj[i], sig);
        sig = row_sum16(sig);
        const double alpha = cj[j];
        double beta = alpha, tau = 0.0, scale = 0.0;
        if (sig != 0.0) {                                              // LAPACK dlarfg
            const double nrm = sqrt(alpha * alpha + sig);
            beta = alpha >= 0 ? -nrm : nrm;
            tau = (beta - alpha) / beta;
            scale = 1.0 / (alpha - beta);
        }
        for (int k = j + 1 + grp; k < nW; k += ngrp) {
            double *ck = Wc + (size_t)k * mW;
            double w = gl == 0 ? ck[j] : 0.0;                          // v[0] = 1
            for (int i = j + 1 + gl; i < mW; i += 16) w = fma(cj[i] * scale, ck[i], w);
            w = row_sum16(w) * tau;
            if (gl == 0) ck[j] -= w;
            for (int i = j + 1 + gl; i < mW; i += 16) ck[i] = fma(-w, cj[i] * scale, ck[i]);
        }
        if (tid == 0) *s_beta = beta;
        __syncthreads();
        for (int i = j + tid; i < mW; i += nthreads) Wc[(size_t)j * mW + i] = i == j ? *s_beta : 0.0;
        __syncthreads();
    }
    // R (upper triangle of the leading block) -> R^T
    for (int t = tid; t < nW * nW; t += nthreads) {
        const int j = t / nW, i = t - j * nW;
        if (i < j) { Wc[(size_t)i * mW + j] = Wc[(size_t)j * mW + i]; Wc[(size_t)j * mW + i] = 0.0; }
    }
    __syncthreads();
}

__constant__ int jac_precond_on = 1;     // TTSK_JACOBI_PRECOND=0 clears it (host: jacobi_lds_mode)

// ---------------------------------------------------------------- Jacobi SVD pinv
// W: mW x nW (mW >= nW) column-major in Wc (column j at Wc + j*mW), V: nW x nW column-major.
// On exit P[i*ldp_i + k*ldp_k] = sum_{j kept} Wc_j[i] * V_j[k] / sigma_j^2.
// With svd_US != nullptr the kernel returns the factors instead of the pseudo-inverse (input taken
// untransposed, l >= r): US (l x r row-major) = U diag(S), S (r) descending, Vt (r x r row-major).
template <int LM>
__global__ __launch_bounds__(1024) void jacobi_pinv_kernel(const double *__restrict__ omega, int64_t l,
                                                           int64_t r, int transposed, double *Wc,
                                                           double *V, double rcond, double *P,
                                                           int *rank_out, double *svd_US,
                                                           double *svd_S, double *svd_Vt,
                                                           const int *run_if_nonzero = nullptr, int64_t om_stride = 0,
                                                           int64_t p_stride = 0)
{
    // a batch of equally spaced matrices: workgroup b takes matrix b (grid 1, strides 0: the plain call)
    omega += (int64_t)blockIdx.x * om_stride;
    P += (int64_t)blockIdx.x * p_stride;
    if (run_if_nonzero) run_if_nonzero += blockIdx.x;
    // queued behind the normal-equations attempt without the host having looked at its verdict (ttsk_pinv_end):
    // nothing to do if that attempt was accepted
    if (run_if_nonzero && *run_if_nonzero == 0) return;
    const int mW = (int)(transposed ? r : l), nW = (int)(transposed ? l : r);
    // LM = 1: W lives in LDS, 2: W and V (the global scratch is then unused).  A template parameter, not a
    // run-time switch: a pointer that may be LDS or global compiles to FLAT loads and stores (67 + 54 of them
    // in this kernel), several times slower than ds_read / ds_write for data that is in LDS.
    // All LDS is dynamic: [sigma^2 (nW doubles) | order (nW ints, padded) | W | V], so that a 100 x 100
    // factor (2 x 80 KB) still fits next to them in the 160 KB of a CU.
    extern __shared__ double jac_lds[];
    double *s_inv2 = jac_lds;
    int *s_ord = reinterpret_cast<int *>(jac_lds + nW);
    double *jac_mat = jac_lds + nW + (nW + 1) / 2;
    if constexpr (LM >= 1) Wc = jac_mat;
    if constexpr (LM >= 2) V = jac_mat + (size_t)mW * nW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwave = blockDim.x >> 6;
    __shared__ int s_rot;
    __shared__ double s_smax;
    const int grp = tid >> 4, gl = tid & 15, ngrp = blockDim.x >> 4;
    // load: W = Omega^T (transposed) or Omega
    for (int t = tid; t < mW * nW; t += blockDim.x) {
        int j = t / mW, i = t - j * mW;
        Wc[t] = transposed ? omega[(int64_t)j * r + i] : omega[(int64_t)i * r + j];
    }
    for (int t = tid; t < nW * nW; t += blockDim.x) V[t] = (t / nW == t % nW) ? 1.0 : 0.0;
    __syncthreads();
    // Pseudo-inverse mode: QR first, then the sweeps run on R^T (nW x nW, lower triangular) without accumulating V
    // (dgejsv's preconditioning: Jacobi on the transposed triangular factor needs fewer sweeps -- 15 -> 10 for a
    // rank-15 100 x 50 sketch -- and the columns are nW instead of mW long).  The right singular vectors are then
    // the normalised columns of the converged matrix, and W V is recomputed from the input.  1.1 -> 0.54 ms for that
    // sketch; the factor mode (svd_US) keeps the plain iteration with its high relative accuracy.
    const bool precond = svd_US == nullptr && jac_precond_on;
    __shared__ double s_beta;
    if (precond) jac_qr_rt(Wc, mW, nW, tid, blockDim.x, &s_beta);
    const int rows = precond ? nW : mW;          // length of the columns the sweeps rotate (column stride stays mW)
    const int np = nW + (nW & 1);  // players (one dummy if odd)
    // LAPACK dgesvj stops at sqrt(m) eps: the computed inner product of two columns of length m carries that
    // much rounding noise, a tighter bound keeps rotating noise until the sweep limit
    const double tol = fmax(4.0, sqrt((double)mW)) * DBL_EPSILON, tol2 = tol * tol;
    const int nm1 = np - 1;
    const int itc = rows <= 64 ? 4 : (rows <= 128 ? 8 : 0);   // mW >= nW
    for (int sweep = 0; sweep < 60; ++sweep) {
        if (tid == 0) { s_rot = 0; s_smax = 0.0; }
        __syncthreads();
        // largest column norm^2 of this sweep (fixed reduction order: per-column sums, then one thread)
        for (int j = grp; j < nW; j += ngrp) {
            const double *wj = Wc + (size_t)j * mW;
            double a = 0;
            for (int i = gl; i < rows; i += 16) a = fma(wj[i], wj[i], a);
            a = row_sum16(a);
            if (gl == 0) s_inv2[j] = a;
        }
        __syncthreads();
        if (tid == 0) {
            double mx = 0;
            for (int j = 0; j < nW; ++j) mx = fmax(mx, s_inv2[j]);
            s_smax = mx;
        }
        __syncthreads();
        const double tiny = 4.0 * mW * DBL_EPSILON, tiny2 = tiny * tiny * s_smax;
        for (int round = 0; round < np - 1; ++round) {
            // one column pair per group of 16 lanes (a DPP row): 64 pairs of a round rotate at once and the
            // three inner products are reduced by four in-register row rotations.  (A whole wave per pair was
            // four sequential pair steps per round at n = 100, each paying six ds_bpermute stages.)
            for (int pi = grp; pi < np / 2; pi += ngrp) {
                int p, q;
                if (pi == 0) { p = nm1; q = round; }
                else {
                    p = round + pi; p -= p >= nm1 ? nm1 : 0;
                    q = round + nm1 - pi; q -= q >= nm1 ? nm1 : 0;
                }
                if (p >= nW || q >= nW) continue;
                if (p > q) { int t = p; p = q; q = t; }
                double *wp = Wc + (size_t)p * mW, *wq = Wc + (size_t)q * mW;
                double *vp = V + (size_t)p * nW, *vq = V + (size_t)q * nW;
                if (precond) {
                    if (itc == 4) jac_pair<4, false>(wp, wq, vp, vq, rows, nW, gl, tol2, tiny2, &s_rot);
                    else if (itc == 8) jac_pair<8, false>(wp, wq, vp, vq, rows, nW, gl, tol2, tiny2, &s_rot);
                    else jac_pair<0, false>(wp, wq, vp, vq, rows, nW, gl, tol2, tiny2, &s_rot);
                } else if (itc == 4) jac_pair<4>(wp, wq, vp, vq, mW, nW, gl, tol2, tiny2, &s_rot);
                else if (itc == 8) jac_pair<8>(wp, wq, vp, vq, mW, nW, gl, tol2, tiny2, &s_rot);
                else jac_pair<0>(wp, wq, vp, vq, mW, nW, gl, tol2, tiny2, &s_rot);
            }
            __syncthreads();
        }
        const int rot = s_rot;
        __syncthreads();
        if (!rot) break;
    }
    if (precond) {
        // V_j = column j of the converged R^T V' over its norm (zero for a column that vanished), then W V from the input
        for (int j = grp; j < nW; j += ngrp) {
            const double *wj = Wc + (size_t)j * mW;
            double a = 0;
            for (int i = gl; i < rows; i += 16) a = fma(wj[i], wj[i], a);
            a = row_sum16(a);
            if (gl == 0) s_inv2[j] = a;
        }
        __syncthreads();
        for (int t = tid; t < nW * nW; t += blockDim.x) {
            const int j = t / nW, k = t - j * nW;
            V[t] = s_inv2[j] > 0.0 ? Wc[(size_t)j * mW + k] / sqrt(s_inv2[j]) : 0.0;
        }
        __syncthreads();
        for (int t = tid; t < mW * nW; t += blockDim.x) {
            const int j = t / mW, i = t - j * mW;
            const double *vj = V + (size_t)j * nW;
            double acc = 0.0;
            if (transposed) for (int k = 0; k < nW; ++k) acc = fma(omega[(int64_t)k * r + i], vj[k], acc);
            else for (int k = 0; k < nW; ++k) acc = fma(omega[(int64_t)i * r + k], vj[k], acc);
            Wc[t] = acc;
        }
        __syncthreads();
    }
    // singular values -> reuse the first nW entries of a shared array
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
    if (svd_US) {
        // order the columns by descending singular value (nW <= 1024, one thread)
        if (tid == 0) {
            for (int j = 0; j < nW; ++j) s_ord[j] = j;
            for (int a = 1; a < nW; ++a) {
                const int key = s_ord[a];
                const double kv = s_inv2[key];
                int b = a - 1;
                while (b >= 0 && s_inv2[s_ord[b]] < kv) { s_ord[b + 1] = s_ord[b]; --b; }
                s_ord[b + 1] = key;
            }
        }
        __syncthreads();
        for (int t = tid; t < mW * nW; t += blockDim.x) {
            const int i = t / nW, k = t - i * nW;
            svd_US[t] = Wc[(size_t)s_ord[k] * mW + i];
        }
        for (int t = tid; t < nW * nW; t += blockDim.x) {
            const int k = t / nW, i = t - k * nW;
            svd_Vt[t] = V[(size_t)s_ord[k] * nW + i];
        }
        for (int k = tid; k < nW; k += blockDim.x) svd_S[k] = sqrt(s_inv2[s_ord[k]]);
        return;
    }
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

// All cross-workgroup reductions of the QR go through per-workgroup partial slots that the
// NEXT launch sums in a fixed order: no atomics, bit-reproducible results.
//   tpart[b]        partial of the tail norm^2 of the current pivot column (nb_t slots)
//   wpart[b*n + k]  partial of w[k] = sum_i v_i M[i][k]                     (nb_w slots)
__device__ __forceinline__ double sum_slots(const double *p, int nslots, int stride)
{
    double s = 0;
    for (int b = 0; b < nslots; ++b) s += p[(size_t)b * stride];
    return s;
}

// tpart[b] = sum_{i in block b, i>j} A[i][j]^2 ; block b covers rows j + 128 b ...
__global__ __launch_bounds__(256) void qr_tail_norm_kernel(const double *A, int64_t m, int64_t n, int64_t j,
                                                           double *tpart)
{
    const int64_t r0 = j + (int64_t)blockIdx.x * QR_ROWS;
    const int64_t r1 = r0 + QR_ROWS < m ? r0 + QR_ROWS : m;
    __shared__ double red[4];
    double acc = 0;
    for (int64_t i = r0 + threadIdx.x; i < r1; i += blockDim.x)
        if (i > j) { double x = A[i * n + j]; acc = fma(x, x, acc); }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) tpart[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// wpart[b][k] = sum_{i in block b} v_i * M[i][k], k in [k0, n); v from column j of A.
__global__ __launch_bounds__(256) void qr_w_kernel(const double *A, const double *M, int64_t m, int64_t n,
                                                   int64_t j, int64_t k0, const double *tpart, int nb_t,
                                                   double *wpart)
{
    const Refl h = make_refl(A[j * n + j], sum_slots(tpart, nb_t, 1));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t r0 = j + (int64_t)blockIdx.x * QR_ROWS;
    const int64_t r1 = r0 + QR_ROWS < m ? r0 + QR_ROWS : m;
    __shared__ double red[4][64];
    for (int64_t kb = k0; kb < n; kb += 64) {
        const int64_t k = kb + lane;
        double acc = 0;
        if (k < n && h.tau != 0.0)
            for (int64_t i = r0 + wave; i < r1; i += 4) {
                double v = (i == j) ? 1.0 : A[i * n + j] * h.scale;
                acc = fma(v, M[i * n + k], acc);
            }
        red[wave][lane] = acc;
        __syncthreads();
        if (wave == 0 && k < n)
            wpart[(size_t)blockIdx.x * n + k] = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
        __syncthreads();
    }
}

// M[i][k] -= tau * v_i * w[k] for i>=j, k in [k0,n); optionally the tail norm partials of
// column j+1 of M (rows > j+1) for the next reflector go to next_tpart[b].
__global__ __launch_bounds__(256) void qr_update_kernel(const double *A, double *M, int64_t m, int64_t n,
                                                        int64_t j, int64_t k0, const double *tpart, int nb_t,
                                                        const double *wpart, int nb_w, double *next_tpart)
{
    const Refl h = make_refl(A[j * n + j], sum_slots(tpart, nb_t, 1));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t r0 = j + (int64_t)blockIdx.x * QR_ROWS;
    const int64_t r1 = r0 + QR_ROWS < m ? r0 + QR_ROWS : m;
    __shared__ double red[4];
    double nacc = 0;
    for (int64_t kb = k0; kb < n; kb += 64) {
        const int64_t k = kb + lane;
        if (k >= n) continue;
        const double tw = h.tau * sum_slots(wpart + k, nb_w, (int)n);
        for (int64_t i = r0 + wave; i < r1; i += 4) {
            double v = (i == j) ? 1.0 : A[i * n + j] * h.scale;
            double x = M[i * n + k];
            if (h.tau != 0.0) { x = fma(-tw, v, x); M[i * n + k] = x; }
            if (next_tpart && k == j + 1 && i > j + 1) nacc = fma(x, x, nacc);
        }
    }
    if (next_tpart) {
        for (int o = 32; o > 0; o >>= 1) nacc += __shfl_xor(nacc, o);
        if (lane == 0) red[wave] = nacc;
        __syncthreads();
        if (threadIdx.x == 0) next_tpart[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
    }
}

__global__ void eye_kernel(double *Q, int64_t m, int64_t n)
{
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < m * n;
         t += (int64_t)gridDim.x * blockDim.x)
        Q[t] = (t / n == t % n) ? 1.0 : 0.0;
}

__global__ void triu_kernel(double *A, int64_t m, int64_t n)
{
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < m * n;
         t += (int64_t)gridDim.x * blockDim.x)
        if (t % n < t / n) A[t] = 0.0;
}

}  // namespace ttsk

namespace ttsk {

// ---- Cholesky-based fast paths ----------------------------------------------------------
// Both solves of the path are overwhelmingly applied to well-conditioned matrices: Omega is an
// (l x r) sketch of full row or column rank, and the matrix orth_step factorises is Psi Omega^+.
// For those, pinv(Omega) = Omega^T (Omega Omega^T)^-1 and the thin QR by CholeskyQR2 are a handful
// of products on the chain kernels plus an n x n Cholesky in one workgroup (n <= 128) -- ~0.1 ms
// instead of 2 ms (one-workgroup Jacobi SVD) and 8 ms (4 n Householder launches) at C3.  The
// Cholesky kernel reports failure (not positive definite, or diag(R) spread beyond cond_tol) and the
// callers then run the robust kernels above on the untouched input, so rank-deficient sketches
// behave exactly as before.

// G (n x n symmetric, row-major) = R^T R; Rinv = R^-1 (upper triangular, dense n x n) and optionally
// Ginv = Rinv Rinv^T = G^-1.  status[0] = 0 ok, 1 rejected.
__device__ void hh_signs_lds(double *B, int n, int ld, int square, double *S, double *shadow, int tid, int nthr = 256);

// Qtop (optional, n <= CHOL_SIGN_MAX): the call is the SECOND factorisation of CholeskyQR2 -- the kernel goes on to form
// the top n x n block of Q = Qtop R^-1 in LDS, reconstructs LAPACK's Householder column signs from it (hh_signs_lds)
// and writes Rinv with its columns scaled by them: three launches of qr_cholesky in one.
// sticky (optional): set to 1 on rejection, never cleared here (deferred verdicts: ttsk_orth_step);
// pminmax (optional): smallest / largest pivot, for callers that combine several blocks (chol_inv_any).
// The factorisation and the inverse of chol_inv_kernel on a matrix that sits in LDS (A: n x n with row stride ld, both
// triangles; xd: n doubles): afterwards X = R^-1 is stored with its strict upper part transposed into A's lower
// triangle (X[i][c], i < c, at A[c][i]) and its diagonal in xd.  Every thread of the workgroup calls it (barriers
// inside); threads beyond the first 256 take part in the recurrence only.  bad / pmin / pmax: the verdict's inputs
// (every thread has them).
__device__ __forceinline__ void chol_lds(double *A, double *xd, const int n, const int ld, double *shadow, const int tid,
                                         const int nthr, int &bad, double &pmin, double &pmax)
{
    const bool core = tid < 256;
    const int ti = tid >> 4, tc = tid & 15, nrow = nthr >> 4;
    auto rcp2 = [](double x) { double r = __builtin_amdgcn_rcp(x); r = r * (2.0 - x * r); return r * (2.0 - x * r); };
    int j = 0;
    for (; j + 1 < n; j += 2) {
        const double *r0 = A + j * ld, *r1 = A + (j + 1) * ld;
        double p0 = r0[j];
        if (!(p0 > 0.0)) { bad = 1; p0 = 1.0; }
        const double pi0 = rcp2(p0);
        const double g = r0[j + 1] * pi0;                      // factor of row j + 1 against row j
        double p1 = fma(-g, r0[j + 1], r1[j + 1]);             // pivot of column j + 1 after step j
        if (!(p1 > 0.0)) { bad = 1; p1 = 1.0; }
        const double pi1 = rcp2(p1);
        pmin = fmin(pmin, fmin(p0, p1));
        pmax = fmax(pmax, fmax(p0, p1));
        for (int i = j + 2 + ti; i < n; i += nrow) {
            const double a0 = r0[i], a1 = fma(-g, a0, r1[i]);  // A[j][i] and A[j+1][i] after step j
            const double f0 = a0 * pi0, f1 = a1 * pi1;
            for (int c = i + tc; c < n; c += 16) {
                const double u1 = fma(-g, r0[c], r1[c]);       // row j + 1 after step j, at c
                A[i * ld + c] = fma(-f1, u1, fma(-f0, r0[c], A[i * ld + c]));
            }
        }
        double *sh = shadow + ((j >> 1) & 1) * 128;
        for (int c = j + 1 + tid; c < n; c += nthr) sh[c] = fma(-g, r0[c], r1[c]);
        __syncthreads();
        for (int c = j + 1 + tid; c < n; c += nthr) A[(j + 1) * ld + c] = sh[c];
    }
    __syncthreads();
    if (j < n) {                                               // odd n: the last pivot
        double piv = A[j * ld + j];
        if (!(piv > 0.0)) { bad = 1; piv = 1.0; }
        pmin = fmin(pmin, piv);
        pmax = fmax(pmax, piv);
    }
    // R[j][c] = row j / r_j; xd[j] = 1 / R[j][j] = 1 / r_j
    if (core && tid < n) xd[tid] = 1.0 / sqrt(A[tid * ld + tid] > 0.0 ? A[tid * ld + tid] : 1.0);
    __syncthreads();
    if (core)
        for (int jj = ti; jj < n; jj += 16) {
            const double sc = xd[jj];
            for (int c = jj + tc; c < n; c += 16) A[jj * ld + c] *= sc;
        }
    __syncthreads();
    // X = R^-1 stays in LDS: its strict upper part X[i][c] (i < c) goes to the unused strict lower triangle of A at
    // A[c][i], its diagonal to xd[].
    // X = R^-1 in 16 x 16 blocks.  (1) The diagonal blocks, all at once: a quad of lanes owns a column and walks the
    // (up to 15) rows of its own block -- one wavefront per 16 columns, in order, no workgroup barrier.  (2) Block rows
    // from the bottom: X_ij = -X_ii (sum_{i<k<=j} R_ik X_kj) on the matrix cores; the accumulator registers of the sum
    // are the B operand of the second product (register kb of a lane holds rows 4 kb + (lane >> 4): exactly k-block
    // kb); one barrier per block row.  (One column per quad over ALL rows was 42 k of the kernel's 118 k cycles at
    // n = 50 -- 900 cycles of dependent LDS reads per row -- and 152 k of 396 k at n = 100; now 20 k and 55 k.)
    const int q4 = tid & 3, col4 = tid >> 2;
    for (int c = core ? col4 : n; c < n; c += 64) {
        const double *xc = A + c * ld;                                  // X[k][c] at A[c][k], k < c
        const int top = c & ~15;
        for (int i = c - 1; i >= top; --i) {
            const double *ri = A + i * ld;
            double acc = 0.0;
            for (int k = i + 1 + q4; k < c; k += 4) acc = fma(ri[k], xc[k], acc);
            acc += jac_dpp<0xB1>(acc);              // quad_perm 1 0 3 2
            acc += jac_dpp<0x4E>(acc);              // quad_perm 2 3 0 1
            if (q4 == 0) A[c * ld + i] = -(acc + ri[c] * xd[c]) * xd[i];
        }
    }
    __syncthreads();
    {
        const int lane = tid & 63, wv = tid >> 6, x16 = lane & 15, kq = lane >> 4;
        const int nt = (n + 15) >> 4;
        // X(r, c), r <= c, from its storage: strict upper part transposed into the lower triangle, diagonal in xd
        auto Xat = [&](int r, int c) -> double {
            if (r >= n || c >= n || r > c) return 0.0;
            return r == c ? xd[c] : A[c * ld + r];
        };
        for (int bi = nt - 2; bi >= 0; --bi) {
            for (int bj = core ? bi + 1 + wv : nt; bj < nt; bj += 4) {
                v4d S = {0.0, 0.0, 0.0, 0.0};
                const int ra = 16 * bi + x16;
                for (int bk = bi + 1; bk <= bj; ++bk)
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb) {
                        const int k = 16 * bk + 4 * kb + kq;
                        const double av = (ra < n && k < n) ? A[ra * ld + k] : 0.0;          // R[ra][k], k > ra
                        S = mfma16(av, Xat(k, 16 * bj + x16), S);
                    }
                v4d Xn = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) Xn = mfma16(Xat(16 * bi + x16, 16 * bi + 4 * kb + kq), S[kb], Xn);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int r = 16 * bi + kq + 4 * jj, c = 16 * bj + x16;
                    if (r < n && c < n) A[c * ld + r] = -Xn[jj];
                }
            }
            __syncthreads();
        }
    }
}

// Launched with 256 threads, or with 1024 (n <= 64, chol_threads()): the extra twelve waves take part in the recurrence
// only -- one row per 16-lane group instead of four, twelve more waves to hide the LDS round trips behind -- and leave.
__global__ __launch_bounds__(1024) void chol_inv_kernel(const double *__restrict__ G, int n, double *__restrict__ Rinv,
                                                       double *__restrict__ Ginv, int *__restrict__ status,
                                                       double cond_tol, int *__restrict__ sticky, double *__restrict__ pminmax,
                                                       const double *__restrict__ Qtop = nullptr, int square = 0, int expand = 1)
{
    extern __shared__ double sm[];
    const int ld = n + 1, tid = threadIdx.x;
    // a batch of independent factorisations: workgroup b takes the matrices n * n * b further on (grid 1: the plain call)
    G += (size_t)blockIdx.x * n * n;
    Rinv += (size_t)blockIdx.x * n * n;
    if (Ginv) Ginv += (size_t)blockIdx.x * n * n;
    status += blockIdx.x;
    double *A = sm;
    double *xd = sm + n * ld;
    const int nthr = blockDim.x;
    const bool core = tid < 256;                       // the threads that run every phase
    for (int e = tid; e < n * n; e += nthr) A[(e / n) * ld + e % n] = G[e];
    __syncthreads();
    // Unscaled right-looking recurrence: row j keeps r_j R[j][:] (r_j^2 = pivot) until the end, the trailing update
    // divides by the pivot instead, every thread reads the pivots itself.  TWO columns per barrier: the pivot of
    // column j + 1 and its row after step j follow from rows j and j + 1 alone, so every thread forms them on the
    // fly and updates its elements with both columns at once; the updated row j + 1 goes through a shadow row (the
    // others still read the old one) and home after the barrier -- nobody reads it again before the scaling pass.
    // (History at n = 50: three barriers per column 75 us of 118; one per column 55 k cycles; two columns 42 k.)
    int bad = 0;
    double pmin = 1e300, pmax = 0.0;                 // pivots r_j^2: the square root is not needed in the loop
    const int ti = tid >> 4, tc = tid & 15, nrow = nthr >> 4;
    // hardware reciprocal + two Newton steps (a full division is ~4x the instructions, on the critical path)
    auto rcp2 = [](double x) { double r = __builtin_amdgcn_rcp(x); r = r * (2.0 - x * r); return r * (2.0 - x * r); };
    __shared__ double shadow[2 * 128];
    int j = 0;
    bool expanded = false;
    if ((Qtop && expand) || expand == 2) {                  // expand == 2: a second factorisation whose signs come later
        // The second factorisation of CholeskyQR2 sees G = I + E with |E| ~ n kappa(A)^2 eps.  For n max|E| <= 1e-8 the
        // factor's inverse is I - Phi(E) (Phi: strict upper triangle + half the diagonal) to within n |E|^2 < 1e-17:
        // no recurrence at all (n / 2 steps of ~1600 cycles otherwise).
        double em = 0.0;
        if (core) {
            for (int e = tid; e < n * n; e += 256) {
                const int i = e / n, c = e - i * n;
                em = nan_max(em, fabs(A[i * ld + c] - (i == c ? 1.0 : 0.0)));
            }
            em = nan_max(em, jac_dpp<0xB1>(em));
            em = nan_max(em, jac_dpp<0x4E>(em));
            if ((tid & 3) == 0) shadow[tid >> 2] = em;
        }
        __syncthreads();
        em = 0.0;
        for (int k = 0; k < 64; ++k) em = nan_max(em, shadow[k]);
        __syncthreads();
        // nan_max keeps a NaN (fmax would drop it and a Gram matrix full of NaN would pass as the identity): NaN compares
        // false here and the recurrence below rejects it
        if (em * n <= 1e-8) {
            if (core)
                for (int e = tid; e < n * n; e += 256) {
                    const int i = e / n, c = e - i * n;
                    if (i < c) A[c * ld + i] = -A[i * ld + c];      // X[i][c] lives at A[c][i]
                    else if (i == c) xd[c] = 1.0 - 0.5 * (A[i * ld + i] - 1.0);
                }
            if (tid == 0) status[0] = 0;
            __syncthreads();
            expanded = true;
        }
    }
    if (expanded && !core) return;
    if (!expanded) {
    chol_lds(A, xd, n, ld, shadow, tid, nthr, bad, pmin, pmax);
    if (!core) return;
    if (tid == 0) {
        const int rej = (bad || pmin < cond_tol * cond_tol * pmax) ? 1 : 0;
        status[0] = rej;
        if (rej && sticky) *sticky = 1;
        if (pminmax) { pminmax[0] = bad ? -1.0 : pmin; pminmax[1] = pmax; }
    }
    }   // !expanded
    // X(r, c) from that storage (zero below the diagonal): the results are written straight from it -- no pass that
    // makes X dense in LDS first
    auto Xe = [&](int r, int c) -> double {
        if (r >= n || c >= n || r > c) return 0.0;
        return r == c ? xd[c] : A[c * ld + r];
    };
    if (Qtop) {
        // top block of Q = Qtop X (X upper triangular), the signs of LAPACK's reflectors from its modified LU, Rinv = X S
        double *B = xd + n, *S = B + n * ld, *Qs = S + n;      // Qs: Qtop staged (one coalesced pass instead of n dependent loads per cell)
        for (int e = tid; e < n * n; e += 256) Qs[(e / n) * ld + e % n] = Qtop[e];
        __syncthreads();
        for (int e = tid; e < n * n; e += 256) {
            const int i = e / n, c = e - i * n;
            const double *qi = Qs + i * ld, *xc = A + c * ld;    // X[k][c], k < c, sits at A[c][k]; the diagonal in xd
            double acc = qi[c] * xd[c];
            for (int k = 0; k < c; ++k) acc = fma(qi[k], xc[k], acc);
            B[i * ld + c] = acc;
        }
        __syncthreads();
        hh_signs_lds(B, n, ld, square, S, shadow, tid);
        for (int i = ti; i < n; i += 16)
            for (int c = tc; c < n; c += 16) Rinv[i * n + c] = Xe(i, c) * S[c];
        return;
    }
    for (int i = ti; i < n; i += 16)
        for (int c = tc; c < n; c += 16) Rinv[i * n + c] = Xe(i, c);
    if (Ginv) {
        // G^-1 = X X^T on the matrix cores: tile (ta, tb), tb >= ta, one per wave and turn, mirrored on the way out;
        // X is upper triangular, so the sum over k starts at the column tile (22 k -> 11 k cycles at n = 50, 125 k ->
        // 30 k at n = 100 against one thread per element)
        const int lane = tid & 63, wv = tid >> 6, x16 = lane & 15, kq = lane >> 4;
        const int nt = (n + 15) >> 4, nkb = (n + 3) >> 2;
        int t = 0;
        for (int ta = 0; ta < nt; ++ta)
            for (int tb = ta; tb < nt; ++tb, ++t) {
                if ((t & 3) != wv) continue;
                v4d acc = {0.0, 0.0, 0.0, 0.0};
                const int ra = 16 * ta + x16, rb = 16 * tb + x16;
                for (int kb = 4 * tb; kb < nkb; ++kb) {
                    const int k = 4 * kb + kq;
                    acc = mfma16(Xe(ra, k), Xe(rb, k), acc);
                }
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int i = 16 * ta + kq + 4 * jj, c = 16 * tb + x16;
                    if (i < n && c < n) { Ginv[i * n + c] = acc[jj]; Ginv[c * n + i] = acc[jj]; }
                }
            }
    }
}

// Column signs that turn the Q of CholeskyQR (R with positive diagonal) into LAPACK's Householder Q:
// the modified LU of the top n x n block of Q (Ballard et al., "Reconstructing Householder vectors
// from TSQR"): S_j = -sgn(pivot_j); for a square matrix the last reflector is the identity.
// Scales the columns of Rinv (n x n) by S in place.
__device__ void hh_signs_lds(double *B, int n, int ld, int square, double *S, double *shadow, int tid, int nthr)
{
    // only the signs are needed: every thread derives the modified pivots itself and the trailing update uses the
    // unscaled columns; two columns per barrier as in chol_inv_kernel (row j + 1 after step j through a shadow row)
    const int ti = tid >> 4, tc = tid & 15, nrow = nthr >> 4;          // (every thread of the workgroup calls this: barriers inside)
    int j = 0;
    for (; j + 1 < n; j += 2) {
        const double *b0 = B + j * ld, *b1 = B + (j + 1) * ld;
        const double sgn0 = b0[j] >= 0.0 ? -1.0 : 1.0;
        const double pinv0 = 1.0 / (b0[j] - sgn0);
        const double g = b1[j] * pinv0;                                  // factor of row j + 1 against row j
        const double piv1 = fma(-g, b0[j + 1], b1[j + 1]);               // pivot of column j + 1 after step j
        double sgn1 = piv1 >= 0.0 ? -1.0 : 1.0;
        if (square && j + 1 == n - 1) sgn1 = -sgn1;
        const double pinv1 = 1.0 / (piv1 - sgn1);
        if (tid == 0) { S[j] = sgn0; S[j + 1] = sgn1; }
        for (int i = j + 2 + ti; i < n; i += nrow) {
            const double f0 = B[i * ld + j] * pinv0;
            const double f1 = fma(-f0, b0[j + 1], B[i * ld + j + 1]) * pinv1;
            for (int c = j + 2 + tc; c < n; c += 16) {
                const double u1 = fma(-g, b0[c], b1[c]);                 // row j + 1 after step j, at c
                B[i * ld + c] = fma(-f1, u1, fma(-f0, b0[c], B[i * ld + c]));
            }
        }
        double *sh = shadow + ((j >> 1) & 1) * 128;
        for (int c = j + 2 + tid; c < n; c += nthr) sh[c] = fma(-g, b0[c], b1[c]);
        __syncthreads();
        for (int c = j + 2 + tid; c < n; c += nthr) B[(j + 1) * ld + c] = sh[c];
    }
    __syncthreads();
    if (j < n) {                                                         // odd n: the last pivot
        double sgn = B[j * ld + j] >= 0.0 ? -1.0 : 1.0;
        if (square && j == n - 1) sgn = -sgn;
        if (tid == 0) S[j] = sgn;
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void hh_sign_scale_kernel(const double *__restrict__ Qtop, int n, int square,
                                                            double *__restrict__ Rinv)
{
    extern __shared__ double sm[];
    const int ld = n + 1, tid = threadIdx.x;
    double *B = sm, *S = sm + n * ld;
    __shared__ double shadow[2 * 128];
    for (int e = tid; e < n * n; e += 256) B[(e / n) * ld + e % n] = Qtop[e];
    __syncthreads();
    hh_signs_lds(B, n, ld, square, S, shadow, tid);
    for (int e = tid; e < n * n; e += 256) Rinv[e] *= S[e % n];
}

// Thin QR of a SMALL, nearly square matrix (m n doubles fit the LDS: the first mode of a sketch whose rank was trimmed to
// the mode size, m = n_0 rows) in one workgroup: Householder with LAPACK's dlarfg signs, Q formed in place as dorg2r does
// -- no condition gate (CholeskyQR2 gives up beyond kappa ~ 1e6, which a square unfolding Psi_0 Omega_0^+ reaches
// easily).  A (m, n) row-major in, Q (m, n) row-major out.  Column-major working copy; every 16-lane group applies a
// reflector to its own columns.
__global__ __launch_bounds__(1024) void small_qr_kernel(double *__restrict__ A, int m, int n)
{
    extern __shared__ double sq[];
    double *W = sq;                    // m x n column-major
    double *tau = sq + (size_t)m * n;  // n
    const int tid = threadIdx.x, grp = tid >> 4, gl = tid & 15, ngrp = 64;
    for (int e = tid; e < m * n; e += 1024) W[(size_t)(e % n) * m + e / n] = A[e];
    __syncthreads();
    for (int j = 0; j < n; ++j) {
        double *cj = W + (size_t)j * m;
        double sig = 0.0;
        for (int i = j + 1 + gl; i < m; i += 16) sig = fma(cj[i], cj[i], sig);
        sig = row_sum16(sig);
        const double alpha = cj[j];
        double beta = alpha, t = 0.0, scale = 0.0;
        if (sig != 0.0) {                                              // LAPACK dlarfg
            const double nrm = sqrt(alpha * alpha + sig);
            beta = alpha >= 0 ? -nrm : nrm;
            t = (beta - alpha) / beta;
            scale = 1.0 / (alpha - beta);
        }
        for (int k = j + 1 + grp; k < n; k += ngrp) {
            double *ck = W + (size_t)k * m;
            double w = gl == 0 ? ck[j] : 0.0;                          // v[0] = 1
            for (int i = j + 1 + gl; i < m; i += 16) w = fma(cj[i] * scale, ck[i], w);
            w = row_sum16(w) * t;
            if (gl == 0) ck[j] -= w;
            for (int i = j + 1 + gl; i < m; i += 16) ck[i] = fma(-w, cj[i] * scale, ck[i]);
        }
        __syncthreads();
        for (int i = j + 1 + tid; i < m; i += 1024) cj[i] *= scale;    // v below the diagonal (v[0] = 1 implied); R is not kept
        if (tid == 0) tau[j] = t;
        __syncthreads();
    }
    // Q = H_0 ... H_{n-1} [I; 0] in place (dorg2r): from the last reflector to the first
    for (int j = n - 1; j >= 0; --j) {
        double *vj = W + (size_t)j * m;
        const double t = tau[j];
        for (int k = j + 1 + grp; k < n; k += ngrp) {                  // columns > j already hold columns of Q (zero above row j + 1)
            double *qk = W + (size_t)k * m;
            double w = 0.0;                                            // row j of column k is still zero
            for (int i = j + 1 + gl; i < m; i += 16) w = fma(vj[i], qk[i], w);
            w = row_sum16(w) * t;
            if (gl == 0) qk[j] = -w;
            for (int i = j + 1 + gl; i < m; i += 16) qk[i] = fma(-w, vj[i], qk[i]);
        }
        __syncthreads();
        for (int i = tid; i < m; i += 1024) vj[i] = i < j ? 0.0 : (i == j ? 1.0 - t : -t * vj[i]);
        __syncthreads();
    }
    for (int e = tid; e < m * n; e += 1024) A[e] = W[(size_t)(e % n) * m + e / n];
}

// The same signs for n beyond one workgroup's LDS (129..256): the working copy is B itself in global memory (L2), one
// column per step, 1024 threads.  Only the signs are needed, so the trailing update uses the unscaled columns.
__global__ __launch_bounds__(1024) void hh_sign_scale_global_kernel(double *__restrict__ B, int n, int square, double *__restrict__ Rinv)
{
    __shared__ double S[CHOL_MAX_N];
    const int tid = threadIdx.x;
    for (int j = 0; j < n; ++j) {
        const double piv = B[j * n + j];
        double sgn = piv >= 0.0 ? -1.0 : 1.0;
        if (square && j == n - 1) sgn = -sgn;
        if (tid == 0) S[j] = sgn;
        const double pinv = 1.0 / (piv - sgn);
        const int rem = n - j - 1;
        for (int e = tid; e < rem * rem; e += 1024) {
            const int i = j + 1 + e / rem, c = j + 1 + e % rem;
            B[i * n + c] = fma(-B[i * n + j] * pinv, B[j * n + c], B[i * n + c]);
        }
        __threadfence_block();
        __syncthreads();
    }
    for (int e = tid; e < n * n; e += 1024) Rinv[e] *= S[e % n];
}

static int small_gemm(int64_t M, int64_t N, int64_t K, const double *A, int64_t a_m, int64_t a_k, const double *B,
                      int64_t b_k, int64_t b_n, double *C, int stream)
{
    ttsk_gemm_desc d{};
    d.batch = 1; d.M = M; d.N = N; d.Ko = 1; d.Ki = K;
    d.a_m = a_m; d.a_ki = a_k; d.b_ki = b_k; d.b_n = b_n; d.c_m = N; d.c_n = 1;
    d.alpha = 1.0;
    return ttsk_gemm(&d, A, B, C, nullptr, stream);
}

static unsigned chol_threads(int n)
{
    static const int wide = [] { const char *e = getenv("TTSK_CHOL_WIDE"); return e ? atoi(e) : 1; }();
    return (wide && n <= (wide == 1 ? 128 : 64)) ? 1024u : 256u;
}

static int launch_chol(const double *G, int n, double *Rinv, double *Ginv, int *status, double cond_tol, hipStream_t st,
                       int *sticky = nullptr, double *pminmax = nullptr, int count = 1)
{
    static PerInit attr;
    if (attr.first()) {
        TTSK_HIP(hipFuncSetAttribute((const void *)chol_inv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
        TTSK_HIP(hipFuncSetAttribute((const void *)hh_sign_scale_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
    }
    hipLaunchKernelGGL(chol_inv_kernel, dim3((unsigned)count), dim3(chol_threads(n)), (size_t)(n * (n + 1) + n) * 8, st, G, n, Rinv, Ginv, status,
                       cond_tol, sticky, pminmax);
    TTSK_LAUNCH_CHECK();
    return TTSK_OK;
}

__global__ void add_diag_kernel(double *E, int n, double v)
{
    E += (size_t)blockIdx.x * n * n;               // batch: one n x n matrix per workgroup
    for (int i = threadIdx.x; i < n; i += blockDim.x) E[(size_t)i * n + i] += v;
}

static int gemm_ex(int64_t M, int64_t N, int64_t K, const double *A, int64_t a_m, int64_t a_k, const double *B, int64_t b_k,
                   int64_t b_n, double *C, int64_t c_m, double alpha, int accumulate, int stream)
{
    ttsk_gemm_desc d{};
    d.batch = 1; d.M = M; d.N = N; d.Ko = 1; d.Ki = K;
    d.a_m = a_m; d.a_ki = a_k; d.b_ki = b_k; d.b_n = b_n; d.c_m = c_m; d.c_n = 1;
    d.alpha = alpha; d.accumulate = accumulate;
    return ttsk_gemm(&d, A, B, C, nullptr, stream);
}

// verdict of a two-block factorisation: both blocks accepted AND the pivots of the whole matrix within cond_tol
__global__ void chol_combine_kernel(const double *pm, const int *st2, double cond_tol, int *status, int *sticky)
{
    const double lo = fmin(pm[0], pm[2]), hi = fmax(pm[1], pm[3]);
    const int rej = (st2[0] || st2[1] || !(lo > 0.0) || lo < cond_tol * cond_tol * hi) ? 1 : 0;
    status[0] = rej;
    if (rej && sticky) *sticky = 1;
}

// The same contract as chol_inv_kernel for n up to 256: beyond 128 (the one-workgroup kernel's LDS) a 2 x 2 block
// factorisation -- R11 = chol(G11), R12 = R11^-T G12, R22 = chol(G22 - R12^T R12), R^-1 = [X11, -X11 R12 X22; 0, X22] --
// i.e. two one-workgroup factorisations and a handful of small products (rank 145 / 290 of scripts/plot_timings.py).
// ws: chol_ws_elems(n) doubles from the caller's arena (nested scratch() calls would move it).
constexpr int CHOL_ONE = 128, CHOL_MAX = 256;
static size_t chol_ws_elems(int n) { return n <= CHOL_ONE ? 0 : (size_t)6 * CHOL_ONE * CHOL_ONE + 16; }

static int chol_inv_any(const double *G, int n, double *Rinv, double *Ginv, int *status, double cond_tol, int stream,
                        hipStream_t st, double *ws, int *sticky = nullptr)
{
    if (n <= CHOL_ONE) return launch_chol(G, n, Rinv, Ginv, status, cond_tol, st, sticky);
    if (n > CHOL_MAX || !ws) return TTSK_ERR_UNSUPPORTED;
    const int n1 = ((n + 1) / 2 + 15) & ~15, n2 = n - n1;
    double *G11 = ws, *X11 = G11 + (size_t)n1 * n1, *R12 = X11 + (size_t)n1 * n1, *S = R12 + (size_t)n1 * n2;
    double *X22 = S + (size_t)n2 * n2, *Y = X22 + (size_t)n2 * n2, *pm = Y + (size_t)n1 * n2;
    int *st2 = (int *)(pm + 4);
    int rc;
    TTSK_HIP(hipMemcpy2DAsync(G11, (size_t)n1 * 8, G, (size_t)n * 8, (size_t)n1 * 8, n1, hipMemcpyDeviceToDevice, st));
    if ((rc = launch_chol(G11, n1, X11, nullptr, st2, 0.0, st, nullptr, pm))) return rc;
    if ((rc = gemm_ex(n1, n2, n1, X11, 1, n1, G + n1, n, 1, R12, n2, 1.0, 0, stream))) return rc;           // R12 = X11^T G12
    TTSK_HIP(hipMemcpy2DAsync(S, (size_t)n2 * 8, G + (size_t)n1 * n + n1, (size_t)n * 8, (size_t)n2 * 8, n2, hipMemcpyDeviceToDevice, st));
    if ((rc = gemm_ex(n2, n2, n1, R12, 1, n2, R12, n2, 1, S, n2, -1.0, 1, stream))) return rc;               // S = G22 - R12^T R12
    if ((rc = launch_chol(S, n2, X22, nullptr, st2 + 1, 0.0, st, nullptr, pm + 2))) return rc;
    hipLaunchKernelGGL(chol_combine_kernel, dim3(1), dim3(1), 0, st, pm, st2, cond_tol, status, sticky);
    TTSK_LAUNCH_CHECK();
    if ((rc = gemm_ex(n1, n2, n2, R12, n2, 1, X22, n2, 1, Y, n2, 1.0, 0, stream))) return rc;                 // Y = R12 X22
    TTSK_HIP(hipMemsetAsync(Rinv, 0, (size_t)n * n * 8, st));
    TTSK_HIP(hipMemcpy2DAsync(Rinv, (size_t)n * 8, X11, (size_t)n1 * 8, (size_t)n1 * 8, n1, hipMemcpyDeviceToDevice, st));
    TTSK_HIP(hipMemcpy2DAsync(Rinv + (size_t)n1 * n + n1, (size_t)n * 8, X22, (size_t)n2 * 8, (size_t)n2 * 8, n2, hipMemcpyDeviceToDevice, st));
    if ((rc = gemm_ex(n1, n2, n1, X11, n1, 1, Y, n2, 1, Rinv + n1, n, -1.0, 0, stream))) return rc;           // X12 = -X11 Y
    if (Ginv && (rc = gemm_ex(n, n, n, Rinv, n, 1, Rinv, 1, n, Ginv, n, 1.0, 0, stream))) return rc;          // G^-1 = X X^T
    return TTSK_OK;
}

bool fast_solves()
{
    static int v = [] { const char *e = getenv("TTSK_FAST_SOLVES"); return e ? atoi(e) : 1; }();
    return v != 0;
}

// pinv(Omega) through the normal equations; 1 = done, 0 = rejected (caller runs the Jacobi SVD)
// Two phases, so that the d - 1 pseudo-inverses of an assembly can be in flight on their own streams
// before the host looks at the first verdict: `begin` queues Gram matrix, factorisation, the product
// Omega^T G^-1 (speculatively: it is overwritten if the factorisation is rejected) and the copy of the
// verdict into a pinned per-stream slot; `verdict` waits for the stream and reads it.
static int *pinv_host_status()
{
    return (int *)persistent_alloc(PA_PINV_HOST, TTSK_NUM_STREAMS * sizeof(int), true, false);
}
static int g_pinv_began[TTSK_NUM_STREAMS];
// the verdict of the attempt on the device, outside the scratch arena (which the Jacobi fallback reuses)
static int *pinv_dev_status(int stream)
{
    int *p = (int *)persistent_alloc(PA_PINV_DEV, TTSK_NUM_STREAMS * sizeof(int), false, false);
    return p ? p + stream : nullptr;
}

// deferred verdicts (ttsk_orth_step): one sticky word per stream, set by any rejected fast-path factorisation since the
// last ttsk_deferred_status
int *deferred_flag(int stream)
{
    int *p = (int *)persistent_alloc(PA_DEFERRED, TTSK_NUM_STREAMS * sizeof(int), false, true);
    return p ? p + stream : nullptr;
}

static size_t pinv_ws_elems(int n, int64_t mx) { return (size_t)3 * n * n + 16 + chol_ws_elems(n) + (size_t)mx * n; }   // mx = max(l, r)

// 1 = attempt queued, 0 = not applicable, < 0 = error.  ws_in: pinv_ws_elems(n) doubles of the caller's, or nullptr
// (then from the stream's arena).  sticky: deferred mode -- no copy of the verdict to the host, the rejection is
// recorded in *sticky and the caller decides at the end.
static int pinv_cholesky_begin(const double *omega, int64_t l, int64_t r, double *pinv, int stream, hipStream_t st,
                               double *ws_in = nullptr, int *sticky = nullptr)
{
    const int n = (int)(l <= r ? l : r);
    int *hs = pinv_host_status();
    int *status = pinv_dev_status(stream);
    if (n > CHOL_MAX || !hs || !status) return 0;
    const int64_t mx = l > r ? l : r;
    double *ws = ws_in ? ws_in : (double *)scratch(stream, SCRATCH_MISC, pinv_ws_elems(n, mx) * 8);
    if (!ws) return TTSK_ERR_HIP;
    double *G = ws, *Rinv = ws + n * n, *Ginv = ws + 2 * n * n, *cws = ws + 3 * n * n + 16;
    double *x1 = cws + chol_ws_elems(n);        // r x l: the refined pseudo-inverse before it replaces the first one
    int rc;
    if (l <= r) rc = small_gemm(l, l, r, omega, r, 1, omega, 1, r, G, stream);          // Omega Omega^T
    else        rc = small_gemm(r, r, l, omega, 1, r, omega, r, 1, G, stream);          // Omega^T Omega
    if (rc) return rc;
    // normal equations square the condition number (error kappa^2 eps); the Newton-Schulz step below brings that back to
    // ~kappa eps, so kappa(Omega) up to 3e4 is accepted (error <= ~1e-11 either way)
    // (the refinement only where a rejection is expensive -- the deferred mode of ttsk_orth_step repeats the whole sketch;
    // ttsk_pinv has the Jacobi kernel queued behind each attempt and keeps the plain gate at kappa = 300)
    const bool refine = sticky != nullptr;
    rc = chol_inv_any(G, n, Rinv, Ginv, status, refine ? 1.0 / 3.0e4 : 1.0 / 300.0, stream, st, n > CHOL_ONE ? cws : nullptr, sticky);
    if (rc) return rc;
    if (!sticky) {
        hs[stream] = 1;
        TTSK_HIP(hipMemcpyAsync(hs + stream, status, sizeof(int), hipMemcpyDeviceToHost, st));
    }
    if (l <= r) rc = small_gemm(r, l, l, omega, 1, r, Ginv, l, 1, pinv, stream);        // X0 = Omega^T G^-1
    else        rc = small_gemm(r, l, r, Ginv, r, 1, omega, 1, r, pinv, stream);        // X0 = G^-1 Omega^T
    if (rc) return rc;
    if (!refine) return 1;
    // One Newton-Schulz step squares the residual of the normal-equations inverse (kappa^2 eps -> ~kappa eps) and keeps
    // the minimum-norm property (X stays in the row / column space of Omega): X1 = X0 (2 I - Omega X0)  (l <= r) or
    // (2 I - X0 Omega) X0.  With it the acceptance gate above can sit at kappa ~ 1e5 instead of 300.
    double *E = G;                      // n x n, free again
    if (l <= r) {
        if ((rc = gemm_ex(l, l, r, omega, r, 1, pinv, l, 1, E, l, -1.0, 0, stream))) return rc;          // E = -Omega X0
        hipLaunchKernelGGL(add_diag_kernel, dim3(1), dim3(256), 0, st, E, (int)l, 2.0);                  // E = 2 I - Omega X0
        TTSK_LAUNCH_CHECK();
        if ((rc = gemm_ex(r, l, l, pinv, l, 1, E, l, 1, x1, l, 1.0, 0, stream))) return rc;               // X1 = X0 E
    } else {
        if ((rc = gemm_ex(r, r, l, pinv, l, 1, omega, r, 1, E, r, -1.0, 0, stream))) return rc;          // E = -X0 Omega
        hipLaunchKernelGGL(add_diag_kernel, dim3(1), dim3(256), 0, st, E, (int)r, 2.0);
        TTSK_LAUNCH_CHECK();
        if ((rc = gemm_ex(r, l, r, E, r, 1, pinv, l, 1, x1, l, 1.0, 0, stream))) return rc;               // X1 = E X0
    }
    TTSK_HIP(hipMemcpyAsync(pinv, x1, (size_t)r * l * 8, hipMemcpyDeviceToDevice, st));
    return 1;
}
// one pseudo-inverse with a deferred verdict (ranks up to 256); ws: pinv_deferred_ws_elems doubles
size_t pinv_deferred_ws_elems(int64_t l, int64_t r) { return pinv_ws_elems((int)(l < r ? l : r), l > r ? l : r); }
int pinv_deferred(const double *omega, int64_t l, int64_t r, double *pinv, int stream, hipStream_t st, double *ws, int *sticky)
{
    return pinv_cholesky_begin(omega, l, r, pinv, stream, st, ws, sticky);
}

// 1 = accepted (pinv is final), 0 = rejected
static int pinv_cholesky_verdict(int64_t l, int64_t r, int stream, hipStream_t st)
{
    TTSK_HIP(hipStreamSynchronize(st));
    const int host_status = pinv_host_status()[stream];
    static int trace = [] { const char *e = getenv("TTSK_GEMM_TRACE"); return e ? atoi(e) : 0; }();
    if (trace) fprintf(stderr, "ttsk_pinv %lld x %lld: normal equations %s\n", (long long)l, (long long)r,
                       host_status ? "rejected -> Jacobi SVD" : "accepted");
    return host_status ? 0 : 1;
}

size_t qr_ws_elems(int64_t m, int n) { return (size_t)m * n + 4 * (size_t)n * n + 16 + chol_ws_elems(n); }

// thin QR by CholeskyQR2 + Householder sign reconstruction; 1 = done, 0 = rejected.  sticky: deferred mode -- the
// factorisation always runs to the end (A is overwritten either way), a rejection is recorded in *sticky.
static int launch_cholqr2_lds(double *M, int64_t m, int n, int *status, double cond_tol, int *sticky, hipStream_t st);

int qr_cholesky(double *A, int64_t m, int64_t n64, int stream, hipStream_t st, double *ws_in, int *sticky, bool unsigned_q)
{
    const int n = (int)n64;
    if (n > CHOL_MAX || m < n) return 0;
    if (m < 2 * n64 && (size_t)m * n + n <= SMALL_QR_MAX) {
        // nearly square and small: Householder in one workgroup, no gate to fail
        static PerInit attr;
        if (attr.first()) {
            TTSK_HIP(hipFuncSetAttribute((const void *)small_qr_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
        }
        hipLaunchKernelGGL(small_qr_kernel, dim3(1), dim3(1024), ((size_t)m * n + n) * 8, st, A, (int)m, n);
        TTSK_LAUNCH_CHECK();
        return unsigned_q ? 2 : 1;        // 2: Q carries LAPACK's signs already
    }
    double *ws = ws_in ? ws_in : (double *)scratch(stream, SCRATCH_MISC, qr_ws_elems(m, n) * 8);
    if (!ws) return TTSK_ERR_HIP;
    double *Q1 = ws, *G = Q1 + (size_t)m * n, *R1 = G + n * n, *R2 = R1 + n * n, *Qtop = R2 + n * n;
    int *status = (int *)(Qtop + n * n);
    static const int small_on = [] { const char *e = getenv("TTSK_CHOLQR2_LDS"); return e ? atoi(e) : 1; }();
    if (unsigned_q && sticky && small_on) {
        // small enough for one workgroup's LDS: the whole CholeskyQR2 in one launch
        const int fr = launch_cholqr2_lds(A, m, n, status, 1e-6, sticky, st);
        if (fr) return fr;
    }
    double *cws = n > CHOL_ONE ? Qtop + n * n + 16 : nullptr;
    int rc;
    if ((rc = small_gemm(n, n, m, A, 1, n, A, n, 1, G, stream))) return rc;              // A^T A
    if ((rc = chol_inv_any(G, n, R1, nullptr, status, 1e-6, stream, st, cws, sticky))) return rc;       // kappa(A) up to ~1e6
    if ((rc = small_gemm(m, n, n, A, n, 1, R1, n, 1, Q1, stream))) return rc;            // Q1 = A R1^-1
    if ((rc = small_gemm(n, n, m, Q1, 1, n, Q1, n, 1, G, stream))) return rc;            // Q1^T Q1
    if (unsigned_q && n > CHOL_ONE) {
        if ((rc = chol_inv_any(G, n, R2, nullptr, status + 1, 0.5, stream, st, cws, sticky))) return rc;    // must be ~identity
    } else if (unsigned_q) {
        // R with positive diagonal only: the caller reconstructs the signs beside the critical path (qr_signs)
        static const int chol_expand = [] { const char *e = getenv("TTSK_CHOL_EXPAND"); return e ? atoi(e) : 1; }();
        hipLaunchKernelGGL(chol_inv_kernel, dim3(1), dim3(256), (size_t)(n * (n + 1) + n) * 8, st, G, n, R2, (double *)nullptr,
                           status + 1, 0.5, sticky, (double *)nullptr, (const double *)nullptr, 0, chol_expand ? 2 : 0);
        TTSK_LAUNCH_CHECK();
    } else if (n <= CHOL_SIGN_MAX) {
        // second factorisation (G ~ identity), top block of Q and the sign reconstruction in ONE kernel
        static const int chol_expand = [] { const char *e = getenv("TTSK_CHOL_EXPAND"); return e ? atoi(e) : 1; }();
        static PerInit attr;
        if (attr.first()) {
            TTSK_HIP(hipFuncSetAttribute((const void *)chol_inv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
        }
        hipLaunchKernelGGL(chol_inv_kernel, dim3(1), dim3(256), (size_t)(3 * n * (n + 1) + 2 * n) * 8, st, G, n, R2, (double *)nullptr,
                           status + 1, 0.5, sticky, (double *)nullptr, (const double *)Q1, m == n64 ? 1 : 0, chol_expand);
        TTSK_LAUNCH_CHECK();
    } else {
    if ((rc = chol_inv_any(G, n, R2, nullptr, status + 1, 0.5, stream, st, cws, sticky))) return rc;    // must be ~identity
    if ((rc = small_gemm(n, n, n, Q1, n, 1, R2, n, 1, Qtop, stream))) return rc;         // top block of Q
    if (n <= CHOL_ONE) {
        hipLaunchKernelGGL(hh_sign_scale_kernel, dim3(1), dim3(256), (size_t)(n * (n + 1) + n) * 8, st, Qtop, n,
                           m == n64 ? 1 : 0, R2);
    } else {
        // beyond one workgroup's LDS: the same modified LU with the working copy in global memory (Qtop itself)
        hipLaunchKernelGGL(hh_sign_scale_global_kernel, dim3(1), dim3(1024), 0, st, Qtop, n, m == n64 ? 1 : 0, R2);
    }
    TTSK_LAUNCH_CHECK();
    }
    if (!sticky) {
        int host_status[2] = {1, 1};
        TTSK_HIP(hipMemcpyAsync(host_status, status, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
        TTSK_HIP(hipStreamSynchronize(st));
        if (host_status[0] || host_status[1]) return 0;
    }
    if ((rc = small_gemm(m, n, n, Q1, n, 1, R2, n, 1, A, stream))) return rc;            // Q = Q1 R2^-1 S
    return 1;
}

// CholeskyQR2 of a SMALL tall matrix in ONE workgroup (m (n + 2) + n (n + 1) + n doubles fit the LDS: the first mode of an
// orthogonalising sketch, 200 x 50 at C3): M (m x n, row-major, in place) -> Q with the signs of CholeskyQR (R's diagonal
// positive; the caller reconstructs Householder's signs with qr_signs).  Both Gram matrices, both factorisations, both
// products without leaving the LDS: one launch instead of eight (Gram + reduce, factorisation, product, twice).
// status[0] / status[1]: verdicts of the two factorisations as chol_inv_kernel gives them (gates cond_tol, 0.5).
constexpr size_t CHOLQR2_LDS_MAX = 19000;       // doubles
__global__ __launch_bounds__(1024) void cholqr2_lds_kernel(double *__restrict__ M, int m, int n, int *__restrict__ status,
                                                           double cond_tol, int *__restrict__ sticky)
{
    extern __shared__ double sm[];
    __shared__ double shadow[2 * 128];
    const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wv = tid >> 6, nw = nthr >> 6, x16 = lane & 15, g = lane >> 4;
    // operands zero-padded to whole tiles, so that the matrix-core loops carry no bounds
    const int nt = (n + 15) >> 4, mt = (m + 15) >> 4, np = 16 * nt, mp = 16 * mt, ldm = np + 2, ld = np + 1;
    double *Ms = sm, *A = sm + (size_t)mp * ldm, *xd = A + (size_t)np * ld;
    const bool core = tid < 256;
    for (int e = tid; e < mp * ldm; e += nthr) Ms[e] = 0.0;
    for (int e = tid; e < np * ld + np; e += nthr) A[e] = 0.0;
    __syncthreads();
    for (int e0 = tid; e0 < m * n; e0 += 8 * nthr) {          // eight loads in flight per thread
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = e0 + u * nthr < m * n ? M[e0 + u * nthr] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + u * nthr;
            if (e < m * n) Ms[(e / n) * ldm + e % n] = v[u];
        }
    }
    __syncthreads();
    // A = Ms^T Ms: one tile pair (t1 <= t2) per wave and turn, the whole column of row blocks
    auto gram = [&]() {
        int q = 0;
        for (int t1 = 0; t1 < nt; ++t1)
            for (int t2 = t1; t2 < nt; ++t2, ++q) {
                if (q % nw != wv) continue;
                v4d acc = {0.0, 0.0, 0.0, 0.0};
                const double *pa = Ms + g * ldm + 16 * t1 + x16, *pb = Ms + g * ldm + 16 * t2 + x16;
                // four k-blocks per step (a row tile of Ms), the next step's operands read before this step's matrix instructions
                double a[4], b[4], an[4], bn[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { a[u] = pa[4 * u * ldm]; b[u] = pb[4 * u * ldm]; }
                for (int s4 = 0; s4 < mt; ++s4) {
                    const int nx = s4 + 1 < mt ? s4 + 1 : s4;
#pragma unroll
                    for (int u = 0; u < 4; ++u) { an[u] = pa[(16 * nx + 4 * u) * ldm]; bn[u] = pb[(16 * nx + 4 * u) * ldm]; }
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc = mfma16(a[u], b[u], acc);
#pragma unroll
                    for (int u = 0; u < 4; ++u) { a[u] = an[u]; b[u] = bn[u]; }
                }
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int r = 16 * t1 + g + 4 * v, c = 16 * t2 + x16;
                    if (r < n && c < n) { A[r * ld + c] = acc[v]; A[c * ld + r] = acc[v]; }
                }
            }
    };
    // X from chol_lds's storage (strict upper part transposed into the lower triangle, diagonal in xd) to a dense upper
    // triangular matrix in A, in place
    auto densify = [&]() {
        for (int e = tid; e < n * n; e += nthr) {
            const int i = e / n, c = e - i * n;
            if (i < c) { const double t = A[c * ld + i]; A[i * ld + c] = t; A[c * ld + i] = 0.0; }
            else if (i == c) A[i * ld + i] = xd[i];
        }
    };
    // rows of Ms (or of the output) <- rows of Ms times A; a wave owns its row tiles
    auto apply = [&](double *out, int ldo) {
        for (int tile = wv; tile < mt; tile += nw) {
            v4d acc[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = v4d{0.0, 0.0, 0.0, 0.0};
            const double *pa = Ms + (16 * tile + x16) * ldm + g, *pb = A + g * ld + x16;
            // the operands of k-block kb + 1 are read before the matrix instructions of k-block kb
            double a = pa[0], b[4], an, bn[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) b[t] = t < nt ? pb[16 * t] : 0.0;
            for (int kb = 0; kb < 4 * nt; ++kb) {
                const int nx = kb + 1 < 4 * nt ? kb + 1 : kb;
                an = pa[4 * nx];
#pragma unroll
                for (int t = 0; t < 4; ++t) bn[t] = t < nt ? pb[4 * nx * ld + 16 * t] : 0.0;
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (t < nt) acc[t] = mfma16(a, b[t], acc[t]);
                a = an;
#pragma unroll
                for (int t = 0; t < 4; ++t) b[t] = bn[t];
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int r = 16 * tile + g + 4 * v, c = 16 * t + x16;
                    if (t < nt && r < m && c < n) out[(size_t)r * ldo + c] = acc[t][v];
                }
        }
    };
    gram();
    __syncthreads();
    int bad = 0;
    double pmin = 1e300, pmax = 0.0;
    chol_lds(A, xd, n, ld, shadow, tid, nthr, bad, pmin, pmax);
    if (tid == 0) {
        const int rej = (bad || pmin < cond_tol * cond_tol * pmax) ? 1 : 0;
        status[0] = rej;
        if (rej && sticky) *sticky = 1;
    }
    densify();
    __syncthreads();
    apply(Ms, ldm);                                        // Q1 = M R1^-1 in place
    __syncthreads();
    gram();                                                // overwrites A's n x n block (both triangles)
    __syncthreads();
    // second factor: I - Phi(E) where the Gram matrix is the identity to 1e-8 / n, the recurrence otherwise
    double em = 0.0;
    if (core) {
        for (int e = tid; e < n * n; e += 256) {
            const int i = e / n, c = e - i * n;
            em = fmax(em, fabs(A[i * ld + c] - (i == c ? 1.0 : 0.0)));
        }
        em = fmax(em, jac_dpp<0xB1>(em));
        em = fmax(em, jac_dpp<0x4E>(em));
        if ((tid & 3) == 0) shadow[tid >> 2] = em;
    }
    __syncthreads();
    em = 0.0;
    for (int k = 0; k < 64; ++k) em = fmax(em, shadow[k]);
    __syncthreads();
    if (em * n <= 1e-8) {
        for (int e = tid; e < n * n; e += nthr) {           // dense X2 = I - Phi(E) in place
            const int i = e / n, c = e - i * n;
            if (i < c) A[i * ld + c] = -A[i * ld + c];
            else if (i == c) A[i * ld + i] = 1.0 - 0.5 * (A[i * ld + i] - 1.0);
        }
        __syncthreads();
        for (int e = tid; e < n * n; e += nthr) {
            const int i = e / n, c = e - i * n;
            if (i > c) A[i * ld + c] = 0.0;
        }
        if (tid == 0) status[1] = 0;
    } else {
        bad = 0; pmin = 1e300; pmax = 0.0;
        chol_lds(A, xd, n, ld, shadow, tid, nthr, bad, pmin, pmax);
        if (tid == 0) {
            const int rej = (bad || pmin < 0.25 * pmax) ? 1 : 0;
            status[1] = rej;
            if (rej && sticky) *sticky = 1;
        }
        densify();
    }
    __syncthreads();
    apply(M, n);                                           // Q = Q1 R2^-1
}

// 1 = queued, 0 = does not fit
static int launch_cholqr2_lds(double *M, int64_t m, int n, int *status, double cond_tol, int *sticky, hipStream_t st)
{
    if (n > 64 || m < n || m > 4096) return 0;
    const size_t np = 16 * (size_t)((n + 15) >> 4), mp = 16 * (size_t)((m + 15) >> 4);
    const size_t elems = mp * (np + 2) + np * (np + 1) + np;
    if (elems > CHOLQR2_LDS_MAX) return 0;
    static PerInit attr;
    if (attr.first()) {
        TTSK_HIP(hipFuncSetAttribute((const void *)cholqr2_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
    }
    const size_t lds = elems * 8;
    hipLaunchKernelGGL(cholqr2_lds_kernel, dim3(1), dim3(1024), lds, st, M, (int)m, n, status, cond_tol, sticky);
    TTSK_LAUNCH_CHECK();
    return 1;
}

// The Householder column signs of Q = D Qc (Qc: top n x n block of a CholeskyQR factor with positive diagonal R, D a
// +-1 scaling of its rows given per group of `rows_per` rows -- the signs of the previous mode's factor, which scale the
// rows of this mode's unfolding): S[c] for the caller to apply whenever it likes.  One workgroup.
__global__ __launch_bounds__(1024) void qr_signs_kernel(const double *__restrict__ Qtop, int n, int square,
                                                       const double *__restrict__ Sprev, int rows_per, double *__restrict__ Sout)
{
    extern __shared__ double sm[];
    const int ld = n + 1, tid = threadIdx.x;
    double *B = sm, *S = sm + n * ld;
    __shared__ double shadow[2 * 128];
    const int nthr = blockDim.x;
    for (int e = tid; e < n * n; e += nthr) {
        const int r = e / n;
        B[r * ld + e % n] = Qtop[e] * (Sprev ? Sprev[r / rows_per] : 1.0);
    }
    __syncthreads();
    hh_signs_lds(B, n, ld, square, S, shadow, tid, nthr);
    for (int c = tid; c < n; c += nthr) Sout[c] = S[c];
}

// the same for 128 < n <= 256: the working copy B (n x n) lives in global memory (L2), one column per step
__global__ __launch_bounds__(1024) void qr_signs_global_kernel(const double *__restrict__ Qtop, int n, int square,
                                                               const double *__restrict__ Sprev, int rows_per,
                                                               double *__restrict__ B, double *__restrict__ Sout)
{
    const int tid = threadIdx.x;
    for (int e = tid; e < n * n; e += 1024) B[e] = Qtop[e] * (Sprev ? Sprev[(e / n) / rows_per] : 1.0);
    __threadfence_block();
    __syncthreads();
    for (int j = 0; j < n; ++j) {
        const double piv = B[j * n + j];
        double sgn = piv >= 0.0 ? -1.0 : 1.0;
        if (square && j == n - 1) sgn = -sgn;
        if (tid == 0) Sout[j] = sgn;
        const double pinv = 1.0 / (piv - sgn);
        const int rem = n - j - 1;
        for (int e = tid; e < rem * rem; e += 1024) {
            const int i = j + 1 + e / rem, c = j + 1 + e % rem;
            B[i * n + c] = fma(-B[i * n + j] * pinv, B[j * n + c], B[i * n + c]);
        }
        __threadfence_block();
        __syncthreads();
    }
}

// work: n * n doubles for n > 128 (may be nullptr otherwise)
int qr_signs(const double *Qtop, int n, int square, const double *Sprev, int rows_per, double *Sout, hipStream_t st, double *work)
{
    if (n > CHOL_ONE) {
        if (n > CHOL_MAX || !work) return 0;
        hipLaunchKernelGGL(qr_signs_global_kernel, dim3(1), dim3(1024), 0, st, Qtop, n, square, Sprev, rows_per, work, Sout);
        TTSK_LAUNCH_CHECK();
        return 1;
    }
    static PerInit attr;
    if (attr.first()) {
        TTSK_HIP(hipFuncSetAttribute((const void *)qr_signs_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
    }
    hipLaunchKernelGGL(qr_signs_kernel, dim3(1), dim3(chol_threads(n)), (size_t)(n * (n + 1) + n) * 8, st, Qtop, n, square, Sprev, rows_per, Sout);
    TTSK_LAUNCH_CHECK();
    return 1;
}

// ---- the same steps for `count` matrices of ONE shape per launch (the tensors of an orthogonalising batch, tt_orth.hip)
struct QrSignsBatch {
    const double *Q[16], *Sprev[16];
    double *Sout[16];
};
__global__ __launch_bounds__(1024) void qr_signs_batch_kernel(QrSignsBatch a, int n, int square, int rows_per)
{
    extern __shared__ double sm[];
    const int ld = n + 1, tid = threadIdx.x;
    double *B = sm, *S = sm + n * ld;
    __shared__ double shadow[2 * 128];
    const int nthr = blockDim.x;
    const double *Qtop = a.Q[blockIdx.x], *Sprev = a.Sprev[blockIdx.x];
    double *Sout = a.Sout[blockIdx.x];
    for (int e = tid; e < n * n; e += nthr) {
        const int r = e / n;
        B[r * ld + e % n] = Qtop[e] * (Sprev ? Sprev[r / rows_per] : 1.0);
    }
    __syncthreads();
    hh_signs_lds(B, n, ld, square, S, shadow, tid, nthr);
    for (int c = tid; c < n; c += nthr) Sout[c] = S[c];
}

// 1 = queued, 0 = outside this path (n beyond one workgroup's LDS, more than 16 matrices)
int qr_signs_batch(int count, const double *const *Qtop, int n, int square, const double *const *Sprev, int rows_per, double *const *Sout,
                   hipStream_t st)
{
    if (n > CHOL_ONE || count < 1 || count > 16) return 0;
    static PerInit attr;
    if (attr.first()) {
        TTSK_HIP(hipFuncSetAttribute((const void *)qr_signs_batch_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
    }
    QrSignsBatch a{};
    for (int b = 0; b < count; ++b) { a.Q[b] = Qtop[b]; a.Sprev[b] = Sprev ? Sprev[b] : nullptr; a.Sout[b] = Sout[b]; }
    hipLaunchKernelGGL(qr_signs_batch_kernel, dim3((unsigned)count), dim3(chol_threads(n)), (size_t)(n * (n + 1) + n) * 8, st, a, n, square, rows_per);
    TTSK_LAUNCH_CHECK();
    return 1;
}

size_t qr_batch_ws_elems(int count, int64_t m, int n) { return (size_t)count * ((size_t)m * n + 3 * (size_t)n * n) + (size_t)count + 16; }

// CholeskyQR2 of `count` tall matrices of one shape (m x n row-major, in place), UNSIGNED factors (R's diagonal positive: the caller
// reconstructs Householder's signs, qr_signs_batch): two Gram products, two factorisations, two triangular products, each ONE
// launch over all matrices (tensor by tensor where a shape has no batched kernel).  Verdicts into *sticky (deferred).
// 1 = queued, 0 = outside this path -- nothing has been queued then.
int qr_cholesky_batch(int count, double *const *A, int64_t m, int n, int stream, hipStream_t st, double *ws, int *sticky)
{
    if (count < 1 || count > 16 || n > CHOL_ONE || m < 2 * (int64_t)n || !ws || !sticky) return 0;
    double *Q1 = ws, *G = Q1 + (size_t)count * m * n, *R1 = G + (size_t)count * n * n, *R2 = R1 + (size_t)count * n * n;
    int *status = (int *)(R2 + (size_t)count * n * n);
    const double *cA[16], *cQ1[16], *cR1[16], *cR2[16];
    double *pQ1[16], *pG[16];
    for (int b = 0; b < count; ++b) {
        cA[b] = A[b]; pQ1[b] = Q1 + (size_t)b * m * n; cQ1[b] = pQ1[b];
        pG[b] = G + (size_t)b * n * n; cR1[b] = R1 + (size_t)b * n * n; cR2[b] = R2 + (size_t)b * n * n;
    }
    auto desc = [](int64_t M, int64_t N, int64_t K, int64_t a_m, int64_t a_k, int64_t b_k, int64_t b_n) {
        ttsk_gemm_desc d{};
        d.batch = 1; d.M = M; d.N = N; d.Ko = 1; d.Ki = K;
        d.a_m = a_m; d.a_ki = a_k; d.b_ki = b_k; d.b_n = b_n; d.c_m = N; d.c_n = 1; d.alpha = 1.0;
        return d;
    };
    const ttsk_gemm_desc gram = desc(n, n, m, 1, n, n, 1), tri = desc(m, n, n, n, 1, n, 1);
    // one product for every matrix: a batched launch where the shape has one (tall: skinny.h; small: small.hip), else one by one
    auto prod = [&](const ttsk_gemm_desc &g, const double *const *X, const double *const *Y, double *const *Z) -> int {
        int r = skinny_try_batch(g, count, X, Y, Z, stream, st);
        if (r == 0) r = small_try_batch(g, count, X, Y, Z, stream, st);
        if (r != 0) return r < 0 ? r : TTSK_OK;
        for (int b = 0; b < count; ++b)
            if ((r = small_gemm(g.M, g.N, g.Ki, X[b], g.a_m, g.a_ki, Y[b], g.b_ki, g.b_n, Z[b], stream))) return r;
        return TTSK_OK;
    };
    int rc;
    if ((rc = prod(gram, cA, cA, pG))) return rc;                                                             // A^T A
    if ((rc = launch_chol(G, n, R1, nullptr, status, 1e-6, st, sticky, nullptr, count))) return rc;           // kappa(A) up to ~1e6
    if ((rc = prod(tri, cA, cR1, pQ1))) return rc;                                                            // Q1 = A R1^-1
    if ((rc = prod(gram, cQ1, cQ1, pG))) return rc;                                                           // Q1^T Q1 ~ identity
    static const int chol_expand = [] { const char *e = getenv("TTSK_CHOL_EXPAND"); return e ? atoi(e) : 1; }();
    hipLaunchKernelGGL(chol_inv_kernel, dim3((unsigned)count), dim3(256), (size_t)(n * (n + 1) + n) * 8, st, (const double *)G, n, R2, (double *)nullptr,
                       status + count, 0.5, sticky, (double *)nullptr, (const double *)nullptr, 0, chol_expand ? 2 : 0);
    TTSK_LAUNCH_CHECK();
    if ((rc = prod(tri, cQ1, cR2, A))) return rc;                                                             // Q = Q1 R2^-1
    return 1;
}

// core[a, i, b] *= sp[a] sn[b] for up to 16 cores in one launch (sp / sn may be nullptr = all ones)
struct SignFix { double *core[16]; const double *sp[16], *sn[16]; int k0[16], nn[16], k1[16]; };
__global__ __launch_bounds__(256) void apply_signs_kernel(SignFix f)
{
    const int q = blockIdx.y;
    double *c = f.core[q];
    const double *sp = f.sp[q], *sn = f.sn[q];
    const int k1 = f.k1[q], rows = f.k0[q] * f.nn[q], nn = f.nn[q];
    // a row (a, i) per 16-lane group and step, its k1 entries 16 at a time
    const int grp = threadIdx.x >> 4, x = threadIdx.x & 15;
    for (int r = blockIdx.x * 16 + grp; r < rows; r += gridDim.x * 16) {
        const double sr = sp ? sp[r / nn] : 1.0;
        double *row = c + (size_t)r * k1;
        for (int b = x; b < k1; b += 16) row[b] *= sn ? sr * sn[b] : sr;
    }
}
int apply_signs(int count, double *const *cores, const double *const *sp, const double *const *sn, const int *k0, const int *nn,
                const int *k1, hipStream_t st)
{
    for (int c0 = 0; c0 < count; c0 += 16) {
        SignFix f{};
        const int cnt = count - c0 < 16 ? count - c0 : 16;
        for (int q = 0; q < cnt; ++q) {
            f.core[q] = cores[c0 + q]; f.sp[q] = sp[c0 + q]; f.sn[q] = sn[c0 + q];
            f.k0[q] = k0[c0 + q]; f.nn[q] = nn[c0 + q]; f.k1[q] = k1[c0 + q];
        }
        hipLaunchKernelGGL(apply_signs_kernel, dim3(160, cnt), dim3(256), 0, st, f);
        TTSK_LAUNCH_CHECK();
    }
    return TTSK_OK;
}

}  // namespace ttsk

using namespace ttsk;

// where the Jacobi working set lives: 2 = W and V in LDS, 1 = W only, 0 = global scratch
static int jacobi_lds_mode(int64_t mW, int64_t nW, size_t *bytes)
{
    static const int off = getenv("TTSK_JACOBI_GLOBAL") ? 1 : 0;
    static PerInit attr_done;
    const size_t cap = 160 * 1024 - 256;            // 160 KB per CU minus the kernel's few static bytes
    const size_t w = (size_t)mW * nW * 8, v = (size_t)nW * nW * 8;
    const size_t small = ((size_t)nW + (nW + 1) / 2) * 8;          // sigma^2 and the sort order
    int mode = 0;
    *bytes = small;
    if (!off) {
        if (small + w + v <= cap) { mode = 2; *bytes = small + w + v; }
        else if (small + w <= cap) { mode = 1; *bytes = small + w; }
    }
    if (attr_done.first()) {
        if (const char *e = getenv("TTSK_JACOBI_PRECOND")) {
            const int v = atoi(e);
            (void)hipMemcpyToSymbol(HIP_SYMBOL(jac_precond_on), &v, sizeof(int));
        }
        if (hipFuncSetAttribute((const void *)jacobi_pinv_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cap) != hipSuccess ||
            hipFuncSetAttribute((const void *)jacobi_pinv_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cap) != hipSuccess ||
            hipFuncSetAttribute((const void *)jacobi_pinv_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cap) != hipSuccess) {
            set_error("jacobi: cannot raise the dynamic LDS limit");
            return -1;
        }
    }
    return mode;
}

extern "C" {

static double pinv_rcond(int64_t l, int64_t r, double rcond)
{
    if (rcond < 0) rcond = DBL_EPSILON;
    // Rank-decision floor: one-sided Jacobi returns the rounding noise of a numerically rank
    // deficient Omega as singular values of size ~eps*||Omega||; gelsd's eps*sigma_max rule then
    // becomes a coin flip and a kept noise direction is amplified by 1/sigma^2.  Anything within
    // 16*sqrt(max(l,r)) of that noise level is treated as zero (documented in DESIGN.md).
    const double floor_ = 16.0 * DBL_EPSILON * sqrt((double)(l > r ? l : r));
    return rcond < floor_ ? floor_ : rcond;
}

int ttsk_pinv_begin(const double *dev_omega, int64_t l, int64_t r, double rcond, double *dev_pinv, int stream)
{
    TTSK_STREAM(st, stream);
    TTSK_ARG(dev_omega && dev_pinv, "ttsk_pinv: NULL argument");
    TTSK_ARG(l >= 1 && r >= 1, "ttsk_pinv: bad shape (%lld, %lld)", (long long)l, (long long)r);
    TTSK_ARG((r >= l ? l : r) <= 1024, "ttsk_pinv: min(l, r) = %lld > 1024 unsupported", (long long)(r >= l ? l : r));
    g_pinv_began[stream] = 0;
    if (fast_solves() && pinv_rcond(l, r, rcond) <= 1e-4) {
        const int fr = pinv_cholesky_begin(dev_omega, l, r, dev_pinv, stream, st);
        if (fr < 0) return fr;
        g_pinv_began[stream] = fr;
    }
    return TTSK_OK;
}

int ttsk_pinv_end(const double *dev_omega, int64_t l, int64_t r, double rcond, double *dev_pinv,
                  int *host_rank, int stream)
{
    TTSK_STREAM(st, stream);
    TTSK_ARG(dev_omega && dev_pinv, "ttsk_pinv: NULL argument");
    TTSK_ARG(l >= 1 && r >= 1, "ttsk_pinv: bad shape (%lld, %lld)", (long long)l, (long long)r);
    const int transposed = r >= l;
    const int64_t mW = transposed ? r : l, nW = transposed ? l : r;
    TTSK_ARG(nW <= 1024, "ttsk_pinv: min(l, r) = %lld > 1024 unsupported", (long long)nW);
    rcond = pinv_rcond(l, r, rcond);
    const int *predicate = nullptr;
    if (g_pinv_began[stream]) {
        g_pinv_began[stream] = 0;
        if (!host_rank) {
            // nobody waits for the rank: the Jacobi kernel is queued behind the attempt and returns at once if the
            // attempt was accepted -- no read-back, the stream keeps running (to_tt: d - 1 of these per call)
            predicate = pinv_dev_status(stream);
        } else {
            const int fr = pinv_cholesky_verdict(l, r, stream, st);
            if (fr < 0) return fr;
            if (fr == 1) {
                *host_rank = (int)(l < r ? l : r);
                return TTSK_OK;
            }
        }
    }
    const size_t ws_elems = (size_t)(mW * nW + nW * nW) + 1;
    double *ws = (double *)scratch(stream, SCRATCH_MISC, ws_elems * 8);
    if (!ws) return TTSK_ERR_HIP;
    int *drank = (int *)(ws + mW * nW + nW * nW);
    size_t jl = 0;
    const int jm = jacobi_lds_mode(mW, nW, &jl);
    if (jm < 0) return TTSK_ERR_HIP;
    auto kern = jm == 2 ? jacobi_pinv_kernel<2> : (jm == 1 ? jacobi_pinv_kernel<1> : jacobi_pinv_kernel<0>);
    hipLaunchKernelGGL(kern, dim3(1), dim3(1024), jl, st, dev_omega, l, r, transposed, ws,
                       ws + mW * nW, rcond, dev_pinv, host_rank ? drank : (int *)nullptr, (double *)nullptr,
                       (double *)nullptr, (double *)nullptr, predicate, (int64_t)0, (int64_t)0);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && host_rank) {
        e = hipMemcpyAsync(host_rank, drank, sizeof(int), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    TTSK_HIP(e);
    return TTSK_OK;
}

// The pseudo-inverses of `count` matrices of ONE shape with ttsk_pinv's contract (gelsd's truncation on rejection), every
// stage of the fast attempt ONE batched launch and the Jacobi kernels queued behind it, each with its own matrix's verdict
// as predicate: 3 + count launches instead of 6 count (assemble_sketched_tt: the launches are what its d - 1 independent
// pseudo-inverses cost).  No read-back.  TTSK_ERR_UNSUPPORTED: min(l, r) > 128, count > 32, TTSK_FAST_SOLVES=0.
int ttsk_pinv_batch(int count, const double *const *dev_omegas, int64_t l, int64_t r, double *const *dev_pinvs, int stream)
{
    TTSK_STREAM(st, stream);
    TTSK_ARG(count >= 1 && count <= SK_MAXB && dev_omegas && dev_pinvs && l >= 1 && r >= 1, "ttsk_pinv_batch: bad argument");
    const int n = (int)(l < r ? l : r);
    // verdicts outside the scratch arena (the Jacobi kernel works there)
    int *vd = (int *)persistent_alloc(PA_PINV_BATCH_VD, TTSK_NUM_STREAMS * SK_MAXB * sizeof(int), false, false);
    if (!fast_solves() || !vd || n > CHOL_ONE || pinv_rcond(l, r, -1.0) > 1e-4) {
        set_error("ttsk_pinv_batch: (%lld x %lld) is outside the batched fast path", (long long)l, (long long)r);
        return TTSK_ERR_UNSUPPORTED;
    }
    int *status = vd + (size_t)stream * SK_MAXB;
    const int transposed = r >= l;
    const int64_t mW = transposed ? r : l, nW = transposed ? l : r;
    const size_t jws = (size_t)(mW * nW + nW * nW) + 1;
    double *ws = (double *)scratch(stream, SCRATCH_MISC, (jws + (size_t)count * 3 * n * n + 64) * 8);
    if (!ws) return TTSK_ERR_HIP;
    double *G0 = ws + jws, *R0 = G0 + (size_t)count * n * n, *I0 = R0 + (size_t)count * n * n;
    const double *Om[SK_MAXB], *cI[SK_MAXB];
    double *G[SK_MAXB], *P[SK_MAXB];
    for (int b = 0; b < count; ++b) {
        TTSK_ARG(dev_omegas[b] && dev_pinvs[b], "ttsk_pinv_batch: NULL matrix %d", b);
        Om[b] = dev_omegas[b]; P[b] = dev_pinvs[b];
        G[b] = G0 + (size_t)b * n * n; cI[b] = I0 + (size_t)b * n * n;
    }
    auto desc = [](int64_t M, int64_t N, int64_t K, int64_t a_m, int64_t a_k, int64_t b_k, int64_t b_n) {
        ttsk_gemm_desc d{};
        d.batch = 1; d.M = M; d.N = N; d.Ko = 1; d.Ki = K;
        d.a_m = a_m; d.a_ki = a_k; d.b_ki = b_k; d.b_n = b_n; d.c_m = N; d.c_n = 1;
        d.alpha = 1.0;
        return d;
    };
    int rc;
#define TTSK_PB(call) do { rc = (call); if (rc < 0) return rc; if (rc == 0) { set_error("ttsk_pinv_batch: product outside the small kernel"); return TTSK_ERR_UNSUPPORTED; } } while (0)
    if (l <= r) {
        TTSK_PB(small_try_batch(desc(l, l, r, r, 1, 1, r), count, Om, Om, G, stream, st));                  // G = Omega Omega^T
        if ((rc = launch_chol(G0, n, R0, I0, status, 1.0 / 300.0, st, nullptr, nullptr, count))) return rc;
        TTSK_PB(small_try_batch(desc(r, l, l, 1, r, l, 1), count, Om, cI, P, stream, st));                  // X = Omega^T G^-1
    } else {
        TTSK_PB(small_try_batch(desc(r, r, l, 1, r, r, 1), count, Om, Om, G, stream, st));                  // G = Omega^T Omega
        if ((rc = launch_chol(G0, n, R0, I0, status, 1.0 / 300.0, st, nullptr, nullptr, count))) return rc;
        TTSK_PB(small_try_batch(desc(r, l, r, r, 1, 1, r), count, cI, Om, P, stream, st));                  // X = G^-1 Omega^T
    }
#undef TTSK_PB
    // rejected ones: the Jacobi kernel on the untouched input (it leaves at once where the attempt was accepted)
    size_t jl = 0;
    const int jm = jacobi_lds_mode(mW, nW, &jl);
    if (jm < 0) return TTSK_ERR_HIP;
    auto kern = jm == 2 ? jacobi_pinv_kernel<2> : (jm == 1 ? jacobi_pinv_kernel<1> : jacobi_pinv_kernel<0>);
    const double rcond = pinv_rcond(l, r, -1.0);
    // equally spaced inputs and outputs, matrices that live in LDS (no shared global scratch): ONE launch, a workgroup per matrix
    // (five launches of a kernel that leaves at once were 23 us of a 0.26 ms to_tt at C3)
    bool spaced = jm == 2 && count >= 2;
    const int64_t os = count >= 2 ? Om[1] - Om[0] : 0, ps = count >= 2 ? P[1] - P[0] : 0;
    for (int b = 2; b < count && spaced; ++b) spaced = Om[b] - Om[b - 1] == os && P[b] - P[b - 1] == ps;
    if (spaced && os >= 0 && ps > 0) {
        hipLaunchKernelGGL(kern, dim3((unsigned)count), dim3(1024), jl, st, Om[0], l, r, transposed, ws, ws + mW * nW, rcond, P[0], (int *)nullptr,
                           (double *)nullptr, (double *)nullptr, (double *)nullptr, (const int *)status, os, ps);
        TTSK_LAUNCH_CHECK();
        return TTSK_OK;
    }
    for (int b = 0; b < count; ++b) {
        hipLaunchKernelGGL(kern, dim3(1), dim3(1024), jl, st, Om[b], l, r, transposed, ws, ws + mW * nW, rcond, P[b], (int *)nullptr,
                           (double *)nullptr, (double *)nullptr, (double *)nullptr, (const int *)(status + b), (int64_t)0, (int64_t)0);
        TTSK_LAUNCH_CHECK();
    }
    return TTSK_OK;
}

int ttsk_pinv(const double *dev_omega, int64_t l, int64_t r, double rcond, double *dev_pinv,
              int *host_rank, int stream)
{
    const int rc = ttsk_pinv_begin(dev_omega, l, r, rcond, dev_pinv, stream);
    if (rc != TTSK_OK) return rc;
    return ttsk_pinv_end(dev_omega, l, r, rcond, dev_pinv, host_rank, stream);
}

// One orthogonalisation step of orthogonal_sketch / hmt_sketch (sketch_dispatch.py:160-174) as ONE call without a
// read-back: Q = qr_thin(Psi_mat pinv(Omega)) (Omega == NULL: qr_thin(Psi_mat)) through the normal equations and
// CholeskyQR2.  The verdicts of the factorisations (Omega not of full rank / too ill conditioned, Psi_mat Omega^+ too
// ill conditioned) are NOT waited for: a rejection sets the stream's deferred flag and the numbers in Q are then
// meaningless; the caller reads the flag once at the end (ttsk_deferred_status) and repeats the sketch on the robust
// kernels (ttsk_pinv / ttsk_qr_thin).  TTSK_ERR_UNSUPPORTED: ranks beyond 256 or TTSK_FAST_SOLVES=0.
int ttsk_orth_step(const double *dev_psi, int64_t m, int64_t r2, const double *dev_omega, int64_t l, double *dev_q, int stream)
{
    TTSK_STREAM(st, stream);
    TTSK_ARG(dev_psi && dev_q && m >= 1 && r2 >= 1, "ttsk_orth_step: bad argument");
    const int64_t k = dev_omega ? l : r2;
    TTSK_ARG(k >= 1 && m >= k, "ttsk_orth_step: cannot orthogonalise a %lld x %lld unfolding", (long long)m, (long long)k);
    int *sticky = deferred_flag(stream);
    const int64_t nmin = dev_omega ? (l < r2 ? l : r2) : 0;
    if (!fast_solves() || !sticky || k > CHOL_MAX || nmin > CHOL_MAX || (dev_omega && pinv_rcond(l, r2, -1.0) > 1e-4)) {
        set_error("ttsk_orth_step: (%lld x %lld, rank %lld) is outside the fast path", (long long)m, (long long)r2, (long long)k);
        return TTSK_ERR_UNSUPPORTED;
    }
    const size_t pw = dev_omega ? pinv_ws_elems((int)nmin, l > r2 ? l : r2) + (size_t)r2 * l : 0;
    double *ws = (double *)scratch(stream, SCRATCH_MISC, (pw + qr_ws_elems(m, (int)k)) * 8);
    if (!ws) return TTSK_ERR_HIP;
    int rc;
    if (dev_omega) {
        double *pinv = ws + pinv_ws_elems((int)nmin, l > r2 ? l : r2);
        rc = pinv_cholesky_begin(dev_omega, l, r2, pinv, stream, st, ws, sticky);
        if (rc < 0) return rc;
        if (rc == 0) { set_error("ttsk_orth_step: pseudo-inverse outside the fast path"); return TTSK_ERR_UNSUPPORTED; }
        if ((rc = small_gemm(m, l, r2, dev_psi, r2, 1, pinv, l, 1, dev_q, stream))) return rc;      // M = Psi_mat Omega^+
    } else if (dev_q != dev_psi) {
        TTSK_HIP(hipMemcpyAsync(dev_q, dev_psi, (size_t)m * r2 * 8, hipMemcpyDeviceToDevice, st));
    }
    rc = qr_cholesky(dev_q, m, k, stream, st, ws + pw, sticky);
    if (rc < 0) return rc;
    if (rc == 0) { set_error("ttsk_orth_step: QR outside the fast path"); return TTSK_ERR_UNSUPPORTED; }
    return TTSK_OK;
}

// The pseudo-inverses of `count` matrices of ONE shape, fast path + Newton-Schulz step, every stage as ONE batched
// launch (7 launches for the d - 1 Omega of an orthogonal sketch instead of 7 each); verdicts deferred to `stream`'s flag.
int ttsk_pinv_batch_deferred(int count, const double *const *dev_omegas, int64_t l, int64_t r, double *const *dev_pinvs, int stream)
{
    TTSK_STREAM(st, stream);
    TTSK_ARG(count >= 1 && count <= SK_MAXB && dev_omegas && dev_pinvs && l >= 1 && r >= 1, "ttsk_pinv_batch_deferred: bad argument");
    int *sticky = deferred_flag(stream);
    const int n = (int)(l < r ? l : r);
    if (!fast_solves() || !sticky || n > CHOL_ONE || pinv_rcond(l, r, -1.0) > 1e-4) {
        set_error("ttsk_pinv_batch_deferred: (%lld x %lld) is outside the batched fast path", (long long)l, (long long)r);
        return TTSK_ERR_UNSUPPORTED;
    }
    const size_t per = (size_t)3 * n * n + (size_t)r * l;
    double *ws = (double *)scratch(stream, SCRATCH_MISC, ((size_t)count * per + 64) * 8);
    if (!ws) return TTSK_ERR_HIP;
    double *G0 = ws, *R0 = G0 + (size_t)count * n * n, *I0 = R0 + (size_t)count * n * n, *X0 = I0 + (size_t)count * n * n;
    int *status = (int *)(X0 + (size_t)count * r * l);
    const double *Om[SK_MAXB], *cG[SK_MAXB], *cI[SK_MAXB], *cX[SK_MAXB];
    double *G[SK_MAXB], *X[SK_MAXB], *P[SK_MAXB];
    for (int b = 0; b < count; ++b) {
        TTSK_ARG(dev_omegas[b] && dev_pinvs[b], "ttsk_pinv_batch_deferred: NULL matrix %d", b);
        Om[b] = dev_omegas[b]; P[b] = dev_pinvs[b];
        G[b] = G0 + (size_t)b * n * n; cG[b] = G[b];
        cI[b] = I0 + (size_t)b * n * n;
        X[b] = X0 + (size_t)b * r * l; cX[b] = X[b];
    }
    auto desc = [](int64_t M, int64_t N, int64_t K, int64_t a_m, int64_t a_k, int64_t b_k, int64_t b_n, double alpha) {
        ttsk_gemm_desc d{};
        d.batch = 1; d.M = M; d.N = N; d.Ko = 1; d.Ki = K;
        d.a_m = a_m; d.a_ki = a_k; d.b_ki = b_k; d.b_n = b_n; d.c_m = N; d.c_n = 1;
        d.alpha = alpha;
        return d;
    };
    int rc;
#define TTSK_PB(call) do { rc = (call); if (rc < 0) return rc; if (rc == 0) { set_error("ttsk_pinv_batch_deferred: product outside the small kernel"); return TTSK_ERR_UNSUPPORTED; } } while (0)
    if (l <= r) {
        TTSK_PB(small_try_batch(desc(l, l, r, r, 1, 1, r, 1.0), count, Om, Om, G, stream, st));                  // G = Omega Omega^T
        if ((rc = launch_chol(G0, n, R0, I0, status, 1.0 / 3.0e4, st, sticky, nullptr, count))) return rc;
        TTSK_PB(small_try_batch(desc(r, l, l, 1, r, l, 1, 1.0), count, Om, cI, X, stream, st));                  // X0 = Omega^T G^-1
        TTSK_PB(small_try_batch(desc(l, l, r, r, 1, l, 1, -1.0), count, Om, cX, G, stream, st));                 // E = -Omega X0
        hipLaunchKernelGGL(add_diag_kernel, dim3((unsigned)count), dim3(256), 0, st, G0, n, 2.0);                // E = 2 I - Omega X0
        TTSK_LAUNCH_CHECK();
        TTSK_PB(small_try_batch(desc(r, l, l, l, 1, l, 1, 1.0), count, cX, cG, P, stream, st));                  // X1 = X0 E
    } else {
        TTSK_PB(small_try_batch(desc(r, r, l, 1, r, r, 1, 1.0), count, Om, Om, G, stream, st));                  // G = Omega^T Omega
        if ((rc = launch_chol(G0, n, R0, I0, status, 1.0 / 3.0e4, st, sticky, nullptr, count))) return rc;
        TTSK_PB(small_try_batch(desc(r, l, r, r, 1, 1, r, 1.0), count, cI, Om, X, stream, st));                  // X0 = G^-1 Omega^T
        TTSK_PB(small_try_batch(desc(r, r, l, l, 1, r, 1, -1.0), count, cX, Om, G, stream, st));                 // E = -X0 Omega
        hipLaunchKernelGGL(add_diag_kernel, dim3((unsigned)count), dim3(256), 0, st, G0, n, 2.0);
        TTSK_LAUNCH_CHECK();
        TTSK_PB(small_try_batch(desc(r, l, r, r, 1, l, 1, 1.0), count, cG, cX, P, stream, st));                  // X1 = E X0
    }
#undef TTSK_PB
    return TTSK_OK;
}

// ttsk_orth_step with the pseudo-inverse already made (ttsk_pinv_batch_deferred): Q = qr_thin(Psi_mat P), P (r2, l)
int ttsk_orth_step_pinv(const double *dev_psi, int64_t m, int64_t r2, const double *dev_pinv, int64_t l, double *dev_q, int stream)
{
    TTSK_STREAM(st, stream);
    TTSK_ARG(dev_psi && dev_pinv && dev_q && m >= 1 && r2 >= 1 && l >= 1, "ttsk_orth_step_pinv: bad argument");
    TTSK_ARG(m >= l, "ttsk_orth_step_pinv: cannot orthogonalise a %lld x %lld unfolding", (long long)m, (long long)l);
    int *sticky = deferred_flag(stream);
    if (!fast_solves() || !sticky || l > CHOL_MAX) {
        set_error("ttsk_orth_step_pinv: rank %lld is outside the fast path", (long long)l);
        return TTSK_ERR_UNSUPPORTED;
    }
    double *ws = (double *)scratch(stream, SCRATCH_DRIVER, qr_ws_elems(m, (int)l) * 8);      // (the pinvs may live in SCRATCH_MISC)
    if (!ws) return TTSK_ERR_HIP;
    int rc;
    if ((rc = small_gemm(m, l, r2, dev_psi, r2, 1, dev_pinv, l, 1, dev_q, stream))) return rc;      // M = Psi_mat Omega^+
    rc = qr_cholesky(dev_q, m, l, stream, st, ws, sticky);
    if (rc < 0) return rc;
    if (rc == 0) { set_error("ttsk_orth_step_pinv: QR outside the fast path"); return TTSK_ERR_UNSUPPORTED; }
    return TTSK_OK;
}

// 1 in *host_flag if a fast-path factorisation queued on `stream` by ttsk_orth_step was rejected since the last call
// (waits for the stream; clears the flag)
int ttsk_deferred_status(int stream, int *host_flag)
{
    TTSK_STREAM(st, stream);
    TTSK_ARG(host_flag, "ttsk_deferred_status: NULL argument");
    int *sticky = deferred_flag(stream);
    TTSK_ARG(sticky, "ttsk_deferred_status: no device flag");
    // through a pinned word: a copy into the caller's pageable int is staged and blocks for ~40 us before the reset is
    // even queued
    int *pinned = (int *)persistent_alloc(PA_DEFERRED_PINNED, TTSK_NUM_STREAMS * sizeof(int), true, false);
    int *dst = pinned ? pinned + stream : host_flag;
    TTSK_HIP(hipMemcpyAsync(dst, sticky, sizeof(int), hipMemcpyDeviceToHost, st));
    TTSK_HIP(hipMemsetAsync(sticky, 0, sizeof(int), st));
    TTSK_HIP(hipStreamSynchronize(st));
    if (pinned) *host_flag = *dst;
    return TTSK_OK;
}

int ttsk_svd_small(const double *dev_A, int64_t m, int64_t n, double *dev_US, double *dev_S, double *dev_Vt,
                   int stream)
{
    TTSK_STREAM(st, stream);
    TTSK_ARG(dev_A && dev_US && dev_S && dev_Vt, "ttsk_svd_small: NULL argument");
    TTSK_ARG(m >= n && n >= 1 && n <= 8192 && m <= (1 << 20), "ttsk_svd_small: need m >= n, 1 <= n <= 8192, got (%lld, %lld)",
             (long long)m, (long long)n);
    // beyond one workgroup's reach (or from TTSK_SVD_GRID_FROM columns on: tests): the whole-chip kernel of svd_grid.hip
    static const int64_t grid_from = [] { const char *e = getenv("TTSK_SVD_GRID_FROM"); return e ? atoll(e) : 1025ll; }();
    if (n >= grid_from) return svd_jacobi_grid(dev_A, m, n, dev_US, dev_S, dev_Vt, stream, st);
    TTSK_ARG(n <= 1024, "ttsk_svd_small: the one-workgroup kernel takes n <= 1024");
    double *ws = (double *)scratch(stream, SCRATCH_MISC, (size_t)(m * n + n * n) * 8);
    if (!ws) return TTSK_ERR_HIP;
    size_t jl = 0;
    const int jm = jacobi_lds_mode(m, n, &jl);
    if (jm < 0) return TTSK_ERR_HIP;
    auto kern = jm == 2 ? jacobi_pinv_kernel<2> : (jm == 1 ? jacobi_pinv_kernel<1> : jacobi_pinv_kernel<0>);
    hipLaunchKernelGGL(kern, dim3(1), dim3(1024), jl, st, dev_A, m, n, 0, ws, ws + m * n, 0.0,
                       (double *)nullptr, (int *)nullptr, dev_US, dev_S, dev_Vt, (const int *)nullptr, (int64_t)0, (int64_t)0);
    TTSK_LAUNCH_CHECK();
    return TTSK_OK;
}

int ttsk_triu(double *A, int64_t m, int64_t n, int stream)
{
    TTSK_STREAM(st, stream);
    TTSK_ARG(A && m >= 1 && n >= 1, "ttsk_triu: bad argument");
    const int64_t blocks = cdiv(m * n, 256);
    hipLaunchKernelGGL(triu_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0, st, A, m, n);
    TTSK_LAUNCH_CHECK();
    return TTSK_OK;
}

int ttsk_qr_thin(double *A, int64_t m, int64_t n, int stream)
{
    TTSK_STREAM(st, stream);
    TTSK_ARG(A, "ttsk_qr_thin: NULL argument");
    TTSK_ARG(m >= n && n >= 1, "ttsk_qr_thin: need m >= n >= 1, got (%lld, %lld)", (long long)m,
             (long long)n);
    if (fast_solves()) {
        const int fr = qr_cholesky(A, m, n, stream, st);
        if (fr < 0) return fr;
        if (fr == 1) return TTSK_OK;
    }
    // scratch: tpart[n][nb] (tail-norm partials per pivot column), wpart[nb][n], Q[m*n]
    const int64_t nb = cdiv(m, QR_ROWS);
    const size_t small = (size_t)n * nb + (size_t)nb * n;
    double *ws = (double *)scratch(stream, SCRATCH_MISC, (small + (size_t)m * n) * 8);
    if (!ws) return TTSK_ERR_HIP;
    double *tpart = ws, *wpart = ws + (size_t)n * nb, *Q = ws + small;
    auto blocks_at = [&](int64_t j) { return (int)cdiv(m - j, QR_ROWS); };
    hipLaunchKernelGGL(qr_tail_norm_kernel, dim3(blocks_at(0)), dim3(256), 0, st, A, m, n, (int64_t)0, tpart);
    // factorisation: reflector j from column j applied to columns j+1..n-1; the update of
    // column j also leaves the tail-norm partials of column j+1 in tpart[j+1][*]
    for (int64_t j = 0; j + 1 < n; ++j) {
        const int nbj = blocks_at(j);
        hipLaunchKernelGGL(qr_w_kernel, dim3(nbj), dim3(256), 0, st, A, A, m, n, j, j + 1, tpart + j * nb,
                           blocks_at(j > 0 ? j - 1 : 0), wpart);
        hipLaunchKernelGGL(qr_update_kernel, dim3(nbj), dim3(256), 0, st, A, A, m, n, j, j + 1,
                           tpart + j * nb, blocks_at(j > 0 ? j - 1 : 0), wpart, nbj, tpart + (j + 1) * nb);
    }
    // Q = H_0 H_1 ... H_{n-1} [I; 0]
    hipLaunchKernelGGL(eye_kernel, dim3(1024), dim3(256), 0, st, Q, m, n);
    for (int64_t j = n - 1; j >= 0; --j) {
        const int nbj = blocks_at(j);
        const int nbt = blocks_at(j > 0 ? j - 1 : 0);
        hipLaunchKernelGGL(qr_w_kernel, dim3(nbj), dim3(256), 0, st, A, Q, m, n, j, j, tpart + j * nb, nbt, wpart);
        hipLaunchKernelGGL(qr_update_kernel, dim3(nbj), dim3(256), 0, st, A, Q, m, n, j, j, tpart + j * nb, nbt,
                           wpart, nbj, (double *)nullptr);
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(A, Q, (size_t)m * n * 8, hipMemcpyDeviceToDevice, st);
    TTSK_HIP(e);
    return TTSK_OK;
}

}  // extern "C"

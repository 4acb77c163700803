// Lab copy of the one-workgroup Jacobi round loop: where do the 2-4 us per round go?
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <vector>
#include <random>
template <int CTRL> __device__ __forceinline__ double jac_dpp(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row_sum16(double x)
{
    x += jac_dpp<0x120 + 8>(x); x += jac_dpp<0x120 + 4>(x); x += jac_dpp<0x120 + 2>(x); x += jac_dpp<0x120 + 1>(x);
    return x;
}
// DIAG: 0 full, 1 no math (fixed c, s), 2 no V update, 3 no stores at all, 4 fast math (rsqrt / rcp based)
template <int IT, int DIAG>
__device__ __forceinline__ void jac_pair(double *wp, double *wq, double *vp, double *vq, const int mW, const int nW,
                                         const int gl, const double tol2, int *s_rot)
{
    double x[IT], y[IT];
    double a = 0, b = 0, g = 0;
#pragma unroll
    for (int it = 0; it < IT; ++it) { const int i = gl + 16 * it; x[it] = i < mW ? wp[i] : 0.0; y[it] = i < mW ? wq[i] : 0.0; }
#pragma unroll
    for (int it = 0; it < IT; ++it) { a = fma(x[it], x[it], a); b = fma(y[it], y[it], b); g = fma(x[it], y[it], g); }
    a = row_sum16(a); b = row_sum16(b); g = row_sum16(g);
    if (g * g <= tol2 * (a * b) || g == 0.0) return;
    if (gl == 0) *s_rot = 1;
    double c, s;
    if (DIAG == 1) { c = 0.8; s = 0.6; }
    else if (DIAG == 4) {
        // tan(2 theta) = 2g / (b - a): t = sign / (|zeta| + sqrt(1 + zeta^2)) with hardware rcp / rsq + one Newton step
        const double d = b - a, h = 2.0 * g;
        const double r2 = d * d + h * h;                        // (2g)^2 (1 + zeta^2)
        double rs = __builtin_amdgcn_rsq(r2); rs = rs * (1.5 - 0.5 * r2 * rs * rs);   // 1 / sqrt(r2)
        const double r = r2 * rs;                               // sqrt(d^2 + h^2)
        // t = h / (d + sign(d) r)  (same root), c = 1 / sqrt(1 + t^2)
        const double den = d + (d >= 0 ? r : -r);
        double rd = __builtin_amdgcn_rcp(den); rd = rd * (2.0 - den * rd);
        const double t = h * rd;
        const double u = 1.0 + t * t;
        double ru = __builtin_amdgcn_rsq(u); ru = ru * (1.5 - 0.5 * u * ru * ru); ru = ru * (1.5 - 0.5 * u * ru * ru);
        c = ru; s = c * t;
    } else {
        const double zeta = (b - a) / (2.0 * g);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        c = 1.0 / sqrt(1.0 + t * t); s = c * t;
    }
    if (DIAG == 3) { if (c == 123.0) wp[0] = s; return; }
#pragma unroll
    for (int it = 0; it < IT; ++it) { const int i = gl + 16 * it; if (i < mW) { wp[i] = c * x[it] - s * y[it]; wq[i] = s * x[it] + c * y[it]; } }
    if (DIAG == 2) return;
#pragma unroll
    for (int it = 0; it < IT; ++it) { const int i = gl + 16 * it; x[it] = i < nW ? vp[i] : 0.0; y[it] = i < nW ? vq[i] : 0.0; }
#pragma unroll
    for (int it = 0; it < IT; ++it) { const int i = gl + 16 * it; if (i < nW) { vp[i] = c * x[it] - s * y[it]; vq[i] = s * x[it] + c * y[it]; } }
}
template <int DIAG>
__global__ __launch_bounds__(1024) void jac_kernel(const double *A, int mW, int nW, double *out, int *sweeps_out, int fixed_sweeps)
{
    extern __shared__ double lds[];
    double *Wc = lds, *V = lds + (size_t)mW * nW;
    const int tid = threadIdx.x;
    __shared__ int s_rot;
    const int grp = tid >> 4, gl = tid & 15, ngrp = blockDim.x >> 4;
    for (int t = tid; t < mW * nW; t += blockDim.x) { int j = t / mW, i = t - j * mW; Wc[t] = A[(size_t)i * nW + j]; }
    for (int t = tid; t < nW * nW; t += blockDim.x) V[t] = (t / nW == t % nW) ? 1.0 : 0.0;
    __syncthreads();
    const int np = nW + (nW & 1), nm1 = np - 1;
    const double tol = fmax(4.0, sqrt((double)mW)) * DBL_EPSILON, tol2 = tol * tol;
    int sweep = 0;
    for (; sweep < (fixed_sweeps ? fixed_sweeps : 60); ++sweep) {
        if (tid == 0) s_rot = 0;
        __syncthreads();
        for (int round = 0; round < np - 1; ++round) {
            for (int pi = grp; pi < np / 2; pi += ngrp) {
                int p, q;
                if (pi == 0) { p = nm1; q = round; }
                else { p = round + pi; p -= p >= nm1 ? nm1 : 0; q = round + nm1 - pi; q -= q >= nm1 ? nm1 : 0; }
                if (p >= nW || q >= nW) continue;
                if (p > q) { int t = p; p = q; q = t; }
                jac_pair<8, DIAG>(Wc + (size_t)p * mW, Wc + (size_t)q * mW, V + (size_t)p * nW, V + (size_t)q * nW, mW, nW, gl, tol2, &s_rot);
            }
            __syncthreads();
        }
        const int rot = s_rot;
        __syncthreads();
        if (!rot && !fixed_sweeps) { ++sweep; break; }
    }
    for (int t = tid; t < mW * nW; t += blockDim.x) out[t] = Wc[t];
    if (tid == 0) *sweeps_out = sweep;
}
template <int DIAG> static void run(const char *label, const double *dA, int m, int n, double *dout, int *dsw, int fixed)
{
    size_t lds = (size_t)(m * n + n * n) * 8;
    hipFuncSetAttribute((const void *)jac_kernel<DIAG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(jac_kernel<DIAG>, dim3(1), dim3(1024), lds, 0, dA, m, n, dout, dsw, fixed);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(jac_kernel<DIAG>, dim3(1), dim3(1024), lds, 0, dA, m, n, dout, dsw, fixed);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    int sw; hipMemcpy(&sw, dsw, 4, hipMemcpyDeviceToHost);
    int rounds = sw * (n + (n & 1) - 1);
    // orthogonality of the result columns
    std::vector<double> W((size_t)m * n); hipMemcpy(W.data(), dout, W.size() * 8, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int p = 0; p < n; p += 7) for (int q = p + 1; q < n; q += 5) {
        double a = 0, b = 0, g = 0;
        for (int i = 0; i < m; ++i) { a += W[(size_t)p * m + i] * W[(size_t)p * m + i]; b += W[(size_t)q * m + i] * W[(size_t)q * m + i]; g += W[(size_t)p * m + i] * W[(size_t)q * m + i]; }
        if (a > 1e-20 && b > 1e-20) worst = fmax(worst, fabs(g) / sqrt(a * b));
    }
    printf("%-28s %dx%d: %8.3f ms, %2d sweeps, %7.3f us per round, worst |cos| %.1e\n", label, m, n, ms, sw, ms * 1e3 / rounds, worst);
}

// Householder QR of Wc (col-major, mW x nW, stride mW) in place: afterwards the upper triangle holds R, the rest zeros.
__device__ void qr_inplace(double *Wc, int mW, int nW, int tid, int nthreads)
{
    const int grp = tid >> 4, gl = tid & 15, ngrp = nthreads >> 4;
    __shared__ double s_beta;
    for (int j = 0; j < nW; ++j) {
        const double *cj = Wc + (size_t)j * mW;
        // every group: the reflector of column j (tail norm), then its own columns
        double sig = 0.0;
        for (int i = j + 1 + gl; i < mW; i += 16) sig = fma(cj[i], cj[i], sig);
        sig = row_sum16(sig);
        const double alpha = cj[j];
        double beta = alpha, tau = 0.0, scale = 0.0;
        if (sig != 0.0) {
            const double nrm = sqrt(alpha * alpha + sig);
            beta = alpha >= 0 ? -nrm : nrm;
            tau = (beta - alpha) / beta;
            scale = 1.0 / (alpha - beta);
        }
        for (int k = j + 1 + grp; k < nW; k += ngrp) {
            double *ck = Wc + (size_t)k * mW;
            double w = gl == 0 ? ck[j] : 0.0;                       // v[0] = 1
            for (int i = j + 1 + gl; i < mW; i += 16) w = fma(cj[i] * scale, ck[i], w);
            w = row_sum16(w) * tau;
            if (gl == 0) ck[j] -= w;
            for (int i = j + 1 + gl; i < mW; i += 16) ck[i] = fma(-w, cj[i] * scale, ck[i]);
        }
        if (tid == 0) s_beta = beta;
        __syncthreads();
        // column j is final: R[j][j] = beta, zeros below
        for (int i = j + tid; i < mW; i += nthreads) Wc[(size_t)j * mW + i] = i == j ? s_beta : 0.0;
        __syncthreads();
    }
}
template <int DIAG>
__global__ __launch_bounds__(1024) void jacqr_kernel(const double *A, int mW, int nW, double *out, int *sweeps_out, int fixed_sweeps)
{
    extern __shared__ double lds[];
    double *Wc = lds, *V = lds + (size_t)mW * nW;
    const int tid = threadIdx.x;
    __shared__ int s_rot;
    const int grp = tid >> 4, gl = tid & 15, ngrp = blockDim.x >> 4;
    for (int t = tid; t < mW * nW; t += blockDim.x) { int j = t / mW, i = t - j * mW; Wc[t] = A[(size_t)i * nW + j]; }
    for (int t = tid; t < nW * nW; t += blockDim.x) V[t] = (t / nW == t % nW) ? 1.0 : 0.0;
    __syncthreads();
    qr_inplace(Wc, mW, nW, tid, blockDim.x);
    if (DIAG == 5) {        // sweep on R^T: swap the triangle
        for (int t = tid; t < nW * nW; t += blockDim.x) {
            const int j = t / nW, i = t - j * nW;
            if (i < j) { const double u = Wc[(size_t)j * mW + i]; Wc[(size_t)i * mW + j] = u; Wc[(size_t)j * mW + i] = 0.0; }
        }
        __syncthreads();
    }
    const int rows = nW;
    const int np = nW + (nW & 1), nm1 = np - 1;
    const double tol = fmax(4.0, sqrt((double)mW)) * DBL_EPSILON, tol2 = tol * tol;
    int sweep = 0;
    for (; sweep < (fixed_sweeps ? fixed_sweeps : 60); ++sweep) {
        if (tid == 0) s_rot = 0;
        __syncthreads();
        for (int round = 0; round < np - 1; ++round) {
            for (int pi = grp; pi < np / 2; pi += ngrp) {
                int p, q;
                if (pi == 0) { p = nm1; q = round; }
                else { p = round + pi; p -= p >= nm1 ? nm1 : 0; q = round + nm1 - pi; q -= q >= nm1 ? nm1 : 0; }
                if (p >= nW || q >= nW) continue;
                if (p > q) { int t = p; p = q; q = t; }
                if (DIAG == 5) { if (rows <= 64) jac_pair<4, 2>(Wc + (size_t)p * mW, Wc + (size_t)q * mW, V, V, rows, nW, gl, tol2, &s_rot); else jac_pair<8, 2>(Wc + (size_t)p * mW, Wc + (size_t)q * mW, V, V, rows, nW, gl, tol2, &s_rot); }
                else if (rows <= 64) jac_pair<4, DIAG>(Wc + (size_t)p * mW, Wc + (size_t)q * mW, V + (size_t)p * nW, V + (size_t)q * nW, rows, nW, gl, tol2, &s_rot);
                else jac_pair<8, DIAG>(Wc + (size_t)p * mW, Wc + (size_t)q * mW, V + (size_t)p * nW, V + (size_t)q * nW, rows, nW, gl, tol2, &s_rot);
            }
            __syncthreads();
        }
        const int rot = s_rot;
        __syncthreads();
        if (!rot && !fixed_sweeps) { ++sweep; break; }
    }
    if (DIAG == 5) {
        __shared__ double s_n2[128];
        for (int j = grp; j < nW; j += ngrp) {
            double a = 0; for (int i = gl; i < rows; i += 16) a = fma(Wc[(size_t)j * mW + i], Wc[(size_t)j * mW + i], a);
            a = row_sum16(a); if (gl == 0) s_n2[j] = a;
        }
        __syncthreads();
        for (int t = tid; t < nW * nW; t += blockDim.x) { const int j = t / nW, k = t - j * nW; V[t] = s_n2[j] > 0 ? Wc[(size_t)j * mW + k] / sqrt(s_n2[j]) : 0.0; }
        __syncthreads();
    }
    // W V from the original matrix
    for (int t = tid; t < mW * nW; t += blockDim.x) {
        const int j = t / mW, i = t - j * mW;
        double acc = 0.0;
        for (int k = 0; k < nW; ++k) acc = fma(A[(size_t)i * nW + k], V[(size_t)j * nW + k], acc);
        out[t] = acc;
    }
    if (tid == 0) *sweeps_out = sweep;
}
template <int DIAG> static void runqr(const char *label, const double *dA, int m, int n, double *dout, int *dsw, int fixed)
{
    size_t lds = (size_t)(m * n + n * n) * 8;
    hipFuncSetAttribute((const void *)jacqr_kernel<DIAG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(jacqr_kernel<DIAG>, dim3(1), dim3(1024), lds, 0, dA, m, n, dout, dsw, fixed);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(jacqr_kernel<DIAG>, dim3(1), dim3(1024), lds, 0, dA, m, n, dout, dsw, fixed);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    int sw; hipMemcpy(&sw, dsw, 4, hipMemcpyDeviceToHost);
    std::vector<double> W((size_t)m * n); hipMemcpy(W.data(), dout, W.size() * 8, hipMemcpyDeviceToHost);
    double worst = 0, smax = 0;
    std::vector<double> nr(n);
    for (int p = 0; p < n; ++p) { double a = 0; for (int i = 0; i < m; ++i) a += W[(size_t)p * m + i] * W[(size_t)p * m + i]; nr[p] = a; smax = fmax(smax, a); }
    for (int p = 0; p < n; ++p) for (int q = p + 1; q < n; ++q) {
        if (nr[p] < 1e-20 * smax || nr[q] < 1e-20 * smax) continue;
        double g = 0; for (int i = 0; i < m; ++i) g += W[(size_t)p * m + i] * W[(size_t)q * m + i];
        worst = fmax(worst, fabs(g) / sqrt(nr[p] * nr[q]));
    }
    printf("%-28s %dx%d: %8.3f ms, %2d sweeps, worst |cos| among live columns %.1e\n", label, m, n, ms, sw, worst);
}

int main()
{
    for (auto mn : {std::pair<int,int>{100, 50}, {100, 100}}) {
        int m = mn.first, n = mn.second;
        std::mt19937_64 rng(1); std::normal_distribution<double> nd;
        std::vector<double> A((size_t)m * n);
        for (auto &v : A) v = nd(rng);
        double *dA, *dout; int *dsw;
        hipMalloc(&dA, A.size() * 8); hipMalloc(&dout, A.size() * 8); hipMalloc(&dsw, 4);
        hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
        run<0>("full", dA, m, n, dout, dsw, 0);
        runqr<0>("QR + jacobi", dA, m, n, dout, dsw, 0);
        runqr<5>("QR + jacobi on R^T, no V", dA, m, n, dout, dsw, 0);
        {   // rank 15
            std::vector<double> L((size_t)m * 15), Rr((size_t)15 * n), B((size_t)m * n, 0.0);
            for (auto &v : L) v = nd(rng);
            for (auto &v : Rr) v = nd(rng);
            for (int i = 0; i < m; ++i) for (int k = 0; k < 15; ++k) for (int j = 0; j < n; ++j) B[(size_t)i * n + j] += L[(size_t)i * 15 + k] * Rr[(size_t)k * n + j];
            double *dB; hipMalloc(&dB, B.size() * 8); hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice);
            run<0>("rank 15: full", dB, m, n, dout, dsw, 0);
            runqr<0>("rank 15: QR + jacobi", dB, m, n, dout, dsw, 0);
            runqr<5>("rank 15: QR + jac R^T no V", dB, m, n, dout, dsw, 0);
            hipFree(dB);
        }
        run<4>("fast rcp / rsq math", dA, m, n, dout, dsw, 0);
        run<0>("full, 8 sweeps fixed", dA, m, n, dout, dsw, 8);
        run<1>("no math, 8 sweeps", dA, m, n, dout, dsw, 8);
        run<2>("no V update, 8 sweeps", dA, m, n, dout, dsw, 8);
        run<3>("no stores, 8 sweeps", dA, m, n, dout, dsw, 8);
        run<4>("fast math, 8 sweeps", dA, m, n, dout, dsw, 8);
    }
    return 0;
}

#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
#include <cmath>
template <int CTRL> __device__ __forceinline__ double quad_dpp(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__global__ __launch_bounds__(256) void chol_inv_kernel(const double *__restrict__ G, int n, double *__restrict__ Rinv,
                                                       double *__restrict__ Ginv, int *__restrict__ status,
                                                       double cond_tol, long long *stamps)
{
#define STAMP(i) do { __syncthreads(); if (threadIdx.x == 0) stamps[i] = __builtin_readcyclecounter(); } while (0)
    STAMP(0);
    extern __shared__ double sm[];
    const int ld = n + 1, tid = threadIdx.x;
    double *A = sm;
    double *xd = sm + n * ld;
    for (int e = tid; e < n * n; e += 256) A[(e / n) * ld + e % n] = G[e];
    __syncthreads();
    // Unscaled right-looking recurrence, ONE barrier per column: row j keeps r_j R[j][:] (r_j^2 = pivot) until
    // the end, the trailing update divides by the pivot instead; every thread reads the pivot itself.  (Pivot
    // square root by one thread + scaling of row j + update were three barriers and a serial stretch per
    // column: 75 of the kernel's 118 us at n = 50.)
    STAMP(1);
    int bad = 0;
    double pmin = 1e300, pmax = 0.0;                 // pivots r_j^2: the square root is not needed in the loop
    const int ti = tid >> 4, tc = tid & 15;
    // two columns per barrier: the pivot of column j + 1 and its updated row follow from rows j and j + 1 alone, every
    // thread forms them itself
    auto rcp2 = [](double x) { double r = __builtin_amdgcn_rcp(x); r = r * (2.0 - x * r); return r * (2.0 - x * r); };
    __shared__ double shadow[256];
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
        pmin = fmin(pmin, fmin(p0, p1)); pmax = fmax(pmax, fmax(p0, p1));
        // rows below: both columns at once
        for (int i = j + 2 + ti; i < n; i += 16) {
            const double a0 = r0[i], a1 = fma(-g, a0, r1[i]);  // A[j][i], updated A[j+1][i]
            const double f0 = a0 * pi0, f1 = a1 * pi1;
            for (int c = i + tc; c < n; c += 16) {
                const double u1 = fma(-g, r0[c], r1[c]);       // updated row j + 1 at c
                A[i * ld + c] = fma(-f1, u1, fma(-f0, r0[c], A[i * ld + c]));
            }
        }
        // row j + 1 after step j goes to a shadow row first (the others still read the old one), and home after the
        // barrier -- nobody reads row j + 1 again before the scaling pass
        double *sh = shadow + ((j >> 1) & 1) * 128;
        for (int c = j + 1 + tid; c < n; c += 256) sh[c] = fma(-g, r0[c], r1[c]);
        __syncthreads();
        for (int c = j + 1 + tid; c < n; c += 256) A[(j + 1) * ld + c] = sh[c];
    }
    __syncthreads();
    if (j < n) {
        double piv = A[j * ld + j];
        if (!(piv > 0.0)) { bad = 1; piv = 1.0; }
        pmin = fmin(pmin, piv); pmax = fmax(pmax, piv);
    }
    STAMP(2);
    if (tid == 0) status[0] = (bad || pmin < cond_tol * cond_tol * pmax) ? 1 : 0;
    // R[j][c] = row j / r_j; xd[j] = 1 / R[j][j] = 1 / r_j
    if (tid < n) xd[tid] = 1.0 / sqrt(A[tid * ld + tid] > 0.0 ? A[tid * ld + tid] : 1.0);
    __syncthreads();
    for (int jj = ti; jj < n; jj += 16) {
        const double sc = xd[jj];
        for (int c = jj + tc; c < n; c += 16) A[jj * ld + c] *= sc;
    }
    __syncthreads();
    // X = R^-1 (upper triangular).  X stays in LDS: its strict upper part X[i][c] (i < c) goes to the unused
    // strict lower triangle of A at A[c][i], its diagonal to xd[].
    // Row i of X from the rows below it, all columns c > i at once, one barrier per row:
    // X[i][c] = -(sum_{i < k <= c} R[i][k] X[k][c]) / R[i][i]; four lanes share a column's sum.  (One thread
    // per COLUMN doing its whole back substitution was 50 lanes of one wave walking 1200 dependent LDS round
    // trips: 0.1 ms at n = 50.)
    STAMP(3);
    // A column of X depends on R and on itself only, and a quad of lanes OWNS its columns (c = quad, quad + 64): the
    // whole back substitution of a column runs inside one wavefront, in order, without a single workgroup barrier
    // (a barrier per row, with the columns re-dealt every row, before: 65 -> 61 us at n = 50).
    // X = R^-1 in 16 x 16 blocks.  (1) the diagonal blocks, all at once: a quad of lanes owns a column and walks up
    // to 15 rows of its own block; (2) block rows from the bottom: X_ij = -X_ii (sum_{i<k<=j} R_ik X_kj) on the matrix
    // cores, the accumulator registers of the sum being the B operand of the second product; one barrier per block row.
    const int q4 = tid & 3, col4 = tid >> 2;
    for (int c = col4; c < n; c += 64) {
        const double *xc = A + c * ld;                                  // X[k][c] at A[c][k], k < c
        const int top = c & ~15;
        for (int i = c - 1; i >= top; --i) {
            const double *ri = A + i * ld;
            double acc = 0.0;
            for (int k = i + 1 + q4; k < c; k += 4) acc = fma(ri[k], xc[k], acc);
            acc += quad_dpp<0xB1>(acc);
            acc += quad_dpp<0x4E>(acc);
            if (q4 == 0) A[c * ld + i] = -(acc + ri[c] * xd[c]) * xd[i];
        }
    }
    __syncthreads();
    {
        typedef double v4d __attribute__((ext_vector_type(4)));
        const int lane = tid & 63, wv = tid >> 6, x16 = lane & 15, kq = lane >> 4;
        const int nt = (n + 15) >> 4;
        // X(r, c) for r <= c from its storage (strict upper part transposed into the lower triangle, diagonal in xd)
        auto Xat = [&](int r, int c) -> double {
            if (r >= n || c >= n || r > c) return 0.0;
            return r == c ? xd[c] : A[c * ld + r];
        };
        for (int bi = nt - 2; bi >= 0; --bi) {
            for (int bj = bi + 1 + wv; bj < nt; bj += 4) {
                v4d S = {0.0, 0.0, 0.0, 0.0};
                const int ra = 16 * bi + x16;
                for (int bk = bi + 1; bk <= bj; ++bk)
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb) {
                        const int k = 16 * bk + 4 * kb + kq;
                        const double av = (ra < n && k < n) ? A[ra * ld + k] : 0.0;          // R[ra][k], k > ra
                        const double bv = Xat(k, 16 * bj + x16);
                        S = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, S, 0, 0, 0);
                    }
                v4d Xn = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    const double av = Xat(16 * bi + x16, 16 * bi + 4 * kb + kq);              // X_ii[m][k]
                    Xn = __builtin_amdgcn_mfma_f64_16x16x4f64(av, S[kb], Xn, 0, 0, 0);
                }
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int r = 16 * bi + kq + 4 * jj, c = 16 * bj + x16;
                    if (r < n && c < n) A[c * ld + r] = -Xn[jj];
                }
            }
            __syncthreads();
        }
    }
    STAMP(4);
    STAMP(5);
    auto Xe = [&](int r, int c) -> double {          // X(r, c): strict upper part transposed in the lower triangle
        if (r >= n || c >= n || r > c) return 0.0;
        return r == c ? xd[c] : A[c * ld + r];
    };
    for (int i = ti; i < n; i += 16)
        for (int c = tc; c < n; c += 16) Rinv[i * n + c] = Xe(i, c);
    STAMP(6);
    if (Ginv) {
        // G^-1 = X X^T on the matrix cores: tile (ti, tc), tc >= ti, one per wave and turn; X is upper triangular,
        // so the sum over k starts at the column tile
        typedef double v4d __attribute__((ext_vector_type(4)));
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
                    const double av = Xe(ra, k);
                    const double bv = Xe(rb, k);
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
                }
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int i = 16 * ta + kq + 4 * jj, c = 16 * tb + x16;
                    if (i < n && c < n) { Ginv[i * n + c] = acc[jj]; Ginv[c * n + i] = acc[jj]; }
                }
            }
    }
    STAMP(7);
}

int main()
{
    for (int n : {50, 100}) {
        std::mt19937_64 rng(1); std::normal_distribution<double> nd;
        std::vector<double> B((size_t)n * 2 * n), G((size_t)n * n, 0.0);
        for (auto &v : B) v = nd(rng);
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double a = 0; for (int k = 0; k < 2 * n; ++k) a += B[(size_t)i * 2 * n + k] * B[(size_t)j * 2 * n + k]; G[(size_t)i * n + j] = a; }
        double *dG, *dR, *dGi; int *dst; long long *dstamp;
        hipMalloc(&dG, G.size() * 8); hipMalloc(&dR, G.size() * 8); hipMalloc(&dGi, G.size() * 8); hipMalloc(&dst, 4); hipMalloc(&dstamp, 64);
        hipMemcpy(dG, G.data(), G.size() * 8, hipMemcpyHostToDevice);
        hipFuncSetAttribute((const void *)chol_inv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(chol_inv_kernel, dim3(1), dim3(256), (size_t)(n * (n + 1) + n) * 8, 0, dG, n, dR, dGi, dst, 1e-6, dstamp);
        hipDeviceSynchronize();
        long long h[8]; hipMemcpy(h, dstamp, 64, hipMemcpyDeviceToHost);
        std::vector<double> Gi((size_t)n * n); hipMemcpy(Gi.data(), dGi, Gi.size() * 8, hipMemcpyDeviceToHost);
        double worst = 0; int st; hipMemcpy(&st, dst, 4, hipMemcpyDeviceToHost);
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double a = 0; for (int k = 0; k < n; ++k) a += G[(size_t)i * n + k] * Gi[(size_t)k * n + j]; worst = fmax(worst, fabs(a - (i == j))); }
        printf("status %d, max |G Ginv - I| = %.2e\n", st, worst);
        printf("n = %d: load %lld | factor %lld | scale %lld | backsub %lld | denseX %lld | storeR %lld | Ginv %lld | total %lld cycles (100 MHz counter?)\n", n,
               h[1] - h[0], h[2] - h[1], h[3] - h[2], h[4] - h[3], h[5] - h[4], h[6] - h[5], h[7] - h[6], h[7] - h[0]);
    }
    return 0;
}

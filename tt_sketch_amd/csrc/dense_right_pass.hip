// Rows of a dense unfolding against a DRM MATRIX, both contiguous along the contracted index:
//
//   C[m][n] = sum_k S[m][k] B[n][k]        S: rows x K (row stride s_row), B: N x K (row stride b_row), N <= 48, K ~ 10^5 .. 10^7
//
//   reference: dense_sketch.py:15-16 / :40-51 with the matrices of dense_gaussian_drm.py:77-80 on the right:
//   Psi_0 = X^{<1>} B_0^T (64 x 16.7 M against 40 x 16.7 M at C2: the tensor and 5.4 GB of Gaussian matrix, once each) and
//   Psi_{mu+1} = Z_mu^{(l n) x rest} B_{mu+1}^T.
//
// The long-K chain kernel (skinny.h) feeds its matrix instructions straight from memory, one 8- or 16-byte load per lane and
// fragment; with 64 + 40 rows that are 134 MB apart it ran at 2 TB/s.  Here a workgroup takes a block of 64 rows of S and a
// range of K, in stages of 64 columns: the S tile (64 x 64) and the B tile (48 x 64) come by global_load_lds -- 512 contiguous
// bytes per row, two rows per instruction -- into one of two LDS images, and are read from there as fragments.  The image has
// no padding (an LDS-DMA instruction writes 1 KB contiguously), so the 16-byte column pairs of row r are stored at position
// pair ^ (r & 31): the 16 rows a fragment read touches then hit 16 different bank groups.
// 8 waves = 4 row tiles x 2 column halves (tile 0 + strips 0, 2 | tile 1 + strips 1, 3 of the columns beyond 32).
// Work per stage: 10 k cycles of the matrix pipes for 57 KB of operands: HBM-bound (the operands once).  Partial results per
// (row block, K range) go to slabs, summed by skinny_r_reduce in range order (no atomics).
#include "common.h"
#include "skinny.h"

namespace ttsk {

namespace {

constexpr int RP_BROWS = 48;     // rows of the B tile (N <= 48; rows beyond N repeat the last one, their columns are not stored)

struct RightPass {
    const double *S, *B;
    double *slab;                // [row block][K range][64][N]
    int64_t rows, s_row, b_row, K;
    int N, nrb, nkc;
};

__device__ __forceinline__ double rp_mfma4(double a, double b, double c)
{
    return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}

__global__ __launch_bounds__(512, 2) void rows_longk_kernel(RightPass a)
{
    extern __shared__ double rp_img[];                                // [2][(64 + RP_BROWS) * 64]: 2 x 57 KB
    constexpr int IMG = (64 + RP_BROWS) * 64;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int x16 = lane & 15, kq = lane >> 4;
    const int rb = blockIdx.x / a.nkc, kc = blockIdx.x - rb * a.nkc;
    const int64_t nst = a.K >> 6;                                     // stages of 64 columns
    const int64_t s_beg = kc * nst / a.nkc, s_end = (kc + 1) * nst / a.nkc;
    const int N = a.N;

    // ---- the loader: instruction i (0 .. 55) brings rows 2 i, 2 i + 1 of the stacked (S block | B) tile; wave w issues
    // i = w, w + 8, ... (7 each).  Lane (h = lane >> 5, s = lane & 31) is slot s of row 2 i + h: column pair s ^ (row & 31).
    const char *src[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const int r = 2 * (w + 8 * j) + (lane >> 5);                  // row of the stacked tile
        const int pair = (lane & 31) ^ (r & 31);
        const double *p;
        if (r < 64) {
            int64_t m = (int64_t)rb * 64 + r;
            if (m >= a.rows) m = a.rows - 1;                          // (a short last block repeats its last row; not stored)
            p = a.S + m * a.s_row;
        } else {
            int n = r - 64;
            if (n >= N) n = N - 1;
            p = a.B + (int64_t)n * a.b_row;
        }
        src[j] = (const char *)(p + 2 * pair);
    }
    auto fill = [&](int64_t st, int buf) {
        const int64_t o = uniform_i64(st * 64 * 8);
#pragma unroll
        for (int j = 0; j < 7; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src[j] + o),
                                             (__attribute__((address_space(3))) void *)(rp_img + buf * IMG + (w + 8 * j) * 128), 16, 0, 0);
    };

    // ---- the wave's share: row tile rt, column tile ct and the strips ct, ct + 2 behind column 32
    const int rt = w & 3, ct = w >> 2;
    const int ns = N > 32 ? (N - 32 + 3) >> 2 : 0;                    // 4-wide strips behind the two full tiles
    const bool tile_on = 16 * ct < N;
    v4d acc = {0.0, 0.0, 0.0, 0.0};
    double accs[2] = {0.0, 0.0};
    // fragment addresses inside an image: element (row r, column k) sits at r * 64 + 2 * ((k >> 1) ^ (r & 31)) + (k & 1);
    // for k-block kb the lane's k = 4 kb + kq, i.e. pair 2 kb + (kq >> 1)
    const int ra = 16 * rt + x16, rbt = 64 + 16 * ct + x16;
    int rs[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) rs[q] = 64 + 32 + 4 * (ct + 2 * q) + (x16 & 3);

    if (s_beg < s_end) fill(s_beg, 0);
    for (int64_t st = s_beg; st < s_end; ++st) {
        const int buf = (int)((st - s_beg) & 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (st + 1 < s_end) fill(st + 1, buf ^ 1);
        const double *I = rp_img + buf * IMG;
#pragma unroll 4
        for (int kb = 0; kb < 16; ++kb) {
            const int pair = 2 * kb + (kq >> 1), lo = kq & 1;
            const double af = I[ra * 64 + 2 * (pair ^ (ra & 31)) + lo];
            if (tile_on) {
                const double bf = I[rbt * 64 + 2 * (pair ^ (rbt & 31)) + lo];
                acc = mfma16(af, bf, acc);
            }
#pragma unroll
            for (int q = 0; q < 2; ++q)
                if (ct + 2 * q < ns) {
                    const double bs = I[rs[q] * 64 + 2 * (pair ^ (rs[q] & 31)) + lo];
                    accs[q] = rp_mfma4(af, bs, accs[q]);
                }
        }
    }

    // ---- slab[rb][kc][m][n]: register t of the tile is row 4 t + kq, column x16; a strip: lane (i = kq, beta, j) row 4 beta + i
    double *slab = a.slab + ((int64_t)rb * a.nkc + kc) * 64 * N;
    if (tile_on) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int m = 16 * rt + 4 * t + kq, n = 16 * ct + x16;
            if (n < N) slab[m * N + n] = acc[t];
        }
    }
    const int beta = (lane >> 2) & 3, j4 = lane & 3;
#pragma unroll
    for (int q = 0; q < 2; ++q)
        if (ct + 2 * q < ns) {
            const int m = 16 * rt + 4 * beta + kq, n = 32 + 4 * (ct + 2 * q) + j4;
            if (n < N) slab[m * N + n] = accs[q];
        }
}

int rp_num_cu()
{
    static int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 256;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) return 256;
        return v;
    }();
    return n;
}

}  // namespace

// 1 = launched, 0 = shape not covered, < 0 = error.  C[m][n] (+)= alpha * sum_k S[m s_row + k] B[n b_row + k]
int rows_longk_try(const double *S, int64_t rows, int64_t s_row, const double *B, int N, int64_t b_row, int64_t K, double *C,
                   int64_t c_row, double alpha, int accumulate, int stream, hipStream_t st)
{
    static const int on = [] { const char *e = getenv("TTSK_ROWS_LONGK"); return e ? atoi(e) : 1; }();
    if (!on || rows < 1 || N < 1 || N > RP_BROWS || K < 4096 || (K & 63) || (s_row & 1) || (b_row & 1) || s_row < K || b_row < K) return 0;
    if (((uintptr_t)S | (uintptr_t)B) & 15) return 0;
    const int64_t nrb = (rows + 63) / 64;
    if (nrb > (1 << 20)) return 0;
    // K ranges: enough workgroups for two per CU, at least 8 stages each
    const int64_t nst = K >> 6;
    int64_t nkc = (2 * rp_num_cu() + nrb - 1) / nrb;
    if (nkc > nst / 8) nkc = nst / 8;
    if (nkc < 1) nkc = 1;
    if (nrb * nkc >= (1ll << 30)) return 0;
    RightPass a{};
    a.S = S; a.B = B; a.rows = rows; a.s_row = s_row; a.b_row = b_row; a.K = K; a.N = N;
    a.nrb = (int)nrb; a.nkc = (int)nkc;
    a.slab = (double *)scratch(stream, SCRATCH_GEMM, (size_t)nrb * nkc * 64 * N * 8 + 64);
    if (!a.slab) return TTSK_ERR_HIP;
    const bool prof = prof_on();
    if (prof) prof_open_named(st, -2, 2.0 * rows * (double)N * K, "rows_longk_kernel");
    static PerInit attr;
    if (attr.first() && hipFuncSetAttribute((const void *)rows_longk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
        set_error("rows_longk: cannot raise the dynamic LDS limit");
        return TTSK_ERR_HIP;
    }
    hipLaunchKernelGGL(rows_longk_kernel, dim3((unsigned)(nrb * nkc)), dim3(512), (size_t)2 * (64 + RP_BROWS) * 64 * 8, st, a);
    int rc = hipGetLastError() == hipSuccess ? TTSK_OK : TTSK_ERR_HIP;
    if (rc == TTSK_OK) {
        ReduceOut ro{};
        ro.C[0] = C;
        rc = launch_r_reduce(st, a.slab, (int)nkc, 64, N, (int)nrb, rows, ro, 1, c_row, 1, alpha, accumulate);
    }
    if (prof) prof_close(st);
    if (rc != TTSK_OK) { set_error("rows_longk: launch failed"); return TTSK_ERR_HIP; }
    return 1;
}

}  // namespace ttsk

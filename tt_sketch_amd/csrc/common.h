// Shared internals of libttsk (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include "ttsk.h"

namespace ttsk {

void set_error(const char *fmt, ...);
hipStream_t stream_of(int s);   // nullptr + error set if invalid / not initialised
int ensure_init();
// grow-only per-stream arenas; nullptr + error on failure.  Slots keep nested users apart:
enum { SCRATCH_DRIVER = 0, SCRATCH_GEMM = 1, SCRATCH_MISC = 2, SCRATCH_ORTH = 3, SCRATCH_SLOTS = 4 };
void *scratch(int stream, int slot, size_t bytes);
// small allocations that live from first use to ttsk_shutdown (status words, verdict slots), one per key: a re-init on
// another device gets fresh ones.  nullptr on failure.
enum { PA_PINV_HOST = 0, PA_PINV_DEV, PA_DEFERRED, PA_PINV_BATCH_VD, PA_DEFERRED_PINNED, PA_SLOTS };
void *persistent_alloc(int key, size_t bytes, bool host, bool zero);
// per-function one-time set-up (hipFuncSetAttribute) that has to be repeated after ttsk_shutdown + re-init
int init_generation();
struct PerInit {
    int gen = -1;
    bool first() { const int g = init_generation(); if (gen == g) return false; gen = g; return true; }
};

#define TTSK_HIP(call)                                                          \
    do {                                                                        \
        hipError_t e_ = (call);                                                 \
        if (e_ != hipSuccess) {                                                 \
            ttsk::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                            __FILE__, __LINE__);                                \
            return TTSK_ERR_HIP;                                                \
        }                                                                       \
    } while (0)

#define TTSK_ARG(cond, ...)                                                     \
    do {                                                                        \
        if (!(cond)) {                                                          \
            ttsk::set_error(__VA_ARGS__);                                       \
            return TTSK_ERR_ARG;                                                \
        }                                                                       \
    } while (0)

#define TTSK_STREAM(var, s)                                                     \
    hipStream_t var = ttsk::stream_of(s);                                       \
    if (!var) return TTSK_ERR_ARG

#define TTSK_LAUNCH_CHECK()                                                     \
    TTSK_HIP(hipGetLastError())

// Fragment reads of tiles p, p + 1 (256 bytes apart) must NOT be paired into ds_read2_b64: its 16-lane groups bank modulo
// 32 dwords, and this layout's interleaved k-pairs (lane stride 16 bytes) then collide two by two -- 16 LDS cycles per pair of
// fragments; two ds_read_b64 (32-lane halves, modulo 64 dwords) are conflict-free on it: 4 cycles (MI355X_MICROARCH.md, LDS
// table).  A volatile access is what the compiler does not combine; the reads keep their place in the instruction stream, which
// is where the look-ahead of the k-block loops wants them anyway.
#define LDS_UNPAIRED(x) (*(const volatile __attribute__((address_space(3))) double *)(&(x)))

typedef double v4d __attribute__((ext_vector_type(4)));

// v_mfma_f64_16x16x4_f64: A lane l holds A[m=l&15][k=l>>4], B lane l holds
// B[k=l>>4][n=l&15]; D reg j of lane l is D[row=(l>>4)+4j][col=l&15].
__device__ __forceinline__ v4d mfma16(double a, double b, v4d c)
{
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// per-launch device timing hooks (implemented in tt_fused.hip, used by ttsk_gemm)
bool prof_on();
void prof_open(hipStream_t st, double flops, int family, int tiles, bool ak, bool bk);
void prof_close(hipStream_t st);
enum { PROF_SAMPLER = 6, PROF_SPARSE = 7, PROF_SOLVE = 8 };
void prof_open_named(hipStream_t st, int cls, double work, const char *name);   // work in the class's own unit

// dense_right_pass.hip: C[m][n] (+)= alpha sum_k S[m][k] B[n][k], both rows contiguous along a long k, n <= 48: 1 = launched, 0 = not covered
int rows_longk_try(const double *S, int64_t rows, int64_t s_row, const double *B, int N, int64_t b_row, int64_t K, double *C,
                   int64_t c_row, double alpha, int accumulate, int stream, hipStream_t st);

// svd_grid.hip: one-sided Jacobi SVD over all compute units (n beyond the one-workgroup kernel)
int svd_jacobi_grid(const double *A, int64_t m, int64_t n, double *US, double *S, double *Vt, int stream, hipStream_t st);

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace ttsk

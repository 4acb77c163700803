// Pieces of linalg.hip the orthogonalising sketch driver (tt_orth.hip) builds on.
#pragma once
#include <cstddef>
#include <cstdint>
#include <hip/hip_runtime.h>

namespace ttsk {

bool fast_solves();                      // TTSK_FAST_SOLVES != 0
int *deferred_flag(int stream);          // the stream's sticky rejection word (ttsk_deferred_status reads and clears it)
size_t qr_ws_elems(int64_t m, int n);    // doubles of workspace qr_cholesky needs
// thin QR in place by CholeskyQR2 with LAPACK's column signs; 1 = queued, 0 = outside the fast path, < 0 = error.
// sticky: deferred verdict (no read-back; a rejection sets *sticky)
int qr_cholesky(double *A, int64_t m, int64_t n, int stream, hipStream_t st, double *ws_in = nullptr, int *sticky = nullptr,
                bool unsigned_q = false);
// unsigned_q: Q comes out with the signs of CholeskyQR (R's diagonal positive) and the sign reconstruction is
// left to the caller (qr_signs on the top n x n block of Q, beside the critical path; apply_signs at the end);
// return value 2 = the one-workgroup Householder kernel ran instead: Q carries LAPACK's signs already.
int qr_signs(const double *Qtop, int n, int square, const double *Sprev, int rows_per, double *Sout, hipStream_t st,
             double *work = nullptr);            // work: n * n doubles when n > 128
// pinv(Omega) (r x l) through the normal equations + one Newton-Schulz step, verdict deferred to *sticky; min(l, r) <= 256.
// 1 = queued, 0 = outside the fast path
size_t pinv_deferred_ws_elems(int64_t l, int64_t r);
int pinv_deferred(const double *omega, int64_t l, int64_t r, double *pinv, int stream, hipStream_t st, double *ws, int *sticky);
int apply_signs(int count, double *const *cores, const double *const *sp, const double *const *sn, const int *k0, const int *nn,
                const int *k1, hipStream_t st);
constexpr int QR_CHOL_MAX_N = 256;       // largest column count of qr_cholesky
// the same steps for `count` <= 16 matrices of one shape per launch; 1 = queued, 0 = outside this path (then nothing was written)
size_t qr_batch_ws_elems(int count, int64_t m, int n);
int qr_cholesky_batch(int count, double *const *A, int64_t m, int n, int stream, hipStream_t st, double *ws, int *sticky);
int qr_signs_batch(int count, const double *const *Qtop, int n, int square, const double *const *Sprev, int rows_per, double *const *Sout,
                   hipStream_t st);
}  // namespace ttsk

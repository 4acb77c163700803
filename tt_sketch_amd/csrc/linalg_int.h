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
int qr_cholesky(double *A, int64_t m, int64_t n, int stream, hipStream_t st, double *ws_in = nullptr, int *sticky = nullptr);
constexpr int QR_CHOL_MAX_N = 256;       // largest column count of qr_cholesky

}  // namespace ttsk

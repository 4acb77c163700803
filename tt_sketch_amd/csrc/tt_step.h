// Fused chain step of the TT x TT-DRM sketch (tt_step.hip); used by the ttsk_tt_sketch driver.
#pragma once
#include <cstdint>

namespace ttsk {

struct StepArgs {
    int s_in, s_out, rho, rhop, r;       // r = 0: no Psi phase
    int q_lo, q_cnt;                     // Psi uses T columns [q_lo, q_lo + q_cnt)
    int avec, cvec, dvec, rvec;          // 16-byte loads allowed for X_k / Cin / D_k / R (set by the launcher)
    int accumulate_psi;
    int64_t x_k, x_a, x_b, x_extent;     // X_k[a][b] = X[k*x_k + a*x_a + b*x_b]
    int64_t ldc, c_extent;               // Cin (s_in x rho), row stride ldc
    int64_t d_q, d_k, d_extent;          // D[q,k,q'] = D[q*d_q + k*d_k + q']
    int64_t ldr, r_extent;               // R (s_out x r), row stride ldr
    int64_t psi_q, psi_k, psi_c;
    const double *X, *Cin, *D, *R;
    double *partial, *Psi;
};

bool tt_step_fits(int64_t s_in, int64_t s_out, int64_t rho, int64_t rhop, int64_t r, int64_t q_cnt,
                  int64_t x_span_elems);
// Out (s_out x rhop, row stride ld_out) = sum over the n mode indices of T_k D_k; Psi optional
int tt_step_launch(bool xkf, int64_t n, StepArgs g, double *Out, int64_t ld_out, int stream);

}  // namespace ttsk

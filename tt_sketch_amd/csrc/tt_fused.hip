// Fused TT-input / TT-DRM step (placeholder until the LDS-resident kernel lands):
// reports TTSK_ERR_UNSUPPORTED so that the host composes the step from ttsk_gemm.
#include "common.h"
using namespace ttsk;
extern "C" int ttsk_tt_step(const ttsk_tt_step_desc *desc, const double *Lin, const double *X,
                            const double *D, const double *R, double *Lout, double *Psi, int stream)
{
    (void)desc; (void)Lin; (void)X; (void)D; (void)R; (void)Lout; (void)Psi; (void)stream;
    set_error("ttsk_tt_step: shape not supported by the fused kernel");
    return TTSK_ERR_UNSUPPORTED;
}

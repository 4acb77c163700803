// Explicit instantiations of chain_step_kernel: ranks of 5 full 16-wide tiles (+ 0..2 strips of 4), with and
// without the T side output; one translation unit per tile count keeps the parallel build short.
#include "chain_fused_inst.h"

namespace ttsk {

int launch_chain_step_c(const ChainStep &a, int nf, int str, bool wt, int ebuf, int unr, size_t lds, int grid, hipStream_t st)
{
    TTSK_CF_CASE(5, 0, 1)
    TTSK_CF_CASE(5, 1, 1)
    TTSK_CF_CASE(5, 2, 1)
    return 1;
}

}  // namespace ttsk

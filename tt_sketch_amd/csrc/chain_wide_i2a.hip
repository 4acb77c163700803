// Explicit instantiations of chain_wide_kernel for chunks of 2 full tiles + 0 strips, output structures of 0..4 full tiles.
#define TTSK_CW_NQF 2
#define TTSK_CW_STRQ 0
#include "chain_wide_inst.h"

namespace ttsk {

int launch_chain_wide_2a(const ChainWide &a, int nn, int sn, bool wt, int unr, size_t lds, int grid, hipStream_t st)
{
    TTSK_CW_OUT_LO
    return 1;
}

}  // namespace ttsk

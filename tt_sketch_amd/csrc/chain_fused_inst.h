// Launch helper shared by the chain_fused_i*.hip instantiation units.
#pragma once
#include "chain_fused.h"

namespace ttsk {

template <int NF, int STR, bool WT, int EBUF, int UNR>
static int launch_cf_one(const ChainStep &a, size_t lds, int grid, hipStream_t st)
{
    auto kern = chain_step_kernel<NF, STR, NF, STR, 5, WT, 1, EBUF, UNR>;
    static bool attr_done = false;
    if (!attr_done) {
        TTSK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), lds, st, a);
    TTSK_LAUNCH_CHECK();
    return TTSK_OK;
}

// one E image for the large structures (no LDS room for two), two for the small ones
#define TTSK_CF_CASE(NF, STR, EBUF)                                                                          \
    if (nf == NF && str == STR && ebuf == EBUF) {                                                            \
        if (unr == 25)                                                                                       \
            return wt ? launch_cf_one<NF, STR, true, EBUF, 25>(a, lds, grid, st)                             \
                      : launch_cf_one<NF, STR, false, EBUF, 25>(a, lds, grid, st);                           \
        return wt ? launch_cf_one<NF, STR, true, EBUF, 5>(a, lds, grid, st)                                  \
                  : launch_cf_one<NF, STR, false, EBUF, 5>(a, lds, grid, st);                                \
    }

}  // namespace ttsk

// Launch helpers shared by the chain_wide_i*.hip instantiation units (one unit per chunk structure).
#pragma once
#include "chain_wide.h"

namespace ttsk {

template <int NQF, int STRQ, int NNF, int STRN, bool WT, int UNR, bool MT2>
static int launch_cw_one(const ChainWide &a, size_t lds, int grid, hipStream_t st)
{
    auto kern = chain_wide_kernel<NQF, STRQ, NNF, STRN, WT, UNR, MT2>;
    static bool attr_done = false;
    if (!attr_done) {
        TTSK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), lds, st, a);
    TTSK_LAUNCH_CHECK();
    return TTSK_OK;
}

// waves with two row tiles exist only where the registers allow (chain_wide.hip asks cw_has_mt2 first)
constexpr bool cw_has_mt2(int nqf, int nnf, int strn) { return nqf <= 3 && nnf + (strn ? 1 : 0) <= 7; }

template <int NQF, int STRQ, int NNF, int STRN>
static int launch_cw_pick(const ChainWide &a, bool wt, int unr, size_t lds, int grid, hipStream_t st)
{
    constexpr bool MT2 = cw_has_mt2(NQF, NNF, STRN);
    if (unr == 25)
        return wt ? launch_cw_one<NQF, STRQ, NNF, STRN, true, 25, MT2>(a, lds, grid, st)
                  : launch_cw_one<NQF, STRQ, NNF, STRN, false, 25, MT2>(a, lds, grid, st);
    return wt ? launch_cw_one<NQF, STRQ, NNF, STRN, true, 5, MT2>(a, lds, grid, st)
              : launch_cw_one<NQF, STRQ, NNF, STRN, false, 5, MT2>(a, lds, grid, st);
}

#define TTSK_CW_OUT(NNF, STRN) \
    if (nn == NNF && sn == STRN) return launch_cw_pick<TTSK_CW_NQF, TTSK_CW_STRQ, NNF, STRN>(a, wt, unr, lds, grid, st);

// every output structure: up to 10 tiles (+ 0..2 strips of 4 columns behind the full tiles), in two halves so that
// each (chunk structure, half) is one translation unit of the parallel build
#define TTSK_CW_OUT_LO                                                                           \
        TTSK_CW_OUT(0, 1) TTSK_CW_OUT(0, 2)                                                      \
        TTSK_CW_OUT(1, 0) TTSK_CW_OUT(1, 1) TTSK_CW_OUT(1, 2)                                    \
        TTSK_CW_OUT(2, 0) TTSK_CW_OUT(2, 1) TTSK_CW_OUT(2, 2)                                    \
        TTSK_CW_OUT(3, 0) TTSK_CW_OUT(3, 1) TTSK_CW_OUT(3, 2)                                    \
        TTSK_CW_OUT(4, 0) TTSK_CW_OUT(4, 1) TTSK_CW_OUT(4, 2)
#define TTSK_CW_OUT_HI                                                                           \
        TTSK_CW_OUT(5, 0) TTSK_CW_OUT(5, 1) TTSK_CW_OUT(5, 2)                                    \
        TTSK_CW_OUT(6, 0) TTSK_CW_OUT(6, 1) TTSK_CW_OUT(6, 2)                                    \
        TTSK_CW_OUT(7, 0) TTSK_CW_OUT(7, 1) TTSK_CW_OUT(7, 2)                                    \
        TTSK_CW_OUT(8, 0) TTSK_CW_OUT(8, 1) TTSK_CW_OUT(8, 2)                                    \
        TTSK_CW_OUT(9, 0) TTSK_CW_OUT(9, 1) TTSK_CW_OUT(9, 2)                                    \
        TTSK_CW_OUT(10, 0)

}  // namespace ttsk

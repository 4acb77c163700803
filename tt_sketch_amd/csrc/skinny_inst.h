// Explicit instantiations of skinny_s_kernel for one ring depth (one translation unit per depth
// keeps the parallel build short).
#pragma once
#include "skinny.h"

namespace ttsk {

template <int NPT, int MODE, int D, int STR>
static int launch_s_one(const SkinnyS &a, size_t lds_bytes, int grid, hipStream_t st)
{
    auto kern = skinny_s_kernel<NPT, MODE, D, STR>;
    static bool attr_done = false;
    if (!attr_done) {
        TTSK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), lds_bytes, st, a);
    TTSK_LAUNCH_CHECK();
    return TTSK_OK;
}

template <int D, int STR>
int launch_skinny_s_depth(const SkinnyS &a, int npt, int spt, size_t lds_bytes, int grid, hipStream_t st)
{
#define TTSK_S_CASE(N) if (npt == N) return spt == 2 ? launch_s_one<N, 2, D, STR>(a, lds_bytes, grid, st) : (spt == 1 ? launch_s_one<N, 1, D, STR>(a, lds_bytes, grid, st) : launch_s_one<N, 0, D, STR>(a, lds_bytes, grid, st))
    TTSK_S_CASE(1); TTSK_S_CASE(2); TTSK_S_CASE(3); TTSK_S_CASE(4);
    TTSK_S_CASE(5); TTSK_S_CASE(6); TTSK_S_CASE(7); TTSK_S_CASE(8);
#undef TTSK_S_CASE
    set_error("skinny_s: no instantiation for %d/%d tiles", npt, spt);
    return TTSK_ERR_ARG;
}

}  // namespace ttsk

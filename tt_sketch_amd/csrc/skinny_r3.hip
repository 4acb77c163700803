// Explicit instantiations of skinny_r_kernel for 8 <= NMT <= 8 row tiles (NNT <= NMT).
#include "skinny.h"

namespace ttsk {

template <int NMT, int NNT>
static int launch_r_one(const SkinnyR &a, int grid, hipStream_t st)
{
    hipLaunchKernelGGL((skinny_r_kernel<NMT, NNT, TTSK_R_DEPTH>), dim3((unsigned)grid), dim3(512), 0, st, a);
    TTSK_LAUNCH_CHECK();
    return TTSK_OK;
}

template <int NMT>
static int launch_r_row(const SkinnyR &a, int nnt, int grid, hipStream_t st)
{
#define TTSK_R_CASE(N) if constexpr (N <= NMT) { if (nnt == N) return launch_r_one<NMT, N>(a, grid, st); }
    TTSK_R_CASE(1) TTSK_R_CASE(2) TTSK_R_CASE(3) TTSK_R_CASE(4) TTSK_R_CASE(5) TTSK_R_CASE(6) TTSK_R_CASE(7) TTSK_R_CASE(8)
#undef TTSK_R_CASE
    set_error("skinny_r: no instantiation for %d x %d tiles", NMT, nnt);
    return TTSK_ERR_ARG;
}

int launch_skinny_r_3(const SkinnyR &a, int nmt, int nnt, int grid, hipStream_t st)
{
    if (nmt == 8) return launch_r_row<8>(a, nnt, grid, st);
    set_error("skinny_r: bad tile count %d", nmt);
    return TTSK_ERR_ARG;
}

}  // namespace ttsk

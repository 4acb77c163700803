// Explicit instantiations of chain_step_kernel (one (tiles, strips) structure per case; the left
// chain's variant also writes T).
#include "chain_fused.h"

namespace ttsk {

template <int NF, int STR, bool WT, int EBUF, int D>
static int launch_cf_one(const ChainStep &a, size_t lds, int grid, hipStream_t st)
{
    auto kern = chain_step_kernel<NF, STR, NF, STR, D, WT, 1, EBUF>;
    static bool attr_done = false;
    if (!attr_done) {
        TTSK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), lds, st, a);
    TTSK_LAUNCH_CHECK();
    return TTSK_OK;
}

template <int NF, int STR, bool WT, int EBUF>
static int launch_cf_depth(const ChainStep &a, int depth, size_t lds, int grid, hipStream_t st)
{
    if (depth == 9) return launch_cf_one<NF, STR, WT, EBUF, 9>(a, lds, grid, st);
    if (depth == 13) return launch_cf_one<NF, STR, WT, EBUF, 13>(a, lds, grid, st);
    return launch_cf_one<NF, STR, WT, EBUF, 5>(a, lds, grid, st);
}

int launch_chain_step(const ChainStep &a, int nf, int str, bool wt, int ebuf, int depth, size_t lds, int grid, hipStream_t st)
{
    if (nf == 6 && str == 1 && !wt && ebuf == 1) return launch_cf_depth<6, 1, false, 1>(a, depth, lds, grid, st);
    if (nf == 3 && str == 1 && wt && ebuf == 2) return launch_cf_depth<3, 1, true, 2>(a, depth, lds, grid, st);
    if (nf == 3 && str == 1 && wt && ebuf == 1) return launch_cf_depth<3, 1, true, 1>(a, depth, lds, grid, st);
    if (nf == 3 && str == 1 && !wt && ebuf == 2) return launch_cf_depth<3, 1, false, 2>(a, depth, lds, grid, st);
    return 1;      // no instantiation for this structure: the caller falls back to the two-launch form
}

}  // namespace ttsk

// Explicit instantiations of chain_sum_kernel with 2 a-tiles per wave in the first product (J, K1 <= 20), T written or not.
#include "chain_sum.h"

namespace ttsk {

template __global__ void chain_sum_kernel<5, 5, 2, false>(ChainSum);
template __global__ void chain_sum_kernel<5, 5, 2, true>(ChainSum);

int launch_chain_sum_2(const ChainSum &a, bool wt, size_t lds, int grid, hipStream_t st)
{
    static PerInit attr;
    if (attr.first()) {
        if (hipFuncSetAttribute((const void *)chain_sum_kernel<5, 5, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void *)chain_sum_kernel<5, 5, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            set_error("chain_sum: cannot raise the dynamic LDS limit");
            return TTSK_ERR_HIP;
        }
    }
    if (wt) hipLaunchKernelGGL((chain_sum_kernel<5, 5, 2, true>), dim3(grid), dim3(512), lds, st, a);
    else hipLaunchKernelGGL((chain_sum_kernel<5, 5, 2, false>), dim3(grid), dim3(512), lds, st, a);
    return hipGetLastError() == hipSuccess ? TTSK_OK : TTSK_ERR_HIP;
}

}  // namespace ttsk

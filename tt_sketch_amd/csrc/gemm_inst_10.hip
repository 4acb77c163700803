// Instantiations of the contraction kernel for one pair of LDS staging layouts
// (A k-fast, B n-fast); split over four files so that hipcc builds them in parallel.
#include "gemm_kernel.h"
namespace ttsk {
template int launch_gemm_layout<true, false>(const GemmLaunch &, hipStream_t);
}

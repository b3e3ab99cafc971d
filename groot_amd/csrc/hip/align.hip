// align.hip -- K3, the graph walk (graphMinion loop + AlignRead hierarchy + DFS).  One of the four translation units of
// libgroot_hip.so (launch.hpp).
#include <hip/hip_runtime.h>

#include "kernels_align.hpp"
#include "kernels_lean.hpp"
#include "launch.hpp"

namespace groot {

void launch_align(uint32_t pw, const AlignArgs &a, dim3 grid, hipStream_t st)
{
    const size_t lds = a.lds_stride_dw ? (size_t)kBlock * a.lds_stride_dw * 4 + 16 : 0;
    if (pw == 3) {
        if (lds) hipLaunchKernelGGL((align_kernel<3, true>), grid, dim3(kBlock), lds, st, a);
        else hipLaunchKernelGGL((align_kernel<3, false>), grid, dim3(kBlock), 0, st, a);
    } else if (pw == 11) {
        if (lds) hipLaunchKernelGGL((align_kernel<11, true>), grid, dim3(kBlock), lds, st, a);
        else hipLaunchKernelGGL((align_kernel<11, false>), grid, dim3(kBlock), 0, st, a);
    }
}

void launch_align_lean(uint32_t pw, const LeanArgs &a, dim3 grid, hipStream_t st)
{
    if (pw != 3) return;
    const size_t lds = (size_t)kBlock * a.lds_stride_dw * 4;
    // (64-bit pieces per node comparison: reads of up to 128 / 160 / 256 bases)
    if (a.max_len <= 128) hipLaunchKernelGGL((align_lean_kernel<3, 4>), grid, dim3(kBlock), lds, st, a);
    else if (a.max_len <= 160) hipLaunchKernelGGL((align_lean_kernel<3, 5>), grid, dim3(kBlock), lds, st, a);
    else hipLaunchKernelGGL((align_lean_kernel<3, 8>), grid, dim3(kBlock), lds, st, a);
}

} // namespace groot

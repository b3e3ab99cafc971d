// seed_fast.hip -- the seed stage's fast paths: sketch_sig_kernel (27-bit signatures, text confirmation), text_lookup_kernel (reads the
// memo knows) and the list pass of sketch_seed_kernel behind either.  One of the four translation units of libgroot_hip.so (launch.hpp).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "kernels_sketch.hpp"
#include "kernels_sig.hpp"
#include "launch.hpp"

namespace groot {

// sketch_sig_kernel and the list pass of sketch_seed_kernel that follows it (or the text lookup): instances for the (sketch size, k) pairs that have a
// strength-reduced sketch_seed_kernel
template <int S, int M5> static void launch_list_sm(const SeedArgs &a, dim3 list_grid, hipStream_t st)
{
    hipLaunchKernelGGL((sketch_seed_kernel<S, 4, false, M5, true>), list_grid, dim3(kBlock), kLdsReads + (size_t)kBlock * a.list_stride_dw * 4, st, a);
}
template <int S, int M5> static void launch_sig_sm(const SeedArgs &a0, uint32_t max_len, hipStream_t st)
{
    // (reads of up to 128 bases: half the registers and instructions in the text comparison)
    // LDS: bad-chunk bits, then one dword of codes per 16 bases of the workgroup's span (+: the kernel reads whole register rows past a read)
    SeedArgs a = a0;
    const uint32_t bs = (uint32_t)kBlock;
    a.lds_read_bytes = (uint32_t)std::min<uint64_t>((uint64_t)bs * max_len + 32, 64 * 1024);
    const size_t lds = kSigCodes + (size_t)((a.lds_read_bytes + 15) / 16) * 4 + 96;
    const dim3 grid((a.n_reads + bs - 1) / bs);
    if (max_len <= 128) hipLaunchKernelGGL((sketch_sig_kernel<S, M5, 8>), grid, dim3(kBlock), lds, st, a);
    else hipLaunchKernelGGL((sketch_sig_kernel<S, M5, (int)kTextMax / 16>), grid, dim3(kBlock), lds, st, a);
}
void launch_list(uint32_t s, const SeedArgs &a, dim3 list_grid, hipStream_t st)
{
    const uint32_t m5 = (uint32_t)(((uint64_t)a.ix.k * GROOT_MULTI_SEED) & 31u);
    if (s == 21 && m5 == 6) return launch_list_sm<21, 6>(a, list_grid, st);
    if (s == 21 && m5 == 10) return launch_list_sm<21, 10>(a, list_grid, st);
    if (s == 21 && m5 == 14) return launch_list_sm<21, 14>(a, list_grid, st);
    if (s == 21 && m5 == 2) return launch_list_sm<21, 2>(a, list_grid, st);
    if (s == 20 && m5 == 6) return launch_list_sm<20, 6>(a, list_grid, st);
    if (s == 30 && m5 == 14) return launch_list_sm<30, 14>(a, list_grid, st);
}
bool sig_supported(uint32_t s, uint32_t max_k, uint32_t k)
{
    const uint32_t m5 = (uint32_t)(((uint64_t)k * GROOT_MULTI_SEED) & 31u);
    if (max_k != 4) return false;
    return (s == 21 && (m5 == 6 || m5 == 10 || m5 == 14 || m5 == 2)) || (s == 20 && m5 == 6) || (s == 30 && m5 == 14);
}
void launch_sig(uint32_t s, const SeedArgs &a, uint32_t max_len, hipStream_t st)
{
    const uint32_t m5 = (uint32_t)(((uint64_t)a.ix.k * GROOT_MULTI_SEED) & 31u);
    if (s == 21 && m5 == 6) return launch_sig_sm<21, 6>(a, max_len, st);
    if (s == 21 && m5 == 10) return launch_sig_sm<21, 10>(a, max_len, st);
    if (s == 21 && m5 == 14) return launch_sig_sm<21, 14>(a, max_len, st);
    if (s == 21 && m5 == 2) return launch_sig_sm<21, 2>(a, max_len, st);
    if (s == 20 && m5 == 6) return launch_sig_sm<20, 6>(a, max_len, st);
    if (s == 30 && m5 == 14) return launch_sig_sm<30, 14>(a, max_len, st);
}

void launch_text_lookup(uint32_t key_dwords, const SeedArgs &a, dim3 grid, size_t lds, hipStream_t st)
{
    switch (key_dwords) {
    case 7: hipLaunchKernelGGL((text_lookup_kernel<7>), grid, dim3(kBlock), lds, st, a); break;
    case 8: hipLaunchKernelGGL((text_lookup_kernel<8>), grid, dim3(kBlock), lds, st, a); break;
    default: hipLaunchKernelGGL((text_lookup_kernel<14>), grid, dim3(kBlock), lds, st, a); break;
    }
}

} // namespace groot

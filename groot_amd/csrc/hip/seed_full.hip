// seed_full.hip -- the instances of sketch_seed_kernel that take a whole batch (K1+K2 full width; `keep_sketches`, groot_hip_sketch,
// batches the signature kernel is not built for).  One of the four translation units of libgroot_hip.so (launch.hpp).
#include <hip/hip_runtime.h>

#include "kernels_sketch.hpp"
#include "launch.hpp"

namespace groot {

template <int S, int MAXK, int M5> static void launch_seed_sm(const SeedArgs &a, bool dump, dim3 grid, size_t lds, hipStream_t st)
{
    if (dump) hipLaunchKernelGGL((sketch_seed_kernel<S, MAXK, true, M5>), grid, dim3(kBlock), lds, st, a);
    else hipLaunchKernelGGL((sketch_seed_kernel<S, MAXK, false, M5>), grid, dim3(kBlock), lds, st, a);
}

bool seed_supported(uint32_t s, uint32_t max_k)
{
    // any `groot index -s / -y` (cmd/index.go:45-49): sizes without a compiled instance run the run-time-sized kernel
    return s >= 1 && s <= (uint32_t)kGenericMaxS && max_k >= 1 && max_k <= s;
}

void launch_seed(uint32_t s, uint32_t max_k, const SeedArgs &a, bool dump, dim3 grid, size_t lds, hipStream_t st)
{
    // low 5 bits of k * multiSeed: kernels specialised on it replace the per-slot 64-bit multiplies by adds
    const uint32_t m5 = (uint32_t)(((uint64_t)a.ix.k * GROOT_MULTI_SEED) & 31u);
    if (max_k == 4) {
        if (s == 21) {   // `groot index` default sketch size, for the common k-mer sizes
            if (m5 == 6) return launch_seed_sm<21, 4, 6>(a, dump, grid, lds, st);     // k = 31 (default), 63
            if (m5 == 10) return launch_seed_sm<21, 4, 10>(a, dump, grid, lds, st);   // k = 41
            if (m5 == 14) return launch_seed_sm<21, 4, 14>(a, dump, grid, lds, st);   // k = 51
            if (m5 == 2) return launch_seed_sm<21, 4, 2>(a, dump, grid, lds, st);     // k = 21
        }
        if (s == 20 && m5 == 6) return launch_seed_sm<20, 4, 6>(a, dump, grid, lds, st);    // travis e2e: -k 31 -s 20
        if (s == 30 && m5 == 14) return launch_seed_sm<30, 4, 14>(a, dump, grid, lds, st);  // pipeline tests: k = 51, s = 30
        switch (s) {
        case 8: return launch_seed_sm<8, 4, -1>(a, dump, grid, lds, st);
        case 10: return launch_seed_sm<10, 4, -1>(a, dump, grid, lds, st);
        case 12: return launch_seed_sm<12, 4, -1>(a, dump, grid, lds, st);
        case 16: return launch_seed_sm<16, 4, -1>(a, dump, grid, lds, st);
        case 20: return launch_seed_sm<20, 4, -1>(a, dump, grid, lds, st);
        case 21: return launch_seed_sm<21, 4, -1>(a, dump, grid, lds, st);
        case 24: return launch_seed_sm<24, 4, -1>(a, dump, grid, lds, st);
        case 28: return launch_seed_sm<28, 4, -1>(a, dump, grid, lds, st);
        case 30: return launch_seed_sm<30, 4, -1>(a, dump, grid, lds, st);
        case 32: return launch_seed_sm<32, 4, -1>(a, dump, grid, lds, st);
        case 36: return launch_seed_sm<36, 4, -1>(a, dump, grid, lds, st);
        case 40: return launch_seed_sm<40, 4, -1>(a, dump, grid, lds, st);
        case 42: return launch_seed_sm<42, 4, -1>(a, dump, grid, lds, st);
        case 48: return launch_seed_sm<48, 4, -1>(a, dump, grid, lds, st);
        case 50: return launch_seed_sm<50, 4, -1>(a, dump, grid, lds, st);
        case 56: return launch_seed_sm<56, 4, -1>(a, dump, grid, lds, st);
        case 64: return launch_seed_sm<64, 4, -1>(a, dump, grid, lds, st);
        default: break;
        }
    }
    launch_seed_sm<0, 0, -1>(a, dump, grid, lds, st);   // run-time sketch size / hash functions per band
}

} // namespace groot

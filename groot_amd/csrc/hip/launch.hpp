// launch.hpp -- kernel dispatch by (sketch size, maxK, k) and path words.  libgroot_hip.so is built from four translation units
// (groot_hip.hip: ctx + C ABI + the small kernels; seed_full.hip: the full-width hashing kernel; seed_fast.hip: signature kernel,
// text lookup, list pass; align.hip: the graph-walk kernels) so that they compile side by side; these are the calls between them.
#pragma once

#include "device_types.hpp"

namespace groot {

// K1+K2 full width (sketch_seed_kernel): any `groot index -s / -y` -- sizes without a compiled instance run the run-time-sized kernel
bool seed_supported(uint32_t s, uint32_t max_k);
void launch_seed(uint32_t s, uint32_t max_k, const SeedArgs &a, bool dump, dim3 grid, size_t lds, hipStream_t st);

// signature kernel; the list pass of the full-width kernel (behind it, or behind the text lookup); the text lookup
bool sig_supported(uint32_t s, uint32_t max_k, uint32_t k);
void launch_sig(uint32_t s, const SeedArgs &a, uint32_t max_len, hipStream_t st);
void launch_list(uint32_t s, const SeedArgs &a, dim3 list_grid, hipStream_t st);
void launch_text_lookup(uint32_t key_dwords, const SeedArgs &a, dim3 grid, size_t lds, hipStream_t st);

// K3
void launch_align(uint32_t pw, const AlignArgs &a, dim3 grid, hipStream_t st);
void launch_align_lean(uint32_t pw, const LeanArgs &a, dim3 grid, hipStream_t st);   // first pass (pw == 3 only)

} // namespace groot

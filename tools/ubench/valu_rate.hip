// valu_rate.hip -- issue-rate microbenchmark for the VALU instructions the sketch kernel is built from (gfx950).
// Each variant runs ITER x 16 independent instructions per thread on distinct registers; rate = wave-instructions per
// cycle per SIMD at the nominal 2.4 GHz (report also cycles per wave-instruction).
//   hipcc --offload-arch=gfx950 -O3 -o build/valu_rate tools/ubench/valu_rate.hip && build/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define ITER 4096

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int V> __global__ __launch_bounds__(256) void k(uint64_t *out, uint32_t seed)
{
    uint64_t a[16];
    uint32_t b[16], c[16];
    for (int i = 0; i < 16; i++) { a[i] = (uint64_t)threadIdx.x * 0x9E3779B97F4A7C15ULL + i + seed; b[i] = (uint32_t)a[i] ^ 0x55u; c[i] = ~b[i]; }
    const uint64_t h = a[3] | 1;
    const uint32_t hs = (uint32_t)h;
    for (int it = 0; it < ITER; it++) {
        if (V == 0) {        // v_lshl_add_u64 a, a, 0, h
#define X(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a[i]) : "v"(h));
            REP16(X)
#undef X
        } else if (V == 1) { // 32-bit add pair with carry
#define X(i) { uint32_t lo = (uint32_t)a[i], hi = (uint32_t)(a[i] >> 32); asm volatile("v_add_co_u32 %0, vcc, %0, %2\n v_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(lo), "+v"(hi) : "v"(hs), "v"(b[0]) : "vcc"); a[i] = lo | ((uint64_t)hi << 32); }
            REP16(X)
#undef X
        } else if (V == 2) { // v_lshrrev_b64
#define X(i) asm volatile("v_lshrrev_b64 %0, 27, %0" : "+v"(a[i]));
            REP16(X)
#undef X
        } else if (V == 3) { // v_xor_b32
#define X(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(b[i]) : "v"(hs));
            REP16(X)
#undef X
        } else if (V == 4) { // v_cmp_lt_u64 + 2 cndmask
#define X(i) { uint32_t lo = (uint32_t)a[i], hi = (uint32_t)(a[i] >> 32); asm volatile("v_cmp_lt_u64 vcc, %2, %3\n v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %5, vcc" : "+v"(lo), "+v"(hi) : "v"(h), "v"(a[i]), "v"(b[i]), "v"(c[i]) : "vcc"); a[i] = lo | ((uint64_t)hi << 32); }
            REP16(X)
#undef X
        } else if (V == 5) { // v_min_u32
#define X(i) asm volatile("v_min_u32 %0, %0, %1" : "+v"(b[i]) : "v"(c[i]));
            REP16(X)
#undef X
        } else if (V == 6) { // v_med3_u32
#define X(i) asm volatile("v_med3_u32 %0, %0, %1, %2" : "+v"(b[i]) : "v"(c[i]), "v"(hs));
            REP16(X)
#undef X
        } else if (V == 7) { // v_bfi_b32
#define X(i) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(b[i]) : "v"(hs), "v"(c[i]));
            REP16(X)
#undef X
        } else if (V == 8) { // v_and_or_b32
#define X(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(b[i]) : "v"(hs), "v"(c[i]));
            REP16(X)
#undef X
        } else if (V == 9) { // v_mul_lo_u32
#define X(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(b[i]) : "v"(hs));
            REP16(X)
#undef X
        } else if (V == 10) { // v_mul_hi_u32
#define X(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(b[i]) : "v"(hs));
            REP16(X)
#undef X
        } else if (V == 11) { // v_mad_u64_u32
#define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(hs), "v"(b[i]) : "vcc");
            REP16(X)
#undef X
        } else if (V == 12) { // v_min3_u32
#define X(i) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(b[i]) : "v"(c[i]), "v"(hs));
            REP16(X)
#undef X
        } else if (V == 13) { // v_cmp_lt_u32 -> sgpr pair (VOP3), no consumer
#define X(i) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(b[i]), "v"(c[i]) : "vcc");
            REP16(X)
#undef X
        } else if (V == 14) { // v_alignbit_b32
#define X(i) asm volatile("v_alignbit_b32 %0, %0, %1, 27" : "+v"(b[i]) : "v"(c[i]));
            REP16(X)
#undef X
        } else if (V == 15) { // v_add_u32
#define X(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(b[i]) : "v"(hs));
            REP16(X)
#undef X
        } else if (V == 16) { // v_lshrrev_b32 + v_xor (two ops)
#define X(i) asm volatile("v_lshrrev_b32 %1, 27, %0\n v_xor_b32 %0, %0, %1" : "+v"(b[i]), "+v"(c[i]));
            REP16(X)
#undef X
        } else if (V == 17) { // v_cmp_lt_u64 alone
#define X(i) asm volatile("v_cmp_lt_u64 vcc, %0, %1" : : "v"(a[i]), "v"(h) : "vcc");
            REP16(X)
#undef X
        } else if (V == 18) { // v_cndmask_b32 alone (vcc fixed)
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(b[i]) : "v"(c[i]) : );
            REP16(X)
#undef X
        } else if (V == 19) { // v_xad_u32 (xor-add)
#define X(i) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(b[i]) : "v"(hs), "v"(c[i]));
            REP16(X)
#undef X
        } else if (V == 20) { // v_mov_b32 dpp row_shr:1 (cross-lane)
#define X(i) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(b[i]) : "v"(c[i]));
            REP16(X)
#undef X
        } else if (V == 21) { // v_pk_add_u16 as a stand-in for packed int
#define X(i) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(b[i]) : "v"(hs));
            REP16(X)
#undef X
        } else if (V == 22) { // v_max_u32 + v_min_u32 pair (sorting network step)
#define X(i) asm volatile("v_max_u32 %1, %0, %2\n v_min_u32 %0, %0, %2" : "+v"(b[i]), "+v"(c[i]) : "v"(hs));
            REP16(X)
#undef X
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < 16; i++) s += a[i] + b[i] + c[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

struct V { const char *name; int ops; void (*fn)(uint64_t *, uint32_t); };

int main()
{
    int ncu = 256;
    hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    const int blocks = ncu * 8;          // 8 waves per SIMD
    uint64_t *out;
    hipMalloc(&out, (size_t)blocks * 256 * 8);
    V vs[] = {
        {"v_lshl_add_u64", 1, k<0>}, {"v_add_co+v_addc_co (2 ops)", 2, k<1>}, {"v_lshrrev_b64", 1, k<2>}, {"v_xor_b32", 1, k<3>},
        {"v_cmp_lt_u64+2 cndmask (3 ops)", 3, k<4>}, {"v_min_u32", 1, k<5>}, {"v_med3_u32", 1, k<6>}, {"v_bfi_b32", 1, k<7>},
        {"v_and_or_b32", 1, k<8>}, {"v_mul_lo_u32", 1, k<9>}, {"v_mul_hi_u32", 1, k<10>}, {"v_mad_u64_u32", 1, k<11>},
        {"v_min3_u32", 1, k<12>}, {"v_cmp_lt_u32", 1, k<13>}, {"v_alignbit_b32", 1, k<14>}, {"v_add_u32", 1, k<15>},
        {"v_lshrrev_b32+v_xor (2 ops)", 2, k<16>}, {"v_cmp_lt_u64", 1, k<17>}, {"v_cndmask_b32", 1, k<18>}, {"v_xad_u32", 1, k<19>},
        {"v_mov_b32_dpp row_shr", 1, k<20>}, {"v_pk_add_u16", 1, k<21>}, {"v_max_u32+v_min_u32 (2 ops)", 2, k<22>},
    };
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%-36s %10s %14s %12s\n", "instruction", "ms", "cyc/wave-inst", "(@2.4GHz)");
    for (auto &v : vs) {
        hipLaunchKernelGGL(v.fn, dim3(blocks), dim3(256), 0, 0, out, 1u);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(v.fn, dim3(blocks), dim3(256), 0, 0, out, 2u);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double wave_insts = (double)blocks * 4 * ITER * 16 * v.ops;       // 4 waves per block
        const double per_simd = wave_insts / (ncu * 4.0);
        const double cyc = ms * 1e-3 * 2.4e9 / per_simd;
        printf("%-36s %10.3f %14.2f\n", v.name, ms, cyc);
    }
    return 0;
}

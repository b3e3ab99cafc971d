// slot_rate.hip -- the two per-(k-mer, slot) update sequences of the sketch kernels in isolation (gfx950, 5 waves/SIMD):
//   legacy : t = acc ^ (acc >> 27); m = min64(m, t); acc += h                 (7 instructions)
//   packed : key = bfi(top 25 bits of acc_hi, j); sc = med3(pk, sc, key); pk = min(pk, key); acc += h   (4 instructions)
//   hipcc --offload-arch=gfx950 -O3 -o build/slot_rate tools/ubench/slot_rate.hip && build/slot_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 2048
template <int V> __global__ __launch_bounds__(256, 5) void k(uint64_t *out, uint32_t seed)
{
    uint64_t m[20];
    uint32_t pk[20], sc[20];
    for (int i = 0; i < 20; i++) { m[i] = ~0ull; pk[i] = ~0u; sc[i] = ~0u; }
    uint64_t h = (uint64_t)threadIdx.x * 0x9E3779B97F4A7C15ULL + seed;
    for (uint32_t j = 0; j < ITER; j++) {
        h = h * 0x2545F4914F6CDD1DULL + j;
        uint64_t acc = h * 0x90b45d39fb6da1e0ULL;
        uint32_t jv = j & 127;
        asm("" : "+v"(jv));
#pragma unroll
        for (int i = 0; i < 20; i++) {
            if (V == 0) {
                const uint64_t t = acc ^ (acc >> 27);
                m[i] = t < m[i] ? t : m[i];
            } else {
                uint32_t key;
                asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(key) : "s"(0xFFFFFF80u), "v"((uint32_t)(acc >> 32)), "v"(jv));
                asm("v_med3_u32 %0, %1, %0, %2" : "+v"(sc[i]) : "v"(pk[i]), "v"(key));
                pk[i] = min(pk[i], key);
            }
            acc += h;
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < 20; i++) s += m[i] + pk[i] + sc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main()
{
    int ncu = 256;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    const int blocks = ncu * 5 * 8;
    uint64_t *out;
    (void)hipMalloc(&out, (size_t)blocks * 256 * 8);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    void (*fns[2])(uint64_t *, uint32_t) = {k<0>, k<1>};
    const char *names[2] = {"legacy: shift, 2 xor, 64-bit min, add", "packed: bfi, med3, min, add"};
    for (int v = 0; v < 2; v++) {
        hipLaunchKernelGGL(fns[v], dim3(blocks), dim3(256), 0, 0, out, 1u);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(fns[v], dim3(blocks), dim3(256), 0, 0, out, 2u);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double slots = (double)blocks * 4 * ITER * 20;
        printf("%-44s %8.3f ms  %6.2f cycles per (wave, slot) at 2.4 GHz\n", names[v], ms, ms * 1e-3 * 2.4e9 / (slots / (ncu * 4.0)));
    }
    return 0;
}

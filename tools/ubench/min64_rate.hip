// min64_rate.hip -- ways to keep a running 64-bit unsigned minimum on gfx950 (no v_min_u64): cycles per update at 8 waves/SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o build/min64_rate tools/ubench/min64_rate.hip && build/min64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITER 4096
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int V> __global__ __launch_bounds__(256) void k(uint64_t *out, uint32_t seed)
{
    uint64_t m[16], r[16];
    for (int i = 0; i < 16; i++) { m[i] = ~0ull - i; r[i] = (uint64_t)threadIdx.x * 0x9E3779B97F4A7C15ULL + i * 77 + seed; }
    for (int it = 0; it < ITER; it++) {
        // a new candidate per slot and iteration (cheap xorshift-ish so that the compiler cannot fold anything)
        if (V == 0) {          // compiler's choice: r < m ? r : m
#define X(i) { r[i] += 0x9E3779B97F4A7C15ULL; m[i] = r[i] < m[i] ? r[i] : m[i]; }
            REP16(X)
#undef X
        } else if (V == 1) {   // v_cmp_lt_u64 vcc + 2 v_cndmask
#define X(i) { r[i] += 0x9E3779B97F4A7C15ULL; uint32_t ml = (uint32_t)m[i], mh = (uint32_t)(m[i] >> 32); \
               asm volatile("v_cmp_lt_u64 vcc, %2, %3\n v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %5, vcc" \
                            : "+v"(ml), "+v"(mh) : "v"(r[i]), "v"(m[i]), "v"((uint32_t)r[i]), "v"((uint32_t)(r[i] >> 32)) : "vcc"); \
               m[i] = ml | ((uint64_t)mh << 32); }
            REP16(X)
#undef X
        } else if (V == 2) {   // v_cmpx_lt_u64 + v_mov_b64 under the narrowed exec + restore
#define X(i) { r[i] += 0x9E3779B97F4A7C15ULL; \
               asm volatile("s_mov_b64 s[20:21], exec\n v_cmpx_lt_u64 %1, %0\n v_mov_b64 %0, %1\n s_mov_b64 exec, s[20:21]" \
                            : "+v"(m[i]) : "v"(r[i]) : "s20", "s21"); }
            REP16(X)
#undef X
        } else if (V == 3) {   // same with two v_mov_b32
#define X(i) { r[i] += 0x9E3779B97F4A7C15ULL; uint32_t ml = (uint32_t)m[i], mh = (uint32_t)(m[i] >> 32); \
               asm volatile("s_mov_b64 s[20:21], exec\n v_cmpx_lt_u64 %2, %3\n v_mov_b32 %0, %4\n v_mov_b32 %1, %5\n s_mov_b64 exec, s[20:21]" \
                            : "+v"(ml), "+v"(mh) : "v"(r[i]), "v"(m[i]), "v"((uint32_t)r[i]), "v"((uint32_t)(r[i] >> 32)) : "s20", "s21"); \
               m[i] = ml | ((uint64_t)mh << 32); }
            REP16(X)
#undef X
        } else if (V == 4) {   // 16 slots share one exec save/restore: save, then 16 x (cmpx, mov, restore)
            asm volatile("s_mov_b64 s[20:21], exec" ::: "s20", "s21");
#define X(i) { r[i] += 0x9E3779B97F4A7C15ULL; \
               asm volatile("v_cmpx_lt_u64 %1, %0\n v_mov_b64 %0, %1\n s_mov_b64 exec, s[20:21]" : "+v"(m[i]) : "v"(r[i])); }
            REP16(X)
#undef X
        } else if (V == 5) {   // the add alone (baseline to subtract)
#define X(i) { r[i] += 0x9E3779B97F4A7C15ULL; asm volatile("" : "+v"(r[i])); }
            REP16(X)
#undef X
        } else if (V == 6) {   // v_cmp_lt_u64 into an SGPR pair + 2 v_cndmask reading that pair
#define X(i) { r[i] += 0x9E3779B97F4A7C15ULL; uint32_t ml = (uint32_t)m[i], mh = (uint32_t)(m[i] >> 32); \
               asm volatile("v_cmp_lt_u64 s[20:21], %2, %3\n v_cndmask_b32 %0, %0, %4, s[20:21]\n v_cndmask_b32 %1, %1, %5, s[20:21]" \
                            : "+v"(ml), "+v"(mh) : "v"(r[i]), "v"(m[i]), "v"((uint32_t)r[i]), "v"((uint32_t)(r[i] >> 32)) : "s20", "s21"); \
               m[i] = ml | ((uint64_t)mh << 32); }
            REP16(X)
#undef X
        } else if (V == 7) {   // v_min_f64 on the raw bits (NOT a correct u64 min: rate only)
#define X(i) { r[i] += 0x9E3779B97F4A7C15ULL; asm volatile("v_min_f64 %0, %0, %1" : "+v"(m[i]) : "v"(r[i])); }
            REP16(X)
#undef X
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < 16; i++) s += m[i] ^ r[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main()
{
    int ncu = 256;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    const int blocks = ncu * 8;
    uint64_t *out;
    (void)hipMalloc(&out, (size_t)blocks * 256 * 8);
    struct { const char *name; void (*fn)(uint64_t *, uint32_t); } vs[] = {
        {"add + (r < m ? r : m)   [compiler]", k<0>}, {"add + v_cmp_lt_u64 vcc + 2 v_cndmask", k<1>}, {"add + save exec, v_cmpx, v_mov_b64, restore", k<2>},
        {"add + save exec, v_cmpx, 2 v_mov_b32, restore", k<3>}, {"add + v_cmpx, v_mov_b64, restore (one save per 16)", k<4>}, {"add only", k<5>},
        {"add + v_cmp_lt_u64 sgpr + 2 v_cndmask sgpr", k<6>}, {"add + v_min_f64 (rate only)", k<7>},
    };
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    uint64_t ref = 0;
    for (auto &v : vs) {
        hipLaunchKernelGGL(v.fn, dim3(blocks), dim3(256), 0, 0, out, 1u);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(v.fn, dim3(blocks), dim3(256), 0, 0, out, 2u);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        uint64_t h[4];
        (void)hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
        if (!ref) ref = h[1];
        const double updates = (double)blocks * 4 * ITER * 16;
        printf("%-52s %8.3f ms  %6.2f cyc/update  result %s\n", v.name, ms, ms * 1e-3 * 2.4e9 / (updates / (ncu * 4.0)), h[1] == ref ? "same" : "DIFFERENT");
    }
    return 0;
}

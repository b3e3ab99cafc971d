// kernels.hpp -- gfx950 (CDNA4, wave64) kernels of the `groot align` hot path.
//
//   sketch_seed_kernel   K1+K2: per read ntHash -> KHF MinHash sketch (registers) -> LSH-Ensemble
//                        containment lookup -> per-read seed slots.          thread per read
//   align_kernel         K3: per read the graphMinion loop: IncrementSubPath call counts and the
//                        hierarchical exact-match DFS alignment.               thread per read
//   order_*_kernel       compact traversal records into canonical (read, ord) order (scan + scatter)
//
// Integer/byte work throughout: no MFMA.  The sketch is VALU bound (64-bit multiply-mix-min per
// (k-mer, slot)); reads are staged through LDS with coalesced 16-byte loads; the index (graphs,
// window sketches, lookup tables: ~100 MB for arg-annot.90) is re-used by every read and stays in
// L2 / Infinity Cache.
#pragma once

#include <hip/hip_runtime.h>

#include "device_types.hpp"

namespace groot {

// ---------------------------------------------------------------------------------------------
// ntHash constants (github.com/will-rowe/nthash v0.2.0; call sites src/minhash/khf.go:38,44)
// ---------------------------------------------------------------------------------------------
#define GROOT_SEED_A 0x3c8bfbb395c60474ULL
#define GROOT_SEED_C 0x3193c18562a02b4cULL
#define GROOT_SEED_G 0x20323ed082572324ULL
#define GROOT_SEED_T 0x295549f54be24456ULL
#define GROOT_MULTI_SEED 0x90b45d39fb6da1faULL
#define GROOT_MULTI_SHIFT 27

__device__ __forceinline__ uint64_t rol64(uint64_t v, unsigned n)
{
    n &= 63;
    return n ? (v << n) | (v >> (64 - n)) : v;
}
__device__ __forceinline__ uint64_t rol1(uint64_t v) { return (v << 1) | (v >> 63); }
__device__ __forceinline__ uint64_t ror1(uint64_t v) { return (v >> 1) | (v << 63); }

// nthash seedTab[b]: the raw byte selects the forward seed; entries 0..7 are the complement table
// reached through (b & 7)
__device__ __forceinline__ uint64_t seed_tab(unsigned b)
{
    switch (b) {
    case 'A': case 'a': case 4: case 5: return GROOT_SEED_A;
    case 'C': case 'c': case 7: return GROOT_SEED_C;
    case 'G': case 'g': case 3: return GROOT_SEED_G;
    case 'T': case 't': case 'U': case 'u': case 1: return GROOT_SEED_T;
    default: return 0;
    }
}

// hash of a whole sketch for the exact-match table (host builds the table with the same function)
__host__ __device__ __forceinline__ uint64_t sketch_hash_step(uint64_t h, uint64_t v)
{
    h = (h ^ v) * 0xff51afd7ed558ccdULL;
    return h ^ (h >> 29);
}
#define GROOT_SKETCH_HASH_INIT 0x9E3779B97F4A7C15ULL
// one byte of a sketch slot for DeviceIndex::band_sig
__host__ __device__ __forceinline__ uint32_t sig8(uint64_t v) { return (uint32_t)((v * 0xD6E8FEB86659FD93ULL) >> 56); }

// 2-bit code of an upper-case base ((b>>1)&3: A=0 C=1 T=2 G=3); 12-bit code of the first 6 bases of r8,
// or -1 if one of them is not ACGT (such a base can still meet the graph's 'N' wildcard)
__host__ __device__ __forceinline__ int kmer6_code(uint64_t r8)
{
    int code = 0;
    for (int i = 0; i < 6; i++) {
        const unsigned b = (unsigned)(r8 >> (8 * i)) & 0xFF;
        if (b != 'A' && b != 'C' && b != 'G' && b != 'T') return -1;
        code |= (int)((b >> 1) & 3) << (2 * i);
    }
    return code;
}
// 8-bit code of the first 4 bases of r8, or -1 if one of them is not ACGT
__host__ __device__ __forceinline__ int kmer4_code(uint64_t r8)
{
    int code = 0;
    for (int i = 0; i < 4; i++) {
        const unsigned b = (unsigned)(r8 >> (8 * i)) & 0xFF;
        if (b != 'A' && b != 'C' && b != 'G' && b != 'T') return -1;
        code |= (int)((b >> 1) & 3) << (2 * i);
    }
    return code;
}
// DeviceIndex::win_prefix of one window: can no level-1/2 start position spell oriented read bases [0,12) = (c0, c1)?
__device__ __forceinline__ bool prefix_absent(const uint32_t *tab, uint64_t c0, uint64_t c1, uint32_t eff)
{
    if (eff < 6) return false;
    const int a = kmer6_code(c0);
    if (a >= 0 && !((tab[a >> 5] >> (a & 31)) & 1u)) return true;
    if (eff < 12) return false;
    const int b = kmer6_code((c0 >> 48) | (c1 << 16));
    return b >= 0 && !((tab[128 + (b >> 5)] >> (b & 31)) & 1u);
}

// the same with the 6-mer codes at hand (all twelve bases are ACGT); needs eff >= 12
__device__ __forceinline__ bool prefix_absent_codes(const uint32_t *tab, uint32_t a, uint32_t b)
{
    if (!((tab[a >> 5] >> (a & 31)) & 1u)) return true;
    return !((tab[128 + (b >> 5)] >> (b & 31)) & 1u);
}

// ---- 8 bases at a time (SWAR on the ASCII bytes; little endian: byte 0 = first base) ----
constexpr uint64_t kLo7 = 0x7F7F7F7F7F7F7F7FULL, kHi1 = 0x8080808080808080ULL, kOnes = 0x0101010101010101ULL;

__device__ __forceinline__ uint64_t ld8(const uint8_t *p)
{
    uint64_t v;
    __builtin_memcpy(&v, p, 8);   // unaligned 8-byte global load
    return v;
}
// bit 7 of byte i set iff byte i of x is non-zero
__device__ __forceinline__ uint64_t nonzero_bytes(uint64_t x) { return (((x & kLo7) + kLo7) | x) & kHi1; }
// bit 7 of byte i set iff graph base i differs from read base i and is not the 'N' wildcard (alignment.go:212-222)
__device__ __forceinline__ uint64_t mismatch8(uint64_t g, uint64_t r)
{
    return nonzero_bytes(g ^ r) & nonzero_bytes(g ^ (kOnes * 'N'));
}
// reverse-complement 8 read bytes: v holds read bytes [e-7, e]; result byte 0 = comp(read[e]).
// A<->T differ by 0x15, C<->G by 0x04, and bit 1 of the ASCII code tells the two pairs apart.  Bytes
// other than ACGT map to bytes other than ACGT, i.e. they never equal a graph base -- the same outcome as
// complementBases' 0 (and 'N' only ever meets the graph's wildcard).
__device__ __forceinline__ uint64_t revcomp8(uint64_t v)
{
    const uint64_t r = __builtin_bswap64(v);
    const uint64_t cg = (r >> 1) & kOnes;
    return r ^ ((cg * 0x04) | ((cg ^ kOnes) * 0x15));
}

// 8 oriented read bases starting at logical index d of the view (rc, clip_lo); bytes past the view's
// end are don't-care (callers mask them).  Never reads before p: the batch buffer may start there.
__device__ __forceinline__ uint64_t read_chunk(const uint8_t *p, uint32_t len, uint32_t rc, uint32_t clip_lo, uint32_t d)
{
    const uint32_t i = d + clip_lo;
    if (!rc) return ld8(p + i);
    const int e = (int)len - 1 - (int)i;                 // oriented base 0 = comp(read[e])
    const uint64_t v = e >= 7 ? ld8(p + (e - 7)) : (e >= 0 ? ld8(p) << (8 * (7 - e)) : 0);
    return revcomp8(v);
}

// first m (<= 8) bases equal under the 'N' wildcard rule?
__device__ __forceinline__ bool prefix_ok(uint64_t g8, uint64_t r8, uint32_t m)
{
    const uint64_t mm = mismatch8(g8, r8);
    return m >= 8 ? mm == 0 : (mm & ((1ULL << (8 * m)) - 1)) == 0;
}
// the same for graph bases known to hold no 'N': plain equality of the first m (1..8) bytes
__device__ __forceinline__ bool prefix_eq(uint64_t g8, uint64_t r8, uint32_t m)
{
    return ((g8 ^ r8) << (8 * (8 - m))) == 0;
}

// bytes [j, j+8) of the 16-byte little-endian window (lo, hi)
__device__ __forceinline__ uint64_t window8(uint64_t lo, uint64_t hi, uint32_t j)
{
    const uint32_t sh = 8 * j;
    return sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
}

// LDS layout of sketch_seed_kernel (bytes)
constexpr uint32_t kLdsTabF = 0;                 // u64[256] seedTab[b]
constexpr uint32_t kLdsTabFout = 2048;           // u64[256] rol(seedTab[b], k)
constexpr uint32_t kLdsTabC = 4096;              // u64[8]   seedTab[c]            c = b & 7
constexpr uint32_t kLdsTabCout = 4096 + 64;      // u64[8]   ror(seedTab[c], 1)
constexpr uint32_t kLdsTabCin = 4096 + 128;      // u64[8]   rol(seedTab[c], k-1)
constexpr uint32_t kLdsReads = 4096 + 192;       // staged read bytes (16-byte aligned)

// both halves of a 32-byte record in flight together, and kept from being split into per-field loads sunk into branches
__device__ __forceinline__ void load32(const void *p, uint4 &a, uint4 &b)
{
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    a = q[0]; b = q[1];
    asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w), "+v"(b.x), "+v"(b.y), "+v"(b.z), "+v"(b.w));
}

// What sketch_sig_kernel fetched ahead for the window it expects to be the read's first seed: the lshe.Key record and the
// four prefix-table words its verdicts need (one round trip together with the text rows instead of two more after them)
struct SeedAhead {
    uint32_t win = kEmpty;     // kEmpty: nothing fetched
    uint4 wa = {}, wb = {};    // WinRec
    uint32_t tf_a = 0, tf_b = 0, tr_a = 0, tr_b = 0;   // prefix-table words of code_f / code_r
};

// per-read bookkeeping every seed kernel ends with
__device__ __forceinline__ void seed_counters(const SeedArgs &a, const uint32_t r, const uint32_t q, const uint32_t n_hits, const bool tabulated = false)
{
    const DeviceIndex &ix = a.ix;
    if (n_hits) {
        if (a.q_seen && ix.q_row[q] == kEmpty) a.q_seen[q] = 1u;   // this kmerCount needs a row of the call-count table
        if (n_hits > a.seed_slots) atomicOr(&a.ctr->flags, kFlagSeedOverflow);
    }
    // sum and maximum of n_hits over the lanes that are here together (ballots per level: reads have one or two seeds), then
    // one pair of atomics per wavefront, sharded: one counter line for all wavefronts costs ~7 ns per atomic, 2.3 ms per
    // 10 M reads.  assign_q_rows_kernel folds the shards into the batch's counter block.
    if (!n_hits && a.trav_cnt) a.trav_cnt[r] = 0;          // the align stage only walks the reads with seeds
    if (a.tab_idx && !tabulated) a.tab_idx[r] = kEmpty;    // its records will not come from the outcome table
    uint32_t total = 0, most = 0;
    for (uint32_t t = 1;; t++) {
        const unsigned long long b = __ballot(n_hits >= t);
        if (!b) break;
        total += (uint32_t)__popcll(b);
        most = t;
    }
    const uint32_t seeded = (uint32_t)__popcll(__ballot(n_hits && !tabulated));   // ... and whose outcome is not tabulated
    const unsigned long long here = __ballot(1);
    if (total && __builtin_amdgcn_mbcnt_hi((uint32_t)(here >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)here, 0u)) == 0) {
        unsigned long long *sh = a.shards + (size_t)(blockIdx.x % kSeedShards) * kSeedShardStride;
        atomicAdd(sh, (unsigned long long)total);
        atomicMax(sh + 1, (unsigned long long)most);
        atomicAdd(sh + 2, (unsigned long long)seeded);
    }
    const unsigned long long tb = __ballot(tabulated);
    if (tb && __builtin_amdgcn_mbcnt_hi((uint32_t)(tb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)tb, 0u)) == 0 && tabulated)
        atomicAdd(a.shards + (size_t)(blockIdx.x % kSeedShards) * kSeedShardStride + 3, (unsigned long long)__popcll(tb));
}

// What every seed kernel leaves behind for one read once its seed windows are known (n_hits of them, the first four in
// s0..s3, the smallest id in min_win): seed_count, the read record with the align stage's verdicts, its scheduling key,
// the batch counters.
__device__ __forceinline__ void seed_epilogue(const SeedArgs &a, const uint32_t r, const uint64_t o0, const uint32_t len, const uint32_t q,
                                              const uint32_t n_hits, const uint32_t min_win, const uint32_t s0, const uint32_t s1,
                                              const uint32_t s2, const uint32_t s3, const bool high, const bool have_codes = false,
                                              const uint32_t code_f = 0, const uint32_t code_r = 0, const SeedAhead *ahead = nullptr, const bool asc = false,
                                              const uint32_t max_win = 0)
{
    // have_codes: the read is all ACGT and at least 12 bases long; code_f / code_r = 2-bit codes of oriented bases [0,12) of
    // the forward read / its reverse complement (base i at bits 2i)
    const DeviceIndex &ix = a.ix;
    a.seed_count[r] = n_hits | (high ? 0x80000000u : 0u);   // bit 31: the read holds a byte > 'T'
    // scheduling key for the align stage: reads are processed in (first seed window, likely orientation) order so
    // that neighbouring lanes walk the same graph nodes in step; reads without seeds sort to the end.  Processing
    // order only -- every output is addressed by read.
    // What the align stage will find for the read's first seed window, per orientation: levels 1-2 cannot start
    // anywhere (prefix tables), the level-3 / level-4 single start position fails its first comparison (alignment.go:72-103).
    // The align stage skips exactly these steps; the sort key groups reads whose orientations have work left, so that
    // neighbouring lanes walk the same graph nodes in step.  Processing order only -- every output is addressed by read.
    uint32_t verdicts = 0;
    if (a.sort_key) {
        uint32_t key = kEmpty;
        if (n_hits) {
            const bool pre = ahead && ahead->win == min_win;
            WinRec wr;
            if (pre) {
                wr.graph = ahead->wa.x; wr.node = ahead->wa.y; wr.offset = ahead->wa.z; wr.l1_hi = ahead->wa.w;
                wr.cn_off = ahead->wb.x; wr.cn_end = ahead->wb.y; wr.seed_s0 = ahead->wb.z; wr.seed_len = ahead->wb.w;
            } else wr = ix.win_rec[min_win];
            const uint32_t *tab = ix.win_prefix + (size_t)min_win * kPrefixWords;
            const uint8_t *p = a.seq + o0;
            const bool in_node = wr.offset < wr.seed_len;         // else levels 3-4 are skipped (alignment.go:199-201)
            const uint64_t g8 = in_node ? ld8(ix.bases + wr.seed_s0 + wr.offset) : 0;
            const uint32_t m34 = min(min(wr.seed_len - wr.offset, len - 1), 8u);
            uint32_t dead = 0;
#pragma unroll
            for (uint32_t t = 0; t < 2; t++) {
                const uint64_t c0 = read_chunk(p, len, t, 0, 0), c1 = read_chunk(p, len, t, 0, 8);
                const uint32_t c12 = t ? code_r : code_f;
                bool no12;
                if (pre) no12 = !(((t ? ahead->tr_a : ahead->tf_a) >> (c12 & 31)) & 1u) || !(((t ? ahead->tr_b : ahead->tf_b) >> ((c12 >> 12) & 31)) & 1u);
                else no12 = have_codes ? prefix_absent_codes(tab, c12 & 0xFFFu, c12 >> 12) : prefix_absent(tab, c0, c1, len);
                uint32_t vt = no12 ? kRecNo12F : 0u;
                if (!in_node || !prefix_ok(g8, (c0 >> 8) | (c1 << 56), m34)) vt |= kRecNo3F;    // read[1:] at (seed, OffSet)
                if (!in_node || !prefix_ok(g8, c0, m34)) vt |= kRecNo4F;                         // read[:len-1] there
                if (vt == (kRecNo12F | kRecNo3F | kRecNo4F)) dead |= 2u >> t;
                verdicts |= vt << (3 * t);
            }
            key = (min_win << 2) | dead;
            if (a.sort_span_bits) {
                // windows spanning a similar number of nodes need similar numbers of DFS steps: keep them together, and
                const uint32_t nn = min(wr.cn_end - wr.cn_off, (1u << a.sort_span_bits) - 1u);
                // longest walks first: the slow chunks are handed out early and the short ones fill the tail of the launch
                key |= (((1u << a.sort_span_bits) - 1u) - nn) << (32u - a.sort_span_bits);
            }
        }
        a.sort_key[r] = key;
        if (a.dfs_list && key != kEmpty) {                    // (few reads are left for the walk: their list is made right here)
            const uint64_t active = __ballot(1);
            const unsigned lane = __lane_id();
            const int leader = __ffsll((unsigned long long)active) - 1;
            uint32_t base = 0;
            if ((int)lane == leader) base = atomicAdd(a.dfs_count, (uint32_t)__popcll(active));
            base = __shfl(base, leader);
            a.dfs_list[base + __popcll(active & ((1ULL << lane) - 1ULL))] = r;
        }
    }
    if (a.read_rec) {
        if (n_hits > kSplitMin && a.long_list) {              // (a read in a hundred; shorter lists are searched as they are)
            const unsigned long long here = __ballot(1);      // one atomic for the lanes that are here together
            const unsigned lane = __lane_id();
            const int leader = __ffsll(here) - 1;
            uint32_t base = 0;
            if ((int)lane == leader) base = atomicAdd(a.long_count, (uint32_t)__popcll(here));
            const uint32_t at = __shfl(base, leader) + (uint32_t)__popcll(here & ((1ULL << lane) - 1ULL));
            if (at < kLongListCap) a.long_list[at] = r;
        }
        uint4 *rq = reinterpret_cast<uint4 *>(a.read_rec + r);
        rq[0] = make_uint4((uint32_t)o0, (uint32_t)(o0 >> 32), len, min(n_hits, kRecSplit - 1u) | verdicts | (asc ? kRecAscending : 0u) | (high ? 0x80000000u : 0u));
        // (more than four seeds: the smallest and the largest window instead of the first two -- the align stage starts at the
        // smallest and knows when nothing is left without looking through the list)
        rq[1] = n_hits > 4 ? make_uint4(min_win, max_win, s2, s3) : make_uint4(s0, s1, s2, s3);
    }
    seed_counters(a, r, q, n_hits);
}

// the same for a read known to be bases [o, o + WindowSize) of a window text row: its verdicts come from the table made at open
__device__ __forceinline__ void seed_epilogue_known(const SeedArgs &a, const uint32_t r, const uint64_t o0, const uint32_t len, const uint32_t q,
                                                    const uint32_t n_hits, const uint32_t min_win, const uint32_t s0, const uint32_t s1,
                                                    const uint32_t s2, const uint32_t s3, const uint32_t vbyte, const uint32_t nodes, const bool asc, const uint32_t max_win)
{
    a.seed_count[r] = n_hits;
    if (a.sort_key) {
        uint32_t key = (min_win << 2) | (vbyte >> 6);
        if (a.sort_span_bits) key |= (((1u << a.sort_span_bits) - 1u) - min(nodes, (1u << a.sort_span_bits) - 1u)) << (32u - a.sort_span_bits);
        a.sort_key[r] = key;
    }
    if (a.read_rec) {
        uint4 *rq = reinterpret_cast<uint4 *>(a.read_rec + r);
        rq[0] = make_uint4((uint32_t)o0, (uint32_t)(o0 >> 32), len, min(n_hits, kRecSplit - 1u) | (a.sort_key ? (vbyte & 0x3Fu) << 24 : 0u) | (asc ? kRecAscending : 0u));
        rq[1] = n_hits > 4 ? make_uint4(min_win, max_win, s2, s3) : make_uint4(s0, s1, s2, s3);
    }
    seed_counters(a, r, q, n_hits);
}

// traversal records of a string with a tabulated outcome (info = its DeviceIndex::sig_info word)
__device__ __forceinline__ uint32_t tab_travs(const DeviceIndex &ix, const uint32_t info)
{
    if (info & kOutNoRec) return 0u;
    const uint32_t f = (info >> kOutTravShift) & 15u;
    if (f != kOutTravLong) return f + 1u;
    return ix.out_tab[(size_t)(info & ((1u << kOutIdxBits) - 1u)) * ix.out_stride_q].w >> 16;   // (a read of a sequence that dozens of graphs share)
}

// the same for a read whose whole graphMinion outcome is tabulated (info = its DeviceIndex::sig_info word): nothing is left for the
// align stage -- no read record, no place in the processing order; order_first_kernel writes its records from the table
__device__ __forceinline__ void seed_epilogue_tab(const SeedArgs &a, const uint32_t r, const uint32_t q, const uint32_t n_hits, const uint32_t info,
                                                  const uint32_t s0, const uint32_t s1, const uint32_t s2, const uint32_t s3)
{
    a.seed_count[r] = n_hits;
    a.sort_key[r] = kEmpty;
    // IncrementSubPath is called once for every seed window of most reads (graphminion.go:60-67; the exceptions: a second seed
    // of a graph that already has its alignment).  Those calls are counted right here, into the batch's own histogram over the
    // windows (fire-and-forget atomics behind the hashing of the other wavefronts; fold_tab_hist_kernel adds the histogram to the
    // call-count table once the batch is known to stand); order_first_kernel takes the others from the table entry.
    uint32_t counted = 0;
    if ((info & kOutAllSeeds) && a.tab_hist && n_hits <= 4u) {
        atomicAdd(&a.tab_hist[s0], 1u);
        if (n_hits > 1u) atomicAdd(&a.tab_hist[s1], 1u);
        if (n_hits > 2u) atomicAdd(&a.tab_hist[s2], 1u);
        if (n_hits > 3u) atomicAdd(&a.tab_hist[s3], 1u);
        counted = kTabCounted;
    }
    a.tab_idx[r] = (info & ((1u << kOutIdxBits) - 1u)) | counted;
    a.trav_cnt[r] = tab_travs(a.ix, info);
    seed_counters(a, r, q, n_hits, true);
}

// ---------------------------------------------------------------------------------------------
// K1+K2
// ---------------------------------------------------------------------------------------------
// (5 workgroups per CU: the register allocator then settles on 81 VGPRs instead of 105 without spilling more, and the
// fifth wave per SIMD lifts VALU issue from 87 % to 90 %; a sixth does not fit the LDS)
#ifndef GROOT_SEED_WAVES
#define GROOT_SEED_WAVES 5
#endif
// M5 >= 0: compile-time value of (k * multiSeed) & 31.  The MultiHash multipliers c_i = i ^ (k*multiSeed) of
// slots i < 32 then equal C0 + (i ^ M5) with C0 = (k*multiSeed) & ~31, so h*c_i for all slots comes from ONE
// 64-bit multiply (h*C0) and a running sum (+h per step) instead of a quarter-rate 64-bit multiply per slot.
// M5 < 0: generic path (any k, any S).
// S = 0 / MAXK = 0: sketch size and hash functions per band are taken from the index at run time (any `groot index -s / -y`,
// cmd/index.go:45-49): the minima then live in an array indexed at run time (private memory), which is correct and slow;
// the sizes people use have compiled instances.
#ifndef GROOT_LSH_ROWS_AHEAD
#define GROOT_LSH_ROWS_AHEAD 1   // (2 and 4 rows fetched together cost more in spilled registers than the round trips they save: 587 / 579 vs 633 Mreads/s on the mixed-length leg)
#endif
constexpr uint32_t kLshHeavyMaxS = 64;   // lsh_heavy_kernel keeps the read's sketch in LDS
constexpr int kGenericMaxS = 256;      // largest sketch the run-time-sized instance handles
constexpr int kGenericMaxBands = kGenericMaxS;   // ... and the most bands (sketch size / maxK >= 1)
// (the S running minima are 2 S registers: at GROOT_SEED_WAVES waves per SIMD (~100 VGPRs) sketch sizes above 30 spilled them --
// 20 ms per 2 M reads at S = 64.  Larger sketches get fewer, larger waves: 3 per SIMD up to S = 48, 2 beyond)
constexpr int seed_waves(int S) { return S == 0 ? 1 : (S <= 30 ? GROOT_SEED_WAVES : (S <= 48 ? 3 : 2)); }
template <int S, int MAXK, bool DUMP, int M5, bool LIST = false>
__global__ __launch_bounds__(kBlock, seed_waves(S)) void sketch_seed_kernel(SeedArgs a)
{
    constexpr int SM = S ? S : kGenericMaxS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t *tabF = reinterpret_cast<uint64_t *>(smem + kLdsTabF);
    uint64_t *tabFout = reinterpret_cast<uint64_t *>(smem + kLdsTabFout);
    uint64_t *tabC = reinterpret_cast<uint64_t *>(smem + kLdsTabC);
    uint64_t *tabCout = reinterpret_cast<uint64_t *>(smem + kLdsTabCout);
    uint64_t *tabCin = reinterpret_cast<uint64_t *>(smem + kLdsTabCin);
    unsigned char *lds_reads = smem + kLdsReads;

    const DeviceIndex &ix = a.ix;
    const unsigned tid = threadIdx.x;
    const uint32_t k = ix.k;
    const int s_ = S ? S : (int)ix.s, maxk_ = MAXK ? MAXK : (int)ix.max_k;
    {
        const uint64_t sd = seed_tab(tid);
        tabF[tid] = sd;
        tabFout[tid] = rol64(sd, k);
        if (tid < 8) {
            tabC[tid] = sd;
            tabCout[tid] = ror1(sd);
            tabCin[tid] = rol64(sd, k - 1);
        }
    }
    // ---- stage this block's reads: one contiguous span, 16 B per lane per load (coalesced) ----
    // (LIST: the reads named by a.todo_list, scattered over the batch: straight from HBM, no staging)
    uint64_t base16 = 0;
    bool in_lds = false;
    if constexpr (!LIST) {
        const uint32_t r0 = blockIdx.x * kBlock;
        const uint32_t r_end = min(r0 + (uint32_t)kBlock, a.n_reads);
        const uint64_t span0 = a.seq_off[r0], span1 = a.seq_off[r_end];
        base16 = span0 & ~15ULL;
        const uint64_t span_bytes = span1 - base16;
        in_lds = span_bytes <= a.lds_read_bytes;
        if (in_lds) {
            const uint4 *src = reinterpret_cast<const uint4 *>(a.seq + base16);
            uint4 *dst = reinterpret_cast<uint4 *>(lds_reads);
            const uint32_t n16 = (uint32_t)((span_bytes + 15) >> 4);
            for (uint32_t i = tid; i < n16; i += kBlock) dst[i] = src[i];
        }
    }
    __syncthreads();

    auto one_read = [&](const uint32_t r) {
    const uint64_t o0 = a.seq_off[r];
    const uint32_t len = (uint32_t)(a.seq_off[r + 1] - o0);
    uint32_t n_hits = 0;
    if (len < k) {                       // NewHasher error -> panic (khf.go:38-41, boss.go:164-166)
        atomicOr(&a.ctr->flags, kFlagShortRead);
        atomicAdd(&a.ctr->short_reads, 1ULL);
        a.seed_count[r] = 0;
        if (a.read_rec) { uint4 *rq = reinterpret_cast<uint4 *>(a.read_rec + r); rq[0] = make_uint4((uint32_t)o0, (uint32_t)(o0 >> 32), len, 0); }
        if (a.sort_key) a.sort_key[r] = kEmpty;
        if (a.trav_cnt) a.trav_cnt[r] = 0;
        if (a.tab_idx) a.tab_idx[r] = kEmpty;
        return;
    }
    if (len > a.max_read_len) {
        atomicOr(&a.ctr->flags, kFlagLongRead);
        a.seed_count[r] = 0;
        if (a.read_rec) { uint4 *rq = reinterpret_cast<uint4 *>(a.read_rec + r); rq[0] = make_uint4((uint32_t)o0, (uint32_t)(o0 >> 32), len, 0); }
        if (a.sort_key) a.sort_key[r] = kEmpty;
        if (a.trav_cnt) a.trav_cnt[r] = 0;
        if (a.tab_idx) a.tab_idx[r] = kEmpty;
        return;
    }
    const uint32_t nk = len - k + 1;
    if (!DUMP && a.ix.max_q && (nk > ix.max_q || ix.q_min_eq[nk] > (uint32_t)s_)) {
        // more k-mers than Containment > t allows at any number of equal slots (reads well beyond the window size): the
        // query cannot return a window, whatever the sketch is -- no hashing
        seed_epilogue(a, r, o0, len, nk, 0, kEmpty, kEmpty, kEmpty, kEmpty, kEmpty, false);
        return;
    }
    // ---- KHF sketch (khf.go:35-55): per slot i, min over k-mers of MultiHash_i(canonical ntHash) ----
    uint64_t m[SM];
#pragma unroll
    for (int i = 0; i < s_; i++) m[i] = ~0ULL;
    const uint64_t M = (uint64_t)k * GROOT_MULTI_SEED;
    unsigned high = 0;                       // any byte > 'T': RevComplement would panic (seqio.go:126)
    auto sketch = [&](const unsigned char *rd) {
        uint64_t fh = 0, rh = 0;
        for (uint32_t j = 0; j < k; j++) {   // ntf64 / ntr64 of the first k-mer in one pass
            const unsigned b = rd[j];
            high |= b > 'T';
            fh = rol1(fh) ^ tabF[b];
            rh ^= rol64(tabC[b & 7], j);
        }
        for (uint32_t j = 0;;) {
            const uint64_t h = fh < rh ? fh : rh;          // canonical
            m[0] = h < m[0] ? h : m[0];
            if (M5 >= 0 && S > 0 && S <= 32) {
                uint64_t acc = h * (M & ~31ULL);           // = h * c_i for the slot with (i ^ M5) == 0
#pragma unroll
                for (int d = 0; d < 32; d++) {
                    const int i = d ^ (M5 & 31);
                    if (i >= 1 && i < S) {
                        const uint64_t t = acc ^ (acc >> GROOT_MULTI_SHIFT);
                        m[i] = t < m[i] ? t : m[i];
                    }
                    acc += h;
                }
            } else {
#pragma unroll
                for (int i = 1; i < s_; i++) {
                    uint64_t t = h * ((uint64_t)i ^ M);
                    t ^= t >> GROOT_MULTI_SHIFT;
                    m[i] = t < m[i] ? t : m[i];
                }
            }
            if (++j == nk) break;
            const unsigned prev = rd[j - 1], end = rd[j + k - 1];
            high |= end > 'T';
            fh = rol1(fh) ^ tabFout[prev] ^ tabF[end];
            rh = ror1(rh) ^ tabCout[prev & 7] ^ tabCin[end & 7];
        }
    };
    if constexpr (LIST) {
        if (a.list_stride_dw && len <= 4 * a.list_stride_dw - 4) {
            // the lane's own copy of its read (odd dword stride: conflict-free): 4-byte loads in flight together instead of
            // two dependent byte loads from HBM per k-mer
            uint32_t *mine = reinterpret_cast<uint32_t *>(lds_reads) + (size_t)tid * a.list_stride_dw;
            for (uint32_t i = 0; i < len; i += 4) {
                uint32_t v;
                __builtin_memcpy(&v, a.seq + o0 + i, 4);     // (reads up to 3 bytes past the read: the batch buffer is padded)
                mine[i >> 2] = v;
            }
            sketch(reinterpret_cast<const unsigned char *>(mine));
        } else sketch(a.seq + o0);
    } else {
        if (in_lds) sketch(lds_reads + (o0 - base16));   // LDS address space
        else sketch(a.seq + o0);                         // span too large for LDS: straight from HBM
    }
    if (DUMP) {
#pragma unroll
        for (int i = 0; i < s_; i++) a.sketch_out[(size_t)r * s_ + i] = m[i];
    }

    // ---- ContainmentIndex.Query (lshe.go:153-175) ----
    const uint32_t q = nk;                                 // kmerCount, boss.go:169
    const uint32_t min_eq = q <= ix.max_q ? ix.q_min_eq[q] : (uint32_t)s_ + 1;
    uint32_t min_win = kEmpty;
    uint32_t s0 = kEmpty, s1 = kEmpty, s2 = kEmpty, s3 = kEmpty;   // first four seeds, for the read record
    bool asc = true;
    uint32_t prev_id = 0, max_win = 0;
    auto hit = [&](uint32_t id) {
        if (n_hits < a.seed_slots) a.seed_win[(size_t)n_hits * a.n_reads + r] = id;
        if (n_hits == 0) s0 = id; else if (n_hits == 1) s1 = id; else if (n_hits == 2) s2 = id; else if (n_hits == 3) s3 = id;
        asc &= n_hits == 0 || id > prev_id;                // (the exact / signature tables return windows in ascending id)
        prev_id = id;
        n_hits++;
        min_win = min(min_win, id);
        max_win = max(max_win, id);
    };
    if (min_eq == (uint32_t)s_) {
        // Containment > t needs every slot equal: windows with an identical sketch.  One probe
        // sequence of the exact-match table (all such windows are consecutive probes).
        uint64_t hs = GROOT_SKETCH_HASH_INIT;
#pragma unroll
        for (int i = 0; i < s_; i++) hs = sketch_hash_step(hs, m[i]);
        const uint32_t tag = (uint32_t)(hs >> 32);
        for (uint32_t slot = (uint32_t)hs & ix.exact_mask;; slot = (slot + 1) & ix.exact_mask) {
            const ExactEntry e = ix.exact[slot];
            if (e.id == kEmpty) break;
            if (e.tag != tag) continue;
            const uint64_t *ws = ix.win_sketch + (size_t)e.id * s_;
            bool same = true;
#pragma unroll
            for (int i = 0; i < s_; i++) same &= ws[i] == m[i];
            if (same) hit(e.id);
        }
    } else if (min_eq < (uint32_t)s_ && a.lsh_list && a.lsh_route != 0) {
        // General LSH Forest query, deferred: the sketch goes to lsh_query_kernel, which deals the rows of equal band prefix of a
        // wavefront's 64 reads over its lanes (here every lane would walk its own rows -- a few to a few hundred -- while the
        // others wait: 8 % of the lane slots doing work)
        const uint64_t active = __ballot(1);
        const unsigned lane = __lane_id();
        const int leader = __ffsll((unsigned long long)active) - 1;
        uint32_t base = 0;
        if ((int)lane == leader) base = atomicAdd(a.lsh_count, (uint32_t)__popcll(active));
        base = __shfl(base, leader);
        const uint32_t pos = base + __popcll(active & ((1ULL << lane) - 1ULL));
        a.lsh_list[pos] = r | (high ? 0x80000000u : 0u);
        uint64_t *sk = a.lsh_sketch + (size_t)pos * s_;
#pragma unroll
        for (int i = 0; i < s_; i++) sk[i] = m[i];
        return;
    } else if (min_eq < (uint32_t)s_) {
        // General LSH Forest query: bands b < L, prefix of K hash values (low 32 bits) per band;
        // a window found through band b is skipped if an earlier band already returned it.
        const int lmax_ = s_ / maxk_;
        const uint32_t K = ix.q_k[q], L = ix.q_l[q];
        const uint32_t n = ix.n_windows;
        const int sl_ = s_ < 32 ? s_ : 32;                  // slots covered by the row signatures
        uint32_t rs[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < sl_; i++) rs[i >> 2] |= sig8(m[i]) << (8 * (i & 3));
        // the rows of equal prefix in every band first: a read with many of them (a sequence that dozens of graphs share) would keep
        // its lane walking while the other 63 wait -- it goes to lsh_query_kernel, which deals a wavefront's rows over its lanes
        // (the run-time-sized instance has room for kGenericMaxBands = kGenericMaxS bands)
        constexpr int LB_ = (S && MAXK) ? (S / MAXK > 0 ? S / MAXK : 1) : kGenericMaxBands;
        uint32_t b_lo[LB_], b_end[LB_];
        uint32_t rows = 0;
#pragma unroll
        for (int b = 0; b < lmax_; b++) {
            if (b >= LB_) break;
            b_lo[b] = n; b_end[b] = n;
            if ((uint32_t)b >= L) continue;
            const uint32_t *keys = ix.band_keys + (size_t)b * n * maxk_;
            auto cmp = [&](uint32_t e) {      // -1 / 0 / +1 : table entry e vs query prefix
                const uint32_t *ke = keys + (size_t)e * maxk_;
#pragma unroll
                for (int j = 0; j < maxk_; j++) {
                    if ((uint32_t)j >= K) break;
                    const uint32_t qv = (uint32_t)m[b * maxk_ + j], kv = ke[j];
                    if (kv != qv) return kv < qv ? -1 : 1;
                }
                return 0;
            };
            // first row of the (sorted) band table with this prefix: hash table over the distinct prefixes
            uint32_t lo = n;
            if (K >= 1) {
                uint64_t hk = GROOT_SKETCH_HASH_INIT;
#pragma unroll
                for (int j = 0; j < maxk_; j++)
                    if ((uint32_t)j < K) hk = sketch_hash_step(hk, (uint32_t)m[b * maxk_ + j]);
                const ExactEntry *tab = ix.band_hash + (((size_t)b * maxk_ + (K - 1)) << ix.band_hash_bits);
                const uint32_t hmask = (1u << ix.band_hash_bits) - 1u, tag = (uint32_t)(hk >> 32);
                for (uint32_t slot = (uint32_t)hk & hmask;; slot = (slot + 1) & hmask) {
                    const ExactEntry e = tab[slot];
                    if (e.id == kEmpty) break;
                    if (e.tag == tag && cmp(e.id) == 0) { lo = e.id; break; }
                }
            }
            b_lo[b] = lo;
            b_end[b] = lo < n ? lo + ix.band_run[((size_t)b * maxk_ + (K - 1)) * n + lo] : n;   // rows with this prefix
            rows += b_end[b] - lo;
        }
        if (a.lsh_list && rows > a.lsh_defer_rows && (uint32_t)s_ <= kLshHeavyMaxS) {
            // (one atomic for the lanes that are here together: the counter is a single address)
            const unsigned long long here = __ballot(1);
            const unsigned lane = __lane_id();
            const int leader = __ffsll(here) - 1;
            uint32_t base = 0;
            if ((int)lane == leader) base = atomicAdd(a.lsh_count, (uint32_t)__popcll(here));
            const uint32_t pos = __shfl(base, leader) + (uint32_t)__popcll(here & ((1ULL << lane) - 1ULL));
            if (pos < a.lsh_cap) {
                a.lsh_list[pos] = r | (high ? 0x80000000u : 0u);
                uint64_t *sk = a.lsh_sketch + (size_t)pos * s_;
#pragma unroll
                for (int i = 0; i < s_; i++) sk[i] = m[i];
                return;
            }
        }
#pragma unroll
        for (int b = 0; b < lmax_; b++) {
            if ((uint32_t)b >= L || b >= LB_) break;
            const uint32_t *ids = ix.band_ids + (size_t)b * n;
            const uint4 *sigs = reinterpret_cast<const uint4 *>(ix.band_sig + (size_t)b * n * 32);
            const uint32_t lo = b_lo[b], e_end = b_end[b];
            // (rows are 32 consecutive bytes each: kRowsAhead of them are fetched together -- the walk is a chain of round trips, 790
            // load instructions per wavefront and read on mixed-length batches, two thirds of the kernel's time spent waiting)
            constexpr uint32_t kRowsAhead = GROOT_LSH_ROWS_AHEAD;
            for (uint32_t e4 = lo; e4 < e_end; e4 += kRowsAhead) {
            uint4 rowa[kRowsAhead], rowb[kRowsAhead];
#pragma unroll
            for (uint32_t i = 0; i < kRowsAhead; i++) {
                const size_t ee = min(e4 + i, e_end - 1u);
                rowa[i] = sigs[2 * ee]; rowb[i] = sigs[2 * ee + 1];
            }
#pragma unroll
            for (uint32_t i = 0; i < kRowsAhead; i++) {
                const uint32_t e = e4 + i;
                if (e >= e_end) break;
                // slots whose signature bytes agree (pad bytes are zero on both sides): an upper bound of the equal slots
                // (this filter is most of the branch's time -- runs of ~40 rows per band: only the dwords that hold slots, and the
                // cheap zero-byte test, which may also flag a byte of value 1 above an equal one: an upper bound still)
                const uint4 sa = rowa[i], sb = rowb[i];
                const uint32_t ws8[8] = {sa.x, sa.y, sa.z, sa.w, sb.x, sb.y, sb.z, sb.w};
                const int nd = (sl_ + 3) >> 2;
                uint32_t same = 0;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    if (i >= nd) break;
                    const uint32_t x = ws8[i] ^ rs[i];
                    same += __popc((x - 0x01010101u) & ~x & 0x80808080u);
                }
                if (same - (4u * (uint32_t)nd - (uint32_t)sl_) + (uint32_t)(s_ - sl_) < min_eq) continue;
                const uint32_t id = ids[e];
                const uint64_t *ws = ix.win_sketch + (size_t)id * s_;
                uint32_t eq = 0;
                bool earlier = false;
#pragma unroll
                for (int bb = 0; bb < lmax_; bb++) {
                    bool pm = true;
#pragma unroll
                    for (int j = 0; j < maxk_; j++) {
                        const uint64_t wv = ws[bb * maxk_ + j];
                        eq += wv == m[bb * maxk_ + j];
                        if ((uint32_t)j < K) pm &= (uint32_t)wv == (uint32_t)m[bb * maxk_ + j];
                    }
                    if (bb < b && pm) earlier = true;
                }
#pragma unroll
                for (int i = lmax_ * maxk_; i < s_; i++) eq += ws[i] == m[i];
                if (!earlier && eq >= min_eq) hit(id);
            }
            }
        }
    }
    seed_epilogue(a, r, o0, len, q, n_hits, min_win, s0, s1, s2, s3, high != 0, false, 0, 0, nullptr, asc, max_win);
    };   // one_read
    if constexpr (LIST) {
        const uint32_t n_todo = *a.todo_count;
        if (!blockIdx.x && !tid) a.ctr->todo_reads = n_todo;
        for (uint32_t i = blockIdx.x * kBlock + tid; i < n_todo; i += gridDim.x * kBlock) one_read(a.todo_list[i]);
    } else {
        const uint32_t r = blockIdx.x * kBlock + tid;
        if (r < a.n_reads) one_read(r);
    }
}

// ---------------------------------------------------------------------------------------------
// K1+K2, fast path: sketch_sig_kernel
// ---------------------------------------------------------------------------------------------
// For reads whose Containment > t needs every sketch slot equal (the exact-table branch of sketch_seed_kernel) the seed
// set is decided without ever forming the 64-bit minima:
//  * MultiHash mixes with t ^= t >> 27, which leaves the top 27 bits of t alone, and truncation is monotone, so
//    top27(min_j mix(t_j)) = min_j top27(t_j) = (min_j hi32(h_j * c_i)) >> 5: a running 32-bit minimum of the raw product's
//    high word gives the top 27 bits of every slot EXACTLY -- a 64-bit add and half a v_min3_u32 per (k-mer, slot) (the
//    compiler pairs two k-mers) instead of seven instructions (add, shift, 2 xor, 64-bit compare, 2 selects);
//  * a window can only equal the read's sketch if its signature (those 27 bits of all S slots) does: the signature table
//    holds every window; no entry -> no seed, rigorously;
//  * an entry is confirmed by TEXT: the window's sketch is the sketch of every WindowSize-mer of the bases it was merged
//    from (graph.go:293-333; re-sketched and compared with Key.Sketch when the ctx is opened), so a read that equals one
//    of them, or its reverse complement (canonical k-mer hashes), has exactly that sketch.  Where in the text to compare is
//    known from the read's smallest k-mer (its position in each text row is in the table entry).  Its seeds are then all
//    windows of the same sketch class, in table order = ascending window id, as the exact table would have returned them;
//  * for a confirmed window-sized read the epilogue's verdicts come from DeviceIndex::sig_verdict -- the full-width seed
//    stage was run on every WindowSize-mer of every text at open;
//  * everything else -- a signature found but no text equal (reads with errors that keep all minimisers, windows merged
//    from another path), bytes other than ACGT (their 2-bit codes say nothing), other lengths / thresholds (LSH-Forest
//    branch), spans too long for the LDS -- goes onto a list and through sketch_seed_kernel<..., LIST> unchanged.
// Reads are staged as 2-bit codes (6.4 KB per 256 x 100 bp instead of 25.6 KB), the rolling hash takes both strands'
// table entries of the entering and the leaving base with one 16-byte LDS read each.
#define GROOT_SIG_HASH_INIT 0x2545F4914F6CDD1DULL
__host__ __device__ __forceinline__ uint64_t sig_hash_step(uint64_t x, uint32_t top27) { return ((x << 13) | (x >> 51)) ^ top27; }
__host__ __device__ __forceinline__ uint64_t sig_hash_fin(uint64_t x)
{
    x *= 0xff51afd7ed558ccdULL;
    return x ^ (x >> 32);
}

// LDS: a static 512-byte table ({leaving, entering} base -> 16-byte entries, at strides 16 and 64, see below; static so
// that its address folds into the ds_read offsets), then dynamic:
constexpr uint32_t kTextBad = 2048;    // text_lookup_kernel: bytes of its bad-group bit set (one bit per 4 bases of a span of up to 64 KB)
constexpr uint32_t kSigBad = 0;        // 4096 bits: 16-byte chunks of the span holding a byte other than ACGT
constexpr uint32_t kSigCodes = 512;    // one dword per 16 bases
#ifndef GROOT_SIG_WAVES
#define GROOT_SIG_WAVES 6
#endif

// four ASCII bases -> four 2-bit codes in bits 0..7; bad collects x ^ "ACTG"[code] (non-zero: some byte is not ACGT)
__device__ __forceinline__ uint32_t codes_of4(uint32_t x, uint32_t &bad)
{
    const uint32_t y = (x >> 1) & 0x03030303u;
    bad |= x ^ __builtin_amdgcn_perm(0x47544341u, 0x47544341u, y);
    return (y * 0x01041040u) >> 24;
}
__device__ __forceinline__ uint64_t seed_of_code(unsigned c)
{
    return c == 0 ? GROOT_SEED_A : c == 1 ? GROOT_SEED_C : c == 2 ? GROOT_SEED_T : GROOT_SEED_G;
}
// append read r to the list of sketch_seed_kernel<..., LIST>: one atomic per wavefront and call site
__device__ __forceinline__ void todo_push(const SeedArgs &a, uint32_t r)
{
    const uint64_t active = __ballot(1);
    const unsigned lane = __lane_id();
    const int leader = __ffsll((unsigned long long)active) - 1;
    uint32_t base = 0;
    if ((int)lane == leader) base = atomicAdd(a.todo_count, (uint32_t)__popcll(active));
    base = __shfl(base, leader);
    a.todo_list[base + __popcll(active & ((1ULL << lane) - 1ULL))] = r;
}

// does sketch j (s words) equal the sketch of window owner[j]?  (the proof of the window texts, groot_hip_open)
__global__ __launch_bounds__(kBlock) void sketch_equal_kernel(const uint64_t *__restrict__ sk, const uint32_t *__restrict__ owner,
                                                              const uint64_t *__restrict__ win_sketch, uint32_t s, uint32_t n, uint8_t *__restrict__ differs)
{
    const uint32_t j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= n) return;
    const uint64_t *a = sk + (size_t)j * s, *b = win_sketch + (size_t)owner[j] * s;
    uint64_t d = 0;
    for (uint32_t i = 0; i < s; i++) d |= a[i] ^ b[i];
    differs[j] = d != 0;
}

// first position of the smallest canonical ntHash among the k-mers of every window text row (ASCII, kTextMax bytes per
// row, two rows per window): sketch_sig_kernel finds where a read lies inside a text from where its own smallest k-mer is
__global__ __launch_bounds__(kBlock) void text_argmin_kernel(const uint8_t *__restrict__ text, const uint32_t *__restrict__ text_len, uint32_t n_rows,
                                                              uint32_t k, uint8_t *__restrict__ pos)
{
    const uint32_t row = blockIdx.x * kBlock + threadIdx.x;
    if (row >= n_rows) return;
    const uint32_t len = text_len[row >> 1];
    const uint8_t *t = text + (size_t)row * kTextMax;
    uint32_t best_pos = 0;
    if (len >= k) {
        uint64_t fh = 0, rh = 0;
        for (uint32_t j = 0; j < k; j++) {
            fh = rol1(fh) ^ seed_tab(t[j]);
            rh ^= rol64(seed_tab(t[j] & 7), j);
        }
        uint64_t best = fh < rh ? fh : rh;
        for (uint32_t j = 1; j + k <= len; j++) {
            fh = rol1(fh) ^ rol64(seed_tab(t[j - 1]), k) ^ seed_tab(t[j + k - 1]);
            rh = ror1(rh) ^ ror1(seed_tab(t[j - 1] & 7)) ^ rol64(seed_tab(t[j + k - 1] & 7), k - 1);
            const uint64_t h = fh < rh ? fh : rh;
            if (h < best) { best = h; best_pos = j; }
        }
    }
    pos[row] = (uint8_t)best_pos;
}

// TW: dwords of a packed read the text comparison handles (reads of up to 16 * TW bases; longer ones take the full-width kernel)
template <int S, int M5, int TW>
__global__ __launch_bounds__(kBlock, GROOT_SIG_WAVES) void sketch_sig_kernel(SeedArgs a)
{
    static_assert(S >= 1 && S <= 32 && M5 >= 0 && M5 < 32, "slots i < 32 with a compile-time (k * multiSeed) & 31 only");
    static_assert(TW >= 1 && 16 * TW <= (int)kTextMax, "a read cannot be longer than a window text");
    constexpr int kTextWords = TW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ __attribute__((aligned(512))) unsigned char tab[512];
    uint32_t *badbits = reinterpret_cast<uint32_t *>(smem + kSigBad);
    uint32_t *codes = reinterpret_cast<uint32_t *>(smem + kSigCodes);
    const DeviceIndex &ix = a.ix;
    const unsigned tid = threadIdx.x;
    const uint32_t k = ix.k;
    // table entries: leaving base c -> {rol(seed[c], k), ror(seed[comp c], 1)}, entering base c -> {seed[c], rol(seed[comp c], k-1)}
    // (ntHash's forward / reverse-strand updates).  Two copies: entries 16 bytes apart for a code sitting at bits 4..5 of a
    // register, 64 bytes apart for one at bits 6..7 -- the address is then ONE v_and of the shifted code word.
    if (tid < 4) {
        const uint64_t f = seed_of_code(tid), fc = seed_of_code(tid ^ 2u);
        const uint64_t of = rol64(f, k), orv = ror1(fc), iv = f, ir = rol64(fc, k - 1);
        const uint4 eo = make_uint4((uint32_t)of, (uint32_t)(of >> 32), (uint32_t)orv, (uint32_t)(orv >> 32));
        const uint4 ei = make_uint4((uint32_t)iv, (uint32_t)(iv >> 32), (uint32_t)ir, (uint32_t)(ir >> 32));
        *reinterpret_cast<uint4 *>(tab + 16 * tid) = eo;
        *reinterpret_cast<uint4 *>(tab + 64 + 16 * tid) = ei;
        *reinterpret_cast<uint4 *>(tab + 256 + 64 * tid) = eo;
        *reinterpret_cast<uint4 *>(tab + 256 + 16 + 64 * tid) = ei;
    }
    if (tid < 128) badbits[tid] = 0;
    __shared__ uint32_t list_cnt, list_base;               // the reads this workgroup leaves to the list pass
    if (tid == 0) list_cnt = 0;
    // ---- stage this block's reads as 2-bit codes: one contiguous span, 16 bases per lane per load ----
    const uint32_t r0 = blockIdx.x * kBlock;
    const uint32_t r_end = min(r0 + (uint32_t)kBlock, a.n_reads);
    const uint64_t span0 = a.seq_off[r0], span1 = a.seq_off[r_end];
    const uint64_t base16 = span0 & ~15ULL;
    const uint64_t span_bytes = span1 - base16;
    const bool in_lds = span_bytes <= a.lds_read_bytes;
    __syncthreads();
    if (in_lds) {
        const uint4 *src = reinterpret_cast<const uint4 *>(a.seq + base16);
        const uint32_t n16 = (uint32_t)((span_bytes + 15) >> 4);
        for (uint32_t i = tid; i < n16; i += kBlock) {
            const uint4 v = src[i];
            uint32_t bad = 0;
            const uint32_t c = codes_of4(v.x, bad) | (codes_of4(v.y, bad) << 8) | (codes_of4(v.z, bad) << 16) | (codes_of4(v.w, bad) << 24);
            codes[i] = c;
            if (bad) atomicOr(&badbits[i >> 5], 1u << (i & 31));
        }
    }
    __syncthreads();
    const uint32_t r = r0 + tid;
    if (r >= a.n_reads) return;
    const uint64_t o0 = a.seq_off[r];
    const uint32_t len = (uint32_t)(a.seq_off[r + 1] - o0);
    const uint32_t q = len - k + 1;                        // kmerCount, boss.go:169
    if (len >= k && len <= a.max_read_len && (q > ix.max_q || ix.q_min_eq[q] > (uint32_t)S)) {
        // Containment > t is out of reach for this many k-mers: no seed, and nothing to hash
        seed_epilogue(a, r, o0, len, q, 0, kEmpty, kEmpty, kEmpty, kEmpty, kEmpty, false);
        return;
    }
    // (len >= WindowSize: only then does the read cover whole WindowSize-mers of a text, whose sketches are proven; a shorter
    // read is a substring with FEWER k-mers -- its minima may differ below the 27 signature bits -- and takes the full-width kernel)
    bool fast = in_lds && len >= k && len >= ix.w && len <= a.max_read_len && len <= 16u * TW && q <= ix.max_q;
    if (fast) fast = ix.q_min_eq[q] == (uint32_t)S;        // else: LSH-Forest branch
    if (fast) {
        const uint32_t c0 = (uint32_t)(o0 - base16) >> 4, c1 = (uint32_t)(o0 - base16 + len - 1) >> 4;
        for (uint32_t w = c0 >> 5; w <= c1 >> 5; w++) {
            uint32_t bits = badbits[w];
            if (w == c0 >> 5) bits &= ~0u << (c0 & 31);
            if (w == c1 >> 5) bits &= ~0u >> (31 - (c1 & 31));
            if (bits) fast = false;
        }
    }
    {
        // the reads left to the list pass (other lengths, bytes other than ACGT, the LSH-Forest branch): counted per workgroup -- one
        // LDS atomic per wavefront, ONE global atomic per workgroup.  (One global atomic per wavefront on the single counter cost
        // 0.9 of the kernel's 1.9 ms on 8 M mixed-length reads, where every wavefront has such reads: ~7 ns each.)  Every thread
        // still here takes part; wavefronts that have left do not count at the barrier.
        const unsigned long long here = __ballot(1), mb = __ballot(!fast);
        const unsigned lane = tid & 63u;
        const int leader = __ffsll(here) - 1;
        uint32_t wave_base = 0;
        if ((int)lane == leader && mb) wave_base = atomicAdd(&list_cnt, (uint32_t)__popcll(mb));
        wave_base = __shfl(wave_base, leader);
        __syncthreads();
        if ((int)lane == leader && mb && wave_base == 0) list_base = atomicAdd(a.todo_count, list_cnt);
        __syncthreads();
        if (!fast) {
            a.todo_list[list_base + wave_base + (uint32_t)__popcll(mb & ((1ULL << lane) - 1ULL))] = r;
            return;
        }
    }

    // ---- top 32 bits of the running minima (khf.go:35-55) ----
    uint32_t m[S];
#pragma unroll
    for (int i = 0; i < S; i++) m[i] = ~0u;
    const uint64_t C0 = ((uint64_t)k * GROOT_MULTI_SEED) & ~31ULL;
    uint64_t fh = 0, rh = 0;
    uint32_t key0 = ~0u, kj = 0;     // smallest (top 24 bits of h | k-mer index): where the read's smallest k-mer is (ties: see the text compare)
    auto ent = [&](uint32_t byte_off) { return *reinterpret_cast<const uint4 *>(tab + byte_off); };
    auto roll = [&](const uint4 eo, const uint4 ei) {
        const uint32_t fl = (uint32_t)fh, fu = (uint32_t)(fh >> 32), rl = (uint32_t)rh, ru = (uint32_t)(rh >> 32);
        const uint32_t nfl = __builtin_amdgcn_alignbit(fl, fu, 31) ^ eo.x ^ ei.x, nfu = __builtin_amdgcn_alignbit(fu, fl, 31) ^ eo.y ^ ei.y;   // rol 1
        const uint32_t nrl = __builtin_amdgcn_alignbit(ru, rl, 1) ^ eo.z ^ ei.z, nru = __builtin_amdgcn_alignbit(rl, ru, 1) ^ eo.w ^ ei.w;     // ror 1
        fh = (uint64_t)nfl | ((uint64_t)nfu << 32);
        rh = (uint64_t)nrl | ((uint64_t)nru << 32);
    };
    auto slots = [&]() {
        const uint64_t h = fh < rh ? fh : rh;              // canonical
        m[0] = min(m[0], (uint32_t)(h >> 32));
        key0 = min(key0, ((uint32_t)(h >> 32) & ~255u) | kj);
        kj++;
        const uint32_t hl = (uint32_t)h, hu = (uint32_t)(h >> 32);
        uint64_t acc = (uint64_t)hl * (uint32_t)C0;        // h * C0 = h * c_i for the slot with (i ^ M5) == 0
        acc += (uint64_t)(hl * (uint32_t)(C0 >> 32) + hu * (uint32_t)C0) << 32;
#pragma unroll
        for (int d = 0; d < 32; d++) {
            const int i = d ^ M5;
            if (i >= 1 && i < S) m[i] = min(m[i], (uint32_t)(acc >> 32));
            acc += h;
        }
    };
    const uint32_t P = 2u * (uint32_t)(o0 - base16);       // bit position of base 0 in `codes`
    {   // first k-mer: bases enter, none leaves
        uint32_t d = P >> 5, lo = codes[d];
        for (uint32_t i = 0; i < k; i += 16) {
            const uint32_t nx = codes[++d];
            uint64_t t = (uint64_t)__builtin_amdgcn_alignbit(nx, lo, P & 31) << 4;
            lo = nx;
            const uint32_t cnt = min(16u, k - i);
            for (uint32_t j = 0; j < cnt; j++) {
                roll(make_uint4(0, 0, 0, 0), ent(64 + ((uint32_t)t & 0x30u)));
                t >>= 2;
            }
        }
    }
    slots();
    {
        uint32_t left = len - k;                           // k-mers still to come
        uint32_t di = (P + 2 * k) >> 5, dn = P >> 5;
        const uint32_t si = (P + 2 * k) & 31, sn = P & 31;
        uint32_t li = codes[di], ln = codes[dn];
        while (left >= 16) {
            const uint32_t ni = codes[++di], nn = codes[++dn];
            const uint32_t wi = __builtin_amdgcn_alignbit(ni, li, si), wo = __builtin_amdgcn_alignbit(nn, ln, sn);
            li = ni; ln = nn;
            uint64_t ti = (uint64_t)wi << 4, to = (uint64_t)wo << 4;     // code of the pair's first base at bits 4..5, second at 6..7
#pragma unroll 1
            for (int p = 0; p < 8; p++) {
                const uint32_t a0 = (uint32_t)to & 0x30u, b0 = (uint32_t)ti & 0x30u, a1 = (uint32_t)to & 0xC0u, b1 = (uint32_t)ti & 0xC0u;
                ti >>= 4; to >>= 4;
                roll(ent(a0), ent(64 + b0));
                slots();
                roll(ent(256 + a1), ent(256 + 16 + b1));
                slots();
            }
            left -= 16;
        }
        if (left) {
            const uint32_t wi = __builtin_amdgcn_alignbit(codes[di + 1], li, si), wo = __builtin_amdgcn_alignbit(codes[dn + 1], ln, sn);
            for (uint32_t j = 0; j < left; j++) {
                roll(ent(((wo >> (2 * j)) & 3u) << 4), ent(64 + (((wi >> (2 * j)) & 3u) << 4)));
                slots();
            }
        }
    }

    // ---- ContainmentIndex.Query (lshe.go:153-175), every slot must be equal ----
    uint64_t x = GROOT_SIG_HASH_INIT;
#pragma unroll
    for (int i = 0; i < S; i++) x = sig_hash_step(x, m[i] >> 5);
    x = sig_hash_fin(x);
    const uint32_t tag = (uint32_t)(x >> 32);
    // the read as packed codes in registers, and a comparison with len bases of a packed text row starting at base o
    uint32_t rdw[kTextWords];
#pragma unroll
    for (int j = 0; j < kTextWords; j++) rdw[j] = __builtin_amdgcn_alignbit(codes[(P >> 5) + j + 1], codes[(P >> 5) + j], P & 31);
    const uint32_t n_full = len >> 4, tail_mask = (1u << (2 * (len & 15))) - 1u;
    auto row_differs = [&](const uint8_t *row, uint32_t o) {
        uint32_t t[kTextWords + 1];
        __builtin_memcpy(t, row + (o >> 2), sizeof t);     // unaligned; runs into the next row, which the masks ignore
        uint32_t diff = 0;
#pragma unroll
        for (int j = 0; j < kTextWords; j++) {
            const uint32_t mask = (uint32_t)j < n_full ? ~0u : ((uint32_t)j == n_full ? tail_mask : 0u);
            diff |= (__builtin_amdgcn_alignbit(t[j + 1], t[j], 2 * (o & 3)) ^ rdw[j]) & mask;
        }
        return diff;
    };
    // 2-bit codes of the first twelve bases of both orientations, for the prefix-table verdicts
    uint32_t code_r = 0;
    const uint32_t code_f = rdw[0] & 0xFFFFFFu;
    if (len >= 12) {
        const uint32_t Q = P + 2 * (len - 12);
        const uint32_t x = __builtin_amdgcn_alignbit(codes[(Q >> 5) + 1], codes[Q >> 5], Q & 31) & 0xFFFFFFu;   // bases len-12 .. len-1
        const uint32_t y = __builtin_bitreverse32(x) >> 8;                                                        // last base first, bit pairs swapped
        code_r = (((y & 0x555555u) << 1) | ((y >> 1) & 0x555555u)) ^ 0xAAAAAAu;                                   // pairs restored, complemented (code ^ 2)
    }
    SeedAhead ahead;
    const bool use_table = ix.sig_info && len == ix.w && a.sort_key;   // the epilogue's answers for window-sized text reads exist already
    uint32_t vbyte = 0, nodes_ahead = 0, first_id = kEmpty;
    bool have_vbyte = false;
    const uint32_t j0 = key0 & 255u;
    uint32_t n_tagged = 0, only_id = kEmpty, cls = kEmpty;
    const uint4 *sig = reinterpret_cast<const uint4 *>(ix.sig);
    for (uint32_t slot = (uint32_t)x & ix.sig_mask;; slot = (slot + 1) & ix.sig_mask) {
        const uint4 e = sig[slot];                         // {tag, id, cls, sig_text_pack(text_len, argmin fwd, argmin rc)}
        if (e.y == kEmpty) break;
        if (e.x != tag) continue;
        n_tagged++;
        only_id = e.y;
        if (n_tagged == 1) first_id = e.y;
        if (n_tagged == 1 && use_table) nodes_ahead = ix.win_nodes[e.y];
        else if (n_tagged == 1 && len >= 12 && a.sort_key) {    // most likely the read's only seed: what the verdicts will need, in flight now
            ahead.win = e.y;
            load32(ix.win_rec + e.y, ahead.wa, ahead.wb);
            const uint32_t *tab = ix.win_prefix + (size_t)e.y * kPrefixWords;
            ahead.tf_a = tab[(code_f & 0xFFFu) >> 5]; ahead.tf_b = tab[128 + (code_f >> 17)];
            ahead.tr_a = tab[(code_r & 0xFFFu) >> 5]; ahead.tr_b = tab[128 + (code_r >> 17)];
        }
        const uint32_t tl = sig_text_len(e.w);
        if (cls != kEmpty || tl < len) continue;
        // the text's smallest k-mer (first occurrence) must be the read's: that fixes the offset, per orientation
        const uint8_t *rows = ix.win_text + (size_t)e.y * (2 * kTextMax / 4);
        const uint32_t of = sig_text_argmin(e.w, 0) - j0, orc = sig_text_argmin(e.w, 1) - j0;
        const bool okf = of <= tl - len, okr = orc <= tl - len;
        uint32_t vf = 0, vr = 0;
        if (use_table) {
            const uint32_t *vt = ix.sig_info + (size_t)e.y * 2 * ix.sig_verdict_stride;
            if (okf) vf = vt[of];
            if (okr) vr = vt[ix.sig_verdict_stride + orc];
        }
        const uint32_t df = okf ? row_differs(rows, of) : 1u, dr = okr ? row_differs(rows + kTextMax / 4, orc) : 1u;
        if (!df || !dr) {
            cls = e.z;
            have_vbyte = use_table;
            vbyte = !df ? vf : vr;
        }
    }
    if (n_tagged && cls == kEmpty) { todo_push(a, r); return; }
    uint32_t n_hits = 0, min_win = kEmpty;
    uint32_t s0 = kEmpty, s1 = kEmpty, s2 = kEmpty, s3 = kEmpty;
    bool asc = true;
    uint32_t prev_id = 0, max_win = 0;
    auto hit = [&](uint32_t id) {
        if (n_hits < a.seed_slots) a.seed_win[(size_t)n_hits * a.n_reads + r] = id;
        if (n_hits == 0) s0 = id; else if (n_hits == 1) s1 = id; else if (n_hits == 2) s2 = id; else if (n_hits == 3) s3 = id;
        asc &= n_hits == 0 || id > prev_id;                // (the exact / signature tables return windows in ascending id)
        prev_id = id;
        n_hits++;
        min_win = min(min_win, id);
        max_win = max(max_win, id);
    };
    if (n_tagged == 1) hit(only_id);
    else if (n_tagged)
        for (uint32_t slot = (uint32_t)x & ix.sig_mask;; slot = (slot + 1) & ix.sig_mask) {
            const uint4 e = sig[slot];
            if (e.y == kEmpty) break;
            if (e.x == tag && e.z == cls) hit(e.y);
        }
    if (have_vbyte && (vbyte & kOutTab) && a.tab_idx) seed_epilogue_tab(a, r, q, n_hits, vbyte, s0, s1, s2, s3);
    else if (have_vbyte) seed_epilogue_known(a, r, o0, len, q, n_hits, min_win, s0, s1, s2, s3, vbyte, min_win == first_id ? nodes_ahead : (uint32_t)ix.win_nodes[min_win], asc, max_win);
    else seed_epilogue(a, r, o0, len, q, n_hits, min_win, s0, s1, s2, s3, false, len >= 12, code_f, code_r, &ahead, asc, max_win);   // all bytes are ACGT
}

// ---------------------------------------------------------------------------------------------
// K1+K2+K3 for reads the index has seen before: text_lookup_kernel
// ---------------------------------------------------------------------------------------------
// A read that IS one of the window-text strings (bases [o, o + WindowSize) of a text row, either orientation) needs no hashing at
// all: groot_hip_open proved per string that its KHF sketch is the window's (the full-width kernel sketched every one of them),
// so ContainmentIndex.Query returns the window's sketch class for it (lshe.go:153-175 at a threshold that needs every slot equal),
// and the outcome table holds what the graphMinion loop does with it.  The strings with a tabulated outcome whose IncrementSubPath
// calls are exactly their seed windows sit in a hash table keyed by the TEXT (2 bits per base): one probe, one 64-byte entry holding
// the text itself -- equality is decided on the bases, never on the hash.  A path string with a few bytes other than ACGT (an N
// in an indexed sequence) has an entry too: those bytes and their positions follow the bases (device_types.hpp text_exc_dwords)
// and are compared like them.  Everything else (no entry: reads with errors, reads from elsewhere, other lengths) goes onto the
// list of sketch_seed_kernel<..., LIST>, which hashes it.
//   entry (64 bytes): [0] tag  [1] DeviceIndex::sig_info word of the string (0 = free slot)  [2..] the string, 16 bases per dword
__host__ __device__ __forceinline__ uint64_t text_hash_step(uint64_t h, uint32_t dw)
{
    h = (h ^ dw) * 0x9E3779B97F4A7C15ULL;
    return h ^ (h >> 29);
}
#define GROOT_TEXT_HASH_INIT 0xD6E8FEB86659FD93ULL
// ---------------------------------------------------------------------------------------------
// K2, LSH-Forest branch, wave-cooperative: lsh_query_kernel
// ---------------------------------------------------------------------------------------------
// ContainmentIndex.Query for reads whose Containment > t needs fewer than all slots equal (lshe.go:153-175: lshensemble's forest
// query with K hash values per band over L bands, then the exact containment test).  A read's candidate rows -- the rows of equal
// K-prefix in each of its L sorted band tables -- number a few to a few hundred; walked per lane, a wavefront loops until its
// slowest lane is through.  Here the hashing kernel hands over the sketches (SeedArgs::lsh_list / lsh_sketch) and a wavefront
// takes 64 of them at a time:
//   A  per lane: the first row and the run length of its prefix in every band (hash table over the prefixes), its signature bytes
//   B  prefix sum of the run lengths over the lanes: T rows in all
//   C  the T rows dealt over the lanes, 64 per step (owner by bisection of the prefix sums in LDS): the 32-byte row signature
//      against the OWNER's signature bytes (LDS) -- ballot + popcount append the survivors to a queue in LDS
//   D  every lane verifies its own survivors against its 64-bit sketch (exact #equal slots, not returned by an earlier band) and
//      writes its seed windows; then the epilogue every seed kernel ends with.
// Same rows, same tests, same order of hits per read as the per-lane branch of sketch_seed_kernel.
constexpr uint32_t kLshQueue = 512;            // survivors a wavefront collects before its lanes verify them
constexpr uint32_t kLshMaxBands = 16;          // bands per read this kernel handles (sketch size / maxK; `groot index` default: 5)
__host__ __device__ inline uint32_t lsh_wave_lds_dw(uint32_t lb) { return 64 * 8 + 65 + 64 + 64 * lb + 64 * (lb + 1) + 64 + 64 + 2 * kLshQueue; }
__global__ __launch_bounds__(kBlock) void lsh_query_kernel(SeedArgs a)
{
    extern __shared__ uint32_t lsh_lds[];
    const DeviceIndex &ix = a.ix;
    const uint32_t S = ix.s, maxk = ix.max_k, LB = ix.l_max, n = ix.n_windows;
    const uint32_t sl = S < 32 ? S : 32, nd = (sl + 3) >> 2;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t *W = lsh_lds + (size_t)wave * lsh_wave_lds_dw(LB);
    uint32_t *rs = W;                          // [64][8] signature bytes of the lanes' sketches
    uint32_t *lbase = rs + 64 * 8;             // [65] exclusive prefix of the lanes' row counts
    uint32_t *lmin = lbase + 65;               // [64] min #equal slots per lane
    uint32_t *blo = lmin + 64;                 // [64][LB] first row per band
    uint32_t *bcum = blo + 64 * LB;            // [64][LB + 1] rows before band b
    uint32_t *qcnt = bcum + 64 * (LB + 1);     // [64] survivors per owner in the queue
    uint32_t *qfirst = qcnt + 64;              // [64] first of them
    uint32_t *queue = qfirst + 64;             // [kLshQueue][2]: owner | band << 8, window
    auto wave_sync = []() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); };
    const uint32_t n_list = a.lsh_cap ? min(*a.lsh_count, a.lsh_cap) : *a.lsh_count;
    for (uint32_t base = (blockIdx.x * (kBlock / 64) + wave) * 64; base < n_list; base += gridDim.x * kBlock) {
        const uint32_t li = base + lane;
        const bool valid = li < n_list;
        // ---- A ----
        uint32_t r = 0, high = 0, len = 0, q = 0, K = 0, L = 0, min_eq = S + 1, rows = 0;
        uint64_t o0 = 0;
        const uint64_t *sk = a.lsh_sketch + (size_t)(valid ? li : 0) * S;
        for (uint32_t i = 0; i < 8; i++) rs[lane * 8 + i] = 0;
        if (valid) {
            const uint32_t e = a.lsh_list[li];
            r = e & 0x7FFFFFFFu; high = e >> 31;
            o0 = a.seq_off[r];
            len = (uint32_t)(a.seq_off[r + 1] - o0);
            q = len - ix.k + 1;
            K = ix.q_k[q]; L = min((uint32_t)ix.q_l[q], LB); min_eq = ix.q_min_eq[q];
#pragma unroll
            for (uint32_t wd = 0; wd < 8; wd++) {
                uint32_t v = 0;
#pragma unroll
                for (uint32_t i = 0; i < 4; i++)
                    if (4 * wd + i < sl) v |= sig8(sk[4 * wd + i]) << (8 * i);
                rs[lane * 8 + wd] = v;
            }
        }
        lmin[lane] = min_eq;
        for (uint32_t b = 0; b < LB; b++) {
            uint32_t lo = n, run = 0;
            if (valid && b < L && K >= 1) {
                const uint32_t *keys = ix.band_keys + (size_t)b * n * maxk;
                uint64_t hk = GROOT_SKETCH_HASH_INIT;
                for (uint32_t j = 0; j < K; j++) hk = sketch_hash_step(hk, (uint32_t)sk[b * maxk + j]);
                const ExactEntry *tab = ix.band_hash + (((size_t)b * maxk + (K - 1)) << ix.band_hash_bits);
                const uint32_t hmask = (1u << ix.band_hash_bits) - 1u, tag = (uint32_t)(hk >> 32);
                for (uint32_t slot = (uint32_t)hk & hmask;; slot = (slot + 1) & hmask) {
                    const ExactEntry e = tab[slot];
                    if (e.id == kEmpty) break;
                    if (e.tag != tag) continue;
                    bool same = true;
                    for (uint32_t j = 0; j < K; j++) same &= keys[(size_t)e.id * maxk + j] == (uint32_t)sk[b * maxk + j];
                    if (same) { lo = e.id; break; }
                }
                if (lo < n) run = ix.band_run[((size_t)b * maxk + (K - 1)) * n + lo];
            }
            blo[lane * LB + b] = lo;
            bcum[lane * (LB + 1) + b] = rows;
            rows += run;
        }
        bcum[lane * (LB + 1) + LB] = rows;
        // ---- B ----
        uint32_t incl = rows;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = __shfl_up(incl, o);
            if ((int)lane >= o) incl += v;
        }
        lbase[lane] = incl - rows;
        const uint32_t T = __shfl(incl, 63);
        if (lane == 0) lbase[64] = T;
        qcnt[lane] = 0; qfirst[lane] = kEmpty;
        wave_sync();
        // per-lane results
        uint32_t n_hits = 0, min_win = kEmpty, max_win = 0, prev_id = 0;
        uint32_t s0 = kEmpty, s1 = kEmpty, s2 = kEmpty, s3 = kEmpty;
        bool asc = true;
        auto hit = [&](uint32_t id) {
            if (n_hits < a.seed_slots) a.seed_win[(size_t)n_hits * a.n_reads + r] = id;
            if (n_hits == 0) s0 = id; else if (n_hits == 1) s1 = id; else if (n_hits == 2) s2 = id; else if (n_hits == 3) s3 = id;
            asc &= n_hits == 0 || id > prev_id;
            prev_id = id;
            n_hits++;
            min_win = min(min_win, id);
            max_win = max(max_win, id);
        };
        uint32_t qn = 0;                                    // survivors in the queue (wave-uniform)
        auto verify = [&]() {                               // ---- D ----
            wave_sync();
            const uint32_t c = qcnt[lane], f = qfirst[lane];
            for (uint32_t j = 0; j < c; j++) {
                const uint32_t tagw = queue[2 * (f + j)], id = queue[2 * (f + j) + 1];
                const uint32_t b = tagw >> 8;
                const uint64_t *ws = ix.win_sketch + (size_t)id * S;
                uint32_t eq = 0;
                bool earlier = false;
                for (uint32_t bb = 0; bb < LB; bb++) {
                    bool pm = true;
                    for (uint32_t j2 = 0; j2 < maxk; j2++) {
                        const uint64_t wv = ws[bb * maxk + j2], mv = sk[bb * maxk + j2];
                        eq += wv == mv;
                        if (j2 < K) pm &= (uint32_t)wv == (uint32_t)mv;
                    }
                    if (bb < b && pm) earlier = true;
                }
                for (uint32_t i = LB * maxk; i < S; i++) eq += ws[i] == sk[i];
                if (!earlier && eq >= min_eq) hit(id);
            }
            wave_sync();
            qcnt[lane] = 0; qfirst[lane] = kEmpty;
            qn = 0;
            wave_sync();
        };
        // ---- C ----
        for (uint32_t t0 = 0; t0 < T; t0 += 64) {
            const uint32_t t = t0 + lane;
            bool pass = false;
            uint32_t owner = 0, band = 0, id = 0;
            if (t < T) {
                uint32_t lo = 0, hi = 64;                   // last lane whose base <= t (bases of lanes without rows repeat: take the last)
                while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (lbase[mid] <= t) lo = mid; else hi = mid; }
                owner = lo;
                const uint32_t local = t - lbase[owner];
                const uint32_t *bc = bcum + owner * (LB + 1);
                while (band + 1 < LB && bc[band + 1] <= local) band++;
                const uint32_t e = blo[owner * LB + band] + (local - bc[band]);
                const uint4 *sg = reinterpret_cast<const uint4 *>(ix.band_sig + ((size_t)band * n + e) * 32);
                const uint4 sa = sg[0], sb = sg[1];
                const uint32_t ws8[8] = {sa.x, sa.y, sa.z, sa.w, sb.x, sb.y, sb.z, sb.w};
                uint32_t same = 0;
#pragma unroll
                for (uint32_t i = 0; i < 8; i++) {
                    if (i >= nd) break;
                    const uint32_t x = ws8[i] ^ rs[owner * 8 + i];
                    same += __popc((x - 0x01010101u) & ~x & 0x80808080u);
                }
                pass = same - (4u * nd - sl) + (S - sl) >= lmin[owner];
                if (pass) id = ix.band_ids[(size_t)band * n + e];
            }
            const unsigned long long pm = __ballot(pass);
            if (pass) {
                const uint32_t pos = qn + (uint32_t)__popcll(pm & ((1ULL << lane) - 1ULL));
                queue[2 * pos] = owner | (band << 8);
                queue[2 * pos + 1] = id;
                atomicAdd(&qcnt[owner], 1u);
                atomicMin(&qfirst[owner], pos);
            }
            qn += (uint32_t)__popcll(pm);
            if (qn + 64 > kLshQueue) verify();
        }
        verify();
        if (valid) seed_epilogue(a, r, o0, len, q, n_hits, min_win, s0, s1, s2, s3, high != 0, false, 0, 0, nullptr, asc, max_win);
    }
}

// K2, LSH-Forest branch in a launch of its own: lsh_lane_kernel -- a lane per read, as in sketch_seed_kernel, but without the hashing
// kernel's registers: the 21 minima are not live here (the sketch sits in HBM, written by the hashing kernel, and is read again
// only to verify a row that passed the signature filter), so kLaneRowsAhead rows are fetched together without a spill and ten
// wavefronts per SIMD hide the trips.  Same rows, same tests, same order of hits as the branch in sketch_seed_kernel
// (lshe.go:153-175); reads with more than lsh_defer_rows rows go on to lsh_heavy_kernel.
constexpr uint32_t kLaneRowsAhead = 4;
template <int NB>
__global__ __launch_bounds__(kBlock) void lsh_lane_kernel(SeedArgs a)
{
    static_assert(NB <= (int)kLshMaxBands, "bands");
    const DeviceIndex &ix = a.ix;
    const uint32_t S = ix.s, maxk = ix.max_k, LB = min(ix.l_max, (uint32_t)NB), n = ix.n_windows;
    const uint32_t sl = S < 32 ? S : 32, nd = (sl + 3) >> 2;
    const uint32_t n_list = min(*a.lsh_count, a.lsh_cap);
    for (uint32_t li = blockIdx.x * kBlock + threadIdx.x; li < n_list; li += gridDim.x * kBlock) {
        const uint32_t e0 = a.lsh_list[li];
        const uint32_t r = e0 & 0x7FFFFFFFu, high = e0 >> 31;
        const uint64_t o0 = a.seq_off[r];
        const uint32_t len = (uint32_t)(a.seq_off[r + 1] - o0);
        const uint32_t q = len - ix.k + 1;
        const uint32_t K = ix.q_k[q], L = min((uint32_t)ix.q_l[q], LB), min_eq = ix.q_min_eq[q];
        const uint64_t *sk = a.lsh_sketch + (size_t)li * S;
        uint32_t rs[8];
#pragma unroll
        for (uint32_t wd = 0; wd < 8; wd++) {
            uint32_t v = 0;
#pragma unroll
            for (uint32_t i = 0; i < 4; i++)
                if (4 * wd + i < sl) v |= sig8(sk[4 * wd + i]) << (8 * i);
            rs[wd] = v;
        }
        uint32_t b_lo[NB], b_end[NB];
        uint32_t rows = 0;
#pragma unroll
        for (uint32_t b = 0; b < (uint32_t)NB; b++) {
            b_lo[b] = n; b_end[b] = n;
            if (b >= L || K < 1) continue;
            const uint32_t *keys = ix.band_keys + (size_t)b * n * maxk;
            uint64_t hk = GROOT_SKETCH_HASH_INIT;
            for (uint32_t j = 0; j < K; j++) hk = sketch_hash_step(hk, (uint32_t)sk[b * maxk + j]);
            const ExactEntry *tab = ix.band_hash + (((size_t)b * maxk + (K - 1)) << ix.band_hash_bits);
            const uint32_t hmask = (1u << ix.band_hash_bits) - 1u, tag = (uint32_t)(hk >> 32);
            uint32_t lo = n;
            for (uint32_t slot = (uint32_t)hk & hmask;; slot = (slot + 1) & hmask) {
                const ExactEntry e = tab[slot];
                if (e.id == kEmpty) break;
                if (e.tag != tag) continue;
                bool same = true;
                for (uint32_t j = 0; j < K; j++) same &= keys[(size_t)e.id * maxk + j] == (uint32_t)sk[b * maxk + j];
                if (same) { lo = e.id; break; }
            }
            b_lo[b] = lo;
            b_end[b] = lo < n ? lo + ix.band_run[((size_t)b * maxk + (K - 1)) * n + lo] : n;
            rows += b_end[b] - lo;
        }
        if (a.heavy_list && rows > a.lsh_defer_rows && S <= kLshHeavyMaxS) {
            const uint32_t pos = atomicAdd(a.heavy_count, 1u);
            if (pos < a.lsh_cap) { a.heavy_list[pos] = li; continue; }
        }
        uint32_t n_hits = 0, min_win = kEmpty, max_win = 0, prev_id = 0;
        uint32_t s0 = kEmpty, s1 = kEmpty, s2 = kEmpty, s3 = kEmpty;
        bool asc = true;
        auto hit = [&](uint32_t id) {
            if (n_hits < a.seed_slots) a.seed_win[(size_t)n_hits * a.n_reads + r] = id;
            if (n_hits == 0) s0 = id; else if (n_hits == 1) s1 = id; else if (n_hits == 2) s2 = id; else if (n_hits == 3) s3 = id;
            asc &= n_hits == 0 || id > prev_id;
            prev_id = id;
            n_hits++;
            min_win = min(min_win, id);
            max_win = max(max_win, id);
        };
#pragma unroll
        for (uint32_t b = 0; b < (uint32_t)NB; b++) {
            if (b >= L) break;
            const uint32_t *ids = ix.band_ids + (size_t)b * n;
            const uint4 *sigs = reinterpret_cast<const uint4 *>(ix.band_sig + (size_t)b * n * 32);
            const uint32_t lo = b_lo[b], e_end = b_end[b];
            for (uint32_t e4 = lo; e4 < e_end; e4 += kLaneRowsAhead) {
                uint4 rowa[kLaneRowsAhead], rowb[kLaneRowsAhead];
#pragma unroll
                for (uint32_t i = 0; i < kLaneRowsAhead; i++) {
                    const size_t ee = min(e4 + i, e_end - 1u);
                    rowa[i] = sigs[2 * ee]; rowb[i] = sigs[2 * ee + 1];
                }
#pragma unroll
                for (uint32_t i = 0; i < kLaneRowsAhead; i++) {
                    const uint32_t e = e4 + i;
                    if (e >= e_end) break;
                    const uint4 sa = rowa[i], sb = rowb[i];
                    const uint32_t ws8[8] = {sa.x, sa.y, sa.z, sa.w, sb.x, sb.y, sb.z, sb.w};
                    uint32_t same = 0;
#pragma unroll
                    for (uint32_t x8 = 0; x8 < 8; x8++) {
                        if (x8 >= nd) break;
                        const uint32_t x = ws8[x8] ^ rs[x8];
                        same += __popc((x - 0x01010101u) & ~x & 0x80808080u);
                    }
                    if (same - (4u * nd - sl) + (S - sl) < min_eq) continue;
                    const uint32_t id = ids[e];
                    const uint64_t *ws = ix.win_sketch + (size_t)id * S;
                    uint32_t eq = 0;
                    bool earlier = false;
                    for (uint32_t bb = 0; bb < LB; bb++) {
                        bool pm = true;
                        for (uint32_t j = 0; j < maxk; j++) {
                            const uint64_t wv = ws[bb * maxk + j], mv = sk[bb * maxk + j];
                            eq += wv == mv;
                            if (j < K) pm &= (uint32_t)wv == (uint32_t)mv;
                        }
                        if (bb < b && pm) earlier = true;
                    }
                    for (uint32_t x = LB * maxk; x < S; x++) eq += ws[x] == sk[x];
                    if (!earlier && eq >= min_eq) hit(id);
                }
            }
        }
        seed_epilogue(a, r, o0, len, q, n_hits, min_win, s0, s1, s2, s3, high != 0, false, 0, 0, nullptr, asc, max_win);
    }
}

// K2, LSH-Forest branch, the heavy reads: lsh_heavy_kernel -- a WAVEFRONT per read.
// The hashing kernels look up a read's rows of equal prefix in all bands before walking any of them; a read with more than
// SeedArgs::lsh_defer_rows of them (a gene family: dozens of alleles times two dozen window offsets) is handed over with its sketch.
// Here the 64 lanes take the read's rows 64 at a time -- signature filter, then the exact count of equal slots against the sketch
// in LDS, both by the lane that holds the row -- and append the windows that pass to the read's seed slots (order of arrival: the
// align stage takes a read's windows in ascending order whatever their order in the list).  Same rows, same tests as the per-lane
// branch of sketch_seed_kernel (lshe.go:153-175).
__global__ __launch_bounds__(kBlock) void lsh_heavy_kernel(SeedArgs a)
{
    __shared__ uint64_t sk_lds[(kBlock / 64) * kLshHeavyMaxS];
    __shared__ uint32_t aux_lds[(kBlock / 64) * (2 * kLshMaxBands + 16)];
    const DeviceIndex &ix = a.ix;
    const uint32_t S = ix.s, maxk = ix.max_k, LB = ix.l_max, n = ix.n_windows;
    const uint32_t sl = S < 32 ? S : 32, nd = (sl + 3) >> 2;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t *m = sk_lds + wave * kLshHeavyMaxS;
    uint32_t *blo = aux_lds + wave * (2 * kLshMaxBands + 16);   // [LB] first row per band
    uint32_t *bcum = blo + kLshMaxBands;                        // [LB + 1] rows before band b
    uint32_t *sc = bcum + kLshMaxBands + 1;                     // [0] hits [1] min [2] max [4..7] the first four
    auto wave_sync = []() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); };
    const uint32_t n_list = a.heavy_list ? min(*a.heavy_count, a.lsh_cap) : min(*a.lsh_count, a.lsh_cap);
    for (uint32_t hi = blockIdx.x * (kBlock / 64) + wave; hi < n_list; hi += gridDim.x * (kBlock / 64)) {
        const uint32_t li = a.heavy_list ? a.heavy_list[hi] : hi;
        const uint32_t e0 = a.lsh_list[li];
        const uint32_t r = e0 & 0x7FFFFFFFu, high = e0 >> 31;
        const uint64_t o0 = a.seq_off[r];
        const uint32_t len = (uint32_t)(a.seq_off[r + 1] - o0);
        const uint32_t q = len - ix.k + 1;
        const uint32_t K = ix.q_k[q], L = min((uint32_t)ix.q_l[q], LB), min_eq = ix.q_min_eq[q];
        const uint64_t *sk = a.lsh_sketch + (size_t)li * S;
        if (lane < S) m[lane] = sk[lane];
        if (lane < 8) sc[lane] = lane == 1 ? kEmpty : 0u;
        wave_sync();
        if (lane < LB) {                                       // lane b: the rows of equal prefix in band b
            const uint32_t b = lane;
            uint32_t lo = n, run = 0;
            if (b < L && K >= 1) {
                const uint32_t *keys = ix.band_keys + (size_t)b * n * maxk;
                uint64_t hk = GROOT_SKETCH_HASH_INIT;
                for (uint32_t j = 0; j < K; j++) hk = sketch_hash_step(hk, (uint32_t)m[b * maxk + j]);
                const ExactEntry *tab = ix.band_hash + (((size_t)b * maxk + (K - 1)) << ix.band_hash_bits);
                const uint32_t hmask = (1u << ix.band_hash_bits) - 1u, tag = (uint32_t)(hk >> 32);
                for (uint32_t slot = (uint32_t)hk & hmask;; slot = (slot + 1) & hmask) {
                    const ExactEntry e = tab[slot];
                    if (e.id == kEmpty) break;
                    if (e.tag != tag) continue;
                    bool same = true;
                    for (uint32_t j = 0; j < K; j++) same &= keys[(size_t)e.id * maxk + j] == (uint32_t)m[b * maxk + j];
                    if (same) { lo = e.id; break; }
                }
                if (lo < n) run = ix.band_run[((size_t)b * maxk + (K - 1)) * n + lo];
            }
            blo[b] = lo;
            bcum[b + 1] = run;
        }
        if (lane == 0) bcum[0] = 0;
        wave_sync();
        if (lane == 0) for (uint32_t b = 0; b < LB; b++) bcum[b + 1] += bcum[b];
        wave_sync();
        const uint32_t T = bcum[LB];
        uint32_t rs[8];
#pragma unroll
        for (uint32_t wd = 0; wd < 8; wd++) {
            uint32_t v = 0;
#pragma unroll
            for (uint32_t i = 0; i < 4; i++)
                if (4 * wd + i < sl) v |= sig8(m[4 * wd + i]) << (8 * i);
            rs[wd] = v;
        }
        for (uint32_t t = lane; t < T; t += 64) {
            uint32_t b = 0;
            while (b + 1 < LB && bcum[b + 1] <= t) b++;
            const uint32_t e = blo[b] + (t - bcum[b]);
            const uint4 *sg = reinterpret_cast<const uint4 *>(ix.band_sig + ((size_t)b * n + e) * 32);
            const uint4 sa = sg[0], sb = sg[1];
            const uint32_t ws8[8] = {sa.x, sa.y, sa.z, sa.w, sb.x, sb.y, sb.z, sb.w};
            uint32_t same = 0;
#pragma unroll
            for (uint32_t i = 0; i < 8; i++) {
                if (i >= nd) break;
                const uint32_t x = ws8[i] ^ rs[i];
                same += __popc((x - 0x01010101u) & ~x & 0x80808080u);
            }
            if (same - (4u * nd - sl) + (S - sl) < min_eq) continue;
            const uint32_t id = ix.band_ids[(size_t)b * n + e];
            const uint64_t *ws = ix.win_sketch + (size_t)id * S;
            uint32_t eq = 0;
            bool earlier = false;
            for (uint32_t bb = 0; bb < LB; bb++) {
                bool pm = true;
                for (uint32_t j = 0; j < maxk; j++) {
                    const uint64_t wv = ws[bb * maxk + j], mv = m[bb * maxk + j];
                    eq += wv == mv;
                    if (j < K) pm &= (uint32_t)wv == (uint32_t)mv;
                }
                if (bb < b && pm) earlier = true;
            }
            for (uint32_t i = LB * maxk; i < S; i++) eq += ws[i] == m[i];
            if (earlier || eq < min_eq) continue;
            const uint32_t pos = atomicAdd(&sc[0], 1u);
            if (pos < a.seed_slots) a.seed_win[(size_t)pos * a.n_reads + r] = id;
            if (pos < 4) sc[4 + pos] = id;
            atomicMin(&sc[1], id);
            atomicMax(&sc[2], id);
        }
        wave_sync();
        if (lane == 0) seed_epilogue(a, r, o0, len, q, sc[0], sc[1], sc[0] > 0 ? sc[4] : kEmpty, sc[0] > 1 ? sc[5] : kEmpty, sc[0] > 2 ? sc[6] : kEmpty,
                                     sc[0] > 3 ? sc[7] : kEmpty, high != 0, false, 0, 0, nullptr, false, sc[2]);
        wave_sync();
    }
}

// fills the text table at open: string j (tw dwords at 2 bits per base, then its bytes other than ACGT: device_types.hpp
// text_exc_dwords) with a non-zero sig_info word claims the first free slot of its probe sequence (compare-and-swap on the entry's
// info word) and writes tag, bases and exceptions; hashed over the bases, twk dwords of them, as the lookup does
__global__ __launch_bounds__(kBlock) void text_table_fill_kernel(const uint32_t *__restrict__ words, const uint32_t *__restrict__ info, uint32_t n, uint32_t tw,
                                                                 uint32_t stride, uint32_t twk, uint32_t *tab, uint32_t mask)
{
    const uint32_t j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= n || !info[j]) return;
    const uint32_t *wd = words + (size_t)j * stride;        // tw dwords of bases, then stride - tw dwords of bytes other than ACGT
    uint64_t h = GROOT_TEXT_HASH_INIT;
    for (uint32_t x = 0; x < twk; x++) h = text_hash_step(h, x < tw ? wd[x] : 0u);
    for (uint32_t slot = (uint32_t)h & mask;; slot = (slot + 1) & mask) {
        uint32_t *e = tab + (size_t)slot * 16;
        if (atomicCAS(e + 1, 0u, info[j]) != 0u) continue;
        e[0] = (uint32_t)(h >> 32);
        for (uint32_t x = 0; x < stride; x++) e[2 + x] = wd[x];
        return;
    }
}

template <int TW>
__global__ __launch_bounds__(kBlock) void text_lookup_kernel(SeedArgs a)
{
    static_assert(TW >= 1 && TW <= 14, "a 64-byte entry holds 224 bases");
    constexpr int XW = (int)text_exc_dwords(TW);           // dwords of (position, byte) pairs behind the bases
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // LDS: one bit per 4 bases of the span (set: a byte other than ACGT among them), then one dword of codes per 16 bases
    uint32_t *badbits = reinterpret_cast<uint32_t *>(smem);
    uint32_t *codes = reinterpret_cast<uint32_t *>(smem + kTextBad);
    const DeviceIndex &ix = a.ix;
    const unsigned tid = threadIdx.x;
    for (uint32_t i = tid; i < kTextBad / 4; i += kBlock) badbits[i] = 0;
    // ---- stage this block's reads as 2-bit codes (as sketch_sig_kernel does): one contiguous span, 16 bases per lane per load ----
    const uint32_t r0 = blockIdx.x * kBlock;
    const uint32_t r_end = min(r0 + (uint32_t)kBlock, a.n_reads);
    const uint64_t span0 = a.seq_off[r0], span1 = a.seq_off[r_end];
    const uint64_t base16 = span0 & ~15ULL;
    const uint64_t span_bytes = span1 - base16;
    const bool in_lds = span_bytes <= a.lds_read_bytes;
    __syncthreads();
    if (in_lds) {
        const uint4 *src = reinterpret_cast<const uint4 *>(a.seq + base16);
        const uint32_t n16 = (uint32_t)((span_bytes + 15) >> 4);
        for (uint32_t i = tid; i < n16; i += kBlock) {
            const uint4 v = src[i];
            uint32_t b0 = 0, b1 = 0, b2 = 0, b3 = 0;
            const uint32_t c = codes_of4(v.x, b0) | (codes_of4(v.y, b1) << 8) | (codes_of4(v.z, b2) << 16) | (codes_of4(v.w, b3) << 24);
            codes[i] = c;
            const uint32_t bad = (b0 ? 1u : 0u) | (b1 ? 2u : 0u) | (b2 ? 4u : 0u) | (b3 ? 8u : 0u);
            if (bad) atomicOr(&badbits[i >> 3], bad << (4 * (i & 7)));
        }
    }
    __syncthreads();
    const uint32_t r = r0 + tid;
    const bool valid = r < a.n_reads;
    uint32_t info = 0;
    uint32_t len = 0;
    bool no_seeds = false;
    if (valid) {
    const uint64_t o0 = a.seq_off[r];
    len = (uint32_t)(a.seq_off[r + 1] - o0);
    {
        // more k-mers than Containment > t allows at any number of equal slots (reads well beyond the window size): the query cannot
        // return a window whatever the sketch is -- answered here, as both hashing kernels do, instead of travelling through the list
        const uint32_t q = len - ix.k + 1;
        if (len >= ix.k && len <= a.max_read_len && ix.max_q && (q > ix.max_q || ix.q_min_eq[q] > ix.s)) {
            seed_epilogue(a, r, o0, len, q, 0, kEmpty, kEmpty, kEmpty, kEmpty, kEmpty, false);
            no_seeds = true;
        }
    }
    bool mine = !no_seeds && in_lds && len == ix.w;
    bool exc = false;
    if (mine) {
        // groups of 4 bases the read touches: a byte other than ACGT in one of them?
        const uint32_t c0 = (uint32_t)(o0 - base16) >> 2, c1 = (uint32_t)(o0 - base16 + len - 1) >> 2;
        for (uint32_t w = c0 >> 5; w <= c1 >> 5; w++) {
            uint32_t bits = badbits[w];
            if (w == c0 >> 5) bits &= ~0u << (c0 & 31);
            if (w == c1 >> 5) bits &= ~0u >> (31 - (c1 & 31));
            if (bits) exc = true;
        }
        if (exc && XW == 0) mine = false;
    }
    if (mine) {
    const uint32_t P = 2u * (uint32_t)(o0 - base16);       // bit position of base 0 in `codes`
    const uint32_t n_full = len >> 4, tail_mask = (1u << (2 * (len & 15))) - 1u;
    uint32_t rdw[TW];
    uint32_t xdw[XW ? XW : 1] = {};
#pragma unroll
    for (int j = 0; j < TW; j++) {
        const uint32_t mask = (uint32_t)j < n_full ? ~0u : ((uint32_t)j == n_full ? tail_mask : 0u);
        rdw[j] = __builtin_amdgcn_alignbit(codes[(P >> 5) + j + 1], codes[(P >> 5) + j], P & 31) & mask;
    }
    if (XW != 0 && exc) {
        // a group of 4 bases with a byte other than ACGT in it (a read in thousands): the bytes themselves, from the read in HBM --
        // position and byte go into the key as they sit in the entry, the 2-bit code of the position is 0
        const uint32_t rel0 = (uint32_t)(o0 - base16);
        const uint32_t c0 = rel0 >> 2, c1 = (rel0 + len - 1) >> 2;
        uint32_t np = 0;
        for (uint32_t g = c0; g <= c1 && mine; g++) {
            if (!((badbits[g >> 5] >> (g & 31)) & 1u)) continue;
            const uint32_t v = *reinterpret_cast<const uint32_t *>(a.seq + base16 + 4ull * g);
            for (uint32_t j = 0; j < 4; j++) {
                const uint32_t b = (v >> (8 * j)) & 0xFFu;
                const uint32_t at = 4 * g + j;
                if (at < rel0 || at >= rel0 + len || b == 'A' || b == 'C' || b == 'G' || b == 'T') continue;
                if (np >= 2u * XW) { mine = false; break; }
                const uint32_t pos = at - rel0;
                const uint32_t pair = ((pos + 1) << 8) | b;
#pragma unroll
                for (int x = 0; x < XW; x++) if ((np >> 1) == (uint32_t)x) xdw[x] |= pair << (16 * (np & 1));
#pragma unroll
                for (int x = 0; x < TW; x++) if ((pos >> 4) == (uint32_t)x) rdw[x] &= ~(3u << (2 * (pos & 15)));
                np++;
            }
        }
    }
    if (mine) {
    uint64_t h = GROOT_TEXT_HASH_INIT;
#pragma unroll
    for (int j = 0; j < TW; j++) h = text_hash_step(h, rdw[j]);
    const uint32_t tag = (uint32_t)(h >> 32);
    const uint4 *tab = ix.text_tab;
    for (uint32_t slot = (uint32_t)h & ix.text_mask;; slot = (slot + 1) & ix.text_mask) {
        const uint4 *e = tab + (size_t)slot * 4;
        constexpr int NQ = (2 + TW + XW + 3) / 4;          // 16-byte words of an entry that hold something
        uint32_t ed[4 * NQ];
#pragma unroll
        for (int i = 0; i < NQ; i++) {
            const uint4 v = e[i];
            ed[4 * i] = v.x; ed[4 * i + 1] = v.y; ed[4 * i + 2] = v.z; ed[4 * i + 3] = v.w;
        }
        // the whole entry in ONE round trip (left alone the compiler loads the tag, tests it, and only then fetches the text)
#pragma unroll
        for (int i = 0; i < 4 * NQ; i++) asm volatile("" : "+v"(ed[i]));
        if (ed[1] == 0) break;                             // free slot: the string is not in the table
        if (ed[0] != tag) continue;
        uint32_t diff = 0;
#pragma unroll
        for (int j = 0; j < TW; j++) diff |= ed[2 + j] ^ rdw[j];
#pragma unroll
        for (int j = 0; j < XW; j++) diff |= ed[2 + TW + j] ^ xdw[j];
        if (!diff) { info = ed[1]; break; }
    }
    }
    }
    }
    // ---- the reads this kernel leaves to the full-width kernel, as a list: counted per workgroup (ballots, one LDS atomic per
    // wavefront), ONE global atomic per workgroup that has any.  (One per wavefront on a single counter cost 1.1 ms per 10 M reads
    // when every wavefront had a miss; a separate stream compaction of per-read marks 0.1 ms.)
    __shared__ uint32_t blk_cnt, blk_base;
    if (tid == 0) blk_cnt = 0;
    __syncthreads();
    const bool miss = valid && !info && !no_seeds;
    const unsigned long long mb = __ballot(miss);
    uint32_t wave_base = 0;
    if ((tid & 63) == 0 && mb) wave_base = atomicAdd(&blk_cnt, (uint32_t)__popcll(mb));
    wave_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_base);
    __syncthreads();
    if (tid == 0 && blk_cnt) blk_base = atomicAdd(a.todo_count, blk_cnt);
    __syncthreads();
    if (miss) a.todo_list[blk_base + wave_base + (uint32_t)__popcll(mb & ((1ULL << (tid & 63)) - 1ULL))] = r;
    if (!valid || !info || no_seeds) return;
    // the read's whole outcome is tabulated; order_first_kernel writes its records and its call counts from the table
    const uint32_t q = len - ix.k + 1;
    if (a.q_seen && ix.q_row[q] == kEmpty) a.q_seen[q] = 1u;
    a.sort_key[r] = kEmpty;
    a.tab_idx[r] = (info & ((1u << kOutIdxBits) - 1u)) | kTabSeedsHere;
    a.trav_cnt[r] = tab_travs(ix, info);
}

// groot_hip_submit_packed: 2 bits per base back to ASCII in HBM (code (byte >> 1) & 3: A=0 C=1 T=2 G=3), 16 bases per
// thread (one 4-byte load, one 16-byte store); bytes other than ACGT are patched in from the exception list afterwards
__global__ __launch_bounds__(kBlock) void unpack_reads_kernel(const uint32_t *__restrict__ packed, uint64_t n_words, uint4 *__restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_words) return;
    const uint32_t w = packed[i];
    uint32_t o[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        uint32_t v = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const uint32_t code = (w >> (8 * q + 2 * b)) & 3u;
            v |= ((0x47544341u >> (8 * code)) & 0xFFu) << (8 * b);   // "ACTG"[code]
        }
        o[q] = v;
    }
    out[i] = make_uint4(o[0], o[1], o[2], o[3]);
}
__global__ __launch_bounds__(kBlock) void patch_reads_kernel(const uint64_t *__restrict__ pos, const uint8_t *__restrict__ byte, uint64_t n,
                                                             uint8_t *__restrict__ seq)
{
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) seq[pos[i]] = byte[i];
}

// The align stage handles a read's seed windows in ascending order (graphminion.go:46-102 ranges over them in the canonical order
// of the windows), one after the other in ONE lane: a read of a sequence that two hundred graphs share keeps its lane -- and the
// launch -- busy for two hundred walks (resfinder.90, reads of 75..150 bases: 0.8 % of the reads bring more than 16 windows; the
// align stage takes 4.7 ms per 2 M reads with them and 2.7 ms without).  Such reads are rare and prepared here, a wavefront per
// read: the list sorted (rank by counting, in LDS; LSH-Forest hits come in band order), then cut at graph boundaries into items
// of at least kSplitMin windows that different lanes of align_kernel take (AlignArgs::vitem).
struct SplitArgs {
    const uint32_t *list, *count;          // SeedArgs::long_list
    const uint32_t *seed_count;
    uint32_t *seed_win;
    uint32_t n_reads, seed_slots;
    ReadRec *read_rec;
    const WinRec *win_rec;
    uint32_t split;                        // 0: sort only (capture pass of groot_hip_open, no_exact_align)
    uint4 *vitem;                          // [vcap]
    uint32_t *vcount;                      // [0] items, [1] split reads
    uint32_t vcap;
    uint4 *split_list;                     // [kLongListCap] {read, first item, items, -}
    DeviceCounters *ctr;
    uint32_t update_weights;
};
__global__ __launch_bounds__(kBlock) void sort_seed_lists_kernel(SplitArgs a)
{
    __shared__ uint32_t lds[(kBlock / 64) * (2 * kSortSeedsMax + kSplitMaxItems + 4)];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t *raw = lds + wave * (2 * kSortSeedsMax + kSplitMaxItems + 4);   // the list as found; then the graph of every sorted window
    uint32_t *sorted = raw + kSortSeedsMax;
    uint32_t *seg = sorted + kSortSeedsMax;                                   // [kSplitMaxItems] end positions, then [0..3] scalars
    auto wave_sync = []() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); };
    const uint32_t n = min(*a.count, kLongListCap);
    for (uint32_t i = blockIdx.x * (kBlock / 64) + wave; i < n; i += gridDim.x * (kBlock / 64)) {
        const uint32_t r = a.list[i];
        const uint32_t full = a.seed_count[r] & 0x7FFFFFFFu;
        const uint32_t cnt = min(full, a.seed_slots);
        if (full > a.seed_slots || cnt > kSortSeedsMax) continue;        // (more seeds than slots: the batch is redone anyway)
        for (uint32_t j = lane; j < cnt; j += 64) raw[j] = a.seed_win[(size_t)j * a.n_reads + r];
        wave_sync();
        for (uint32_t j = lane; j < cnt; j += 64) {
            const uint32_t v = raw[j];
            uint32_t rank = 0;
            for (uint32_t x = 0; x < cnt; x++) { const uint32_t o = raw[x]; rank += (o < v || (o == v && x < j)) ? 1u : 0u; }
            sorted[rank] = v;
            a.seed_win[(size_t)rank * a.n_reads + r] = v;
        }
        wave_sync();
        uint32_t flags = kRecAscending;
        if (a.split && cnt > kSplitMin) {
            for (uint32_t j = lane; j < cnt; j += 64) raw[j] = a.win_rec[sorted[j]].graph;
            wave_sync();
            if (lane == 0) {
                const uint32_t target = max(kSplitMin, (cnt + kSplitMaxItems - 1) / kSplitMaxItems);
                uint32_t lo = 0, x = 0, ns = 0, graphs = 0;
                while (x < cnt) {
                    const uint32_t g = raw[x];
                    while (x < cnt && raw[x] == g) x++;
                    graphs++;
                    if (x - lo >= target || x == cnt) { seg[ns++] = x; lo = x; }
                }
                // (a short last item joins the one before it)
                if (ns > 1 && seg[ns - 1] - seg[ns - 2] < kSplitMin / 2) { seg[ns - 2] = seg[ns - 1]; ns--; }
                uint32_t j0 = kEmpty;
                if (ns > 1) {
                    j0 = atomicAdd(&a.vcount[0], ns - 1);
                    if (j0 > a.vcap || ns - 1 > a.vcap - j0) {                 // no room: the slots taken stay empty, the read stays whole
                        for (uint32_t j = j0; j < min(j0 + ns - 1, a.vcap); j++) a.vitem[j] = make_uint4(kEmpty, 0, 0, 0);
                        j0 = kEmpty;
                    }
                }
                seg[kSplitMaxItems] = ns; seg[kSplitMaxItems + 1] = j0; seg[kSplitMaxItems + 2] = graphs;
            }
            wave_sync();
            const uint32_t ns = seg[kSplitMaxItems], j0 = seg[kSplitMaxItems + 1];
            if (ns > 1 && j0 != kEmpty) {
                for (uint32_t k = 1 + lane; k < ns; k += 64) a.vitem[j0 + k - 1] = make_uint4(r, seg[k - 1], seg[k], 0);
                if (lane == 0) {
                    const uint32_t si = atomicAdd(&a.vcount[1], 1u);      // (at most one per entry of the list: si < kLongListCap)
                    a.split_list[si] = make_uint4(r, j0, ns - 1, 0);
                    if (a.update_weights && seg[kSplitMaxItems + 2] > 1) atomicAdd(&a.ctr->multimapped, 1ull);   // boss.go:195-200
                    uint32_t &cf = a.read_rec[r].cnt_flags;
                    cf = (cf & ~kRecCountMask) | seg[0] | kRecSplit;
                }
            }
            wave_sync();
        }
        if (lane == 0) a.read_rec[r].cnt_flags |= flags;
    }
}

// after the align stage: the records of a split read's items follow each other in the read's (read, ord) run -- every item learns
// how many records the read's earlier items made, the read's count becomes the sum
__global__ __launch_bounds__(kBlock) void split_fix_kernel(const uint4 *__restrict__ split_list, const uint32_t *__restrict__ vcount, uint4 *__restrict__ vitem,
                                                           uint32_t *__restrict__ trav_cnt, uint32_t n_reads, DeviceCounters *ctr)
{
    const uint32_t ns = min(vcount[1], kLongListCap);
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < ns; i += gridDim.x * kBlock) {
        const uint4 sl = split_list[i];
        uint32_t base = trav_cnt[sl.x];
        for (uint32_t j = sl.y; j < sl.y + sl.z; j++) {
            vitem[j].w = base;
            base += trav_cnt[n_reads + j];
        }
        if (base > 0xFFFFu) atomicOr(&ctr->flags, kFlagOrdOverflow);
        trav_cnt[sl.x] = base;
    }
}

// ... and the first record of every item goes to its place (the later ones: order_ovf_kernel)
__global__ __launch_bounds__(kBlock) void order_split_kernel(const uint4 *__restrict__ vitem, const uint32_t *__restrict__ vcount, uint32_t vcap,
                                                             const uint32_t *__restrict__ trav_cnt, const uint32_t *__restrict__ off, const groot_trav *__restrict__ first,
                                                             const uint64_t *__restrict__ mask_first, uint32_t n_reads, uint32_t first_read_id, groot_trav *__restrict__ out,
                                                             uint64_t *__restrict__ mask_out, uint32_t cap, uint32_t pw_in, uint32_t pw_out, DeviceCounters *ctr)
{
    const uint32_t nv = min(vcount[0], vcap);
    for (uint32_t j = blockIdx.x * kBlock + threadIdx.x; j < nv; j += gridDim.x * kBlock) {
        const uint4 vi = vitem[j];
        if (vi.x == kEmpty || trav_cnt[n_reads + j] == 0) continue;
        const uint32_t i = off[vi.x] + vi.w;
        if (i >= cap) { atomicOr(&ctr->flags, kFlagTravOverflow); continue; }
        groot_trav t = first[n_reads + j];
        t.read_id = first_read_id + vi.x;
        t.ord = (uint16_t)vi.w;
        out[i] = t;
        for (uint32_t w = 0; w < pw_out; w++) mask_out[(size_t)i * pw_out + w] = mask_first[(size_t)(n_reads + j) * pw_in + w];
    }
}

// One row of the call-count table per kmerCount that occurs among seeded reads (IncrementSubPath's numKmers,
// graphminion.go:60-67): rows are handed out in ascending kmerCount order within a batch, after the seed stage and before
// the align stage.  More kmerCounts than rows: kFlagQOverflow, the align stage does nothing, the host grows the table
// and re-runs the batch.
__global__ __launch_bounds__(64) void assign_q_rows_kernel(uint32_t *q_seen, uint32_t *q_row, uint32_t *q_of_row, uint32_t *n_rows, uint32_t cap,
                                                            uint32_t max_q, DeviceCounters *ctr, unsigned long long *shards, uint32_t *long_count)
{
    // one wavefront: lane i folds shard i of the seed kernels' counters, then the kmerCounts are taken 64 at a time
    if (blockIdx.x) return;
    if (!threadIdx.x) *long_count = 0;                      // (sort_seed_lists_kernel ran just before: ready for the next batch)
    const uint32_t lane = threadIdx.x;
    {
        unsigned long long seeds = 0, most = 0, seeded = 0, tabbed = 0;
        for (uint32_t i = lane; i < kSeedShards; i += 64) {
            unsigned long long *sh = shards + (size_t)i * kSeedShardStride;
            seeds += sh[0];
            most = sh[1] > most ? sh[1] : most;
            seeded += sh[2];
            tabbed += sh[3];
            sh[0] = 0; sh[1] = 0; sh[2] = 0; sh[3] = 0;
        }
        for (int o = 32; o; o >>= 1) {
            seeds += __shfl_xor(seeds, o);
            seeded += __shfl_xor(seeded, o);
            tabbed += __shfl_xor(tabbed, o);
            const unsigned long long other = __shfl_xor(most, o);
            most = other > most ? other : most;
        }
        if (!lane) {
            ctr->seeds += seeds;
            ctr->seeded_reads += (unsigned int)seeded;
            ctr->tab_reads += (unsigned int)tabbed;
            if (most > ctr->max_seeds) ctr->max_seeds = (unsigned int)most;
        }
    }
    uint32_t need = *n_rows;                               // (every lane reads the same value; lane 0 writes it back at the end)
    for (uint32_t q0 = 0; q0 <= max_q; q0 += 64) {
        const uint32_t q = q0 + lane;
        bool wants = false;
        if (q <= max_q && q_seen[q]) {
            q_seen[q] = 0;
            wants = q_row[q] == kEmpty;
        }
        const unsigned long long b = __ballot(wants);       // rows are handed out in ascending kmerCount order
        if (wants) {
            const uint32_t row = need + __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
            if (row < cap) { q_row[q] = row; q_of_row[row] = q; }
        }
        need += (uint32_t)__popcll(b);
    }
    if (!lane) {
        if (need > cap) atomicOr(&ctr->flags, kFlagQOverflow);
        *n_rows = min(need, cap);
        ctr->q_rows = need;
    }
}

// offsets of a batch whose reads all have the same length (then no length array travels): off[i] = i * len, i in [0, n]
__global__ __launch_bounds__(kBlock) void uniform_offsets_kernel(uint64_t *__restrict__ off, uint32_t n, uint32_t len)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i <= n) off[i] = (uint64_t)i * len;
}

// ---------------------------------------------------------------------------------------------
// K3
// ---------------------------------------------------------------------------------------------
// seqio.go:17-23 complementBases: anything but ACGTN becomes 0 (never equals a graph base)
__device__ __forceinline__ unsigned comp_base(unsigned b)
{
    switch (b) {
    case 'A': return 'T';
    case 'T': return 'A';
    case 'C': return 'G';
    case 'G': return 'C';
    case 'N': return 'N';
    default: return 0;
    }
}

// sum v over the workgroup; result valid in thread 0
__device__ __forceinline__ unsigned long long block_sum(unsigned long long v, unsigned long long *lds4)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const unsigned wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) lds4[wave] = v;
    __syncthreads();
    return lds4[0] + lds4[1] + lds4[2] + lds4[3];
}

// K3 as a per-lane state machine with wave-coherent phase scheduling.
//   graphMinion loop (graphminion.go:46-102) -> AlignRead hierarchy (alignment.go:13-159)
//   -> performAlignment / dfsRecursive / processTraversal (alignment.go:162-317)
// Each lane is in one of three phases and advances by ONE step when its phase is executed:
//   FETCH  take the next read (its record, in processing order; bases staged in the lane's LDS slice) / pick the
//          read's next seed window, count IncrementSubPath, apply the seed stage's verdicts on hierarchy levels
//   SCAN   test up to 16 candidate start offsets of one node against the read prefix (SWAR, 4-base
//          filter then exact 8-base check of the lowest survivor); walks the hierarchy levels 1..4
//   DFS    match up to 32 bases of one graph node, choose the next neighbour, emit / backtrack
// Per iteration the wave executes only the phase holding the most lanes (ballot + popcount in SALU), so
// lanes in different reads / levels / depths never serialise each other's loops and every executed
// instruction runs at the best available lane fill.  A wavefront takes 64 consecutive reads of the processing order
// at a time (they share a seed window, hence the graph nodes they walk) and asks for more when all lanes are done.
enum : uint32_t { PH_FETCH, PH_SCAN, PH_DFS, PH_WAIT, PH_DONE };
#ifndef GROOT_REFILL
#define GROOT_REFILL 64
#endif
constexpr int kRefill = GROOT_REFILL;          // waiting lanes that trigger a refill
#ifndef GROOT_WAVE_CHUNK
#define GROOT_WAVE_CHUNK 128
#endif
constexpr uint32_t kWaveChunk = GROOT_WAVE_CHUNK;   // consecutive slots a wave takes before asking for more (multiple of 64)
#ifndef GROOT_SMALL_CHUNK_SHARE
#define GROOT_SMALL_CHUNK_SHARE(n) ((n) >> 3)
#endif

// 0x80 in byte j iff byte j of x equals c, or is the 'N' wildcard
__device__ __forceinline__ uint64_t match_or_n(uint64_t x, unsigned c)
{
    return ~(nonzero_bytes(x ^ (kOnes * c)) & nonzero_bytes(x ^ (kOnes * 'N'))) & kHi1;
}
// 0x80 in the low n bytes (n may exceed 8 or be <= 0)
__device__ __forceinline__ uint64_t low_bytes(int n) { return n <= 0 ? 0 : (n >= 8 ? kHi1 : (kHi1 >> (8 * (8 - n)))); }

// a NodeRec held in registers as dwords (16-byte loads; every access below uses a constant index)
template <int PW> struct RecRegs {
    static constexpr int NQ = (int)(sizeof(NodeRec<PW>) / 16);
    uint32_t d[NQ * 4];
    __device__ __forceinline__ void load(const NodeRec<PW> *rp)
    {
        const uint4 *q = reinterpret_cast<const uint4 *>(rp);
#pragma unroll
        for (int i = 0; i < NQ; i++) {
            const uint4 v = q[i];
            d[4 * i] = v.x; d[4 * i + 1] = v.y; d[4 * i + 2] = v.z; d[4 * i + 3] = v.w;
        }
        // The whole record is wanted NOW, in one round trip.  Left to itself the compiler sinks field loads into the
        // branches that use them (seq_off after the length test, first8 folded into a pointer select with the bases
        // load), which turns one DFS step into three dependent trips to L2.
#pragma unroll
        for (int i = 0; i < NQ * 4; i++) asm volatile("" : "+v"(d[i]));
    }
    __device__ __forceinline__ uint32_t seq_off() const { return d[0]; }
    __device__ __forceinline__ uint32_t seq_len() const { return d[1]; }
    __device__ __forceinline__ uint32_t deg() const { return d[2] & 0x7FFFFFFFu; }
    __device__ __forceinline__ bool wild() const { return (d[2] >> 31) != 0; }     // the node holds an 'N'
    __device__ __forceinline__ unsigned child_first(int e) const { return (d[3] >> (8 * e)) & 0xFFu; }
    __device__ __forceinline__ uint64_t first8() const { return (uint64_t)d[4] | ((uint64_t)d[5] << 32); }
    __device__ __forceinline__ uint32_t edge(int e) const { return d[6 + e]; }
    __device__ __forceinline__ uint64_t mask(int i) const { return (uint64_t)d[10 + 2 * i] | ((uint64_t)d[11 + 2 * i] << 32); }
};

// LDSR: the oriented read of every lane is staged in LDS when its first DFS of that orientation starts
// (lane-private slice of lds_stride_dw dwords, odd stride = conflict-free across lanes); DFS steps then read
// their 8-base chunks with three ds_read_b32 + two alignbit instead of going back to the Infinity Cache /
// HBM for the read's line and re-doing the reverse complement at every step.
#ifndef GROOT_ALIGN_WAVES
#define GROOT_ALIGN_WAVES 4   // 4 waves/SIMD = at most 128 VGPRs (5 or 6 spill and are slower; 3 hide too little latency)
#endif
#ifndef GROOT_ALIGN_WAVES_WIDE
#define GROOT_ALIGN_WAVES_WIDE 3   // 704-bit path sets: at 4 waves (128 VGPRs) the kernel spills 260 bytes per lane and is 40 % slower (tools/wide_probe.py)
#endif
template <int PW, bool LDSR>
__global__ __launch_bounds__(kBlock, PW > 3 ? GROOT_ALIGN_WAVES_WIDE : GROOT_ALIGN_WAVES) void align_kernel(AlignArgs a)
{
    using Rec = NodeRec<PW>;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_reads[];
    __shared__ unsigned long long red[4];
    uint32_t *my_lds = lds_reads + (size_t)threadIdx.x * a.lds_stride_dw;
    const DeviceIndex &ix = a.ix;
    const Rec *recs = reinterpret_cast<const Rec *>(a.node_rec);
    const uint32_t gtid = blockIdx.x * kBlock + threadIdx.x;
    // the seed stage ran out of per-read slots (or the call-count table out of rows): the host grows them and re-runs the whole batch
    if (a.ctr->flags & (kFlagSeedOverflow | kFlagQOverflow)) return;
    unsigned long long alns = 0, mapped = 0, multimapped = 0, panics = 0;
#ifdef GROOT_WORK_COUNTERS
    uint32_t ev = 0;                                       // events of this lane in the current wave iteration
    uint32_t wc_iter = 0, wc_round0 = 0;                   // wave iterations so far / at the last refill
    unsigned long long wc_t[3] = {0, 0, 0};                // wall-clock ticks (100 MHz) per phase, wave-uniform
    uint32_t wc_n[3] = {0, 0, 0};                          // steps per phase
#define GROOT_EV(i) (ev |= 1u << (i))
#else
#define GROOT_EV(i) ((void)0)
#endif

    uint32_t phase = PH_WAIT;
    // reads are handed out per wavefront: chunks of kWaveChunk consecutive (sorted) slots, round-robin over the
    // waves of the grid, consecutive slots to the lanes that ask together
    // (a wavefront that runs out takes the next chunk from a global cursor: no static shares, so no wave idles while
    // another still holds several chunks)
    // Rounds of 64 slots are handed out through two cursors.  The first eighth of the order holds the longest walks (one
    // round of them can take a quarter of the launch): cursor 0 hands those out one round at a time; once it has run past
    // them, cursor 1 hands out the rest kWaveChunk slots at a time.  (Only atomics touch the cursors: an atomic LOAD at
    // agent scope in this loop halves the kernel's speed.)
    // reads without seeds sort last and have nothing to do here (the seed stage zeroed their traversal counts)
    // (items of split reads come first: slot j < nv is AlignArgs::vitem[j], slot nv + i is position i of the processing order)
    const uint32_t nv = a.vitem ? min((uint32_t)__builtin_amdgcn_readfirstlane((int)*a.vcount), a.vcap) : 0u;
    const uint32_t n_todo = nv + (a.perm ? min(a.n_reads, (uint32_t)__builtin_amdgcn_readfirstlane((int)a.ctr->seeded_reads)) : a.n_reads);   // (scalar: it bounds every refill)
    // Lanes per round.  A round lasts as long as its slowest read, so when there are fewer reads than 64 per resident wavefront
    // (most of the batch was answered from the outcome table: what is left are the hard reads) the rounds are made smaller
    // and spread over all wavefronts: the launch then ends with the slowest read instead of the slowest sum of rounds.
    uint32_t U = 64;
    if (a.round_lanes) U = a.round_lanes;
    else
        while (U > 1u && n_todo < U * (gridDim.x * (uint32_t)(kBlock / 64))) U >>= 1;
    const uint32_t n_rounds = (n_todo + U - 1u) / U;
    // (odd on purpose: with an even count the two-round chunks behind the head start at multiples of 128 slots and the kernel is
    // 6 % slower -- measured both ways, cause not established)
    const uint32_t head_rounds = GROOT_SMALL_CHUNK_SHARE(n_rounds) | 1u;
    // The head of the order holds the longest walks.  When the reads of a batch do not march in step (a.head_lanes != 0: mixed
    // read lengths) a round of 64 of them lasts as long as their steps laid end to end -- one such round was a quarter of the
    // launch --, so the head is handed out in rounds of a.head_lanes reads; the tail keeps full rounds.
    const uint32_t Uh = a.head_lanes ? min(a.head_lanes, U) : U;
    const uint32_t head_slots = min(head_rounds * U, n_todo);
    const uint32_t head_small = (head_slots + Uh - 1u) / Uh;
    uint32_t chunk_len = 0, chunk_base = 0;                // slots in the current chunk, its first slot (wave-uniform)
    bool head_done = head_small == 0;                      // wave-uniform
    auto take_chunk = [&]() {
        uint32_t c = 0, tail = 0;
        if ((threadIdx.x & 63) == 0) {
            if (!head_done) c = atomicAdd(a.ovf_cnt + kOvfShards, 1u);
            if (head_done || c >= head_small) {
                tail = 1;
                c = atomicAdd(a.ovf_cnt + kOvfShards + 1, kWaveChunk / 64u);
            }
        }
        c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
        tail = (uint32_t)__builtin_amdgcn_readfirstlane((int)tail);
        if (tail) {
            head_done = true;
            // (saturating: past the end the base only has to be >= n_todo)
            const unsigned long long b = (unsigned long long)head_slots + (unsigned long long)c * U;
            chunk_base = b > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)b;
            chunk_len = (kWaveChunk / 64u) * U;
        } else {
            chunk_base = c * Uh;
            chunk_len = min(Uh, head_slots - chunk_base);
        }
    };
    take_chunk();
    uint32_t chunk_pos = 0;                                // slots of the chunk already handed out (wave-uniform)
    uint32_t slot = 0, r = 0;
    // ---- read ----
    bool have_read = false;
    const uint8_t *p = nullptr;
    uint32_t len = 0, cnt = 0, qrow = 0, n_graphs = 0, ord = 0, read_id = 0;
    uint32_t sd0 = kEmpty, sd1 = kEmpty, sd2 = kEmpty, sd3 = kEmpty;   // the read's first four seed windows
    uint32_t high_byte = 0;                                // RevComplement would panic on this read
    uint32_t cls = 0;                                      // kRec* verdicts of the read record >> 24; bit 6: they apply to w
    long long last = -1;                                   // last seed window handled (ascending window id order)
    uint32_t done_graph = kEmpty, cur_graph = kEmpty;
    bool group_rc_called = false;
    // ---- seed / hierarchy ----
    uint32_t w = 0, g = 0, seed = 0, seed_s0 = 0, seed_len = 0, off0 = 0, l1_hi = 0, cn_cur = 0, cn_begin = 0, cn_end = 0;
    uint32_t rc = 0, level = 1;
    uint32_t sc_node = 0, sc_s0 = 0, sc_len = 0, sc_pos = 0, sc_end = 0;   // range being scanned
    uint32_t clip_lo = 0, eff = 0, tflags = 0;
    uint64_t pre8 = 0;
    // ---- DFS ----
    uint32_t node0 = 0, noff0 = 0, cur = 0, coff = 0, dist = 0, sp = 0, emitted = 0;
    uint64_t cur8 = 0;                                     // oriented read bases [dist, dist+8)
    uint64_t mask[PW];
#pragma unroll
    for (int i = 0; i < PW; i++) mask[i] = 0;

    // oriented bases [d, d+8) of the current view during DFS
    // LDSR: the lane's slice holds 8 zero bytes, then the read as it came (forward), staged when the read was fetched.
    // Oriented bases [i, i+8) are slice bytes [8+i, 16+i) forward, or the reverse complement of slice bytes [len-i, len-i+8).
    auto dfs_chunk = [&](uint32_t d) -> uint64_t {
        if (!LDSR) return read_chunk(p, len, rc, clip_lo, d);
        const uint32_t i = d + clip_lo;
        const uint32_t o = rc ? len - i : 8u + i;
        const uint32_t *wp = my_lds + (o >> 2);
        const uint32_t x0 = wp[0], x1 = wp[1], x2 = wp[2];
        const uint32_t sh = (o & 3u) * 8u;
        const uint64_t v = (uint64_t)__funnelshift_r(x0, x1, sh) | ((uint64_t)__funnelshift_r(x1, x2, sh) << 32);
        return rc ? revcomp8(v) : v;
    };
    auto set_view = [&](uint32_t clip_lo_, uint32_t eff_, uint32_t clip_flag) {
        clip_lo = clip_lo_; eff = eff_;
        tflags = (rc ? GROOT_TRAV_RC : 0u) | clip_flag;
        pre8 = dfs_chunk(0);
    };
    auto scan_range = [&](uint32_t node, uint32_t s0, uint32_t nlen, uint32_t from, uint32_t to) {
        sc_node = node; sc_s0 = s0; sc_len = nlen; sc_pos = from; sc_end = to;
    };
    // verdict of the seed stage for the current orientation (f = kRecNo12F / kRecNo3F / kRecNo4F)
    auto verdict = [&](uint32_t f) -> bool { return (cls & 0x40u) && ((cls >> (rc ? 3 : 0)) & (f >> 24)); };
    // 1. seed offset shuffling (alignment.go:34-45).  Returns true when levels 1 and 2 cannot start anywhere for this
    // orientation (prefix tables): the ranges are left empty and the caller moves on through the hierarchy.
    auto start_orientation = [&](uint32_t t) -> bool {
        rc = t; level = 1;
        set_view(0, len, 0);
        scan_range(seed, seed_s0, seed_len, off0, l1_hi);
        phase = PH_SCAN;
        bool no;
        if (cls & 0x40u) no = verdict(kRecNo12F);
        else no = prefix_absent(ix.win_prefix + (size_t)w * kPrefixWords, pre8, eff >= 12 ? dfs_chunk(8) : 0, eff);
        if (no) { level = 2; cn_cur = cn_end; sc_pos = sc_end = 0; }
        return no;
    };
    // the current scan range is used up: move through the hierarchy until a non-empty range or the end
    auto next_range = [&]() {
        for (;;) {
            if (level == 1) { level = 2; cn_cur = cn_begin; }
            else if (level == 2) cn_cur++;
            else if (level == 3) {
                level = 4;                                  // 4. hard clip the last base (:87-103)
                if (verdict(kRecNo4F)) continue;            // its single start position fails the first comparison
                set_view(0, len - 1, GROOT_TRAV_END_CLIP);
                scan_range(seed, seed_s0, seed_len, off0, off0 + 1);
                return;
            } else {
                // AlignRead found nothing in this orientation: graphminion.go:94 RevComplement
                if (!group_rc_called) {                     // first RevComplement of this minion's copy of the read
                    group_rc_called = true;
                    if (high_byte) panics++;                 // seqio.go:126 index out of range
                }
                if (rc == 0) {
                    if (start_orientation(1)) continue;
                } else phase = PH_FETCH;                     // both orientations failed: next mapping
                return;
            }
            if (level == 2) {                               // 2. seed node shuffling (:47-70): offsets 0..10
                // Contained nodes none of whose offsets 0..10 can spell the first four read bases would each cost a SCAN step
                // that finds nothing (its 4-base filter is the same test): DeviceIndex::node_pre4 says so per node, four nodes
                // per pair of trips.  A read that fails everywhere walks every contained node of every seed window in both
                // orientations -- it is the slowest read of its batch, and the launch lasts as long as it does.
                if (ix.node_pre4 && eff >= 4u && cn_cur < cn_end) {
                    const int c4 = kmer4_code(pre8);
                    if (c4 >= 0) {
                        const uint32_t wi = (uint32_t)c4 >> 5, bi = (uint32_t)c4 & 31u;
                        while (cn_cur < cn_end) {
                            const uint32_t left = cn_end - cn_cur;
                            const uint32_t n0 = ix.cn_node[cn_cur], n1 = left > 1 ? ix.cn_node[cn_cur + 1] : n0, n2 = left > 2 ? ix.cn_node[cn_cur + 2] : n0,
                                           n3 = left > 3 ? ix.cn_node[cn_cur + 3] : n0;
                            const uint32_t w0 = ix.node_pre4[(size_t)n0 * 8 + wi], w1 = ix.node_pre4[(size_t)n1 * 8 + wi], w2 = ix.node_pre4[(size_t)n2 * 8 + wi],
                                           w3 = ix.node_pre4[(size_t)n3 * 8 + wi];
                            uint32_t hit = 4;
                            if (left > 3 && ((w3 >> bi) & 1u)) hit = 3;
                            if (left > 2 && ((w2 >> bi) & 1u)) hit = 2;
                            if (left > 1 && ((w1 >> bi) & 1u)) hit = 1;
                            if ((w0 >> bi) & 1u) hit = 0;
                            cn_cur += min(hit, left);
                            if (hit < 4) break;
                        }
                    }
                }
                if (cn_cur < cn_end) {
                    const uint32_t node = ix.cn_node[cn_cur];
                    const uint32_t nlen = recs[node].seq_len;
                    scan_range(node, recs[node].seq_off, nlen, 0, min(nlen, 11u));
                    return;
                }
                level = 3;                                  // 3. hard clip the first base (:72-85)
                if (off0 >= seed_len) { level = 4; continue; }   // :199-201 holds for levels 3 and 4 alike
                if (verdict(kRecNo3F)) continue;
                set_view(1, len - 1, GROOT_TRAV_START_CLIP);
                scan_range(seed, seed_s0, seed_len, off0, off0 + 1);
                return;
            }
        }
    };

    for (;;) {
#if defined(GROOT_WORK_COUNTERS) && GROOT_WORK_COUNTERS != 2     // (2 = phase timing only: the tally itself costs time)
        for (int e = 0; e < 32; e++) {                         // convergent point: tally the previous iteration
            const unsigned long long b = __ballot((ev >> e) & 1u);
            if (b && (threadIdx.x & 63) == 0) {
                atomicAdd(&a.ctr->dbg[e], 1ull);
                atomicAdd(&a.ctr->dbg[32 + e], (unsigned long long)__popcll(b));
            }
        }
        ev = 0;
#endif
#ifdef GROOT_WORK_COUNTERS
        wc_iter++;
#endif
        // ---- run the phase holding the most lanes (wave-uniform; ballots and popcounts are SALU) ----
        const unsigned long long bf = __ballot(phase == PH_FETCH), bs = __ballot(phase == PH_SCAN), bd = __ballot(phase == PH_DFS);
        {
            // Lanes that finished their read wait until kRefill of them have gathered (or nothing else is left to
            // run), then take the next consecutive slots together.  Reads are sorted by (first seed window,
            // orientation class), so lanes that start together do near-identical work and share phases.
            const unsigned long long bw = __ballot(phase == PH_WAIT);
            const int cw = __popcll(bw);
            const uint32_t Uc = max(1u, min(chunk_len, min(U, 64u)));   // lanes a round of the current chunk fills
            if (cw >= (int)((64u - Uc) + max(1u, a.refill * Uc / 64u)) || (cw && !(bf | bs | bd))) {
#if defined(GROOT_WORK_COUNTERS) && GROOT_WORK_COUNTERS != 2
                if ((threadIdx.x & 63) == 0 && wc_iter > 1) atomicAdd(&a.ctr->dbg[128 + min(63u, (wc_iter - wc_round0) / 2)], 1ull);   // round length
                if ((threadIdx.x & 63) == 0) atomicMax(&a.ctr->dbg[63], (unsigned long long)(wc_iter - wc_round0));          // longest round
                wc_round0 = wc_iter;
#elif defined(GROOT_WORK_COUNTERS)
                wc_round0 = wc_iter;
#endif
                const uint64_t base = chunk_base;
                if (base >= n_todo) {                          // this wave's share is used up
                    if (phase == PH_WAIT) phase = PH_DONE;
                } else {
                    const uint32_t room = chunk_len - chunk_pos;
                    if (phase == PH_WAIT) {
                        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(bw >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bw, 0u));
                        const uint64_t sl = base + chunk_pos + rank;
                        if (rank < room) {
                            if (sl < n_todo) { slot = (uint32_t)sl; phase = PH_FETCH; GROOT_EV(20); }
                            else phase = PH_DONE;
                        }
                    }
                    chunk_pos += min((uint32_t)cw, room);
                    if (chunk_pos >= chunk_len) { chunk_pos = 0; take_chunk(); }
                }
                continue;
            }
            if (!(bf | bs | bd)) break;                         // no lane has work and none waits
        }
        const int cf = __popcll(bf), cs = __popcll(bs), cd = __popcll(bd);
        const uint32_t run = (cd >= cs && cd >= cf) ? PH_DFS : (cs >= cf ? PH_SCAN : PH_FETCH);
        if (phase != run) continue;
        GROOT_EV(run);                                          // events 0,1,2: a step of FETCH / SCAN / DFS
#ifdef GROOT_WORK_COUNTERS
        const unsigned long long wc_t0 = wall_clock64();
        const uint32_t wc_steps0 = wc_iter;
#endif
        bool advance = false;                                   // leave the current scan range (one call site: the code is large)

        if (run == PH_FETCH) {
#if defined(GROOT_WORK_COUNTERS) && GROOT_WORK_COUNTERS == 2
#define GROOT_SUBT(i) do { const unsigned long long t__ = wall_clock64(); if ((threadIdx.x & 63) == (__ffsll((unsigned long long)__ballot(1)) - 1)) atomicAdd(&a.ctr->dbg[40 + (i)], t__ - wc_sub); wc_sub = t__; } while (0)
            unsigned long long wc_sub = wall_clock64();
#else
#define GROOT_SUBT(i) ((void)0)
#endif
            if (!have_read) {
                GROOT_EV(3);
                const bool virt = slot < nv;                   // an item of a split read: seed positions [vlo, vhi) of its ascending list
                uint32_t vlo = 0, vhi = 0;
                if (virt) {
                    const uint4 vi = a.vitem[slot];
                    r = vi.x; vlo = vi.y; vhi = vi.z;
                    if (r == kEmpty) { phase = PH_WAIT; continue; }   // (found no room: its read is handled whole)
                } else {
                    const uint32_t so = slot - nv;
                    r = a.perm ? a.perm[so] : so;              // reads in (first seed window, orientation) order
                }
                uint4 ra, rb;                                     // one 32-byte record per read
                load32(a.read_rec + r, ra, rb);                  // (gathering the records into processing order first costs more than this dependent trip)
                const uint32_t sc = ra.w;
                cnt = min(sc & (kRecSplit - 1u), a.seed_slots);   // overflow already flagged; batch is re-run
                cls = a.perm ? (sc >> 24) & 0x3Fu : 0x80u;     // bit 7: no verdicts without the seed stage's sort keys
                if (sc & kRecAscending) cls |= 0x100u;         // bit 8: the read's seed list is in ascending window order
                // bit 9: the windows come from the list, not from the read record; bit 10: an item (mapped / multimapped are counted
                // with the read's first item); bit 11: a split read (multimapped was counted when it was split)
                if (sc & kRecSplit) cls |= 0xA00u;
                if (virt) { cnt = min(vhi, a.seed_slots); cls = 0x80u | 0x100u | 0x200u | 0x400u; }
                if (cnt == 0) { a.trav_cnt[r] = 0; phase = PH_WAIT; continue; }
                high_byte = sc >> 31;
                len = ra.z;
                p = a.seq + ((uint64_t)ra.x | ((uint64_t)ra.y << 32));
                sd0 = rb.x; sd1 = rb.y; sd2 = rb.z; sd3 = rb.w;
                if (cls & 0x200u) sd0 = vlo;                   // (ascending list: sd0 is the position in it; else sd0 / sd1 = smallest / largest window)
                else if (cnt > 4 && (cls & 0x100u)) sd0 = 0;
                GROOT_SUBT(0);
                if (LDSR && 2 + 4 * ((len + 27) >> 4) > a.lds_stride_dw) {   // longer than the max_len the batch was submitted with
                    atomicOr(&a.ctr->flags, kFlagLongRead);
                    a.trav_cnt[(cls & 0x400u) ? a.n_reads + slot : r] = 0;
                    phase = PH_WAIT;
                    continue;
                }
                qrow = ix.q_row[len - ix.k + 1];              // graphminion.go:60 kmerCount -> its row of the call-count table
                read_id = a.first_read_id + ((cls & 0x400u) ? a.n_reads + slot : r);   // (an item labels its records with its own slot: order_ovf_kernel)
                n_graphs = 0; ord = 0; last = -1;
                done_graph = kEmpty; cur_graph = kEmpty; group_rc_called = false;
                have_read = true;
                if (LDSR) {                                   // stage the read: 64 bytes per pass, the four loads in flight together
                    my_lds[0] = 0; my_lds[1] = 0;
                    for (uint32_t b = 0; b < len; b += 64) {       // reads at most 15 bytes past the read's end
                        const uint4 *src = reinterpret_cast<const uint4 *>(p + b);   // unaligned 16-byte global loads
                        const bool h1 = b + 16 < len, h2 = b + 32 < len, h3 = b + 48 < len;
                        uint4 v0, v1 = {}, v2 = {}, v3 = {};
                        __builtin_memcpy(&v0, src, 16);
                        if (h1) __builtin_memcpy(&v1, src + 1, 16);
                        if (h2) __builtin_memcpy(&v2, src + 2, 16);
                        if (h3) __builtin_memcpy(&v3, src + 3, 16);
                        uint32_t *d = my_lds + 2 + (b >> 2);
                        d[0] = v0.x; d[1] = v0.y; d[2] = v0.z; d[3] = v0.w;
                        if (h1) { d[4] = v1.x; d[5] = v1.y; d[6] = v1.z; d[7] = v1.w; }
                        if (h2) { d[8] = v2.x; d[9] = v2.y; d[10] = v2.z; d[11] = v2.w; }
                        if (h3) { d[12] = v3.x; d[13] = v3.y; d[14] = v3.z; d[15] = v3.w; }
                    }
                }
            }
            GROOT_SUBT(1);
            // seeds in canonical order = ascending window id (graph, Node, OffSet, list position)
            uint32_t nw = kEmpty;
            if (cnt <= 4 && !(cls & 0x200u)) {                // the seeds travel in the read record
                if ((long long)sd0 > last && sd0 < nw) nw = sd0;
                if (cnt > 1 && (long long)sd1 > last && sd1 < nw) nw = sd1;
                if (cnt > 2 && (long long)sd2 > last && sd2 < nw) nw = sd2;
                if (cnt > 3 && (long long)sd3 > last && sd3 < nw) nw = sd3;
            } else if (cls & 0x100u) {
                // An ascending list is walked, not searched (a read of a sequence that many graphs share brings a hundred seed
                // windows: looking through all of them for every one of them made it the slowest read of its batch by far).
                // sd0 = first position not handled yet; after a graph is done `last` has jumped past its windows: bisect.
                uint32_t lo = sd0;
                uint32_t cand = lo < cnt ? a.seed_win[(size_t)lo * a.n_reads + r] : kEmpty;
                if (lo < cnt && (long long)cand <= last) {
                    uint32_t hi = cnt;
                    lo++;
                    while (lo < hi) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if ((long long)a.seed_win[(size_t)mid * a.n_reads + r] > last) hi = mid; else lo = mid + 1;
                    }
                    cand = lo < cnt ? a.seed_win[(size_t)lo * a.n_reads + r] : kEmpty;
                }
                if (lo < cnt) nw = cand;
                sd0 = lo + 1;
            } else if (last < 0) nw = sd0;                    // the smallest window, from the read record
            else if ((long long)sd1 > last)                   // (else nothing is left: no look at the list)
                for (uint32_t j = 0; j < cnt; j++) {
                    const uint32_t cand = a.seed_win[(size_t)j * a.n_reads + r];
                    if ((long long)cand > last && cand < nw) nw = cand;
                }
            if (nw == kEmpty) {                               // every seed of the read handled
                GROOT_EV(4);
#if defined(GROOT_WORK_COUNTERS) && GROOT_WORK_COUNTERS != 2
                atomicAdd(&a.ctr->dbg[64 + min(63u, (wc_iter - wc_round0) / 2)], 1ull);   // when in its round the lane finished
#endif
#if defined(GROOT_WORK_COUNTERS) && GROOT_WORK_COUNTERS == 2
                if (wc_iter - wc_round0 >= 100) {              // slow reads, by name (meaningful with GROOT_ROUND_LANES=1)
                    const unsigned long long sl = atomicAdd(&a.ctr->dbg[128], 1ull);
                    if (sl < 60) a.ctr->dbg[129 + sl] = (unsigned long long)r | ((unsigned long long)(wc_iter - wc_round0) << 32);
                }
#endif
                a.trav_cnt[(cls & 0x400u) ? a.n_reads + slot : r] = ord;
                if (!(cls & 0x400u)) mapped++;                // boss.go:195-200
                if (!(cls & 0xC00u) && n_graphs > 1) multimapped++;
                if (a.incr_cnt && n_graphs > 1) a.incr_cnt[r] |= 0x80000000u;   // (capture pass of groot_hip_open; the lane owns the read)
                have_read = false;
                phase = PH_WAIT;
                continue;
            }
            cls = (cls & ~0x40u) | ((last < 0 && !(cls & 0x80u)) ? 0x40u : 0u);   // bit 6: w is the read's first seed window
            w = nw; last = nw;
            uint4 wa, wb;                                     // the whole lshe.Key in one 32-byte load
            load32(ix.win_rec + w, wa, wb);
            g = wa.x;
            GROOT_SUBT(2);
            if (g != cur_graph) { cur_graph = g; n_graphs++; group_rc_called = false; }
            if (g == done_graph) continue;                    // graphminion.go:96-98: stop after the first alignment
            if (a.update_weights) {                            // :67 IncrementSubPath
                // neighbouring lanes mostly hold reads of the same window (processing order): one atomic per distinct cell
                // among the lanes that are here together instead of one per lane (0.44 of 3.15 ms per 10 M reads)
                const uint64_t cell = (uint64_t)qrow * ix.n_windows + w;
                for (bool pending = true; pending;) {
                    const uint64_t first = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)cell) |
                                           ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(cell >> 32)) << 32);
                    const unsigned long long same = __ballot(cell == first);
                    if (cell == first) {
                        if (__builtin_amdgcn_mbcnt_hi((uint32_t)(same >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)same, 0u)) == 0)
                            atomicAdd(&a.attempts[first], (uint32_t)__popcll(same));
                        pending = false;
                    }
                }
            }
            GROOT_SUBT(3);
            if (a.incr_cnt) {              // capture pass: which windows had IncrementSubPath called, in call order
                const uint32_t n = a.incr_cnt[r];
                a.incr_cnt[r] = n + 1;
                if (n < a.incr_cap) a.incr_win[(size_t)r * a.incr_cap + n] = w;
            }
            if (a.no_align) continue;                         // :70-72
            seed = wa.y; off0 = wa.z;
            l1_hi = wa.w;                                     // alignment.go:36 and :199-201, folded at open
            cn_begin = wb.x; cn_end = wb.y;
            seed_s0 = wb.z; seed_len = wb.w;
            GROOT_EV(5);
            advance = start_orientation(0);
            GROOT_SUBT(4);
        } else if (run == PH_SCAN) {
            if (sc_pos >= sc_end) { GROOT_EV(6); advance = true; }   // only after a DFS that used the range's last offset
            else {
            // up to 16 start offsets sc_pos.. of node sc_node: which can spell the first bases of the read?
            const uint8_t *gb = ix.bases + sc_s0 + sc_pos;
            const uint64_t w0 = ld8(gb), w1 = ld8(gb + 8), w2 = ld8(gb + 16);
            const uint32_t npos = min(16u, sc_end - sc_pos);
            const int room = (int)(sc_len - sc_pos);          // bases from sc_pos to the node end
            uint64_t c_lo = low_bytes((int)npos), c_hi = low_bytes((int)npos - 8);
            const uint32_t kf = min(4u, eff);
#pragma unroll
            for (int b = 0; b < 4; b++) {
                if ((uint32_t)b >= kf) break;
                const unsigned rb = (unsigned)(pre8 >> (8 * b)) & 0xFF;
                // positions whose base b lies inside the node must match it; past the node end the DFS decides
                const uint64_t need_lo = low_bytes(room - b), need_hi = low_bytes(room - b - 8);
                c_lo &= match_or_n(window8(w0, w1, b), rb) | ~need_lo;
                c_hi &= match_or_n(window8(w1, w2, b), rb) | ~need_hi;
            }
            uint32_t j = 16;
            if (c_lo) j = (uint32_t)__builtin_ctzll(c_lo) >> 3;
            else if (c_hi) j = 8 + ((uint32_t)__builtin_ctzll(c_hi) >> 3);
            if (j >= npos) {
                GROOT_EV(7);
                sc_pos += npos;
                advance = sc_pos >= sc_end;                   // set up the next range in this step: no empty one
            } else {
                // exact 8-base check of the lowest survivor (alignment.go:203-223 would fail here otherwise)
                const uint64_t g8 = j < 8 ? window8(w0, w1, j) : window8(w1, w2, j - 8);
                const uint32_t off = sc_pos + j;
                sc_pos = off + 1;
                if (!prefix_ok(g8, pre8, min(min(sc_len - off, eff), 8u))) {
                    GROOT_EV(8);
                    advance = sc_pos >= sc_end;
                } else {
                    node0 = sc_node; noff0 = off; cur = sc_node; coff = off; dist = 0; sp = 0; emitted = 0;
                    cur8 = pre8;
#pragma unroll
                    for (int i = 0; i < PW; i++) mask[i] = ~0ULL;
                    phase = PH_DFS;
                    GROOT_EV(10);
                }
            }
            }
        } else {
            // ---- DFS: match up to 32 bases of node `cur` from offset coff (dfsRecursive, alignment.go:203-223) ----
            // The lanes stay in here for as long as the scheduling rule above would pick the phase again (lanes only leave
            // it for FETCH or SCAN, both counted below), which saves the ballots and the refill logic per step.
            int nd, nf, ns;
            do {
            if (phase == PH_DFS) {
            RecRegs<PW> rec;
            rec.load(recs + cur);
            // The common step, on its own: a whole short node (<= 8 bases, from its first base, no 'N') matches, the read goes on,
            // some path is left and exactly one neighbour can take the next base.  Everything is in the record: no graph bases, no
            // stack, nothing to report.  Whatever does not fit falls through to the general step below, state untouched; a
            // wavefront whose lanes all fit skips that code altogether (it is most of this kernel's instructions).
            // (Letting the lanes that fit run ahead, step after step, while the others wait is slower: 3.08 vs 2.53 ms -- a step
            // is a trip to L2 whatever it computes, and the general step hides some of it.)
            bool fast_done = false;
            {
                const uint32_t take = min(rec.seq_len(), eff - dist);
                const uint32_t rdeg = rec.deg();
                if (coff == 0 && take >= 1 && take <= 8 && take == rec.seq_len() && dist + take < eff && !rec.wild() && rdeg >= 1 && rdeg <= 4 &&
                    prefix_eq(rec.first8(), cur8, take)) {
                    uint64_t nm[PW];
                    bool any = false;
#pragma unroll
                    for (int i = 0; i < PW; i++) { nm[i] = mask[i] & rec.mask(i); any |= nm[i] != 0; }
                    const uint64_t c8 = dfs_chunk(dist + take);
                    const unsigned nextb = (unsigned)c8 & 0xFF;
                    uint32_t hits = 0, pick = 0;
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const unsigned cf1 = rec.child_first(e);
                        const bool m = (uint32_t)e < rdeg && (cf1 == 'N' || cf1 == nextb);
                        hits += m;
                        if (m) pick = rec.edge(e);
                    }
                    if (any && hits == 1) {
#pragma unroll
                        for (int i = 0; i < PW; i++) mask[i] = nm[i];
                        dist += take; cur8 = c8; cur = pick; coff = 0;
                        fast_done = true;
                    }
                }
            }
            if (!fast_done) {
            if (coff != 0 && !(cur == node0 && coff == noff0 && dist == 0)) GROOT_EV(18);
            const uint32_t take = min(rec.seq_len() - coff, eff - dist);
            const uint32_t nb = min(take, 32u);
            bool ok = true;
            if (nb) {
                const uint8_t *gb = ix.bases + rec.seq_off() + coff;
                const uint64_t ga = coff == 0 ? rec.first8() : ld8(gb);
                ok = prefix_ok(ga, cur8, min(nb, 8u));
                for (uint32_t i = 8; ok && i < nb; i += 8)
                    ok = prefix_ok(ld8(gb + i), dfs_chunk(dist + i), nb - i);
            }
            bool backtrack = !ok;
            if (!ok) GROOT_EV(12);
            if (take > 8) GROOT_EV(11);
            if (take > 32) GROOT_EV(21);
            if (ok) {
                dist += nb; coff += nb;
                cur8 = dfs_chunk(dist);
                if (nb == take) {                              // node consumed (or read finished)
                    GROOT_EV(13);
                    bool any = false;
#pragma unroll
                    for (int i = 0; i < PW; i++) { mask[i] &= rec.mask(i); any |= mask[i] != 0; }
                    const uint32_t rdeg = rec.deg();
                    if (dist == eff || rdeg == 0) {             // :229-236 report the traversal
                        if (any) {
                            GROOT_EV(14);
                            if (ord) GROOT_EV(19);
                            groot_trav t;
                            t.read_id = read_id; t.graph_id = g; t.node = node0; t.offset = noff0;
                            t.ord = (uint16_t)ord;
                            t.flags = (uint8_t)(tflags | (emitted == 0 ? GROOT_TRAV_FIRST : 0));
                            t.reserved = 0;
                            if (ord == 0) {                    // the common case: no allocation at all
                                const uint32_t os = (cls & 0x400u) ? a.n_reads + slot : r;
                                a.trav_first[os] = t;
#pragma unroll
                                for (int i = 0; i < PW; i++) a.mask_first[(size_t)os * PW + i] = mask[i];
                            } else {
                                const uint32_t shard = blockIdx.x & (kOvfShards - 1);
                                const uint32_t slot = atomicAdd(&a.ovf_cnt[shard], 1u);
                                if (slot < a.ovf_cap) {
                                    const size_t o = (size_t)shard * a.ovf_cap + slot;
                                    a.ovf_trav[o] = t;
#pragma unroll
                                    for (int i = 0; i < PW; i++) a.ovf_mask[o * PW + i] = mask[i];
                                } else atomicOr(&a.ctr->flags, kFlagOvfOverflow);
                            }
                            if (ord >= 0xFFFFu) atomicOr(&a.ctr->flags, kFlagOrdOverflow);
                            ord++;
#pragma unroll
                            for (int i = 0; i < PW; i++) alns += __popcll(mask[i]);
                            emitted++;
                        }
                        backtrack = true;
                    } else if (!any) backtrack = true;         // no path left: descendants cannot yield ids
                    else {
                        // :242-252 neighbours in OutEdges order; a neighbour whose first base cannot match the
                        // next read base dies in its first comparison, so it is skipped without being visited
                        const unsigned nextb = (unsigned)cur8 & 0xFF;
                        uint32_t first = kEmpty, more = kEmpty;
                        if (rdeg <= 4) {
#pragma unroll
                            for (int e = 3; e >= 0; e--) {
                                const unsigned cf1 = rec.child_first(e);
                                if ((uint32_t)e < rdeg && (cf1 == 'N' || cf1 == nextb)) { more = first; first = e; }
                            }
                        } else { first = 0; more = 1; }
                        if (first == kEmpty) backtrack = true;
                        else {
                            if (more != kEmpty) {              // further candidates stay pending
                                GROOT_EV(15);
                                const size_t si = (size_t)sp * a.n_threads + gtid;
                                a.stk_hdr[si] = (uint64_t)cur | ((uint64_t)more << 32) | ((uint64_t)dist << 48);
#pragma unroll
                                for (int i = 0; i < PW; i++) a.stk_mask[si * PW + i] = mask[i];
                                sp++;
                            }
                            if (rdeg <= 4) {                   // select, not index: keeps the record in registers
                                cur = rec.edge(0);
                                if (first == 1) cur = rec.edge(1);
                                if (first == 2) cur = rec.edge(2);
                                if (first == 3) cur = rec.edge(3);
                            } else cur = ix.edges[rec.edge(0) + first];
                            coff = 0;
                        }
                    }
                }
            }
            if (backtrack) {
                GROOT_EV(16);
                if (sp == 0) {                                 // performAlignment is over
                    if (emitted) {                             // alignment found for (read, graph)
                        done_graph = g; phase = PH_FETCH;
                        // graphminion.go:96-98 passes over the graph's other seeds: they are the windows up to the graph's last one
                        // (a read below the window size can bring a hundred of them: one FETCH step instead of one each)
                        if (ix.graph_win_end) last = (long long)ix.graph_win_end[g] - 1;
                    }
                    else phase = PH_SCAN;
                } else {                                       // resume at the newest pending neighbour
                    GROOT_EV(17);
                    const size_t si = (size_t)(sp - 1) * a.n_threads + gtid;
                    const uint64_t hdr = a.stk_hdr[si];
                    const uint32_t pn = (uint32_t)hdr, e = (uint32_t)(hdr >> 32) & 0xFFFFu;
                    dist = (uint32_t)(hdr >> 48);
#pragma unroll
                    for (int i = 0; i < PW; i++) mask[i] = a.stk_mask[si * PW + i];
                    cur8 = dfs_chunk(dist);
                    RecRegs<PW> pr;
                    pr.load(recs + pn);
                    const uint32_t deg = pr.deg();
                    uint32_t more = kEmpty;
                    if (deg <= 4) {
                        const unsigned nextb = (unsigned)cur8 & 0xFF;
#pragma unroll
                        for (int e2 = 3; e2 >= 1; e2--) {
                            const unsigned cf1 = pr.child_first(e2);
                            if ((uint32_t)e2 > e && (uint32_t)e2 < deg && (cf1 == 'N' || cf1 == nextb)) more = e2;
                        }
                        cur = pr.edge(0);
                        if (e == 1) cur = pr.edge(1);
                        if (e == 2) cur = pr.edge(2);
                        if (e == 3) cur = pr.edge(3);
                    } else {
                        if (e + 1 < deg) more = e + 1;
                        cur = ix.edges[pr.edge(0) + e];
                    }
                    coff = 0;
                    if (more == kEmpty) sp--;
                    else a.stk_hdr[si] = (uint64_t)pn | ((uint64_t)more << 32) | ((uint64_t)dist << 48);
                }
            }
            }   // general step
            }
#ifdef GROOT_WORK_COUNTERS
            wc_iter++;                                         // (events of the steps inside this loop are merged)
#endif
            nd = __popcll(__ballot(phase == PH_DFS));
            nf = cf + __popcll(__ballot(phase == PH_FETCH));
            ns = cs + __popcll(__ballot(phase == PH_SCAN));
            } while (nd > 0 && nd >= ns && nd >= nf);
        }
        if (advance) next_range();
#ifdef GROOT_WORK_COUNTERS
        {   // wall-clock ticks (100 MHz) and steps of this phase execution (wave-uniform values)
            const unsigned long long dt = wall_clock64() - wc_t0;
            const uint32_t st = run == PH_DFS ? wc_iter - wc_steps0 : 1u;
            if (run == PH_FETCH) { wc_t[0] += dt; wc_n[0] += st; } else if (run == PH_SCAN) { wc_t[1] += dt; wc_n[1] += st; } else { wc_t[2] += dt; wc_n[2] += st; }
        }
#endif
    }

#ifdef GROOT_WORK_COUNTERS
    if ((threadIdx.x & 63) == 0)
        for (int i = 0; i < 3; i++) { atomicAdd(&a.ctr->dbg[24 + i], wc_t[i]); atomicAdd(&a.ctr->dbg[27 + i], (unsigned long long)wc_n[i]); }
#endif
    alns = block_sum(alns, red);
    mapped = block_sum(mapped, red);
    multimapped = block_sum(multimapped, red);
    panics = block_sum(panics, red);
    if (threadIdx.x == 0) {
        if (alns) atomicAdd(&a.ctr->alignments, alns);
        if (a.update_weights) {
            if (mapped) atomicAdd(&a.ctr->mapped, mapped);
            if (multimapped) atomicAdd(&a.ctr->multimapped, multimapped);
            if (panics) atomicAdd(&a.ctr->revcomp_panics, panics);
        }
    }
}

// ordered traversal records -> the 12-byte form the copy-out sends (20 B -> 12 B per record over PCIe)
__global__ __launch_bounds__(kBlock) void trav_pack_kernel(const groot_trav *__restrict__ in, const DeviceCounters *__restrict__ ctr, uint32_t cap,
                                                           uint32_t first_read_id, groot_ctrav *__restrict__ out)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= cap || i >= ctr->n_trav) return;
    const groot_trav t = in[i];
    out[i] = groot_ctrav{t.node, t.offset, ((t.read_id - first_read_id) & 0x00FFFFFFu) | ((uint32_t)t.flags << 24)};
}

// dst += src over n uint32 (call-count tables of ctxs that share a device, groot_hip_attempts_allreduce)
__global__ __launch_bounds__(kBlock) void add_u32_kernel(uint32_t *__restrict__ dst, const uint32_t *__restrict__ src, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) dst[i] += src[i];
}

// the seed stage's histogram of IncrementSubPath calls of tabulated reads (SeedArgs::tab_hist) -> the row of their kmerCount in the
// call-count table; after assign_q_rows_kernel, when the row exists and the batch's overflow flags are known.  Zeroes the histogram.
__global__ __launch_bounds__(kBlock) void fold_tab_hist_kernel(uint32_t *__restrict__ hist, uint32_t *__restrict__ attempts, const uint32_t *__restrict__ q_row,
                                                               uint32_t q_tab, uint32_t n_windows, const DeviceCounters *ctr, uint32_t update_weights)
{
    const bool live = update_weights && !(ctr->flags & (kFlagSeedOverflow | kFlagQOverflow));
    const uint32_t row = live ? q_row[q_tab] : kEmpty;
    for (uint32_t w = blockIdx.x * kBlock + threadIdx.x; w < n_windows; w += gridDim.x * kBlock) {
        const uint32_t v = hist[w];
        if (!v) continue;
        hist[w] = 0;
        if (row != kEmpty) attempts[(size_t)row * n_windows + w] += v;
    }
}

// ---- ordering: (read, ord) order without a sort -------------------------------------------------
// off = exclusive scan of trav_cnt (rocprim); record (r, ord) lands at off[r] + ord.
__global__ void order_total_kernel(const uint32_t *off, const uint32_t *cnt, uint32_t n, DeviceCounters *ctr)
{
    if (n) ctr->n_trav = off[n - 1] + cnt[n - 1];
}

// tab_idx[r] != kEmpty: the read's records come from the outcome table (DeviceIndex::out_tab) instead of the align stage -- cnt[r]
// entries from tab_idx[r] on --, and this kernel does what the align stage does for the others: the IncrementSubPath call counts
// (graphminion.go:60-67) and the read / alignment counters (boss.go:195-200).
struct OrderTabArgs {
    uint32_t *seed_count, *seed_win;   // [n], [slots][n]: written here for the reads text_lookup_kernel answered (kTabSeedsHere)
    uint32_t seed_slots, exp;
    const uint32_t *tab_idx;     // [n] or null
    const uint4 *out_tab;
    uint32_t stride_q, first_read_id;
    uint32_t update_weights;
    uint32_t *attempts;          // [rows][n_windows]
    const uint32_t *q_row;
    uint32_t q_tab, n_windows;   // kmerCount of the tabulated reads (WindowSize - k + 1)
};
__global__ __launch_bounds__(kBlock) void order_first_kernel(const groot_trav *first, const uint64_t *mask_first, const uint32_t *off,
                                                           const uint32_t *cnt, uint32_t n, groot_trav *out, uint64_t *mask_out,
                                                           uint32_t cap, uint32_t pw_in, uint32_t pw_out, DeviceCounters *ctr, OrderTabArgs t)
{
    __shared__ unsigned long long red[4];
    unsigned long long alns = 0, mapped = 0, multimapped = 0, seeds = 0;
    // (a pass whose seed stage ran out of slots / table rows is repeated as a whole: nothing may be counted in it)
    const bool live = !(ctr->flags & (kFlagSeedOverflow | kFlagQOverflow));
    // (grid-stride: the three counter atomics per workgroup below share one line, ~7 ns each -- a few thousand workgroups, not 40 000)
    for (uint32_t r = blockIdx.x * kBlock + threadIdx.x; r < n; r += gridDim.x * kBlock) {
    const uint32_t tw = t.tab_idx ? t.tab_idx[r] : kEmpty;
    const uint32_t ti = tw == kEmpty ? kEmpty : tw & ((1u << kOutIdxBits) - 1u);
    const bool count_here = (t.update_weights & 1u) && !(tw & kTabCounted) && !(t.exp & 1u);   // (else the seed stage counted the read's calls)
    const bool seeds_here = tw != kEmpty && (tw & kTabSeedsHere) && !(t.exp & 2u);              // its seed windows are the table's call-count windows
    uint32_t ns = 0;
    if (ti != kEmpty && live) {
        const uint32_t nt = cnt[r], i = off[r];
        const bool fits = i < cap && nt <= cap - i;
        if (!fits) atomicOr(&ctr->flags, kFlagTravOverflow);
        const uint32_t row = t.update_weights ? t.q_row[t.q_tab] : 0u;
        uint32_t n_ent = nt ? nt : 1u;                        // (a string without traversals still has its calls, seeds and counters)
        for (uint32_t j = 0; j < n_ent; j++) {
            const uint4 *e = t.out_tab + (size_t)(ti + j) * t.stride_q;
            uint4 h = e[0];                                // node, offset, graph (| entries without a record << 20 in the first), flags | ...
            if (j == 0) { n_ent += h.z >> 20; h.z &= 0xFFFFFu; }
            const uint4 x = e[1];                          // two call-count windows, first path word
            const uint4 y = pw_out > 1 ? e[2] : make_uint4(0, 0, 0, 0);   // path words 1, 2
            if (count_here) {
                if (x.x != kEmpty) atomicAdd(&t.attempts[(size_t)row * t.n_windows + x.x], 1u);
                if (x.y != kEmpty) atomicAdd(&t.attempts[(size_t)row * t.n_windows + x.y], 1u);
            }
            if (seeds_here) ns += (h.w >> 10) & 7u;        // seed windows in the entry (groot_hip_read_seeds takes the windows themselves from the host's copy of the table)
            if (j == 0) { mapped += (h.w >> 9) & 1u; multimapped += (h.w >> 8) & 1u; }
            if (j < nt) {                                  // one sam.Record per path of the traversal (alignment.go:114-156)
                alns += __popc(x.z) + __popc(x.w) + __popc(y.x) + __popc(y.y) + __popc(y.z) + __popc(y.w);
                if (pw_out > 3) {
                    const uint32_t *ew = reinterpret_cast<const uint32_t *>(e) + kOutHdrDw;
                    for (uint32_t w = 6; w < 2 * pw_out; w++) alns += __popc(ew[w]);
                }
            }
            if (!fits || j >= nt || (t.exp & 4u)) continue;
            groot_trav tr;
            tr.read_id = t.first_read_id + r; tr.graph_id = h.z; tr.node = h.x; tr.offset = h.y;
            tr.ord = (uint16_t)j; tr.flags = (uint8_t)h.w; tr.reserved = 0;
            uint64_t *mo = mask_out + (size_t)(i + j) * pw_out;
            if (!(t.exp & 8u)) {
                // (the records are written once and read by the copy-out or the next stage of the caller: streaming stores keep
                // them from pushing the outcome table out of L2 / MALL)
                static_assert(sizeof(groot_trav) == 20, "record is five dwords");
                uint32_t tw5[5];
                __builtin_memcpy(tw5, &tr, 20);
                uint32_t *po = reinterpret_cast<uint32_t *>(out + i + j);
#pragma unroll
                for (int d = 0; d < 5; d++) __builtin_nontemporal_store(tw5[d], po + d);
                __builtin_nontemporal_store((uint64_t)x.z | ((uint64_t)x.w << 32), mo);
                if (pw_out > 1) __builtin_nontemporal_store((uint64_t)y.x | ((uint64_t)y.y << 32), mo + 1);
                if (pw_out > 2) __builtin_nontemporal_store((uint64_t)y.z | ((uint64_t)y.w << 32), mo + 2);
                const uint32_t *ew = reinterpret_cast<const uint32_t *>(e) + kOutHdrDw;
                for (uint32_t w = 3; w < pw_out; w++) mo[w] = (uint64_t)ew[2 * w] | ((uint64_t)ew[2 * w + 1] << 32);
                continue;
            }
            out[i + j] = tr;
            mo[0] = (uint64_t)x.z | ((uint64_t)x.w << 32);
            if (pw_out > 1) {
                mo[1] = (uint64_t)y.x | ((uint64_t)y.y << 32);
                if (pw_out > 2) mo[2] = (uint64_t)y.z | ((uint64_t)y.w << 32);
                const uint32_t *ew = reinterpret_cast<const uint32_t *>(e) + kOutHdrDw;
                for (uint32_t w = 3; w < pw_out; w++) mo[w] = (uint64_t)ew[2 * w] | ((uint64_t)ew[2 * w + 1] << 32);
            }
        }
        if (seeds_here) seeds += ns;
    } else if (ti == kEmpty && cnt[r] != 0) {
        const uint32_t i = off[r];
        if (i >= cap) atomicOr(&ctr->flags, kFlagTravOverflow);
        else {
            out[i] = first[r];
            for (uint32_t w = 0; w < pw_out; w++) mask_out[(size_t)i * pw_out + w] = mask_first[(size_t)r * pw_in + w];
        }
    }
    }
    if (!t.tab_idx) return;                                // (uniform)
    alns = block_sum(alns, red);
    mapped = block_sum(mapped, red);
    multimapped = block_sum(multimapped, red);
    seeds = block_sum(seeds, red);
    if (threadIdx.x == 0) {
        if (alns) atomicAdd(&ctr->alignments, alns);
        if (seeds) atomicAdd(&ctr->seeds, seeds);
        if (t.update_weights) {
            if (mapped) atomicAdd(&ctr->mapped, mapped);
            if (multimapped) atomicAdd(&ctr->multimapped, multimapped);
        }
    }
}

__global__ __launch_bounds__(kBlock) void order_ovf_kernel(const groot_trav *ovf, const uint64_t *ovf_mask, const uint32_t *ovf_cnt,
                                                         uint32_t ovf_cap, const uint32_t *off, uint32_t first_read_id, groot_trav *out,
                                                         uint64_t *mask_out, uint32_t cap, uint32_t pw_in, uint32_t pw_out,
                                                         DeviceCounters *ctr, const uint4 *vitem, uint32_t n_reads)
{
    const uint32_t shard = blockIdx.y;
    const uint32_t slot = blockIdx.x * kBlock + threadIdx.x;
    if (slot >= min(ovf_cnt[shard], ovf_cap)) return;
    const size_t o = (size_t)shard * ovf_cap + slot;
    groot_trav t = ovf[o];
    uint32_t rid = t.read_id - first_read_id, ord = t.ord;
    if (rid >= n_reads) {                                  // a record of an item of a split read (AlignArgs::vitem)
        const uint4 vi = vitem[rid - n_reads];
        rid = vi.x; ord += vi.w;
        t.read_id = first_read_id + rid;
        t.ord = (uint16_t)ord;
    }
    const uint32_t i = off[rid] + ord;
    if (i >= cap) { atomicOr(&ctr->flags, kFlagTravOverflow); return; }
    out[i] = t;
    for (uint32_t w = 0; w < pw_out; w++) mask_out[(size_t)i * pw_out + w] = ovf_mask[o * pw_in + w];
}

// ---- compact path sets for the copy-out --------------------------------------------------------------------
// A traversal's path set needs only as many 64-bit words as its graph has paths (one word for 579 of the 583 arg-annot.90
// graphs, three for the widest): the copy-out carries ceil(paths(graph) / 64) words per traversal instead of path_words,
// 33 instead of 46 bytes per read over PCIe.  words[i] for traversal i (0 beyond the batch's count), an exclusive scan of
// them (rocprim), then the copy; every 256th offset is kept as a checkpoint for the host.
// (a record slot may hold anything when an overflow list filled up -- the batch is redone then -- hence the range checks)
__global__ __launch_bounds__(kBlock) void mask_words_kernel(const groot_trav *__restrict__ trav, const DeviceCounters *ctr, uint32_t cap,
                                                          const uint8_t *__restrict__ graph_words, uint32_t n_graphs, uint32_t *__restrict__ words)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= cap) return;
    uint32_t w = 0;
    if (i < min(ctr->n_trav, cap)) {
        const uint32_t g = trav[i].graph_id;
        w = g < n_graphs ? graph_words[g] : 0u;
    }
    words[i] = w;
}
__global__ __launch_bounds__(kBlock) void mask_compact_kernel(const groot_trav *__restrict__ trav, const uint64_t *__restrict__ mask, uint32_t pw_in,
                                                            DeviceCounters *ctr, uint32_t cap, const uint8_t *__restrict__ graph_words, uint32_t n_graphs,
                                                            const uint32_t *__restrict__ off, uint64_t *__restrict__ out, uint32_t *__restrict__ ckpt)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t n = min(ctr->n_trav, cap);
    if (i >= n) return;
    const uint32_t g = trav[i].graph_id;
    const uint32_t o = off[i], w = g < n_graphs ? graph_words[g] : 0u;
    for (uint32_t x = 0; x < w; x++) out[(size_t)o + x] = mask[(size_t)i * pw_in + x];
    if ((i & 255u) == 0) ckpt[i >> 8] = o;
    if (i == n - 1) ctr->mask_words = o + w;
}

// sketch-only entry (groot_hip_sketch): reuse K1 with an index that has no windows
} // namespace groot

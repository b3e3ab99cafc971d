// kernels_common.hpp -- what the gfx950 (CDNA4, wave64) kernels of the `groot align` hot path share: ntHash / MultiHash constants, the table hashes
// the host builds its tables with, 8-bases-at-a-time byte comparison, and the epilogue every seed kernel ends a read with.
//
//   kernels_sig.hpp      sketch_sig_kernel  K1+K2 for reads of at least the window size on the every-slot-equal branch of Query: a few slots of the
//                                           sketch (kSigG), signature index, confirmation by text.  text_lookup_kernel: reads the memo knows.
//   kernels_sketch.hpp   sketch_seed_kernel K1+K2 full width: all S slots at 64 bits, exact table / LSH Forest; LIST instances take the reads the
//                                           kernels above leave.                                                thread per read
//   kernels_align.hpp    align_kernel       K3: the graphMinion loop -- IncrementSubPath call counts, hierarchical exact-match DFS; persistent,
//                                           per-lane state machine with wave-coherent phase scheduling
//   kernels_misc.hpp     heavy LSH-Forest reads, seed-list sort / split, call-count rows, ordering of the traversal records, table builders of open
//
// Integer / byte work throughout: no MFMA.  Hashing is bound by VALU issue, the graph walk by dependent trips to L2; reads are staged through LDS with
// coalesced 16-byte loads; the index is reused by every read and stays in L2 / Infinity Cache.  
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
// LSH-Forest rows (DeviceIndex::band_sig): 5 bits per sketch slot, six slots to a dword (bits 30, 31 zero), four dwords = the first 24 slots
__host__ __device__ __forceinline__ uint32_t sig5(uint64_t v) { return (uint32_t)((v * 0xD6E8FEB86659FD93ULL) >> 59); }
constexpr uint32_t kRowSlots = 24, kRowBytes = 16;
// slots whose 5-bit fields agree in one dword of a row and of the read (an UPPER bound: the borrow of the zero-field test may also flag a field
// of value 1 right above an equal one; fields neither side uses are zero on both and are taken off by the caller)
__device__ __forceinline__ uint32_t row_same6(uint32_t w, uint32_t r)
{
    const uint32_t x = w ^ r;
    return (uint32_t)__builtin_popcount((x - 0x02108421u) & ~x & 0x21084210u);
}

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
// DeviceIndex::node_l2b: the two bits an 8-mer (2 bits per base, base i at bits 2i) sets / tests in a start position's 64-bit set
__host__ __device__ __forceinline__ uint64_t l2_bloom_bits(uint32_t code16)
{
    const uint32_t x = code16 * 0x9E3779B1u;
    return (1ull << (x >> 26)) | (1ull << ((x >> 20) & 63u));
}
// 16-bit code of the first 8 bases of r8, or -1 if one of them is not ACGT
__host__ __device__ __forceinline__ int kmer8_code(uint64_t r8)
{
    int code = 0;
    for (int i = 0; i < 8; i++) {
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
                                              const uint32_t max_win = 0, const uint32_t len_flags = 0)
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
                else if (!ix.win_prefix) no12 = false;            // (the tables are still being built: groot_hip_open_flags)
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
                key |= (((1u << a.sort_span_bits) - 1u) - nn) << a.sort_span_shift;
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
        rq[0] = make_uint4((uint32_t)o0, (uint32_t)(o0 >> 32), len | len_flags, min(n_hits, kRecSplit - 1u) | verdicts | (asc ? kRecAscending : 0u) | (high ? 0x80000000u : 0u));
        // (more than four seeds: the smallest and the largest window instead of the first two -- the align stage starts at the
        // smallest and knows when nothing is left without looking through the list)
        rq[1] = n_hits > 4 ? make_uint4(min_win, max_win, s2, s3) : make_uint4(s0, s1, s2, s3);
    }
    seed_counters(a, r, q, n_hits);
}

// the same for a read known to be bases [o, o + WindowSize) of a window text row: its verdicts come from the table made at open
__device__ __forceinline__ void seed_epilogue_known(const SeedArgs &a, const uint32_t r, const uint64_t o0, const uint32_t len, const uint32_t q,
                                                    const uint32_t n_hits, const uint32_t min_win, const uint32_t s0, const uint32_t s1,
                                                    const uint32_t s2, const uint32_t s3, const uint32_t vbyte, const uint32_t nodes, const bool asc, const uint32_t max_win,
                                                    const uint32_t len_flags = 0)
{
    a.seed_count[r] = n_hits;
    if (a.sort_key) {
        uint32_t key = (min_win << 2) | (vbyte >> 6);
        if (a.sort_span_bits) key |= (((1u << a.sort_span_bits) - 1u) - min(nodes, (1u << a.sort_span_bits) - 1u)) << a.sort_span_shift;
        a.sort_key[r] = key;
    }
    if (a.read_rec) {
        uint4 *rq = reinterpret_cast<uint4 *>(a.read_rec + r);
        rq[0] = make_uint4((uint32_t)o0, (uint32_t)(o0 >> 32), len | len_flags, min(n_hits, kRecSplit - 1u) | (a.sort_key ? (vbyte & 0x3Fu) << 24 : 0u) | (asc ? kRecAscending : 0u));
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

// ---- helpers the host side shares with the kernels (table hashes, LDS layouts) ----
#define GROOT_SIG_HASH_INIT 0x2545F4914F6CDD1DULL
__host__ __device__ __forceinline__ uint64_t sig_hash_step(uint64_t x, uint32_t top27)
{
    x = (x ^ top27) * 0x9E3779B97F4A7C15ULL;
    return x ^ (x >> 29);
}
// sketch_sig_kernel's signature covers kSigG of the S sketch slots (round 5): slot 0 -- the smallest canonical ntHash itself, no multiply -- and
// the kSigG - 1 slots whose MultiHash multipliers i ^ (k * multiSeed) come first in the kernel's running sum h * C0, h * C0 + h, ...
// (slot i sits at step d = i ^ M5, M5 = (k * multiSeed) & 31).  A window whose sketch equals a read's has the same value in THOSE
// slots: no table entry -> no seed, as rigorously as with all S slots; an entry is confirmed by text as before, and a read whose
// entry cannot be confirmed takes the full-width kernel.  The other S - kSigG slots are never computed for reads this kernel decides.
#ifndef GROOT_SIG_G
#define GROOT_SIG_G 13
#endif
constexpr int kSigG = GROOT_SIG_G;
// step d of the j-th signature slot (j = 1..): the j-th smallest d for which slot d ^ m5 exists; -1 if the sketch has too few slots
__host__ __device__ constexpr int sig_step(int j, int s, int m5)
{
    int cnt = 0;
    for (int d = 0; d < 32; d++) {
        const int i = d ^ m5;
        if (i >= 1 && i < s && ++cnt == j) return d;
    }
    return -1;
}
// what the signature keeps of a 64-bit sketch value: the top 24 bits of slot 0 (the kernel tracks the position of the read's smallest
// k-mer in the low byte of that slot's running minimum), the top 27 of the others (MultiHash's t ^= t >> 27 leaves them alone)
__host__ __device__ constexpr uint32_t sig_part(int j, uint64_t v) { return j == 0 ? (uint32_t)(v >> 40) : (uint32_t)(v >> 37); }
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
__host__ __device__ __forceinline__ uint64_t text_hash_step(uint64_t h, uint32_t dw)
{
    h = (h ^ dw) * 0x9E3779B97F4A7C15ULL;
    return h ^ (h >> 29);
}
#define GROOT_TEXT_HASH_INIT 0xD6E8FEB86659FD93ULL
constexpr uint32_t kLshHeavyMaxS = 64;   // lsh_heavy_kernel keeps the read's sketch in LDS
constexpr int kGenericMaxS = 256;      // largest sketch the run-time-sized instance of sketch_seed_kernel handles
constexpr int kGenericMaxBands = kGenericMaxS;   // ... and the most bands (sketch size / maxK >= 1)
// align_kernel is persistent: workgroups per CU (= waves per SIMD) it is compiled and launched for
constexpr int kAlignWaves = 4;         // 4 waves/SIMD = at most 128 VGPRs (5 or 6 spill and are slower; 3 hide too little latency)
constexpr int kAlignWavesWide = 2;     // 704-bit path sets (eleven words in registers + as many per node record): 2 waves = 256 VGPRs, no spills (3 waves: 78 spilled VGPRs for 5 % more speed on graphs of more than 192 paths; 4 waves: 260 bytes of scratch per lane, 40 % slower)
constexpr uint32_t kLshMaxBands = 16;          // bands per read lsh_heavy_kernel handles (sketch size / maxK; `groot index` default: 5)

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

} // namespace groot

#pragma once

#include "kernels_common.hpp"

namespace groot {

// ---------------------------------------------------------------------------------------------
// K1+K2, fast path: sketch_sig_kernel
// ---------------------------------------------------------------------------------------------
// For reads whose Containment > t needs every sketch slot equal (the exact-table branch of sketch_seed_kernel) the seed set is decided
// without ever forming the sketch:
//  * MultiHash mixes with t ^= t >> 27, which leaves the top 27 bits of t alone, and truncation is monotone, so
//    top27(min_j mix(t_j)) = min_j top27(t_j) = (min_j hi32(h_j * c_i)) >> 5: a running 32-bit minimum of the raw product's high word gives the
//    top 27 bits of a slot EXACTLY -- one 64-bit add (v_lshl_add_u64) and half a v_min3_u32 per (k-mer, slot) instead of seven instructions;
//  * round 5, LAZY: only kSigG of the S slots are computed (slot 0 = the smallest canonical hash itself, no multiply; then the slots that come first
//    in the running sum h * C0 + d * h).  A window whose sketch equals the read's agrees in THOSE slots too: no window with this partial signature ->
//    no seed, as rigorously as with all S; the other slots are never needed, because
//  * an entry is confirmed by TEXT: the window's sketch is the sketch of every WindowSize-mer of the bases it was merged from (graph.go:293-333;
//    re-sketched and compared with Key.Sketch when the ctx is opened), so a read that equals one of them, or its reverse complement (canonical
//    k-mer hashes), has exactly that sketch.  Where in the text to compare is known from the read's smallest k-mer (its position in each text
//    row is in the entry).  Its seeds are then all windows of the same sketch class, in ascending window id as the exact table returns them;
//  * the signature index has two levels (device_types.hpp SigEntry): a directory over the distinct signatures, and the windows of a signature back
//    to back, sorted by (class, id), 32 bytes each with the verdict bytes of their first eight offsets inline.  A read first runs through its
//    group with a few instructions per entry (does the text's smallest k-mer sit where the read's does?) and only compares text for a candidate;
//  * ntHash's rolling update takes ONE 16-byte LDS read and four XORs per k-mer: the table is keyed by the (leaving, entering) pair of bases and
//    holds both strands' values already XORed together;
//  * for a confirmed window-sized read the epilogue's verdicts come from the table made at open (the full-width seed stage was run on every
//    WindowSize-mer of every text);
//  * everything else -- a signature found but no text equal (reads with errors that keep those minimisers, windows merged from another path), bytes
//    other than ACGT, other lengths / thresholds (LSH-Forest branch), spans too long for the LDS -- goes onto a list and through
//    sketch_seed_kernel<..., LIST> unchanged.
// Measured (10 M x 100 bp, alone on the chip): 1.65 ms with all 21 slots (round 4) -> 1.25 ms with 13; hashing alone is 0.86 ms (G = 9) .. 1.26 ms
// (G = 20), the rest is the directory / group / text trips.  Fewer slots hash faster and confirm slower (neighbouring windows of a sequence share
// most minimisers: a signature over G slots has ~21 / G windows of a path in its group): G = 5 / 7 / 9 / 13 / 20 -> 2.0 / 1.7 / 1.3 / 1.25 / 1.5 ms.
// Reads are staged as 2-bit codes (6.4 KB per 256 x 100 bp instead of 25.6 KB) -- by the workgroups that will hash any: the reads are classified by length first.
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

#ifndef GROOT_SIG_MIN_LANES
#define GROOT_SIG_MIN_LANES 16
#endif
#ifndef GROOT_SIG_WALK_MAX
#define GROOT_SIG_WALK_MAX 16
#endif
// windows of a signature's group a read looks at before it gives up and takes the full-width kernel (a wavefront waits for its longest walk; on
// arg-annot.90 one group in a hundred is larger)
constexpr uint32_t kSigWalkMax = GROOT_SIG_WALK_MAX;
constexpr uint32_t kSigMinLanes = GROOT_SIG_MIN_LANES;   // (break-even: ~4 300 wave-instructions of hashing against ~180 per read in the list pass)
// TW: dwords of a packed read the text comparison handles (reads of up to 16 * TW bases; longer ones take the full-width kernel)
// BS: threads per workgroup.  64: every wavefront is a workgroup of its own and retires -- frees its registers and LDS for the next one -- as soon as ITS
// reads are done (the walk through a signature's group of windows is a chain of trips to memory whose length differs from read to read; in a workgroup
// of four wavefronts three wait for the slowest).  256: one global atomic per 256 reads for the list of reads left to the full-width kernel, which is
// what counts in a batch where every wavefront has such reads (mixed read lengths).
template <int S, int M5, int TW, int BS = kBlock, int G = kSigG>
#ifndef GROOT_SIG_WAVES_LONG
#define GROOT_SIG_WAVES_LONG 4    // the instance for reads of up to 256 bases compares 16 dwords of text per orientation: 4 waves = 128 VGPRs, no spills (at 5 it spilled 28, at 6 34-60)
#endif
__global__ __launch_bounds__(BS, (TW > 8 ? GROOT_SIG_WAVES_LONG : GROOT_SIG_WAVES) * (kBlock / BS)) void sketch_sig_kernel(SeedArgs a)
{
    static_assert(BS == 64 || BS == kBlock, "one wavefront or four per workgroup");
    static_assert(S >= 1 && S <= 32 && M5 >= 0 && M5 < 32, "slots i < 32 with a compile-time (k * multiSeed) & 31 only");
    static_assert(TW >= 1 && 16 * TW <= (int)kTextMax, "a read cannot be longer than a window text");
    static_assert(G >= 1 && (G == 1 || sig_step(G - 1, S, M5) >= 0), "the sketch has fewer slots than the signature wants");
#ifdef GROOT_SIG_PRIO
    __builtin_amdgcn_s_setprio(GROOT_SIG_PRIO);
#endif
    constexpr int kTextWords = TW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ __attribute__((aligned(512))) unsigned char tab[512];
    uint32_t *badbits = reinterpret_cast<uint32_t *>(smem + kSigBad);
    uint32_t *codes = reinterpret_cast<uint32_t *>(smem + kSigCodes);
    const DeviceIndex &ix = a.ix;
    const unsigned tid = threadIdx.x;
    const uint32_t k = ix.k;
    // ntHash's rolling update XORs two table values into each strand's hash: one for the base that leaves the k-mer, one for the base that enters
    // (forward: rol(seed[out], k) ^ seed[in]; reverse strand: ror(seed[comp out], 1) ^ rol(seed[comp in], k - 1)).  The table is keyed by the PAIR
    // (out | in << 2) and holds both strands' values already XORed together: one 16-byte LDS read and four XORs per k-mer.  16 entries at byte
    // offset 16 * pair (the address is ONE v_and of the shifted code word); behind them the four entries of a base that only enters (first k-mer).
    if (tid < 16) {
        const unsigned o = tid & 3u, i = tid >> 2;
        const uint64_t fo = seed_of_code(o), fco = seed_of_code(o ^ 2u), fi = seed_of_code(i), fci = seed_of_code(i ^ 2u);
        const uint64_t f = rol64(fo, k) ^ fi, rv = ror1(fco) ^ rol64(fci, k - 1);
        *reinterpret_cast<uint4 *>(tab + 16 * tid) = make_uint4((uint32_t)f, (uint32_t)(f >> 32), (uint32_t)rv, (uint32_t)(rv >> 32));
        if (tid < 4) {
            const uint64_t iv = seed_of_code(tid), ir = rol64(seed_of_code(tid ^ 2u), k - 1);
            *reinterpret_cast<uint4 *>(tab + 256 + 16 * tid) = make_uint4((uint32_t)iv, (uint32_t)(iv >> 32), (uint32_t)ir, (uint32_t)(ir >> 32));
        }
    }
    for (uint32_t i = tid; i < 128; i += BS) badbits[i] = 0;
    __shared__ uint32_t list_cnt, list_base;               // the reads this workgroup leaves to the list pass
    if (tid == 0) list_cnt = 0;
    // ---- stage this block's reads as 2-bit codes: one contiguous span, 16 bases per lane per load ----
    const uint32_t r0 = blockIdx.x * BS;
    const uint32_t r_end = min(r0 + (uint32_t)BS, a.n_reads);
    const uint64_t span0 = a.seq_off[r0], span1 = a.seq_off[r_end];
    const uint64_t base16 = span0 & ~15ULL;
    const uint64_t span_bytes = span1 - base16;
    const bool in_lds = span_bytes <= a.lds_read_bytes;
    const uint32_t r = r0 + tid;
    // (no thread leaves before the barriers below: threads without a read, or with one that is answered at once, are predicated off instead)
    const bool valid = r < a.n_reads;
    const uint64_t o0 = valid ? a.seq_off[r] : 0;
    const uint32_t len = valid ? (uint32_t)(a.seq_off[r + 1] - o0) : 0;
    const uint32_t q = len - k + 1;                        // kmerCount, boss.go:169
    // (len >= WindowSize: only then does the read cover whole WindowSize-mers of a text, whose sketches are proven; a shorter
    // read is a substring with FEWER k-mers -- its minima may differ below the 27 signature bits -- and takes the full-width kernel)
    bool fast = valid && in_lds && len >= k && len >= ix.w && len <= a.max_read_len && len <= 16u * TW && q <= ix.max_q;
    const uint32_t qme = (valid && len >= k && q <= ix.max_q) ? ix.q_min_eq[q] : 0u;
    if (fast) fast = qme == (uint32_t)S;                   // else: out of reach (answered below) or the LSH-Forest branch
    // A wavefront hashes at the price of 64 reads however few of its lanes take part (below), and a workgroup stages its 256 reads as 2-bit codes
    // whether anybody hashes them or not: 100+ bytes per read from HBM, and on a batch of mixed read lengths -- 2-3 % window-sized reads, one or two
    // to a wavefront -- nobody does.  The reads are therefore classified by LENGTH first, and a workgroup none of whose wavefronts will hash
    // stages nothing (round 5; the barrier that follows the table set-up carries the vote).
#ifdef GROOT_REP_BAD_LIST
    // The arrangement of round 4's c6ce697 (python __graft_entry__.py rep; tools/rep_bad_list.sh): the wavefront's vote is taken BEFORE the staging and
    // not again behind the bad-base check.  In that variant a dozen reads per 10 M that take the late push (todo_push below) got a slot of the list
    // that was never written; tests/test_signature_path.py::test_no_read_is_left_out_by_the_seed_stage is the guard that sees it.
    if ((uint32_t)__popcll(__ballot(fast)) < kSigMinLanes) fast = false;
    const bool stage = __syncthreads_or(fast ? 1 : 0) != 0;
#else
    const bool stage = __syncthreads_or((uint32_t)__popcll(__ballot(fast)) >= kSigMinLanes) != 0;
#endif
    if (stage && in_lds) {
        const uint4 *src = reinterpret_cast<const uint4 *>(a.seq + base16);
        const uint32_t n16 = (uint32_t)((span_bytes + 15) >> 4);
        for (uint32_t i = tid; i < n16; i += BS) {
            const uint4 v = src[i];
            uint32_t bad = 0;
            const uint32_t c = codes_of4(v.x, bad) | (codes_of4(v.y, bad) << 8) | (codes_of4(v.z, bad) << 16) | (codes_of4(v.w, bad) << 24);
            codes[i] = c;
            if (bad) atomicOr(&badbits[i >> 5], 1u << (i & 31));
        }
    }
    __syncthreads();
    bool answered = !valid;
    if (valid && len >= k && len <= a.max_read_len && (q > ix.max_q || qme > (uint32_t)S)) {
        // Containment > t is out of reach for this many k-mers: no seed, and nothing to hash
        seed_epilogue(a, r, o0, len, q, 0, kEmpty, kEmpty, kEmpty, kEmpty, kEmpty, false);
        answered = true;
    }
    fast = fast && stage;
    if (fast) {
        const uint32_t c0 = (uint32_t)(o0 - base16) >> 4, c1 = (uint32_t)(o0 - base16 + len - 1) >> 4;
        for (uint32_t w = c0 >> 5; w <= c1 >> 5; w++) {
            uint32_t bits = badbits[w];
            if (w == c0 >> 5) bits &= ~0u << (c0 & 31);
            if (w == c1 >> 5) bits &= ~0u >> (31 - (c1 & 31));
            if (bits) fast = false;
        }
    }
    // A wavefront hashes at the price of 64 reads however few of its lanes take part: in a batch of mixed read lengths 2-3 % of the reads are
    // window-sized, four in five wavefronts hold one or two of them, and the kernel cost 1.1 ms per 8 M reads (2.9 beside the align stage) to
    // answer 0.2 M.  Below kSigMinLanes such lanes the wavefront leaves its reads to the list pass, which takes them packed 64 to a wavefront.
#ifndef GROOT_REP_BAD_LIST
    if ((uint32_t)__popcll(__ballot(fast)) < kSigMinLanes) fast = false;
#endif
    {
        // the reads left to the list pass (other lengths, bytes other than ACGT, the LSH-Forest branch): counted per workgroup -- one
        // LDS atomic per wavefront, ONE global atomic per workgroup.  (One global atomic per wavefront on the single counter cost
        // 0.9 of the kernel's 1.9 ms on 8 M mixed-length reads, where every wavefront has such reads: ~7 ns each.)  Every thread
        // takes part in the barriers.
        const unsigned long long here = __ballot(1), mb = __ballot(!fast && !answered);
        const unsigned lane = tid & 63u;
        const int leader = __ffsll(here) - 1;
        uint32_t wave_base = 0;
        if ((int)lane == leader && mb) wave_base = atomicAdd(&list_cnt, (uint32_t)__popcll(mb));
        wave_base = __shfl(wave_base, leader);
        __syncthreads();
        if ((int)lane == leader && mb && wave_base == 0) list_base = atomicAdd(a.todo_count, list_cnt);
        __syncthreads();
        if (answered) return;
        if (!fast) {
            a.todo_list[list_base + wave_base + (uint32_t)__popcll(mb & ((1ULL << lane) - 1ULL))] = r;
            return;
        }
    }

    // ---- the signature slots' running minima, top 32 bits (khf.go:35-55) ----
    // m[0] = min over the k-mers of (top 24 bits of the canonical hash | k-mer index): slot 0 of the sketch IS the smallest canonical hash, and the
    // index of the k-mer that holds it says where the read lies inside a window text.  m[j], j >= 1: slot sig_step(j) ^ M5.
    uint32_t m[G];
#pragma unroll
    for (int i = 0; i < G; i++) m[i] = ~0u;
    const uint64_t C0 = ((uint64_t)k * GROOT_MULTI_SEED) & ~31ULL;
    uint64_t fh = 0, rh = 0;
    uint32_t kj = 0;
    auto ent = [&](uint32_t byte_off) { return *reinterpret_cast<const uint4 *>(tab + byte_off); };
    auto roll = [&](const uint4 e) {
        const uint32_t fl = (uint32_t)fh, fu = (uint32_t)(fh >> 32), rl = (uint32_t)rh, ru = (uint32_t)(rh >> 32);
        const uint32_t nfl = __builtin_amdgcn_alignbit(fl, fu, 31) ^ e.x, nfu = __builtin_amdgcn_alignbit(fu, fl, 31) ^ e.y;   // rol 1
        const uint32_t nrl = __builtin_amdgcn_alignbit(ru, rl, 1) ^ e.z, nru = __builtin_amdgcn_alignbit(rl, ru, 1) ^ e.w;     // ror 1
        fh = (uint64_t)nfl | ((uint64_t)nfu << 32);
        rh = (uint64_t)nrl | ((uint64_t)nru << 32);
    };
    auto slots = [&]() {
        const uint64_t h = fh < rh ? fh : rh;              // canonical
        const uint32_t hl = (uint32_t)h, hu = (uint32_t)(h >> 32);
        m[0] = min(m[0], (hu & ~255u) | kj);
        kj++;
        if constexpr (G > 1) {
            uint64_t acc = (uint64_t)hl * (uint32_t)C0;    // h * C0 = h * c_i for the slot with (i ^ M5) == 0
            acc += (uint64_t)(hl * (uint32_t)(C0 >> 32) + hu * (uint32_t)C0) << 32;
            int at = 0;
#pragma unroll
            for (int j = 1; j < G; j++) {
                const int d = sig_step(j, S, M5);
                acc += (uint64_t)(d - at) * h;             // (d - at: a compile-time 0, 1 or small step over slots the sketch does not have)
                at = d;
                m[j] = min(m[j], (uint32_t)(acc >> 32));
            }
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
                roll(ent(256 + ((uint32_t)t & 0x30u)));
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
            const uint32_t wi = __builtin_amdgcn_alignbit(ni, li, si), wo = __builtin_amdgcn_alignbit(nn, ln, sn);   // 16 entering / 16 leaving codes
            li = ni; ln = nn;
            // pairs (out | in << 2) of the even k-mers 0, 2, .. 14 in the nibbles of pe, of the odd ones in po; both pre-shifted to bits 4..7
            uint64_t pe = (uint64_t)((wo & 0x33333333u) | ((wi << 2) & 0xCCCCCCCCu)) << 4;
            uint64_t po = (uint64_t)(((wo >> 2) & 0x33333333u) | (wi & 0xCCCCCCCCu)) << 4;
#pragma unroll 1
            for (int p = 0; p < 8; p++) {
                const uint32_t a0 = (uint32_t)pe & 0xF0u, a1 = (uint32_t)po & 0xF0u;
                pe >>= 4; po >>= 4;
                roll(ent(a0));
                slots();
                roll(ent(a1));
                slots();
            }
            left -= 16;
        }
        if (left) {
            const uint32_t wi = __builtin_amdgcn_alignbit(codes[di + 1], li, si), wo = __builtin_amdgcn_alignbit(codes[dn + 1], ln, sn);
            for (uint32_t j = 0; j < left; j++) {
                roll(ent((((wo >> (2 * j)) & 3u) | (((wi >> (2 * j)) & 3u) << 2)) << 4));
                slots();
            }
        }
    }

    // ---- ContainmentIndex.Query (lshe.go:153-175), every slot must be equal ----
    uint64_t x = GROOT_SIG_HASH_INIT;
    x = sig_hash_step(x, m[0] >> 8);
#pragma unroll
    for (int i = 1; i < G; i++) x = sig_hash_step(x, m[i] >> 5);
    x = sig_hash_fin(x);
    const uint32_t tag = (uint32_t)(x >> 32);
    // the read as packed codes in registers, and a comparison with len bases of a packed text row starting at base o
    uint32_t rdw[kTextWords];
#pragma unroll
    for (int j = 0; j < kTextWords; j++) rdw[j] = __builtin_amdgcn_alignbit(codes[(P >> 5) + j + 1], codes[(P >> 5) + j], P & 31);
    // ... and for the first pass of the align stage, which then stages the read from two 16-byte loads instead of len unaligned bytes
    uint32_t len_flags = 0;
    if (a.packed) {
        static_assert(kTextWords % 4 == 0, "whole 16-byte words");
        uint4 *pk = a.packed + (size_t)r * a.packed_q;
#pragma unroll
        for (int j = 0; j < kTextWords / 4; j++)
            if ((uint32_t)j < a.packed_q) pk[j] = make_uint4(rdw[4 * j], rdw[4 * j + 1], rdw[4 * j + 2], rdw[4 * j + 3]);
        len_flags = kRecPacked;
    }
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
    uint32_t vbyte = 0, nodes_ahead = 0, conf_id = kEmpty;
    bool have_vbyte = false;
    const uint32_t j0 = m[0] & 255u;
    // directory: the group of windows with this signature, if any (buckets of two; a free place ends the search)
    uint32_t first = kEmpty;
    for (uint32_t b = (uint32_t)x & ix.sig_mask;; b = (b + 1) & ix.sig_mask) {
        const uint4 d = ix.sig_dir[b];
        if (d.y != kEmpty && d.x == tag) { first = d.y; break; }
        if (d.y == kEmpty) break;
        if (d.w != kEmpty && d.z == tag) { first = d.w; break; }
        if (d.w == kEmpty) break;
    }
    uint32_t n_hits = 0, min_win = kEmpty;
    uint32_t s0 = kEmpty, s1 = kEmpty, s2 = kEmpty, s3 = kEmpty;
    bool asc = true;
    uint32_t prev_id = 0, max_win = 0;
    auto hit = [&](uint32_t id) {
        if (n_hits < a.seed_slots) a.seed_win[(size_t)n_hits * a.n_reads + r] = id;
        if (n_hits == 0) s0 = id; else if (n_hits == 1) s1 = id; else if (n_hits == 2) s2 = id; else if (n_hits == 3) s3 = id;
        asc &= n_hits == 0 || id > prev_id;                // (a class's windows follow each other in ascending id)
        prev_id = id;
        n_hits++;
        min_win = min(min_win, id);
        max_win = max(max_win, id);
    };
    if (first != kEmpty) {
        // The group: windows that share the kSigG signature slots -- neighbouring windows of a sequence and its alleles, half a dozen on
        // arg-annot.90 --, sorted by (sketch class, id).  The first window whose text holds the read confirms it; the read's seeds are then ALL
        // windows of that window's class (identical 64-bit sketches), which sit around it.
        const uint4 *grp = reinterpret_cast<const uint4 *>(ix.sig) + 2 * (size_t)first;
        uint32_t n_grp = 1, cls = kEmpty, run0 = 0, run_cls = kEmpty, i = 0;
        // Two nested loops, so that a wavefront pays for a text comparison only when a lane HAS a candidate: the inner one is a few instructions per
        // entry -- the text's smallest k-mer (first occurrence) must be the read's, which fixes where the read would lie in the text, per orientation;
        // most windows of a group (the same sequence a few bases on) fail that --, the outer one compares.  (One loop that did both ran the comparison's
        // instructions for every entry any lane looked at: 1 600 of the kernel's 4 400 instructions per wavefront.)
        while (cls == kEmpty) {
            uint4 e = make_uint4(0, 0, 0, 0);               // {id, class, sig_text_pack(text_len, argmin fwd, argmin rc) | kSigInline, entries left | nodes << 24}
            uint32_t of = 0, orc = 0, tl = 0;
            bool okf = false, okr = false;
            for (; i < n_grp && i < kSigWalkMax; i++) {
                e = grp[2 * i];
                if (i == 0) n_grp = e.w & 0xFFFFFFu;
                if (e.y != run_cls) { run_cls = e.y; run0 = i; }
                tl = sig_text_len(e.z);
                of = sig_text_argmin(e.z, 0) - j0; orc = sig_text_argmin(e.z, 1) - j0;
                okf = tl >= len && of <= tl - len; okr = tl >= len && orc <= tl - len;
                if (okf || okr) break;
            }
            if (!(okf || okr)) break;                        // the group holds no (further) window whose text could be the read
            // the candidate: verdict bytes (second half of the entry, or the table for offsets beyond 7) and the text row(s), in flight together
            const uint4 ev = grp[2 * i + 1];
            const uint8_t *rows = ix.win_text + (size_t)e.x * (2 * kTextMax / 4);
            uint32_t vf = 0, vr = 0;
            if (use_table) {
                const bool inl = (e.z & kSigInline) != 0;
                const uint32_t *vt = ix.sig_info + (size_t)e.x * 2 * ix.sig_verdict_stride;
                if (okf) vf = inl && of < 8u ? ((of < 4u ? ev.x : ev.y) >> (8u * (of & 3u))) & 0xFFu : vt[of];
                if (okr) vr = inl && orc < 8u ? ((orc < 4u ? ev.z : ev.w) >> (8u * (orc & 3u))) & 0xFFu : vt[ix.sig_verdict_stride + orc];
                nodes_ahead = e.w >> 24;
            } else if (len >= 12 && a.sort_key) {          // what the verdicts will need if this is the read's first seed
                ahead.win = e.x;
                load32(ix.win_rec + e.x, ahead.wa, ahead.wb);
                const uint32_t *tabp = ix.win_prefix + (size_t)e.x * kPrefixWords;
                ahead.tf_a = tabp[(code_f & 0xFFFu) >> 5]; ahead.tf_b = tabp[128 + (code_f >> 17)];
                ahead.tr_a = tabp[(code_r & 0xFFFu) >> 5]; ahead.tr_b = tabp[128 + (code_r >> 17)];
            }
            // one comparison for the orientation that fits (both fit for one text in a thousand: then a second one)
            const bool first_f = okf;
            uint32_t d0 = row_differs(first_f ? rows : rows + kTextMax / 4, first_f ? of : orc);
            bool hit_f = first_f && !d0, hit_r = !first_f && !d0;
            if (d0 && okf && okr) hit_r = row_differs(rows + kTextMax / 4, orc) == 0;
            if (hit_f || hit_r) {
                cls = e.y;
                conf_id = e.x;
                have_vbyte = use_table;
                vbyte = hit_f ? vf : vr;
            }
            i++;
        }
        if (cls == kEmpty) { todo_push(a, r); return; }      // a window with this signature, none whose text is the read: the full-width kernel decides
        for (uint32_t i = run0; i < n_grp; i++) {
            const uint4 e = grp[2 * i];
            if (e.y != cls) break;
            hit(e.x);
        }
    }
    if (have_vbyte && (vbyte & kOutTab) && a.tab_idx) seed_epilogue_tab(a, r, q, n_hits, vbyte, s0, s1, s2, s3);
    else if (have_vbyte) seed_epilogue_known(a, r, o0, len, q, n_hits, min_win, s0, s1, s2, s3, vbyte, min_win == conf_id ? nodes_ahead : (uint32_t)ix.win_nodes[min_win], asc, max_win, len_flags);
    else seed_epilogue(a, r, o0, len, q, n_hits, min_win, s0, s1, s2, s3, false, len >= 12, code_f, code_r, &ahead, asc, max_win, len_flags);   // all bytes are ACGT
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

} // namespace groot

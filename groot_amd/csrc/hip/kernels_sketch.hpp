#pragma once

#include "kernels_common.hpp"

namespace groot {

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
// (the S running minima are 2 S registers: at GROOT_SEED_WAVES waves per SIMD (~100 VGPRs) sketch sizes above 30 spilled them --
// 20 ms per 2 M reads at S = 64.  Larger sketches get fewer, larger waves: 3 per SIMD up to S = 48, 2 beyond)
#ifndef GROOT_LIST_WAVES
#define GROOT_LIST_WAVES 4      // the list instance: 4 waves = 128 VGPRs (6 spilled with four 16-byte rows of the LSH-Forest walk and their window ids fetched together).
                                // History: 5 waves spilled 25-50 VGPRs; 3 waves with four rows ahead took the list pass of a mixed-length batch from 4.0 to 3.6 ms; 4 waves x 2 rows,
                                // measured once the signature kernel left its sparse wavefronts' reads to this pass: mixed t = 0.99 1 093 -> 1 176, t = 0.90 606 -> 653 Mreads/s
                                // (4 x 4: 1 174 / 649, 4 x 1: 1 154 / 647, 3 x 8: 1 109 / 607, 5 x 2: 1 073 / 611)
#endif
#ifndef GROOT_LIST_ROWS_AHEAD
#define GROOT_LIST_ROWS_AHEAD 4
#endif
// (round 5: sketch sizes 22..30 at 4 waves -- they spilled 31..95 VGPRs at 5 --, 45 and more at 2 -- 48 spilled 293 at 3; tools/kernel_meta.sh)
constexpr int seed_waves(int S, bool list = false) { return S == 0 ? 1 : (S <= 30 ? (list ? (S <= 24 ? GROOT_LIST_WAVES : 3) : (S <= 21 ? GROOT_SEED_WAVES : 4)) : (S <= 44 ? 3 : 2)); }
template <int S, int MAXK, bool DUMP, int M5, bool LIST = false>
__global__ __launch_bounds__(kBlock, seed_waves(S, LIST)) void sketch_seed_kernel(SeedArgs a)
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
    } else if (min_eq < (uint32_t)s_) {
        // General LSH Forest query: bands b < L, prefix of K hash values (low 32 bits) per band;
        // a window found through band b is skipped if an earlier band already returned it.
        const int lmax_ = s_ / maxk_;
        const uint32_t K = ix.q_k[q], L = ix.q_l[q];
        const uint32_t n = ix.n_windows;
        const int sl_ = s_ < (int)kRowSlots ? s_ : (int)kRowSlots;   // slots covered by the row signatures
        uint32_t rs[4] = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < sl_; i++) rs[i / 6] |= sig5(m[i]) << (5 * (i % 6));
        // the rows of equal prefix in every band first: a read with many of them (a sequence that dozens of graphs share) would keep
        // its lane walking while the other 63 wait -- it goes to lsh_heavy_kernel, a wavefront per read
        // (the run-time-sized instance has room for kGenericMaxBands = kGenericMaxS bands)
        constexpr int LB_ = (S && MAXK) ? (S / MAXK > 0 ? S / MAXK : 1) : kGenericMaxBands;
        uint32_t b_lo[LB_], b_end[LB_];
        uint32_t rows = 0;
        // first row of the (sorted) band table with a band's prefix: hash table over the distinct prefixes, open addressing from `slot` on
        auto probe = [&](int b, uint32_t slot, uint32_t tag) -> uint32_t {
            const uint32_t *keys = ix.band_keys + (size_t)b * n * maxk_;
            const ExactEntry *tab = ix.band_hash + (((size_t)b * maxk_ + (K - 1)) << ix.band_hash_bits);
            const uint32_t hmask = (1u << ix.band_hash_bits) - 1u;
            for (slot &= hmask;; slot = (slot + 1) & hmask) {
                const ExactEntry e = tab[slot];
                if (e.id == kEmpty) return n;
                if (e.tag != tag) continue;
                const uint32_t *ke = keys + (size_t)e.id * maxk_;
                bool eqk = true;
#pragma unroll
                for (int j = 0; j < maxk_; j++)
                    if ((uint32_t)j < K) eqk &= ke[j] == (uint32_t)m[b * maxk_ + j];
                if (eqk) return e.id;
            }
        };
        auto band_hash_of = [&](int b) {
            uint64_t hk = GROOT_SKETCH_HASH_INIT;
#pragma unroll
            for (int j = 0; j < maxk_; j++)
                if ((uint32_t)j < K) hk = sketch_hash_step(hk, (uint32_t)m[b * maxk_ + j]);
            return hk;
        };
        if constexpr (S && MAXK && LIST) {
            // (round 5) the bands' look-ups side by side: the first table entry of every band in one round of loads, then the keys and the run length
            // of the rows they name in a second -- two trips to memory for all bands instead of three per band, one after the other (hash entry ->
            // keys of its row -> length of the run).  A band whose first entry is somebody else's (open addressing) goes on alone.
            const uint32_t hmask = (1u << ix.band_hash_bits) - 1u;
            const uint32_t Kc = K ? K : 1u;
            uint32_t slot0[LB_], tag0[LB_];
            ExactEntry e0[LB_];
#pragma unroll
            for (int b = 0; b < LB_; b++) {
                const uint64_t hk = band_hash_of(b);
                slot0[b] = (uint32_t)hk & hmask; tag0[b] = (uint32_t)(hk >> 32);
                e0[b] = ix.band_hash[((((size_t)b * MAXK + (Kc - 1)) << ix.band_hash_bits)) + slot0[b]];
            }
            uint32_t kv[LB_][MAXK], run0[LB_];
#pragma unroll
            for (int b = 0; b < LB_; b++) {
                const bool cand = (uint32_t)b < L && K >= 1 && e0[b].id != kEmpty && e0[b].tag == tag0[b];
                const uint32_t idc = cand ? e0[b].id : 0u;
                const uint32_t *ke = ix.band_keys + ((size_t)b * n + idc) * MAXK;
#pragma unroll
                for (int j = 0; j < MAXK; j++) kv[b][j] = ke[j];
                run0[b] = ix.band_run[((size_t)b * MAXK + (Kc - 1)) * n + idc];
            }
#pragma unroll
            for (int b = 0; b < LB_; b++) {
                b_lo[b] = n; b_end[b] = n;
                if ((uint32_t)b >= L || K < 1) continue;
                if (e0[b].id == kEmpty) continue;
                bool eqk = e0[b].tag == tag0[b];
#pragma unroll
                for (int j = 0; j < MAXK; j++)
                    if ((uint32_t)j < K) eqk &= kv[b][j] == (uint32_t)m[b * MAXK + j];
                uint32_t lo = e0[b].id, len_run = run0[b];
                if (!eqk) {
                    lo = probe(b, slot0[b] + 1, tag0[b]);
                    len_run = lo < n ? ix.band_run[((size_t)b * MAXK + (K - 1)) * n + lo] : 0;
                }
                b_lo[b] = lo;
                b_end[b] = lo < n ? lo + len_run : n;
                rows += b_end[b] - lo;
            }
        } else {
#pragma unroll
            for (int b = 0; b < lmax_; b++) {
                if (b >= LB_) break;
                b_lo[b] = n; b_end[b] = n;
                if ((uint32_t)b >= L) continue;
                uint32_t lo = n;
                if (K >= 1) {
                    const uint64_t hk = band_hash_of(b);
                    lo = probe(b, (uint32_t)hk, (uint32_t)(hk >> 32));
                }
                b_lo[b] = lo;
                b_end[b] = lo < n ? lo + ix.band_run[((size_t)b * maxk_ + (K - 1)) * n + lo] : n;   // rows with this prefix
                rows += b_end[b] - lo;
            }
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
            const uint4 *sigs = reinterpret_cast<const uint4 *>(ix.band_sig + (size_t)b * n * kRowBytes);
            const uint32_t lo = b_lo[b], e_end = b_end[b];
            // (rows are 16 consecutive bytes each: kRowsAhead of them are fetched together -- the walk is a chain of round trips, two thirds of
            // the kernel's time on mixed-length batches spent waiting)
            constexpr uint32_t kRowsAhead = LIST ? GROOT_LIST_ROWS_AHEAD : GROOT_LSH_ROWS_AHEAD;
            for (uint32_t e4 = lo; e4 < e_end; e4 += kRowsAhead) {
            uint4 rowa[kRowsAhead];
#pragma unroll
            for (uint32_t i = 0; i < kRowsAhead; i++) rowa[i] = sigs[min(e4 + i, e_end - 1u)];
            // (the rows' window ids come along in the same round trip: on mixed-length reads of resfinder.90 four rows in ten pass the filter, and nine
            // in ten of those are seeds -- id, then sketch, was two trips per seed in the lane that holds up its wavefront)
            uint32_t rid[kRowsAhead];
#pragma unroll
            for (uint32_t i = 0; i < kRowsAhead; i++) rid[i] = ids[min(e4 + i, e_end - 1u)];
#pragma unroll
            for (uint32_t i = 0; i < kRowsAhead; i++) {
                const uint32_t e = e4 + i;
                if (e >= e_end) break;
                // slots whose signature fields agree: an upper bound of the equal slots (this filter is most of the branch's time -- runs of
                // ~40 rows per read: only the dwords that hold slots)
                const uint4 sa = rowa[i];
                const uint32_t ws4[4] = {sa.x, sa.y, sa.z, sa.w};
                const int nd = (sl_ + 5) / 6;
                uint32_t same = 0;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    if (i >= nd) break;
                    same += row_same6(ws4[i], rs[i]);
                }
                if (same - (6u * (uint32_t)nd - (uint32_t)sl_) + (uint32_t)(s_ - sl_) < min_eq) continue;
                const uint32_t id = rid[i];
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

} // namespace groot

#pragma once

#include "kernels_common.hpp"

namespace groot {

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
    const uint32_t sl = S < kRowSlots ? S : kRowSlots, nd = (sl + 5) / 6;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t *m = sk_lds + wave * kLshHeavyMaxS;
    uint32_t *blo = aux_lds + wave * (2 * kLshMaxBands + 16);   // [LB] first row per band
    uint32_t *bcum = blo + kLshMaxBands;                        // [LB + 1] rows before band b
    uint32_t *sc = bcum + kLshMaxBands + 1;                     // [0] hits [1] min [2] max [4..7] the first four
    auto wave_sync = []() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); };
    const uint32_t n_list = min(*a.lsh_count, a.lsh_cap);
    for (uint32_t hi = blockIdx.x * (kBlock / 64) + wave; hi < n_list; hi += gridDim.x * (kBlock / 64)) {
        const uint32_t li = hi;
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
        uint32_t rs[4];
#pragma unroll
        for (uint32_t wd = 0; wd < 4; wd++) {
            uint32_t v = 0;
#pragma unroll
            for (uint32_t i = 0; i < 6; i++)
                if (6 * wd + i < sl) v |= sig5(m[6 * wd + i]) << (5 * i);
            rs[wd] = v;
        }
        for (uint32_t t = lane; t < T; t += 64) {
            uint32_t b = 0;
            while (b + 1 < LB && bcum[b + 1] <= t) b++;
            const uint32_t e = blo[b] + (t - bcum[b]);
            const uint4 sa = *reinterpret_cast<const uint4 *>(ix.band_sig + ((size_t)b * n + e) * kRowBytes);
            const uint32_t ws4[4] = {sa.x, sa.y, sa.z, sa.w};
            uint32_t same = 0;
#pragma unroll
            for (uint32_t i = 0; i < 4; i++) {
                if (i >= nd) break;
                same += row_same6(ws4[i], rs[i]);
            }
            if (same - (6u * nd - sl) + (S - sl) < min_eq) continue;
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

// the memo has turned DeviceIndex::sig_info into pointers to tabulated outcomes: the verdict bytes inlined in the signature entries no longer say it all
__global__ __launch_bounds__(kBlock) void sig_inline_off_kernel(SigEntry *ent, uint32_t n)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) ent[i].text_len &= ~kSigInline;
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
                                                           uint32_t *__restrict__ trav_cnt, uint32_t n_reads, DeviceCounters *ctr, groot_trav *__restrict__ first,
                                                           uint64_t *__restrict__ mask_first, uint32_t pw, uint32_t first_read_id)
{
    const uint32_t ns = min(vcount[1], kLongListCap);
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < ns; i += gridDim.x * kBlock) {
        const uint4 sl = split_list[i];
        const uint32_t own = trav_cnt[sl.x];
        uint32_t base = own, first_with = kEmpty;
        for (uint32_t j = sl.y; j < sl.y + sl.z; j++) {
            const uint32_t n = trav_cnt[n_reads + j];
            vitem[j].w = base;
            if (n && first_with == kEmpty) first_with = j;
            base += n;
        }
        if (base > 0xFFFFu) atomicOr(&ctr->flags, kFlagOrdOverflow);
        trav_cnt[sl.x] = base;
        // order_first_kernel copies slot r of every read with records: when the read's own item made none, that slot must hold the
        // read's first record -- the first record of its first item that has any -- and not whatever an earlier batch left there
        if (!own && first_with != kEmpty) {
            groot_trav t = first[n_reads + first_with];
            t.read_id = first_read_id + sl.x;
            t.ord = 0;
            first[sl.x] = t;
            for (uint32_t w = 0; w < pw; w++) mask_first[(size_t)sl.x * pw + w] = mask_first[(size_t)(n_reads + first_with) * pw + w];
        }
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
        // (atomic: the align and order stages of the batch before run beside this kernel on the align stream and add to the same cells)
        if (row != kEmpty) atomicAdd(&attempts[(size_t)row * n_windows + w], v);
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
    const bool count_here = (t.update_weights & 1u) && !(tw & kTabCounted);   // (else the seed stage counted the read's calls)
    const bool seeds_here = tw != kEmpty && (tw & kTabSeedsHere);              // its seed windows are the table's call-count windows
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
            if (!fits || j >= nt) continue;
            groot_trav tr;
            tr.read_id = t.first_read_id + r; tr.graph_id = h.z; tr.node = h.x; tr.offset = h.y;
            tr.ord = (uint16_t)j; tr.flags = (uint8_t)h.w; tr.reserved = 0;
            uint64_t *mo = mask_out + (size_t)(i + j) * pw_out;
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
// A traversal's path set needs only as many bits as its graph has paths (three per graph on average on arg-annot.90, 704 for the
// widest graph supported): the copy-out carries ceil(paths(graph) / 8) BYTES per traversal instead of path_words 64-bit words --
// 13.6 instead of 24.5 (round 3: whole words) or 46 bytes per read over PCIe.  words[i] for traversal i (0 beyond the batch's count), an exclusive scan of
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
                                                            const uint32_t *__restrict__ off, uint8_t *__restrict__ out, uint32_t *__restrict__ ckpt)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t n = min(ctr->n_trav, cap);
    if (i >= n) return;
    const uint32_t g = trav[i].graph_id;
    const uint32_t o = off[i], w = g < n_graphs ? graph_words[g] : 0u;
    for (uint32_t x = 0; x < w; x++) out[(size_t)o + x] = (uint8_t)(mask[(size_t)i * pw_in + (x >> 3)] >> (8 * (x & 7u)));   // (w bytes)
    if ((i & 255u) == 0) ckpt[i >> 8] = o;
    if (i == n - 1) ctr->mask_words = o + w;
}


} // namespace groot

namespace groot {

// ---- DeviceIndex::win_prefix, built on the device (groot_hip_open) ------------------------------------------------------------------
// Which read prefixes can AlignRead's levels 1-2 start on?  Two tables per window: the 6-mer codes (2 bits per base, A=0 C=1 T=2 G=3) of
// oriented read bases [0,6) and [6,12) that some level-1 / level-2 start position of the window (alignment.go:34-70) can spell --
// following every out-edge, with the graph's 'N' and the graph ends (a read may hang off a sink, alignment.go:229-236) as wildcards.
// Sound filters: a read whose code is absent from either table cannot pass performAlignment from any of those starts.
// Two steps: (1) per graph base position and table the 4096-bit set of codes a walk from there spells (a thread per position and table,
// depth-first with explicit frames: a frame per node entered and per 'N' branched on, at most two dozen); (2) per window the union over
// its start positions (a workgroup per window, a thread per 32-bit word, the positions' sets read as whole 512-byte rows).
struct PrefixBuildArgs {
    const uint8_t *bases;
    const uint32_t *seq_off, *edge_off, *edges;   // [n_nodes + 1], [n_nodes + 1], [n_edges]
    uint32_t n_nodes, p0, p1;                      // the positions of this pass: [p0, p1) -- the bases of a run of whole graphs
    uint32_t *pos_bits;                            // [p1 - p0][2][128]
};
constexpr int kPrefixK = 6, kPrefixFrames = 40;
__global__ __launch_bounds__(kBlock) void prefix_positions_kernel(PrefixBuildArgs a)
{
    const uint32_t gid = blockIdx.x * kBlock + threadIdx.x;
    if (gid >= 2u * (a.p1 - a.p0)) return;
    const uint32_t pos = a.p0 + (gid >> 1), tb = gid & 1u;
    const int d0 = (int)tb * kPrefixK;
    uint32_t *bits = a.pos_bits + (size_t)gid * 128;
    // the node holding the position: the last node whose first base is at or before it (empty nodes hold none)
    uint32_t lo = 0, hi = a.n_nodes;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (a.seq_off[mid] <= pos) lo = mid; else hi = mid;
    }
    while (lo + 1 < a.n_nodes && a.seq_off[lo + 1] <= pos) lo++;
    // frames: {node, offset, depth | code << 8, cursor}; cursor = next out-edge (or next base of an 'N') to try, kEmpty = the frame has not run yet
    uint32_t f_node[kPrefixFrames], f_off[kPrefixFrames], f_dc[kPrefixFrames], f_cur[kPrefixFrames];
    int sp = 0;
    f_node[0] = lo; f_off[0] = pos - a.seq_off[lo]; f_dc[0] = 0; f_cur[0] = kEmpty; sp = 1;
    auto set_code = [&](uint32_t code) { bits[code >> 5] |= 1u << (code & 31u); };
    while (sp > 0) {
        const int f = sp - 1;
        const uint32_t node = f_node[f];
        const uint32_t s0 = a.seq_off[node], len = a.seq_off[node + 1] - s0;
        if (f_cur[f] == kEmpty) {
            // first visit: spell the node's bases from the frame's offset
            uint32_t off = f_off[f];
            int depth = (int)(f_dc[f] & 0xFFu);
            uint32_t code = f_dc[f] >> 8;
            bool wild = false;
            while (off < len && depth < d0 + kPrefixK) {
                if (depth >= d0) {
                    const uint8_t b = a.bases[s0 + off];
                    if (b == 'N') { wild = true; break; }
                    code |= (uint32_t)((b >> 1) & 3u) << (2 * (depth - d0));
                }
                depth++; off++;
            }
            f_off[f] = off; f_dc[f] = (uint32_t)depth | (code << 8);
            if (wild) { f_cur[f] = 0x80000000u; continue; }             // branch over the four bases at `off`
            if (depth == d0 + kPrefixK) { set_code(code); sp--; continue; }
            const uint32_t e0 = a.edge_off[node], e1 = a.edge_off[node + 1];
            if (e0 == e1) {                                             // a sink: every completion counts
                const int have = max(0, depth - d0);
                const uint32_t low = code & ((1u << (2 * have)) - 1u);
                for (uint32_t x = 0; x < (1u << (2 * (kPrefixK - have))); x++) set_code(low | (x << (2 * have)));
                sp--;
                continue;
            }
            f_cur[f] = e0;
            continue;
        }
        if (f_cur[f] & 0x80000000u) {                                   // an 'N' at f_off: the next of the four bases
            const uint32_t c = f_cur[f] & 3u, done = (f_cur[f] >> 2) & 1u;
            if (done) { sp--; continue; }
            f_cur[f] = c == 3u ? (0x80000000u | 4u) : (0x80000000u | (c + 1u));
            const int depth = (int)(f_dc[f] & 0xFFu);
            const uint32_t code = f_dc[f] >> 8;
            if (sp < kPrefixFrames) {
                f_node[sp] = node; f_off[sp] = f_off[f] + 1; f_dc[sp] = (uint32_t)(depth + 1) | ((code | (c << (2 * (depth - d0)))) << 8); f_cur[sp] = kEmpty;
                sp++;
            } else for (uint32_t w = 0; w < 128; w++) bits[w] = ~0u;    // (deeper than any real graph: anything goes -- sound)
            continue;
        }
        // out-edges, one per visit
        const uint32_t e1 = a.edge_off[node + 1];
        if (f_cur[f] >= e1) { sp--; continue; }
        const uint32_t child = a.edges[f_cur[f]++];
        if (sp < kPrefixFrames) {
            f_node[sp] = child; f_off[sp] = 0; f_dc[sp] = f_dc[f]; f_cur[sp] = kEmpty;
            sp++;
        } else for (uint32_t w = 0; w < 128; w++) bits[w] = ~0u;
    }
}

// (2) the union over a window's start positions: level 1 = offsets [OffSet, l1_hi) of the seed node, level 2 = offsets 0..10 of every
// contained node.  One workgroup of 256 threads per window: thread i owns word i of the window's two tables.
__global__ __launch_bounds__(kBlock) void prefix_windows_kernel(const WinRec *__restrict__ win_rec, const uint4 *__restrict__ cn_pre, const uint32_t *__restrict__ seq_off,
                                                                const uint32_t *__restrict__ pos_bits, uint32_t p0, uint32_t w0, uint32_t w1, uint32_t *__restrict__ out)
{
    const uint32_t w = w0 + blockIdx.x;
    if (w >= w1) return;
    const WinRec wr = win_rec[w];
    const uint32_t i = threadIdx.x;                                     // table (i >> 7), word (i & 127): a position's two tables are 256 consecutive words
    uint32_t acc = 0;
    for (uint32_t o = wr.offset; o < wr.l1_hi; o++) acc |= pos_bits[(size_t)(wr.seed_s0 + o - p0) * 256 + i];
    for (uint32_t c = wr.cn_off; c < wr.cn_end; c++) {
        const uint4 e1 = cn_pre[2 * (size_t)c + 1];
        const uint32_t s0 = seq_off[e1.z], n = min(e1.w, 11u);
        for (uint32_t o = 0; o < n; o++) acc |= pos_bits[(size_t)(s0 + o - p0) * 256 + i];
    }
    out[(size_t)w * kPrefixWords + i] = acc;
}

} // namespace groot

// kernels.hpp -- gfx950 (CDNA4, wave64) kernels of the `groot align` hot path.
//
//   sketch_seed_kernel   K1+K2: per read ntHash -> KHF MinHash sketch (registers) -> LSH-Ensemble
//                        containment lookup -> per-read seed slots.          thread per read
//   align_kernel         K3: per read the graphMinion loop: IncrementSubPath call counts and the
//                        hierarchical exact-match DFS alignment.               thread per read
//   gather_trav_kernel   reorder traversal records into canonical (read, ord) order
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

// LDS layout of sketch_seed_kernel (bytes)
constexpr uint32_t kLdsTabF = 0;                 // u64[256] seedTab[b]
constexpr uint32_t kLdsTabFout = 2048;           // u64[256] rol(seedTab[b], k)
constexpr uint32_t kLdsTabC = 4096;              // u64[8]   seedTab[c]            c = b & 7
constexpr uint32_t kLdsTabCout = 4096 + 64;      // u64[8]   ror(seedTab[c], 1)
constexpr uint32_t kLdsTabCin = 4096 + 128;      // u64[8]   rol(seedTab[c], k-1)
constexpr uint32_t kLdsReads = 4096 + 192;       // staged read bytes (16-byte aligned)

// ---------------------------------------------------------------------------------------------
// K1+K2
// ---------------------------------------------------------------------------------------------
template <int S, int MAXK, bool DUMP>
__global__ __launch_bounds__(kBlock) void sketch_seed_kernel(SeedArgs a)
{
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
    const uint32_t r0 = blockIdx.x * kBlock;
    const uint32_t r_end = min(r0 + (uint32_t)kBlock, a.n_reads);
    const uint64_t span0 = a.seq_off[r0], span1 = a.seq_off[r_end];
    const uint64_t base16 = span0 & ~15ULL;
    const uint64_t span_bytes = span1 - base16;
    const bool in_lds = span_bytes <= a.lds_read_bytes;
    if (in_lds) {
        const uint4 *src = reinterpret_cast<const uint4 *>(a.seq + base16);
        uint4 *dst = reinterpret_cast<uint4 *>(lds_reads);
        const uint32_t n16 = (uint32_t)((span_bytes + 15) >> 4);
        for (uint32_t i = tid; i < n16; i += kBlock) dst[i] = src[i];
    }
    __syncthreads();

    const uint32_t r = r0 + tid;
    if (r >= a.n_reads) return;
    const uint64_t o0 = a.seq_off[r];
    const uint32_t len = (uint32_t)(a.seq_off[r + 1] - o0);
    uint32_t n_hits = 0;
    if (len < k) {                       // NewHasher error -> panic (khf.go:38-41, boss.go:164-166)
        atomicOr(&a.ctr->flags, kFlagShortRead);
        atomicAdd(&a.ctr->short_reads, 1ULL);
        a.seed_count[r] = 0;
        return;
    }
    if (len > a.max_read_len) {
        atomicOr(&a.ctr->flags, kFlagLongRead);
        a.seed_count[r] = 0;
        return;
    }
    // ---- KHF sketch (khf.go:35-55): per slot i, min over k-mers of MultiHash_i(canonical ntHash) ----
    uint64_t m[S];
#pragma unroll
    for (int i = 0; i < S; i++) m[i] = ~0ULL;
    const uint64_t M = (uint64_t)k * GROOT_MULTI_SEED;
    const uint32_t nk = len - k + 1;
    auto sketch = [&](const unsigned char *rd) {
        uint64_t fh = 0, rh = 0;
        for (uint32_t j = 0; j < k; j++) {   // ntf64 / ntr64 of the first k-mer in one pass
            const unsigned b = rd[j];
            fh = rol1(fh) ^ tabF[b];
            rh ^= rol64(tabC[b & 7], j);
        }
        for (uint32_t j = 0;;) {
            const uint64_t h = fh < rh ? fh : rh;          // canonical
            m[0] = h < m[0] ? h : m[0];
#pragma unroll
            for (int i = 1; i < S; i++) {
                uint64_t t = h * ((uint64_t)i ^ M);
                t ^= t >> GROOT_MULTI_SHIFT;
                m[i] = t < m[i] ? t : m[i];
            }
            if (++j == nk) break;
            const unsigned prev = rd[j - 1], end = rd[j + k - 1];
            fh = rol1(fh) ^ tabFout[prev] ^ tabF[end];
            rh = ror1(rh) ^ tabCout[prev & 7] ^ tabCin[end & 7];
        }
    };
    if (in_lds) sketch(lds_reads + (o0 - base16));   // LDS address space
    else sketch(a.seq + o0);                         // span too large for LDS: straight from HBM
    if (DUMP) {
#pragma unroll
        for (int i = 0; i < S; i++) a.sketch_out[(size_t)r * S + i] = m[i];
    }

    // ---- ContainmentIndex.Query (lshe.go:153-175) ----
    const uint32_t q = nk;                                 // kmerCount, boss.go:169
    const uint32_t min_eq = q <= ix.max_q ? ix.q_min_eq[q] : (uint32_t)S + 1;
    auto hit = [&](uint32_t id) {
        if (n_hits < a.seed_slots) a.seed_win[(size_t)n_hits * a.n_reads + r] = id;
        n_hits++;
    };
    if (min_eq == (uint32_t)S) {
        // Containment > t needs every slot equal: windows with an identical sketch.  One probe
        // sequence of the exact-match table (all such windows are consecutive probes).
        uint64_t hs = GROOT_SKETCH_HASH_INIT;
#pragma unroll
        for (int i = 0; i < S; i++) hs = sketch_hash_step(hs, m[i]);
        const uint32_t tag = (uint32_t)(hs >> 32);
        for (uint32_t slot = (uint32_t)hs & ix.exact_mask;; slot = (slot + 1) & ix.exact_mask) {
            const ExactEntry e = ix.exact[slot];
            if (e.id == kEmpty) break;
            if (e.tag != tag) continue;
            const uint64_t *ws = ix.win_sketch + (size_t)e.id * S;
            bool same = true;
#pragma unroll
            for (int i = 0; i < S; i++) same &= ws[i] == m[i];
            if (same) hit(e.id);
        }
    } else if (min_eq < (uint32_t)S) {
        // General LSH Forest query: bands b < L, prefix of K hash values (low 32 bits) per band;
        // a window found through band b is skipped if an earlier band already returned it.
        constexpr int LMAX = S / MAXK;
        const uint32_t K = ix.q_k[q], L = ix.q_l[q];
        const uint32_t n = ix.n_windows;
#pragma unroll
        for (int b = 0; b < LMAX; b++) {
            if ((uint32_t)b >= L) break;
            const uint32_t *keys = ix.band_keys + (size_t)b * n * MAXK;
            const uint32_t *ids = ix.band_ids + (size_t)b * n;
            auto cmp = [&](uint32_t e) {      // -1 / 0 / +1 : table entry e vs query prefix
                const uint32_t *ke = keys + (size_t)e * MAXK;
#pragma unroll
                for (int j = 0; j < MAXK; j++) {
                    if ((uint32_t)j >= K) break;
                    const uint32_t qv = (uint32_t)m[b * MAXK + j], kv = ke[j];
                    if (kv != qv) return kv < qv ? -1 : 1;
                }
                return 0;
            };
            uint32_t lo = 0, hi = n;
            while (lo < hi) {
                const uint32_t mid = lo + ((hi - lo) >> 1);
                if (cmp(mid) >= 0) hi = mid; else lo = mid + 1;
            }
            for (uint32_t e = lo; e < n && cmp(e) == 0; e++) {
                const uint32_t id = ids[e];
                const uint64_t *ws = ix.win_sketch + (size_t)id * S;
                uint32_t eq = 0;
                bool earlier = false;
#pragma unroll
                for (int bb = 0; bb < LMAX; bb++) {
                    bool pm = true;
#pragma unroll
                    for (int j = 0; j < MAXK; j++) {
                        const uint64_t wv = ws[bb * MAXK + j];
                        eq += wv == m[bb * MAXK + j];
                        if ((uint32_t)j < K) pm &= (uint32_t)wv == (uint32_t)m[bb * MAXK + j];
                    }
                    if (bb < b && pm) earlier = true;
                }
#pragma unroll
                for (int i = LMAX * MAXK; i < S; i++) eq += ws[i] == m[i];
                if (!earlier && eq >= min_eq) hit(id);
            }
        }
    }
    a.seed_count[r] = n_hits;
    if (n_hits) {
        atomicAdd(&a.ctr->seeds, (unsigned long long)n_hits);
        atomicMax(&a.ctr->max_seeds, n_hits);
        if (n_hits > a.seed_slots) atomicOr(&a.ctr->flags, kFlagSeedOverflow);
    }
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

struct ReadRef {
    const uint8_t *p;   // forward read
    uint32_t len;       // full length
    uint32_t rc;        // orientation
    uint32_t clip_lo;   // bases hard-clipped at the start of the oriented read
    uint32_t eff;       // effective length being aligned
    __device__ __forceinline__ unsigned at(uint32_t d) const
    {
        const uint32_t i = d + clip_lo;
        return rc ? comp_base(p[len - 1 - i]) : p[i];
    }
};

struct EmitCtx {
    uint32_t local_read, read_id, graph, flags;
    uint32_t ord;              // per-read running traversal counter
    unsigned long long alns;   // per-thread popcount sum
};

// performAlignment (alignment.go:162-193): every traversal from (node0, off0) spelling the read
// (dfsRecursive, :196-254), kept with the path set present in all of its nodes (processTraversal,
// :263-317).  Iterative DFS in OutEdges order; only pending alternatives are stacked; branches whose
// path set is already empty are cut (they can only yield traversals without ids).
template <int PW>
__device__ uint32_t perform_alignment(const AlignArgs &a, const uint32_t stack_tid, const ReadRef &rd, const uint32_t node0,
                                      const uint32_t off0, EmitCtx &ec)
{
    const DeviceIndex &ix = a.ix;
    uint32_t emitted = 0, sp = 0;
    uint32_t cur = node0, off = off0, dist = 0;
    uint64_t mask[PW];
#pragma unroll
    for (int w = 0; w < PW; w++) mask[w] = ~0ULL;
    {
        const uint32_t nlen = ix.node_seq_off[node0 + 1] - ix.node_seq_off[node0];
        if (off0 >= nlen) return 0;                                   // alignment.go:199-201
    }
    for (;;) {
        const uint32_t s0 = ix.node_seq_off[cur], s1 = ix.node_seq_off[cur + 1];
        const uint32_t avail = s1 - s0 - off;
        const uint32_t take = min(avail, rd.eff - dist);
        bool ok = true;
        const uint8_t *gb = ix.bases + s0 + off;
        for (uint32_t i = 0; i < take; i++) {
            const unsigned g = gb[i];
            if (g != 'N' && g != rd.at(dist + i)) { ok = false; break; }   // :212-222
        }
        if (ok) {
            dist += take;
            bool any = false;
#pragma unroll
            for (int w = 0; w < PW; w++) {
                mask[w] &= ix.node_mask[(size_t)cur * PW + w];
                any |= mask[w] != 0;
            }
            const uint32_t e0 = ix.node_edge_off[cur], e1 = ix.node_edge_off[cur + 1];
            if (dist == rd.eff || e0 == e1) {                         // :229-236 report the traversal
                if (any) {
                    const uint32_t slot = atomicAdd(&a.ctr->n_trav, 1u);
                    if (slot < a.trav_cap) {
                        groot_trav t;
                        t.read_id = ec.read_id; t.graph_id = ec.graph; t.node = node0; t.offset = off0;
                        t.ord = (uint16_t)ec.ord;
                        t.flags = (uint8_t)(ec.flags | (emitted == 0 ? GROOT_TRAV_FIRST : 0));
                        t.reserved = 0;
                        a.trav[slot] = t;
#pragma unroll
                        for (int w = 0; w < PW; w++) a.trav_mask[(size_t)slot * PW + w] = mask[w];
                        a.trav_key[slot] = ((uint64_t)ec.local_read << 16) | (ec.ord & 0xFFFFu);
                    } else {
                        atomicOr(&a.ctr->flags, kFlagTravOverflow);
                    }
                    if (ec.ord >= 0xFFFFu) atomicOr(&a.ctr->flags, kFlagOrdOverflow);
                    ec.ord++;
#pragma unroll
                    for (int w = 0; w < PW; w++) ec.alns += __popcll(mask[w]);
                    emitted++;
                }
            } else if (any) {
                if (e1 - e0 > 1) {                                    // alternatives e0+1.. stay pending
                    const size_t si = (size_t)sp * a.n_threads + stack_tid;
                    a.stk_hdr[si] = (uint64_t)cur | (1ULL << 32) | ((uint64_t)dist << 48);
#pragma unroll
                    for (int w = 0; w < PW; w++) a.stk_mask[si * PW + w] = mask[w];
                    sp++;
                }
                cur = ix.edges[e0];
                off = 0;
                continue;
            }
        }
        if (sp == 0) break;                                           // backtrack to the newest pending edge
        const size_t si = (size_t)(sp - 1) * a.n_threads + stack_tid;
        const uint64_t hdr = a.stk_hdr[si];
        const uint32_t pn = (uint32_t)hdr, next = (uint32_t)(hdr >> 32) & 0xFFFFu;
        dist = (uint32_t)(hdr >> 48);
#pragma unroll
        for (int w = 0; w < PW; w++) mask[w] = a.stk_mask[si * PW + w];
        const uint32_t e0 = ix.node_edge_off[pn], deg = ix.node_edge_off[pn + 1] - e0;
        cur = ix.edges[e0 + next];
        off = 0;
        if (next + 1 == deg) sp--;
        else a.stk_hdr[si] = (uint64_t)pn | ((uint64_t)(next + 1) << 32) | ((uint64_t)dist << 48);
    }
    return emitted;
}

// AlignRead (alignment.go:13-159) for one orientation of the read against one seed window
template <int PW>
__device__ uint32_t align_read(const AlignArgs &a, const uint32_t stack_tid, const uint8_t *p, const uint32_t len,
                               const uint32_t rc, const uint32_t w, EmitCtx &ec)
{
    const DeviceIndex &ix = a.ix;
    const uint32_t seed = ix.win_node[w], off0 = ix.win_offset[w];
    const uint32_t seed_len = ix.node_seq_off[seed + 1] - ix.node_seq_off[seed];
    ReadRef rd{p, len, rc, 0, len};
    const uint32_t base_flags = rc ? GROOT_TRAV_RC : 0;
    ec.flags = base_flags;
    // 1. seed offset shuffling (:34-45): offsets past the node end fail immediately
    {
        const uint64_t last = (uint64_t)off0 + ix.win_merge_span[w] + ix.w;
        for (uint32_t off = off0; off <= last && off < seed_len; off++) {
            const uint32_t n = perform_alignment<PW>(a, stack_tid, rd, seed, off, ec);
            if (n) return n;
        }
    }
    // 2. seed node shuffling (:47-70): ContainedNodes ascending SegmentID, offsets 0..10
    for (uint32_t c = ix.win_cn_off[w]; c < ix.win_cn_off[w + 1]; c++) {
        const uint32_t node = ix.cn_node[c];
        const uint32_t nlen = ix.node_seq_off[node + 1] - ix.node_seq_off[node];
        for (uint32_t off = 0; off <= 10 && off < nlen; off++) {
            const uint32_t n = perform_alignment<PW>(a, stack_tid, rd, node, off, ec);
            if (n) return n;
        }
    }
    // 3. hard clip the first base (:72-85)
    rd.clip_lo = 1; rd.eff = len - 1;
    ec.flags = base_flags | GROOT_TRAV_START_CLIP;
    {
        const uint32_t n = perform_alignment<PW>(a, stack_tid, rd, seed, off0, ec);
        if (n) return n;
    }
    // 4. hard clip the last base (:87-103)
    rd.clip_lo = 0;
    ec.flags = base_flags | GROOT_TRAV_END_CLIP;
    return perform_alignment<PW>(a, stack_tid, rd, seed, off0, ec);
}

template <int PW>
__global__ __launch_bounds__(kBlock) void align_kernel(AlignArgs a)
{
    const DeviceIndex &ix = a.ix;
    const uint32_t gtid = blockIdx.x * kBlock + threadIdx.x;
    // the seed stage ran out of per-read slots: the host grows them and re-runs the whole batch
    if (a.ctr->flags & kFlagSeedOverflow) return;
    unsigned long long alns = 0;
    for (uint32_t r = gtid; r < a.n_reads; r += a.n_threads) {
        uint32_t cnt = a.seed_count[r];
        if (cnt == 0) continue;
        if (cnt > a.seed_slots) cnt = a.seed_slots;       // overflow already flagged; batch is re-run
        const uint64_t o0 = a.seq_off[r];
        const uint32_t len = (uint32_t)(a.seq_off[r + 1] - o0);
        const uint8_t *p = a.seq + o0;
        const uint32_t q = len - ix.k + 1;                // graphminion.go:60 kmerCount
        EmitCtx ec{r, a.first_read_id + r, 0, 0, 0, 0};
        uint32_t n_graphs = 0;
        int high_byte = -1;                               // lazily: does RevComplement panic on this read?
        // seeds in canonical order = ascending window id (graph, Node, OffSet, list position)
        long long last = -1;
        uint32_t done_graph = kEmpty;                     // graph whose minion already found an alignment
        uint32_t cur_graph = kEmpty;
        bool group_rc_called = false;
        for (uint32_t it = 0; it < cnt; it++) {
            uint32_t w = kEmpty;
            for (uint32_t j = 0; j < cnt; j++) {
                const uint32_t cand = a.seed_win[(size_t)j * a.n_reads + r];
                if ((long long)cand > last && cand < w) w = cand;
            }
            if (w == kEmpty) break;                       // duplicates cannot occur; defensive
            last = w;
            const uint32_t g = ix.win_graph[w];
            if (g != cur_graph) { cur_graph = g; n_graphs++; group_rc_called = false; }
            if (g == done_graph) continue;                // graphminion.go:96-98 break after first alignment
            if (a.update_weights) atomicAdd(&a.attempts[(size_t)q * ix.n_windows + w], 1u);   // :67 IncrementSubPath
            if (a.no_align) continue;                     // :70-72
            ec.graph = g;
            bool found = false;
            for (uint32_t t = 0; t < 2; t++) {            // :76-95 forward, then reverse complement
                if (align_read<PW>(a, threadIdx.x + blockIdx.x * kBlock, p, len, t, w, ec)) { found = true; break; }
                if (!group_rc_called) {                   // first RevComplement of this minion's copy of the read
                    group_rc_called = true;
                    if (high_byte < 0) {
                        high_byte = 0;
                        for (uint32_t i = 0; i < len; i++) high_byte |= p[i] > 'T';
                    }
                    if (high_byte && a.update_weights) atomicAdd(&a.ctr->revcomp_panics, 1ULL);
                }
            }
            if (found) done_graph = g;
        }
        if (a.update_weights) {
            atomicAdd(&a.ctr->mapped, 1ULL);                              // boss.go:195-200
            if (n_graphs > 1) atomicAdd(&a.ctr->multimapped, 1ULL);
        }
        alns += ec.alns;
    }
    if (alns) atomicAdd(&a.ctr->alignments, alns);
}

// canonical order: perm[i] = index of the i-th record after sorting keys
__global__ __launch_bounds__(kBlock) void gather_trav_kernel(const groot_trav *in, const uint64_t *mask_in, const uint32_t *perm,
                                                           groot_trav *out, uint64_t *mask_out, uint32_t n, uint32_t pw_in,
                                                           uint32_t pw_out)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = perm[i];
    out[i] = in[s];
    for (uint32_t w = 0; w < pw_out; w++) mask_out[(size_t)i * pw_out + w] = mask_in[(size_t)s * pw_in + w];
}

__global__ __launch_bounds__(kBlock) void iota_kernel(uint32_t *p, uint32_t n)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) p[i] = i;
}

// sketch-only entry (groot_hip_sketch): reuse K1 with an index that has no windows
} // namespace groot

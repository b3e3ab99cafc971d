#pragma once

#include "kernels_common.hpp"

namespace groot {

// ---------------------------------------------------------------------------------------------
// K3, first pass: the reads whose whole graphMinion loop (graphminion.go:46-102) is ONE seed window and whose walks never have a
// second neighbour to come back to -- most reads of an error-free batch.  A thread per read, in the processing order of the seed
// stage (neighbouring lanes hold reads of the same window: they walk the same nodes in step), no phase scheduling, no stack:
//   * the read is staged in the lane's LDS slice at 2 bits per base (A=0 C=1 T=2 G=3, the code of the signature kernel); 32 bases of
//     either strand come out of it with three ds_read_b32 and two v_alignbit (the reverse complement: v_bfrev + a pair swap + xor);
//   * the graph side is 2 bits per base too: LeanNode holds a node's first 32 bases, `bases2` the rest -- a node of a hundred bases
//     is compared in four 64-bit steps of ONE trip (align_kernel: 8 ASCII bases per comparison, 32 per step);
//   * AlignRead's hierarchy (alignment.go:13-110) is followed candidate by candidate exactly as align_kernel does it (same filters:
//     the seed stage's verdicts, the first min(8, ...) bases inside the node, the 8-mer set of the start position), so a read that
//     finishes here produces, bit for bit, what align_kernel would have produced for it;
//   * whatever does not fit -- more than one seed window, a byte other than ACGT, a node with an 'N', two neighbours that both take
//     the next base (dfsRecursive would come back to the second: alignment.go:242-252) -- is left UNTOUCHED: the read's slot is
//     flagged, a stream compaction keeps the flagged slots in processing order, and align_kernel walks them as before.
// 64 VGPRs: eight wavefronts per SIMD, and workgroups that retire -- the next batch's hashing kernels get their share of the chip.
// ---------------------------------------------------------------------------------------------

#ifndef GROOT_LEAN_WAVES
#define GROOT_LEAN_WAVES 6
#endif
constexpr int kLeanWaves = GROOT_LEAN_WAVES;

__device__ __forceinline__ uint64_t lean_lowmask(uint32_t n) { return n >= 32u ? ~0ull : ((1ull << (2u * n)) - 1ull); }
// order of the 32 two-bit fields reversed, every base complemented (A<->T = 0<->2, C<->G = 1<->3: code ^ 2)
__device__ __forceinline__ uint64_t lean_revcomp32(uint64_t v)
{
    uint64_t r = __brevll(v);
    r = ((r & 0xAAAAAAAAAAAAAAAAull) >> 1) | ((r & 0x5555555555555555ull) << 1);
    return r ^ 0xAAAAAAAAAAAAAAAAull;
}
// 64 bits from bit position `bit` of a dword array
__device__ __forceinline__ uint64_t lean_bits64(const uint32_t *wp, uint32_t bit)
{
    const uint32_t *q = wp + (bit >> 5);
    const uint32_t x0 = q[0], x1 = q[1], x2 = q[2];
    const uint32_t sh = bit & 31u;
    return (uint64_t)__funnelshift_r(x0, x1, sh) | ((uint64_t)__funnelshift_r(x1, x2, sh) << 32);
}

template <int PW>
__global__ __launch_bounds__(kBlock, kLeanWaves) void align_lean_kernel(LeanArgs a)
{
    static_assert(PW == 3, "LeanNode holds three path words");
    extern __shared__ __attribute__((aligned(16))) uint32_t lean_lds[];
    __shared__ unsigned long long red[4];
    uint32_t *my = lean_lds + (size_t)threadIdx.x * a.lds_stride_dw;
    const uint32_t slot = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t n_todo = min(a.n_reads, (uint32_t)__builtin_amdgcn_readfirstlane((int)a.ctr->seeded_reads));
    // the seed stage ran out of slots or rows: the host grows them and runs the batch again (align_kernel returns at once, too)
    const bool stale = (a.ctr->flags & (kFlagSeedOverflow | kFlagQOverflow)) != 0;

    enum : uint32_t { ST_GEN, ST_WALK, ST_DONE, ST_DEFER, ST_IDLE };
    uint32_t st = ST_IDLE;
    uint32_t r = 0, len = 0, w = 0, g = 0, vbits = 0;
    uint32_t seed = 0, seed_s0 = 0, seed_len = 0, off0 = 0, l1_hi = 0, cn_begin = 0, cn_end = 0;
    if (slot < n_todo && !stale) {
        r = a.perm[slot];
        uint4 ra, rb;
        load32(a.read_rec + r, ra, rb);
        const uint32_t sc = ra.w;
        len = ra.z;
        w = rb.x;
        vbits = (sc >> 24) & 0x3Fu;
        st = ST_GEN;
        // one seed window, no byte > 'T' (RevComplement would panic on it: align_kernel counts that), a read the slice holds
        if ((sc & kRecCountMask) != 1u || (sc >> 31) || len > a.max_len || len < 12u) st = ST_DEFER;
        else {
            const uint8_t *p = a.seq + ((uint64_t)ra.x | ((uint64_t)ra.y << 32));
            uint4 wa, wb;
            load32(a.win_rec + w, wa, wb);
            const uint32_t ok = a.win_ok[w];
            // ---- stage the read: 16 bases = one 16-byte load = one dword of codes ----
            my[0] = 0; my[1] = 0;
            uint32_t bad = 0;
            const uint32_t nd = (len + 15u) >> 4;
            for (uint32_t i = 0; i < nd; i++) {                // reads at most 15 bytes past the read's end
                uint4 v;
                __builtin_memcpy(&v, p + 16u * i, 16);
                const uint32_t x[4] = {v.x, v.y, v.z, v.w};
                uint32_t codes = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t c = (x[j] >> 1) & 0x03030303u;
                    // anything but ACGT: the byte the code stands for differs from the byte that is there
                    uint32_t diff = __builtin_amdgcn_perm(0u, 0x47544341u, c) ^ x[j];
                    const int left = (int)len - (int)(16u * i + 4u * j);     // bytes of this dword inside the read
                    if (left < 4) diff = left <= 0 ? 0u : diff & ((1u << (8 * left)) - 1u);
                    bad |= diff;
                    uint32_t t = c | (c >> 6);
                    t = (t | (t >> 12)) & 0xFFu;
                    codes |= t << (8 * j);
                }
                my[2 + i] = codes;
            }
            my[2 + nd] = 0; my[3 + nd] = 0;
            g = wa.x; seed = wa.y; off0 = wa.z; l1_hi = wa.w;
            cn_begin = wb.x; cn_end = wb.y; seed_s0 = wb.z; seed_len = wb.w;
            if (bad || !ok) st = ST_DEFER;
        }
    }

    // ---- the view of the read a hierarchy level works on: orientation, clip, effective length (alignment.go:72-103) ----
    uint32_t rc = 0, level = 0, clip = 0, eff = len;
    // oriented view bases [d, d + 32) at 2 bits each (bits past the view's end are don't-care)
    auto chunk = [&](uint32_t d) -> uint64_t {
        const uint32_t i = d + clip;
        // forward: the read starts at bit 64 of the slice; reverse complement: oriented bases [i, i+32) are the reverse complement
        // of read bases [len-i-32, len-i), which start at bit 64 + 2 (len-i-32) = 2 (len-i) (>= 0: two zero dwords come first)
        const uint64_t v = lean_bits64(my, rc ? 2u * (len - i) : 64u + 2u * i);
        return rc ? lean_revcomp32(v) : v;
    };
    uint32_t p16 = 0;             // first eight bases of the view
    uint64_t need = 0;            // their two bits in a start position's 8-mer set
    // candidate cursor: level 1 -- pos = next offset of the seed node; level 2 -- pos = ContainedNodes entry, sub = next offset in it
    uint32_t pos = 0, sub = 0;
    // walk
    uint32_t node0 = 0, noff0 = 0, cur = 0, coff = 0, dist = 0;
    uint64_t cur64 = 0, m0 = 0, m1 = 0, m2 = 0;
    bool emitted = false;

    // verdict of the seed stage on the read's (only) seed window, current orientation: bit 0 levels 1-2, bit 1 level 3, bit 2 level 4
    auto verdict = [&](uint32_t bit) -> bool { return ((vbits >> (rc ? 3 : 0)) >> bit) & 1u; };
    auto set_view = [&]() {
        clip = level == 3u ? 1u : 0u;
        eff = len - (level >= 3u ? 1u : 0u);
        p16 = (uint32_t)chunk(0) & 0xFFFFu;
        need = l2_bloom_bits(p16);
    };
    // move to the next level / orientation that has a candidate range; ST_DONE when nothing is left (no alignment for the read)
    auto enter = [&](uint32_t lv) {
        for (;;) {
            if (lv == 1u) {
                level = 1; set_view();
                if (verdict(0)) { lv = 3; continue; }          // levels 1 and 2 cannot start anywhere (prefix tables)
                pos = off0;
                if (pos < l1_hi) return;
                lv = 2; continue;
            }
            if (lv == 2u) {
                level = 2; pos = cn_begin; sub = 0;
                if (pos < cn_end) return;
                lv = 3; continue;
            }
            if (lv == 3u) {                                    // alignment.go:72-85: the first base clipped, at (seed, OffSet)
                if (off0 >= seed_len) { lv = 5; continue; }    // :199-201 holds for levels 3 and 4 alike
                if (verdict(1)) { lv = 4; continue; }
                level = 3; set_view(); pos = 0;
                return;
            }
            if (lv == 4u) {                                    // :87-103: the last base clipped
                if (verdict(2)) { lv = 5; continue; }
                level = 4; set_view(); pos = 0;
                return;
            }
            // AlignRead found nothing in this orientation: graphminion.go:94 RevComplement
            if (rc == 0) { rc = 1; lv = 1; continue; }
            st = ST_DONE;
            return;
        }
    };
    auto begin_walk = [&](uint32_t node, uint32_t off) {
        node0 = node; noff0 = off; cur = node; coff = off; dist = 0;
        cur64 = chunk(0);
        m0 = m1 = m2 = ~0ull;
        st = ST_WALK;
    };
    // a start position whose first min(8, bases left in the node, eff) bases equal the view's: the 8-mer set, then the walk
    auto try_start = [&](uint32_t node, uint32_t off) {
        if (a.node_l2b && off <= 10u && eff >= 8u && (a.node_l2b[(size_t)node * 11 + off] & need) != need) return;
        begin_walk(node, off);
    };
    if (st == ST_GEN) enter(1);

    while (__ballot(st == ST_GEN || st == ST_WALK)) {
        if (st == ST_GEN) {
            if (level == 1u) {
                // up to 24 start offsets pos.. of the seed node against the first eight bases of the view
                const uint64_t G = lean_bits64(a.bases2, 2u * (seed_s0 + pos));
                const uint32_t n = min(24u, l1_hi - pos);
                uint32_t j = 0;
                for (; j < n; j++) {
                    const uint32_t m = min(min(seed_len - (pos + j), eff), 8u);
                    if ((((uint32_t)(G >> (2u * j)) ^ p16) & ((1u << (2u * m)) - 1u)) == 0u) break;
                }
                if (j < n) { const uint32_t off = pos + j; pos = off + 1u; try_start(seed, off); }
                else pos += n;
                if (st == ST_GEN && pos >= l1_hi) enter(2);
            } else if (level == 2u) {
                // offsets sub..10 of ContainedNodes entry pos (alignment.go:47-70), from its 16-byte prefix record
                const uint4 e = a.cn_pre2[pos];
                const uint64_t G = (uint64_t)e.x | ((uint64_t)(e.y & 0xFFFFu) << 32);
                const uint32_t nlen = e.y >> 16, n = min(nlen, 11u);
                uint32_t j = sub;
                for (; j < n; j++) {
                    const uint32_t m = min(min(nlen - j, eff), 8u);
                    if ((((uint32_t)(G >> (2u * j)) ^ p16) & ((1u << (2u * m)) - 1u)) == 0u) break;
                }
                if (j < n) { sub = j + 1u; try_start(e.z, j); }
                else sub = n;
                if (sub >= n) { pos++; sub = 0; }
                if (st == ST_GEN && pos >= cn_end) enter(3);
            } else {
                // levels 3 and 4: the single start position (seed, OffSet)
                if (pos == 0u) {
                    pos = 1;
                    const uint32_t G = (uint32_t)lean_bits64(a.bases2, 2u * (seed_s0 + off0));
                    const uint32_t m = min(min(seed_len - off0, eff), 8u);
                    if (((G ^ p16) & ((1u << (2u * m)) - 1u)) == 0u) try_start(seed, off0);
                }
                if (st == ST_GEN) enter(level + 1u);
            }
        } else if (st == ST_WALK) {
            // ---- one node of the walk (dfsRecursive, alignment.go:196-254, for a read that never has two neighbours to choose from) ----
            const uint4 *q = reinterpret_cast<const uint4 *>(a.nodes + cur);
            const uint4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
            const uint32_t seq_off = q0.x, seq_len = q0.y, dk = q0.z;
            bool fail = false;
            if (dk & kLeanNo) st = ST_DEFER;
            else {
                const uint32_t take = min(seq_len - coff, eff - dist);
                if (coff == 0u && take <= 32u) {
                    const uint64_t f32 = (uint64_t)q2.x | ((uint64_t)q2.y << 32);
                    fail = ((f32 ^ cur64) & lean_lowmask(take)) != 0ull;
                } else {
                    for (uint32_t done = 0; done < take; done += 32u) {
                        const uint64_t gb = lean_bits64(a.bases2, 2u * (seq_off + coff + done));
                        const uint64_t rd = done ? chunk(dist + done) : cur64;
                        if ((gb ^ rd) & lean_lowmask(take - done)) { fail = true; break; }
                    }
                }
                if (!fail) {
                    dist += take;
                    m0 &= (uint64_t)q2.z | ((uint64_t)q2.w << 32);
                    m1 &= (uint64_t)q3.x | ((uint64_t)q3.y << 32);
                    m2 &= (uint64_t)q3.z | ((uint64_t)q3.w << 32);
                    const bool any = (m0 | m1 | m2) != 0ull;
                    const uint32_t deg = dk & 7u;
                    if (dist == eff || deg == 0u) {               // :229-236 report the traversal
                        if (any) {
                            groot_trav t;
                            t.read_id = a.first_read_id + r; t.graph_id = g; t.node = node0; t.offset = noff0;
                            t.ord = 0;
                            t.flags = (uint8_t)((rc ? GROOT_TRAV_RC : 0u) | (level == 3u ? GROOT_TRAV_START_CLIP : level == 4u ? GROOT_TRAV_END_CLIP : 0u) | GROOT_TRAV_FIRST);
                            t.reserved = 0;
                            a.trav_first[r] = t;
                            a.mask_first[(size_t)r * PW] = m0; a.mask_first[(size_t)r * PW + 1] = m1; a.mask_first[(size_t)r * PW + 2] = m2;
                            emitted = true;
                            st = ST_DONE;
                        } else fail = true;
                    } else if (!any) fail = true;                 // no path left: descendants cannot yield ids
                    else {
                        cur64 = chunk(dist);
                        const uint32_t nextb = (uint32_t)cur64 & 3u;
                        const uint32_t ed[4] = {q1.x, q1.y, q1.z, q1.w};
                        uint32_t hits = 0, pick = 0;
                        bool wild = false;
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const uint32_t code = (dk >> (8 + 4 * e)) & 15u;
                            if ((uint32_t)e < deg) {
                                wild |= code == 4u;
                                if (code == nextb) { hits++; pick = ed[e]; }
                            }
                        }
                        if (wild || hits > 1u) st = ST_DEFER;   // an 'N' ahead, or a second neighbour to come back to: align_kernel's business
                        else if (hits == 0u) fail = true;
                        else { cur = pick; coff = 0; }
                    }
                }
                if (fail) {                                       // the start position yields nothing: on with the hierarchy
                    st = ST_GEN;
                    if (level == 1u) { if (pos >= l1_hi) enter(2); }
                    else if (level == 2u) { if (pos >= cn_end) enter(3); }
                    else enter(level + 1u);
                }
            }
        }
    }

    // ---- what the read leaves behind ----
    const bool fin = st == ST_DONE;
    if (slot < a.n_reads) a.defer[slot] = st == ST_DEFER ? 1 : 0;
    unsigned long long alns = 0, mapped = 0;
    if (fin) {
        a.trav_cnt[r] = emitted ? 1u : 0u;
        mapped = 1;                                               // boss.go:195-200
        if (emitted) alns = (unsigned long long)(__popcll(m0) + __popcll(m1) + __popcll(m2));
    }
    if (a.update_weights) {                                       // graphminion.go:67 IncrementSubPath, once: the read's only seed window
        // neighbouring lanes mostly hold reads of the same window: one atomic per RUN of equal cells among the lanes that finished here
        const uint64_t cell = (uint64_t)(fin ? a.q_row[len - a.k + 1u] : 0u) * a.n_windows + w;
        const unsigned long long here = __ballot(fin);
        const unsigned lane_ = threadIdx.x & 63u;
        const unsigned long long below = here & ((1ull << lane_) - 1ull);
        const int prev = below ? 63 - __builtin_clzll(below) : -1;
        const uint32_t plo = __shfl((uint32_t)cell, prev < 0 ? (int)lane_ : prev), phi = __shfl((uint32_t)(cell >> 32), prev < 0 ? (int)lane_ : prev);
        const bool head = fin && (prev < 0 || plo != (uint32_t)cell || phi != (uint32_t)(cell >> 32));
        const unsigned long long heads = __ballot(head);
        if (head) {
            const unsigned long long after = lane_ == 63u ? 0ull : heads & ~((2ull << lane_) - 1ull);
            const unsigned long long run = here & ~((1ull << lane_) - 1ull) & (after ? ((1ull << (__ffsll(after) - 1)) - 1ull) : ~0ull);
            atomicAdd(&a.attempts[cell], (uint32_t)__popcll(run));
        }
    }
    alns = block_sum(alns, red);
    mapped = block_sum(mapped, red);
    if (threadIdx.x == 0) {
        if (alns) atomicAdd(&a.ctr->alignments, alns);
        if (a.update_weights && mapped) atomicAdd(&a.ctr->mapped, mapped);
        if (mapped) atomicAdd(&a.ctr->lean_reads, (unsigned int)mapped);
    }
}

} // namespace groot

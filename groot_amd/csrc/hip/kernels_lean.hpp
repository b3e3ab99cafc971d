#pragma once

#include "kernels_common.hpp"

namespace groot {

// ---------------------------------------------------------------------------------------------
// K3, first pass: the reads whose whole graphMinion loop (graphminion.go:46-102) is ONE seed window and whose walks never have a
// second neighbour to come back to -- five reads in six of an error-free batch.  A thread per read, in the processing order of the
// seed stage (neighbouring lanes hold reads of the same window: they walk the same nodes in step), no phase scheduling, no stack:
//   * the read is staged in the lane's LDS slice at 2 bits per base (A=0 C=1 T=2 G=3, the code of the signature kernel), one strand
//     at a time (the slice is reverse-complemented in place when AlignRead's forward hierarchy has failed: graphminion.go:94);
//     32 bases of the current view come out of it with three ds_read_b32 and two v_alignbit;
//   * the graph side is 2 bits per base too: LeanNode holds a node's first 32 bases and LeanExt the next 224; a walk step is ONE trip
//     to memory whatever the node's length (record + extension, or record + `bases2` for a walk that starts inside a node, + the
//     start position's 8-mer set, all in flight together) -- align_kernel compares 8 ASCII bases at a time, 32 per step;
//   * AlignRead's hierarchy (alignment.go:13-110) is followed candidate by candidate exactly as align_kernel does it (same filters:
//     the seed stage's verdicts, the first min(8, ...) bases inside the node, the 8-mer set of the start position), so a read that
//     finishes here produces, bit for bit, what align_kernel would have produced for it;
//   * whatever does not fit -- more than one seed window, a byte other than ACGT, a node with an 'N', two neighbours that both take
//     the next base (dfsRecursive would come back to the second: alignment.go:242-252) -- is left UNTOUCHED: the read's slot is
//     flagged, a stream compaction keeps the flagged slots in processing order, and align_kernel walks them as before.
// ---------------------------------------------------------------------------------------------

#ifndef GROOT_LEAN_WAVES
#define GROOT_LEAN_WAVES 6
#endif
constexpr int kLeanWaves = GROOT_LEAN_WAVES;

__device__ __forceinline__ uint64_t lean_lowmask(int n) { return n <= 0 ? 0ull : (n >= 32 ? ~0ull : ((1ull << (2 * n)) - 1ull)); }
// order of the 16 two-bit fields reversed, every base complemented (A<->T = 0<->2, C<->G = 1<->3: code ^ 2)
__device__ __forceinline__ uint32_t lean_revcomp16(uint32_t v)
{
    uint32_t r = __brev(v);
    r = ((r & 0xAAAAAAAAu) >> 1) | ((r & 0x55555555u) << 1);
    return r ^ 0xAAAAAAAAu;
}
__device__ __forceinline__ uint64_t lean_funnel(uint32_t x0, uint32_t x1, uint32_t x2, uint32_t sh)
{
    return (uint64_t)__funnelshift_r(x0, x1, sh) | ((uint64_t)__funnelshift_r(x1, x2, sh) << 32);
}
// 64 bits from bit position `bit` of a dword array
__device__ __forceinline__ uint64_t lean_bits64(const uint32_t *wp, uint32_t bit)
{
    const uint32_t *q = wp + (bit >> 5);
    return lean_funnel(q[0], q[1], q[2], bit & 31u);
}

// NCH: 64-bit pieces a node comparison looks at (reads of up to 32 * NCH bases)
template <int PW, int NCH>
__global__ __launch_bounds__(kBlock, kLeanWaves) void align_lean_kernel(LeanArgs a)
{
    static_assert(PW == 3, "LeanNode holds three path words");
    static_assert(NCH >= 1 && NCH <= 8, "LeanExt holds bases [32, 256)");
    extern __shared__ __attribute__((aligned(16))) uint32_t lean_lds[];
    __shared__ unsigned long long red[4];
    uint32_t *my = lean_lds + (size_t)threadIdx.x * a.lds_stride_dw;
    uint32_t *W = my + (a.lds_stride_dw - 7u);                 // seed, OffSet, l1_hi, cn_begin, cn_end, seed_s0, seed_len of the read's window
    enum : int { W_SEED, W_OFF0, W_L1HI, W_CNB, W_CNE, W_S0, W_SLEN };
    const uint32_t slot = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t n_todo = min(a.n_reads, (uint32_t)__builtin_amdgcn_readfirstlane((int)a.ctr->seeded_reads));
    // the seed stage ran out of slots or rows: the host grows them and runs the batch again (align_kernel returns at once, too)
    const bool stale = (a.ctr->flags & (kFlagSeedOverflow | kFlagQOverflow)) != 0;
#if defined(GROOT_WORK_COUNTERS) && GROOT_WORK_COUNTERS == 4
    const unsigned long long lw_t0 = wall_clock64();
#endif
    enum : uint32_t { ST_ADV, ST_GEN, ST_WALK, ST_DONE, ST_DEFER, ST_IDLE };
    uint32_t st = ST_IDLE;
    uint32_t r = 0, len = 0, w = 0, vbits = 0;
    if (slot < n_todo && !stale) {
        r = a.perm[slot];
        uint4 ra, rb;
        load32(a.read_rec + r, ra, rb);
        const uint32_t sc = ra.w;
        len = ra.z & ~kRecPacked;
        w = rb.x;
        vbits = (sc >> 24) & 0x3Fu;
        st = ST_ADV;
        // one seed window, no byte > 'T' (RevComplement would panic on it: align_kernel counts that), a read the slice holds
        if ((sc & kRecCountMask) != 1u || (sc >> 31) || len > a.max_len || len < 12u) st = ST_DEFER;
        else {
            const uint8_t *p = a.seq + ((uint64_t)ra.x | ((uint64_t)ra.y << 32));
            uint4 wa, wb;
            load32(a.win_rec + w, wa, wb);
            const uint32_t ok = a.win_ok[w];
            // ---- stage the read at 16 bases per dword ----
            my[0] = 0; my[1] = 0;
            uint32_t bad = 0;
            const uint32_t nd = (len + 15u) >> 4;
            if ((ra.z & kRecPacked) && a.packed) {             // the signature kernel left its codes (all ACGT): packed_q 16-byte loads
                const uint4 *pk = a.packed + (size_t)r * a.packed_q;
                for (uint32_t i = 0; 4u * i < nd; i++) {
                    const uint4 v = pk[i];
                    my[2 + 4 * i] = v.x;
                    if (4u * i + 1u < nd) my[3 + 4 * i] = v.y;
                    if (4u * i + 2u < nd) my[4 + 4 * i] = v.z;
                    if (4u * i + 3u < nd) my[5 + 4 * i] = v.w;
                }
            } else
            for (uint32_t i = 0; i < nd; i++) {                // 16 bases = one 16-byte load (at most 15 bytes past the read's end) = one dword of codes
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
            W[W_SEED] = wa.y; W[W_OFF0] = wa.z; W[W_L1HI] = wa.w;
            W[W_CNB] = wb.x; W[W_CNE] = wb.y; W[W_S0] = wb.z; W[W_SLEN] = wb.w;
            if (bad || !ok) st = ST_DEFER;
        }
    }

    // ---- the view of the read a hierarchy level works on: orientation, clip, effective length (alignment.go:72-103) ----
    uint32_t rc = 0, level = 0, eff = len;
    uint32_t sbit = 64;           // bit of the slice where the view's first base sits
    // view bases [d, d + 32) at 2 bits each (bits past the view's end are don't-care)
    auto chunk = [&](uint32_t d) -> uint64_t { return lean_bits64(my, sbit + 2u * d); };
    uint32_t p16 = 0;             // first eight bases of the view
    uint64_t need = 0;            // their two bits in a start position's 8-mer set
    // candidate cursor: level 1 -- pos = next offset of the seed node, lim = l1_hi; level 2 -- pos = ContainedNodes entry, lim = cn_end,
    // sub = next offset in the entry; levels 3, 4 -- pos = 0 before the single start position, lim = 1
    uint32_t pos = 0, sub = 0, lim = 0;
    // walk
    uint32_t node0 = 0, noff0 = 0, cur = 0, cur_s0 = 0, coff = 0, dist = 0;
    bool cur_long = false;
    uint64_t cur64 = 0, m0 = 0, m1 = 0, m2 = 0;
    bool emitted = false;

    // verdict of the seed stage on the read's (only) seed window, current orientation: bit 0 levels 1-2, bit 1 level 3, bit 2 level 4
    auto verdict = [&](uint32_t bit) -> bool { return ((vbits >> (rc ? 3 : 0)) >> bit) & 1u; };
    // move to level lv / the next one that has a candidate range / the other orientation; ST_DONE when nothing is left
    auto enter = [&](uint32_t lv) {
        const uint32_t off0 = W[W_OFF0], seed_len = W[W_SLEN];
        bool view = false;
        for (;;) {
            if (lv == 1u) {
                level = 1; view = true;
                if (!verdict(0)) {                                 // (else levels 1 and 2 cannot start anywhere: prefix tables)
                    pos = off0; lim = W[W_L1HI];
                    if (pos < lim) break;
                    lv = 2; continue;
                }
                lv = 3; continue;
            }
            if (lv == 2u) {
                level = 2; pos = W[W_CNB]; lim = W[W_CNE]; sub = 0;
                if (pos < lim) break;
                lv = 3; continue;
            }
            if (lv == 3u) {                                        // alignment.go:72-85: the first base clipped, at (seed, OffSet)
                if (off0 >= seed_len) { lv = 5; continue; }        // :199-201 holds for levels 3 and 4 alike
                if (verdict(1)) { lv = 4; continue; }
                level = 3; view = true; pos = 0; lim = 1;
                break;
            }
            if (lv == 4u) {                                        // :87-103: the last base clipped
                if (verdict(2)) { lv = 5; continue; }
                level = 4; view = true; pos = 0; lim = 1;
                break;
            }
            // AlignRead found nothing in this orientation: graphminion.go:94 RevComplement
            if (rc == 0) {
                rc = 1; lv = 1;
                // the slice becomes the reverse complement, padded to whole dwords in FRONT: dword k <- revcomp16(dword nd-1-k)
                const uint32_t nd = (len + 15u) >> 4;
                for (uint32_t k2 = 0; 2u * k2 < nd; k2++) {
                    const uint32_t x = my[2 + k2], y = my[1 + nd - k2];
                    my[2 + k2] = lean_revcomp16(y);
                    my[1 + nd - k2] = lean_revcomp16(x);
                }
                continue;
            }
            st = ST_DONE;
            return;
        }
        st = ST_GEN;
        if (view) {
            sbit = 64u + (rc ? 2u * (16u * ((len + 15u) >> 4) - len) : 0u) + (level == 3u ? 2u : 0u);
            eff = len - (level >= 3u ? 1u : 0u);
            p16 = (uint32_t)chunk(0) & 0xFFFFu;
            need = l2_bloom_bits(p16);
        }
    };
#if defined(GROOT_WORK_COUNTERS) && GROOT_WORK_COUNTERS == 4
    // dbg[140..]: iterations with a level-1 / level-2 / level-3-4 / walk lane; lane-steps of each; wave iterations; ticks (100 MHz) staging / loop
    unsigned long long lw[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long lw_t1 = wall_clock64();
#define LEAN_EV(i, pred) do { const unsigned long long b__ = __ballot(pred); if (b__) { lw[i]++; lw[(i) + 4] += (unsigned long long)__popcll(b__); } } while (0)
#else
#define LEAN_EV(i, pred) ((void)0)
#endif

#ifdef GROOT_LEAN_PROBE      // (tools: 1 = every read is left to align_kernel right after staging: what the prologue costs)
    if (GROOT_LEAN_PROBE == 1 && st <= ST_WALK) st = ST_DEFER;
#endif
    while (__ballot(st <= ST_WALK)) {
        if (st == ST_ADV) enter(level + 1u);
#if defined(GROOT_WORK_COUNTERS) && GROOT_WORK_COUNTERS == 4
        lw[8]++;
        LEAN_EV(0, st == ST_GEN && level == 1u); LEAN_EV(1, st == ST_GEN && level == 2u); LEAN_EV(2, st == ST_GEN && level >= 3u); LEAN_EV(3, st == ST_WALK);
#endif
        if (st == ST_GEN) {
            // ---- the next start position of the level whose first min(8, bases left in the node, eff) bases equal the view's ----
            bool have = false;
            uint32_t c_node = 0, c_off = 0, c_s0 = 0, c_len = 0;
            if (level == 2u) {
                // offsets sub..10 of ContainedNodes entry pos (alignment.go:47-70), from its 16-byte prefix record
                const uint4 e = a.cn_pre2[pos];
                const uint64_t G = (uint64_t)e.x | ((uint64_t)(e.y & 0xFFFFu) << 32);
                const uint32_t nlen = e.y >> 16, n = min(nlen, 11u);
                uint32_t j = sub;
                for (; j < n; j++) {
                    const uint32_t m = min(min(nlen - j, eff), 8u);
                    if ((((uint32_t)(G >> (2u * j)) ^ p16) & ((1u << (2u * m)) - 1u)) == 0u) break;
                }
                if (j < n) { have = true; c_node = e.z; c_off = j; c_s0 = e.w; c_len = nlen; sub = j + 1u; }
                else sub = n;
                if (sub >= n) { pos++; sub = 0; }
            } else {
                // level 1: up to 24 start offsets pos.. of the seed node (alignment.go:34-45); levels 3, 4: the one at OffSet
                const uint32_t s0 = W[W_S0], slen = W[W_SLEN];
                const uint32_t from = level == 1u ? pos : W[W_OFF0];
                const uint64_t G = lean_bits64(a.bases2, 2u * (s0 + from));
                const uint32_t n = level == 1u ? min(24u, lim - pos) : 1u;
                uint32_t j = 0;
                for (; j < n; j++) {
                    const uint32_t m = min(min(slen - (from + j), eff), 8u);
                    if ((((uint32_t)(G >> (2u * j)) ^ p16) & ((1u << (2u * m)) - 1u)) == 0u) break;
                }
                if (j < n) { have = true; c_node = W[W_SEED]; c_off = from + j; c_s0 = s0; c_len = slen; }
                pos = level == 1u ? (j < n ? c_off + 1u : pos + n) : 1u;
            }
            if (have) {
                node0 = c_node; noff0 = c_off; cur = c_node; cur_s0 = c_s0; coff = c_off; dist = 0;
                cur_long = c_len > 32u;
                cur64 = chunk(0);
                m0 = m1 = m2 = ~0ull;
                st = ST_WALK;
            } else if (pos >= lim) st = ST_ADV;
        } else if (st == ST_WALK) {
            // ---- one node of the walk (dfsRecursive, alignment.go:196-254, for a read that never has two neighbours to choose from) ----
            // everything the step may need is asked for at once: the record; for a walk that starts inside its node the graph bases from
            // `bases2`, else the extension of a long node; for a start position at offset <= 10 its 8-mer set
            const uint4 *q = reinterpret_cast<const uint4 *>(a.nodes + cur);
            const uint4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
            uint64_t gch[NCH];
#pragma unroll
            for (int c = 0; c < NCH; c++) gch[c] = 0;
            uint64_t l2b = ~0ull;
            if (dist == 0u && coff <= 10u && a.node_l2b) l2b = a.node_l2b[(size_t)cur * 11 + coff];
            if (coff != 0u) {
                const uint32_t bit = 2u * (cur_s0 + coff);
                const uint32_t *bp = a.bases2 + (bit >> 5);
                uint32_t x[2 * NCH + 1];
#pragma unroll
                for (int c = 0; c < 2 * NCH + 1; c++) x[c] = bp[c];
#pragma unroll
                for (int c = 0; c < NCH; c++) gch[c] = lean_funnel(x[2 * c], x[2 * c + 1], x[2 * c + 2], bit & 31u);
            } else if (cur_long) {
                const uint64_t *ep = a.ext[cur].b;
#pragma unroll
                for (int c = 1; c < NCH; c++) gch[c] = ep[c - 1];
            }
            const uint32_t seq_len = q0.y, dk = q0.z;
            if (coff == 0u) gch[0] = (uint64_t)q2.x | ((uint64_t)q2.y << 32);
            bool fail = (l2b & need) != need;                     // the start position cannot spell the view's first eight bases
            if (dk & kLeanNo) st = ST_DEFER;
            else if (!fail) {
                const uint32_t take = min(seq_len - coff, eff - dist);
                uint64_t diff = (gch[0] ^ cur64) & lean_lowmask((int)take);
                if (__ballot(take > 32u)) {
#pragma unroll
                    for (int c = 1; c < NCH; c++)
                        if (take > 32u * c) diff |= (gch[c] ^ chunk(dist + 32u * c)) & lean_lowmask((int)take - 32 * c);
                }
                fail = diff != 0ull;
                if (!fail) {
                    dist += take;
                    m0 &= (uint64_t)q2.z | ((uint64_t)q2.w << 32);
                    m1 &= (uint64_t)q3.x | ((uint64_t)q3.y << 32);
                    m2 &= (uint64_t)q3.z | ((uint64_t)q3.w << 32);
                    const bool any = (m0 | m1 | m2) != 0ull;
                    const uint32_t deg = dk & 7u;
                    if (dist == eff || deg == 0u) {               // :229-236 report the traversal
                        if (any) { emitted = true; st = ST_DONE; }
                        else fail = true;
                    } else if (!any) fail = true;                 // no path left: descendants cannot yield ids
                    else {
                        cur64 = chunk(dist);
                        const uint32_t nextb = (uint32_t)cur64 & 3u;
                        const uint32_t ed[4] = {q1.x, q1.y, q1.z, q1.w};
                        uint32_t hits = 0, pick = 0;
                        bool wild = false, plong = false;
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const uint32_t code = (dk >> (8 + 4 * e)) & 15u;
                            if ((uint32_t)e < deg) {
                                wild |= code == 4u;
                                if (code == nextb) { hits++; pick = ed[e]; plong = (dk >> (24 + e)) & 1u; }
                            }
                        }
                        if (wild || hits > 1u) st = ST_DEFER;   // an 'N' ahead, or a second neighbour to come back to: align_kernel's business
                        else if (hits == 0u) fail = true;
                        else { cur = pick; coff = 0; cur_long = plong; }
                    }
                }
            }
            if (fail && st == ST_WALK) st = pos >= lim ? ST_ADV : ST_GEN;   // the start position yields nothing: on with the hierarchy
        }
    }

#if defined(GROOT_WORK_COUNTERS) && GROOT_WORK_COUNTERS == 4
    if ((threadIdx.x & 63) == 0) {
        const unsigned long long t2 = wall_clock64();
        for (int i = 0; i < 9; i++) atomicAdd(&a.ctr->dbg[140 + i], lw[i]);
        atomicAdd(&a.ctr->dbg[149], lw_t1 - lw_t0);
        atomicAdd(&a.ctr->dbg[150], t2 - lw_t1);
        atomicAdd(&a.ctr->dbg[151], 1ull);
    }
#endif
    // ---- what the read leaves behind ----
    const bool fin = st == ST_DONE;
    if (slot < a.n_reads) a.defer[slot] = st == ST_DEFER ? 1 : 0;
    unsigned long long alns = 0, mapped = 0;
    if (fin) {
        a.trav_cnt[r] = emitted ? 1u : 0u;
        mapped = 1;                                               // boss.go:195-200
        if (emitted) {
            groot_trav t;
            t.read_id = a.first_read_id + r; t.graph_id = a.win_rec[w].graph; t.node = node0; t.offset = noff0;
            t.ord = 0;
            t.flags = (uint8_t)((rc ? GROOT_TRAV_RC : 0u) | (level == 3u ? GROOT_TRAV_START_CLIP : level == 4u ? GROOT_TRAV_END_CLIP : 0u) | GROOT_TRAV_FIRST);
            t.reserved = 0;
            a.trav_first[r] = t;
            a.mask_first[(size_t)r * PW] = m0; a.mask_first[(size_t)r * PW + 1] = m1; a.mask_first[(size_t)r * PW + 2] = m2;
            alns = (unsigned long long)(__popcll(m0) + __popcll(m1) + __popcll(m2));
        }
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

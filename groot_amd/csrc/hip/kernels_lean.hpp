#pragma once

#include "kernels_common.hpp"

namespace groot {

// ---------------------------------------------------------------------------------------------
// K3, first pass: the reads with at most four seed windows whose walks have at most two neighbours pending at a time -- all but a
// few in a hundred of an error-free batch.  A thread per read, in the processing order of the seed stage (neighbouring lanes hold
// reads of the same window: they walk the same nodes in step), no phase scheduling among the lanes of a wavefront:
//   * the read is staged in the lane's LDS slice at 2 bits per base (A=0 C=1 T=2 G=3, the code of the signature kernel), one strand
//     at a time (the slice is reverse-complemented in place when AlignRead's forward hierarchy has failed: graphminion.go:94);
//     32 bases of the current view come out of it with three ds_read_b32 and two v_alignbit;
//   * the graph side is 2 bits per base too: LeanNode holds a node's first 32 bases and LeanExt the next 224; a walk step is ONE trip
//     to memory whatever the node's length (record + extension, or record + `bases2` for a walk that starts inside a node, + the
//     start position's 8-mer set, all in flight together) -- align_kernel compares 8 ASCII bases at a time, 32 per step;
//   * AlignRead's hierarchy (alignment.go:13-110) is followed candidate by candidate exactly as align_kernel does it (same filters:
//     the seed stage's verdicts, the first min(8, ...) bases inside the node, the 8-mer set of the start position), so a read that
//     finishes here produces, bit for bit, what align_kernel would have produced for it;
//   * the graphMinion loop (graphminion.go:46-102) over the read's windows in ascending order, a graph done after its first alignment;
//     a node where two neighbours take the next base (dfsRecursive comes back to the second: alignment.go:242-252) leaves the second
//     on a stack of two entries per read in HBM;
//   * whatever does not fit -- more than four seed windows, a byte other than ACGT, an 'N' in the graph, three neighbours that take
//     the next base, a third pending neighbour -- is left to align_kernel: the read's slot is flagged, a stream compaction keeps the
//     flagged slots in processing order, and align_kernel handles the read from scratch.  What the first pass has written for such a
//     read by then are traversal records that align_kernel writes again, bit for bit, to the same places (ord 0: the read's own
//     slot; ord >= 1: the overflow list, placed by (read, ord)); everything that counts -- IncrementSubPath calls, mapped /
//     multimapped / alignments -- is kept in registers until the read is finished here.
// ---------------------------------------------------------------------------------------------

#ifndef GROOT_LEAN_WAVES
#define GROOT_LEAN_WAVES 5
#endif
constexpr int kLeanWaves = GROOT_LEAN_WAVES;
#ifndef GROOT_LEAN_MAX_ITER
#define GROOT_LEAN_MAX_ITER 96
#endif
constexpr uint32_t kLeanMaxIter = GROOT_LEAN_MAX_ITER;   // steps (candidate ranges, walk nodes) a read may take in the first pass

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
    // behind the read: the current window's seed, OffSet, l1_hi, cn_begin, cn_end, seed_s0, seed_len; the read's seed windows, ascending
    uint32_t *W = my + (a.lds_stride_dw - 11u);
    enum : int { W_SEED, W_OFF0, W_L1HI, W_CNB, W_CNE, W_S0, W_SLEN, W_WIN };
    // (a chain of dependent trips to memory with a few dozen instructions between two of them, beside the next batch's hashing kernels, which keep the
    // issue ports busy: raised, a walking wavefront goes first when it can go at all -- as in align_kernel)
#ifndef GROOT_LEAN_PRIO
#define GROOT_LEAN_PRIO -1
#endif
#if GROOT_LEAN_PRIO >= 0
    __builtin_amdgcn_s_setprio(GROOT_LEAN_PRIO);
#endif
    const uint32_t slot = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t n_todo = min(a.n_reads, (uint32_t)__builtin_amdgcn_readfirstlane((int)a.ctr->seeded_reads));
    // the seed stage ran out of slots or rows: the host grows them and runs the batch again (align_kernel returns at once, too)
    const bool stale = (a.ctr->flags & (kFlagSeedOverflow | kFlagQOverflow)) != 0;
#if defined(GROOT_WORK_COUNTERS) && GROOT_WORK_COUNTERS == 4
    const unsigned long long lw_t0 = wall_clock64();
#endif
    enum : uint32_t { ST_ADV, ST_GEN, ST_WALK, ST_SEED, ST_DONE, ST_DEFER, ST_IDLE };
    uint32_t st = ST_IDLE;
#if defined(GROOT_WORK_COUNTERS) && GROOT_WORK_COUNTERS == 4
    uint32_t why = 0;          // why the read was left to align_kernel: 1 more than four seeds, 3 byte > 'T' / length, 4 a byte other than ACGT, 5 a window with an 'N', 6 a node with an 'N', 7 an 'N' ahead, 8 three neighbours, 9 a third pending neighbour
#define LEAN_WHY(x) (why = (x))
#else
#define LEAN_WHY(x) ((void)0)
#endif
    uint32_t r = 0, len = 0, cnt = 0, vbits = 0;
    uint32_t g = kEmpty;          // graph of the current window
    if (slot < n_todo && !stale) {
        r = a.perm[slot];
        // (asked for together with the record: the read's codes, if the signature kernel left them)
        uint4 pk0 = make_uint4(0, 0, 0, 0), pk1 = pk0, pk2 = pk0, pk3 = pk0;
        if (a.packed) {
            const uint4 *pk = a.packed + (size_t)r * a.packed_q;
            pk0 = pk[0]; pk1 = pk[1];
            if (a.packed_q > 2u) { pk2 = pk[2]; pk3 = pk[3]; }
        }
        uint4 ra, rb;
        load32(a.read_rec + r, ra, rb);
        const uint32_t sc = ra.w;
        len = ra.z & ~kRecPacked;
        cnt = sc & (kRecSplit - 1u);
        vbits = (sc >> 24) & 0x3Fu;
        st = ST_ADV;
        // at most four seed windows (they travel in the record), no byte > 'T' (RevComplement would panic on it: align_kernel counts that), a read the slice holds
        if (cnt == 0u || cnt > 4u || (sc & kRecSplit) || (sc >> 31) || len > a.max_len || len < 12u) { st = ST_DEFER; LEAN_WHY(cnt > 4u ? 1 : 3); }
        else {
            // the windows in ascending order (graphminion.go:50 sorts the seeds; the build's canonical order is the window id)
            uint32_t s0 = rb.x, s1 = cnt > 1u ? rb.y : kEmpty, s2 = cnt > 2u ? rb.z : kEmpty, s3 = cnt > 3u ? rb.w : kEmpty;
            { uint32_t t; if (s0 > s1) { t = s0; s0 = s1; s1 = t; } if (s2 > s3) { t = s2; s2 = s3; s3 = t; } if (s0 > s2) { t = s0; s0 = s2; s2 = t; }
              if (s1 > s3) { t = s1; s1 = s3; s3 = t; } if (s1 > s2) { t = s1; s1 = s2; s2 = t; } }
            const uint8_t *p = a.seq + ((uint64_t)ra.x | ((uint64_t)ra.y << 32));
            uint4 wa, wb;
            load32(a.win_rec + s0, wa, wb);
            // (one trip: the first window's record and every window's flag)
            const uint32_t ok = (uint32_t)a.win_ok[s0] & a.win_ok[cnt > 1u ? s1 : s0] & a.win_ok[cnt > 2u ? s2 : s0] & a.win_ok[cnt > 3u ? s3 : s0];
            // ---- stage the read at 16 bases per dword ----
            my[0] = 0; my[1] = 0;
            uint32_t bad = 0;
            const uint32_t nd = (len + 15u) >> 4;
            if ((ra.z & kRecPacked) && a.packed) {             // the signature kernel left its codes (all ACGT)
                const uint32_t c[16] = {pk0.x, pk0.y, pk0.z, pk0.w, pk1.x, pk1.y, pk1.z, pk1.w, pk2.x, pk2.y, pk2.z, pk2.w, pk3.x, pk3.y, pk3.z, pk3.w};
#pragma unroll
                for (int i = 0; i < 16; i++)
                    if ((uint32_t)i < nd) my[2 + i] = c[i];
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
            g = wa.x;
            W[W_SEED] = wa.y; W[W_OFF0] = wa.z; W[W_L1HI] = wa.w;
            W[W_CNB] = wb.x; W[W_CNE] = wb.y; W[W_S0] = wb.z; W[W_SLEN] = wb.w;
            W[W_WIN] = s0; W[W_WIN + 1] = s1; W[W_WIN + 2] = s2; W[W_WIN + 3] = s3;
            if (bad || !ok) { st = ST_DEFER; LEAN_WHY(bad ? 4 : 5); }
        }
    }

    // ---- graphMinion loop (graphminion.go:46-102) ----
    uint32_t si = 1;              // seed windows taken so far (the first one's record came with the read)
    uint32_t done_graph = kEmpty, n_graphs = 1, ord = 0, calls = 1;   // calls: bit i = IncrementSubPath was called for window i (:67)
    uint32_t alns = 0;
    // ---- the view of the read a hierarchy level works on: orientation, clip, effective length (alignment.go:72-103) ----
    uint32_t rc = 0, level = 0, eff = len;
    uint32_t sbit = 64;           // bit of the slice where the view's first base sits
    // view bases [d, d + 32) at 2 bits each (bits past the view's end are don't-care)
    auto chunk = [&](uint32_t d) -> uint64_t { return lean_bits64(my, sbit + 2u * d); };
    // the slice becomes the reverse complement of what it holds, padded to whole dwords at the other end: dword k <- revcomp16(dword nd-1-k)
    auto flip = [&]() {
        const uint32_t nd = (len + 15u) >> 4;
        for (uint32_t k2 = 0; 2u * k2 < nd; k2++) {
            const uint32_t x = my[2 + k2], y = my[1 + nd - k2];
            my[2 + k2] = lean_revcomp16(y);
            my[1 + nd - k2] = lean_revcomp16(x);
        }
        rc ^= 1u;
    };
    uint32_t p16 = 0;             // first eight bases of the view
    uint64_t need = 0;            // their two bits in a start position's 8-mer set
    // candidate cursor: level 1 -- pos = next offset of the seed node, lim = l1_hi; level 2 -- pos = ContainedNodes entry, lim = cn_end,
    // sub = next offset in the entry; levels 3, 4 -- pos = 0 before the single start position, lim = 1
    uint32_t pos = 0, sub = 0, lim = 0;
    // walk
    uint32_t node0 = 0, noff0 = 0, cur = 0, cur_s0 = 0, coff = 0, dist = 0;
    bool cur_long = false;
    uint64_t cur64 = 0, m0 = 0, m1 = 0, m2 = 0;
    uint32_t emitted = 0, sp = 0; // traversals of this AlignRead call; pending neighbours on the read's stack
    uint4 *stk = a.stk + (size_t)slot * 4;

    // verdict of the seed stage on the read's FIRST seed window, current orientation: bit 0 levels 1-2, bit 1 level 3, bit 2 level 4
    auto verdict = [&](uint32_t bit) -> bool { return si == 1u && (((vbits >> (rc ? 3 : 0)) >> bit) & 1u); };
    // move to level lv / the next one that has a candidate range / the other orientation; ST_SEED when AlignRead found nothing for the window
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
            if (rc == 0) { flip(); lv = 1; continue; }
            st = ST_SEED;                                          // neither orientation: the read's next window (the slice is flipped back there)
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
    if (GROOT_LEAN_PROBE == 1 && st <= ST_SEED) st = ST_DEFER;
#endif

    uint32_t iters = 0;
    while (__ballot(st <= ST_SEED)) {
        // a read that is still at it after kLeanMaxIter steps is one of the hard ones (it fails through the hierarchy, candidate after candidate, a trip
        // to memory each): align_kernel has the means for those (cooperative scans, fork / join) and a wavefront here would wait for it
        if (st <= ST_SEED && ++iters > kLeanMaxIter) { st = ST_DEFER; LEAN_WHY(10); }
        if (st == ST_SEED) {
            // ---- the read's next seed window, if any (graphminion.go:52-100) ----
            if (si >= cnt) st = ST_DONE;
            else {
                const uint32_t w = W[W_WIN + si];
                uint4 wa, wb;
                load32(a.win_rec + w, wa, wb);
                si++;
                if (wa.x != g) { g = wa.x; n_graphs++; }
                if (g != done_graph) {                             // :96-98 a graph is done after its first alignment
                    calls |= 1u << (si - 1u);                      // :67 IncrementSubPath
                    W[W_SEED] = wa.y; W[W_OFF0] = wa.z; W[W_L1HI] = wa.w;
                    W[W_CNB] = wb.x; W[W_CNE] = wb.y; W[W_S0] = wb.z; W[W_SLEN] = wb.w;
                    if (rc) flip();                                // the minion's copy of the read is forward again (two flips, :94)
                    level = 0; emitted = 0;
                    st = ST_ADV;
                }
            }
        }
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
        }
        if (st == ST_WALK) {
            // ---- one node of the walk (dfsRecursive, alignment.go:196-254) ----
            // everything the step may need is asked for at once: the record; for a walk that starts inside its node the graph bases from
            // `bases2`, else the extension of a long node; for a start position at offset <= 10 its 8-mer set
            const uint4 *q = reinterpret_cast<const uint4 *>(a.nodes + cur);
#if defined(GROOT_LEAN_PROBE) && GROOT_LEAN_PROBE == 2   // (tools: one 16-byte load of four less per step -- what do the loads cost?  wrong results on graphs of more than 64 paths)
            const uint4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = make_uint4(~0u, ~0u, ~0u, ~0u);
#else
            const uint4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
#endif
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
            bool over = (l2b & need) != need;                     // this branch of the walk is over (here: the start position cannot spell the view's first eight bases)
            if (dk & kLeanNo) { st = ST_DEFER; LEAN_WHY(6); }
            else if (!over) {
                const uint32_t take = min(seq_len - coff, eff - dist);
                uint64_t diff = (gch[0] ^ cur64) & lean_lowmask((int)take);
                if (__ballot(take > 32u)) {
#pragma unroll
                    for (int c = 1; c < NCH; c++)
                        if (take > 32u * c) diff |= (gch[c] ^ chunk(dist + 32u * c)) & lean_lowmask((int)take - 32 * c);
                }
                over = diff != 0ull;
                if (!over) {
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
                            t.ord = (uint16_t)ord;
                            t.flags = (uint8_t)((rc ? GROOT_TRAV_RC : 0u) | (level == 3u ? GROOT_TRAV_START_CLIP : level == 4u ? GROOT_TRAV_END_CLIP : 0u) | (emitted == 0u ? GROOT_TRAV_FIRST : 0u));
                            t.reserved = 0;
                            if (ord == 0u) {
                                a.trav_first[r] = t;
                                a.mask_first[(size_t)r * PW] = m0; a.mask_first[(size_t)r * PW + 1] = m1; a.mask_first[(size_t)r * PW + 2] = m2;
                            } else {
                                const uint32_t shard = blockIdx.x & (kOvfShards - 1);
                                const uint32_t at = atomicAdd(&a.ovf_cnt[shard], 1u);
                                if (at < a.ovf_cap) {
                                    const size_t o = (size_t)shard * a.ovf_cap + at;
                                    a.ovf_trav[o] = t;
                                    a.ovf_mask[o * PW] = m0; a.ovf_mask[o * PW + 1] = m1; a.ovf_mask[o * PW + 2] = m2;
                                } else atomicOr(&a.ctr->flags, kFlagOvfOverflow);
                            }
                            ord++; emitted++;
                            alns += (uint32_t)(__popcll(m0) + __popcll(m1) + __popcll(m2));
                        }
                        over = true;
                    } else if (!any) over = true;                 // no path left: descendants cannot yield ids
                    else {
                        // :242-252 neighbours in OutEdges order; one whose first base differs from the read's next base dies in its first comparison
                        cur64 = chunk(dist);
                        const uint32_t nextb = (uint32_t)cur64 & 3u;
                        const uint32_t ed[4] = {q1.x, q1.y, q1.z, q1.w};
                        uint32_t hits = 0, pick = 0, alt = 0;
                        bool wild = false;
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const uint32_t code = (dk >> (8 + 4 * e)) & 15u;
                            if ((uint32_t)e < deg) {
                                wild |= code == 4u;
                                if (code == nextb) {
                                    const uint32_t nx = ed[e] | (((dk >> (24 + e)) & 1u) << 31);    // bit 31: a long node
                                    if (hits == 0u) pick = nx; else alt = nx;
                                    hits++;
                                }
                            }
                        }
                        if (wild || hits > 2u || (hits == 2u && sp == 2u)) { st = ST_DEFER; LEAN_WHY(wild ? 7 : hits > 2u ? 8 : 9); }   // align_kernel's business
                        else if (hits == 0u) over = true;
                        else {
                            if (hits == 2u) {                      // the second neighbour stays pending
                                stk[2 * sp] = make_uint4(alt, dist, (uint32_t)m0, (uint32_t)(m0 >> 32));
                                stk[2 * sp + 1] = make_uint4((uint32_t)m1, (uint32_t)(m1 >> 32), (uint32_t)m2, (uint32_t)(m2 >> 32));
                                sp++;
                            }
                            cur = pick & 0x7FFFFFFFu; coff = 0; cur_long = pick >> 31;
                        }
                    }
                }
            }
            if (over && st == ST_WALK) {
                if (sp) {                                          // resume at the newest pending neighbour
                    sp--;
                    const uint4 h0 = stk[2 * sp], h1 = stk[2 * sp + 1];
                    cur = h0.x & 0x7FFFFFFFu; cur_long = h0.x >> 31; coff = 0; dist = h0.y;
                    m0 = (uint64_t)h0.z | ((uint64_t)h0.w << 32);
                    m1 = (uint64_t)h1.x | ((uint64_t)h1.y << 32);
                    m2 = (uint64_t)h1.z | ((uint64_t)h1.w << 32);
                    cur64 = chunk(dist);
                } else if (emitted) { done_graph = g; st = ST_SEED; }      // performAlignment is over with an alignment for (read, graph)
                else st = pos >= lim ? ST_ADV : ST_GEN;                    // the start position yields nothing: on with the hierarchy
            }
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
    for (uint32_t y = 1; y <= 10; y++) {
        const unsigned long long b = __ballot(st == ST_DEFER && why == y);
        if (b && (threadIdx.x & 63) == 0) atomicAdd(&a.ctr->dbg[160 + y], (unsigned long long)__popcll(b));
    }
    { const unsigned long long b = __ballot(st == ST_DONE && !ord); if (b && (threadIdx.x & 63) == 0) atomicAdd(&a.ctr->dbg[160], (unsigned long long)__popcll(b)); }
#endif
    // ---- what the read leaves behind ----
    const bool fin = st == ST_DONE;
    if (slot < n_todo) a.defer[slot] = st == ST_DEFER ? 1 : 0;   // (slots from n_todo on hold reads without seeds: LeanLeft)
    unsigned long long n_alns = 0, mapped = 0, multimapped = 0;
    if (fin) {
        a.trav_cnt[r] = ord;
        mapped = 1;                                               // boss.go:195-200
        multimapped = n_graphs > 1u ? 1 : 0;
        n_alns = alns;
    }
    if (a.update_weights) {                                       // graphminion.go:67 IncrementSubPath for the windows it was called on
        // neighbouring lanes mostly hold reads of the same first window: one atomic per RUN of equal cells among the lanes that finished here
        const uint32_t qrow = fin ? a.q_row[len - a.k + 1u] : 0u;
        const uint64_t cell = (uint64_t)qrow * a.n_windows + (fin ? W[W_WIN] : 0u);
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
        if (fin && (calls >> 1))                                  // (the first window is always called: bit 0)
            for (uint32_t i = 1; i < cnt; i++)
                if ((calls >> i) & 1u) atomicAdd(&a.attempts[(uint64_t)qrow * a.n_windows + W[W_WIN + i]], 1u);
    }
    n_alns = block_sum(n_alns, red);
    mapped = block_sum(mapped, red);
    multimapped = block_sum(multimapped, red);
    if (threadIdx.x == 0) {
        if (n_alns) atomicAdd(&a.ctr->alignments, n_alns);
        if (a.update_weights && mapped) atomicAdd(&a.ctr->mapped, mapped);
        if (a.update_weights && multimapped) atomicAdd(&a.ctr->multimapped, multimapped);
        if (mapped) atomicAdd(&a.ctr->lean_reads, (unsigned int)mapped);
    }
}

} // namespace groot

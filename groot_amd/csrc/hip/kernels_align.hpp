#pragma once

#include "kernels_common.hpp"

namespace groot {

// ---------------------------------------------------------------------------------------------
// K3
// ---------------------------------------------------------------------------------------------

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
enum : uint32_t { PH_FETCH, PH_SCAN, PH_DFS, PH_WAIT, PH_DONE, PH_FORK, PH_FORKWAIT, PH_JOIN };
// A read whose window of a graph has failed through the whole hierarchy in both orientations and that brings more windows of that graph
// asks the idle lanes of its wavefront (PH_FORK) to try those, a window each, as a dry run (PH_FORKWAIT while they do; they end in PH_JOIN)
#ifndef GROOT_FORK_MIN
#define GROOT_FORK_MIN 3
#endif
constexpr uint32_t kForkMin = GROOT_FORK_MIN;   // windows left in the read's list from which it asks
constexpr uint32_t kClsHelper = 0x1000u;        // cls bit 12: the lane tries a window for another lane's read (no output, no counts)
constexpr uint32_t kCoopMin = 6;       // contained nodes from which level 2 of AlignRead asks its wavefront for a cooperative scan
constexpr uint32_t kCoopWant = 0xFFFFFFFEu;
#ifndef GROOT_COOP_MAX_ASK
#define GROOT_COOP_MAX_ASK 8
#endif
constexpr uint32_t kCoopMaxAsk = GROOT_COOP_MAX_ASK;     // lanes of a wavefront that may ask in the same iteration
#ifndef GROOT_TL_LATE_US
#define GROOT_TL_LATE_US 3000
#endif
#ifndef GROOT_WAVE_CHUNK
#define GROOT_WAVE_CHUNK 128
#endif
constexpr uint32_t kWaveChunk = GROOT_WAVE_CHUNK;   // (rounds per trip to the cursor = kWaveChunk / 64; one when the reads do not march in step)   // consecutive slots a wave takes before asking for more (multiple of 64)

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
template <int PW, bool LDSR>
__global__ __launch_bounds__(kBlock, PW > 3 ? kAlignWavesWide : kAlignWaves) void align_kernel(AlignArgs a)
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
#ifndef GROOT_ALIGN_PRIO
#define GROOT_ALIGN_PRIO 3
#endif
    // The walk is a chain of dependent steps with a few hundred instructions between two trips to memory, and it runs beside the next
    // batch's hashing kernels, which keep every SIMD's issue port busy: at equal priority each instruction of a walking wavefront waits its turn
    // behind five hashing wavefronts.  Raised, the walking wavefronts -- mostly waiting for memory anyway -- go first when they can go at all
    // (worth a few per cent: the 5-6 us a wave iteration takes are its own dependent instructions and trips to memory).
#if GROOT_ALIGN_PRIO >= 0
    __builtin_amdgcn_s_setprio(GROOT_ALIGN_PRIO);
#endif
    unsigned long long alns = 0, mapped = 0, multimapped = 0, panics = 0;
#ifdef GROOT_WORK_COUNTERS
    uint32_t ev = 0;                                       // events of this lane in the current wave iteration
    uint32_t wc_iter = 0, wc_round0 = 0;                   // wave iterations so far / at the last refill
    unsigned long long wc_t[3] = {0, 0, 0};                // wall-clock ticks (100 MHz) per phase, wave-uniform
    uint32_t wc_n[3] = {0, 0, 0};                          // steps per phase
    uint32_t wc_n0[3] = {0, 0, 0};                         // ... when the lane's current read began
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
    const uint32_t n_todo = nv + (a.perm ? min(a.n_reads, (uint32_t)__builtin_amdgcn_readfirstlane((int)(a.n_perm ? *a.n_perm : a.ctr->seeded_reads))) : a.n_reads);   // (scalar: it bounds every refill)
    // Lanes per round.  A round lasts as long as its slowest read, so when there are fewer reads than 64 per resident wavefront
    // (most of the batch was answered from the outcome table: what is left are the hard reads) the rounds are made smaller
    // and spread over all wavefronts: the launch then ends with the slowest read instead of the slowest sum of rounds.
    // Reads that do not march in step (mixed read lengths: a.head_lanes != 0) get rounds of 32, one round per trip to the cursor: a round
    // of 64 of them lasts as long as their steps laid end to end, and a wavefront that took 128 such slots just before the cursor ran dry
    // was still walking at 5.9 ms when all others had ended by 2.5 (timeline build, 8 M reads of 75..150 bases: align stage 5.9 -> 3.4 ms)
#ifndef GROOT_MIXED_ROUND_LANES
#define GROOT_MIXED_ROUND_LANES 32
#endif
    uint32_t U = a.head_lanes ? (uint32_t)GROOT_MIXED_ROUND_LANES : 64u;
    if (a.round_lanes) U = a.round_lanes;
    else
        while (U > 1u && n_todo < U * (gridDim.x * (uint32_t)(kBlock / 64))) U >>= 1;
    const uint32_t n_rounds = (n_todo + U - 1u) / U;
    // (odd on purpose: with an even count the two-round chunks behind the head start at multiples of 128 slots and the kernel is
    // 6 % slower -- measured both ways, cause not established)
    const uint32_t head_rounds = (n_rounds >> 3) | 1u;
    // The head of the order holds the longest walks.  When the reads of a batch do not march in step (a.head_lanes != 0: mixed
    // read lengths) a round of 64 of them lasts as long as their steps laid end to end -- one such round was a quarter of the
    // launch --, so the head is handed out in rounds of a.head_lanes reads; the tail keeps full rounds.
    const uint32_t Uh = a.head_lanes ? min(a.head_lanes, U) : U;
    const uint32_t head_slots = min(head_rounds * U, n_todo);
    const uint32_t head_small = (head_slots + Uh - 1u) / Uh;
    uint32_t chunk_len = 0, chunk_base = 0;                // slots in the current chunk, its first slot (wave-uniform)
    bool head_done = head_small == 0;                      // wave-uniform
    const uint32_t chunk_rounds = a.head_lanes ? 1u : kWaveChunk / 64u;
    auto take_chunk = [&]() {
        uint32_t c = 0, tail = 0;
        if ((threadIdx.x & 63) == 0) {
            if (!head_done) c = atomicAdd(a.ovf_cnt + kOvfShards, 1u);
            if (head_done || c >= head_small) {
                tail = 1;
                c = atomicAdd(a.ovf_cnt + kOvfShards + 1, chunk_rounds);
            }
        }
        c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
        tail = (uint32_t)__builtin_amdgcn_readfirstlane((int)tail);
        if (tail) {
            head_done = true;
            // (saturating: past the end the base only has to be >= n_todo)
            const unsigned long long b = (unsigned long long)head_slots + (unsigned long long)c * U;
            chunk_base = b > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)b;
            chunk_len = chunk_rounds * U;
        } else {
            chunk_base = c * Uh;
            chunk_len = min(Uh, head_slots - chunk_base);
        }
    };
    take_chunk();
#if defined(GROOT_WORK_COUNTERS) && GROOT_WORK_COUNTERS == 3
    const unsigned long long tl_start = wall_clock64();    // timeline of the launch: when do the wavefronts end, and how long was their last round?
    unsigned long long tl_round = tl_start, tl_coop = 0, tl_fork = 0, tl_read0 = 0;
    uint32_t tl_it0 = 0, tl_st0 = 0;
    uint32_t tl_rounds = 0, tl_ncoop = 0;
#endif
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
    // the view of the read a level works on follows from the level: 3 clips the first base, 4 the last (alignment.go:72-103)
    auto clip_lo = [&]() -> uint32_t { return level == 3u ? 1u : 0u; };
    auto eff = [&]() -> uint32_t { return len - (level >= 3u ? 1u : 0u); };
    // upper bound of the read bases any branch of level 1's DFS from (seed, OffSet) has matched, per orientation: level 4 walks the same
    // nodes from the same position with the last base clipped (alignment.go:87-103), so it can only succeed where level 1 got to len - 1
    uint32_t reach = 0;
    // fork / join: a helper keeps its owner's lane (bits 0-5), the phase it goes back to (bits 8-10) and its result (bits 12-13: 0 failed,
    // 1 aligned, 2 the window belongs to another graph); an owner keeps 1 + the largest window of its list known to fail
    uint32_t fk = 0;
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
        if (!LDSR) return read_chunk(p, len, rc, clip_lo(), d);
        const uint32_t i = d + clip_lo();
        const uint32_t o = rc ? len - i : 8u + i;
        const uint32_t *wp = my_lds + (o >> 2);
        const uint32_t x0 = wp[0], x1 = wp[1], x2 = wp[2];
        const uint32_t sh = (o & 3u) * 8u;
        const uint64_t v = (uint64_t)__funnelshift_r(x0, x1, sh) | ((uint64_t)__funnelshift_r(x1, x2, sh) << 32);
        return rc ? revcomp8(v) : v;
    };
    auto set_view = [&]() { pre8 = dfs_chunk(0); };           // (after rc / level have been set)
    auto scan_range = [&](uint32_t node, uint32_t s0, uint32_t nlen, uint32_t from, uint32_t to) {
        sc_node = node; sc_s0 = s0; sc_len = nlen; sc_pos = from; sc_end = to;
    };
    // verdict of the seed stage for the current orientation (f = kRecNo12F / kRecNo3F / kRecNo4F)
    auto verdict = [&](uint32_t f) -> bool { return (cls & 0x40u) && ((cls >> (rc ? 3 : 0)) & (f >> 24)); };
    // 1. seed offset shuffling (alignment.go:34-45).  Returns true when levels 1 and 2 cannot start anywhere for this
    // orientation (prefix tables): the ranges are left empty and the caller moves on through the hierarchy.
    auto start_orientation = [&](uint32_t t) -> bool {
        rc = t; level = 1; reach = 0;
        set_view();
        scan_range(seed, seed_s0, seed_len, off0, l1_hi);
        phase = PH_SCAN;
        bool no;
        if (cls & 0x40u) no = verdict(kRecNo12F);
        else no = ix.win_prefix && prefix_absent(ix.win_prefix + (size_t)w * kPrefixWords, pre8, eff() >= 12 ? dfs_chunk(8) : 0, eff());   // (null: the tables are still being built, groot_hip_open_flags)
        if (no) { level = 2; cn_cur = cn_end; sc_pos = sc_end = 0; sc_node = kEmpty; }
        return no;
    };
    // the current scan range is used up: move through the hierarchy until a non-empty range or the end
    // (no loads in here: level 2 reads DeviceIndex::cn_pre, whose next entry is the next address)
    auto next_range = [&]() {
        for (;;) {
            if (level == 1) {                               // 2. seed node shuffling (:47-70): offsets 0..10 of every contained node
                level = 2; cn_cur = cn_begin; sc_pos = 0; sc_end = 11;
                // (sc_node / sc_s0 / sc_len are free during level 2: a window with many contained nodes asks the wavefront to look at 64 of
                // them at once -- sc_node = kCoopWant, then the first entry of the block; sc_s0 | sc_len << 32 = the entries worth a visit)
                sc_node = cn_end - cn_cur >= kCoopMin ? kCoopWant : kEmpty;
                if (cn_cur < cn_end) return;
            } else if (level == 2) {
                level = 3;                                  // 3. hard clip the first base (:72-85)
                if (off0 >= seed_len) { level = 4; continue; }   // :199-201 holds for levels 3 and 4 alike
                if (verdict(kRecNo3F)) continue;
                set_view();
                scan_range(seed, seed_s0, seed_len, off0, off0 + 1);
                return;
            } else if (level == 3) {
                level = 4;                                  // 4. hard clip the last base (:87-103)
                if (off0 >= seed_len) continue;
                if (verdict(kRecNo4F)) continue;            // its single start position fails the first comparison
                // Level 4 is level 1's first DFS again, one base shorter: a branch that died (mismatch, no path left, no neighbour for the next
                // base) before it had matched len - 1 bases dies the same way here, and a sink it reached would have been level 1's alignment.
                // (start positions the filters turned down matched fewer than 12 bases; len >= 14 keeps both views above that)
#ifndef GROOT_NO_LEVEL4_RULE                               // (tools/cross_check.py builds the kernel without the rule and without the fork to compare at scale)
                if (len >= 14u && reach + 1u < len) continue;
#endif
                set_view();
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
                } else if (cls & kClsHelper) { have_read = false; phase = PH_JOIN; }   // (dry run: result 0 = the window fails)
                else if ((cls & 0x100u) && (cnt > 4u || (cls & 0x200u)) && sd0 + kForkMin <= cnt) phase = PH_FORK;   // more windows wait in the read's ascending list
                else phase = PH_FETCH;                       // both orientations failed: next mapping
                return;
            }
        }
    };
    // which of the start offsets [0, npos) of a base string (w0, w1, w2 = its first 24 bytes; `room` of them lie inside the node) can
    // spell the first min(4, eff) read bases?  0x80 per surviving offset: offsets 0..7 in lo, 8..15 in hi
    auto filter16x = [](const uint64_t pre, const uint32_t ef, const uint64_t w0, const uint64_t w1, const uint64_t w2, const int npos, const int room,
                        uint64_t &c_lo, uint64_t &c_hi) {
        c_lo = low_bytes(npos); c_hi = low_bytes(npos - 8);
        const uint32_t kf = min(4u, ef);
#pragma unroll
        for (int b = 0; b < 4; b++) {
            if ((uint32_t)b >= kf) break;
            const unsigned rb = (unsigned)(pre >> (8 * b)) & 0xFF;
            // positions whose base b lies inside the node must match it; past the node end the DFS decides
            const uint64_t need_lo = low_bytes(room - b), need_hi = low_bytes(room - b - 8);
            c_lo &= match_or_n(window8(w0, w1, b), rb) | ~need_lo;
            c_hi &= match_or_n(window8(w1, w2, b), rb) | ~need_hi;
        }
    };
    auto filter16 = [&](const uint64_t w0, const uint64_t w1, const uint64_t w2, const int npos, const int room, uint64_t &c_lo, uint64_t &c_hi) {
        filter16x(pre8, eff(), w0, w1, w2, npos, room, c_lo, c_hi);
    };
    auto first16 = [](const uint64_t c_lo, const uint64_t c_hi) -> uint32_t {
        if (c_lo) return (uint32_t)__builtin_ctzll(c_lo) >> 3;
        if (c_hi) return 8u + ((uint32_t)__builtin_ctzll(c_hi) >> 3);
        return 16u;
    };
    // can no DFS from (node, off) spell the first eight bases of the current view?  (DeviceIndex::node_l2b; sound: false when in doubt)
    auto cannot_start = [&](uint32_t node, uint32_t off) -> bool {
        if (!ix.node_l2b || off > 10u || eff() < 8u) return false;
        const int c8 = kmer8_code(pre8);
        if (c8 < 0) return false;
        const uint64_t need = l2_bloom_bits((uint32_t)c8);
        return (ix.node_l2b[(size_t)node * 11 + off] & need) != need;
    };
    auto begin_dfs = [&](uint32_t node, uint32_t off) {
        node0 = node; noff0 = off; cur = node; coff = off; dist = 0; sp = 0; emitted = 0;
        cur8 = pre8;
#pragma unroll
        for (int i = 0; i < PW; i++) mask[i] = ~0ULL;
        phase = PH_DFS;
    };

    for (;;) {
#if defined(GROOT_WORK_COUNTERS) && GROOT_WORK_COUNTERS == 1     // (2 = phase timing only: the tally itself costs time)
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
        // ---- cooperative level-2 scans: all 64 lanes look at one lane's contained nodes, an entry each ----
        // A read that fails at a window of a variant-dense region walks 50-65 contained nodes of a few bases, eleven start offsets each, in
        // both orientations: two entries per SCAN step and a step per start position that survives the 4-base filter -- it was the slowest
        // read of its batch (130-500 SCAN steps), and the launch lasts as long as it does.  Here every lane, whatever it is doing for
        // its own read, applies the requester's filters (4 bases, the in-node bases, the start position's 8-mer set) to entry
        // cn_cur + lane; the ballot of the entries with a start position left is all the requester visits afterwards.
        // (when many lanes ask at once -- reads that march in step -- each scanning its own entries is the parallel way: they are told so)
#if defined(GROOT_WORK_COUNTERS) && GROOT_WORK_COUNTERS == 3
        const unsigned long long tl_c0 = wall_clock64();
#endif
        unsigned long long bh = __ballot(phase == PH_SCAN && level == 2u && sc_node == kCoopWant);
#if defined(GROOT_WORK_COUNTERS) && GROOT_WORK_COUNTERS == 3
        tl_ncoop += (uint32_t)__popcll(bh);
#endif
        if (__popcll(bh) > (int)kCoopMaxAsk) {
            if (phase == PH_SCAN && level == 2u && sc_node == kCoopWant) sc_node = kEmpty;
            bh = 0;
        }
        for (; bh; bh &= bh - 1) {
            const int L = __ffsll(bh) - 1;
            const uint32_t q0 = __shfl(cn_cur, L), q1 = __shfl(cn_end, L), qeff = __shfl(eff(), L);
            const uint64_t qpre = (uint64_t)(uint32_t)__shfl((int)(uint32_t)pre8, L) | ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(pre8 >> 32), L) << 32);
            const int c8 = qeff >= 8u && ix.node_l2b ? kmer8_code(qpre) : -1;
            const uint64_t need8 = c8 >= 0 ? l2_bloom_bits((uint32_t)c8) : 0ull;
            const uint32_t e = q0 + (threadIdx.x & 63u);
            bool has = false;
            if (e < q1) {
                const uint4 *ep = ix.cn_pre + 2 * (size_t)e;
                const uint4 a0 = ep[0], a1 = ep[1];
                const uint64_t w0 = (uint64_t)a0.x | ((uint64_t)a0.y << 32), w1 = (uint64_t)a0.z | ((uint64_t)a0.w << 32), w2 = (uint64_t)a1.x | ((uint64_t)a1.y << 32);
                const uint32_t node = a1.z, nlen = a1.w;
                uint64_t c_lo, c_hi;
                filter16x(qpre, qeff, w0, w1, w2, (int)min(nlen, 11u), (int)nlen, c_lo, c_hi);
                for (uint32_t j = first16(c_lo, c_hi); j < 16u; j = first16(c_lo, c_hi)) {
                    if (j < 8) c_lo &= ~(0x80ull << (8 * j)); else c_hi &= ~(0x80ull << (8 * (j - 8)));
                    const uint64_t g8 = j < 8 ? window8(w0, w1, j) : window8(w1, w2, j - 8);
                    if (!prefix_ok(g8, qpre, min(min(nlen - j, qeff), 8u))) continue;
                    if (c8 >= 0 && (ix.node_l2b[(size_t)node * 11 + j] & need8) != need8) continue;
                    has = true;
                    break;
                }
            }
            const unsigned long long bits = __ballot(has);
            if ((int)(threadIdx.x & 63u) == L) { sc_node = q0; sc_s0 = (uint32_t)bits; sc_len = (uint32_t)(bits >> 32); }
        }
#if defined(GROOT_WORK_COUNTERS) && GROOT_WORK_COUNTERS == 3
        const unsigned long long tl_c1 = wall_clock64();
        tl_coop += tl_c1 - tl_c0;
#endif
        // ---- fork: the windows a failing read has left in its list are tried by the wavefront's idle lanes, one each, as dry runs ----
        // graphminion.go:46-102 takes a read's windows of a graph one after the other until one aligns; for a read that fails them all --
        // 13 windows of one gene family, five hierarchy levels, two orientations -- that chain IS the launch's duration, while most lanes
        // of its wavefront have nothing to do (rounds of the head of the order fill 16 lanes; the tail of the launch fewer).  The
        // helpers only answer "would AlignRead succeed on window j" (no records, no counts); the owner then passes over the windows known to
        // fail with one FETCH step each (IncrementSubPath is still called for them, in order: :60-67) and aligns the first that succeeds itself.
        if (__ballot(phase >= PH_FORK)) {                       // (one ballot per iteration when nobody forks)
        for (unsigned long long bq = __ballot(phase == PH_FORK); bq; bq &= bq - 1) {
            const int L = __ffsll(bq) - 1;
            const unsigned long long idle = __ballot(phase == PH_WAIT || phase == PH_DONE);
            const uint32_t q_lo = __shfl(sd0, L), q_cnt = __shfl(cnt, L);
            const uint32_t nh = min((uint32_t)__popcll(idle), q_cnt - q_lo);
            if (nh < 2u) {                                     // nobody to help: carry on alone
                if ((int)(threadIdx.x & 63u) == L) phase = PH_FETCH;
                continue;
            }
            const uint32_t q_r = __shfl(r, L), q_len = __shfl(len, L), q_g = __shfl(g, L), q_hb = __shfl(high_byte, L);
            const uint32_t q_plo = __shfl((uint32_t)(uintptr_t)p, L), q_phi = __shfl((uint32_t)((uintptr_t)p >> 32), L);
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
            if ((phase == PH_WAIT || phase == PH_DONE) && rank < nh) {
                fk = (uint32_t)L | (phase << 8);
                r = q_r; len = q_len; high_byte = q_hb;
                p = reinterpret_cast<const uint8_t *>((uintptr_t)q_plo | ((uintptr_t)q_phi << 32));
                if (LDSR) {                                    // the owner's staged read (one LDS address for all helpers: a broadcast)
                    const uint32_t *src = lds_reads + (size_t)(threadIdx.x - (threadIdx.x & 63u) + (uint32_t)L) * a.lds_stride_dw;
                    const uint32_t nd = min(a.lds_stride_dw, 2u + 4u * ((q_len + 27u) >> 4));
                    for (uint32_t i = 0; i < nd; i++) my_lds[i] = src[i];
                }
                // a read of ONE window, w: the FETCH phase sets it up like any other (verdicts from the prefix tables)
                cnt = 1; cls = 0x80u | kClsHelper;
                sd0 = a.seed_win[(size_t)(q_lo + rank) * a.n_reads + q_r];
                last = -1; cur_graph = q_g; done_graph = kEmpty; group_rc_called = true;
                have_read = true;
                phase = PH_FETCH;
            }
            if ((int)(threadIdx.x & 63u) == L) phase = PH_FORKWAIT;
        }
        // ---- join: an owner goes on when the lowest helper that did not fail is known and every helper below it has failed ----
        for (unsigned long long bj = __ballot(phase == PH_FORKWAIT); bj; bj &= bj - 1) {
            const int L = __ffsll(bj) - 1;
            const bool mine = (cls & kClsHelper) && (int)(fk & 63u) == L;
            const unsigned long long bm = __ballot(mine), bdone = __ballot(mine && phase == PH_JOIN), bhit = __ballot(mine && phase == PH_JOIN && (fk >> 12) != 0u);
            const unsigned long long below = bhit ? bm & ((1ull << (__ffsll(bhit) - 1)) - 1ull) : bm;
            if (below & ~bdone) continue;                       // someone whose answer matters is still walking
            // windows of the list up to the last helper below the hit are known to fail (the list ascends: a bound on the window id says it)
            const uint32_t w_top = below ? __shfl(w, 63 - __builtin_clzll(below)) + 1u : 0u;
            if (mine) { phase = (fk >> 8) & 7u; cls = 0; have_read = false; fk = 0; }   // (helpers above the first hit are called off)
            if ((int)(threadIdx.x & 63u) == L) { fk = w_top; phase = PH_FETCH; }
        }
        }
#if defined(GROOT_WORK_COUNTERS) && GROOT_WORK_COUNTERS == 3
        tl_fork += wall_clock64() - tl_c1;
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
#if defined(GROOT_WORK_COUNTERS) && GROOT_WORK_COUNTERS == 1
                if ((threadIdx.x & 63) == 0 && wc_iter > 1) atomicAdd(&a.ctr->dbg[128 + min(63u, (wc_iter - wc_round0) / 2)], 1ull);   // round length
                if ((threadIdx.x & 63) == 0) atomicMax(&a.ctr->dbg[63], (unsigned long long)(wc_iter - wc_round0));          // longest round
                wc_round0 = wc_iter;
#elif defined(GROOT_WORK_COUNTERS)
                wc_round0 = wc_iter;
#endif
#if defined(GROOT_WORK_COUNTERS) && GROOT_WORK_COUNTERS == 3
                if (chunk_base < n_todo) { tl_round = wall_clock64(); tl_rounds++; }
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
        // (running the smaller phases in the same iteration as well -- every lane a step per iteration -- was measured in round 4: the
        // phases' trips to memory then follow each other inside the iteration and nothing is gained: 3.1 -> 3.5 ms on reads with errors)
        // (a bound on the iterations a lane may wait for its phase -- its phase runs next once it has waited 4 / 8 / 16 -- was measured in
        // round 4 on all kernel-path workloads: no difference; the wavefronts that end late are not starved, they hold reads that started late)
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
#if defined(GROOT_WORK_COUNTERS) && GROOT_WORK_COUNTERS == 2
                wc_n0[0] = wc_n[0]; wc_n0[1] = wc_n[1]; wc_n0[2] = wc_n[2];
#endif
#if defined(GROOT_WORK_COUNTERS) && GROOT_WORK_COUNTERS == 3
                tl_read0 = wall_clock64(); tl_it0 = wc_iter; tl_st0 = wc_n[0] + wc_n[1] + wc_n[2];
#endif
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
                len = ra.z & ~kRecPacked;
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
                n_graphs = 0; ord = 0; last = -1; fk = 0;
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
            if (nw == kEmpty && (cls & kClsHelper)) { have_read = false; phase = PH_JOIN; continue; }   // (not reached: next_range ends a dry run)
            if (nw == kEmpty) {                               // every seed of the read handled
                GROOT_EV(4);
#if defined(GROOT_WORK_COUNTERS) && GROOT_WORK_COUNTERS == 1
                atomicAdd(&a.ctr->dbg[64 + min(63u, (wc_iter - wc_round0) / 2)], 1ull);   // when in its round the lane finished
#endif
#if defined(GROOT_WORK_COUNTERS) && GROOT_WORK_COUNTERS == 2
                if (wc_iter - wc_round0 >= 300) {              // slow reads, by name (meaningful with one read per round): read | iterations, then its FETCH / SCAN / DFS steps
                    const unsigned long long sl = atomicAdd(&a.ctr->dbg[128], 1ull);
                    if (sl < 30) {
                        a.ctr->dbg[129 + 2 * sl] = (unsigned long long)r | ((unsigned long long)(wc_iter - wc_round0) << 32);
                        a.ctr->dbg[130 + 2 * sl] = (unsigned long long)(wc_n[0] - wc_n0[0]) | ((unsigned long long)(wc_n[1] - wc_n0[1]) << 20) | ((unsigned long long)(wc_n[2] - wc_n0[2]) << 40);
                    }
                }
#endif
#if defined(GROOT_WORK_COUNTERS) && GROOT_WORK_COUNTERS == 3
                {   // reads that end late in the launch, by name: read | windows | item?, start and end (us), wave iterations and own steps meanwhile
                    const unsigned long long te = wall_clock64();
                    if (te - tl_start >= (unsigned long long)GROOT_TL_LATE_US * 100ull) {
                        const unsigned long long sl = atomicAdd(&a.ctr->dbg[17], 1ull);
                        if (sl < 14) {
                            a.ctr->dbg[18 + 3 * sl] = (unsigned long long)r | ((unsigned long long)cnt << 32) | ((unsigned long long)((cls >> 10) & 1u) << 63);
                            a.ctr->dbg[19 + 3 * sl] = ((tl_read0 - tl_start) / 100ull) | (((te - tl_start) / 100ull) << 32);
                            a.ctr->dbg[20 + 3 * sl] = (unsigned long long)(wc_iter - tl_it0) | ((unsigned long long)(wc_n[0] + wc_n[1] + wc_n[2] - tl_st0) << 32) | ((unsigned long long)ord << 48) | ((unsigned long long)n_graphs << 56);
                        }
                    }
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
            if (cls & kClsHelper) {
                if (g != cur_graph) { fk |= 2u << 12; have_read = false; phase = PH_JOIN; continue; }   // another graph's window: its owner takes it as it comes
            } else {
            if (g != cur_graph) { cur_graph = g; n_graphs++; group_rc_called = false; }
            if (g == done_graph) continue;                    // graphminion.go:96-98: stop after the first alignment
            }
            if (a.update_weights && !(cls & kClsHelper)) {     // :67 IncrementSubPath
                // neighbouring lanes mostly hold reads of the same window (processing order): one atomic per RUN of equal cells among the lanes
                // that are here together -- a lane whose cell differs from that of the lane before it adds the length of its run (round 4 looped over
                // the distinct cells)
                const uint64_t cell = (uint64_t)qrow * ix.n_windows + w;
                const unsigned long long here = __ballot(1);
                const unsigned lane_ = threadIdx.x & 63u;
                const unsigned long long below = here & ((1ull << lane_) - 1ull);
                const int prev = below ? 63 - __builtin_clzll(below) : -1;                 // the lane before this one among those here
                const uint32_t plo = __shfl((uint32_t)cell, prev < 0 ? (int)lane_ : prev), phi = __shfl((uint32_t)(cell >> 32), prev < 0 ? (int)lane_ : prev);
                const bool head = prev < 0 || plo != (uint32_t)cell || phi != (uint32_t)(cell >> 32);
                const unsigned long long heads = __ballot(head);
                if (head) {
                    const unsigned long long after = lane_ == 63u ? 0ull : heads & ~((2ull << lane_) - 1ull);   // the next run's head
                    const unsigned long long run = here & ~((1ull << lane_) - 1ull) & (after ? ((1ull << (__ffsll(after) - 1)) - 1ull) : ~0ull);
                    atomicAdd(&a.attempts[cell], (uint32_t)__popcll(run));
                }
            }
            GROOT_SUBT(3);
            if (a.incr_cnt && !(cls & kClsHelper)) {   // capture pass: which windows had IncrementSubPath called, in call order
                const uint32_t n = a.incr_cnt[r];
                a.incr_cnt[r] = n + 1;
                if (n < a.incr_cap) a.incr_win[(size_t)r * a.incr_cap + n] = w;
            }
            if (a.no_align) continue;                         // :70-72
            if (!(cls & kClsHelper) && w < fk) continue;       // a helper has found that AlignRead fails on this window in both orientations
            seed = wa.y; off0 = wa.z;
            l1_hi = wa.w;                                     // alignment.go:36 and :199-201, folded at open
            cn_begin = wb.x; cn_end = wb.y;
            seed_s0 = wb.z; seed_len = wb.w;
            GROOT_EV(5);
            advance = start_orientation(0);
            GROOT_SUBT(4);
        } else if (run == PH_SCAN) {
            if (level == 2) {
                // 2. seed node shuffling: offsets sc_pos..10 of ContainedNodes entry cn_cur and 0..10 of the next one, from their
                // 32-byte prefix records (24 bases, node, length) -- no trip to the node list, the node records or the graph bases
                bool visit = true;
                if (cn_cur < cn_end && sc_node != kEmpty) {
                    // after a cooperative scan: only the entries of the block [sc_node, sc_node + 64) whose bit is set
                    const uint32_t rel = cn_cur - sc_node;
                    unsigned long long bits = (unsigned long long)sc_s0 | ((unsigned long long)sc_len << 32);
                    bits = rel < 64u ? bits & ~((1ull << rel) - 1ull) : 0ull;
                    if (!bits) {
                        cn_cur = min(sc_node + 64u, cn_end); sc_pos = 0;
                        sc_node = kCoopWant;                     // (the next block, if there is one)
                        visit = false;
                    } else {
                        const uint32_t en = sc_node + (uint32_t)__builtin_ctzll(bits);
                        if (en != cn_cur) { cn_cur = en; sc_pos = 0; }
                    }
                }
                if (cn_cur >= cn_end) { GROOT_EV(6); advance = true; }
                else if (visit) {
                    const uint4 *e = ix.cn_pre + 2 * (size_t)cn_cur;
                    const bool two = cn_cur + 1 < cn_end && sc_node == kEmpty;
                    uint4 a0 = e[0], a1 = e[1], b0 = make_uint4(0, 0, 0, 0), b1 = make_uint4(0, 0, 0, 0);
                    if (two) { b0 = e[2]; b1 = e[3]; }
                    uint64_t c_lo, c_hi;
                    uint64_t w0 = (uint64_t)a0.x | ((uint64_t)a0.y << 32), w1 = (uint64_t)a0.z | ((uint64_t)a0.w << 32), w2 = (uint64_t)a1.x | ((uint64_t)a1.y << 32);
                    uint32_t node = a1.z, nlen = a1.w;
                    filter16(w0, w1, w2, (int)min(nlen, 11u), (int)nlen, c_lo, c_hi);
                    c_lo &= ~low_bytes((int)sc_pos); c_hi &= ~low_bytes((int)sc_pos - 8);
                    uint32_t j = first16(c_lo, c_hi);
                    if (j >= 16u && two) {                     // nothing (left) in this node: the next one
                        cn_cur++; sc_pos = 0;
                        w0 = (uint64_t)b0.x | ((uint64_t)b0.y << 32); w1 = (uint64_t)b0.z | ((uint64_t)b0.w << 32); w2 = (uint64_t)b1.x | ((uint64_t)b1.y << 32);
                        node = b1.z; nlen = b1.w;
                        filter16(w0, w1, w2, (int)min(nlen, 11u), (int)nlen, c_lo, c_hi);
                        j = first16(c_lo, c_hi);
                    }
                    if (j >= 16u) {
                        GROOT_EV(7);
                        cn_cur++; sc_pos = 0;
                        advance = cn_cur >= cn_end;
                    } else {
                        // exact 8-base check of the lowest survivor (alignment.go:203-223 would fail here otherwise)
                        const uint64_t g8 = j < 8 ? window8(w0, w1, j) : window8(w1, w2, j - 8);
                        sc_pos = j + 1; sc_end = min(nlen, 11u);
                        if (!prefix_ok(g8, pre8, min(min(nlen - j, eff()), 8u)) || cannot_start(node, j)) {
                            GROOT_EV(8);
                            if (sc_pos >= sc_end) { cn_cur++; sc_pos = 0; advance = cn_cur >= cn_end; }
                        } else {
                            begin_dfs(node, j);
                            GROOT_EV(10);
                        }
                    }
                }
            } else if (sc_pos >= sc_end) { GROOT_EV(6); advance = true; }
            else {
                // levels 1, 3, 4: up to 16 start offsets sc_pos.. of node sc_node: which can spell the first bases of the read?
                const uint8_t *gb = ix.bases + sc_s0 + sc_pos;
                const uint32_t npos = min(16u, sc_end - sc_pos);
                const uint64_t w0 = ld8(gb), w1 = ld8(gb + 8), w2 = ld8(gb + 16);
                uint64_t c_lo, c_hi;
                filter16(w0, w1, w2, (int)npos, (int)(sc_len - sc_pos), c_lo, c_hi);   // (room = bases from sc_pos to the node end)
                const uint32_t j = first16(c_lo, c_hi);
                const uint64_t g8 = j < 8 ? window8(w0, w1, j) : window8(w1, w2, j - 8);
                if (j >= npos) {
                    GROOT_EV(7);
                    sc_pos += npos;
                    advance = sc_pos >= sc_end;                   // set up the next range in this step: no empty one
                } else {
                    // exact 8-base check of the lowest survivor (alignment.go:203-223 would fail here otherwise)
                    const uint32_t off = sc_pos + j;
                    sc_pos = off + 1;
                    if (!prefix_ok(g8, pre8, min(min(sc_len - off, eff()), 8u)) || cannot_start(sc_node, off)) {
                        GROOT_EV(8);
                        advance = sc_pos >= sc_end;
                    } else {
                        begin_dfs(sc_node, off);
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
                const uint32_t take = min(rec.seq_len(), eff() - dist);
                const uint32_t rdeg = rec.deg();
                if (coff == 0 && take >= 1 && take <= 8 && take == rec.seq_len() && dist + take < eff() && !rec.wild() && rdeg >= 1 && rdeg <= 4 &&
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
            const uint32_t take = min(rec.seq_len() - coff, eff() - dist);
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
                    if (dist == eff() || rdeg == 0) {             // :229-236 report the traversal
                        if (any && (cls & kClsHelper)) {       // dry run: the window aligns -- that is all its owner wants to know
                            fk |= 1u << 12; sp = 0; emitted = 1;
                        } else if (any) {
                            GROOT_EV(14);
                            if (ord) GROOT_EV(19);
                            groot_trav t;
                            t.read_id = read_id; t.graph_id = g; t.node = node0; t.offset = noff0;
                            t.ord = (uint16_t)ord;
                            t.flags = (uint8_t)((rc ? GROOT_TRAV_RC : 0u) | (level == 3u ? GROOT_TRAV_START_CLIP : level == 4u ? GROOT_TRAV_END_CLIP : 0u) | (emitted == 0 ? GROOT_TRAV_FIRST : 0));
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
                if (level == 1u && noff0 == off0) reach = max(reach, ok ? dist : dist + nb - 1u);   // (a failed comparison leaves dist where it was)
                if (sp == 0) {                                 // performAlignment is over
                    if (emitted && (cls & kClsHelper)) { have_read = false; phase = PH_JOIN; }
                    else if (emitted) {                        // alignment found for (read, graph)
                        done_graph = g; phase = PH_FETCH;
                        // graphminion.go:96-98 passes over the graph's other seeds: they are the windows up to the graph's last one
                        // (a read below the window size can bring a hundred of them: one FETCH step instead of one each)
                        if (ix.graph_win_end) last = (long long)ix.graph_win_end[g] - 1;
                    }
                    else {
                        phase = PH_SCAN;
                        // (the start position just tried was the range's last one: the next range is set up right away instead of by a SCAN
                        // step of its own -- 30 % of the SCAN lane-steps of a batch of reads with errors were such empty ones)
                        if (sc_pos >= sc_end) {
                            if (level == 2) { cn_cur++; sc_pos = 0; advance = cn_cur >= cn_end; }
                            else advance = true;
                        }
                    }
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

#if defined(GROOT_WORK_COUNTERS) && GROOT_WORK_COUNTERS == 3
    if ((threadIdx.x & 63) == 0 && tl_rounds) {
        const unsigned long long te = wall_clock64(), d = te - tl_start, dl = te - tl_round;   // 100 MHz ticks; buckets of 50 us
        atomicAdd(&a.ctr->dbg[0], (unsigned long long)wc_iter);
        atomicMax(&a.ctr->dbg[1], (unsigned long long)wc_iter);
        atomicAdd(&a.ctr->dbg[2], d);
        atomicMax(&a.ctr->dbg[3], d);
        atomicAdd(&a.ctr->dbg[4], 1ull);
        atomicAdd(&a.ctr->dbg[5], (unsigned long long)tl_rounds);
        atomicAdd(&a.ctr->dbg[8], tl_coop); atomicAdd(&a.ctr->dbg[9], tl_fork); atomicAdd(&a.ctr->dbg[16], (unsigned long long)tl_ncoop);
        for (int i = 0; i < 3; i++) { atomicAdd(&a.ctr->dbg[10 + i], wc_t[i]); atomicAdd(&a.ctr->dbg[13 + i], (unsigned long long)wc_n[i]); }
        atomicAdd(&a.ctr->dbg[64 + min(63ull, d / 5000ull)], 1ull);
        atomicAdd(&a.ctr->dbg[128 + min(63ull, dl / 5000ull)], 1ull);
    }
#elif defined(GROOT_WORK_COUNTERS)
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


} // namespace groot

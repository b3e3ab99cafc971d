// device_types.hpp -- argument blocks shared by the gfx950 kernels and the host-side ctx code
#pragma once

#include <cstdint>

#include <hip/hip_runtime.h>

#include "groot_hip.h"

namespace groot {

constexpr int kBlock = 256;           // threads per workgroup: 4 wave64, one per SIMD
constexpr uint32_t kEmpty = 0xFFFFFFFFu;

// error / status bits the kernels OR into DeviceCounters::flags
enum : uint32_t {
    kFlagShortRead = 1u,      // len < k            (reference: panic, boss.go:164-166)
    kFlagLongRead = 2u,       // len > max_read_len (device stacks / LDS are sized for it)
    kFlagSeedOverflow = 4u,   // a read produced more seeds than max_seeds_per_read slots
    kFlagTravOverflow = 8u,   // more traversal records than the output buffer holds
    kFlagOrdOverflow = 16u,   // > 65535 traversals for one read
    kFlagOvfOverflow = 32u,   // a shard of the overflow traversal list is full
    kFlagQOverflow = 64u,     // more distinct kmerCounts among the seeded reads than the call-count table has rows
};

constexpr uint32_t kIncrCap = 8;      // AlignArgs::incr_win slots per read in the capture pass of groot_hip_open (a second pass with kIncrCapBig takes the few strings that need more)
constexpr uint32_t kIncrCapBig = 2048;
constexpr uint32_t kSeedShards = 64, kSeedShardStride = 16;   // seed-stage counters: one 128-byte line per shard
constexpr uint32_t kOvfShards = 256;  // overflow traversal lists, picked by workgroup id: spreads the atomics

struct DeviceCounters {
    unsigned long long mapped, multimapped, alignments, seeds, revcomp_panics, short_reads;
    unsigned int n_trav;        // traversal records emitted (may exceed capacity: then kFlagTravOverflow)
    unsigned int max_seeds;     // largest per-read seed count seen
    unsigned int flags;
    unsigned int q_rows;        // rows of the call-count table in use after this batch (may exceed its capacity: then kFlagQOverflow)
    unsigned int mask_words;    // 64-bit words of the compact path sets of this batch (mask_compact_kernel)
    unsigned int todo_reads;    // reads sketch_sig_kernel handed to the full-width kernel (0 when that kernel ran alone)
    unsigned int seeded_reads;  // reads with at least one seed whose alignment is not tabulated: the align stage's share of the processing order (they sort first)
    unsigned int tab_reads;     // reads the signature kernel answered from the outcome table
    unsigned int lean_reads;    // reads align_lean_kernel finished
    unsigned long long dbg[192]; // work counters (only with -DGROOT_WORK_COUNTERS): [e] wave iterations with event e, [32+e] lanes with it,
                                 // [64+b] lanes finishing their read b*2 iterations into the round, [128+b] rounds of that length
};

// Everything a DFS step needs about one graph node in one aligned record (64 B for PW<=3, 128 B for PW=11):
// GrootGraphNode{SegmentLength, Sequence[:8], OutEdges, PathIDs} (src/graph/node.go:13-22).
template <int PW> struct alignas(16) NodeRec {
    uint32_t seq_off;          // into DeviceIndex::bases
    uint32_t seq_len;
    uint32_t deg;              // bits 0..30 out-degree (> 4: edges[0] is the offset into DeviceIndex::edges, child_first unused); bit 31: the node holds an 'N'
    uint8_t child_first[4];    // first base of each embedded neighbour
    uint64_t first8;           // first 8 bases of the node (bytes past seq_len are don't-care)
    uint32_t edges[4];         // OutEdges order (global node indices)
    uint64_t mask[PW];         // path ids through the node
};
static_assert(sizeof(NodeRec<3>) == 64, "node record must be one 64-byte line");
static_assert(sizeof(NodeRec<11>) == 128, "wide node record must be 128 bytes");

// lshe.Key of one window, the fields AlignRead needs, in one 32-byte record (one load instead of six)
struct alignas(16) WinRec {
    uint32_t graph;        // Key.GraphID
    uint32_t node;         // global node index of Key.Node
    uint32_t offset;       // Key.OffSet
    uint32_t l1_hi;        // exclusive end of the level-1 offsets: min(len(node), OffSet + MergeSpan + WindowSize + 1)
    uint32_t cn_off, cn_end;   // ContainedNodes range in cn_node
    uint32_t seed_s0, seed_len;   // base offset / length of Key.Node
};
static_assert(sizeof(WinRec) == 32, "window record is two 16-byte words");

// what the align stage needs to start a read, written by the seed stage in one 32-byte record
struct alignas(16) ReadRec {
    uint64_t seq_off;      // first base of the read in the batch buffer
    uint32_t len;          // | kRecPacked
    uint32_t cnt_flags;    // seeds (bits 0..23) | kRec* verdicts of the seed stage on the FIRST seed window | byte > 'T' (bit 31)
    uint32_t seed[4];      // the first four seed windows (all of them for 99.9% of reads); more: seed_win slots
};
static_assert(sizeof(ReadRec) == 32, "read record is two 16-byte words");

// cnt_flags bits 24..29: the hierarchy level(s) of AlignRead that cannot produce a start position for the forward read
// (F) / its reverse complement (R) in the read's first seed window, as established by the seed stage
constexpr uint32_t kRecCountMask = 0x00FFFFFFu;
constexpr uint32_t kRecNo12F = 1u << 24, kRecNo3F = 1u << 25, kRecNo4F = 1u << 26;   // R = F << 3
constexpr uint32_t kRecPacked = 1u << 31;                   // ReadRec::len: the read is all ACGT and its 2-bit codes are in SeedArgs::packed (sketch_sig_kernel wrote them: the align stage's first pass stages it from there)
constexpr uint32_t kRecAscending = 1u << 30;                // cnt_flags: the read's seed windows were written in ascending order
constexpr uint32_t kLongListCap = 1u << 20, kSortSeedsMax = 512;   // sort_seed_lists_kernel: reads per batch, seed windows per read
// A read with more than kSplitMin seed windows is handled by several lanes of the align stage: its ascending window list is cut at
// graph boundaries into items of at least kSplitMin windows (graphminion.go:46-102 treats the graphs of a read independently of
// each other; only the order of the records and the per-read counters tie them together, and those are put right afterwards).
constexpr uint32_t kSplitMin = 8, kSplitMaxItems = 256;
constexpr uint32_t kRecSplit = 1u << 23;                    // cnt_flags: the read's first item ends at the count in bits 0..22; the rest are AlignArgs::vitem entries
constexpr uint32_t kPrefixWords = 256;   // words per window in DeviceIndex::win_prefix

// a traversal record on its way to the host: what cannot be derived there (graph = graph of the node, ord = position among
// the read's records, read id = first_read_id + position in the batch)
struct groot_ctrav {
    uint32_t node, offset;
    uint32_t read_flags;       // position of the read in the batch (24 bits) | GROOT_TRAV_* flags << 24
};
static_assert(sizeof(groot_ctrav) == 12, "packed traversal record is 12 bytes");

// Outcome table entry (DeviceIndex::out_tab), as dwords: what graphMinion + AlignRead (graphminion.go:46-102, alignment.go:13-159)
// produce for a read that IS bases [o, o + WindowSize) of a window text row -- every such read is the same string with the same seed
// windows, so the outcome is a function of (window, row, o) and the ctx works it out once, at open, by running the align stage itself
// on the string.  One entry per traversal record:
//   [0] node  [1] offset  [2] graph  [3] flags (GROOT_TRAV_*, bits 0..7) | multimapped << 8 | mapped << 9 (first entry) | seed windows in this entry << 10 | traversals of the string << 16 (first entry)
//       (entries with node = kEmpty carry no record: calls, seeds and counters of a string without traversals, or calls and seeds
//       beyond what the traversal entries hold; their number sits in bits 20..31 of [2] of the string's first entry)
//   [4],[5] windows whose IncrementSubPath the read triggers (kEmpty = none; a string's calls are spread over its entries)
//   [6..] path set, pw 64-bit words (lo, hi)
//   last four dwords of the entry: seed windows of the read (kEmpty = none; spread over the string's entries like the calls)
constexpr uint32_t kOutHdrDw = 6, kOutSeedDw = 4;
constexpr uint32_t kOutTab = 0x80000000u;                  // sig_info: the string's outcome is tabulated
// sig_info bits 27..30: traversals - 1, or 15 = more than fifteen: the number is in the string's first entry ([3] bits 16..31)
constexpr uint32_t kOutMaxTrav = 4095, kOutIdxBits = 25, kOutTravShift = 27, kOutTravLong = 15;
constexpr uint32_t kOutNoRec = 1u << 25;                   // sig_info: the string has seed windows (or none) but AlignRead reports nothing for it: one entry, no record
constexpr uint32_t kOutAllSeeds = 1u << 26;                // sig_info: ... and IncrementSubPath is called exactly once for each of the read's seed windows
constexpr uint32_t kTabSeedsHere = 0x40000000u;            // SeedArgs::tab_idx: text_lookup_kernel answered the read; order_first_kernel writes its seeds too
constexpr uint32_t kTabCounted = 0x80000000u;              // SeedArgs::tab_idx: the seed stage has counted the read's IncrementSubPath calls
// (entries are padded to a power of two of at least 64 bytes: a gather touches one 64-byte sector, never two)
__host__ __device__ inline uint32_t out_stride_q(uint32_t pw)
{
    uint32_t q = 4;
    while (q * 16 < (kOutHdrDw + kOutSeedDw) * 4 + 8 * pw) q <<= 1;
    return q;
}

// Text-table entry (64 bytes): [0] tag  [1] sig_info word  [2, 2 + tw) the string at 2 bits per base, zero-padded  [2 + tw, 2 + tw + xw) the
// string's bytes other than ACGT, in ascending position: (position + 1) << 8 | byte, two per dword, low half first, the rest 0 --
// their 2-bit codes in the string are 0.  xw = the dwords left in the 16-byte word the string ends in, or a whole one.
// tw = text_key_dwords(dwords the WindowSize bases take): the widths text_lookup_kernel is built for.
__host__ __device__ constexpr uint32_t text_key_dwords(uint32_t dw) { return dw <= 7 ? 7u : dw <= 8 ? 8u : 14u; }
__host__ __device__ constexpr uint32_t text_exc_dwords(uint32_t tw)
{
    return tw > 14 ? 0u : ((2 + tw) % 4 ? 4 - (2 + tw) % 4 : (tw <= 10 ? 4u : 0u));
}

// exact-match table entry: windows whose whole sketch equals the query's
struct ExactEntry { uint32_t tag; uint32_t id; };

// signature index (sketch_sig_kernel): the windows grouped by the signature of their sketch (kSigG of its slots, kernels_common.hpp).
// DeviceIndex::sig_dir: open addressing over the DISTINCT signatures, buckets of two {tag, first entry} pairs (one 16-byte load; first = kEmpty: free);
// DeviceIndex::sig: one entry per window, the windows of a signature back to back, sorted by (sketch class, window id) -- the windows of
// one class (identical 64-bit sketches: a read's seed set) are neighbours, in ascending id as the exact table returns them.
struct alignas(32) SigEntry {
    uint32_t id;           // window
    uint32_t cls;          // windows with identical 64-bit sketches share a class
    uint32_t text_len;     // bits 0..9: bases of the window's text that were verified at open (0: the window cannot confirm a read);
                           // bits 10..19 / 20..29: first position of the smallest canonical k-mer hash in the forward / reverse-complement row;
                           // bit 30 (kSigInline): `verdict` below holds what DeviceIndex::sig_info says for offsets 0..7 of both rows
    uint32_t group;        // bits 0..23: entries of this signature's group from this one on (the first entry holds the group's size);
                           // bits 24..31: min(255, contained nodes of the window) (DeviceIndex::win_nodes: the span class of the scheduling key)
    uint8_t verdict[2][8]; // the verdict byte of the strings at offsets 0..7 of the forward / reverse-complement row (most windows merge fewer than
                           // eight WindowSize-mers): one 32-byte entry answers what took three more trips (sig_info, win_nodes) for most reads
};
static_assert(sizeof(SigEntry) == 32, "signature entry is two 16-byte words");
constexpr uint32_t kSigInline = 1u << 30;
constexpr uint32_t kTextMax = 256;       // bases kept per window text (window + merged neighbours), per orientation
// SigEntry::text_len fields
__host__ __device__ inline uint32_t sig_text_pack(uint32_t len, uint32_t argmin_fwd, uint32_t argmin_rc) { return len | (argmin_fwd << 10) | (argmin_rc << 20); }
__host__ __device__ inline uint32_t sig_text_len(uint32_t v) { return v & 1023u; }
__host__ __device__ inline uint32_t sig_text_argmin(uint32_t v, uint32_t row) { return (v >> (10 + 10 * row)) & 1023u; }

// graph + window arrays resident in HBM (replicated per GPU)
struct DeviceIndex {
    uint32_t k, s, w, num_window_kmers, n_windows, n_nodes, pw; // pw = path words on the device (>= view.path_words)
    const uint32_t *edges;          // only for nodes with more than 4 OutEdges (NodeRec embeds the rest)
    const uint8_t *bases;
    const uint32_t *win_graph, *cn_node;
    const uint32_t *graph_win_end;  // [n_graphs] one past the last window of the graph (windows are numbered graph by graph); null if they are not
    const WinRec *win_rec;          // [n_windows]
    const uint64_t *win_sketch;     // [n_windows*s]
    // per window: which read prefixes (6-mer codes of oriented bases [0,6) and [6,12), 2 bits per base A=0 C=1 T=2 G=3)
    // can be spelled from any level-1 / level-2 start position of AlignRead (alignment.go:34-70): 2 x 4096 bits
    const uint32_t *win_prefix;
    // per ContainedNodes entry (cn_node order: windows one after the other, ascending SegmentID within a window) a 32-byte prefix
    // record = two uint4: bytes 0..23 the node's first 24 bases (0 past its end), dword 6 the global node index, dword 7 its length --
    // level 2 of AlignRead (alignment.go:47-70: offsets 0..10 of every contained node) reads nothing else until a start position matches
    const uint4 *cn_pre;
    // per (node, start offset 0..10): a 64-bit set (two bits per member, l2_bloom_bits) of the 8-mers a DFS from there can spell --
    // following every out-edge, the graph's 'N' and everything behind a sink counting as any base (dfsRecursive, alignment.go:193-254).
    // A start position whose set lacks the read's first eight bases cannot produce a traversal: level 2 tries offsets 0..10 of every
    // contained node, and on nodes of a few bases the in-node comparison lets most of them through.  [n_nodes][11]; null = none
    const uint64_t *node_l2b;
    // lookup structures
    const ExactEntry *exact;        // open addressing, exact_mask+1 slots
    uint32_t exact_mask;
    const uint32_t *band_keys;      // [l_max][n_windows][max_k] low-32 hash values, sorted per band
    const uint32_t *band_ids;       // [l_max][n_windows]
    // per (band b, prefix length K): open-addressing table from the hash of the band's first K values to the first row of
    // band_keys with that prefix -- [(b * max_k + K-1) << band_hash_bits | slot], reusing ExactEntry {tag, id = row}
    const ExactEntry *band_hash;
    uint32_t band_hash_bits;
    // per band row: five bits per sketch slot (the first 24 slots, six to a dword), sig5() of the slot's 64-bit value.  Rows of equal prefix
    // are neighbours, so a candidate is first judged on this 16-byte neighbourhood read (32 bytes of one byte per slot until round 5: half the
    // lines and round trips of the walk): only windows whose signature agrees in enough slots have their 8*S-byte sketch fetched
    const uint8_t *band_sig;        // [l_max][n_windows][16]
    const uint32_t *band_run;       // [l_max][max_k][n_windows]: rows sharing the K-prefix of row e, counted from e (valid at first rows)
    uint32_t max_k, l_max;
    // per kmerCount q (0..max_q): LSH params and the smallest #equal slots with Containment > t
    const uint8_t *q_k, *q_l;
    const uint16_t *q_min_eq;
    uint32_t max_q;
    // IncrementSubPath call counts are kept per (kmerCount, window) only for the kmerCounts that occur: q_row[q] = row of
    // the call-count table holding kmerCount q, kEmpty until a seeded read with that q shows up (assign_q_rows_kernel)
    const uint32_t *q_row;          // [max_q + 1]
    // sketch_sig_kernel: signature table (open addressing, windows inserted in ascending id like `exact`) and per window
    // the bases it was sketched from -- WindowSize + MergeSpan of them along its first Ref path, 2 bits per base
    // ((byte >> 1) & 3, base j at bits 2*(j%4) of byte j/4): forward row at byte w*2*kTextMax/4, reverse complement kTextMax/4 later; null = not available
    const SigEntry *sig;
    const uint4 *sig_dir;           // [sig_mask + 1] buckets {tag0, first0, tag1, first1}
    uint32_t sig_mask;
    const uint8_t *win_text;
    // what the seed stage's epilogue works out for a read that IS bases [o, o + WindowSize) of a text row (every such read is
    // the same string): verdict bits (kRecNo* >> 24, both orientations) | dead-orientation class << 6, as the full-width kernel
    // produced them for exactly that string when the ctx was opened -- [(window * 2 + row) * sig_verdict_stride + o]; null = none
    // One u32 per such string since round 3.  Bit 31 clear: bits 0..7 = that verdict byte.  Bit 31 set: the whole outcome of the
    // graphMinion loop for the string is tabulated (kOutTab): bits 27..30 = traversals - 1, bit 26 = kOutAllSeeds, bit 25 = kOutNoRec, bits 0..24 = index of its first OutEntry.
    const uint32_t *sig_info;
    uint32_t sig_verdict_stride;
    // AlignRead outcomes of window-text strings (groot_hip_open ran the align stage on every one of them): out_stride_q 16-byte
    // words per entry, entries of a string back to back in `ord` order
    const uint4 *out_tab;
    uint32_t out_stride_q;
    // text_lookup_kernel: window-text strings with a tabulated outcome, keyed by their bases (open addressing, text_mask + 1 entries of 64 bytes); null = none
    const uint4 *text_tab;
    uint32_t text_mask;
    const uint8_t *win_nodes;       // [n_windows] min(255, contained nodes of the window): the span class of the scheduling key
};

struct SeedArgs {
    DeviceIndex ix;
    const uint8_t *seq;
    const uint64_t *seq_off;
    uint32_t n_reads, max_read_len;
    uint32_t lds_read_bytes;     // bytes of LDS available for staging the block's reads
    uint32_t seed_slots;         // H
    uint32_t *seed_count;        // [n_reads]
    uint32_t *seed_win;          // [H][n_reads] slot-major
    uint64_t *sketch_out;        // [n_reads*s] or null
    uint32_t *sort_key;          // [n_reads] (node span class, first seed window, orientation class), kEmpty without seeds; or null
    uint32_t sort_span_bits;     // bits of sort_key that hold the span class 2^bits-1 - min(contained nodes of the window, 2^bits-1); 0 = none
    uint32_t sort_span_shift;    // ... and where they sit: right above window << 2 | class, so that the sort runs over as few bits as the index needs
    ReadRec *read_rec;           // [n_reads]
    uint32_t *q_seen;            // [max_q + 1] set for every kmerCount of a seeded read that has no row yet; or null
    uint32_t *trav_cnt;          // [n_reads] traversal counts of the align stage: zeroed here for reads without seeds (it skips them); or null
    unsigned long long *shards;  // [kSeedShards][kSeedShardStride]: {sum of seeds, largest per-read seed count} per shard of workgroups
    uint32_t *tab_idx;           // [n_reads] first OutEntry (| kTabCounted) of reads whose outcome is tabulated, kEmpty for the others; or null: no table
    uint32_t *tab_hist;          // [n_windows] IncrementSubPath calls of tabulated reads counted by the seed stage in this batch; or null
    // reads on the LSH-Forest branch of Query whose rows of equal band prefix number more than lsh_defer_rows, handed by the hashing
    // kernels to lsh_heavy_kernel (a wavefront per read): read | byte > 'T' << 31, and their sketches ([position in the list][s] u64);
    // null: every lane walks its own rows
    uint32_t *lsh_list, *lsh_count;
    uint64_t *lsh_sketch;
    uint32_t *dfs_list, *dfs_count;   // reads with a scheduling key, appended by the seed epilogue (processing order of the align stage when few are left); or null
    uint32_t *long_list, *long_count; // reads with more than four seed windows that were not found in ascending order (sort_seed_lists_kernel); up to kLongListCap
    uint4 *packed;               // [n_reads][packed_q] the reads sketch_sig_kernel decides, 16 bases per dword (for align_lean_kernel); null: not wanted
    uint32_t packed_q;           // 16-byte words per read in `packed`
    uint32_t *todo_list;         // [n_reads] reads sketch_sig_kernel leaves to sketch_seed_kernel<..., LIST>
    uint32_t *todo_count;        // [1]
    uint32_t lsh_defer_rows, lsh_cap;
    uint32_t list_stride_dw;     // dwords of LDS per lane of the LIST kernel for its own copy of the read (odd), 0 = read from HBM
    DeviceCounters *ctr;
};

struct AlignArgs {
    DeviceIndex ix;
    const uint8_t *seq;
    const uint64_t *seq_off;
    uint32_t n_reads, first_read_id;
    uint32_t seed_slots;
    const uint32_t *seed_count;
    const uint32_t *seed_win;
    const uint32_t *perm;        // [n_reads] processing order (reads sorted by sort_key) or null
    const ReadRec *read_rec;     // [n_reads], by read
    uint32_t no_align, update_weights;
    const void *node_rec;        // NodeRec<pw>[n_nodes]
    uint32_t *attempts;          // [rows][n_windows], row = ix.q_row[kmerCount]
    // traversal output: traversal 0 of read r goes to slot r; later ones (ord >= 1, ~4% of them)
    // to a sharded overflow list.  order_* kernels then compact both into (read, ord) order.
    groot_trav *trav_first;      // [n_reads]
    uint64_t *mask_first;        // [n_reads*pw]
    uint32_t *trav_cnt;          // [n_reads] traversals emitted for read r
    groot_trav *ovf_trav;        // [kOvfShards][ovf_cap]
    uint64_t *ovf_mask;          // [kOvfShards*ovf_cap*pw]
    uint32_t *ovf_cnt;           // [kOvfShards] list lengths, then [kOvfShards], [kOvfShards+1] = cursors over the sorted slots (align_kernel)
    uint32_t ovf_cap;
    // DFS stacks: entry d of thread t lives at [d*n_threads + t]
    uint64_t *stk_hdr;
    uint64_t *stk_mask;          // [(d*n_threads + t)*pw + word]
    uint32_t n_threads, stk_depth;
    uint32_t lds_stride_dw;      // dwords of LDS per lane for the staged read (odd), 0 = reads stay in global memory
    // groot_hip_open's capture pass for the outcome table: per read the windows whose IncrementSubPath was called, in call order
    // ([n_reads][kIncrCap], count in incr_cnt[r] bits 0..30, bit 31 = the read touched more than one graph); null otherwise
    uint32_t *incr_cnt, *incr_win;
    uint32_t incr_cap;           // slots per read in incr_win
    uint32_t head_lanes;         // lanes per round in the head of the processing order (the longest walks); 0 = as everywhere else
    uint32_t round_lanes;        // lanes a wavefront fills per round; 0 = 64, fewer when the batch leaves the align stage little to do (see the kernel)
    // items of split reads (kSplitMin): {read, first seed position, end position, records of the read's earlier items}; they come
    // first in the processing order (slot j < min(*vcount, vcap) is item j), and item j writes its count / first record to
    // slot n_reads + j of trav_cnt / trav_first / mask_first and labels its overflow records with read n_reads + j
    const uint4 *vitem;
    const uint32_t *vcount;
    uint32_t vcap;
    const uint32_t *n_perm;      // entries of perm when it is the list the first pass left (align_lean_kernel + stream compaction); null: DeviceCounters::seeded_reads
    uint32_t refill;             // waiting lanes that make a wavefront take new reads: 64 (all of them) for batches of one read length, 32 for mixed ones
    DeviceCounters *ctr;
};

// ---- K3, first pass (kernels_lean.hpp) ----
// a graph node for the lean walk: 64 bytes, four 16-byte loads in flight together
struct alignas(16) LeanNode {
    uint32_t seq_off;      // index of the node's first base in DeviceIndex::bases / LeanArgs::bases2
    uint32_t seq_len;
    uint32_t deg_kids;     // bits 0..2 out-degree (<= 4); bit 31: not for this pass (an 'N' in the node, more than four OutEdges);
                           // bits 8..23: four bits per OutEdge: 0..3 = code of the neighbour's first base, 4 = it is an 'N', 8 = the neighbour is empty;
                           // bits 24..27: OutEdge e leads to a node of more than 32 bases (its LeanExt is wanted with its record)
    uint32_t pad;
    uint32_t edges[4];     // OutEdges order (global node indices)
    uint64_t first32;      // first min(32, seq_len) bases, 2 bits each
    uint64_t mask[3];      // path ids through the node
};
static_assert(sizeof(LeanNode) == 64, "lean node record is one 64-byte line");
constexpr uint32_t kLeanNo = 0x80000000u;
// bases [32, 256) of a node, 2 bits each: fetched together with the node's record when the node is known to be long, so that a walk
// step is ONE trip to memory whatever the node's length
struct alignas(16) LeanExt {
    uint64_t b[7];
    uint64_t pad;
};
static_assert(sizeof(LeanExt) == 64, "lean node extension is one 64-byte line");
constexpr uint32_t kLeanMaxLen = 256;   // longest read the first pass takes

struct LeanArgs {
    const LeanNode *nodes;
    const LeanExt *ext;
    const uint32_t *bases2;        // every graph base at 2 bits, 16 to a dword, in DeviceIndex::bases order (+ slack for 17 dwords from any base)
    const uint4 *cn_pre2;          // per ContainedNodes entry: {bases [0,16), bases [16,24) | min(len, 65535) << 16, node, index of its first base}
    const uint64_t *node_l2b;      // DeviceIndex::node_l2b (null: none)
    const uint8_t *win_ok;         // [n_windows] 1: neither the window's node nor any contained node holds an 'N'
    const WinRec *win_rec;
    const uint32_t *q_row;
    const uint8_t *seq;
    const uint4 *packed;           // SeedArgs::packed of the batch (reads whose record says kRecPacked); null: none
    uint32_t packed_q;
    const uint32_t *perm;          // processing order
    const ReadRec *read_rec;
    uint32_t n_reads, first_read_id, n_windows, k;
    uint32_t update_weights;
    uint32_t lds_stride_dw;        // lean_stride_dw(max_len)
    uint32_t max_len;              // longest read the slices hold
    uint32_t *attempts;
    groot_trav *trav_first;
    uint64_t *mask_first;
    uint32_t *trav_cnt;
    uint8_t *defer;                // [n_reads] by slot: 1 = left to align_kernel
    uint4 *stk;                    // [n_reads][2][2] by slot: pending neighbours of the walk {node | long << 31, dist, path set}
    groot_trav *ovf_trav;          // AlignArgs::ovf_* (traversals with ord >= 1)
    uint64_t *ovf_mask;
    uint32_t *ovf_cnt;
    uint32_t ovf_cap;
    DeviceCounters *ctr;
};

// LDS dwords per lane: 2 zero dwords, the read at 16 bases per dword (one strand at a time), 2 zero dwords, 7 dwords of the current window's record,
// the read's (up to four) seed windows; odd
__host__ __device__ inline uint32_t lean_stride_dw(uint32_t max_len) { return (((max_len + 15u) >> 4) + 15u) | 1u; }

} // namespace groot

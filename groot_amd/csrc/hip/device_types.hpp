// device_types.hpp -- argument blocks shared by the gfx950 kernels and the host-side ctx code
#pragma once

#include <cstdint>

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
};

struct DeviceCounters {
    unsigned long long mapped, multimapped, alignments, seeds, revcomp_panics, short_reads;
    unsigned int n_trav;        // traversal records emitted (may exceed capacity: then kFlagTravOverflow)
    unsigned int max_seeds;     // largest per-read seed count seen
    unsigned int flags;
    unsigned int pad;
};

// exact-match table entry: windows whose whole sketch equals the query's
struct ExactEntry { uint32_t tag; uint32_t id; };

// graph + window arrays resident in HBM (replicated per GPU)
struct DeviceIndex {
    uint32_t k, s, w, num_window_kmers, n_windows, n_nodes, pw; // pw = path words on the device (>= view.path_words)
    const uint32_t *node_seq_off, *node_edge_off, *edges;
    const uint8_t *bases;
    const uint64_t *node_mask;      // [n_nodes*pw]
    const uint32_t *win_graph, *win_node, *win_offset, *win_merge_span, *win_cn_off, *cn_node;
    const uint64_t *win_sketch;     // [n_windows*s]
    // lookup structures
    const ExactEntry *exact;        // open addressing, exact_mask+1 slots
    uint32_t exact_mask;
    const uint32_t *band_keys;      // [l_max][n_windows][max_k] low-32 hash values, sorted per band
    const uint32_t *band_ids;       // [l_max][n_windows]
    uint32_t max_k, l_max;
    // per kmerCount q (0..max_q): LSH params and the smallest #equal slots with Containment > t
    const uint8_t *q_k, *q_l;
    const uint16_t *q_min_eq;
    uint32_t max_q;
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
    uint32_t no_align, update_weights;
    uint32_t *attempts;          // [(max_q+1)*n_windows]
    // traversal output (unsorted; sorted by key afterwards)
    groot_trav *trav;
    uint64_t *trav_mask;         // [cap*pw]
    uint64_t *trav_key;          // (local read index << 16) | ord
    uint32_t trav_cap;
    // DFS stacks: entry d of thread t lives at [d*n_threads + t]
    uint64_t *stk_hdr;
    uint64_t *stk_mask;          // [(d*n_threads + t)*pw + word]
    uint32_t n_threads, stk_depth;
    DeviceCounters *ctr;
};

} // namespace groot

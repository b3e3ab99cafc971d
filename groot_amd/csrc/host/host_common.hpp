// host_common.hpp -- shared bits of libgroot_host.so (no GPU code in this library)
#pragma once

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <mutex>
#include <string>
#include <vector>

#include "groot_host.h"

namespace groot {

// CPUs this process may really use: min(affinity mask, cgroup CPU quota) -- containers often expose every hardware thread
// of the box (std::thread::hardware_concurrency) while granting a fraction of them; oversubscribing that quota only adds
// throttling stalls.  The "0 = all cores" defaults of this library mean this number.
unsigned usable_cpus();

// thread-local last-error string behind groot_host_last_error()
int set_error(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

// ---- host ntHash/KHF used only at index-build time for graph windows (graph.go:292-296) ------
// Arithmetic of github.com/will-rowe/nthash v0.2.0 (call sites src/minhash/khf.go:38,44).
struct NtHashTables {
    uint64_t seed[256];
    NtHashTables();
};
const NtHashTables &nthash_tables();

inline uint64_t rol64(uint64_t v, unsigned n) { n &= 63; return n ? (v << n) | (v >> (64 - n)) : v; }
inline uint64_t ror64(uint64_t v, unsigned n) { n &= 63; return n ? (v >> n) | (v << (64 - n)) : v; }

// canonical ntHash of every k-mer of seq; out.size() = len-k+1.  false if k>len or k==0 or k>64
bool nthash_all(const uint8_t *seq, size_t len, unsigned k, std::vector<uint64_t> &out);
// MultiHash variant i>=1 of canonical hash h (i==0 is h itself)
inline uint64_t multihash(uint64_t h, uint64_t i, unsigned k)
{
    uint64_t t = h * (i ^ (uint64_t(k) * 0x90b45d39fb6da1faULL));
    return t ^ (t >> 27);
}

// ---- in-memory graph (GrootGraph, src/graph/graph.go:18-34) ----------------------------------
struct Node {                       // GrootGraphNode, src/graph/node.go:13-22
    uint32_t seg_id = 0;
    std::string seq;
    std::vector<uint32_t> out;      // OutEdges (segment ids)
    std::vector<uint32_t> path_ids; // PathIDs
    std::vector<uint32_t> pos;      // Position[path_ids[i]]
    double kmer_freq = 0.0;
};

struct Window {                     // lshe.Key, src/lshe/lshe.go:17-28
    uint32_t graph = 0, node_seg = 0, offset = 0, merge_span = 0;
    std::vector<std::pair<uint32_t, uint32_t>> contained; // (segment id, count) ascending segment id
    std::vector<uint32_t> ref;
    std::vector<uint64_t> sketch;
};

struct Graph {
    uint32_t id = 0;
    bool masked = false;
    std::vector<Node> nodes;                 // SortedNodes
    std::vector<std::string> path_names;     // Paths
    std::vector<uint32_t> path_len;          // Lengths
    std::vector<Window> windows;             // canonical order within the graph
};

// raw GFA-level description handed to create_groot_graph (what gfa.GFA carries)
struct RawSegment { uint32_t name; std::string seq; double kc = 0.0; };
struct RawGraph {
    std::vector<RawSegment> segments;                    // GFA order
    std::vector<std::pair<uint32_t, uint32_t>> links;    // (from, to) GFA order
    std::vector<std::pair<std::string, std::vector<uint32_t>>> paths;
};

int read_msa_file(const std::string &file, RawGraph &out);   // gfa.ReadMSA + gfa.MSA2GFA
int read_gfa_file(const std::string &file, RawGraph &out);   // graph.LoadGFA
int create_groot_graph(const RawGraph &raw, uint32_t id, Graph &g);                 // graph.go:37-147
struct WindowSketcher { groot_sketch_fn fn; void *user; std::mutex *mu; };
int window_graph(Graph &g, unsigned w, unsigned k, unsigned s, const WindowSketcher *sketcher = nullptr);   // graph.go:229-396

} // namespace groot

struct groot_index;
namespace groot {
int check_index_params(const groot_index_params *p);
// Store + windows -> flat arrays of include/groot_index.h (window ids in canonical order: graphs ascending, then Graph::windows order)
int flatten_graphs(std::vector<Graph> &graphs, const groot_index_params &p, groot_index **out);
} // namespace groot

// owning storage behind the C handle
struct groot_index {
    groot_index_view v{};
    std::vector<uint32_t> graph_node_off, graph_path_off, node_seg_id, node_seq_off, node_edge_off, node_np_off, edges,
        np_path, np_pos, path_len, path_name_off, win_graph, win_node, win_offset, win_merge_span, win_cn_off, cn_node,
        cn_count, win_ref_off, win_ref;
    std::vector<uint8_t> graph_masked, bases;
    std::vector<uint64_t> node_mask, win_sketch;
    std::string path_names;
    void bind(); // point v.* at the vectors
};

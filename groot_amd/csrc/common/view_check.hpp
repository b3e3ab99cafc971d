// view_check.hpp -- one O(n) consistency pass over a groot_index_view, shared by the host library (after loading a
// .gidx / gob index) and the device library (before uploading a caller-supplied view).  A truncated or corrupt index
// must come back as GROOT_E_FORMAT, not as out-of-bounds reads on the host or the GPU.
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <string>

#include "groot_index.h"

namespace groot {

// empty string = consistent; otherwise the first inconsistency found
inline std::string check_index_view(const groot_index_view *v)
{
    char buf[256];
    auto bad = [&](const char *fmt, unsigned long long a = 0, unsigned long long b = 0) {
        snprintf(buf, sizeof buf, fmt, a, b);
        return std::string(buf);
    };
    if (!v) return "null view";
    if (v->kmer_size == 0 || v->sketch_size == 0) return "k-mer size / sketch size is zero";
    // offsets: monotone, first 0, last = payload size
    auto offsets = [&](const uint32_t *off, uint64_t n, uint64_t total, const char *name) -> std::string {
        if (!off) return n ? bad((std::string(name) + " is null").c_str()) : std::string();
        if (off[0] != 0) return bad((std::string(name) + "[0] is not 0").c_str());
        for (uint64_t i = 0; i < n; i++)
            if (off[i + 1] < off[i]) return bad((std::string(name) + " not monotone at %llu").c_str(), i);
        if (off[n] != total) return bad((std::string(name) + " ends at %llu, payload holds %llu").c_str(), off[n], total);
        return std::string();
    };
    std::string e;
    // payload arrays that hold something must be there
    {
        const struct { const void *p; uint64_t n; const char *name; } need[] = {
            {v->bases, v->n_bases, "bases"}, {v->edges, v->n_edges, "edges"}, {v->np_path, v->n_np, "np_path"}, {v->np_pos, v->n_np, "np_pos"},
            {v->node_mask, (uint64_t)v->n_nodes * v->path_words, "node_mask"}, {v->node_seg_id, v->n_nodes, "node_seg_id"},
            {v->path_len, v->n_paths, "path_len"}, {v->path_names, v->n_name_bytes, "path_names"},
            {v->win_graph, v->n_windows, "win_graph"}, {v->win_node, v->n_windows, "win_node"}, {v->win_offset, v->n_windows, "win_offset"},
            {v->win_merge_span, v->n_windows, "win_merge_span"}, {v->win_sketch, (uint64_t)v->n_windows * v->sketch_size, "win_sketch"},
            {v->cn_node, v->n_cn, "cn_node"}, {v->win_ref, v->n_wref, "win_ref"}};
        for (const auto &x : need)
            if (x.n && !x.p) return bad((std::string(x.name) + " is null although its count is %llu").c_str(), x.n);
    }
    if (v->n_windows && (v->window_size == 0 || v->window_size < v->kmer_size)) return bad("window size %llu below the k-mer size %llu", v->window_size, v->kmer_size);
    if (v->n_windows && v->num_window_kmers != v->window_size - v->kmer_size + 1)
        return bad("num_window_kmers=%llu does not belong to window size %llu", v->num_window_kmers, v->window_size);
    if (v->n_graphs) {
        if (!(e = offsets(v->graph_node_off, v->n_graphs, v->n_nodes, "graph_node_off")).empty()) return e;
        if (!(e = offsets(v->graph_path_off, v->n_graphs, v->n_paths, "graph_path_off")).empty()) return e;
    }
    if (v->n_nodes) {
        if (!(e = offsets(v->node_seq_off, v->n_nodes, v->n_bases, "node_seq_off")).empty()) return e;
        if (!(e = offsets(v->node_edge_off, v->n_nodes, v->n_edges, "node_edge_off")).empty()) return e;
        if (!(e = offsets(v->node_np_off, v->n_nodes, v->n_np, "node_np_off")).empty()) return e;
    }
    if (v->n_paths && !(e = offsets(v->path_name_off, v->n_paths, v->n_name_bytes, "path_name_off")).empty()) return e;
    if (v->n_windows) {
        if (!(e = offsets(v->win_cn_off, v->n_windows, v->n_cn, "win_cn_off")).empty()) return e;
        if (!(e = offsets(v->win_ref_off, v->n_windows, v->n_wref, "win_ref_off")).empty()) return e;
    }
    // path bitsets must hold every local path id
    uint32_t max_paths = 0;
    for (uint32_t g = 0; g < v->n_graphs; g++) max_paths = std::max(max_paths, v->graph_path_off[g + 1] - v->graph_path_off[g]);
    if ((uint64_t)v->path_words * 64 < max_paths) return bad("path_words=%llu cannot hold %llu paths", v->path_words, max_paths);
    // index arrays
    for (uint64_t i = 0; i < v->n_edges; i++)
        if (v->edges[i] >= v->n_nodes) return bad("edges[%llu]=%llu out of range", i, v->edges[i]);
    // edges and windows stay inside their graph
    for (uint32_t g = 0; g < v->n_graphs; g++) {
        const uint32_t n0 = v->graph_node_off[g], n1 = v->graph_node_off[g + 1], np = v->graph_path_off[g + 1] - v->graph_path_off[g];
        for (uint32_t n = n0; n < n1; n++) {
            for (uint32_t x = v->node_edge_off[n]; x < v->node_edge_off[n + 1]; x++)
                if (v->edges[x] < n0 || v->edges[x] >= n1) return bad("edge %llu of node %llu leaves its graph", x, n);
            for (uint32_t x = v->node_np_off[n]; x < v->node_np_off[n + 1]; x++)
                if (v->np_path[x] >= np) return bad("np_path[%llu]=%llu out of range", x, v->np_path[x]);
        }
    }
    for (uint32_t w = 0; w < v->n_windows; w++) {
        const uint32_t g = v->win_graph[w];
        if (g >= v->n_graphs) return bad("win_graph[%llu]=%llu out of range", w, g);
        const uint32_t n0 = v->graph_node_off[g], n1 = v->graph_node_off[g + 1], np = v->graph_path_off[g + 1] - v->graph_path_off[g];
        if (v->win_node[w] < n0 || v->win_node[w] >= n1) return bad("win_node[%llu]=%llu outside its graph", w, v->win_node[w]);
        for (uint32_t c = v->win_cn_off[w]; c < v->win_cn_off[w + 1]; c++)
            if (v->cn_node[c] < n0 || v->cn_node[c] >= n1) return bad("cn_node[%llu]=%llu outside its graph", c, v->cn_node[c]);
        for (uint32_t c = v->win_ref_off[w]; c < v->win_ref_off[w + 1]; c++)
            if (v->win_ref[c] >= np) return bad("win_ref[%llu]=%llu out of range", c, v->win_ref[c]);
    }
    return std::string();
}

} // namespace groot

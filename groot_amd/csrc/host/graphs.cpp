// graphs.cpp -- host steps after the device path: traversal -> record expansion, graph weighting
// replay, pruning and GFA output.
//   src/graph/alignment.go:113-156,263-317   records from traversals
//   src/graph/graph.go:401-451               IncrementSubPath
//   src/graph/graph.go:455-525               Prune
//   src/graph/graphio.go:19-112              SaveGraphAsGFA (dialect pinned by src/graph/test2.gfa)
#include "host_common.hpp"

#include <algorithm>
#include <thread>

#include <cstring>
#include <ctime>

using namespace groot;

extern "C" {

int groot_host_expand_alns(const groot_index_view *ix, const groot_trav *travs, const uint64_t *masks, uint64_t n_trav,
                           groot_aln *out, uint64_t cap, uint64_t *n_out)
{
    if (!ix || !n_out || (n_trav && (!travs || !masks))) return set_error(GROOT_E_INVALID, "null argument");
    const uint32_t pw = ix->path_words;
    uint64_t n = 0;
    for (uint64_t t = 0; t < n_trav; t++) {
        const groot_trav &tr = travs[t];
        if (tr.node >= ix->n_nodes || tr.graph_id >= ix->n_graphs) return set_error(GROOT_E_INVALID, "traversal %llu refers outside the index", (unsigned long long)t);
        const uint32_t np0 = ix->node_np_off[tr.node], np1 = ix->node_np_off[tr.node + 1];
        bool first = (tr.flags & GROOT_TRAV_FIRST) != 0;
        for (uint32_t w = 0; w < pw; w++) {
            uint64_t m = masks[t * pw + w];
            while (m) {
                const uint32_t p = w * 64 + (uint32_t)__builtin_ctzll(m);
                m &= m - 1;
                uint32_t pos = 0;
                bool have = false;
                for (uint32_t j = np0; j < np1; j++)
                    if (ix->np_path[j] == p) { pos = ix->np_pos[j] + tr.offset; have = true; break; }   // alignment.go:296
                if (!have) return set_error(GROOT_E_INVALID, "traversal %llu: path %u does not pass its first node", (unsigned long long)t, p);
                if (out && n < cap) {
                    groot_aln &a = out[n];
                    a.read_id = tr.read_id; a.graph_id = tr.graph_id; a.path_id = p;
                    a.ref_id = ix->graph_path_off[tr.graph_id] + p;
                    a.pos = pos;
                    a.start_clip = (tr.flags & GROOT_TRAV_START_CLIP) ? 1 : 0;
                    a.end_clip = (tr.flags & GROOT_TRAV_END_CLIP) ? 1 : 0;
                    a.rc = (tr.flags & GROOT_TRAV_RC) ? 1 : 0;
                    a.secondary = first ? 0 : 1;       // alignment.go:147-149
                }
                first = false;
                n++;
            }
        }
    }
    *n_out = n;
    return GROOT_OK;
}

// graph.go:401-451
static inline void increment_sub_path(const groot_index_view *ix, uint32_t w, double num_kmers, double *kf, uint64_t *kt)
{
    const uint32_t c0 = ix->win_cn_off[w], c1 = ix->win_cn_off[w + 1];
    if (c1 - c0 == 1) {                       // a single segment takes all the k-mers; KmerTotal untouched (:409-422)
        kf[ix->cn_node[c0]] += num_kmers;
        return;
    }
    double total = 0.0;
    for (uint32_t c = c0; c < c1; c++) {
        const uint32_t nd = ix->cn_node[c];
        total += double(ix->node_seq_off[nd + 1] - ix->node_seq_off[nd]);
    }
    for (uint32_t c = c0; c < c1; c++) {
        const uint32_t nd = ix->cn_node[c];
        const double seg_len = double(ix->node_seq_off[nd + 1] - ix->node_seq_off[nd]);
        kf[nd] += ((seg_len / total) * num_kmers) * double(ix->cn_count[c]);
    }
    kt[ix->win_graph[w]] += uint64_t(num_kmers);
}

int groot_host_weights(const groot_index_view *ix, const uint32_t *attempts, uint32_t n_q, double *kf, uint64_t *kt)
{
    if (!ix || !kf || !kt || (n_q && !attempts)) return set_error(GROOT_E_INVALID, "null argument");
    memset(kf, 0, sizeof(double) * ix->n_nodes);
    memset(kt, 0, sizeof(uint64_t) * ix->n_graphs);
    // rows (kmerCount values) that were used at all
    std::vector<uint32_t> rows;
    for (uint32_t q = 0; q < n_q; q++) {
        const uint32_t *row = attempts + (size_t)q * ix->n_windows;
        bool any = false;
        for (uint32_t w = 0; w < ix->n_windows && !any; w++) any = row[w] != 0;
        if (any) rows.push_back(q);
    }
    for (uint32_t w = 0; w < ix->n_windows; w++)
        for (uint32_t q : rows) {
            const uint32_t c = attempts[(size_t)q * ix->n_windows + w];
            for (uint32_t i = 0; i < c; i++) increment_sub_path(ix, w, double(q), kf, kt);
        }
    return GROOT_OK;
}

int groot_host_unpack_masks(const groot_index_view *ix, const groot_trav *travs, uint64_t n_trav, const uint8_t *compact, uint64_t *out)
{
    if (!ix || (n_trav && (!travs || !compact || !out))) return set_error(GROOT_E_INVALID, "null argument");
    const uint32_t pw = ix->path_words;
    uint64_t o = 0;
    for (uint64_t i = 0; i < n_trav; i++) {
        const uint32_t g = travs[i].graph_id;
        if (g >= ix->n_graphs) return set_error(GROOT_E_INVALID, "traversal %llu: graph id out of range", (unsigned long long)i);
        const uint32_t nb = std::min<uint32_t>(8 * pw, std::max<uint32_t>(1, (ix->graph_path_off[g + 1] - ix->graph_path_off[g] + 7) / 8));
        for (uint32_t x = 0; x < pw; x++) out[i * pw + x] = 0;
        memcpy(out + i * pw, compact + o, nb);             // (little endian: byte b of the set is bits 8b.. of word b / 8)
        o += nb;
    }
    return GROOT_OK;
}

int groot_host_weights_rows(const groot_index_view *ix, const uint32_t *q_values, uint32_t n_rows, const uint32_t *counts, double *kf, uint64_t *kt)
{
    if (!ix || !kf || !kt || (n_rows && (!q_values || !counts))) return set_error(GROOT_E_INVALID, "null argument");
    for (uint32_t r = 1; r < n_rows; r++)
        if (q_values[r] <= q_values[r - 1]) return set_error(GROOT_E_INVALID, "kmerCounts must be strictly ascending");
    memset(kf, 0, sizeof(double) * ix->n_nodes);
    memset(kt, 0, sizeof(uint64_t) * ix->n_graphs);
    // canonical order of the float additions: windows ascending, kmerCounts ascending within a window, one call at a time
    // (graph.go:401-451 adds a share per call).  A window only touches nodes and the counter of its own graph, so graphs are replayed
    // side by side when the windows are numbered graph by graph (they are: the index builder emits them so) -- same sums, bit for bit.
    auto replay = [&](uint32_t w0, uint32_t w1) {
        for (uint32_t w = w0; w < w1; w++)
            for (uint32_t r = 0; r < n_rows; r++) {
                const uint32_t c = counts[(size_t)r * ix->n_windows + w];
                for (uint32_t i = 0; i < c; i++) increment_sub_path(ix, w, double(q_values[r]), kf, kt);
            }
    };
    bool grouped = true;
    for (uint32_t w = 1; w < ix->n_windows; w++) grouped &= ix->win_graph[w] >= ix->win_graph[w - 1];
    const unsigned nt = grouped ? std::min<unsigned>(usable_cpus(), 16) : 1;
    if (nt <= 1 || ix->n_windows < 4096) { replay(0, ix->n_windows); return GROOT_OK; }
    // cut points at graph boundaries, about equal numbers of calls per piece
    std::vector<uint64_t> calls(ix->n_windows + 1, 0);
    for (uint32_t w = 0; w < ix->n_windows; w++) {
        uint64_t c = 0;
        for (uint32_t r = 0; r < n_rows; r++) c += counts[(size_t)r * ix->n_windows + w];
        calls[w + 1] = calls[w] + c + 1;
    }
    std::vector<uint32_t> cut{0};
    for (unsigned t = 1; t < nt; t++) {
        uint32_t w = (uint32_t)(std::lower_bound(calls.begin(), calls.end(), calls.back() * t / nt) - calls.begin());
        w = std::min(w, ix->n_windows);
        while (w > 0 && w < ix->n_windows && ix->win_graph[w] == ix->win_graph[w - 1]) w++;       // on to the next graph's first window
        if (w > cut.back() && w < ix->n_windows) cut.push_back(w);
    }
    cut.push_back(ix->n_windows);
    std::vector<std::thread> th;
    for (size_t i = 0; i + 1 < cut.size(); i++) th.emplace_back(replay, cut[i], cut[i + 1]);
    for (auto &x : th) x.join();
    return GROOT_OK;
}

int groot_host_prune(const groot_index_view *ix, const double *kf, double min_cov, uint8_t *graph_kept, uint8_t *path_kept,
                     uint8_t *node_removed)
{
    if (!ix || !kf || !graph_kept || !path_kept || !node_removed) return set_error(GROOT_E_INVALID, "null argument");
    memset(node_removed, 0, ix->n_nodes);
    for (uint32_t g = 0; g < ix->n_graphs; g++) {
        const uint32_t p0 = ix->graph_path_off[g], p1 = ix->graph_path_off[g + 1];
        for (uint32_t p = p0; p < p1; p++) path_kept[p] = 1;
        uint32_t removed_paths = 0;
        for (uint32_t n = ix->graph_node_off[g]; n < ix->graph_node_off[g + 1]; n++) {
            const double per_base = kf[n] / double(ix->node_seq_off[n + 1] - ix->node_seq_off[n]);   // :463
            if (per_base < min_cov)                                                                 // :466-471
                for (uint32_t j = ix->node_np_off[n]; j < ix->node_np_off[n + 1]; j++) {
                    const uint32_t p = p0 + ix->np_path[j];
                    if (path_kept[p]) { path_kept[p] = 0; removed_paths++; }
                    node_removed[n] = 1;
                }
        }
        graph_kept[g] = removed_paths == p1 - p0 ? 0 : 1;                                           // :475-477
    }
    return GROOT_OK;
}

int groot_host_save_gfa(const groot_index_view *ix, uint32_t g, const double *kf, const uint8_t *path_kept,
                        const uint8_t *node_removed, uint64_t total_kmers, const char *timestamp, const char *file_name,
                        int *written)
{
    if (!ix || !kf || !file_name || g >= ix->n_graphs) return set_error(GROOT_E_INVALID, "bad argument");
    if (written) *written = 0;
    const uint32_t n0 = ix->graph_node_off[g], n1 = ix->graph_node_off[g + 1];
    bool used = false;
    for (uint32_t n = n0; n < n1; n++)
        if (!(node_removed && node_removed[n]) && kf[n] > 0) { used = true; break; }
    if (!used) return GROOT_OK;                                                                    // graphio.go:67-69
    char stamp[64];
    if (!timestamp) {
        time_t now = time(nullptr);
        struct tm tmv;
        localtime_r(&now, &tmv);
        strftime(stamp, sizeof stamp, "%a %b %e %H:%M:%S %Y", &tmv);   // Go "Mon Jan _2 15:04:05 2006"
        timestamp = stamp;
    }
    std::string out;
    out += "H\tVN:Z:1\n";
    out += "#\tvariation graph created by groot (version " + std::string(groot_host_version()) + ") at: " + timestamp + "\n";
    out += "#\tthis graph is approximately weighted using k-mer frequencies from projected read sketches (total k-mers projected across all graphs: " +
           std::to_string(total_kmers) + ")\n";
    // the gfa writer emits all segments, then all links, then all paths
    std::string links;
    for (uint32_t n = n0; n < n1; n++) {
        if (node_removed && node_removed[n]) continue;                                              // :34-36
        const std::string seg = std::to_string(ix->node_seg_id[n]);
        const uint32_t s0 = ix->node_seq_off[n], s1 = ix->node_seq_off[n + 1];
        out += "S\t" + seg + "\t";
        out.append((const char *)ix->bases + s0, s1 - s0);
        out += "\tLN:i:" + std::to_string(s1 - s0) + "\tKC:i:" + std::to_string((long long)kf[n]) + "\n";   // :49 int(KmerFreq)
        for (uint32_t e = ix->node_edge_off[n]; e < ix->node_edge_off[n + 1]; e++) {
            const uint32_t to = ix->edges[e];
            if (node_removed && node_removed[to]) continue;                                         // Prune drops edges to deleted nodes (graph.go:506-513)
            links += "L\t" + seg + "\t+\t" + std::to_string(ix->node_seg_id[to]) + "\t+\t0M\n";
        }
    }
    out += links;
    const uint32_t p0 = ix->graph_path_off[g], p1 = ix->graph_path_off[g + 1];
    for (uint32_t p = p0; p < p1; p++) {
        if (path_kept && !path_kept[p]) continue;                                                   // Lengths[id]==0 (:73-75)
        std::string segs, overlaps;
        for (uint32_t n = n0; n < n1; n++) {
            if (node_removed && node_removed[n]) continue;
            for (uint32_t j = ix->node_np_off[n]; j < ix->node_np_off[n + 1]; j++)
                if (ix->np_path[j] == p - p0) {
                    if (!segs.empty()) { segs += ","; overlaps += ","; }
                    segs += std::to_string(ix->node_seg_id[n]) + "+";
                    overlaps += std::to_string(ix->node_seq_off[n + 1] - ix->node_seq_off[n]) + "M";
                    break;
                }
        }
        out += "P\t";
        out.append(ix->path_names + ix->path_name_off[p], ix->path_name_off[p + 1] - ix->path_name_off[p]);
        out += "\t" + segs + "\t" + overlaps + "\n";
    }
    FILE *f = fopen(file_name, "wb");
    if (!f) return set_error(GROOT_E_IO, "cannot create %s", file_name);
    const size_t wr = fwrite(out.data(), 1, out.size(), f);
    if (fclose(f) != 0 || wr != out.size()) return set_error(GROOT_E_IO, "write to %s failed", file_name);
    if (written) *written = 1;
    return GROOT_OK;
}

} // extern "C"

// index.cpp -- `groot index` side: MSA/GFA -> GrootGraph -> sketched windows -> flat index.
//
// Restates (does not copy) the reference's index pipeline so that there is an index to align
// against where no Go-built groot.gg / groot.lshe exists:
//   src/pipeline/index.go:37-71   MSAconverter      (gfa.ReadMSA, gfa.MSA2GFA, CreateGrootGraph, mask)
//   src/graph/graph.go:37-218     CreateGrootGraph, topoSort, traverse
//   src/graph/graph.go:575-644    GetPaths (Position), Graph2Seqs (Lengths)
//   src/graph/graph.go:229-396    WindowGraph (run-length merge, both quirks kept)
//   src/pipeline/index.go:184-211 SketchIndexer ("g%dn%do%d-%d" lookup keys)
// gfa.MSA2GFA lives in github.com/will-rowe/gfa (not in /root/reference); its node numbering is
// Go-map-order dependent, so segment ids produced here are deterministic
// but not those of a Go-built index -- parity is defined on id-free coordinates.
#include "host_common.hpp"
#include "../common/cpus.hpp"
#include "../common/view_check.hpp"

#include <algorithm>
#include <sched.h>
#include <atomic>
#include <cstring>
#include <dirent.h>
#include <fstream>
#include <map>
#include <mutex>
#include <sstream>
#include <thread>
#include <unordered_map>

namespace groot {

static thread_local std::string g_err;

int set_error(int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

// ---------------------------------------------------------------------------------------------
// ntHash (will-rowe/nthash v0.2.0)
// ---------------------------------------------------------------------------------------------
NtHashTables::NtHashTables()
{
    const uint64_t A = 0x3c8bfbb395c60474ULL, C = 0x3193c18562a02b4cULL, G = 0x20323ed082572324ULL,
                   T = 0x295549f54be24456ULL;
    for (auto &s : seed) s = 0;
    // slots 0..7 double as the complement table, fetched with (base & 0x07)
    seed[1] = T; seed[3] = G; seed[4] = A; seed[5] = A; seed[7] = C;
    seed['A'] = seed['a'] = A;
    seed['C'] = seed['c'] = C;
    seed['G'] = seed['g'] = G;
    seed['T'] = seed['t'] = T;
    seed['U'] = seed['u'] = T;
}
const NtHashTables &nthash_tables()
{
    static const NtHashTables t;
    return t;
}

bool nthash_all(const uint8_t *seq, size_t len, unsigned k, std::vector<uint64_t> &out)
{
    out.clear();
    if (k == 0 || k > 64 || k > len) return false;
    const uint64_t *tab = nthash_tables().seed;
    uint64_t fh = 0, rh = 0;
    for (unsigned i = 0; i < k; i++) fh = rol64(fh, 1) ^ tab[seq[i]];
    for (unsigned i = 0; i < k; i++) rh = rol64(rh, 1) ^ tab[seq[k - 1 - i] & 7];
    out.reserve(len - k + 1);
    out.push_back(std::min(fh, rh));
    for (size_t i = 1; i + k <= len; i++) {
        uint8_t prev = seq[i - 1], end = seq[i + k - 1];
        fh = rol64(fh, 1) ^ rol64(tab[prev], k) ^ tab[end];
        rh = ror64(rh, 1) ^ ror64(tab[prev & 7], 1) ^ rol64(tab[end & 7], k - 1);
        out.push_back(std::min(fh, rh));
    }
    return true;
}

// ---------------------------------------------------------------------------------------------
// MSA -> raw graph  (gfa.ReadMSA + gfa.MSA2GFA restated: SURVEY Appendix A.1)
// ---------------------------------------------------------------------------------------------
int read_msa_file(const std::string &file, RawGraph &out)
{
    std::ifstream in(file);
    if (!in) return set_error(GROOT_E_IO, "cannot open MSA file %s", file.c_str());
    std::vector<std::pair<std::string, std::string>> rows;
    std::string line;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty()) continue;
        if (line[0] == '>') {
            std::string id = line.substr(1);
            size_t sp = id.find_first_of(" \t");
            if (sp != std::string::npos) id.resize(sp);
            rows.emplace_back(id, std::string());
        } else {
            if (rows.empty()) return set_error(GROOT_E_FORMAT, "%s: sequence before first header", file.c_str());
            rows.back().second += line;
        }
    }
    // the vsearch consensus row is not a sequence of the cluster
    rows.erase(std::remove_if(rows.begin(), rows.end(), [](const auto &r) { return r.first == "consensus"; }), rows.end());
    if (rows.empty()) return set_error(GROOT_E_FORMAT, "%s: no sequences", file.c_str());
    const size_t ncol = rows[0].second.size();
    for (auto &r : rows)
        if (r.second.size() != ncol) return set_error(GROOT_E_FORMAT, "%s: rows of unequal length", file.c_str());

    // one node per distinct non-gap letter per column, numbered by first appearance (row order)
    struct MNode { char base; std::vector<uint32_t> rows; std::vector<uint32_t> out, in; };
    std::vector<MNode> nodes;
    std::vector<std::vector<uint32_t>> row_nodes(rows.size());
    for (size_t c = 0; c < ncol; c++) {
        int col_node[256];
        std::fill(std::begin(col_node), std::end(col_node), -1);
        for (size_t r = 0; r < rows.size(); r++) {
            unsigned char b = (unsigned char)rows[r].second[c];
            if (b == '-' || b == '.') continue;
            if (col_node[b] < 0) {
                col_node[b] = (int)nodes.size();
                nodes.push_back(MNode{(char)b, {}, {}, {}});
            }
            nodes[col_node[b]].rows.push_back((uint32_t)r);
            row_nodes[r].push_back((uint32_t)col_node[b]);
        }
    }
    // edges between consecutive nodes of every row (unique)
    for (auto &rn : row_nodes)
        for (size_t i = 0; i + 1 < rn.size(); i++) {
            auto &o = nodes[rn[i]].out;
            if (std::find(o.begin(), o.end(), rn[i + 1]) == o.end()) {
                o.push_back(rn[i + 1]);
                nodes[rn[i + 1]].in.push_back(rn[i]);
            }
        }
    // squash maximal non-branching chains whose members carry the same rows
    std::vector<int> head_of(nodes.size(), -1);     // chain head for every node
    std::vector<uint32_t> seg_of(nodes.size(), 0);  // 1-based segment id
    std::vector<std::vector<uint32_t>> chains;
    for (uint32_t n = 0; n < nodes.size(); n++) {
        if (head_of[n] >= 0) continue;
        // n is a head unless it can be absorbed by its single predecessor (then it was already absorbed,
        // because predecessors are created in earlier columns)
        std::vector<uint32_t> chain{n};
        head_of[n] = (int)n;
        uint32_t cur = n;
        while (nodes[cur].out.size() == 1) {
            uint32_t nx = nodes[cur].out[0];
            if (nodes[nx].in.size() != 1 || nodes[nx].rows != nodes[cur].rows || head_of[nx] >= 0) break;
            chain.push_back(nx);
            head_of[nx] = (int)n;
            cur = nx;
        }
        chains.push_back(std::move(chain));
    }
    out = RawGraph{};
    for (size_t s = 0; s < chains.size(); s++) {
        RawSegment seg;
        seg.name = (uint32_t)s + 1;
        for (uint32_t n : chains[s]) { seg.seq.push_back(nodes[n].base); seg_of[n] = seg.name; }
        out.segments.push_back(std::move(seg));
    }
    for (size_t s = 0; s < chains.size(); s++) {
        uint32_t tail = chains[s].back();
        for (uint32_t nx : nodes[tail].out) out.links.emplace_back((uint32_t)s + 1, seg_of[nx]);
    }
    for (size_t r = 0; r < rows.size(); r++) {
        std::vector<uint32_t> segs;
        for (uint32_t n : row_nodes[r])
            if (segs.empty() || segs.back() != seg_of[n]) segs.push_back(seg_of[n]);
        if (segs.empty()) return set_error(GROOT_E_FORMAT, "%s: row %s is all gaps", file.c_str(), rows[r].first.c_str());
        out.paths.emplace_back(rows[r].first, std::move(segs));
    }
    return GROOT_OK;
}

// ---------------------------------------------------------------------------------------------
// GFA -> raw graph (graph.LoadGFA, graphio.go:115-138; dialect of src/graph/test.gfa / test2.gfa)
// ---------------------------------------------------------------------------------------------
static std::vector<std::string> split(const std::string &s, char d)
{
    std::vector<std::string> f;
    size_t a = 0;
    for (;;) {
        size_t b = s.find(d, a);
        if (b == std::string::npos) { f.push_back(s.substr(a)); break; }
        f.push_back(s.substr(a, b - a));
        a = b + 1;
    }
    return f;
}

int read_gfa_file(const std::string &file, RawGraph &out)
{
    std::ifstream in(file);
    if (!in) return set_error(GROOT_E_IO, "cannot open GFA file %s", file.c_str());
    out = RawGraph{};
    std::string line;
    auto to_id = [&](const std::string &s, uint32_t &id) {
        if (s.empty()) return false;
        char *e = nullptr;
        unsigned long v = strtoul(s.c_str(), &e, 10);
        if (*e) return false;
        id = (uint32_t)v;
        return true;
    };
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty()) continue;
        auto f = split(line, '\t');
        if (f[0] == "S") {
            if (f.size() < 3) return set_error(GROOT_E_FORMAT, "%s: short S line", file.c_str());
            RawSegment s;
            if (!to_id(f[1], s.name))   // graph.go:59-62 strconv.Atoi
                return set_error(GROOT_E_FORMAT, "could not convert segment name from GFA into an int for groot graph: %s", f[1].c_str());
            s.seq = f[2];
            for (size_t i = 3; i < f.size(); i++)
                if (f[i].rfind("KC:i:", 0) == 0) s.kc = atof(f[i].c_str() + 5);
            out.segments.push_back(std::move(s));
        } else if (f[0] == "L") {
            if (f.size() < 5) return set_error(GROOT_E_FORMAT, "%s: short L line", file.c_str());
            uint32_t a, b;
            if (!to_id(f[1], a) || !to_id(f[3], b)) return set_error(GROOT_E_FORMAT, "%s: non-integer link", file.c_str());
            out.links.emplace_back(a, b);
        } else if (f[0] == "P") {
            if (f.size() < 3) return set_error(GROOT_E_FORMAT, "%s: short P line", file.c_str());
            std::vector<uint32_t> segs;
            for (auto &t : split(f[2], ',')) {
                if (t.empty()) continue;
                std::string n = t;
                if (n.back() == '+') n.pop_back();   // graph.go:120
                uint32_t id;
                if (!to_id(n, id)) return set_error(GROOT_E_FORMAT, "%s: bad path segment %s", file.c_str(), t.c_str());
                segs.push_back(id);
            }
            out.paths.emplace_back(f[1], std::move(segs));
        }
    }
    if (out.segments.empty()) return set_error(GROOT_E_FORMAT, "%s: no segments", file.c_str());
    return GROOT_OK;
}

// ---------------------------------------------------------------------------------------------
// CreateGrootGraph (graph.go:37-147) + topoSort/traverse (:150-218) + GetPaths/Graph2Seqs (:575-644)
// ---------------------------------------------------------------------------------------------
static char base_check(char c)   // seqio.go:72-91
{
    switch (c) {
    case 'a': case 'A': return 'A';
    case 'c': case 'C': return 'C';
    case 'g': case 'G': return 'G';
    case 't': case 'T': return 'T';
    case 'n': case 'N': return 'N';
    default: return 'N';
    }
}

int create_groot_graph(const RawGraph &raw, uint32_t id, Graph &g)
{
    g = Graph{};
    g.id = id;
    std::vector<Node> pre(raw.segments.size());
    std::unordered_map<uint32_t, uint32_t> lookup;
    for (size_t i = 0; i < raw.segments.size(); i++) {
        Node &n = pre[i];
        n.seg_id = raw.segments[i].name;
        n.seq = raw.segments[i].seq;
        for (auto &c : n.seq) c = base_check(c);
        n.kmer_freq = raw.segments[i].kc;
        if (n.seq.empty()) return set_error(GROOT_E_FORMAT, "graph %u: empty segment %u", id, n.seg_id);
        if (!lookup.emplace(n.seg_id, (uint32_t)i).second)
            return set_error(GROOT_E_FORMAT, "graph contains duplicate nodes (identical segment IDs)");
    }
    for (auto &l : raw.links) {
        auto a = lookup.find(l.first), b = lookup.find(l.second);
        if (a == lookup.end() || b == lookup.end()) return set_error(GROOT_E_FORMAT, "graph %u: link to unknown segment", id);
        pre[a->second].out.push_back(l.second);
    }
    for (size_t p = 0; p < raw.paths.size(); p++) {
        g.path_names.push_back(raw.paths[p].first);
        for (uint32_t seg : raw.paths[p].second) {
            auto it = lookup.find(seg);
            if (it == lookup.end()) return set_error(GROOT_E_FORMAT, "graph %u: path through unknown segment %u", id, seg);
            pre[it->second].path_ids.push_back((uint32_t)p);
        }
    }
    if (pre.size() > 1) {
        // topoSort: every node (GFA order) is a start; traverse visits OutEdges sorted descending
        // (this permanently reorders OutEdges, graph.go:203) and prepends finished nodes.
        std::vector<uint8_t> state(pre.size(), 0);   // 0 = in nodeMap, 1 = seen (on stack), 2 = done
        std::vector<uint32_t> finished;              // post-order; SortedNodes = reverse
        finished.reserve(pre.size());
        struct Frame { uint32_t node; size_t edge; };
        std::vector<Frame> stack;
        for (uint32_t start = 0; start < pre.size(); start++) {
            if (state[start] != 0) continue;
            state[start] = 1;
            std::sort(pre[start].out.begin(), pre[start].out.end(), std::greater<uint32_t>());
            stack.push_back({start, 0});
            while (!stack.empty()) {
                Frame &f = stack.back();
                Node &n = pre[f.node];
                if (f.edge < n.out.size()) {
                    uint32_t child = lookup[n.out[f.edge++]];
                    if (state[child] == 0) {
                        state[child] = 1;
                        std::sort(pre[child].out.begin(), pre[child].out.end(), std::greater<uint32_t>());
                        stack.push_back({child, 0});
                    }
                } else {
                    state[f.node] = 2;
                    finished.push_back(f.node);
                    stack.pop_back();
                }
            }
        }
        g.nodes.reserve(pre.size());
        for (auto it = finished.rbegin(); it != finished.rend(); ++it) g.nodes.push_back(std::move(pre[*it]));
    } else {
        g.nodes = std::move(pre);
    }
    // GetPaths: Position[pathID] = running length over SortedNodes that list the path
    g.path_len.assign(g.path_names.size(), 0);
    for (auto &n : g.nodes) n.pos.assign(n.path_ids.size(), 0);
    for (uint32_t p = 0; p < g.path_names.size(); p++) {
        uint32_t ref_len = 0;
        for (auto &n : g.nodes)
            for (size_t j = 0; j < n.path_ids.size(); j++)
                if (n.path_ids[j] == p) {
                    n.pos[j] = ref_len;
                    ref_len += (uint32_t)n.seq.size();
                }
        g.path_len[p] = ref_len;
    }
    return GROOT_OK;
}

// ---------------------------------------------------------------------------------------------
// WindowGraph (graph.go:229-396)
// ---------------------------------------------------------------------------------------------
int window_graph(Graph &g, unsigned w, unsigned k, unsigned s, const WindowSketcher *sketcher)
{
    g.windows.clear();
    // key "g%dn%do%d" -> windows at that node+offset, in arrival order (paths ascending = one
    // valid goroutine schedule, chosen as canonical)
    std::map<std::pair<uint32_t, uint32_t>, std::vector<Window>> lookup;
    std::vector<uint64_t> kh;
    for (uint32_t p = 0; p < g.path_names.size(); p++) {
        const uint32_t plen = g.path_len[p];
        if (plen < w) return set_error(GROOT_E_INVALID, "graph contains sequence < window size");
        std::string pseq;
        std::vector<uint32_t> segs(plen), offs(plen);
        {
            uint32_t it = 0;
            for (auto &n : g.nodes)
                for (uint32_t pid : n.path_ids)
                    if (pid == p)
                        for (uint32_t o = 0; o < n.seq.size(); o++) {
                            if (it >= plen) return set_error(GROOT_E_FORMAT, "windowing did not traverse entire path");
                            segs[it] = n.seg_id; offs[it] = o; it++;
                            pseq.push_back(n.seq[o]);
                        }
            if (it != plen) return set_error(GROOT_E_FORMAT, "windowing did not traverse entire path");
        }
        const uint32_t num_windows = plen - w + 1, wk = w - k + 1;
        std::vector<uint64_t> mh, all_sk;
        if (sketcher && sketcher->fn) {
            // external sketcher (the device's RunMinHash mirror): every window of the path as one batch
            std::vector<uint8_t> cat((size_t)num_windows * w);
            std::vector<uint64_t> offs64(num_windows + 1);
            for (uint32_t i = 0; i < num_windows; i++) {
                memcpy(cat.data() + (size_t)i * w, pseq.data() + i, w);
                offs64[i] = (uint64_t)i * w;
            }
            offs64[num_windows] = (uint64_t)num_windows * w;
            all_sk.resize((size_t)num_windows * s);
            std::lock_guard<std::mutex> lock(*sketcher->mu);
            if (sketcher->fn(sketcher->user, cat.data(), offs64.data(), num_windows, all_sk.data()))
                return set_error(GROOT_E_DEVICE, "window sketch callback failed");
        } else {
            // per-k-mer MultiHash values once per path (a k-mer's ntHash does not depend on where the
            // rolling started), then each window's KHF sketch = per-slot min over its w-k+1 k-mers
            if (!nthash_all((const uint8_t *)pseq.data(), pseq.size(), k, kh))
                return set_error(GROOT_E_INVALID, "k-mer size %u does not fit path of length %u", k, plen);
            const size_t nk = kh.size();
            mh.resize(nk * s);
            for (size_t j = 0; j < nk; j++) {
                mh[j * s] = kh[j];
                for (unsigned i = 1; i < s; i++) mh[j * s + i] = multihash(kh[j], i, k);
            }
        }
        Window holder;
        bool sketch_sent = false;
        std::vector<uint64_t> sk(s);
        std::map<uint32_t, uint32_t> cn;   // holder.ContainedNodes
        auto emit = [&]() {
            Window out = holder;
            out.contained.assign(cn.begin(), cn.end());
            auto &lst = lookup[{out.node_seg, out.offset}];
            for (auto &ex : lst)
                if (ex.sketch == out.sketch) {
                    // quirk 2 (graph.go:361-373): only the ContainedNodes map is shared with the stored
                    // window; the Ref append and MergeSpan max go to a loop-variable copy and are lost
                    std::map<uint32_t, uint32_t> m(ex.contained.begin(), ex.contained.end());
                    for (auto &kv : out.contained) m[kv.first] += kv.second;
                    ex.contained.assign(m.begin(), m.end());
                    return;
                }
            lst.push_back(std::move(out));
        };
        for (uint32_t i = 0; i < num_windows; i++) {
            if (!all_sk.empty()) std::copy(all_sk.begin() + (size_t)i * s, all_sk.begin() + (size_t)(i + 1) * s, sk.begin());
            else {
                for (unsigned x = 0; x < s; x++) sk[x] = UINT64_MAX;
                for (uint32_t j = i; j < i + wk; j++)
                    for (unsigned x = 0; x < s; x++) sk[x] = std::min(sk[x], mh[(size_t)j * s + x]);
            }
            bool merge = false;
            if (i != 0) {
                if (holder.sketch != sk) { emit(); sketch_sent = true; }   // :303-305
                else merge = true;
            }
            if (!merge) {                                                   // :312-323
                holder = Window{};
                holder.graph = g.id; holder.node_seg = segs[i]; holder.offset = offs[i];
                holder.ref = {p}; holder.sketch = sk; holder.merge_span = 0;
                cn.clear();
            }
            for (uint32_t y = i; y < i + w; y++) cn[segs[y]]++;             // :326-328
            if (merge) holder.merge_span++;                                 // :331-333
            // quirk 1 (:336-338): the final holder is only sent if nothing was sent before
            if (!sketch_sent && i == num_windows - 1) emit();
        }
    }
    for (auto &kv : lookup)
        for (auto &win : kv.second) g.windows.push_back(std::move(win));
    // std::map iterates (segment id, offset) ascending and list order is the "-%d" suffix: canonical
    if (g.windows.empty()) return set_error(GROOT_E_FORMAT, "no sketches produced after windowing graph %u", g.id);
    return GROOT_OK;
}

} // namespace groot

using namespace groot;

// ---------------------------------------------------------------------------------------------
// flatten + C ABI
// ---------------------------------------------------------------------------------------------
void groot_index::bind()
{
    v.n_graphs = (uint32_t)graph_masked.size();
    v.n_nodes = (uint32_t)node_seg_id.size();
    v.n_edges = (uint32_t)edges.size();
    v.n_paths = (uint32_t)path_len.size();
    v.n_windows = (uint32_t)win_graph.size();
    v.n_bases = bases.size();
    v.n_np = np_path.size();
    v.n_cn = cn_node.size();
    v.n_wref = win_ref.size();
    v.n_name_bytes = path_names.size();
    v.graph_node_off = graph_node_off.data(); v.graph_path_off = graph_path_off.data(); v.graph_masked = graph_masked.data();
    v.node_seg_id = node_seg_id.data(); v.node_seq_off = node_seq_off.data(); v.node_edge_off = node_edge_off.data();
    v.node_np_off = node_np_off.data(); v.node_mask = node_mask.data(); v.bases = bases.data(); v.edges = edges.data();
    v.np_path = np_path.data(); v.np_pos = np_pos.data(); v.path_len = path_len.data();
    v.path_name_off = path_name_off.data(); v.path_names = path_names.data();
    v.win_graph = win_graph.data(); v.win_node = win_node.data(); v.win_offset = win_offset.data();
    v.win_merge_span = win_merge_span.data(); v.win_cn_off = win_cn_off.data(); v.cn_node = cn_node.data();
    v.cn_count = cn_count.data(); v.win_ref_off = win_ref_off.data(); v.win_ref = win_ref.data();
    v.win_sketch = win_sketch.data();
}

int groot::flatten_graphs(std::vector<Graph> &graphs, const groot_index_params &p, groot_index **out)
{
    auto idx = new groot_index();
    auto &v = idx->v;
    v.kmer_size = p.kmer_size; v.sketch_size = p.sketch_size; v.window_size = p.window_size;
    v.num_part = p.num_part; v.max_k = p.max_k; v.num_window_kmers = p.window_size - p.kmer_size + 1;
    size_t max_paths = 1;
    for (auto &g : graphs) max_paths = std::max(max_paths, g.path_names.size());
    v.path_words = (uint32_t)((max_paths + 63) / 64);
    idx->graph_node_off.push_back(0);
    idx->graph_path_off.push_back(0);
    idx->node_seq_off.push_back(0);
    idx->node_edge_off.push_back(0);
    idx->node_np_off.push_back(0);
    idx->path_name_off.push_back(0);
    idx->win_cn_off.push_back(0);
    idx->win_ref_off.push_back(0);
    for (auto &g : graphs) {
        const uint32_t node_base = (uint32_t)idx->node_seg_id.size();
        std::unordered_map<uint32_t, uint32_t> lookup;   // NodeLookup
        for (uint32_t i = 0; i < g.nodes.size(); i++) lookup[g.nodes[i].seg_id] = node_base + i;
        for (auto &n : g.nodes) {
            idx->node_seg_id.push_back(n.seg_id);
            idx->bases.insert(idx->bases.end(), n.seq.begin(), n.seq.end());
            idx->node_seq_off.push_back((uint32_t)idx->bases.size());
            for (uint32_t e : n.out) idx->edges.push_back(lookup.at(e));
            idx->node_edge_off.push_back((uint32_t)idx->edges.size());
            size_t m0 = idx->node_mask.size();
            idx->node_mask.resize(m0 + v.path_words, 0);
            for (size_t j = 0; j < n.path_ids.size(); j++) {
                idx->np_path.push_back(n.path_ids[j]);
                idx->np_pos.push_back(n.pos[j]);
                idx->node_mask[m0 + n.path_ids[j] / 64] |= 1ULL << (n.path_ids[j] % 64);
            }
            idx->node_np_off.push_back((uint32_t)idx->np_path.size());
        }
        idx->graph_node_off.push_back((uint32_t)idx->node_seg_id.size());
        for (size_t q = 0; q < g.path_names.size(); q++) {
            idx->path_len.push_back(g.path_len[q]);
            idx->path_names += g.path_names[q];
            idx->path_name_off.push_back((uint32_t)idx->path_names.size());
        }
        idx->graph_path_off.push_back((uint32_t)idx->path_len.size());
        idx->graph_masked.push_back(g.masked ? 1 : 0);
        for (auto &w : g.windows) {
            idx->win_graph.push_back(g.id);
            idx->win_node.push_back(lookup.at(w.node_seg));
            idx->win_offset.push_back(w.offset);
            idx->win_merge_span.push_back(w.merge_span);
            for (auto &kv : w.contained) {
                idx->cn_node.push_back(lookup.at(kv.first));
                idx->cn_count.push_back(kv.second);
            }
            idx->win_cn_off.push_back((uint32_t)idx->cn_node.size());
            idx->win_ref.insert(idx->win_ref.end(), w.ref.begin(), w.ref.end());
            idx->win_ref_off.push_back((uint32_t)idx->win_ref.size());
            idx->win_sketch.insert(idx->win_sketch.end(), w.sketch.begin(), w.sketch.end());
        }
    }
    idx->bind();
    *out = idx;
    return GROOT_OK;
}

int groot::check_index_params(const groot_index_params *p)
{
    if (!p) return set_error(GROOT_E_INVALID, "null index params");
    if (p->kmer_size == 0 || p->kmer_size > 64) return set_error(GROOT_E_UNSUPPORTED, "k-mer size must be in [1,64]");
    if (p->kmer_size > p->window_size) return set_error(GROOT_E_INVALID, "supplied k-mer size greater than read length");
    if (p->sketch_size == 0 || p->max_k == 0 || p->num_part == 0) return set_error(GROOT_E_INVALID, "sketch size, maxK and numPart must be > 0");
    if (p->sketch_size < p->max_k) return set_error(GROOT_E_INVALID, "sketch size smaller than maxK");
    return GROOT_OK;
}

static int build_from_files(const char *const *files, uint32_t n_files, const groot_index_params *p, bool gfa, groot_index **out,
                            const WindowSketcher *sketcher = nullptr)
{
    if (int rc = check_index_params(p)) return rc;
    if (!files || !n_files || !out) return set_error(GROOT_E_INVALID, "no input files");
    std::vector<Graph> graphs(n_files);
    std::vector<std::string> errs(n_files);
    std::vector<int> rcs(n_files, 0);
    unsigned nt = p->n_threads ? p->n_threads : usable_cpus();
    nt = std::min<unsigned>(nt, n_files);
    std::atomic<uint32_t> next{0};
    auto work = [&]() {
        for (;;) {
            uint32_t i = next.fetch_add(1);
            if (i >= n_files) break;
            RawGraph raw;
            int rc = gfa ? read_gfa_file(files[i], raw) : read_msa_file(files[i], raw);
            if (!rc) rc = create_groot_graph(raw, i, graphs[i]);
            if (!rc) {
                // src/pipeline/index.go:58-66: mask graphs holding a sequence shorter than the window
                for (uint32_t len : graphs[i].path_len)
                    if (len < p->window_size) { graphs[i].masked = true; break; }
                if (!graphs[i].masked) rc = window_graph(graphs[i], p->window_size, p->kmer_size, p->sketch_size, sketcher);
            }
            if (rc) { rcs[i] = rc; errs[i] = groot_host_last_error(); }
        }
    };
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++) th.emplace_back(work);
    for (auto &t : th) t.join();
    for (uint32_t i = 0; i < n_files; i++)
        if (rcs[i]) return set_error(rcs[i], "%s: %s", files[i], errs[i].c_str());
    size_t n_sketched = 0;
    for (auto &g : graphs) n_sketched += g.masked ? 0 : 1;
    if (!n_sketched) return set_error(GROOT_E_INVALID, "could not create and sketch any graphs");
    return flatten_graphs(graphs, *p, out);
}

// ---- .gidx file: header + raw arrays ---------------------------------------------------------
static const char GIDX_MAGIC[8] = {'G', 'R', 'O', 'O', 'T', 'I', 'X', '1'};

template <class T> static void put_vec(std::ofstream &o, const std::vector<T> &v)
{
    uint64_t n = v.size();
    o.write((const char *)&n, 8);
    if (n) o.write((const char *)v.data(), (std::streamsize)(n * sizeof(T)));
}
template <class T> static bool get_vec(std::ifstream &i, std::vector<T> &v)
{
    uint64_t n = 0;
    if (!i.read((char *)&n, 8)) return false;
    if (n > (1ULL << 36)) return false;
    v.resize(n);
    if (n && !i.read((char *)v.data(), (std::streamsize)(n * sizeof(T)))) return false;
    return true;
}

namespace groot {
unsigned usable_cpus() { return granted_cpus(); }
} // namespace groot

extern "C" {

uint32_t groot_host_usable_cpus(void) { return groot::usable_cpus(); }

const char *groot_host_last_error(void) { return g_err.c_str(); }
const char *groot_host_version(void) { return "1.1.2"; }   // src/version/version.go:5-17

void groot_index_params_default(groot_index_params *p)
{
    if (!p) return;
    p->kmer_size = 31; p->sketch_size = 21; p->window_size = 100; p->num_part = 8; p->max_k = 4;
    p->max_sketch_span = 30; p->n_threads = 0; p->reserved = 0;
}

int groot_index_build_msa_files(const char *const *files, uint32_t n, const groot_index_params *p, groot_index **out)
{
    return build_from_files(files, n, p, false, out);
}
int groot_index_build_gfa_files(const char *const *files, uint32_t n, const groot_index_params *p, groot_index **out)
{
    return build_from_files(files, n, p, true, out);
}

int groot_index_build_msa_dir(const char *msa_dir, const groot_index_params *p, groot_index **out)
{
    return groot_index_build_msa_dir_with(msa_dir, p, nullptr, nullptr, out);
}

int groot_index_build_msa_dir_with(const char *msa_dir, const groot_index_params *p, groot_sketch_fn fn, void *user, groot_index **out)
{
    if (!msa_dir) return set_error(GROOT_E_INVALID, "null msa dir");
    DIR *d = opendir(msa_dir);
    if (!d) return set_error(GROOT_E_IO, "cannot open directory %s", msa_dir);
    std::vector<std::string> names;
    while (dirent *e = readdir(d)) {
        std::string n = e->d_name;   // cmd/index.go:143 Glob(msaDir + "/cluster*.msa")
        if (n.size() > 11 && n.rfind("cluster", 0) == 0 && n.compare(n.size() - 4, 4, ".msa") == 0) names.push_back(n);
    }
    closedir(d);
    if (names.empty())
        return set_error(GROOT_E_INVALID, "no MSA files found that passed the file checks (make sure filenames follow 'cluster-DD.msa' convention)");
    std::sort(names.begin(), names.end());
    std::vector<std::string> full;
    for (auto &n : names) full.push_back(std::string(msa_dir) + "/" + n);
    std::vector<const char *> ptrs;
    for (auto &f : full) ptrs.push_back(f.c_str());
    std::mutex mu;
    WindowSketcher sk{fn, user, &mu};
    return build_from_files(ptrs.data(), (uint32_t)ptrs.size(), p, false, out, fn ? &sk : nullptr);
}

void groot_index_get_view(const groot_index *idx, groot_index_view *view)
{
    if (idx && view) *view = idx->v;
}
void groot_index_free(groot_index *idx) { delete idx; }

int groot_host_window_sketch(const uint8_t *seq, uint32_t len, uint32_t k, uint32_t s, uint64_t *sketch)
{
    std::vector<uint64_t> kh;
    if (!seq || !sketch || s == 0) return set_error(GROOT_E_INVALID, "bad sketch arguments");
    if (!nthash_all(seq, len, k, kh)) return set_error(GROOT_E_SHORT_READ, "k size is greater than sequence length (%u vs %u)", k, len);
    for (uint32_t i = 0; i < s; i++) sketch[i] = UINT64_MAX;
    for (uint64_t h : kh) {
        sketch[0] = std::min(sketch[0], h);
        for (uint32_t i = 1; i < s; i++) sketch[i] = std::min(sketch[i], multihash(h, i, k));
    }
    return GROOT_OK;
}

int groot_index_save(const groot_index *idx, const char *path)
{
    if (!idx || !path) return set_error(GROOT_E_INVALID, "null argument");
    std::ofstream o(path, std::ios::binary);
    if (!o) return set_error(GROOT_E_IO, "cannot create %s", path);
    o.write(GIDX_MAGIC, 8);
    uint32_t hdr[8] = {idx->v.kmer_size, idx->v.sketch_size, idx->v.window_size, idx->v.num_part,
                       idx->v.max_k, idx->v.num_window_kmers, idx->v.path_words, 0};
    o.write((const char *)hdr, sizeof hdr);
    put_vec(o, idx->graph_node_off); put_vec(o, idx->graph_path_off); put_vec(o, idx->graph_masked);
    put_vec(o, idx->node_seg_id); put_vec(o, idx->node_seq_off); put_vec(o, idx->node_edge_off);
    put_vec(o, idx->node_np_off); put_vec(o, idx->node_mask); put_vec(o, idx->bases); put_vec(o, idx->edges);
    put_vec(o, idx->np_path); put_vec(o, idx->np_pos); put_vec(o, idx->path_len); put_vec(o, idx->path_name_off);
    std::vector<char> names(idx->path_names.begin(), idx->path_names.end());
    put_vec(o, names);
    put_vec(o, idx->win_graph); put_vec(o, idx->win_node); put_vec(o, idx->win_offset); put_vec(o, idx->win_merge_span);
    put_vec(o, idx->win_cn_off); put_vec(o, idx->cn_node); put_vec(o, idx->cn_count); put_vec(o, idx->win_ref_off);
    put_vec(o, idx->win_ref); put_vec(o, idx->win_sketch);
    if (!o) return set_error(GROOT_E_IO, "write to %s failed", path);
    return GROOT_OK;
}

int groot_index_load(const char *path, groot_index **out)
{
    if (!path || !out) return set_error(GROOT_E_INVALID, "null argument");
    std::ifstream i(path, std::ios::binary);
    if (!i) return set_error(GROOT_E_IO, "cannot open %s", path);
    char magic[8];
    uint32_t hdr[8];
    if (!i.read(magic, 8) || memcmp(magic, GIDX_MAGIC, 8) != 0 || !i.read((char *)hdr, sizeof hdr))
        return set_error(GROOT_E_FORMAT, "%s is not a groot-hip index", path);
    auto idx = new groot_index();
    idx->v.kmer_size = hdr[0]; idx->v.sketch_size = hdr[1]; idx->v.window_size = hdr[2]; idx->v.num_part = hdr[3];
    idx->v.max_k = hdr[4]; idx->v.num_window_kmers = hdr[5]; idx->v.path_words = hdr[6];
    std::vector<char> names;
    bool ok = get_vec(i, idx->graph_node_off) && get_vec(i, idx->graph_path_off) && get_vec(i, idx->graph_masked) &&
              get_vec(i, idx->node_seg_id) && get_vec(i, idx->node_seq_off) && get_vec(i, idx->node_edge_off) &&
              get_vec(i, idx->node_np_off) && get_vec(i, idx->node_mask) && get_vec(i, idx->bases) && get_vec(i, idx->edges) &&
              get_vec(i, idx->np_path) && get_vec(i, idx->np_pos) && get_vec(i, idx->path_len) && get_vec(i, idx->path_name_off) &&
              get_vec(i, names) && get_vec(i, idx->win_graph) && get_vec(i, idx->win_node) && get_vec(i, idx->win_offset) &&
              get_vec(i, idx->win_merge_span) && get_vec(i, idx->win_cn_off) && get_vec(i, idx->cn_node) &&
              get_vec(i, idx->cn_count) && get_vec(i, idx->win_ref_off) && get_vec(i, idx->win_ref) && get_vec(i, idx->win_sketch);
    if (ok) {
        idx->path_names.assign(names.begin(), names.end());
        const size_t ng = idx->graph_masked.size(), nn = idx->node_seg_id.size(), nw = idx->win_graph.size();
        ok = idx->graph_node_off.size() == ng + 1 && idx->graph_path_off.size() == ng + 1 && idx->node_seq_off.size() == nn + 1 &&
             idx->node_edge_off.size() == nn + 1 && idx->node_np_off.size() == nn + 1 &&
             idx->node_mask.size() == nn * idx->v.path_words && idx->path_name_off.size() == idx->path_len.size() + 1 &&
             idx->win_node.size() == nw && idx->win_offset.size() == nw && idx->win_merge_span.size() == nw &&
             idx->win_cn_off.size() == nw + 1 && idx->win_ref_off.size() == nw + 1 &&
             idx->win_sketch.size() == nw * idx->v.sketch_size && idx->cn_count.size() == idx->cn_node.size() &&
             idx->np_pos.size() == idx->np_path.size();
    }
    if (!ok) {
        delete idx;
        return set_error(GROOT_E_FORMAT, "%s is truncated or corrupt", path);
    }
    idx->bind();
    // array lengths agree; now the contents: every index in range, offsets monotone and ending at their payload
    const std::string why = groot::check_index_view(&idx->v);
    if (!why.empty()) {
        delete idx;
        return set_error(GROOT_E_FORMAT, "%s is corrupt: %s", path, why.c_str());
    }
    *out = idx;
    return GROOT_OK;
}

int groot_index_view_check(const groot_index_view *view)
{
    const std::string why = groot::check_index_view(view);
    if (!why.empty()) return set_error(GROOT_E_FORMAT, "inconsistent index view: %s", why.c_str());
    return GROOT_OK;
}

} // extern "C"

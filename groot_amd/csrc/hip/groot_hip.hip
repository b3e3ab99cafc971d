// groot_hip.hip -- libgroot_hip.so: ctx management + the C ABI of include/groot_hip.h.
// gfx950 only; no CPU fallback anywhere in this library.
#include <cstring>   // before rocprim: its texture iterator calls host memset

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "kernels.hpp"

using namespace groot;

// ---------------------------------------------------------------------------------------------
// ctx
// ---------------------------------------------------------------------------------------------
template <class T> struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    hipError_t alloc(size_t count)
    {
        release();
        n = count;
        return hipMalloc((void **)&p, std::max<size_t>(count, 1) * sizeof(T));
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    ~DevBuf() { release(); }
};

struct groot_ctx {
    int device = 0;
    std::string err;
    groot_params prm{};
    uint32_t s = 0, k = 0, max_k = 0, l_max = 0, pw_view = 0, pw = 0, n_windows = 0, max_q = 0, band_hash_bits = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    hipEvent_t ev[6]{};
    bool profiling = false;
    groot_stage_ms ms{};

    // index in HBM
    DevBuf<uint32_t> win_prefix, edges, win_graph, cn_node,
        band_keys, band_ids;
    DevBuf<ExactEntry> band_hash;
    DevBuf<uint8_t> band_sig;
    DevBuf<uint32_t> band_run;
    DevBuf<uint8_t> bases, q_k, q_l;
    DevBuf<uint16_t> q_min_eq;
    DevBuf<uint64_t> win_sketch;
    DevBuf<unsigned char> node_rec;
    DevBuf<WinRec> win_rec;
    DevBuf<ExactEntry> exact;
    DeviceIndex dix{};

    // batch state
    DevBuf<uint8_t> seq, exc_byte;
    DevBuf<uint32_t> packed;
    DevBuf<uint64_t> exc_pos;
    DevBuf<uint64_t> seq_off;
    const uint8_t *cur_seq = nullptr;
    const uint64_t *cur_off = nullptr;
    uint32_t n_reads = 0, first_read_id = 0, batch_max_len = 0;
    bool submitted = false, finished = false;
    uint32_t seed_slots = 0;
    DevBuf<uint32_t> seed_count, seed_win, sort_key, sort_key_out, perm_in, perm;
    DevBuf<char> sort_tmp;
    DevBuf<ReadRec> read_rec, read_rec_sorted;
    DevBuf<uint64_t> sketches;
    DevBuf<DeviceCounters> ctr;
    DeviceCounters hctr{};
    // traversal output
    uint32_t trav_cap = 0, ovf_cap = 0;
    DevBuf<groot_trav> trav_first, ovf_trav, trav_sorted;
    DevBuf<uint64_t> mask_first, ovf_mask, trav_mask_sorted;
    DevBuf<uint32_t> trav_cnt, trav_off, ovf_cnt;
    DevBuf<char> scan_tmp;
    uint32_t n_trav = 0;
    // DFS stacks
    uint32_t align_threads = 0, stk_depth = 0;
    DevBuf<uint64_t> stk_hdr, stk_mask;
    // weights
    DevBuf<uint32_t> attempts;
    uint32_t *attempts_ptr = nullptr;   // own buffer or a caller-bound one
};

static thread_local std::string g_open_err;

static int fail(groot_ctx *ctx, int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    else g_open_err = buf;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                             \
    do {                                                                                                \
        hipError_t e__ = (expr);                                                                        \
        if (e__ != hipSuccess) return fail(ctx, GROOT_E_DEVICE, "%s: %s", #expr, hipGetErrorString(e__)); \
    } while (0)

template <class T> static hipError_t upload(DevBuf<T> &d, const T *src, size_t n, size_t pad = 0)
{
    hipError_t e = d.alloc(n + pad);
    if (e != hipSuccess) return e;
    if (pad) {
        e = hipMemset(d.p, 0, (n + pad) * sizeof(T));
        if (e != hipSuccess) return e;
    }
    if (n) e = hipMemcpy(d.p, src, n * sizeof(T), hipMemcpyHostToDevice);
    return e;
}

// ---------------------------------------------------------------------------------------------
// LSH Ensemble parameters (github.com/ekzhu/lshensemble v1.1.0: OptimalKL, Containment), computed
// once per possible kmerCount at open -- the reference caches them per (x, q, t) at query time.
// ---------------------------------------------------------------------------------------------
namespace {

struct KLProb {
    int x, q, l, k;
    double p(double t) const { return 1.0 - std::pow(1.0 - std::pow(t / (1.0 + double(x) / double(q) - t), double(k)), double(l)); }
};

template <class F> double integrate(F f, double a, double b, double precision)
{
    double area = 0.0;
    for (double x = a; x < b; x += precision) area += f(x + 0.5 * precision) * precision;
    return area;
}

void optimal_kl(int max_k, int max_l, int x, int q, double t, int &opt_k, int &opt_l)
{
    const double prec = 0.01;
    double min_err = 1.7976931348623157e308;
    opt_k = 0; opt_l = 0;
    const double xq = double(x) / double(q);
    for (int l = 1; l <= max_l; l++)
        for (int k = 1; k <= max_k; k++) {
            KLProb pr{x, q, l, k};
            double fp = 0.0, fn = 0.0;
            if (xq >= 1.0) {
                fp = integrate([&](double v) { return pr.p(v); }, 0.0, t, prec);
                fn = integrate([&](double v) { return 1.0 - pr.p(v); }, t, 1.0, prec);
            } else if (xq >= t) {
                fp = integrate([&](double v) { return pr.p(v); }, 0.0, t, prec);
                fn = integrate([&](double v) { return 1.0 - pr.p(v); }, t, xq, prec);
            }
            const double err = fn + fp;
            if (min_err > err) { min_err = err; opt_k = k; opt_l = l; }
        }
}

// smallest eq in [1, s] with Containment(eq) > t (monotone in eq); s+1 if none
uint32_t min_equal_slots(uint32_t s, int q_size, int x_size, double t)
{
    if (q_size == 0 || x_size == 0) return s + 1;
    for (uint32_t eq = 1; eq <= s; eq++) {
        const double jaccard = double(eq) / double(s);
        const double c = (double(x_size) / double(q_size) + 1.0) * jaccard / (1.0 + jaccard);
        if (c > t) return eq;
    }
    return s + 1;
}

uint32_t round_pw(uint32_t pw)
{
    for (uint32_t c : {3u, 11u})   // NodeRec<3> = 64 B, NodeRec<11> = 128 B
        if (pw <= c) return c;
    return 0;
}

// Per window: every 5-base prefix a read must start with for AlignRead's level 1 (seed node, offsets
// OffSet..OffSet+MergeSpan+WindowSize inside the node, alignment.go:34-45) or level 2 (ContainedNodes,
// offsets 0..10, :47-70) to have any chance: the spellings of 5 bases from each such start position,
// following every OutEdge at node ends ('N' spells anything; a sink before 5 bases accepts anything, as
// dfsRecursive reports a traversal that runs off the graph, :229).  Sound: never clears a spellable prefix.
// Which read prefixes can AlignRead's levels 1-2 start on?  Two tables per window: 6-mer codes (2 bits per base,
// A=0 C=1 T=2 G=3) of oriented read bases [0,6) and [6,12) that some level-1 / level-2 start position of the window
// (alignment.go:34-70) can spell -- following every out-edge, with the graph's 'N' and the graph ends (a read may
// hang off a sink, alignment.go:229-236) as wildcards.  Sound filters: a read whose code is absent from either table
// cannot pass performAlignment from any of those starts.
struct PrefixTables {
    static constexpr int K = 6, T = 2;
    const groot_index_view *v;
    // per table and graph position: the codes spelled from there, as (code | have << 12); have < K = the walk fell off
    // a sink after `have` coded bases and every completion counts
    std::vector<uint32_t> start[T];      // [n_bases + 1]
    std::vector<uint16_t> ent[T];

    struct Walker {
        const groot_index_view *v;
        int d0;                          // first coded depth of this table
        std::vector<uint16_t> *out;
        void walk(uint32_t node, uint32_t off, int depth, int code)
        {
            const uint32_t s0 = v->node_seq_off[node], len = v->node_seq_off[node + 1] - s0;
            while (off < len && depth < d0 + K) {
                if (depth >= d0) {
                    const uint8_t b = v->bases[s0 + off];
                    if (b == 'N') {
                        for (int c = 0; c < 4; c++) walk(node, off + 1, depth + 1, code | (c << (2 * (depth - d0))));
                        return;
                    }
                    code |= (int)((b >> 1) & 3) << (2 * (depth - d0));
                }
                depth++; off++;
            }
            if (depth == d0 + K) { out->push_back((uint16_t)(code | (K << 12))); return; }
            const uint32_t e0 = v->node_edge_off[node], e1 = v->node_edge_off[node + 1];
            if (e0 == e1) { out->push_back((uint16_t)(code | (std::max(0, depth - d0) << 12))); return; }
            for (uint32_t e = e0; e < e1; e++) walk(v->edges[e], 0, depth, code);
        }
    };

    void positions(unsigned nt)
    {
        // pass 1 per node (threads take nodes round-robin), then stitched into one CSR per table
        std::vector<std::vector<uint16_t>> per_node[T];
        std::vector<std::vector<uint32_t>> per_node_cnt[T];
        for (int t = 0; t < T; t++) { per_node[t].resize(v->n_nodes); per_node_cnt[t].resize(v->n_nodes); }
        std::vector<std::thread> th;
        for (unsigned x = 0; x < nt; x++)
            th.emplace_back([&, x]() {
                for (uint32_t n = x; n < v->n_nodes; n += nt) {
                    const uint32_t len = v->node_seq_off[n + 1] - v->node_seq_off[n];
                    for (int t = 0; t < T; t++) {
                        Walker wk{v, t * K, &per_node[t][n]};
                        per_node_cnt[t][n].resize(len);
                        for (uint32_t o = 0; o < len; o++) {
                            const size_t before = per_node[t][n].size();
                            wk.walk(n, o, 0, 0);
                            auto &e = per_node[t][n];
                            std::sort(e.begin() + before, e.end());
                            e.erase(std::unique(e.begin() + before, e.end()), e.end());
                            per_node_cnt[t][n][o] = (uint32_t)(e.size() - before);
                        }
                    }
                }
            });
        for (auto &x : th) x.join();
        for (int t = 0; t < T; t++) {
            start[t].assign(v->n_bases + 1, 0);
            size_t total = 0;
            for (uint32_t n = 0; n < v->n_nodes; n++) total += per_node[t][n].size();
            ent[t].reserve(total);
            for (uint32_t n = 0; n < v->n_nodes; n++) {      // node order = order of `bases`
                const uint32_t s0 = v->node_seq_off[n];
                uint32_t run = (uint32_t)ent[t].size();
                for (size_t o = 0; o < per_node_cnt[t][n].size(); o++) { start[t][s0 + o] = run; run += per_node_cnt[t][n][o]; }
                ent[t].insert(ent[t].end(), per_node[t][n].begin(), per_node[t][n].end());
            }
            start[t][v->n_bases] = (uint32_t)ent[t].size();
        }
    }
    // nodes are stored back to back in `bases` and the CSR was filled in that order: p's entries end where p+1's begin
    void add(int t, size_t pos, uint32_t *bits) const
    {
        for (uint32_t i = start[t][pos]; i < start[t][pos + 1]; i++) {
            const int code = ent[t][i] & 0xFFF, have = ent[t][i] >> 12;
            if (have >= K) { bits[code >> 5] |= 1u << (code & 31); continue; }
            const int free_bits = 2 * (K - have);
            for (int x = 0; x < (1 << free_bits); x++) {
                const int c = (code & ((1 << (2 * have)) - 1)) | (x << (2 * have));
                bits[c >> 5] |= 1u << (c & 31);
            }
        }
    }
    void window(uint32_t w, uint32_t *out) const     // out: T * 128 words
    {
        const uint32_t seed = v->win_node[w], off0 = v->win_offset[w];
        const uint32_t seed_len = v->node_seq_off[seed + 1] - v->node_seq_off[seed];
        const uint64_t last = (uint64_t)off0 + v->win_merge_span[w] + v->window_size;
        const uint32_t hi = (uint32_t)std::min<uint64_t>(seed_len, last + 1);
        for (int t = 0; t < T; t++) {
            uint32_t *bits = out + t * 128;
            for (uint32_t o = off0; o < hi; o++) add(t, (size_t)v->node_seq_off[seed] + o, bits);
            for (uint32_t c = v->win_cn_off[w]; c < v->win_cn_off[w + 1]; c++) {
                const uint32_t n = v->cn_node[c];
                const uint32_t nlen = v->node_seq_off[n + 1] - v->node_seq_off[n];
                for (uint32_t o = 0; o < std::min(nlen, 11u); o++) add(t, (size_t)v->node_seq_off[n] + o, bits);
            }
        }
    }
};

template <int PW> void build_node_records(const groot_index_view *v, std::vector<unsigned char> &out)
{
    std::vector<NodeRec<PW>> recs(v->n_nodes);
    for (uint32_t n = 0; n < v->n_nodes; n++) {
        NodeRec<PW> &r = recs[n];
        memset(&r, 0, sizeof r);
        r.seq_off = v->node_seq_off[n];
        r.seq_len = v->node_seq_off[n + 1] - v->node_seq_off[n];
        const uint32_t e0 = v->node_edge_off[n], deg = v->node_edge_off[n + 1] - e0;
        r.deg = deg;
        if (deg <= 4) {
            for (uint32_t e = 0; e < deg; e++) {
                const uint32_t c = v->edges[e0 + e];
                r.edges[e] = c;
                r.child_first[e] = v->bases[v->node_seq_off[c]];
            }
        } else {
            r.edges[0] = e0;
        }
        for (uint32_t i = 0; i < 8 && i < r.seq_len; i++) r.first8 |= (uint64_t)v->bases[r.seq_off + i] << (8 * i);
        for (uint32_t w = 0; w < v->path_words; w++) r.mask[w] = v->node_mask[(size_t)n * v->path_words + w];
    }
    out.resize(recs.size() * sizeof(NodeRec<PW>));
    if (!recs.empty()) memcpy(out.data(), recs.data(), out.size());
}

} // namespace

// ---------------------------------------------------------------------------------------------
// kernel dispatch by (sketch size, maxK) and path words
// ---------------------------------------------------------------------------------------------
template <int S, int MAXK, int M5> static void launch_seed_sm(const SeedArgs &a, bool dump, dim3 grid, size_t lds, hipStream_t st)
{
    if (dump) hipLaunchKernelGGL((sketch_seed_kernel<S, MAXK, true, M5>), grid, dim3(kBlock), lds, st, a);
    else hipLaunchKernelGGL((sketch_seed_kernel<S, MAXK, false, M5>), grid, dim3(kBlock), lds, st, a);
}

static bool seed_supported(uint32_t s, uint32_t max_k)
{
    if (max_k != 4) return false;
    switch (s) {
    case 8: case 10: case 12: case 16: case 20: case 21: case 24: case 28: case 30: case 32: case 36: case 40: case 42: case 48: case 50:
    case 56: case 64: return true;
    default: return false;
    }
}

static void launch_seed(uint32_t s, const SeedArgs &a, bool dump, dim3 grid, size_t lds, hipStream_t st)
{
    // low 5 bits of k * multiSeed: kernels specialised on it replace the per-slot 64-bit multiplies by adds
    const uint32_t m5 = (uint32_t)(((uint64_t)a.ix.k * GROOT_MULTI_SEED) & 31u);
    if (s == 21) {   // `groot index` default sketch size, for the common k-mer sizes
        if (m5 == 6) return launch_seed_sm<21, 4, 6>(a, dump, grid, lds, st);     // k = 31 (default), 63
        if (m5 == 10) return launch_seed_sm<21, 4, 10>(a, dump, grid, lds, st);   // k = 41
        if (m5 == 14) return launch_seed_sm<21, 4, 14>(a, dump, grid, lds, st);   // k = 51
        if (m5 == 2) return launch_seed_sm<21, 4, 2>(a, dump, grid, lds, st);     // k = 21
    }
    if (s == 20 && m5 == 6) return launch_seed_sm<20, 4, 6>(a, dump, grid, lds, st);    // travis e2e: -k 31 -s 20
    if (s == 30 && m5 == 14) return launch_seed_sm<30, 4, 14>(a, dump, grid, lds, st);  // pipeline tests: k = 51, s = 30
    switch (s) {
    case 8: launch_seed_sm<8, 4, -1>(a, dump, grid, lds, st); break;
    case 10: launch_seed_sm<10, 4, -1>(a, dump, grid, lds, st); break;
    case 12: launch_seed_sm<12, 4, -1>(a, dump, grid, lds, st); break;
    case 16: launch_seed_sm<16, 4, -1>(a, dump, grid, lds, st); break;
    case 20: launch_seed_sm<20, 4, -1>(a, dump, grid, lds, st); break;
    case 21: launch_seed_sm<21, 4, -1>(a, dump, grid, lds, st); break;
    case 24: launch_seed_sm<24, 4, -1>(a, dump, grid, lds, st); break;
    case 28: launch_seed_sm<28, 4, -1>(a, dump, grid, lds, st); break;
    case 30: launch_seed_sm<30, 4, -1>(a, dump, grid, lds, st); break;
    case 32: launch_seed_sm<32, 4, -1>(a, dump, grid, lds, st); break;
    case 36: launch_seed_sm<36, 4, -1>(a, dump, grid, lds, st); break;
    case 40: launch_seed_sm<40, 4, -1>(a, dump, grid, lds, st); break;
    case 42: launch_seed_sm<42, 4, -1>(a, dump, grid, lds, st); break;
    case 48: launch_seed_sm<48, 4, -1>(a, dump, grid, lds, st); break;
    case 50: launch_seed_sm<50, 4, -1>(a, dump, grid, lds, st); break;
    case 56: launch_seed_sm<56, 4, -1>(a, dump, grid, lds, st); break;
    case 64: launch_seed_sm<64, 4, -1>(a, dump, grid, lds, st); break;
    default: break;
    }
}

static void launch_align(uint32_t pw, const AlignArgs &a, dim3 grid, hipStream_t st)
{
    const size_t lds = a.lds_stride_dw ? (size_t)kBlock * a.lds_stride_dw * 4 + 16 : 0;
    if (pw == 3) {
        if (lds) hipLaunchKernelGGL((align_kernel<3, true>), grid, dim3(kBlock), lds, st, a);
        else hipLaunchKernelGGL((align_kernel<3, false>), grid, dim3(kBlock), 0, st, a);
    } else if (pw == 11) {
        if (lds) hipLaunchKernelGGL((align_kernel<11, true>), grid, dim3(kBlock), lds, st, a);
        else hipLaunchKernelGGL((align_kernel<11, false>), grid, dim3(kBlock), 0, st, a);
    }
}

// ---------------------------------------------------------------------------------------------
// batch execution
// ---------------------------------------------------------------------------------------------
static constexpr uint32_t kMaxLdsReadBytes = 64 * 1024;

static int alloc_seed_slots(groot_ctx *c, uint32_t slots)
{
    c->seed_slots = slots;
    HIP_TRY(c, c->seed_win.alloc((size_t)slots * c->prm.max_batch_reads));
    return GROOT_OK;
}

static int alloc_trav(groot_ctx *c, uint32_t cap)
{
    c->trav_cap = cap;
    HIP_TRY(c, c->trav_sorted.alloc(cap));
    HIP_TRY(c, c->trav_mask_sorted.alloc((size_t)cap * c->pw_view));
    return GROOT_OK;
}

static int alloc_ovf(groot_ctx *c, uint32_t cap_per_shard)
{
    c->ovf_cap = cap_per_shard;
    HIP_TRY(c, c->ovf_trav.alloc((size_t)kOvfShards * cap_per_shard));
    HIP_TRY(c, c->ovf_mask.alloc((size_t)kOvfShards * cap_per_shard * c->pw));
    return GROOT_OK;
}

#ifndef GROOT_SPAN_BITS
#define GROOT_SPAN_BITS 6
#endif
static int launch_seed_stage(groot_ctx *c)
{
    SeedArgs a{};
    a.ix = c->dix;
    a.seq = c->cur_seq;
    a.seq_off = c->cur_off;
    a.n_reads = c->n_reads;
    a.max_read_len = c->prm.max_read_len;
    const uint64_t want = (uint64_t)kBlock * c->batch_max_len + 32;
    a.lds_read_bytes = (uint32_t)std::min<uint64_t>(want, kMaxLdsReadBytes);
    a.seed_slots = c->seed_slots;
    a.seed_count = c->seed_count.p;
    a.seed_win = c->seed_win.p;
    a.sketch_out = c->prm.keep_sketches ? c->sketches.p : nullptr;
    a.sort_key = c->sort_key.p;
    a.read_rec = c->read_rec.p;
    a.ctr = c->ctr.p;
    // processing order of the align stage: reads sorted by (node span of the first seed window, that window, likely
    // orientation).  key = span << (32-span_bits) | window << 2 | class; reads without seeds carry 0xFFFFFFFF and sort last
    unsigned win_bits = 3;                                  // 2 class bits + one bit above the largest window id
    for (uint32_t v = c->n_windows; v; v >>= 1) win_bits++;
    win_bits = std::min(32u, win_bits);
    a.sort_span_bits = std::min((unsigned)GROOT_SPAN_BITS, 32u - win_bits);
    const unsigned end_bit = a.sort_span_bits ? 32u : win_bits;
    const dim3 grid((c->n_reads + kBlock - 1) / kBlock);
    const size_t lds = kLdsReads + ((a.lds_read_bytes + 15) & ~15u);
    launch_seed(c->s, a, c->prm.keep_sketches != 0, grid, lds, c->stream);
    HIP_TRY(c, hipGetLastError());
    if (c->profiling) HIP_TRY(c, hipEventRecord(c->ev[5], c->stream));
    size_t tmp_bytes = 0;
    // keys are (window << 2 | class) below 2^end_bit, or 0xFFFFFFFF for reads without seeds: sorting the low
    // end_bit bits keeps those last as long as bit end_bit-1.. are all ones for them, which they are
    HIP_TRY(c, rocprim::radix_sort_pairs(nullptr, tmp_bytes, c->sort_key.p, c->sort_key_out.p, c->perm_in.p, c->perm.p, c->n_reads, 0,
                                         end_bit, c->stream));
    if (tmp_bytes > c->sort_tmp.n) HIP_TRY(c, c->sort_tmp.alloc(tmp_bytes));
    HIP_TRY(c, rocprim::radix_sort_pairs(c->sort_tmp.p, tmp_bytes, c->sort_key.p, c->sort_key_out.p, c->perm_in.p, c->perm.p,
                                         c->n_reads, 0, end_bit, c->stream));
    hipLaunchKernelGGL(gather_recs_kernel, grid, dim3(kBlock), 0, c->stream, c->perm.p, c->read_rec.p, c->read_rec_sorted.p, c->n_reads);
    HIP_TRY(c, hipGetLastError());
    return GROOT_OK;
}

static int launch_align_stage(groot_ctx *c, bool update_weights)
{
    AlignArgs a{};
    a.ix = c->dix;
    a.seq = c->cur_seq;
    a.seq_off = c->cur_off;
    a.n_reads = c->n_reads;
    a.first_read_id = c->first_read_id;
    a.seed_slots = c->seed_slots;
    a.seed_count = c->seed_count.p;
    a.seed_win = c->seed_win.p;
    a.perm = c->perm.p;
    a.read_rec = c->read_rec_sorted.p;   // in perm order (gather_recs_kernel)
    a.no_align = c->prm.no_exact_align;
    a.update_weights = update_weights ? 1 : 0;
    a.attempts = c->attempts_ptr;
    a.node_rec = c->node_rec.p;
    a.trav_first = c->trav_first.p;
    a.mask_first = c->mask_first.p;
    a.trav_cnt = c->trav_cnt.p;
    a.ovf_trav = c->ovf_trav.p;
    a.ovf_mask = c->ovf_mask.p;
    a.ovf_cnt = c->ovf_cnt.p;
    a.ovf_cap = c->ovf_cap;
    a.stk_hdr = c->stk_hdr.p;
    a.stk_mask = c->stk_mask.p;
    const uint32_t blocks = std::min<uint32_t>((c->n_reads + kBlock - 1) / kBlock, c->align_threads / kBlock);
    a.n_threads = blocks * kBlock;
    a.stk_depth = c->stk_depth;
    // stage reads in LDS when 256 lanes x (longest read + slack) stays within 64 KB (<= 2 workgroups... per CU budget)
    {
        // 8 zero bytes, then the read in whole 16-byte pieces up to 12 bytes past its end; odd dword stride = no bank conflicts
        const uint32_t stride = (2 + 4 * ((c->batch_max_len + 27) / 16)) | 1u;
        a.lds_stride_dw = (size_t)kBlock * stride * 4 <= 64 * 1024 ? stride : 0;
    }
    a.ctr = c->ctr.p;
    HIP_TRY(c, hipMemsetAsync(c->ovf_cnt.p, 0, (kOvfShards + 2) * sizeof(uint32_t), c->stream));   // + the two chunk cursors
    launch_align(c->pw, a, dim3(blocks), c->stream);
    HIP_TRY(c, hipGetLastError());
    return GROOT_OK;
}

// traversal records -> (read, ord) order: exclusive scan of the per-read counts, then two scatters
static int launch_order_stage(groot_ctx *c)
{
    const uint32_t n = c->n_reads;
    size_t tmp_bytes = 0;
    HIP_TRY(c, rocprim::exclusive_scan(nullptr, tmp_bytes, c->trav_cnt.p, c->trav_off.p, 0u, n, rocprim::plus<uint32_t>(), c->stream));
    if (tmp_bytes > c->scan_tmp.n) HIP_TRY(c, c->scan_tmp.alloc(tmp_bytes));
    HIP_TRY(c, rocprim::exclusive_scan(c->scan_tmp.p, tmp_bytes, c->trav_cnt.p, c->trav_off.p, 0u, n, rocprim::plus<uint32_t>(), c->stream));
    hipLaunchKernelGGL(order_total_kernel, dim3(1), dim3(1), 0, c->stream, c->trav_off.p, c->trav_cnt.p, n, c->ctr.p);
    hipLaunchKernelGGL(order_first_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream, c->trav_first.p,
                       c->mask_first.p, c->trav_off.p, c->trav_cnt.p, n, c->trav_sorted.p, c->trav_mask_sorted.p, c->trav_cap, c->pw,
                       c->pw_view, c->ctr.p);
    hipLaunchKernelGGL(order_ovf_kernel, dim3((c->ovf_cap + kBlock - 1) / kBlock, kOvfShards), dim3(kBlock), 0, c->stream,
                       c->ovf_trav.p, c->ovf_mask.p, c->ovf_cnt.p, c->ovf_cap, c->trav_off.p, c->first_read_id, c->trav_sorted.p,
                       c->trav_mask_sorted.p, c->trav_cap, c->pw, c->pw_view, c->ctr.p);
    HIP_TRY(c, hipGetLastError());
    return GROOT_OK;
}

static int run_batch_async(groot_ctx *c)
{
    HIP_TRY(c, hipMemsetAsync(c->ctr.p, 0, sizeof(DeviceCounters), c->stream));
    if (c->profiling) HIP_TRY(c, hipEventRecord(c->ev[1], c->stream));
    if (int rc = launch_seed_stage(c)) return rc;
    if (c->profiling) HIP_TRY(c, hipEventRecord(c->ev[2], c->stream));
    if (int rc = launch_align_stage(c, true)) return rc;
    if (c->profiling) HIP_TRY(c, hipEventRecord(c->ev[3], c->stream));
    if (int rc = launch_order_stage(c)) return rc;
    if (c->profiling) HIP_TRY(c, hipEventRecord(c->ev[4], c->stream));
    return GROOT_OK;
}

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" {

void groot_params_default(groot_params *p)
{
    if (!p) return;
    memset(p, 0, sizeof *p);
    p->containment_threshold = 0.99;   // cmd/align.go:47
    p->max_read_len = 256;
    p->max_batch_reads = 1u << 20;
    p->max_seeds_per_read = 8;
}

int groot_hip_device_count(int *n)
{
    if (!n) return GROOT_E_INVALID;
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        *n = 0;
        return fail(nullptr, GROOT_E_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *n = c;
    return GROOT_OK;
}

const char *groot_hip_last_error(const groot_ctx *ctx) { return ctx ? ctx->err.c_str() : g_open_err.c_str(); }

void groot_hip_close(groot_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (auto &e : ctx->ev)
        if (e) (void)hipEventDestroy(e);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

static int open_impl(groot_ctx *c, int device_id, const groot_index_view *v, const groot_params *p)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(c, GROOT_E_DEVICE, "no HIP device available (libgroot_hip has no CPU fallback)");
    if (device_id < 0 || device_id >= ndev) return fail(c, GROOT_E_INVALID, "device %d out of range (%d devices)", device_id, ndev);
    if (!v) return fail(c, GROOT_E_INVALID, "null index view");
    c->device = device_id;
    HIP_TRY(c, hipSetDevice(device_id));
    groot_params d;
    groot_params_default(&d);
    c->prm = p ? *p : d;
    if (!c->prm.max_read_len) c->prm.max_read_len = d.max_read_len;
    if (!c->prm.max_batch_reads) c->prm.max_batch_reads = d.max_batch_reads;
    if (!c->prm.max_seeds_per_read) c->prm.max_seeds_per_read = d.max_seeds_per_read;
    if (!c->prm.max_batch_bases) c->prm.max_batch_bases = (uint64_t)c->prm.max_batch_reads * c->prm.max_read_len;
    if (c->prm.max_read_len > 65535) return fail(c, GROOT_E_UNSUPPORTED, "max_read_len must be <= 65535");
    if (v->kmer_size == 0 || v->kmer_size > 64) return fail(c, GROOT_E_UNSUPPORTED, "k-mer size %u not in [1,64]", v->kmer_size);
    if (c->prm.max_read_len < v->kmer_size) return fail(c, GROOT_E_INVALID, "max_read_len smaller than the k-mer size");
    if (!seed_supported(v->sketch_size, v->max_k))
        return fail(c, GROOT_E_UNSUPPORTED, "sketch size %u with maxK %u has no compiled kernel (see launch_seed)", v->sketch_size, v->max_k);
    c->s = v->sketch_size; c->k = v->kmer_size; c->max_k = v->max_k; c->l_max = v->sketch_size / v->max_k;
    c->pw_view = v->path_words; c->pw = round_pw(v->path_words);
    if (!c->pw) return fail(c, GROOT_E_UNSUPPORTED, "graphs with more than 704 paths are not supported (path_words=%u)", v->path_words);
    c->n_windows = v->n_windows;
    c->max_q = c->prm.max_read_len - c->k + 1;

    HIP_TRY(c, hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    for (auto &e : c->ev) HIP_TRY(c, hipEventCreate(&e));

    // ---- graphs + windows -> HBM ----
    HIP_TRY(c, upload(c->edges, v->edges, v->n_edges));
    HIP_TRY(c, upload(c->bases, v->bases, v->n_bases, 64));   // kernels read 8-byte windows up to 24 bytes past a node start
    {
        std::vector<unsigned char> recs;
        if (c->pw == 3) build_node_records<3>(v, recs);
        else build_node_records<11>(v, recs);
        HIP_TRY(c, upload(c->node_rec, recs.data(), recs.size()));
    }
    {
        std::vector<uint32_t> k5((size_t)v->n_windows * kPrefixWords, 0);
        const unsigned nt = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
        PrefixTables pt;
        pt.v = v;
        pt.positions(nt);
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++)
            th.emplace_back([&, t]() {
                for (uint32_t w = t; w < v->n_windows; w += nt) pt.window(w, k5.data() + (size_t)w * kPrefixWords);
            });
        for (auto &x : th) x.join();
        HIP_TRY(c, upload(c->win_prefix, k5.data(), k5.size()));
    }
    HIP_TRY(c, upload(c->win_graph, v->win_graph, v->n_windows));
    {
        std::vector<WinRec> wr(v->n_windows);
        for (uint32_t w = 0; w < v->n_windows; w++) {
            const uint32_t node = v->win_node[w];
            const uint32_t nlen = v->node_seq_off[node + 1] - v->node_seq_off[node];
            const uint64_t last = (uint64_t)v->win_offset[w] + v->win_merge_span[w] + v->window_size;
            wr[w] = WinRec{v->win_graph[w], node, v->win_offset[w], (uint32_t)std::min<uint64_t>(nlen, last + 1), v->win_cn_off[w],
                           v->win_cn_off[w + 1], v->node_seq_off[node], nlen};
        }
        HIP_TRY(c, upload(c->win_rec, wr.data(), wr.size()));
    }
    HIP_TRY(c, upload(c->cn_node, v->cn_node, v->n_cn));
    HIP_TRY(c, upload(c->win_sketch, v->win_sketch, (size_t)v->n_windows * v->sketch_size, 2));

    // ---- lookup structures (the reference bootstraps its LSH forests at load too, lshe.go:95-147) ----
    const uint32_t n = v->n_windows, s = v->sketch_size;
    {   // exact-match table
        uint32_t cap = 16;
        while (cap < 2 * (uint64_t)n) cap <<= 1;
        std::vector<ExactEntry> tab(cap, ExactEntry{0, kEmpty});
        for (uint32_t w = 0; w < n; w++) {
            uint64_t h = GROOT_SKETCH_HASH_INIT;
            for (uint32_t i = 0; i < s; i++) h = sketch_hash_step(h, v->win_sketch[(size_t)w * s + i]);
            uint32_t slot = (uint32_t)h & (cap - 1);
            while (tab[slot].id != kEmpty) slot = (slot + 1) & (cap - 1);
            tab[slot] = ExactEntry{(uint32_t)(h >> 32), w};
        }
        HIP_TRY(c, upload(c->exact, tab.data(), tab.size()));
        c->dix.exact_mask = cap - 1;
    }
    {   // LSH forest band tables: per band the low-32 hash values of its max_k slots, sorted
        const uint32_t mk = v->max_k, lmax = c->l_max;
        std::vector<uint32_t> keys((size_t)lmax * n * mk), ids((size_t)lmax * n), order(n);
        for (uint32_t b = 0; b < lmax; b++) {
            std::iota(order.begin(), order.end(), 0u);
            const uint64_t *sk = v->win_sketch;
            std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
                for (uint32_t j = 0; j < mk; j++) {
                    const uint32_t a = (uint32_t)sk[(size_t)x * s + b * mk + j], bb = (uint32_t)sk[(size_t)y * s + b * mk + j];
                    if (a != bb) return a < bb;
                }
                return x < y;
            });
            for (uint32_t e = 0; e < n; e++) {
                ids[(size_t)b * n + e] = order[e];
                for (uint32_t j = 0; j < mk; j++)
                    keys[((size_t)b * n + e) * mk + j] = (uint32_t)sk[(size_t)order[e] * s + b * mk + j];
            }
        }
        HIP_TRY(c, upload(c->band_keys, keys.data(), keys.size()));
        HIP_TRY(c, upload(c->band_ids, ids.data(), ids.size()));
        // hash tables over the distinct K-prefixes of every band: the query finds the first matching row with one or two
        // probes instead of a binary search of ~log2(n) dependent loads
        uint32_t bits = 4;
        while ((1ull << bits) < 2 * (uint64_t)n) bits++;
        c->band_hash_bits = bits;
        const uint32_t cap = 1u << bits;
        std::vector<ExactEntry> tab((size_t)lmax * mk * cap, ExactEntry{0, kEmpty});
        for (uint32_t b = 0; b < lmax; b++)
            for (uint32_t K = 1; K <= mk; K++) {
                ExactEntry *t = tab.data() + (((size_t)b * mk + (K - 1)) << bits);
                for (uint32_t e = 0; e < n; e++) {
                    const uint32_t *ke = &keys[((size_t)b * n + e) * mk];
                    if (e && std::equal(ke, ke + K, ke - mk)) continue;      // same prefix as the previous row
                    uint64_t h = GROOT_SKETCH_HASH_INIT;
                    for (uint32_t j = 0; j < K; j++) h = sketch_hash_step(h, ke[j]);
                    uint32_t slot = (uint32_t)h & (cap - 1);
                    while (t[slot].id != kEmpty) slot = (slot + 1) & (cap - 1);
                    t[slot] = ExactEntry{(uint32_t)(h >> 32), e};
                }
            }
        HIP_TRY(c, upload(c->band_hash, tab.data(), tab.size()));
        std::vector<uint8_t> sig((size_t)lmax * n * 32, 0);
        const uint32_t sl = std::min<uint32_t>(s, 32);
        for (uint32_t b = 0; b < lmax; b++)
            for (uint32_t e = 0; e < n; e++) {
                const uint64_t *ws = v->win_sketch + (size_t)ids[(size_t)b * n + e] * s;
                uint8_t *row = &sig[((size_t)b * n + e) * 32];
                for (uint32_t i = 0; i < sl; i++) row[i] = (uint8_t)sig8(ws[i]);
            }
        HIP_TRY(c, upload(c->band_sig, sig.data(), sig.size(), 32));
        std::vector<uint32_t> run((size_t)lmax * mk * n, 0);
        for (uint32_t b = 0; b < lmax; b++)
            for (uint32_t K = 1; K <= mk; K++) {
                uint32_t *rn = run.data() + ((size_t)b * mk + (K - 1)) * n;
                for (uint32_t e = n; e-- > 0;) {             // backwards: length of the run of equal K-prefixes starting at e
                    const uint32_t *ke = &keys[((size_t)b * n + e) * mk];
                    rn[e] = (e + 1 < n && std::equal(ke, ke + K, ke + mk)) ? rn[e + 1] + 1 : 1;
                }
            }
        HIP_TRY(c, upload(c->band_run, run.data(), run.size()));
    }
    {   // per kmerCount: (K, L) of the partitions (all have Upper = NumWindowKmers) and min #equal slots
        std::vector<uint8_t> qk(c->max_q + 1, 0), ql(c->max_q + 1, 0);
        std::vector<uint16_t> qm(c->max_q + 1, (uint16_t)(s + 1));
        for (uint32_t q = 1; q <= c->max_q; q++) {
            int K, L;
            optimal_kl((int)v->max_k, (int)c->l_max, (int)v->num_window_kmers, (int)q, c->prm.containment_threshold, K, L);
            qk[q] = (uint8_t)K; ql[q] = (uint8_t)L;
            qm[q] = (uint16_t)min_equal_slots(s, (int)q, (int)v->num_window_kmers, c->prm.containment_threshold);
        }
        HIP_TRY(c, upload(c->q_k, qk.data(), qk.size()));
        HIP_TRY(c, upload(c->q_l, ql.data(), ql.size()));
        HIP_TRY(c, upload(c->q_min_eq, qm.data(), qm.size()));
    }
    DeviceIndex &x = c->dix;
    x.k = v->kmer_size; x.s = s; x.w = v->window_size; x.num_window_kmers = v->num_window_kmers;
    x.n_windows = n; x.n_nodes = v->n_nodes; x.pw = c->pw;
    x.edges = c->edges.p; x.bases = c->bases.p;
    x.win_prefix = c->win_prefix.p; x.win_graph = c->win_graph.p; x.win_rec = c->win_rec.p; x.cn_node = c->cn_node.p;
    x.win_sketch = c->win_sketch.p; x.exact = c->exact.p; x.band_keys = c->band_keys.p; x.band_ids = c->band_ids.p;
    x.band_hash = c->band_hash.p; x.band_hash_bits = c->band_hash_bits; x.band_sig = c->band_sig.p; x.band_run = c->band_run.p;
    x.max_k = v->max_k; x.l_max = c->l_max; x.q_k = c->q_k.p; x.q_l = c->q_l.p; x.q_min_eq = c->q_min_eq.p; x.max_q = c->max_q;

    // ---- batch buffers ----
    const uint32_t R = c->prm.max_batch_reads;
    HIP_TRY(c, c->seq.alloc(c->prm.max_batch_bases + 64));
    HIP_TRY(c, c->seq_off.alloc((size_t)R + 1));
    HIP_TRY(c, c->seed_count.alloc(R));
    HIP_TRY(c, c->sort_key.alloc(R));
    HIP_TRY(c, c->read_rec.alloc(R));
    HIP_TRY(c, c->read_rec_sorted.alloc(R));
    HIP_TRY(c, c->sort_key_out.alloc(R));
    HIP_TRY(c, c->perm.alloc(R));
    {
        std::vector<uint32_t> iota(R);
        std::iota(iota.begin(), iota.end(), 0u);
        HIP_TRY(c, upload(c->perm_in, iota.data(), iota.size()));
    }
    if (int rc = alloc_seed_slots(c, c->prm.max_seeds_per_read)) return rc;
    if (c->prm.keep_sketches) HIP_TRY(c, c->sketches.alloc((size_t)R * s));
    HIP_TRY(c, c->ctr.alloc(1));
    HIP_TRY(c, c->trav_first.alloc(R));
    HIP_TRY(c, c->mask_first.alloc((size_t)R * c->pw));
    HIP_TRY(c, c->trav_cnt.alloc(R));
    HIP_TRY(c, c->trav_off.alloc(R));
    HIP_TRY(c, c->ovf_cnt.alloc(kOvfShards + 2));
    if (int rc = alloc_trav(c, std::max<uint32_t>(1024, R + R / 4))) return rc;
    if (int rc = alloc_ovf(c, std::max<uint32_t>(256, R / kOvfShards / 4))) return rc;
    // the align kernel is persistent: exactly the workgroups that are resident at once (GROOT_ALIGN_WAVES per SIMD = per CU)
    int n_cu = 256;
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, c->device);
    uint32_t per_cu = GROOT_ALIGN_WAVES;
    if (const char *e = getenv("GROOT_ALIGN_GRID_PER_CU")) per_cu = std::max(1, std::min(16, atoi(e)));   // experiments (tools/overlap_probe.py)
    c->align_threads = std::min<uint32_t>(((R + kBlock - 1) / kBlock) * kBlock, (uint32_t)std::max(n_cu, 1) * per_cu * kBlock);
    c->stk_depth = c->prm.max_read_len;
    HIP_TRY(c, c->stk_hdr.alloc((size_t)c->stk_depth * c->align_threads));
    HIP_TRY(c, c->stk_mask.alloc((size_t)c->stk_depth * c->align_threads * c->pw));
    HIP_TRY(c, c->attempts.alloc((size_t)(c->max_q + 1) * n));
    HIP_TRY(c, hipMemset(c->attempts.p, 0, (size_t)(c->max_q + 1) * n * sizeof(uint32_t)));
    c->attempts_ptr = c->attempts.p;
    HIP_TRY(c, hipDeviceSynchronize());
    return GROOT_OK;
}

int groot_hip_open(groot_ctx **out, int device_id, const groot_index_view *idx, const groot_params *p)
{
    if (!out) return fail(nullptr, GROOT_E_INVALID, "null out pointer");
    *out = nullptr;
    groot_ctx *c = new groot_ctx();
    int rc = open_impl(c, device_id, idx, p);
    if (rc) {
        g_open_err = c->err;
        groot_hip_close(c);
        return rc;
    }
    *out = c;
    return GROOT_OK;
}

int groot_hip_set_stream(groot_ctx *c, void *hip_stream)
{
    if (!c) return GROOT_E_INVALID;
    if (c->submitted && !c->finished) return fail(c, GROOT_E_STATE, "cannot change stream while a batch is in flight");
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return GROOT_OK;
}

int groot_hip_set_profiling(groot_ctx *c, int enable)
{
    if (!c) return GROOT_E_INVALID;
    c->profiling = enable != 0;
    return GROOT_OK;
}

static int begin_batch(groot_ctx *c, uint32_t n_reads, uint32_t first_read_id)
{
    if (c->submitted && !c->finished) return fail(c, GROOT_E_STATE, "previous batch not collected: call groot_hip_wait first");
    if (n_reads > c->prm.max_batch_reads) return fail(c, GROOT_E_NOSPACE, "batch of %u reads exceeds max_batch_reads=%u", n_reads, c->prm.max_batch_reads);
    HIP_TRY(c, hipSetDevice(c->device));
    c->n_reads = n_reads; c->first_read_id = first_read_id;
    c->submitted = true; c->finished = false; c->n_trav = 0;
    memset(&c->hctr, 0, sizeof c->hctr);
    memset(&c->ms, 0, sizeof c->ms);
    return GROOT_OK;
}

// offsets must not decrease; *max_len = the longest read (branch-free pass so that it vectorises: 10 M reads per batch)
static int check_offsets(groot_ctx *c, const uint64_t *seq_off, uint32_t n_reads, uint32_t *max_len)
{
    uint64_t longest = 0, bad = 0;
    for (uint32_t i = 0; i < n_reads; i++) {
        bad |= (uint64_t)(seq_off[i + 1] < seq_off[i]);
        longest = std::max(longest, seq_off[i + 1] - seq_off[i]);
    }
    if (bad) {
        uint32_t i = 0;
        while (seq_off[i + 1] >= seq_off[i]) i++;
        c->submitted = false;
        return fail(c, GROOT_E_INVALID, "seq_off not monotone at read %u", i);
    }
    *max_len = (uint32_t)std::min<uint64_t>(longest, 0xFFFFFFFFu);
    return GROOT_OK;
}

int groot_hip_submit(groot_ctx *c, const uint8_t *seq_concat, const uint64_t *seq_off, uint32_t n_reads, uint32_t first_read_id)
{
    if (!c) return GROOT_E_INVALID;
    if (n_reads && (!seq_concat || !seq_off)) return fail(c, GROOT_E_INVALID, "null read buffers");
    if (int rc = begin_batch(c, n_reads, first_read_id)) { return rc; }
    if (!n_reads) return GROOT_OK;
    const uint64_t total = seq_off[n_reads] - seq_off[0];
    if (seq_off[0] != 0) { c->submitted = false; return fail(c, GROOT_E_INVALID, "seq_off[0] must be 0"); }
    if (total > c->prm.max_batch_bases) { c->submitted = false; return fail(c, GROOT_E_NOSPACE, "batch of %llu bases exceeds max_batch_bases=%llu", (unsigned long long)total, (unsigned long long)c->prm.max_batch_bases); }
    uint32_t max_len = 0;
    if (int rc = check_offsets(c, seq_off, n_reads, &max_len)) return rc;
    c->batch_max_len = std::min(max_len, c->prm.max_read_len);
    if (c->profiling) HIP_TRY(c, hipEventRecord(c->ev[0], c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->seq.p, seq_concat, total, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->seq_off.p, seq_off, ((size_t)n_reads + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
    c->cur_seq = c->seq.p; c->cur_off = c->seq_off.p;
    return run_batch_async(c);
}

int groot_hip_submit_packed(groot_ctx *c, const uint8_t *packed, const uint64_t *seq_off, uint32_t n_reads, uint32_t first_read_id,
                            const uint64_t *exc_pos, const uint8_t *exc_byte, uint64_t n_exc)
{
    if (!c) return GROOT_E_INVALID;
    if (n_reads && (!packed || !seq_off)) return fail(c, GROOT_E_INVALID, "null read buffers");
    if (n_exc && (!exc_pos || !exc_byte)) return fail(c, GROOT_E_INVALID, "null exception list");
    if (int rc = begin_batch(c, n_reads, first_read_id)) return rc;
    if (!n_reads) return GROOT_OK;
    const uint64_t total = seq_off[n_reads];
    if (seq_off[0] != 0) { c->submitted = false; return fail(c, GROOT_E_INVALID, "seq_off[0] must be 0"); }
    if (total > c->prm.max_batch_bases) { c->submitted = false; return fail(c, GROOT_E_NOSPACE, "batch of %llu bases exceeds max_batch_bases=%llu", (unsigned long long)total, (unsigned long long)c->prm.max_batch_bases); }
    uint32_t max_len = 0;
    if (int rc = check_offsets(c, seq_off, n_reads, &max_len)) return rc;
    for (uint64_t i = 0; i < n_exc; i++)
        if (exc_pos[i] >= total) { c->submitted = false; return fail(c, GROOT_E_INVALID, "exception %llu lies outside the batch", (unsigned long long)i); }
    c->batch_max_len = std::min(max_len, c->prm.max_read_len);
    const uint64_t n_words = (total + 15) / 16;                  // 16 bases per packed word
    if (c->packed.n < n_words) HIP_TRY(c, c->packed.alloc((c->prm.max_batch_bases + 15) / 16 + 1));
    if (c->exc_pos.n < n_exc) { HIP_TRY(c, c->exc_pos.alloc(n_exc + n_exc / 4 + 1024)); HIP_TRY(c, c->exc_byte.alloc(n_exc + n_exc / 4 + 1024)); }
    if (c->profiling) HIP_TRY(c, hipEventRecord(c->ev[0], c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->packed.p, packed, (size_t)((total + 3) / 4), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->seq_off.p, seq_off, ((size_t)n_reads + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(unpack_reads_kernel, dim3((unsigned)((n_words + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream, c->packed.p, n_words,
                       reinterpret_cast<uint4 *>(c->seq.p));
    if (n_exc) {
        HIP_TRY(c, hipMemcpyAsync(c->exc_pos.p, exc_pos, n_exc * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(c->exc_byte.p, exc_byte, n_exc, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(patch_reads_kernel, dim3((unsigned)((n_exc + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream, c->exc_pos.p,
                           c->exc_byte.p, n_exc, c->seq.p);
    }
    HIP_TRY(c, hipGetLastError());
    c->cur_seq = c->seq.p; c->cur_off = c->seq_off.p;
    return run_batch_async(c);
}

int groot_hip_submit_device(groot_ctx *c, const void *d_seq, const void *d_seq_off, uint32_t n_reads, uint32_t first_read_id,
                            uint32_t max_len)
{
    if (!c) return GROOT_E_INVALID;
    if (n_reads && (!d_seq || !d_seq_off)) return fail(c, GROOT_E_INVALID, "null device buffers");
    if (((uintptr_t)d_seq & 15) != 0) return fail(c, GROOT_E_INVALID, "d_seq must be 16-byte aligned");
    if (int rc = begin_batch(c, n_reads, first_read_id)) return rc;
    if (!n_reads) return GROOT_OK;
    c->batch_max_len = max_len ? std::min(max_len, c->prm.max_read_len) : c->prm.max_read_len;
    c->cur_seq = (const uint8_t *)d_seq; c->cur_off = (const uint64_t *)d_seq_off;
    if (c->profiling) HIP_TRY(c, hipEventRecord(c->ev[0], c->stream));
    return run_batch_async(c);
}

static void fill_counts(groot_ctx *c, groot_counts *out)
{
    if (!out) return;
    out->received = c->n_reads;           // boss.go:194 receivedReads++ for every read
    out->mapped = c->hctr.mapped;
    out->multimapped = c->hctr.multimapped;
    out->alignments = c->hctr.alignments;
    out->seeds = c->hctr.seeds;
    out->travs = c->n_trav;
    out->revcomp_panics = c->hctr.revcomp_panics;
    out->short_reads = c->hctr.short_reads;
}

int groot_hip_wait(groot_ctx *c, groot_counts *counts)
{
    if (!c) return GROOT_E_INVALID;
    if (!c->submitted) return fail(c, GROOT_E_STATE, "no batch submitted");
    if (c->finished) { fill_counts(c, counts); return GROOT_OK; }
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->n_reads == 0) { c->finished = true; fill_counts(c, counts); return GROOT_OK; }
    auto fetch = [&](DeviceCounters &dst) -> int {
        HIP_TRY(c, hipMemcpyAsync(&dst, c->ctr.p, sizeof(DeviceCounters), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        return GROOT_OK;
    };
    if (int rc = fetch(c->hctr)) return rc;
    for (int attempt = 0;; attempt++) {
        if (attempt > 8) return fail(c, GROOT_E_NOSPACE, "output buffers keep overflowing (flags=0x%x)", c->hctr.flags);
        if (c->hctr.flags & kFlagSeedOverflow) {
            // a read had more seeds than slots; the align stage saw the flag and did nothing.  Grow and redo.
            if (int rc = alloc_seed_slots(c, c->hctr.max_seeds + 4)) return rc;
            if (int rc = run_batch_async(c)) return rc;
            if (int rc = fetch(c->hctr)) return rc;
            continue;
        }
        if (c->hctr.flags & kFlagOvfOverflow) {
            // a shard of the overflow traversal list filled up: enlarge and re-emit traversals only
            // (weights and read counters of the first pass stand)
            const DeviceCounters first = c->hctr;
            if (int rc = alloc_ovf(c, c->ovf_cap * 4)) return rc;
            HIP_TRY(c, hipMemsetAsync(c->ctr.p, 0, sizeof(DeviceCounters), c->stream));
            if (int rc = launch_align_stage(c, false)) return rc;
            if (int rc = launch_order_stage(c)) return rc;
            DeviceCounters second{};
            if (int rc = fetch(second)) return rc;
            c->hctr = first;
            c->hctr.n_trav = second.n_trav; c->hctr.alignments = second.alignments;
            c->hctr.flags = (first.flags & ~(kFlagOvfOverflow | kFlagTravOverflow)) | second.flags;
            continue;
        }
        if (c->hctr.flags & kFlagTravOverflow) {
            // the ordered output buffer is too small: the raw records are intact, only the ordering is redone
            if (int rc = alloc_trav(c, c->hctr.n_trav + c->hctr.n_trav / 8 + 1024)) return rc;
            HIP_TRY(c, hipMemsetAsync(&c->ctr.p->flags, 0, sizeof(unsigned int), c->stream));
            if (int rc = launch_order_stage(c)) return rc;
            DeviceCounters again{};
            if (int rc = fetch(again)) return rc;
            c->hctr.flags = (c->hctr.flags & ~kFlagTravOverflow) | again.flags;
            continue;
        }
        break;
    }
    c->n_trav = c->hctr.n_trav;
#ifdef GROOT_WORK_COUNTERS
    for (int e = 0; e < 32; e++)
        if (c->hctr.dbg[e]) fprintf(stderr, "[groot work] event %2d: wave iterations %llu lanes %llu\n", e, c->hctr.dbg[e], c->hctr.dbg[32 + e]);
    fprintf(stderr, "[groot work] longest round: %llu wave iterations\n", c->hctr.dbg[63]);
    for (int ph = 0; ph < 3; ph++)
        fprintf(stderr, "[groot work] phase %d: %llu steps, %.2f us per step (wall clock, per wave)\n", ph, c->hctr.dbg[27 + ph],
                c->hctr.dbg[27 + ph] ? (double)c->hctr.dbg[24 + ph] / 100.0 / (double)c->hctr.dbg[27 + ph] : 0.0);
    for (int h = 0; h < 2; h++) {
        fprintf(stderr, "[groot work] %s (buckets of 2 iterations):", h ? "round length" : "lane finish");
        for (int b = 0; b < 64; b++) fprintf(stderr, " %llu", c->hctr.dbg[64 + 64 * h + b]);
        fprintf(stderr, "\n");
    }
#endif
    if (c->profiling) {
        (void)hipEventElapsedTime(&c->ms.h2d, c->ev[0], c->ev[1]);
        (void)hipEventElapsedTime(&c->ms.sketch_seed, c->ev[1], c->ev[5]);
        (void)hipEventElapsedTime(&c->ms.schedule, c->ev[5], c->ev[2]);
        (void)hipEventElapsedTime(&c->ms.align, c->ev[2], c->ev[3]);
        (void)hipEventElapsedTime(&c->ms.sort, c->ev[3], c->ev[4]);
        (void)hipEventElapsedTime(&c->ms.total, c->ev[0], c->ev[4]);
    }
    c->finished = true;
    fill_counts(c, counts);
    if (c->hctr.flags & kFlagLongRead) return fail(c, GROOT_E_NOSPACE, "a read is longer than max_read_len=%u", c->prm.max_read_len);
    if (c->hctr.flags & kFlagOrdOverflow) return fail(c, GROOT_E_NOSPACE, "a read produced more than 65535 traversals");
    if (c->hctr.flags & kFlagShortRead)
        return fail(c, GROOT_E_SHORT_READ, "k size is greater than sequence length for %llu read(s) (the reference panics: boss.go:164-166)", c->hctr.short_reads);
    if (c->hctr.revcomp_panics)
        return fail(c, GROOT_E_REVCOMP, "%llu read(s) hold a byte > 'T' and reached RevComplement (the reference panics: seqio.go:126)", c->hctr.revcomp_panics);
    return GROOT_OK;
}

int groot_hip_read_seeds(groot_ctx *c, groot_seed *out, uint64_t cap, uint64_t *n)
{
    if (!c || !n) return GROOT_E_INVALID;
    if (!c->finished) return fail(c, GROOT_E_STATE, "no finished batch");
    HIP_TRY(c, hipSetDevice(c->device));
    const uint32_t R = c->n_reads;
    std::vector<uint32_t> cnt(R), win((size_t)c->seed_slots * R);
    if (R) {
        HIP_TRY(c, hipMemcpy(cnt.data(), c->seed_count.p, (size_t)R * 4, hipMemcpyDeviceToHost));
        for (uint32_t j = 0; j < c->seed_slots; j++)
            HIP_TRY(c, hipMemcpy(win.data() + (size_t)j * R, c->seed_win.p + (size_t)j * R, (size_t)R * 4, hipMemcpyDeviceToHost));
    }
    uint64_t total = 0;
    std::vector<uint32_t> tmp;
    for (uint32_t r = 0; r < R; r++) {
        const uint32_t m = std::min(cnt[r] & 0x7FFFFFFFu, c->seed_slots);
        tmp.clear();
        for (uint32_t j = 0; j < m; j++) tmp.push_back(win[(size_t)j * R + r]);
        std::sort(tmp.begin(), tmp.end());
        for (uint32_t w : tmp) {
            if (out && total < cap) out[total] = groot_seed{c->first_read_id + r, w};
            total++;
        }
    }
    *n = total;
    return GROOT_OK;
}

int groot_hip_read_travs(groot_ctx *c, groot_trav *out, uint64_t *masks, uint64_t cap, uint64_t *n)
{
    if (!c || !n) return GROOT_E_INVALID;
    if (!c->finished) return fail(c, GROOT_E_STATE, "no finished batch");
    HIP_TRY(c, hipSetDevice(c->device));
    *n = c->n_trav;
    const uint64_t m = std::min<uint64_t>(cap, c->n_trav);
    if (m && out) HIP_TRY(c, hipMemcpy(out, c->trav_sorted.p, m * sizeof(groot_trav), hipMemcpyDeviceToHost));
    if (m && masks) HIP_TRY(c, hipMemcpy(masks, c->trav_mask_sorted.p, m * c->pw_view * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return GROOT_OK;
}

int groot_hip_read_sketches(groot_ctx *c, uint64_t *out, uint64_t cap_reads, uint64_t *n_reads)
{
    if (!c || !n_reads) return GROOT_E_INVALID;
    if (!c->finished) return fail(c, GROOT_E_STATE, "no finished batch");
    if (!c->prm.keep_sketches) return fail(c, GROOT_E_STATE, "ctx was opened without keep_sketches");
    HIP_TRY(c, hipSetDevice(c->device));
    *n_reads = c->n_reads;
    const uint64_t m = std::min<uint64_t>(cap_reads, c->n_reads);
    if (m && out) HIP_TRY(c, hipMemcpy(out, c->sketches.p, m * c->s * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return GROOT_OK;
}

int groot_hip_stage_ms(groot_ctx *c, groot_stage_ms *out)
{
    if (!c || !out) return GROOT_E_INVALID;
    *out = c->ms;
    return GROOT_OK;
}

int groot_hip_attempts_shape(groot_ctx *c, uint32_t *n_q, uint32_t *n_windows)
{
    if (!c) return GROOT_E_INVALID;
    if (n_q) *n_q = c->max_q + 1;
    if (n_windows) *n_windows = c->n_windows;
    return GROOT_OK;
}

int groot_hip_attempts_device(groot_ctx *c, void **d_counts, uint64_t *n_elems)
{
    if (!c || !d_counts) return GROOT_E_INVALID;
    *d_counts = c->attempts_ptr;
    if (n_elems) *n_elems = (uint64_t)(c->max_q + 1) * c->n_windows;
    return GROOT_OK;
}

int groot_hip_attempts_read(groot_ctx *c, uint32_t *out, uint64_t n_elems)
{
    if (!c || !out) return GROOT_E_INVALID;
    const uint64_t have = (uint64_t)(c->max_q + 1) * c->n_windows;
    if (n_elems < have) return fail(c, GROOT_E_NOSPACE, "need room for %llu counts", (unsigned long long)have);
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (have) HIP_TRY(c, hipMemcpy(out, c->attempts_ptr, have * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return GROOT_OK;
}

int groot_hip_attempts_reset(groot_ctx *c)
{
    if (!c) return GROOT_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemsetAsync(c->attempts_ptr, 0, (size_t)(c->max_q + 1) * c->n_windows * sizeof(uint32_t), c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return GROOT_OK;
}

int groot_hip_attempts_bind(groot_ctx *c, void *d_counts, uint64_t n_elems)
{
    if (!c) return GROOT_E_INVALID;
    if (c->submitted && !c->finished) return fail(c, GROOT_E_STATE, "a batch is in flight");
    if (!d_counts) { c->attempts_ptr = c->attempts.p; return GROOT_OK; }
    const uint64_t need = (uint64_t)(c->max_q + 1) * c->n_windows;
    if (n_elems < need) return fail(c, GROOT_E_NOSPACE, "bound buffer holds %llu counts, need %llu", (unsigned long long)n_elems, (unsigned long long)need);
    c->attempts_ptr = (uint32_t *)d_counts;
    return GROOT_OK;
}

int groot_hip_sketch(groot_ctx *c, const uint8_t *seq_concat, const uint64_t *seq_off, uint32_t n, uint64_t *out)
{
    if (!c || !out || (n && (!seq_concat || !seq_off))) return GROOT_E_INVALID;
    if (c->submitted && !c->finished) return fail(c, GROOT_E_STATE, "a batch is in flight");
    if (!n) return GROOT_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    if (n > c->prm.max_batch_reads) return fail(c, GROOT_E_NOSPACE, "more sequences than max_batch_reads");
    const uint64_t total = seq_off[n];
    if (total > c->prm.max_batch_bases) return fail(c, GROOT_E_NOSPACE, "more bases than max_batch_bases");
    uint32_t max_len = 0;
    for (uint32_t i = 0; i < n; i++) max_len = std::max<uint32_t>(max_len, (uint32_t)(seq_off[i + 1] - seq_off[i]));
    DevBuf<uint64_t> sk;
    HIP_TRY(c, sk.alloc((size_t)n * c->s));
    HIP_TRY(c, hipMemcpyAsync(c->seq.p, seq_concat, total, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->seq_off.p, seq_off, ((size_t)n + 1) * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->ctr.p, 0, sizeof(DeviceCounters), c->stream));
    SeedArgs a{};
    a.ix = c->dix;
    a.ix.max_q = 0;   // no lookup: every read gets min_eq = S+1
    a.seq = c->seq.p; a.seq_off = c->seq_off.p; a.n_reads = n; a.max_read_len = c->prm.max_read_len;
    a.lds_read_bytes = (uint32_t)std::min<uint64_t>((uint64_t)kBlock * std::min(max_len, c->prm.max_read_len) + 32, kMaxLdsReadBytes);
    a.seed_slots = c->seed_slots; a.seed_count = c->seed_count.p; a.seed_win = c->seed_win.p;
    a.sketch_out = sk.p; a.sort_key = nullptr; a.read_rec = nullptr; a.ctr = c->ctr.p;
    launch_seed(c->s, a, true, dim3((n + kBlock - 1) / kBlock), kLdsReads + ((a.lds_read_bytes + 15) & ~15u), c->stream);
    HIP_TRY(c, hipGetLastError());
    DeviceCounters h{};
    HIP_TRY(c, hipMemcpyAsync(&h, c->ctr.p, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(out, sk.p, (size_t)n * c->s * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (h.flags & kFlagShortRead) return fail(c, GROOT_E_SHORT_READ, "k size is greater than sequence length");
    if (h.flags & kFlagLongRead) return fail(c, GROOT_E_NOSPACE, "a sequence is longer than max_read_len=%u", c->prm.max_read_len);
    return GROOT_OK;
}

} // extern "C"

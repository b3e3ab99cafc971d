// groot_hip.hip -- libgroot_hip.so: ctx management + the C ABI of include/groot_hip.h.
// gfx950 only; no CPU fallback anywhere in this library.
#include <cstring>   // before rocprim: its texture iterator calls host memset

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <atomic>
#include <chrono>
#include <deque>
#include <memory>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "../common/cpus.hpp"
#include "../common/view_check.hpp"
#include "kernels_misc.hpp"
#include "launch.hpp"

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
    hipError_t reserve(size_t count) { return count <= n && p ? hipSuccess : alloc(count); }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    ~DevBuf() { release(); }
};

// page-locked host memory: the only kind hipMemcpyAsync really overlaps with kernels
template <class T> struct PinBuf {
    T *p = nullptr;
    size_t n = 0;
    hipError_t alloc(size_t count)
    {
        release();
        n = count;
        return hipHostMalloc((void **)&p, std::max<size_t>(count, 1) * sizeof(T), hipHostMallocDefault);
    }
    hipError_t reserve(size_t count) { return count <= n && p ? hipSuccess : alloc(count); }
    void release()
    {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        n = 0;
    }
    ~PinBuf() { release(); }
};

// The environment switches of the shipped library, read when a ctx is opened (everything else that used to be tunable from the
// environment was an experiment and went in round 4: DESIGN.md "Removed").  The three NO_* switch a tier of the seed stage off (tests
// compare the tiers with each other and with the CPU checker); TEST_SMALL_BUFFERS starts every growable buffer and list too small, so that
// a test batch walks the grow-and-redo and the fall-back paths; OPEN_STATS prints where groot_hip_open spent its time.
struct Knobs {
    bool no_outcome_table = false, no_text_table = false, no_sig = false, force_rccl = false, small_buffers = false, open_stats = false, poison = false, lean = false;
    static Knobs read()
    {
        Knobs k;
        k.no_outcome_table = getenv("GROOT_NO_OUTCOME_TABLE") != nullptr; k.no_text_table = getenv("GROOT_NO_TEXT_TABLE") != nullptr;
        k.no_sig = getenv("GROOT_NO_SIG") != nullptr;                     k.force_rccl = getenv("GROOT_FORCE_RCCL") != nullptr;
        k.small_buffers = getenv("GROOT_TEST_SMALL_BUFFERS") != nullptr;  k.open_stats = getenv("GROOT_OPEN_STATS") != nullptr;
        k.poison = getenv("GROOT_TEST_POISON") != nullptr;
        k.lean = getenv("GROOT_LEAN") != nullptr;                         // the align stage WITH its first pass (kernels_lean.hpp; off by default: DESIGN.md section 3)
        return k;
    }
};

// One batch in flight.  Inputs and outputs are per slot (copy-in of batch b+1 and copy-out of batch b-1 overlap the
// kernels of batch b); everything the kernels only use between themselves is shared by all slots (one compute stream).
struct Slot {
    enum State { FREE, ACQUIRED, IN_FLIGHT, D2H_ISSUED, COLLECTED };
    State state = FREE;
    uint64_t ticket = 0;
    uint32_t n_reads = 0, first_read_id = 0, max_len = 0;
    uint32_t set = 0;                      // groot_ctx::ws the batch runs through
    uint32_t uniform_len = 0;              // IN_PACKED16: every read has this length (0 = lengths differ): no length array on the wire
    bool mixed_len = false;                // the reads are known to differ in length (the align stage then refills its wavefronts earlier)
    bool text_used = false;                // text_lookup_kernel ran first (the list behind it goes through the full-width kernel)
    bool sig_used = false;                 // the signature kernel ran in front of the full-width kernel for this batch
    uint32_t packed_q = 0;                 // SeedArgs::packed_q of the batch's signature kernel (0: it left no codes)
    bool lean_used = false;                // align_lean_kernel ran in front of align_kernel for this batch
    bool one_len = false;                  // the reads are known to have max_len bases each, or the caller said so (submit_device with max_len)
    uint64_t n_bases = 0, n_exc = 0;
    enum Input { IN_ASCII, IN_PACKED, IN_PACKED16, IN_DEVICE } input = IN_ASCII;
    const uint8_t *ext_seq = nullptr;      // IN_DEVICE
    const uint64_t *ext_off = nullptr;
    // pinned staging (inputs)
    PinBuf<uint8_t> h_bases;               // ASCII or packed bases
    PinBuf<uint16_t> h_len;
    PinBuf<uint64_t> h_off, h_exc_pos;
    PinBuf<uint8_t> h_exc_byte;
    // HBM inputs
    DevBuf<uint16_t> d_len;
    DevBuf<uint32_t> d_packed;
    bool exc_at_home = false;                      // the batch's exception list is read from the pinned staging by the kernel that applies it
    DevBuf<uint8_t> d_seq, d_exc_byte;
    DevBuf<uint64_t> d_off, d_exc_pos;
    // outputs
    uint32_t trav_cap = 0;
    DevBuf<groot_trav> d_trav;
    DevBuf<uint64_t> d_mask;
    DevBuf<DeviceCounters> d_ctr;
    PinBuf<DeviceCounters> h_ctr;
    PinBuf<groot_trav> h_trav;                     // what collect hands out (expanded on the host from h_ctrav when the records travel packed)
    DevBuf<groot_ctrav> d_ctrav;                   // 12-byte records for the copy-out
    PinBuf<groot_ctrav> h_ctrav;
    PinBuf<uint8_t> h_mask;                        // COMPACT path sets: ceil(paths(graph) / 8) bytes per traversal
    PinBuf<uint32_t> h_ckpt;                       // offset into h_mask of every 256th traversal
    DevBuf<uint8_t> d_cmask;                       // the compact copy the copy-out takes (host-result mode)
    DevBuf<uint32_t> d_mwords, d_moff, d_ckpt;
    uint32_t n_trav = 0, copied = 0;               // records of the batch / records the copy-out enqueued at submit covers
    uint64_t n_mask_bytes = 0, copied_bytes = 0;
    bool host_results = false;             // the traversal records of this batch are in h_trav / h_mask
    hipEvent_t ev_seed = nullptr;          // behind the batch's seed stage on the compute stream: its align stage waits for it
    hipEvent_t ev_h2d0 = nullptr, ev_h2d = nullptr, ev_compute = nullptr, ev_ctr = nullptr, ev_d2h0 = nullptr, ev_d2h = nullptr;
    hipEvent_t ev[14]{};                   // [7..8] around the first seed kernel, [9..10] around order_first_kernel, [11] start of the align stage (align stream), [12] behind the list pass
                                           // [0..6] stage boundaries on the compute stream (profiling)
    groot_counts counts{};
    int status = GROOT_OK;
    std::string status_msg;
    groot_stage_ms ms{};
    const uint8_t *seq() const { return input == IN_DEVICE ? ext_seq : d_seq.p; }
    const uint64_t *off() const { return input == IN_DEVICE ? ext_off : d_off.p; }
};

// One of the two sets of buffers a batch's seed stage fills for its align and order stages (groot_ctx::ws)
struct WorkSet {
    DevBuf<uint32_t> seed_count, seed_win, perm, perm_count, trav_cnt, tab_idx;
    DevBuf<uint32_t> perm2, perm2_count;                 // the slots align_lean_kernel left, in processing order, and how many
    DevBuf<uint8_t> defer;                               // LeanArgs::defer
    DevBuf<uint4> packed;                                // SeedArgs::packed
    DevBuf<ReadRec> read_rec;
    DevBuf<uint4> vitem, split_list;                     // AlignArgs::vitem, sort_seed_lists_kernel
    DevBuf<uint32_t> vcount;                             // [0] items, [1] split reads of the batch
    DevBuf<groot_trav> trav_first;
    DevBuf<uint64_t> mask_first, sketches;
    hipEvent_t ev_free = nullptr;          // on the align stream behind the order stage of the batch that used the set last
    bool used = false;
    Slot *owner = nullptr;                 // whose seeds / sketches the set holds
    uint64_t ticket = 0;
};

struct groot_ctx {
    int device = 0;
    std::string err;
    groot_params prm{};
    Knobs kn;
    uint32_t s = 0, k = 0, max_k = 0, l_max = 0, pw_view = 0, pw = 0, n_windows = 0, max_q = 0, band_hash_bits = 0;
    hipEvent_t h2d_last = nullptr;         // the copy-in of the newest host-fed batch (its slot's event)
    Slot *newest = nullptr;                // the newest submitted batch (groot_hip_redo_status)
    hipEvent_t last_compute = nullptr;     // behind the order stage of the newest batch (groot_hip_stream_join)
    hipStream_t own_stream = nullptr, stream = nullptr, astream = nullptr, h2d_stream = nullptr, d2h_stream = nullptr;   // stream: seed stage (the caller's, if given); astream: align + order stage
    bool profiling = false;

    // index in HBM
    DevBuf<uint32_t> graph_win_end;
    DevBuf<uint4> cn_pre;                  // DeviceIndex::cn_pre
    DevBuf<uint64_t> node_l2b;             // DeviceIndex::node_l2b
    DevBuf<uint32_t> win_prefix, edges, win_graph, cn_node,
        band_keys, band_ids;
    DevBuf<ExactEntry> band_hash;
    DevBuf<uint8_t> band_sig;
    DevBuf<uint32_t> band_run;
    DevBuf<uint8_t> bases, q_k, q_l;
    DevBuf<uint16_t> q_min_eq;
    DevBuf<uint64_t> win_sketch;
    DevBuf<unsigned char> node_rec;
    DevBuf<LeanExt> lean_ext;
    DevBuf<LeanNode> lean_nodes;           // first pass of the align stage (kernels_lean.hpp): nodes, graph bases and ContainedNodes prefixes at 2 bits per base
    DevBuf<uint32_t> bases2;
    DevBuf<uint4> cn_pre2;
    DevBuf<uint8_t> win_ok;
    DevBuf<uint4> lean_stk;                // LeanArgs::stk (align stream)
    bool lean = false;
    DevBuf<WinRec> win_rec;
    DevBuf<ExactEntry> exact;
    DevBuf<SigEntry> sig;                  // sketch_sig_kernel: signature index + window texts (absent: that kernel is not used)
    DevBuf<uint4> sig_dir;
    DevBuf<uint8_t> win_text, win_nodes;
    DevBuf<uint32_t> sig_info;             // per window-text string: verdict byte, or where its tabulated outcome is (DeviceIndex::sig_info)
    DevBuf<uint4> out_tab;                 // AlignRead outcomes of the window-text strings (DeviceIndex::out_tab)
    std::vector<uint32_t> h_out_tab;       // the host's copy (groot_hip_read_seeds: seed windows of reads the text lookup answered)
    uint64_t out_strings = 0, out_tabulated = 0, out_entries = 0;   // strings that confirm reads / of them tabulated / table entries
    double out_build_ms = 0, open_ms = 0;
    uint32_t incr_cap = kIncrCap;
    bool tab_capture = false;              // the capture pass of groot_hip_open is running (align stage records the IncrementSubPath windows)
    DevBuf<uint32_t> tab_idx, tab_hist, incr_cnt, incr_win;
    DevBuf<uint4> text_tab;                // text_lookup_kernel: strings with a tabulated outcome, keyed by their bases
    uint64_t text_entries = 0;
    uint32_t batches_without_text = 0, text_retry_gap = 8;   // the lookup is tried again after this many batches without it; the gap doubles (up to 256) while it keeps missing
    double text_hit_frac = 1.0;            // share of the latest batch's reads the outcome table answered: picks the first kernel of the seed stage
    std::vector<uint16_t> h_q_min_eq;      // host copy of DeviceIndex::q_min_eq: which seed kernel a batch of one read length gets
    uint32_t sig_disabled = 0;             // windows whose text did not reproduce Key.Sketch (they cannot confirm reads)
    DeviceIndex dix{};

    // groot_hip_open_flags(GROOT_OPEN_BACKGROUND): the prefix tables and the signature index are built on a thread of its own while the
    // first batches already run (through the full-width kernel, without the seed stage's verdicts: same results, a little slower);
    // what it builds is described in bg_dix and moves into dix between two batches (install_background)
    std::thread bg;
    std::atomic<int> bg_state{0};          // 0 nothing pending, 1 running, 2 finished, 3 failed, 4 abandoned
    std::atomic<bool> bg_cancel{false};    // groot_hip_open_abandon / groot_hip_close: the builder stops at its next checkpoint
    int bg_rc = 0;
    std::string bg_err;
    DeviceIndex bg_dix{};
    uint32_t bg_seed_slots = 0, bg_max_read_len = 0;   // what the builder thread may know of the ctx's mutable state: copies taken before it starts
    hipStream_t bg_stream = nullptr;
    DevBuf<unsigned long long> bg_shards;
    // where the table builders of groot_hip_open work: the ctx's own index description / compute stream / shard counters, or the
    // background thread's
    DeviceIndex *build_dix = nullptr;
    hipStream_t build_stream = nullptr;
    unsigned long long *build_shards = nullptr;

    // pipeline
    std::vector<std::unique_ptr<Slot>> slots;
    std::deque<Slot *> inflight;           // submission order: IN_FLIGHT / D2H_ISSUED
    uint64_t next_ticket = 1;
    Slot *waited = nullptr;                // the batch groot_hip_wait collected (released by the next submit / wait)
    double todo_frac = 1.0;                // share of the latest finished batch's reads that the first seed kernel left to the list pass
    double lean_left_frac = 1.0;           // share of the latest finished batch's reads that the first pass of the align stage left to the second
    uint32_t n_cu = 256;
    double dfs_frac = 1.0;                 // share of the latest finished batch's reads that needed the align stage's graph walk (the rest: no seeds / tabulated outcomes)
    double trav_per_read = 1.25;           // traversal records per read of the latest finished batch: sizes the next copy-out
    double bytes_per_trav = 0;             // compact path-set bytes per traversal, likewise (0 = not seen yet: 8 * path_words)
    bool packed_travs = false;             // the copy-out sends 12-byte records (batches of at most 2^24 reads), collect expands them
    std::vector<uint32_t> h_node_graph;    // graph of every node (the expansion)
    DevBuf<uint8_t> graph_words;           // ceil(paths / 8) per graph: BYTES of a traversal's compact path set
    std::vector<uint8_t> h_graph_words;
    // What the seed stage of a batch leaves for its align and order stages lives in one of TWO work sets, taken in turn: the seed
    // stage of batch b+1 (compute stream) runs beside the align + order stages of batch b (align stream) -- the reference's sketching
    // minions and graph minions run side by side too (boss.go:134-203, graphminion.go:46-102).  Hashing is VALU-issue bound, the
    // graph walk waits on dependent loads: they want different resources.
    WorkSet ws[2];
    uint32_t next_set = 0;

    // shared work buffers: used on ONE of the two streams only, inside one stage
    uint32_t seed_slots = 0;
    DevBuf<uint32_t> sort_key, sort_key_out, perm_in, todo_list, todo_count;   // seed stage
    DevBuf<uint32_t> long_list, long_count;              // SeedArgs::long_list (seed stage)
    uint32_t vcap = 0;
    uint32_t lsh_defer_rows = 0, lsh_cap = 0;   // SeedArgs::lsh_defer_rows
    DevBuf<unsigned long long> seed_shards;
    DevBuf<uint32_t> lsh_list, lsh_count;  // reads on the LSH-Forest branch with many candidate rows + their sketches, for lsh_heavy_kernel (seed stage)
    DevBuf<uint64_t> lsh_sketch;
    DevBuf<char> sort_tmp, in_tmp;         // rocprim scratch of the seed stage / of the input decoding (compute stream)
    uint32_t ovf_cap = 0;
    DevBuf<groot_trav> ovf_trav;           // align + order stage
    DevBuf<uint64_t> ovf_mask;
    DevBuf<uint32_t> trav_off, ovf_cnt;
    DevBuf<char> scan_tmp;                 // rocprim scratch of the order stage (align stream)
    // DFS stacks
    uint32_t align_threads = 0, stk_depth = 0;
    DevBuf<uint64_t> stk_hdr, stk_mask;
    // IncrementSubPath call counts: [rows][n_windows], one row per kmerCount that occurred
    DevBuf<uint32_t> attempts, q_row, q_seen, q_of_row, q_nrows;
    uint32_t *attempts_ptr = nullptr;      // own buffer or the caller's (groot_hip_attempts_layout)
    uint32_t att_cap = 0;                  // rows the table can hold
    bool att_external = false;
};

// A ctx drives four HIP streams at once -- seed stage, align + order stage, copy-in, copy-out -- beside whatever the host process
// uses itself.  The HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that share one
// run one after the other: with a fifth stream in the process the copy-in and the copy-out of neighbouring batches took turns
// (host-fed rate 1 355 -> 717 Mreads/s).  Ask for eight before the runtime reads the setting (first HIP call of the process); a
// value the user has set stands.
__attribute__((constructor)) static void groot_hw_queues() { setenv("GPU_MAX_HW_QUEUES", "8", 0); }

static thread_local std::string g_open_err;

static thread_local bool tl_background = false;       // this thread is a ctx's background builder

static int fail(groot_ctx *ctx, int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) (tl_background ? ctx->bg_err : ctx->err) = buf;
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

template <int PW> void build_node_records(const groot_index_view *v, std::vector<unsigned char> &out)
{
    std::vector<NodeRec<PW>> recs(v->n_nodes);
    for (uint32_t n = 0; n < v->n_nodes; n++) {
        NodeRec<PW> &r = recs[n];
        memset(&r, 0, sizeof r);
        r.seq_off = v->node_seq_off[n];
        r.seq_len = v->node_seq_off[n + 1] - v->node_seq_off[n];
        const uint32_t e0 = v->node_edge_off[n], deg = v->node_edge_off[n + 1] - e0;
        bool wild = false;
        for (uint32_t i = 0; i < r.seq_len; i++) wild |= v->bases[r.seq_off + i] == 'N';
        r.deg = deg | (wild ? 0x80000000u : 0u);        // bit 31: the node holds an 'N' (the wildcard of alignment.go:212-222)
        if (deg <= 4) {
            for (uint32_t e = 0; e < deg; e++) {
                const uint32_t c = v->edges[e0 + e];
                r.edges[e] = c;
                r.child_first[e] = v->node_seq_off[c] < v->node_seq_off[c + 1] ? v->bases[v->node_seq_off[c]] : (uint8_t)0;   // (an empty node spells nothing)
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
// batch execution
// ---------------------------------------------------------------------------------------------
static constexpr uint32_t kMaxLdsReadBytes = 64 * 1024;

static int alloc_seed_slots(groot_ctx *c, uint32_t slots)
{
    c->seed_slots = slots;
    // (both work sets get new buffers with a new stride: whatever seeds they held are gone -- groot_hip_read_seeds must not find an owner)
    for (WorkSet &w : c->ws) { HIP_TRY(c, w.seed_win.alloc((size_t)slots * c->prm.max_batch_reads)); w.owner = nullptr; w.ticket = 0; }
    return GROOT_OK;
}

static int alloc_trav(groot_ctx *c, Slot *s, uint32_t cap)
{
    s->trav_cap = cap;
    HIP_TRY(c, s->d_trav.alloc((size_t)cap + 2));
    HIP_TRY(c, s->d_mask.alloc((size_t)cap * c->pw_view + 2));
    if (!c->prm.results_on_device) {
        HIP_TRY(c, s->h_trav.alloc((size_t)cap + 2));
        HIP_TRY(c, s->h_mask.alloc((size_t)cap * c->pw_view * 8 + 16));
        HIP_TRY(c, s->h_ckpt.alloc((size_t)cap / 256 + 2));
        HIP_TRY(c, s->d_cmask.alloc((size_t)cap * c->pw_view * 8 + 16));
        HIP_TRY(c, s->d_mwords.alloc(cap));
        HIP_TRY(c, s->d_moff.alloc(cap));
        HIP_TRY(c, s->d_ckpt.alloc((size_t)cap / 256 + 2));
        if (c->packed_travs) {
            HIP_TRY(c, s->d_ctrav.alloc((size_t)cap + 2));
            HIP_TRY(c, s->h_ctrav.alloc((size_t)cap + 2));
        }
    }
    return GROOT_OK;
}

static int alloc_ovf(groot_ctx *c, uint32_t cap_per_shard)
{
    c->ovf_cap = cap_per_shard;
    HIP_TRY(c, c->ovf_trav.alloc((size_t)kOvfShards * cap_per_shard));
    HIP_TRY(c, c->ovf_mask.alloc((size_t)kOvfShards * cap_per_shard * c->pw));
    return GROOT_OK;
}

// call-count table with room for `rows` kmerCounts; existing rows are kept (device-to-device copy)
static int grow_attempts(groot_ctx *c, uint32_t rows)
{
    if (c->att_external) return fail(c, GROOT_E_NOSPACE, "a kmerCount outside the fixed layout of groot_hip_attempts_layout occurred");
    DevBuf<uint32_t> bigger;
    HIP_TRY(c, bigger.alloc((size_t)rows * c->n_windows));
    HIP_TRY(c, hipMemset(bigger.p, 0, (size_t)rows * c->n_windows * sizeof(uint32_t)));
    if (c->attempts.p && c->att_cap)
        HIP_TRY(c, hipMemcpy(bigger.p, c->attempts.p, (size_t)std::min(rows, c->att_cap) * c->n_windows * sizeof(uint32_t), hipMemcpyDeviceToDevice));
    std::swap(c->attempts.p, bigger.p);
    std::swap(c->attempts.n, bigger.n);
    c->attempts_ptr = c->attempts.p;
    c->att_cap = rows;
    return GROOT_OK;
}

constexpr double kSparseBelow = 0.05;   // share of a batch left for the graph walk below which the processing order is a stream compaction and half the persistent grid runs
struct HasKey {     // reads the seed stage left for the align stage's graph walk carry a scheduling key
    const uint32_t *key;
    __host__ __device__ bool operator()(uint32_t r) const { return key[r] != kEmpty; }
};
#ifndef GROOT_SPAN_BITS
#define GROOT_SPAN_BITS 6
#endif
// The list pass copies each read into its lane's LDS slice while five workgroups per CU still fit (reads up to ~104 bases: 1.77 vs 1.57
// Greads/s on 100-base reads with errors); for longer reads it reads the bases from HBM -- the copy would cost the fifth wavefront
// per SIMD and a dependent trip (seed stage of 8 M reads of 75..150 bases: 7.0 ms with the copy, 5.9 ms without).
static uint32_t list_lds_stride(uint32_t stride_dw)
{
    return kLdsReads + (uint64_t)kBlock * stride_dw * 4 <= 32 * 1024 ? stride_dw : 0;
}
// workgroups of the two wavefront-per-read kernels of the seed stage's tail (grid-stride over their lists): many small shares level out reads that
// cost between a few and a few thousand rows (round 5, 8 M reads of 75..150 bases, t = 0.99 / 0.90: heavy reads 512 / 2 048 / 8 192 / 32 768 workgroups
// -> 1 306 / 1 330 / 1 347 / 1 356 and 667 / 728 / 750 / 748 Mreads/s; seed-list sort 512 / 2 048 / 8 192 -> 1 283 / 1 330 / 1 360 and 680 / 728 / 748)
#ifndef GROOT_HEAVY_BLOCKS
#define GROOT_HEAVY_BLOCKS 16384u
#endif
#ifndef GROOT_SORTLIST_BLOCKS
#define GROOT_SORTLIST_BLOCKS 16384
#endif
static int launch_seed_stage(groot_ctx *c, Slot *s, bool update_weights)
{
    WorkSet *w = &c->ws[s->set];
    SeedArgs a{};
    s->packed_q = 0;
    a.ix = c->dix;
    a.seq = s->seq();
    a.seq_off = s->off();
    a.n_reads = s->n_reads;
    a.max_read_len = c->prm.max_read_len;
    const uint64_t want = (uint64_t)kBlock * s->max_len + 32;
    a.lds_read_bytes = (uint32_t)std::min<uint64_t>(want, kMaxLdsReadBytes);
    a.seed_slots = c->seed_slots;
    a.seed_count = w->seed_count.p;
    a.seed_win = w->seed_win.p;
    a.sketch_out = c->prm.keep_sketches ? w->sketches.p : nullptr;
    a.sort_key = c->sort_key.p;
    a.read_rec = w->read_rec.p;
    a.q_seen = c->q_seen.p;
    a.trav_cnt = w->trav_cnt.p;
    a.shards = c->seed_shards.p;
    a.ctr = s->d_ctr.p;
    a.long_list = c->long_list.p; a.long_count = c->long_count.p;
    if (c->dix.out_tab) {                                   // reads the signature kernel finds in the outcome table say so here
        a.tab_idx = w->tab_idx.p;
        a.tab_hist = c->tab_hist.p;
    }
    // processing order of the align stage: reads sorted by (node span of the first seed window, that window, likely
    // orientation).  key = span << (32-span_bits) | window << 2 | class; reads without seeds carry 0xFFFFFFFF and sort last
    unsigned win_bits = 2;                                  // 2 class bits + the bits of the largest window id
    for (uint32_t v = c->n_windows ? c->n_windows - 1 : 0; v; v >>= 1) win_bits++;
    win_bits = std::min(32u, std::max(3u, win_bits));
    // (round 5: the span class sits right above the window bits, five bits when that makes the sorted range 24 bits -- three passes of the radix sort
    // instead of four; the two class bits below the window are not sorted on: reads of one window are neighbours either way)
    a.sort_span_bits = std::min((unsigned)GROOT_SPAN_BITS, 32u - win_bits);
    if (win_bits - 2u + a.sort_span_bits > 24u && win_bits - 2u + 4u <= 24u) a.sort_span_bits = 24u - (win_bits - 2u);
    // (reads without seeds carry 0xFFFFFFFF: all ones in the span field, which no window has -- every window contains a node -- so they sort last)
    a.sort_span_shift = win_bits;
    // (the low window bits matter: sorted without the lowest 4 / 8 of them -- two radix passes instead of three -- configs[2] through the kernels ran at
    // 1 832 / 1 406 instead of 2 035 Mreads/s, the align kernel 3.6 / 5.4 ms instead of 3.1: neighbouring windows walk the same nodes)
    const unsigned begin_bit = 2u, end_bit = win_bits + a.sort_span_bits;
    const dim3 grid((s->n_reads + kBlock - 1) / kBlock);
    // workgroups of the list pass (grid-stride over the list).  Round 5: as many as the list is expected to need at a read per thread -- the latest
    // finished batch says how long it was --, not the 1 280 (five per CU) that are resident at once: reads of the LSH-Forest branch cost between a
    // few and a few hundred row visits, a workgroup that walks eight or nine sets of 256 of them in a fixed order ends when its slowest sets add up,
    // and workgroups dealt out as CUs become free level that (8 M reads of 75..150 bases, 2.84 M on the list, t = 0.99 / 0.90: 1 024 -> 1 229 / 665,
    // 1 280 -> 1 255 / 699, 2 560 -> 1 294 / 710, 5 120 -> 1 340 / 726, 20 480 -> 1 346 / 730 Mreads/s).  At least 1 280, so that a batch that
    // differs from the one before is not left with a handful.
#ifndef GROOT_LIST_BLOCKS_MIN
#define GROOT_LIST_BLOCKS_MIN 1280
#endif
    const uint32_t list_blocks = std::max<uint32_t>(GROOT_LIST_BLOCKS_MIN, (uint32_t)std::min<double>(4.0e6, c->todo_frac * 1.25 * (double)s->n_reads / kBlock + 1.0));
    // a batch of one read length that is not on the exact-table branch (lower thresholds, reads shorter than the windows) would
    // send every read through the list: the full-width kernel alone is 25-30 % faster then (tools/threshold_probe.py)
    bool sig_useful = true;
    if (s->one_len && s->max_len >= c->k) {
        const uint32_t q = s->max_len - c->k + 1;
        sig_useful = q < c->h_q_min_eq.size() && c->h_q_min_eq[q] == c->s;
    }
    s->sig_used = c->dix.sig && !c->prm.keep_sketches && sig_useful;
    // Which kernel sees the batch first?  When the outcome table answered most of the latest batch, the text lookup (no hashing at
    // all; what it does not find goes through the full-width kernel, read by read); else the signature kernel as before.
    // (the share is only known exactly while the lookup runs: it is tried again after 8, 16, ... 256 batches)
    const double list_below = kSparseBelow;
    const bool list_mode = c->dfs_frac < list_below;       // few reads need the graph walk (the latest batch says so)
    const bool text_try = c->text_hit_frac >= 0.7 || ++c->batches_without_text >= c->text_retry_gap;
    s->text_used = !c->prm.keep_sketches && c->dix.text_tab && c->dix.out_tab && text_try && s->max_len >= c->dix.w && !c->tab_capture;
    if (s->text_used) c->batches_without_text = 0;
    if (c->lsh_list.p && !c->prm.keep_sketches) {
        a.lsh_list = c->lsh_list.p; a.lsh_count = c->lsh_count.p; a.lsh_sketch = c->lsh_sketch.p;
        a.lsh_defer_rows = c->lsh_defer_rows; a.lsh_cap = c->lsh_cap;
        HIP_TRY(c, hipMemsetAsync(c->lsh_count.p, 0, 2 * sizeof(uint32_t), c->stream));
    }
    HIP_TRY(c, hipMemsetAsync(w->vcount.p, 0, 2 * sizeof(uint32_t), c->stream));
    if (c->profiling) HIP_TRY(c, hipEventRecord(s->ev[7], c->stream));
    if (s->text_used) {
        a.todo_list = c->todo_list.p;
        a.todo_count = c->todo_count.p;
        const uint32_t stride_dw = ((s->max_len + 3) / 4 + 1) | 1u;
        a.list_stride_dw = list_lds_stride(stride_dw);
        const size_t lds = kTextBad + (size_t)((a.lds_read_bytes + 15) / 16) * 4 + 96;
        HIP_TRY(c, hipMemsetAsync(c->todo_count.p, 0, sizeof(uint32_t), c->stream));
        if (list_mode) {       // the reads left for the graph walk are a subset of the lookup's misses: the list pass appends them itself
            a.dfs_list = w->perm.p; a.dfs_count = w->perm_count.p;
            HIP_TRY(c, hipMemsetAsync(w->perm_count.p, 0, sizeof(uint32_t), c->stream));
        }
        launch_text_lookup(text_key_dwords((c->dix.w + 15) / 16), a, grid, lds, c->stream);
        HIP_TRY(c, hipGetLastError());
        if (c->profiling) HIP_TRY(c, hipEventRecord(s->ev[8], c->stream));
        launch_list(c->s, a, dim3(std::min<uint32_t>(grid.x, list_blocks)), c->stream);
    } else if (s->sig_used) {
        // signature kernel first; what it cannot decide goes through the full-width kernel, read by read
        a.todo_list = c->todo_list.p;
        a.todo_count = c->todo_count.p;
        const uint32_t stride_dw = ((s->max_len + 3) / 4 + 1) | 1u;     // the LIST pass copies each read into its lane's LDS slice
        a.list_stride_dw = list_lds_stride(stride_dw);
        HIP_TRY(c, hipMemsetAsync(c->todo_count.p, 0, sizeof(uint32_t), c->stream));
        // (the reads it decides it also leaves as 2-bit codes for the first pass of the align stage)
        if (c->lean && !c->tab_capture && w->packed.p) { a.packed = w->packed.p; a.packed_q = s->max_len <= 128 ? 2u : 4u; }
        s->packed_q = a.packed ? a.packed_q : 0;
        launch_sig(c->s, a, s->max_len, c->stream);
        HIP_TRY(c, hipGetLastError());
        if (c->profiling) HIP_TRY(c, hipEventRecord(s->ev[8], c->stream));
        launch_list(c->s, a, dim3(std::min<uint32_t>(grid.x, list_blocks)), c->stream);
    } else {
        const size_t lds = kLdsReads + ((a.lds_read_bytes + 15) & ~15u);
        launch_seed(c->s, c->max_k, a, c->prm.keep_sketches != 0, grid, lds, c->stream);
        HIP_TRY(c, hipGetLastError());
        if (c->profiling) HIP_TRY(c, hipEventRecord(s->ev[8], c->stream));
    }
    // the reads of the LSH-Forest branch with many candidate rows: a wavefront each
    if (a.lsh_list) hipLaunchKernelGGL(lsh_heavy_kernel, dim3(std::min<uint32_t>(grid.x, GROOT_HEAVY_BLOCKS)), dim3(kBlock), 0, c->stream, a);
    HIP_TRY(c, hipGetLastError());
    if (c->profiling) HIP_TRY(c, hipEventRecord(s->ev[12], c->stream));   // (the list pass behind the first kernel + the heavy LSH-Forest reads)
    // seed lists of more than four windows that are not ascending (LSH-Forest hits come in band order): sorted, a wavefront per read
    // ... and the longest ones cut into items that different lanes of the align stage take
    {
        SplitArgs sa{};
        sa.list = c->long_list.p; sa.count = c->long_count.p; sa.seed_count = w->seed_count.p; sa.seed_win = w->seed_win.p;
        sa.n_reads = s->n_reads; sa.seed_slots = c->seed_slots; sa.read_rec = w->read_rec.p; sa.win_rec = c->dix.win_rec;
        sa.split = c->vcap && !c->tab_capture && !c->prm.no_exact_align;
        sa.vitem = w->vitem.p; sa.vcount = w->vcount.p; sa.vcap = c->vcap; sa.split_list = w->split_list.p;
        sa.ctr = s->d_ctr.p; sa.update_weights = update_weights ? 1 : 0;
        hipLaunchKernelGGL(sort_seed_lists_kernel, dim3(GROOT_SORTLIST_BLOCKS), dim3(kBlock), 0, c->stream, sa);
    }
    if (c->profiling) HIP_TRY(c, hipEventRecord(s->ev[2], c->stream));
    hipLaunchKernelGGL(assign_q_rows_kernel, dim3(1), dim3(64), 0, c->stream, c->q_seen.p, c->q_row.p, c->q_of_row.p, c->q_nrows.p, c->att_cap,
                       c->max_q, s->d_ctr.p, c->seed_shards.p, c->long_count.p);
    if (a.tab_hist) {
        hipLaunchKernelGGL(fold_tab_hist_kernel, dim3(std::min<uint32_t>((c->n_windows + kBlock - 1) / kBlock, 1024u)), dim3(kBlock), 0, c->stream, c->tab_hist.p,
                           c->attempts_ptr, c->q_row.p, c->dix.w - c->k + 1, c->n_windows, s->d_ctr.p, update_weights ? 1u : 0u);
        HIP_TRY(c, hipGetLastError());
    }
    if (list_mode && a.dfs_list) return GROOT_OK;           // (the processing order was written by the seed stage)
    if (list_mode) {
        // Few reads need the graph walk (the latest batch says so; most are answered from the outcome table or have no seeds): sorting
        // ten million keys to order a few of them costs more than their order saves.  The processing order is then simply the reads
        // with a key, ascending -- one stream compaction (0.05 instead of 0.45 ms per 10 M reads).  Processing order only.
        size_t tb = 0;
        HasKey pred{c->sort_key.p};
        rocprim::counting_iterator<uint32_t> ids(0u);
        HIP_TRY(c, rocprim::select(nullptr, tb, ids, w->perm.p, w->perm_count.p, (size_t)s->n_reads, pred, c->stream));
        if (tb > c->sort_tmp.n) {
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            HIP_TRY(c, c->sort_tmp.alloc(tb + tb / 4));
        }
        HIP_TRY(c, rocprim::select(c->sort_tmp.p, tb, ids, w->perm.p, w->perm_count.p, (size_t)s->n_reads, pred, c->stream));
        return GROOT_OK;
    }
    size_t tmp_bytes = 0;
    // keys are (window << 2 | class) below 2^end_bit, or 0xFFFFFFFF for reads without seeds: sorting the low
    // end_bit bits keeps those last as long as bit end_bit-1.. are all ones for them, which they are
    HIP_TRY(c, rocprim::radix_sort_pairs(nullptr, tmp_bytes, c->sort_key.p, c->sort_key_out.p, c->perm_in.p, w->perm.p, s->n_reads, begin_bit,
                                         end_bit, c->stream));
    if (tmp_bytes > c->sort_tmp.n) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        HIP_TRY(c, c->sort_tmp.alloc(tmp_bytes + tmp_bytes / 4));
    }
    HIP_TRY(c, rocprim::radix_sort_pairs(c->sort_tmp.p, tmp_bytes, c->sort_key.p, c->sort_key_out.p, c->perm_in.p, w->perm.p,
                                         s->n_reads, begin_bit, end_bit, c->stream));
    return GROOT_OK;
}

// which slots of the processing order does align_kernel take after the first pass?  Those the first pass flagged, and those it did not get to
// (its grid is sized by the latest batch: slots beyond it)
struct LeanLeft {
    const uint8_t *defer;
    const DeviceCounters *ctr;
    uint32_t lean_slots;
    __device__ uint8_t operator()(uint32_t i) const { return i < ctr->seeded_reads && (i >= lean_slots || defer[i]) ? 1 : 0; }
};

static int launch_align_stage(groot_ctx *c, Slot *s, bool update_weights)
{
    WorkSet *w = &c->ws[s->set];
    AlignArgs a{};
    a.ix = c->dix;
    a.seq = s->seq();
    a.seq_off = s->off();
    a.n_reads = s->n_reads;
    a.first_read_id = s->first_read_id;
    a.seed_slots = c->seed_slots;
    a.seed_count = w->seed_count.p;
    a.seed_win = w->seed_win.p;
    a.perm = w->perm.p;
    a.read_rec = w->read_rec.p;
    a.no_align = c->prm.no_exact_align;
    a.update_weights = update_weights ? 1 : 0;
    a.attempts = c->attempts_ptr;
    a.node_rec = c->node_rec.p;
    a.trav_first = w->trav_first.p;
    a.mask_first = w->mask_first.p;
    a.trav_cnt = w->trav_cnt.p;
    a.ovf_trav = c->ovf_trav.p;
    a.ovf_mask = c->ovf_mask.p;
    a.ovf_cnt = c->ovf_cnt.p;
    a.ovf_cap = c->ovf_cap;
    if (c->vcap) { a.vitem = w->vitem.p; a.vcount = w->vcount.p; a.vcap = c->vcap; }
    a.stk_hdr = c->stk_hdr.p;
    a.stk_mask = c->stk_mask.p;
    uint32_t blocks = std::min<uint32_t>((s->n_reads + kBlock - 1) / kBlock, c->align_threads / kBlock);
    // (few reads left for the walk -- the latest batch says so: half the persistent grid starts and drains 0.05 ms sooner and the
    // slowest read, not the number of wavefronts, sets the duration anyway)
    // (... unless there are reads enough to give every wavefront of the whole grid a round of 16: reads that fail -- reads with an error that
    // kept their minimisers -- do not march in step, and more wavefronts with fewer of them each end sooner)
    if (c->dfs_frac < kSparseBelow && c->dfs_frac * (double)s->n_reads < 16.0 * (double)(blocks * (kBlock / 64))) blocks = std::max(1u, blocks / 2);
    // (round 5: when fewer than six reads in ten need the walk -- reads with errors: the exact ones seed, the others do not -- the hashing kernels of the
    // next batch are the longer stage, and a persistent grid of two workgroups per CU leaves them half the registers: configs[2] with 1 % substitutions
    // 2 345 -> 2 680 Mreads/s; no difference on mixed-length batches; on error-free reads, where every read is walked, the full grid is 4 % faster)
    else if (c->dfs_frac >= kSparseBelow && c->dfs_frac < 0.6 && !s->mixed_len && blocks >= 4) blocks /= 2;
    // (the first pass takes most reads: what it left in the latest batch sizes the persistent grid of the second -- a wavefront per 64 reads left,
    // at least one workgroup per CU; the registers it does not hold go to the next batch's hashing kernels)
    if (c->lean && !c->tab_capture && c->lean_left_frac < 0.25) {
        uint32_t want = (uint32_t)(c->lean_left_frac * 1.25 * (double)s->n_reads / 64.0 / (kBlock / 64)) + 1u;
        blocks = std::max(1u, std::min(blocks, std::max(want, c->n_cu)));
    }
    a.n_threads = blocks * kBlock;
    a.stk_depth = c->stk_depth;
    // stage reads in LDS when 256 lanes x (longest read + slack) stays within 64 KB
    {
        // 8 zero bytes, then the read in whole 16-byte pieces up to 12 bytes past its end; odd dword stride = no bank conflicts
        const uint32_t stride = (2 + 4 * ((s->max_len + 27) / 16)) | 1u;
        a.lds_stride_dw = (size_t)kBlock * stride * 4 <= 64 * 1024 ? stride : 0;
    }
    if (c->tab_capture) {
        a.incr_cnt = c->incr_cnt.p; a.incr_win = c->incr_win.p; a.incr_cap = c->incr_cap;
        HIP_TRY(c, hipMemsetAsync(c->incr_cnt.p, 0, (size_t)s->n_reads * sizeof(uint32_t), c->astream));
    }
    a.head_lanes = s->mixed_len ? 16u : 0u;              // (8: best at 2 M reads before the items of split reads took the head; 16: 2.9 / 5.2 ms at 2 M / 8 M reads, 8 gave 3.05 / 5.6)
    // (round 5: 48 when fewer than six reads in ten are walked -- reads with errors: some align at once, some fail through the hierarchy, and lanes that
    // have finished take new reads before the whole round has: configs[2] with 1 % substitutions, memo off, 2 625 -> 2 740 Mreads/s; on error-free reads,
    // which march in step, 64 stays: 1 966 against 1 892 / 1 895 / 1 908 at 32 / 48 / 56)
#ifndef GROOT_REFILL_ERR
#define GROOT_REFILL_ERR 48       // (2 / 16 / 32 / 48: 2 674 / 2 712 / 2 736 / 2 755 Mreads/s on configs[2] with 1 % substitutions, memo off)
#endif
#ifndef GROOT_REFILL_MIXED
#define GROOT_REFILL_MIXED 2      // mixed read lengths: a lane that has finished takes its next read at once -- no rounds (2 / 8 / 16 / 32 / 48: 1 258 / 1 253 / 1 249 / 1 208 /
                                  // 1 172 Mreads/s at t = 0.99 on 8 M reads of 75..150 bases, 694 / 678 / 680 / 676 / 653 at t = 0.90; batches of 2 M: 850-890 -> 912)
#endif
    a.refill = s->mixed_len ? (uint32_t)GROOT_REFILL_MIXED : (c->dfs_frac >= kSparseBelow && c->dfs_frac < 0.6 ? (uint32_t)GROOT_REFILL_ERR : 64u);                   // reads of many lengths finish their walks far apart (tools/mixed_probe.py)
#ifdef GROOT_WORK_COUNTERS
    if (const char *e = getenv("GROOT_DEV_ROUND")) a.round_lanes = (uint32_t)atoi(e);   // instrumented builds only (tools/slow_reads_probe.py: one read per round)
#endif
    a.ctr = s->d_ctr.p;
    HIP_TRY(c, hipMemsetAsync(c->ovf_cnt.p, 0, (kOvfShards + 2) * sizeof(uint32_t), c->astream));   // + the two chunk cursors
    // First pass (kernels_lean.hpp): a thread per read in processing order finishes the reads of one seed window whose walks never branch;
    // the slots it leaves are flagged, a stream compaction keeps them in processing order, and align_kernel takes that list.
    const uint32_t lean_stride = lean_stride_dw(s->max_len);
    // (GROOT_LEAN=1: every batch that has reads to walk.  Measured, DESIGN.md section 3: alone on the chip the two passes take 2.2 + 0.7 ms where align_kernel
    // takes 3.0 on configs[2]; beside the next batch's hashing kernels the step is between 1.5 % shorter and 8 % longer from box to box, and batches of
    // mixed lengths or of reads with errors are slower -- hence off by default.)
    s->lean_used = c->lean && !c->tab_capture && s->n_reads && s->max_len <= kLeanMaxLen && c->dfs_frac >= 0.02;
    if (s->lean_used) {
        LeanArgs l{};
        l.nodes = c->lean_nodes.p; l.ext = c->lean_ext.p; l.bases2 = c->bases2.p; l.cn_pre2 = c->cn_pre2.p; l.win_ok = c->win_ok.p;
        l.win_rec = c->dix.win_rec; l.node_l2b = c->dix.node_l2b; l.q_row = c->q_row.p;
        l.seq = s->seq(); l.packed = s->packed_q ? w->packed.p : nullptr; l.packed_q = s->packed_q; l.perm = w->perm.p; l.read_rec = w->read_rec.p;
        l.n_reads = s->n_reads; l.first_read_id = s->first_read_id; l.n_windows = c->n_windows; l.k = c->k;
        l.update_weights = update_weights ? 1 : 0;
        l.lds_stride_dw = lean_stride; l.max_len = s->max_len;
        l.attempts = c->attempts_ptr;
        l.trav_first = w->trav_first.p; l.mask_first = w->mask_first.p; l.trav_cnt = w->trav_cnt.p;
        l.defer = w->defer.p; l.ctr = s->d_ctr.p;
        l.stk = c->lean_stk.p; l.ovf_trav = c->ovf_trav.p; l.ovf_mask = c->ovf_mask.p; l.ovf_cnt = c->ovf_cnt.p; l.ovf_cap = c->ovf_cap;
        // workgroups for the reads expected to have seeds (the latest batch says how many: they come first in the processing order); the slots
        // beyond them, if the batch has more, go to align_kernel like the flagged ones
        const uint32_t lean_blocks = std::min<uint32_t>((s->n_reads + kBlock - 1) / kBlock, (uint32_t)(c->dfs_frac * 1.05 * (double)s->n_reads / kBlock) + 64u);
        launch_align_lean(c->pw, l, dim3(lean_blocks), c->astream);
        HIP_TRY(c, hipGetLastError());
        size_t tb = 0;
        rocprim::counting_iterator<uint32_t> ids(0u);
        auto flags = rocprim::make_transform_iterator(ids, LeanLeft{w->defer.p, s->d_ctr.p, lean_blocks * (uint32_t)kBlock});
        HIP_TRY(c, rocprim::select(nullptr, tb, w->perm.p, flags, w->perm2.p, w->perm2_count.p, (size_t)s->n_reads, c->astream));
        if (tb > c->scan_tmp.n) {
            HIP_TRY(c, hipStreamSynchronize(c->astream));
            HIP_TRY(c, c->scan_tmp.alloc(tb + tb / 4));
        }
        HIP_TRY(c, rocprim::select(c->scan_tmp.p, tb, w->perm.p, flags, w->perm2.p, w->perm2_count.p, (size_t)s->n_reads, c->astream));
        a.perm = w->perm2.p;
        a.n_perm = w->perm2_count.p;
        if (c->profiling) HIP_TRY(c, hipEventRecord(s->ev[13], c->astream));
    }
    launch_align(c->pw, a, dim3(blocks), c->astream);
    HIP_TRY(c, hipGetLastError());
    return GROOT_OK;
}

// traversal records -> (read, ord) order: exclusive scan of the per-read counts, then two scatters into the slot's output
static int launch_order_stage(groot_ctx *c, Slot *s, bool update_weights)
{
    WorkSet *w = &c->ws[s->set];
    const uint32_t n = s->n_reads;
    size_t tmp_bytes = 0;
    // split reads: their items' counts become the read's count, every item learns where its records go in the read's run
    if (c->vcap) hipLaunchKernelGGL(split_fix_kernel, dim3(256), dim3(kBlock), 0, c->astream, w->split_list.p, w->vcount.p, w->vitem.p, w->trav_cnt.p, n, s->d_ctr.p,
                                    w->trav_first.p, w->mask_first.p, c->pw, s->first_read_id);
    HIP_TRY(c, rocprim::exclusive_scan(nullptr, tmp_bytes, w->trav_cnt.p, c->trav_off.p, 0u, n, rocprim::plus<uint32_t>(), c->astream));
    if (tmp_bytes > c->scan_tmp.n) {
        HIP_TRY(c, hipStreamSynchronize(c->astream));
        HIP_TRY(c, c->scan_tmp.alloc(tmp_bytes + tmp_bytes / 4));
    }
    HIP_TRY(c, rocprim::exclusive_scan(c->scan_tmp.p, tmp_bytes, w->trav_cnt.p, c->trav_off.p, 0u, n, rocprim::plus<uint32_t>(), c->astream));
    hipLaunchKernelGGL(order_total_kernel, dim3(1), dim3(1), 0, c->astream, c->trav_off.p, w->trav_cnt.p, n, s->d_ctr.p);
    OrderTabArgs ot{};
    if (c->dix.out_tab) {
        ot.tab_idx = w->tab_idx.p; ot.out_tab = c->dix.out_tab; ot.stride_q = c->dix.out_stride_q; ot.first_read_id = s->first_read_id;
        ot.update_weights = update_weights ? 1 : 0;
        ot.attempts = c->attempts_ptr; ot.q_row = c->q_row.p;
        ot.q_tab = c->dix.w - c->k + 1; ot.n_windows = c->n_windows;
    }
    if (c->profiling) HIP_TRY(c, hipEventRecord(s->ev[9], c->astream));
    hipLaunchKernelGGL(order_first_kernel, dim3(std::min<uint32_t>((n + kBlock - 1) / kBlock, 2048u)), dim3(kBlock), 0, c->astream, w->trav_first.p,
                       w->mask_first.p, c->trav_off.p, w->trav_cnt.p, n, s->d_trav.p, s->d_mask.p, s->trav_cap, c->pw,
                       c->pw_view, s->d_ctr.p, ot);
    if (c->profiling) HIP_TRY(c, hipEventRecord(s->ev[10], c->astream));
    if (c->vcap) hipLaunchKernelGGL(order_split_kernel, dim3(std::min<uint32_t>((c->vcap + kBlock - 1) / kBlock, 256u)), dim3(kBlock), 0, c->astream, w->vitem.p, w->vcount.p, c->vcap, w->trav_cnt.p,
                                    c->trav_off.p, w->trav_first.p, w->mask_first.p, n, s->first_read_id, s->d_trav.p, s->d_mask.p, s->trav_cap, c->pw, c->pw_view, s->d_ctr.p);
    hipLaunchKernelGGL(order_ovf_kernel, dim3((c->ovf_cap + kBlock - 1) / kBlock, kOvfShards), dim3(kBlock), 0, c->astream,
                       c->ovf_trav.p, c->ovf_mask.p, c->ovf_cnt.p, c->ovf_cap, c->trav_off.p, s->first_read_id, s->d_trav.p,
                       s->d_mask.p, s->trav_cap, c->pw, c->pw_view, s->d_ctr.p, w->vitem.p, n);
    HIP_TRY(c, hipGetLastError());
    if (!c->prm.results_on_device) {
        // compact path sets for the copy-out (kernels.hpp): words per traversal, their exclusive scan, the copy
        const dim3 g((s->trav_cap + kBlock - 1) / kBlock);
        hipLaunchKernelGGL(mask_words_kernel, g, dim3(kBlock), 0, c->astream, s->d_trav.p, s->d_ctr.p, s->trav_cap, c->graph_words.p, (uint32_t)c->h_graph_words.size(),
                           s->d_mwords.p);
        size_t tb = 0;
        HIP_TRY(c, rocprim::exclusive_scan(nullptr, tb, s->d_mwords.p, s->d_moff.p, 0u, s->trav_cap, rocprim::plus<uint32_t>(), c->astream));
        if (tb > c->scan_tmp.n) {
            HIP_TRY(c, hipStreamSynchronize(c->astream));
            HIP_TRY(c, c->scan_tmp.alloc(tb + tb / 4));
        }
        HIP_TRY(c, rocprim::exclusive_scan(c->scan_tmp.p, tb, s->d_mwords.p, s->d_moff.p, 0u, s->trav_cap, rocprim::plus<uint32_t>(), c->astream));
        hipLaunchKernelGGL(mask_compact_kernel, g, dim3(kBlock), 0, c->astream, s->d_trav.p, s->d_mask.p, c->pw_view, s->d_ctr.p, s->trav_cap,
                           c->graph_words.p, (uint32_t)c->h_graph_words.size(), s->d_moff.p, s->d_cmask.p, s->d_ckpt.p);
        if (c->packed_travs) hipLaunchKernelGGL(trav_pack_kernel, g, dim3(kBlock), 0, c->astream, s->d_trav.p, s->d_ctr.p, s->trav_cap, s->first_read_id, s->d_ctrav.p);
        HIP_TRY(c, hipGetLastError());
    }
    return GROOT_OK;
}

// sketch+seed -> schedule (compute stream) | align -> order (align stream) for the batch of slot s
static int run_batch_async(groot_ctx *c, Slot *s, bool update_weights)
{
    WorkSet *w = &c->ws[s->set];
    // compute stream: the seed stage, once the batch that used this work set last is through its order stage
    if (w->used) HIP_TRY(c, hipStreamWaitEvent(c->stream, w->ev_free, 0));
    HIP_TRY(c, hipMemsetAsync(s->d_ctr.p, 0, sizeof(DeviceCounters), c->stream));
    if (c->kn.poison) {
        // GROOT_TEST_POISON: what the seed stage writes per read is wiped first.  A work set keeps the values of the batch that used it last, and a
        // stream of equal batches hides a read that no kernel of the seed stage handled -- round 4's one-in-a-million signature kernel dropped a
        // dozen reads per 10 M from the list of the full-width pass, visible only in the first batch through each work set (tools/first_use_check.py)
        const size_t n = s->n_reads;
        HIP_TRY(c, hipMemsetAsync(w->seed_count.p, 0, n * sizeof(uint32_t), c->stream));
        HIP_TRY(c, hipMemsetAsync(w->read_rec.p, 0, n * sizeof(ReadRec), c->stream));
        HIP_TRY(c, hipMemsetAsync(w->trav_cnt.p, 0, n * sizeof(uint32_t), c->stream));
        HIP_TRY(c, hipMemsetAsync(c->sort_key.p, 0xFF, n * sizeof(uint32_t), c->stream));
        if (w->tab_idx.p) HIP_TRY(c, hipMemsetAsync(w->tab_idx.p, 0xFF, n * sizeof(uint32_t), c->stream));
        HIP_TRY(c, hipMemsetAsync(c->todo_list.p, 0, n * sizeof(uint32_t), c->stream));
    }
    if (c->profiling) HIP_TRY(c, hipEventRecord(s->ev[1], c->stream));
    if (int rc = launch_seed_stage(c, s, update_weights)) return rc;
    if (c->profiling) HIP_TRY(c, hipEventRecord(s->ev[3], c->stream));
    HIP_TRY(c, hipEventRecord(s->ev_seed, c->stream));
    // align stream: graph walk and ordering, beside the seed stage of the next batch
    HIP_TRY(c, hipStreamWaitEvent(c->astream, s->ev_seed, 0));
    if (c->profiling) HIP_TRY(c, hipEventRecord(s->ev[11], c->astream));
    if (int rc = launch_align_stage(c, s, update_weights)) return rc;
    if (c->profiling) HIP_TRY(c, hipEventRecord(s->ev[4], c->astream));
    if (int rc = launch_order_stage(c, s, update_weights)) return rc;
    if (c->profiling) HIP_TRY(c, hipEventRecord(s->ev[5], c->astream));
    HIP_TRY(c, hipEventRecord(w->ev_free, c->astream));
    w->used = true;
    w->owner = s;
    w->ticket = s->ticket;
    return GROOT_OK;
}

// lengths on the wire -> u64 offsets in HBM
struct LenToU64 {
    __host__ __device__ uint64_t operator()(uint16_t v) const { return (uint64_t)v; }
};

constexpr uint64_t kExcAtHomeBytes = 2u << 20;

static void par_copy(void *dst, const void *src, size_t bytes)
{
    const size_t kMin = 8u << 20;
    unsigned nt = (unsigned)std::min<size_t>(8, bytes / kMin);
    if (nt <= 1) { memcpy(dst, src, bytes); return; }
    std::vector<std::thread> th;
    const size_t per = ((bytes + nt - 1) / nt + 63) & ~(size_t)63;
    for (unsigned t = 0; t < nt; t++) {
        const size_t lo = std::min(bytes, t * per), hi = std::min(bytes, lo + per);
        if (lo < hi) th.emplace_back([=]() { memcpy((char *)dst + lo, (const char *)src + lo, hi - lo); });
    }
    for (auto &x : th) x.join();
}

// input staging + output buffers of a slot, sized once for the ctx's batch capacity
static int ensure_slot(groot_ctx *c, Slot *s, Slot::Input in, uint64_t n_exc)
{
    const uint32_t R = c->prm.max_batch_reads;
    const uint64_t B = c->prm.max_batch_bases;
    if (!s->d_ctr.p) {
        HIP_TRY(c, s->d_ctr.alloc(1));
        HIP_TRY(c, s->h_ctr.alloc(1));
        // (GROOT_TEST_SMALL_BUFFERS: start with buffers that every batch outgrows, so that the tests walk the grow-and-redo paths)
        const bool tiny = c->kn.small_buffers;
        if (int rc = alloc_trav(c, s, tiny ? 64u : std::max<uint32_t>(1024, R + R / 4))) return rc;
    }
    if (in == Slot::IN_DEVICE) return GROOT_OK;
    HIP_TRY(c, s->d_seq.reserve(B + 64));
    HIP_TRY(c, s->d_off.reserve((size_t)R + 1));
    if (in == Slot::IN_ASCII) {
        HIP_TRY(c, s->h_bases.reserve(B + 64));
        HIP_TRY(c, s->h_off.reserve((size_t)R + 1));
        return GROOT_OK;
    }
    HIP_TRY(c, s->h_bases.reserve((B + 3) / 4 + 64));
    HIP_TRY(c, s->d_packed.reserve((B + 15) / 16 + 1));
    if (in == Slot::IN_PACKED) HIP_TRY(c, s->h_off.reserve((size_t)R + 1));
    else {
        HIP_TRY(c, s->h_len.reserve(R));
        HIP_TRY(c, s->d_len.reserve(R));
    }
    const uint64_t exc_cap = std::max<uint64_t>(n_exc + n_exc / 4, std::max<uint64_t>(4096, B / 256));
    if (s->h_exc_pos.n < std::max<uint64_t>(n_exc, 1)) {
        HIP_TRY(c, s->h_exc_pos.alloc(exc_cap)); HIP_TRY(c, s->h_exc_byte.alloc(exc_cap));
        HIP_TRY(c, s->d_exc_pos.alloc(exc_cap)); HIP_TRY(c, s->d_exc_byte.alloc(exc_cap));
    }
    return GROOT_OK;
}

static void release_slot(groot_ctx *c, Slot *s)
{
    s->state = Slot::FREE;
    s->host_results = false;
    if (c->waited == s) c->waited = nullptr;
}

static Slot *free_slot(groot_ctx *c)
{
    if (c->waited) release_slot(c, c->waited);      // one-batch-at-a-time callers never release explicitly
    for (auto &s : c->slots)
        if (s->state == Slot::FREE) return s.get();
    return nullptr;
}

static int install_background(groot_ctx *c, bool wait);

// copy-in, decode, kernels, counter copy-out of slot s: everything asynchronous
static int enqueue(groot_ctx *c, Slot *s)
{
    HIP_TRY(c, hipSetDevice(c->device));
    if (int rc = install_background(c, false)) return rc;
    s->status = GROOT_OK; s->status_msg.clear();
    s->n_trav = 0; s->host_results = false;
    memset(&s->counts, 0, sizeof s->counts);
    memset(&s->ms, 0, sizeof s->ms);
    s->ticket = c->next_ticket++;
    if (s->n_reads == 0) {      // nothing to run: completes at once
        memset(s->h_ctr.p, 0, sizeof(DeviceCounters));
        HIP_TRY(c, hipEventRecord(s->ev_ctr, c->d2h_stream));
        s->state = Slot::IN_FLIGHT;
        c->inflight.push_back(s);
        return GROOT_OK;
    }
    if (s->input != Slot::IN_DEVICE) {
        hipStream_t h = c->h2d_stream;
        // One copy-in at a time: the caller waits here for the copy-in of the batch before.  (Every copy queued on a stream is given an SDMA
        // engine when it is submitted and waits THERE for the copy before it; the copy-out of a finished batch then lands behind such a
        // waiting copy-in and takes 11-15 ms instead of 3.5.  A host-fed stream ran at 1 950 Mreads/s with three batches in flight, 1 600
        // with four, 1 250 with five; with this wait it runs at 1 900-1 950 whatever the depth.  Reading the staging from a kernel instead
        // of the copy engine was tried as well: its 64-byte read requests crowd the link's upstream direction and the copy-out halves.)
        if (c->h2d_last) HIP_TRY(c, hipEventSynchronize(c->h2d_last));
        c->h2d_last = s->ev_h2d;
        if (c->profiling) HIP_TRY(c, hipEventRecord(s->ev_h2d0, h));
        if (s->input == Slot::IN_ASCII) {
            HIP_TRY(c, hipMemcpyAsync(s->d_seq.p, s->h_bases.p, s->n_bases, hipMemcpyHostToDevice, h));
            HIP_TRY(c, hipMemcpyAsync(s->d_off.p, s->h_off.p, ((size_t)s->n_reads + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, h));
        } else {
            HIP_TRY(c, hipMemcpyAsync(s->d_packed.p, s->h_bases.p, (size_t)((s->n_bases + 3) / 4), hipMemcpyHostToDevice, h));
            if (s->input == Slot::IN_PACKED)
                HIP_TRY(c, hipMemcpyAsync(s->d_off.p, s->h_off.p, ((size_t)s->n_reads + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, h));
            else if (!s->uniform_len)
                HIP_TRY(c, hipMemcpyAsync(s->d_len.p, s->h_len.p, (size_t)s->n_reads * sizeof(uint16_t), hipMemcpyHostToDevice, h));
            // A short exception list stays at home: patch_reads_kernel reads it through the pinned mapping.  (Copies of that size are done by
            // blit kernels, and the two of them kept the copy-in stream busy for 0.5 ms between one batch's bases and the next batch's: a
            // tenth of the batch period of a host-fed stream.)
            s->exc_at_home = s->n_exc * 9 <= kExcAtHomeBytes;
            if (s->n_exc && !s->exc_at_home) {
                HIP_TRY(c, hipMemcpyAsync(s->d_exc_pos.p, s->h_exc_pos.p, s->n_exc * sizeof(uint64_t), hipMemcpyHostToDevice, h));
                HIP_TRY(c, hipMemcpyAsync(s->d_exc_byte.p, s->h_exc_byte.p, s->n_exc, hipMemcpyHostToDevice, h));
            }
        }
        HIP_TRY(c, hipEventRecord(s->ev_h2d, h));
        HIP_TRY(c, hipStreamWaitEvent(c->stream, s->ev_h2d, 0));
    }
    if (c->profiling) HIP_TRY(c, hipEventRecord(s->ev[0], c->stream));
    if (s->input == Slot::IN_PACKED || s->input == Slot::IN_PACKED16) {
        const uint64_t n_words = (s->n_bases + 15) / 16;                  // 16 bases per packed word
        if (n_words)
            hipLaunchKernelGGL(unpack_reads_kernel, dim3((unsigned)((n_words + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream, s->d_packed.p,
                               n_words, reinterpret_cast<uint4 *>(s->d_seq.p));
        if (s->n_exc)
            hipLaunchKernelGGL(patch_reads_kernel, dim3((unsigned)((s->n_exc + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream,
                               s->exc_at_home ? s->h_exc_pos.p : s->d_exc_pos.p, s->exc_at_home ? s->h_exc_byte.p : s->d_exc_byte.p, s->n_exc, s->d_seq.p);
        HIP_TRY(c, hipGetLastError());
        if (s->input == Slot::IN_PACKED16 && s->uniform_len) {
            hipLaunchKernelGGL(uniform_offsets_kernel, dim3((s->n_reads + kBlock) / kBlock), dim3(kBlock), 0, c->stream, s->d_off.p, s->n_reads,
                               s->uniform_len);
            HIP_TRY(c, hipGetLastError());
        } else if (s->input == Slot::IN_PACKED16) {
            HIP_TRY(c, hipMemsetAsync(s->d_off.p, 0, sizeof(uint64_t), c->stream));
            auto in = rocprim::make_transform_iterator(s->d_len.p, LenToU64());
            size_t tmp_bytes = 0;
            HIP_TRY(c, rocprim::inclusive_scan(nullptr, tmp_bytes, in, s->d_off.p + 1, s->n_reads, rocprim::plus<uint64_t>(), c->stream));
            if (tmp_bytes > c->in_tmp.n) {
                HIP_TRY(c, hipStreamSynchronize(c->stream));
                HIP_TRY(c, c->in_tmp.alloc(tmp_bytes + tmp_bytes / 4));
            }
            HIP_TRY(c, rocprim::inclusive_scan(c->in_tmp.p, tmp_bytes, in, s->d_off.p + 1, s->n_reads, rocprim::plus<uint64_t>(), c->stream));
        }
    }
    s->set = c->next_set;
    c->next_set ^= 1u;
    if (int rc = run_batch_async(c, s, true)) return rc;
    HIP_TRY(c, hipEventRecord(s->ev_compute, c->astream));
    c->last_compute = s->ev_compute;
    c->newest = s;
    // Copy-out on its own stream with no host in between.  The record count is only known on the device, and asking for it
    // would put a host round trip between the last kernel and the copy; so the copy engine is given a PREDICTED count now
    // -- records per read of the latest finished batch, plus a margin -- and collect fetches the rest in the rare batch
    // that has more.  (A device-driven copy kernel writing straight into pinned host memory gets the exact size too, but
    // its posted PCIe writes back up into the write path the other kernels share: the next batch's first memory-bound
    // kernel stalled until the copy was through.  Measured, dropped.)
    HIP_TRY(c, hipStreamWaitEvent(c->d2h_stream, s->ev_compute, 0));
    if (!c->prm.results_on_device) {
        if (c->profiling) HIP_TRY(c, hipEventRecord(s->ev_d2h0, c->d2h_stream));
        // margin: a few standard deviations of a per-read count over n reads, at least 1 %
        const double margin = 1.0 + std::max(0.01, 4.0 / std::sqrt((double)s->n_reads + 1.0));
        const uint64_t predicted = (uint64_t)((double)s->n_reads * c->trav_per_read * margin) + 1024;
        s->copied = (uint32_t)std::min<uint64_t>(predicted, s->trav_cap);
        const double bpt = c->bytes_per_trav > 0 ? c->bytes_per_trav : 8.0 * (double)c->pw_view;
        s->copied_bytes = std::min<uint64_t>((uint64_t)((double)s->copied * bpt * margin) + 4096, (uint64_t)s->trav_cap * c->pw_view * 8);
        if (c->packed_travs) HIP_TRY(c, hipMemcpyAsync(s->h_ctrav.p, s->d_ctrav.p, (size_t)s->copied * sizeof(groot_ctrav), hipMemcpyDeviceToHost, c->d2h_stream));
        else HIP_TRY(c, hipMemcpyAsync(s->h_trav.p, s->d_trav.p, (size_t)s->copied * sizeof(groot_trav), hipMemcpyDeviceToHost, c->d2h_stream));
        HIP_TRY(c, hipMemcpyAsync(s->h_mask.p, s->d_cmask.p, (size_t)s->copied_bytes, hipMemcpyDeviceToHost, c->d2h_stream));
        HIP_TRY(c, hipMemcpyAsync(s->h_ckpt.p, s->d_ckpt.p, ((size_t)s->copied / 256 + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, c->d2h_stream));
        if (c->profiling) HIP_TRY(c, hipEventRecord(s->ev_d2h, c->d2h_stream));
    }
    HIP_TRY(c, hipMemcpyAsync(s->h_ctr.p, s->d_ctr.p, sizeof(DeviceCounters), hipMemcpyDeviceToHost, c->d2h_stream));
    HIP_TRY(c, hipEventRecord(s->ev_ctr, c->d2h_stream));
    s->state = Slot::IN_FLIGHT;
    c->inflight.push_back(s);
    return GROOT_OK;
}

// 12-byte records (read position | flags, node, offset) -> groot_trav: the graph is the node's, ord counts the records of a read
static void expand_travs(const groot_ctx *c, Slot *s)
{
    const size_t n = s->n_trav;
    const groot_ctrav *in = s->h_ctrav.p;
    groot_trav *out = s->h_trav.p;
    const uint32_t first = s->first_read_id;
    const uint32_t *node_graph = c->h_node_graph.data();
    auto span = [=](size_t lo, size_t hi) {
        if (lo >= hi) return;
        uint32_t ord = 0, prev = ~0u;
        if (lo) {                                          // records of the same read before this span
            prev = in[lo - 1].read_flags & 0x00FFFFFFu;
            if ((in[lo].read_flags & 0x00FFFFFFu) == prev) {
                size_t j = lo;
                while (j > 0 && (in[j - 1].read_flags & 0x00FFFFFFu) == prev) j--;
                ord = (uint32_t)(lo - j) - 1;              // ord of the record at lo - 1
            }
        }
        for (size_t i = lo; i < hi; i++) {
            const uint32_t pos = in[i].read_flags & 0x00FFFFFFu;
            ord = pos == prev ? ord + 1 : 0;
            prev = pos;
            groot_trav t;
            t.read_id = first + pos; t.graph_id = node_graph[in[i].node]; t.node = in[i].node; t.offset = in[i].offset;
            t.ord = (uint16_t)ord; t.flags = (uint8_t)(in[i].read_flags >> 24); t.reserved = 0;
            out[i] = t;
        }
    };
    const unsigned nt = (unsigned)std::min<size_t>(std::min(16u, granted_cpus()), n / (1u << 18));   // (memory bound: 2-3 ms per 10 M records)
    if (nt <= 1) { span(0, n); return; }
    std::vector<std::thread> th;
    const size_t per = (n + nt - 1) / nt;
    for (unsigned t = 0; t < nt; t++) th.emplace_back(span, std::min(n, t * per), std::min(n, (t + 1) * per));
    for (auto &x : th) x.join();
}

// The counters of slot s have arrived: grow-and-redo on overflow, then start the copy-out of its traversal records.
static int finish_counters(groot_ctx *c, Slot *s)
{
    DeviceCounters &h = *s->h_ctr.p;
    auto refetch = [&](DeviceCounters &dst) -> int {
        HIP_TRY(c, hipMemcpyAsync(s->h_ctr.p, s->d_ctr.p, sizeof(DeviceCounters), hipMemcpyDeviceToHost, c->astream));
        HIP_TRY(c, hipStreamSynchronize(c->astream));
        dst = *s->h_ctr.p;
        return GROOT_OK;
    };
    DeviceCounters first = h;
    bool have_first = false;              // weights + read counters already taken from an earlier pass
    bool redone = false;
    for (int attempt = 0; s->n_reads; attempt++) {
        const uint32_t fl = h.flags;
        if (!(fl & (kFlagSeedOverflow | kFlagQOverflow | kFlagOvfOverflow | kFlagTravOverflow))) break;
        if (attempt > 8) return fail(c, GROOT_E_NOSPACE, "output buffers keep overflowing (flags=0x%x)", fl);
        // Later batches may already have run through the shared work buffers: let them finish, grow, and redo this
        // batch as a whole.  A pass whose align stage did nothing (seed slots / table rows ran out) is simply repeated;
        // after any other overflow the weights and read counters of the first pass stand and only records are re-made.
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->astream));
        bool redo_weights = !have_first;
        if (fl & (kFlagSeedOverflow | kFlagQOverflow)) {
            if (fl & kFlagSeedOverflow) { if (int rc = alloc_seed_slots(c, h.max_seeds + 4)) return rc; }
            if (fl & kFlagQOverflow) {
                if (c->att_external) {
                    s->status = GROOT_E_NOSPACE;
                    s->status_msg = "a kmerCount outside the fixed layout of groot_hip_attempts_layout occurred";
                    h.n_trav = 0;
                    break;
                }
                if (int rc = grow_attempts(c, std::max(h.q_rows, c->att_cap * 2))) return rc;
            }
        } else {
            if (!have_first) { first = h; have_first = true; }
            redo_weights = false;
            if (fl & kFlagOvfOverflow) { if (int rc = alloc_ovf(c, c->ovf_cap * 4)) return rc; }
            if (fl & kFlagTravOverflow) { if (int rc = alloc_trav(c, s, h.n_trav + h.n_trav / 8 + 1024)) return rc; }
        }
        if (int rc = run_batch_async(c, s, redo_weights)) return rc;
        redone = true;
        DeviceCounters again{};
        if (int rc = refetch(again)) return rc;
        if (have_first) {
            const uint32_t keep = first.flags & ~(kFlagOvfOverflow | kFlagTravOverflow | kFlagSeedOverflow | kFlagQOverflow);
            DeviceCounters merged = first;
            merged.n_trav = again.n_trav; merged.alignments = again.alignments; merged.seeds = again.seeds; merged.max_seeds = again.max_seeds;
            merged.flags = keep | again.flags;
            merged.q_rows = again.q_rows;
            merged.mask_words = again.mask_words;
            merged.todo_reads = again.todo_reads; merged.tab_reads = again.tab_reads; merged.lean_reads = again.lean_reads;
            h = merged;
        } else h = again;
    }
    s->n_trav = s->n_reads ? h.n_trav : 0;
    groot_counts &o = s->counts;
    o.received = s->n_reads;              // boss.go:194 receivedReads++ for every read
    o.mapped = h.mapped; o.multimapped = h.multimapped; o.alignments = h.alignments; o.seeds = h.seeds;
    o.travs = s->n_trav; o.revcomp_panics = h.revcomp_panics; o.short_reads = h.short_reads;
    o.full_sketch_reads = (s->sig_used || s->text_used) ? h.todo_reads : s->n_reads;
    o.walked_reads = h.seeded_reads;
    o.lean_reads = h.lean_reads;
    if (s->status == GROOT_OK) {
        char buf[256];
        if (h.flags & kFlagLongRead) { s->status = GROOT_E_NOSPACE; snprintf(buf, sizeof buf, "a read is longer than max_read_len=%u", c->prm.max_read_len); s->status_msg = buf; }
        else if (h.flags & kFlagOrdOverflow) { s->status = GROOT_E_NOSPACE; s->status_msg = "a read produced more than 65535 traversals"; }
        else if (h.flags & kFlagShortRead) {
            s->status = GROOT_E_SHORT_READ;
            snprintf(buf, sizeof buf, "k size is greater than sequence length for %llu read(s) (the reference panics: boss.go:164-166)", h.short_reads);
            s->status_msg = buf;
        } else if (h.revcomp_panics) {
            s->status = GROOT_E_REVCOMP;
            snprintf(buf, sizeof buf, "%llu read(s) hold a byte > 'T' and reached RevComplement (the reference panics: seqio.go:126)", h.revcomp_panics);
            s->status_msg = buf;
        }
    }
#if defined(GROOT_WORK_COUNTERS) && GROOT_WORK_COUNTERS == 3
    if (h.dbg[4]) {
        fprintf(stderr, "[groot timeline] wavefronts with work %llu, rounds %llu, iterations: mean %.0f max %llu; wave duration: mean %.3f ms max %.3f ms\n", h.dbg[4], h.dbg[5],
                (double)h.dbg[0] / (double)h.dbg[4], h.dbg[1], (double)h.dbg[2] / (double)h.dbg[4] / 1e5, (double)h.dbg[3] / 1e5);
        for (unsigned long long i = 0; i < 14 && i < h.dbg[17]; i++)
            fprintf(stderr, "[groot timeline] late read %llu%s: %llu windows, %llu graphs, %llu traversals, from %llu to %llu us, %llu wave iterations, %llu steps of its own\n", h.dbg[18 + 3 * i] & 0xFFFFFFFFull,
                    (h.dbg[18 + 3 * i] >> 63) ? " (item)" : "", (h.dbg[18 + 3 * i] >> 32) & 0x7FFFFFFFull, h.dbg[20 + 3 * i] >> 56, (h.dbg[20 + 3 * i] >> 48) & 0xFFull, h.dbg[19 + 3 * i] & 0xFFFFFFFFull, h.dbg[19 + 3 * i] >> 32,
                    h.dbg[20 + 3 * i] & 0xFFFFFFFFull, (h.dbg[20 + 3 * i] >> 32) & 0xFFFFull);
        fprintf(stderr, "[groot timeline] late reads in all: %llu\n", h.dbg[17]);
        fprintf(stderr, "[groot timeline] ms summed over wavefronts: all %.1f = cooperative scans %.1f (%llu services) + fork/join %.1f + FETCH %.1f (%llu steps) + SCAN %.1f (%llu) + DFS %.1f (%llu) + rest\n",
                (double)h.dbg[2] / 1e5, (double)h.dbg[8] / 1e5, h.dbg[16], (double)h.dbg[9] / 1e5, (double)h.dbg[10] / 1e5, h.dbg[13], (double)h.dbg[11] / 1e5, h.dbg[14], (double)h.dbg[12] / 1e5, h.dbg[15]);
        for (int hh = 0; hh < 2; hh++) {
            fprintf(stderr, "[groot timeline] %s (buckets of 50 us):", hh ? "length of a wavefront's last round" : "wavefront ends after");
            for (int b = 0; b < 64; b++) fprintf(stderr, " %llu", h.dbg[64 + 64 * hh + b]);
            fprintf(stderr, "\n");
        }
    }
#elif defined(GROOT_WORK_COUNTERS) && GROOT_WORK_COUNTERS == 4
    if (h.dbg[151]) {
        const double nw = (double)h.dbg[151];
        fprintf(stderr, "[groot lean] wavefronts %llu; per wavefront: %.1f iterations (with a level-1 / level-2 / level-3-4 / walk lane: %.1f / %.1f / %.1f / %.1f); lane-steps per wavefront: %.0f / %.0f / %.0f / %.0f; staging %.2f us, loop %.2f us\n",
                h.dbg[151], (double)h.dbg[148] / nw, (double)h.dbg[140] / nw, (double)h.dbg[141] / nw, (double)h.dbg[142] / nw, (double)h.dbg[143] / nw,
                (double)h.dbg[144] / nw, (double)h.dbg[145] / nw, (double)h.dbg[146] / nw, (double)h.dbg[147] / nw, (double)h.dbg[149] / nw / 100.0, (double)h.dbg[150] / nw / 100.0);
        fprintf(stderr, "[groot lean] finished without an alignment %llu; left to align_kernel: seeds > 4: %llu, byte > T / length: %llu, byte other than ACGT: %llu, window with an N: %llu, node with an N: %llu, N ahead: %llu, three neighbours: %llu, third pending: %llu, too many steps: %llu\n",
                h.dbg[160], h.dbg[161], h.dbg[163], h.dbg[164], h.dbg[165], h.dbg[166], h.dbg[167], h.dbg[168], h.dbg[169], h.dbg[170]);
    }
#elif defined(GROOT_WORK_COUNTERS)
    for (int e = 0; e < 32; e++)
        if (h.dbg[e]) fprintf(stderr, "[groot work] event %2d: wave iterations %llu lanes %llu\n", e, h.dbg[e], h.dbg[32 + e]);
    fprintf(stderr, "[groot work] longest round: %llu wave iterations\n", h.dbg[63]);
#if GROOT_WORK_COUNTERS == 2
    fprintf(stderr, "[groot work] slow reads (%llu):", h.dbg[128]);
    for (int i = 0; i < 30 && (unsigned long long)i < h.dbg[128]; i++)
        fprintf(stderr, " %llu:%llu:%llu/%llu/%llu", h.dbg[129 + 2 * i] & 0xFFFFFFFFull, h.dbg[129 + 2 * i] >> 32, h.dbg[130 + 2 * i] & 0xFFFFFull,
                (h.dbg[130 + 2 * i] >> 20) & 0xFFFFFull, h.dbg[130 + 2 * i] >> 40);
    fprintf(stderr, "\n");
#endif
    for (int i = 0; i < 5; i++) fprintf(stderr, "[groot work] FETCH part %d: %.1f ms summed over wavefronts\n", i, (double)h.dbg[40 + i] / 1e5);
    for (int ph = 0; ph < 3; ph++)
        fprintf(stderr, "[groot work] phase %d: %llu steps, %.2f us per step (wall clock, per wave)\n", ph, h.dbg[27 + ph],
                h.dbg[27 + ph] ? (double)h.dbg[24 + ph] / 100.0 / (double)h.dbg[27 + ph] : 0.0);
    for (int hh = 0; hh < 2; hh++) {
        fprintf(stderr, "[groot work] %s (buckets of 2 iterations):", hh ? "round length" : "lane finish");
        for (int b = 0; b < 64; b++) fprintf(stderr, " %llu", h.dbg[64 + 64 * hh + b]);
        fprintf(stderr, "\n");
    }
#endif
    // the records were copied out by copy_out_kernel right behind the kernels; after a redo they are fetched again here
    if (s->n_reads) c->trav_per_read = (double)s->n_trav / (double)s->n_reads;
    if (s->n_reads && !c->tab_capture) c->dfs_frac = (double)h.seeded_reads / (double)s->n_reads;
    if (s->n_reads && !c->tab_capture && s->lean_used) c->lean_left_frac = (double)(h.seeded_reads - std::min(h.seeded_reads, h.lean_reads)) / (double)s->n_reads;
    if (s->n_reads && (s->sig_used || s->text_used)) c->todo_frac = (double)h.todo_reads / (double)s->n_reads;
    if (s->n_reads && !c->tab_capture && c->dix.text_tab) {
        c->text_hit_frac = s->text_used ? 1.0 - (double)h.todo_reads / (double)s->n_reads : (double)h.tab_reads / (double)s->n_reads;
        // (a batch that tried the lookup in vain sent all its reads through the list pass: on a stream the memo cannot answer --
        // mixed read lengths, another organism -- the next try comes later and later)
        if (s->text_used) c->text_retry_gap = c->text_hit_frac >= 0.7 ? 8u : std::min(256u, c->text_retry_gap * 2u);
    }
    s->n_mask_bytes = s->n_trav ? h.mask_words : 0;
    if (!c->prm.results_on_device && s->n_trav) {
        c->bytes_per_trav = (double)s->n_mask_bytes / (double)s->n_trav;
        const uint32_t have = redone ? 0 : std::min(s->copied, s->n_trav);      // a redo re-made the records: fetch them all
        if (have < s->n_trav) {
            if (c->packed_travs) HIP_TRY(c, hipMemcpy(s->h_ctrav.p + have, s->d_ctrav.p + have, (size_t)(s->n_trav - have) * sizeof(groot_ctrav), hipMemcpyDeviceToHost));
            else HIP_TRY(c, hipMemcpy(s->h_trav.p + have, s->d_trav.p + have, (size_t)(s->n_trav - have) * sizeof(groot_trav), hipMemcpyDeviceToHost));
            HIP_TRY(c, hipMemcpy(s->h_ckpt.p, s->d_ckpt.p, ((size_t)s->n_trav / 256 + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost));
        }
        const uint64_t have_w = redone ? 0 : std::min<uint64_t>(s->copied_bytes, s->n_mask_bytes);
        if (have_w < s->n_mask_bytes)
            HIP_TRY(c, hipMemcpy(s->h_mask.p + have_w, s->d_cmask.p + have_w, (size_t)(s->n_mask_bytes - have_w), hipMemcpyDeviceToHost));
        if (c->packed_travs) expand_travs(c, s);
        s->host_results = true;
    }
    s->state = Slot::D2H_ISSUED;
    return GROOT_OK;
}

// Move batches whose counters have arrived on to their copy-out, in submission order: blocking up to and including
// `must`, opportunistically (event query) beyond it or when must is null.
static int progress(groot_ctx *c, Slot *must = nullptr)
{
    bool blocking = must != nullptr;
    for (Slot *s : c->inflight) {
        if (s->state == Slot::IN_FLIGHT) {
            if (blocking) HIP_TRY(c, hipEventSynchronize(s->ev_ctr));
            else {
                const hipError_t q = hipEventQuery(s->ev_ctr);
                if (q == hipErrorNotReady) break;
                if (q != hipSuccess) return fail(c, GROOT_E_DEVICE, "hipEventQuery: %s", hipGetErrorString(q));
            }
            if (int rc = finish_counters(c, s)) return rc;
        }
        if (s == must) blocking = false;
    }
    return GROOT_OK;
}

static int collect_impl(groot_ctx *c, Slot **out)
{
    if (c->inflight.empty()) return fail(c, GROOT_E_STATE, "no batch submitted");
    HIP_TRY(c, hipSetDevice(c->device));
    Slot *s = c->inflight.front();
    if (s->state == Slot::IN_FLIGHT) {       // waits for the batch's counters, which travel behind its records
        if (int rc = progress(c, s)) return rc;
    }
    if (c->profiling && s->n_reads) {
        if (s->input != Slot::IN_DEVICE) (void)hipEventElapsedTime(&s->ms.h2d, s->ev_h2d0, s->ev_h2d);
        (void)hipEventElapsedTime(&s->ms.unpack, s->ev[0], s->ev[1]);
        (void)hipEventElapsedTime(&s->ms.sketch_seed, s->ev[1], s->ev[2]);
        (void)hipEventElapsedTime(&s->ms.schedule, s->ev[2], s->ev[3]);
        (void)hipEventElapsedTime(&s->ms.align, s->ev[11], s->ev[4]);
        (void)hipEventElapsedTime(&s->ms.sort, s->ev[4], s->ev[5]);
        (void)hipEventElapsedTime(&s->ms.total, s->ev[0], s->ev[5]);
        (void)hipEventElapsedTime(&s->ms.first_seed_kernel, s->ev[7], s->ev[8]);
        (void)hipEventElapsedTime(&s->ms.order_kernel, s->ev[9], s->ev[10]);
        (void)hipEventElapsedTime(&s->ms.list_pass, s->ev[8], s->ev[12]);
        s->ms.lean_pass = 0;
        if (s->lean_used) (void)hipEventElapsedTime(&s->ms.lean_pass, s->ev[11], s->ev[13]);
        (void)hipEventElapsedTime(&s->ms.wall, s->ev[1], s->ev[5]);
        if (!c->prm.results_on_device) (void)hipEventElapsedTime(&s->ms.d2h, s->ev_d2h0, s->ev_d2h);
    }
    c->inflight.pop_front();
    s->state = Slot::COLLECTED;
    *out = s;
    (void)progress(c);       // keep the copy-outs of the batches behind it going
    return GROOT_OK;
}

static int drain(groot_ctx *c)     // everything submitted has finished on the device (results stay collectable)
{
    HIP_TRY(c, hipSetDevice(c->device));
    if (!c->inflight.empty()) { if (int rc = progress(c, c->inflight.back())) return rc; }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->astream));
    HIP_TRY(c, hipStreamSynchronize(c->d2h_stream));
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
    p->pipeline_depth = 3;
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
    ctx->bg_cancel = true;
    if (ctx->bg.joinable()) ctx->bg.join();
    if (ctx->bg_stream) { (void)hipStreamSynchronize(ctx->bg_stream); (void)hipStreamDestroy(ctx->bg_stream); }
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->astream) (void)hipStreamSynchronize(ctx->astream);
    if (ctx->h2d_stream) (void)hipStreamSynchronize(ctx->h2d_stream);
    if (ctx->d2h_stream) (void)hipStreamSynchronize(ctx->d2h_stream);
    for (auto &s : ctx->slots) {
        for (hipEvent_t e : {s->ev_seed, s->ev_h2d0, s->ev_h2d, s->ev_compute, s->ev_ctr, s->ev_d2h0, s->ev_d2h})
            if (e) (void)hipEventDestroy(e);
        for (auto &e : s->ev)
            if (e) (void)hipEventDestroy(e);
    }
    ctx->slots.clear();
    for (WorkSet &w : ctx->ws)
        if (w.ev_free) (void)hipEventDestroy(w.ev_free);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    if (ctx->astream) (void)hipStreamDestroy(ctx->astream);
    if (ctx->h2d_stream) (void)hipStreamDestroy(ctx->h2d_stream);
    if (ctx->d2h_stream) (void)hipStreamDestroy(ctx->d2h_stream);
    delete ctx;
}

// ---------------------------------------------------------------------------------------------
// sketch_sig_kernel's side of the index: window texts, proven against Key.Sketch, and the signature table
// ---------------------------------------------------------------------------------------------
// n WindowSize-mers of window texts (concatenated; owner[j] = their window) through the full-width kernel twice on one upload:
// sketches (compared with Key.Sketch on the device: differs[j]) and the whole seed stage (per read the record's cnt_flags word and the scheduling key without
// span bits: first seed window << 2 | dead-orientation class)
static int text_pass(groot_ctx *c, const uint8_t *seqs, const uint32_t *owner, uint32_t n, uint32_t len, uint8_t *differs, uint32_t *cnt_flags,
                     uint32_t *keys)
{
    DevBuf<uint64_t> off, sk;
    DevBuf<uint8_t> seq, bad;
    DevBuf<DeviceCounters> ctr;
    DevBuf<uint32_t> cnt, win, key, own;
    DevBuf<ReadRec> rec;
    const uint64_t total = (uint64_t)n * len;
    HIP_TRY(c, off.alloc((size_t)n + 1));
    HIP_TRY(c, sk.alloc((size_t)n * c->s));
    HIP_TRY(c, seq.alloc(total + 64));
    HIP_TRY(c, bad.alloc(n));
    HIP_TRY(c, ctr.alloc(1));
    HIP_TRY(c, cnt.alloc(n));
    HIP_TRY(c, own.alloc(n));
    // (one read of the ctx's seed slots: on the background builder's thread the caller's thread may grow them meanwhile -- finish_counters)
    const uint32_t seed_slots = tl_background ? c->bg_seed_slots : c->seed_slots;
    HIP_TRY(c, win.alloc((size_t)seed_slots * n));
    HIP_TRY(c, key.alloc(n));
    HIP_TRY(c, rec.alloc(n));
    HIP_TRY(c, hipMemcpyAsync(seq.p, seqs, total, hipMemcpyHostToDevice, c->build_stream));
    HIP_TRY(c, hipMemcpyAsync(own.p, owner, (size_t)n * 4, hipMemcpyHostToDevice, c->build_stream));
    HIP_TRY(c, hipMemsetAsync(ctr.p, 0, sizeof(DeviceCounters), c->build_stream));
    const dim3 grid((n + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(uniform_offsets_kernel, dim3(n / kBlock + 1), dim3(kBlock), 0, c->build_stream, off.p, n, len);
    SeedArgs a{};
    a.ix = *c->build_dix;
    a.seq = seq.p; a.seq_off = off.p; a.n_reads = n; a.max_read_len = std::max(len, tl_background ? c->bg_max_read_len : c->prm.max_read_len);
    a.lds_read_bytes = (uint32_t)std::min<uint64_t>((uint64_t)kBlock * len + 32, kMaxLdsReadBytes);
    a.seed_slots = seed_slots; a.seed_count = cnt.p; a.seed_win = win.p;
    a.ctr = ctr.p; a.shards = c->build_shards;
    const size_t lds = kLdsReads + ((a.lds_read_bytes + 15) & ~15u);
    SeedArgs d = a;                                         // sketches only: no lookup
    d.ix.max_q = 0; d.sketch_out = sk.p;
    launch_seed(c->s, c->max_k, d, true, grid, lds, c->build_stream);
    hipLaunchKernelGGL(sketch_equal_kernel, grid, dim3(kBlock), 0, c->build_stream, sk.p, own.p, c->win_sketch.p, c->s, n, bad.p);
    a.sort_key = key.p; a.sort_span_bits = 0; a.read_rec = rec.p;
    launch_seed(c->s, c->max_k, a, false, grid, lds, c->build_stream);
    HIP_TRY(c, hipGetLastError());
    std::vector<ReadRec> h(n);
    HIP_TRY(c, hipMemcpyAsync(h.data(), rec.p, (size_t)n * sizeof(ReadRec), hipMemcpyDeviceToHost, c->build_stream));
    HIP_TRY(c, hipMemcpyAsync(keys, key.p, (size_t)n * 4, hipMemcpyDeviceToHost, c->build_stream));
    HIP_TRY(c, hipMemcpyAsync(differs, bad.p, n, hipMemcpyDeviceToHost, c->build_stream));
    HIP_TRY(c, hipStreamSynchronize(c->build_stream));
    HIP_TRY(c, hipMemsetAsync(c->build_shards, 0, (size_t)kSeedShards * kSeedShardStride * sizeof(unsigned long long), c->build_stream));
    HIP_TRY(c, hipStreamSynchronize(c->build_stream));   // (nobody folds them here)
    for (uint32_t i = 0; i < n; i++) cnt_flags[i] = h[i].cnt_flags;
    return GROOT_OK;
}

static int enqueue(groot_ctx *c, Slot *s);
static int collect_impl(groot_ctx *c, Slot **out);
static int take_slot(groot_ctx *c, uint32_t n_reads, Slot **out);

// Outcome table (DeviceIndex::out_tab, device_types.hpp OutEntry) and text table (DeviceIndex::text_tab).
// What this ctx does with a read -- which windows ContainmentIndex.Query returns, which of them get IncrementSubPath, which
// traversals AlignRead reports, in which order (lshe.go:153-175, graphminion.go:46-102, alignment.go:13-159) -- is a function of
// the read's bases alone.  So the strings real reads are most likely to BE, every WindowSize-mer of every indexed sequence path on
// both strands, go through the ctx's own pipeline once, as ordinary batches (signature kernel -> full-width kernel -> sort ->
// align_kernel -> ordering), the align stage additionally noting the windows it counted, and every string with 1..16 traversals gets
// its records stored: a memo of the pipeline's own results, nothing else.  At run time a read that equals such a string is
// answered by text_lookup_kernel (keyed by the bases) or by the signature kernel (sig_info of the window-text strings, which are
// path strings too) and never reaches the align stage.
namespace {
struct StringSet {                          // distinct strings at 2 bits per base, tw dwords each; open addressing over their hashes
    uint32_t tw = 0;
    std::vector<uint32_t> words;            // [n * tw]
    std::vector<uint32_t> slots;            // index + 1, 0 = free
    size_t n = 0;
    uint32_t mask = 0;
    void init(uint32_t tw_, size_t expect)
    {
        tw = tw_;
        uint32_t cap = 1024;
        while (cap < 2 * expect) cap <<= 1;
        slots.assign(cap, 0);
        mask = cap - 1;
        words.reserve(expect * tw);
    }
    static uint64_t hash(const uint32_t *w, uint32_t tw)
    {
        uint64_t h = GROOT_TEXT_HASH_INIT;
        for (uint32_t j = 0; j < tw; j++) h = text_hash_step(h, w[j]);
        return h;
    }
    // index of the string, inserting it if `insert`; -1 if absent
    long find(const uint32_t *w, bool insert)
    {
        const uint64_t h = hash(w, tw);
        for (uint32_t s = (uint32_t)(h ^ (h >> 32)) & mask;; s = (s + 1) & mask) {
            if (!slots[s]) {
                if (!insert) return -1;
                words.insert(words.end(), w, w + tw);
                slots[s] = (uint32_t)++n;
                return (long)n - 1;
            }
            if (!memcmp(&words[(size_t)(slots[s] - 1) * tw], w, (size_t)tw * 4)) return (long)slots[s] - 1;
        }
    }
};
// bases [i, i + len) of a sequence packed at 2 bits per base (16 per dword, trailing dwords zero-padded)
inline void pack_at(const std::vector<uint32_t> &packed, size_t i, uint32_t len, uint32_t tw, uint32_t *out)
{
    const size_t d = i >> 4;
    const uint32_t sh = 2 * (uint32_t)(i & 15);
    for (uint32_t j = 0; j < tw; j++) {
        const uint64_t two = (uint64_t)packed[d + j] | ((uint64_t)packed[d + j + 1] << 32);
        out[j] = (uint32_t)(two >> sh);
    }
    const uint32_t full = len >> 4, tail = len & 15;
    if (full < tw) out[full] &= tail ? (1u << (2 * tail)) - 1u : 0u;
    for (uint32_t j = full + 1; j < tw; j++) out[j] = 0;
}
} // namespace

static int build_outcome_table(groot_ctx *c, const groot_index_view *v, const std::vector<uint8_t> &text, const std::vector<uint32_t> &tlen,
                               std::vector<uint32_t> &info, uint32_t w, uint32_t vstride)
{
    const bool stats = c->kn.open_stats;
    auto t_lap = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!stats) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[groot open]     memo: %-22s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_lap).count());
        t_lap = now;
    };
    const uint32_t n = c->n_windows, pw = c->pw_view;
    const uint32_t sq = out_stride_q(pw);
    const uint32_t tw = (w + 15) / 16;
    const uint32_t chunk = (uint32_t)std::min<uint64_t>(std::min<uint64_t>(c->prm.max_batch_reads, 1u << 20), c->prm.max_batch_bases / w);
    if (!chunk) return GROOT_OK;
    // ---- 1. the strings: every WindowSize-mer of every path, both strands, each once ----
    // (the text table serves reads of exactly WindowSize bases whose kmerCount puts Query on the every-slot-equal branch)
    const uint32_t q_w = w - c->k + 1;
    const bool text_ok = w <= 224 && q_w < c->h_q_min_eq.size() && !c->kn.no_text_table;
    // strings with a few bytes other than ACGT (a path through an N): bases with code 0 at those positions, then the bytes and their
    // positions as the text table keeps them (device_types.hpp text_exc_dwords) -- only the text lookup can find these
    const uint32_t twk = text_key_dwords(tw);               // dwords of bases in a text-table entry (zero-padded)
    const uint32_t xw = text_ok ? text_exc_dwords(twk) : 0;
    StringSet set, xset;
    {
        uint64_t expect = 0;
        for (uint32_t p = 0; p < v->n_paths; p++) expect += v->path_len[p] >= w ? 2 * (uint64_t)(v->path_len[p] - w + 1) : 0;
        // The memo's budget (groot_params.memo_budget_mb): per string at most one outcome entry (16 * stride bytes; strings with
        // several traversals are few), one text-table entry at load factor 1/2 (128 bytes), and -- while it is built -- the string
        // set on the host (4 tw + 8 bytes).  An index whose path strings need more is opened without the memo.
        const uint64_t budget = (uint64_t)(c->prm.memo_budget_mb ? c->prm.memo_budget_mb : GROOT_MEMO_DEFAULT_MB) << 20;
        const uint64_t need = expect * ((uint64_t)sq * 16 + 128 + 4 * tw + 8);
        if (need > budget || expect >= (1ull << 30)) {
            if (stats) fprintf(stderr, "[groot open]     memo: skipped, %llu path strings need about %llu MiB (budget %llu MiB)\n", (unsigned long long)expect,
                               (unsigned long long)(need >> 20), (unsigned long long)(budget >> 20));
            return GROOT_OK;
        }
        set.init(tw, (size_t)expect);
        xset.init(twk + xw, 4096);
        std::vector<uint8_t> seq, strand[2];
        std::vector<uint32_t> pk[2];
        std::vector<uint32_t> bad_before[2];                // number of bytes other than ACGT before position i
        std::vector<uint32_t> high_before[2];               // ... of bytes above 'T': RevComplement panics on such a read (seqio.go:126) -- a string holding
                                                            // one stays out of the memo (its batch status would drop the whole capture chunk with it)
        uint32_t buf[16 + 4];
        for (uint32_t g = 0; g < v->n_graphs; g++)
            for (uint32_t lp = 0; lp < v->graph_path_off[g + 1] - v->graph_path_off[g]; lp++) {
                seq.clear();
                for (uint32_t node = v->graph_node_off[g]; node < v->graph_node_off[g + 1]; node++) {   // a path visits its nodes in ascending order (graph.go:243-262)
                    if (!((v->node_mask[(size_t)node * v->path_words + (lp >> 6)] >> (lp & 63)) & 1ULL)) continue;
                    seq.insert(seq.end(), v->bases + v->node_seq_off[node], v->bases + v->node_seq_off[node + 1]);
                }
                const size_t L = seq.size();
                if (L < w) continue;
                for (int st = 0; st < 2; st++) {
                    pk[st].assign(L / 16 + tw + 3, 0);
                    bad_before[st].assign(L + 1, 0);
                    high_before[st].assign(L + 1, 0);
                    strand[st].resize(L);
                    for (size_t i = 0; i < L; i++) {
                        uint8_t b = st ? seq[L - 1 - i] : seq[i];       // (reverse strand: ACGT complemented, any other byte as it is)
                        const bool acgt = b == 'A' || b == 'C' || b == 'G' || b == 'T';
                        if (st && acgt) b = b == 'A' ? 'T' : b == 'C' ? 'G' : b == 'G' ? 'C' : 'A';
                        strand[st][i] = b;
                        bad_before[st][i + 1] = bad_before[st][i] + (acgt ? 0 : 1);
                        high_before[st][i + 1] = high_before[st][i] + (b > 'T' ? 1 : 0);
                        if (acgt) pk[st][i >> 4] |= (uint32_t)((b >> 1) & 3u) << (2 * (i & 15));
                    }
                    for (size_t i = 0; i + w <= L; i++) {
                        const uint32_t nb = bad_before[st][i + w] - bad_before[st][i];
                        if (nb > 2 * xw || high_before[st][i + w] != high_before[st][i]) continue;
                        pack_at(pk[st], i, w, tw, buf);
                        if (!nb) { (void)set.find(buf, true); continue; }
                        for (uint32_t x = tw; x < twk + xw; x++) buf[x] = 0;
                        for (uint32_t x = 0, np = 0; x < w; x++)
                            if (bad_before[st][i + x + 1] != bad_before[st][i + x]) {
                                buf[twk + (np >> 1)] |= (((x + 1) << 8) | strand[st][i + x]) << (16 * (np & 1));
                                np++;
                            }
                        (void)xset.find(buf, true);
                    }
                }
            }
    }
    lap("path strings");
    const size_t NS = set.n, NX = xset.n, NT = NS + NX;     // string ids: the ACGT strings, then the ones with exceptions
    if (!NT) return GROOT_OK;
    // ---- 2. the pipeline, once per string ----
    DevBuf<uint8_t> d_seq;
    DevBuf<uint64_t> d_off;
    HIP_TRY(c, d_seq.alloc((size_t)chunk * w + 64));
    HIP_TRY(c, d_off.alloc((size_t)chunk + 1));
    std::vector<uint32_t> tab;                              // entries, sq * 4 dwords each
    std::vector<uint32_t> sinfo(NT, 0);                     // sig_info word per string (0 = not tabulated)
    std::vector<uint8_t> in_text(NT, 0);                    // ... and it may go into the text table
    std::vector<uint8_t> seqs((size_t)chunk * w);
    std::vector<uint64_t> offs((size_t)chunk + 1);
    for (uint32_t i = 0; i <= chunk; i++) offs[i] = (uint64_t)i * w;
    HIP_TRY(c, hipMemcpy(d_off.p, offs.data(), ((size_t)chunk + 1) * sizeof(uint64_t), hipMemcpyHostToDevice));
    std::vector<groot_trav> travs;
    std::vector<uint64_t> masks;
    std::vector<uint32_t> icnt, iwin, nseeds, seedw;
    std::vector<size_t> big;                                // strings with more calls / seeds than the first pass keeps: second pass
    c->out_strings = NT; c->out_tabulated = c->out_entries = 0;
    int rc_all = GROOT_OK;
    c->tab_capture = true;
    static const char kBase[4] = {'A', 'C', 'T', 'G'};
    // one batch: strings ids[0..m) through the pipeline; incr_cap call-count windows and up to seed_rows seed windows kept per string
    auto run = [&](const size_t *ids, uint32_t m, uint32_t incr_cap, bool second_pass) -> int {
        {
            const unsigned nt = (unsigned)std::max<size_t>(1, std::min<size_t>(std::min(32u, granted_cpus()), m / 4096));
            std::vector<std::thread> th;
            for (unsigned t = 0; t < nt; t++) th.emplace_back([&, t]() {
                for (uint32_t j = (uint32_t)((uint64_t)m * t / nt); j < (uint32_t)((uint64_t)m * (t + 1) / nt); j++) {
                    const uint32_t *pwd = ids[j] < NS ? &set.words[ids[j] * tw] : &xset.words[(ids[j] - NS) * (twk + xw)];
                    uint8_t *dst = &seqs[(size_t)j * w];
                    for (uint32_t x = 0; x < w; x++) dst[x] = (uint8_t)kBase[(pwd[x >> 4] >> (2 * (x & 15))) & 3u];
                    if (ids[j] >= NS)
                        for (uint32_t np = 0; np < 2 * xw; np++) {
                            const uint32_t pair = (pwd[twk + (np >> 1)] >> (16 * (np & 1))) & 0xFFFFu;
                            if (pair) dst[(pair >> 8) - 1] = (uint8_t)pair;
                        }
                }
            });
            for (auto &x : th) x.join();
        }
        c->incr_cap = incr_cap;
        HIP_TRY(c, c->incr_cnt.reserve(m));
        HIP_TRY(c, c->incr_win.reserve((size_t)m * incr_cap));
        Slot *s = nullptr;
        if (int rc = take_slot(c, m, &s)) return rc;
        if (int rc = ensure_slot(c, s, Slot::IN_DEVICE, 0)) return rc;      // (resident input: no staging is allocated for this)
        HIP_TRY(c, hipMemcpy(d_seq.p, seqs.data(), (size_t)m * w, hipMemcpyHostToDevice));
        s->input = Slot::IN_DEVICE; s->n_reads = m; s->first_read_id = 0; s->mixed_len = false; s->one_len = true;
        s->n_bases = 0; s->n_exc = 0; s->max_len = w; s->uniform_len = 0;
        s->ext_seq = d_seq.p; s->ext_off = d_off.p;
        if (int rc = enqueue(c, s)) return rc;
        Slot *done = nullptr;
        if (int rc = collect_impl(c, &done)) return rc;
        const bool ok = done->status == GROOT_OK;
        const uint32_t nt = done->n_trav;
        const uint32_t seed_rows = second_pass ? c->seed_slots : std::min<uint32_t>(4 * kOutSeedDw, c->seed_slots);
        travs.resize(nt); masks.resize((size_t)nt * pw); icnt.resize(m); iwin.resize((size_t)m * incr_cap);
        nseeds.resize(m); seedw.resize((size_t)seed_rows * m);
        if (ok) {
            if (nt) {
                HIP_TRY(c, hipMemcpy(travs.data(), done->d_trav.p, (size_t)nt * sizeof(groot_trav), hipMemcpyDeviceToHost));
                HIP_TRY(c, hipMemcpy(masks.data(), done->d_mask.p, (size_t)nt * pw * sizeof(uint64_t), hipMemcpyDeviceToHost));
            }
            HIP_TRY(c, hipMemcpy(nseeds.data(), c->ws[done->set].seed_count.p, (size_t)m * 4, hipMemcpyDeviceToHost));
            HIP_TRY(c, hipMemcpy2D(seedw.data(), (size_t)m * 4, c->ws[done->set].seed_win.p, (size_t)m * 4, (size_t)m * 4, seed_rows, hipMemcpyDeviceToHost));
            HIP_TRY(c, hipMemcpy(icnt.data(), c->incr_cnt.p, (size_t)m * 4, hipMemcpyDeviceToHost));
            HIP_TRY(c, hipMemcpy(iwin.data(), c->incr_win.p, (size_t)m * incr_cap * 4, hipMemcpyDeviceToHost));
        }
        release_slot(c, done);
        if (!ok) return GROOT_OK;
        size_t t0 = 0;
        for (uint32_t j = 0; j < m; j++) {                  // records come in (read, ord) order
            size_t t1 = t0;
            while (t1 < nt && travs[t1].read_id == j) t1++;
            const size_t sid = ids[j];
            const uint32_t cnt = (uint32_t)(t1 - t0), ni = icnt[j] & 0x7FFFFFFFu, nsd = nseeds[j] & 0x7FFFFFFFu;
            if (!second_pass && (ni > incr_cap || nsd > seed_rows)) { big.push_back(sid); t0 = t1; continue; }
            const size_t first = tab.size() / (sq * 4);
            // entries: one per traversal (at least one), and as many more -- without a record -- as the string's IncrementSubPath calls
            // (two per entry) and seed windows (four per entry) need: at lower containment thresholds a read brings several windows
            // of one graph and one traversal
            const bool seeds_fit = nsd <= c->seed_slots && nsd <= seed_rows;
            const uint32_t n_ent = std::max(std::max(cnt, 1u), std::max((ni + 1) / 2, seeds_fit ? (nsd + kOutSeedDw - 1) / kOutSeedDw : 0u));
            const uint32_t n_extra = n_ent - std::max(cnt, 1u);
            if (cnt <= kOutMaxTrav && ni <= incr_cap && n_extra < (1u << 12) && first + n_ent < (1u << kOutIdxBits)) {
                for (uint32_t e = 0; e < n_ent; e++) {
                    const size_t b = tab.size();
                    tab.resize(b + sq * 4, 0);
                    if (e < cnt) {
                        const groot_trav &t = travs[t0 + e];
                        tab[b] = t.node; tab[b + 1] = t.offset; tab[b + 2] = t.graph_id; tab[b + 3] = (uint32_t)t.flags;
                    } else tab[b] = kEmpty;
                    if (e == 0) tab[b + 2] = (tab[b + 2] & 0xFFFFFu) | (n_extra << 20);   // (graph ids stay below 2^20: checked at open)
                    // multimapped / mapped as the align stage counts them (boss.go:195-200): a read with seeds is mapped
                    if (e == 0) tab[b + 3] |= ((icnt[j] >> 31) ? 0x100u : 0u) | (nsd ? 0x200u : 0u) | (cnt << 16);
                    tab[b + 4] = 2 * e < ni ? iwin[(size_t)j * incr_cap + 2 * e] : kEmpty;
                    tab[b + 5] = 2 * e + 1 < ni ? iwin[(size_t)j * incr_cap + 2 * e + 1] : kEmpty;
                    uint32_t here = 0;                      // seed windows in this entry: bits 10..12 of [3]
                    for (uint32_t x = 0; x < kOutSeedDw; x++) {
                        const bool has = seeds_fit && kOutSeedDw * e + x < nsd;
                        tab[b + sq * 4 - kOutSeedDw + x] = has ? seedw[(size_t)(kOutSeedDw * e + x) * m + j] : kEmpty;
                        here += has;
                    }
                    tab[b + 3] |= here << 10;
                    for (uint32_t x = 0; x < pw; x++) {
                        tab[b + kOutHdrDw + 2 * x] = e < cnt ? (uint32_t)masks[(t0 + e) * pw + x] : 0u;
                        tab[b + kOutHdrDw + 2 * x + 1] = e < cnt ? (uint32_t)(masks[(t0 + e) * pw + x] >> 32) : 0u;
                    }
                }
                // are the IncrementSubPath calls exactly the string's seed windows, once each?  (then the signature kernel counts them itself)
                bool all_seeds = ni == nsd && ni <= 4 && ni <= c->seed_slots;
                if (all_seeds) {
                    uint32_t a4[4], b4[4];
                    for (uint32_t x = 0; x < ni; x++) { a4[x] = iwin[(size_t)j * incr_cap + x]; b4[x] = seedw[(size_t)x * m + j]; }
                    std::sort(a4, a4 + ni); std::sort(b4, b4 + ni);
                    all_seeds = std::equal(a4, a4 + ni, b4) && std::adjacent_find(a4, a4 + ni) == a4 + ni;
                }
                sinfo[sid] = kOutTab | (cnt ? std::min(cnt - 1, kOutTravLong) << kOutTravShift : kOutNoRec) | (all_seeds ? kOutAllSeeds : 0u) | (uint32_t)first;
                in_text[sid] = seeds_fit && text_ok;
                c->out_tabulated++;
            }
            t0 = t1;
        }
        return GROOT_OK;
    };
    {
        std::vector<size_t> ids(chunk);
        for (size_t s0 = 0; s0 < NT && !rc_all; s0 += chunk) {
            const uint32_t m = (uint32_t)std::min<size_t>(chunk, NT - s0);
            std::iota(ids.begin(), ids.begin() + m, s0);
            rc_all = run(ids.data(), m, kIncrCap, false);
        }
        // second pass: the few strings whose reads bring dozens of seed windows (a sequence shared by many graphs): everything kept
        const uint32_t chunk2 = std::min<uint32_t>(chunk, 4096);
        for (size_t s0 = 0; s0 < big.size() && !rc_all; s0 += chunk2)
            rc_all = run(big.data() + s0, (uint32_t)std::min<size_t>(chunk2, big.size() - s0), kIncrCapBig, true);
    }
    c->tab_capture = false;
    c->incr_cnt.release(); c->incr_win.release();
    // the capture batches counted IncrementSubPath calls and claimed rows of the call-count table: back to the state of a fresh ctx
    {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        std::vector<uint32_t> none(c->max_q + 2, kEmpty);
        HIP_TRY(c, hipMemcpy(c->q_row.p, none.data(), none.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemset(c->q_seen.p, 0, (size_t)(c->max_q + 2) * 4));
        HIP_TRY(c, hipMemset(c->q_nrows.p, 0, 4));
        if (c->att_cap) HIP_TRY(c, hipMemset(c->attempts_ptr, 0, (size_t)c->att_cap * c->n_windows * sizeof(uint32_t)));
        for (WorkSet &w : c->ws) w.owner = nullptr;
        c->trav_per_read = 1.25; c->bytes_per_trav = 0; c->dfs_frac = 1.0; c->todo_frac = 1.0;
    }
    if (rc_all) return rc_all;
    lap("pipeline on the strings");
    c->out_entries = tab.size() / (sq * 4);
    if (!c->out_entries) return GROOT_OK;
    // ---- 3. sig_info of the window-text strings (the signature kernel's way into the table): they are path strings ----
    {
        const unsigned nt = std::min(32u, granted_cpus());
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++) th.emplace_back([&, t]() {
        uint32_t buf[16];
        std::vector<uint32_t> pk;
        for (uint32_t i = t; i < n; i += nt) {                  // (the set is only read here)
            if (!tlen[i]) continue;
            for (uint32_t row = 0; row < 2; row++) {
                const uint8_t *src = &text[(size_t)i * 2 * kTextMax + row * kTextMax];
                pk.assign(tlen[i] / 16 + tw + 3, 0);
                for (uint32_t x = 0; x < tlen[i]; x++) pk[x >> 4] |= (uint32_t)((src[x] >> 1) & 3u) << (2 * (x & 15));
                for (uint32_t o = 0; o + w <= tlen[i]; o++) {
                    pack_at(pk, o, w, tw, buf);
                    const long j = set.find(buf, false);
                    if (j >= 0 && sinfo[(size_t)j]) info[((size_t)i * 2 + row) * vstride + o] = sinfo[(size_t)j];
                }
            }
        }
        });
        for (auto &x : th) x.join();
    }
    lap("sig_info");
    tab.resize(tab.size() + 16, 0);
    HIP_TRY(c, c->out_tab.alloc(c->out_entries * sq + 4));
    HIP_TRY(c, hipMemcpy(c->out_tab.p, tab.data(), (c->out_entries * sq + 4) * sizeof(uint4), hipMemcpyHostToDevice));
    c->h_out_tab = std::move(tab);
    HIP_TRY(c, hipMemcpy(c->sig_info.p, info.data(), info.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(sig_inline_off_kernel, dim3((unsigned)((c->sig.n + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream, c->sig.p, (uint32_t)c->sig.n);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (WorkSet &w : c->ws) HIP_TRY(c, w.tab_idx.alloc(c->prm.max_batch_reads));
    HIP_TRY(c, c->tab_hist.alloc(c->n_windows));
    HIP_TRY(c, hipMemset(c->tab_hist.p, 0, (size_t)c->n_windows * sizeof(uint32_t)));
    c->dix.out_tab = c->out_tab.p;
    c->dix.out_stride_q = sq;
    // ---- 4. text table: 64-byte entries {tag, sig_info word, bases}, keyed by the bases ----
    // (at ANY containment threshold: the memo is the pipeline's own output for the string under this ctx's parameters, whichever
    // branch of Query produced its seeds)
    if (text_ok) {
        size_t ns = 0;
        for (size_t j = 0; j < NT; j++) ns += in_text[j];
        uint32_t cap = 1024;
        while (cap < 2 * ns) cap <<= 1;
        // filled on the device: the strings are uploaded as they sit in the set, every thread claims a slot for its string with a
        // compare-and-swap on the entry's sig_info word (0 = free) and writes tag and bases behind it
        DevBuf<uint32_t> d_words, d_xwords, d_info;
        for (size_t j = 0; j < NT; j++) if (!in_text[j]) sinfo[j] = 0;      // (sinfo is not needed past this point)
        HIP_TRY(c, upload(d_words, set.words.data(), NS * tw));
        HIP_TRY(c, upload(d_xwords, xset.words.data(), NX * (twk + xw)));
        HIP_TRY(c, upload(d_info, sinfo.data(), NT));
        HIP_TRY(c, c->text_tab.alloc((size_t)cap * 4));
        HIP_TRY(c, hipMemsetAsync(c->text_tab.p, 0, (size_t)cap * 64, c->stream));
        if (NS) hipLaunchKernelGGL(text_table_fill_kernel, dim3((unsigned)((NS + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream, d_words.p, d_info.p, (uint32_t)NS, tw,
                                   tw, twk, reinterpret_cast<uint32_t *>(c->text_tab.p), cap - 1);
        if (NX) hipLaunchKernelGGL(text_table_fill_kernel, dim3((unsigned)((NX + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream, d_xwords.p, d_info.p + NS, (uint32_t)NX, twk,
                                   twk + xw, twk, reinterpret_cast<uint32_t *>(c->text_tab.p), cap - 1);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        c->dix.text_tab = c->text_tab.p;
        c->dix.text_mask = cap - 1;
        c->text_entries = ns;
    }
    lap("text table");
    if (stats)
        fprintf(stderr, "[groot open]   memo: %llu of %llu distinct path strings tabulated (%llu in the second pass), %llu entries of %u bytes; text table: %llu strings in %u slots of 64 bytes\n",
                (unsigned long long)c->out_tabulated, (unsigned long long)c->out_strings, (unsigned long long)big.size(), (unsigned long long)c->out_entries, sq * 16,
                (unsigned long long)c->text_entries, c->dix.text_tab ? c->dix.text_mask + 1 : 0u);
    return GROOT_OK;
}

static int build_signature_index(groot_ctx *c, const groot_index_view *v, const std::vector<uint32_t> &sketch_class)
{
    const uint32_t n = v->n_windows, s = v->sketch_size, w = v->window_size, k = v->kmer_size;
    if (c->kn.no_sig || !sig_supported(s, v->max_k, k) || w > kTextMax || w < k || !n || n >= (1u << 24)) return GROOT_OK;   // (SigEntry::group: 24 bits)
    const bool open_stats = c->kn.open_stats;
    auto t_lap = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!open_stats) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[groot open]   sig: %-22s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_lap).count());
        t_lap = now;
    };
    // 1. the bases every window was sketched from: WindowSize + MergeSpan of them along its first Ref path, starting at
    //    (Key.Node, Key.OffSet) -- WindowGraph walks a path through the graph's nodes in order (graph.go:243-262) and merges
    //    consecutive windows of equal sketch into the first one (:293-333).  Texts stop at a base other than ACGT.
    std::vector<uint8_t> text((size_t)n * 2 * kTextMax + 64, 0);
    std::vector<uint32_t> tlen(n, 0);
    const unsigned nt = std::min(32u, granted_cpus());
    {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++)
            th.emplace_back([&, t]() {
                for (uint32_t i = t; i < n; i += nt) {
                    if (v->win_ref_off[i] == v->win_ref_off[i + 1]) continue;
                    const uint32_t g = v->win_graph[i], p = v->win_ref[v->win_ref_off[i]];
                    const uint32_t n1 = v->graph_node_off[g + 1];
                    const uint32_t want = (uint32_t)std::min<uint64_t>(kTextMax, (uint64_t)w + v->win_merge_span[i]);
                    uint8_t *fw = &text[(size_t)i * 2 * kTextMax], *rc = fw + kTextMax;
                    uint32_t node = v->win_node[i], off = v->win_offset[i], got = 0;
                    bool stop = false;
                    for (; !stop && got < want && node < n1; node++, off = 0) {
                        if (!((v->node_mask[(size_t)node * v->path_words + (p >> 6)] >> (p & 63)) & 1ULL)) {
                            if (node == v->win_node[i]) stop = true;      // the window's own node is not on its path?
                            continue;
                        }
                        const uint32_t s0 = v->node_seq_off[node], nlen = v->node_seq_off[node + 1] - s0;
                        for (; off < nlen && got < want; off++) {
                            const uint8_t b = v->bases[s0 + off];
                            if (b != 'A' && b != 'C' && b != 'G' && b != 'T') { stop = true; break; }
                            fw[got++] = b;
                        }
                    }
                    if (got < w) { memset(fw, 0, kTextMax); continue; }
                    tlen[i] = got;
                    for (uint32_t j = 0; j < got; j++) {
                        const uint8_t b = fw[got - 1 - j];
                        rc[j] = b == 'A' ? 'T' : b == 'C' ? 'G' : b == 'G' ? 'C' : 'A';
                    }
                }
            });
        for (auto &x : th) x.join();
    }
    lap("texts");
    if (c->bg_cancel) return GROOT_OK;   // (background build abandoned: nothing of it is installed)
    // 2. proof and verdicts, one pass: every WindowSize-mer of both rows must reproduce Key.Sketch through the full-width kernel
    //    (a window whose text does not is left without one: its reads take the full-width kernel), and what the full-width
    //    seed stage's epilogue says about the same strings (the reads the signature kernel confirms ARE these strings) --
    //    verdict bits and dead-orientation class, one byte each.  A text whose own bases do not come back with a seed is dropped.
    const uint32_t vstride = kTextMax - w + 1;
    std::vector<uint32_t> verdict((size_t)n * 2 * vstride + 16, 0);   // DeviceIndex::sig_info (verdict bytes first, tabulated outcomes in step 5)
    {
        const uint32_t chunk = 1u << 20;
        std::vector<uint8_t> seqs, differs;
        std::vector<uint32_t> owner, flags, keys;
        std::vector<size_t> where;
        auto flush = [&]() -> int {
            if (owner.empty()) return GROOT_OK;
            flags.resize(owner.size()); keys.resize(owner.size()); differs.resize(owner.size());
            if (int rc = text_pass(c, seqs.data(), owner.data(), (uint32_t)owner.size(), w, differs.data(), flags.data(), keys.data())) return rc;
            for (size_t j = 0; j < owner.size(); j++) {
                if (differs[j] || !(flags[j] & kRecCountMask) || keys[j] == kEmpty) { tlen[owner[j]] = 0; continue; }
                verdict[where[j]] = ((flags[j] >> 24) & 0x3Fu) | ((keys[j] & 3u) << 6);
            }
            seqs.clear(); owner.clear(); where.clear();
            return GROOT_OK;
        };
        for (uint32_t i = 0; i < n; i++) {
            if (!tlen[i]) continue;
            for (uint32_t row = 0; row < 2; row++)
                for (uint32_t o = 0; o + w <= tlen[i]; o++) {
                    const uint8_t *src = &text[(size_t)i * 2 * kTextMax + row * kTextMax + o];
                    seqs.insert(seqs.end(), src, src + w);
                    owner.push_back(i);
                    where.push_back(((size_t)i * 2 + row) * vstride + o);
                }
            if (owner.size() >= chunk)
                if (int rc = flush()) return rc;
        }
        if (int rc = flush()) return rc;
        for (uint32_t i = 0; i < n; i++)
            if (!tlen[i]) memset(&text[(size_t)i * 2 * kTextMax], 0, 2 * kTextMax);
    }
    lap("proof + verdicts");
    if (c->bg_cancel) return GROOT_OK;   // (background build abandoned: nothing of it is installed)
    // 3. where the smallest k-mer of every text row is (first occurrence), and the rows at 2 bits per base
    std::vector<uint8_t> argmin((size_t)n * 2, 0);
    {
        DevBuf<uint8_t> d_text, d_pos;
        DevBuf<uint32_t> d_len;
        HIP_TRY(c, upload(d_text, text.data(), text.size()));
        HIP_TRY(c, upload(d_len, tlen.data(), tlen.size()));
        HIP_TRY(c, d_pos.alloc((size_t)n * 2));
        hipLaunchKernelGGL(text_argmin_kernel, dim3((2 * n + kBlock - 1) / kBlock), dim3(kBlock), 0, c->build_stream, d_text.p, d_len.p, 2 * n, k, d_pos.p);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipMemcpyAsync(argmin.data(), d_pos.p, argmin.size(), hipMemcpyDeviceToHost, c->build_stream));
        HIP_TRY(c, hipStreamSynchronize(c->build_stream));
    }
    std::vector<uint8_t> packed((size_t)n * 2 * (kTextMax / 4) + 64, 0);
    for (uint32_t i = 0; i < n; i++)
        for (uint32_t row = 0; row < 2; row++) {
            const uint8_t *src = &text[((size_t)i * 2 + row) * kTextMax];
            uint8_t *dst = &packed[((size_t)i * 2 + row) * (kTextMax / 4)];
            for (uint32_t j = 0; j < tlen[i]; j++) dst[j >> 2] |= (uint8_t)(((src[j] >> 1) & 3u) << (2 * (j & 3)));
        }
    c->sig_disabled = 0;
    for (uint32_t i = 0; i < n; i++) c->sig_disabled += tlen[i] == 0;
    std::vector<uint8_t> nodes(n);
    for (uint32_t i = 0; i < n; i++) nodes[i] = (uint8_t)std::min<uint32_t>(255, v->win_cn_off[i + 1] - v->win_cn_off[i]);
    lap("argmin + packing");
    if (c->bg_cancel) return GROOT_OK;   // (background build abandoned: nothing of it is installed)
    // 4. signature index: the windows grouped by the hash of their signature (kSigG slots of the sketch: kernels_common.hpp sig_step -- the slots the
    //    kernel computes), a group's windows sorted by (sketch class, id); a directory over the distinct signatures says where each group starts
    const uint32_t m5 = (uint32_t)(((uint64_t)k * GROOT_MULTI_SEED) & 31u);
    std::vector<uint64_t> key(n);
    for (uint32_t i = 0; i < n; i++) {
        uint64_t x = GROOT_SIG_HASH_INIT;
        x = sig_hash_step(x, sig_part(0, v->win_sketch[(size_t)i * s]));
        for (int j = 1; j < kSigG; j++) x = sig_hash_step(x, sig_part(j, v->win_sketch[(size_t)i * s + (uint32_t)(sig_step(j, (int)s, (int)m5) ^ (int)m5)]));
        key[i] = sig_hash_fin(x);
    }
    std::vector<uint32_t> order(n);
    std::iota(order.begin(), order.end(), 0u);
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
        if (key[a] != key[b]) return key[a] < key[b];
        if (sketch_class[a] != sketch_class[b]) return sketch_class[a] < sketch_class[b];
        return a < b;
    });
    std::vector<SigEntry> ent((size_t)n + 8, SigEntry{kEmpty, kEmpty, 0, 0, {{0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0}}});
    uint32_t n_keys = 0;
    for (uint32_t i = 0; i < n;) {
        uint32_t j = i;
        while (j < n && key[order[j]] == key[order[i]]) j++;
        for (uint32_t x = i; x < j; x++) {
            const uint32_t w_ = order[x];
            SigEntry e{w_, sketch_class[w_], sig_text_pack(tlen[w_], argmin[2 * w_], argmin[2 * w_ + 1]) | kSigInline, (j - x) | ((uint32_t)nodes[w_] << 24), {{0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0}}};
            for (uint32_t row = 0; row < 2; row++)
                for (uint32_t o = 0; o < 8 && o < vstride; o++) e.verdict[row][o] = (uint8_t)verdict[((size_t)w_ * 2 + row) * vstride + o];
            ent[x] = e;
        }
        n_keys++;
        i = j;
    }
    uint32_t cap = 16;                                       // buckets of two: load <= 1/4
    while ((uint64_t)cap * 2 < 4 * (uint64_t)n_keys) cap <<= 1;
    std::vector<uint32_t> dir((size_t)cap * 4);
    for (size_t i = 0; i < (size_t)cap; i++) { dir[4 * i] = 0; dir[4 * i + 1] = kEmpty; dir[4 * i + 2] = 0; dir[4 * i + 3] = kEmpty; }
    for (uint32_t i = 0; i < n; i += ent[i].group & 0xFFFFFFu) {
        const uint64_t x = key[order[i]];
        for (uint32_t b = (uint32_t)x & (cap - 1);; b = (b + 1) & (cap - 1)) {
            uint32_t *q = &dir[(size_t)b * 4];
            if (q[1] == kEmpty) { q[0] = (uint32_t)(x >> 32); q[1] = i; break; }
            if (q[3] == kEmpty) { q[2] = (uint32_t)(x >> 32); q[3] = i; break; }
        }
    }
    HIP_TRY(c, upload(c->sig, ent.data(), ent.size()));
    HIP_TRY(c, upload(c->sig_dir, reinterpret_cast<const uint4 *>(dir.data()), (size_t)cap));
    HIP_TRY(c, upload(c->win_text, packed.data(), packed.size()));
    HIP_TRY(c, upload(c->sig_info, verdict.data(), verdict.size()));
    HIP_TRY(c, upload(c->win_nodes, nodes.data(), nodes.size(), 4));
    c->build_dix->sig_info = c->sig_info.p;
    c->build_dix->sig_verdict_stride = vstride;
    c->build_dix->win_nodes = c->win_nodes.p;
    lap("tables + uploads");
    c->build_dix->sig = c->sig.p;
    c->build_dix->sig_dir = c->sig_dir.p;
    c->build_dix->sig_mask = cap - 1;
    c->build_dix->win_text = c->win_text.p;
    // 5. outcome table: the align stage itself, once, on every string that confirms reads
    if (c->build_dix == &c->dix && c->dix.sig_info && !c->kn.no_outcome_table && c->prm.memo_budget_mb != GROOT_MEMO_OFF && !c->prm.no_exact_align && !c->prm.keep_sketches && w <= c->prm.max_read_len && v->n_graphs < (1u << 20)) {
        const auto t0 = std::chrono::steady_clock::now();
        if (int rc = build_outcome_table(c, v, text, tlen, verdict, w, vstride)) return rc;
        c->out_build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        lap("outcome table");
    }
    return GROOT_OK;
}

// DeviceIndex::win_prefix: per window, which 6-mers its level-1 / level-2 start positions can spell as read bases [0, 6) and [6, 12)
// (kernels_misc.hpp prefix_positions_kernel / prefix_windows_kernel; on the host this took three core-seconds for arg-annot.90)
static int build_prefix_tables(groot_ctx *c, const groot_index_view *v)
{
    if (!v->n_windows || !v->n_nodes || !v->n_bases) return GROOT_OK;      // (an index without windows: groot_hip_sketch only)
    DevBuf<uint32_t> d_seq_off, d_edge_off, pos_bits;
    HIP_TRY(c, upload(d_seq_off, v->node_seq_off, (size_t)v->n_nodes + 1));
    HIP_TRY(c, upload(d_edge_off, v->node_edge_off, (size_t)v->n_nodes + 1));
    HIP_TRY(c, c->win_prefix.alloc((size_t)v->n_windows * kPrefixWords));
    static_assert(kPrefixWords == kBlock, "a thread per word of a window's two tables");
    // a pass per run of whole graphs holding up to 2 M bases (1 KB of sets per base position): the windows of a graph only start in its
    // own nodes, and nodes and windows are stored graph by graph (else: one pass over everything)
    bool grouped = true;
    for (uint32_t w = 1; w < v->n_windows; w++) grouped &= v->win_graph[w] >= v->win_graph[w - 1];
    const uint64_t kPassBases = 2u << 20;
    uint32_t g0 = 0, w0 = 0;
    while (g0 < v->n_graphs) {
        uint32_t g1 = g0 + 1;
        const uint32_t p0 = v->node_seq_off[v->graph_node_off[g0]];
        if (!grouped) g1 = v->n_graphs;
        else while (g1 < v->n_graphs && (uint64_t)v->node_seq_off[v->graph_node_off[g1 + 1]] - p0 <= kPassBases) g1++;
        const uint32_t p1 = v->node_seq_off[v->graph_node_off[g1]];
        uint32_t w1 = w0;
        if (!grouped) w1 = v->n_windows;
        else while (w1 < v->n_windows && v->win_graph[w1] < g1) w1++;
        if (p1 > p0 && w1 > w0) {
            const size_t words = (size_t)(p1 - p0) * 256 + 256;
            HIP_TRY(c, pos_bits.reserve(words));
            HIP_TRY(c, hipMemsetAsync(pos_bits.p, 0, words * sizeof(uint32_t), c->build_stream));
            PrefixBuildArgs a{};
            a.bases = c->bases.p; a.seq_off = d_seq_off.p; a.edge_off = d_edge_off.p; a.edges = c->edges.p;
            a.n_nodes = v->n_nodes; a.p0 = p0; a.p1 = p1; a.pos_bits = pos_bits.p;
            hipLaunchKernelGGL(prefix_positions_kernel, dim3((unsigned)((2 * (uint64_t)(p1 - p0) + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->build_stream, a);
            hipLaunchKernelGGL(prefix_windows_kernel, dim3(w1 - w0), dim3(kBlock), 0, c->build_stream, c->win_rec.p, c->cn_pre.p, d_seq_off.p, pos_bits.p, p0, w0, w1, c->win_prefix.p);
            HIP_TRY(c, hipGetLastError());
            HIP_TRY(c, hipStreamSynchronize(c->build_stream));
        } else if (w1 > w0) HIP_TRY(c, hipMemsetAsync(c->win_prefix.p + (size_t)w0 * kPrefixWords, 0, (size_t)(w1 - w0) * kPrefixWords * sizeof(uint32_t), c->build_stream));
        g0 = g1; w0 = w1;
    }
    HIP_TRY(c, hipStreamSynchronize(c->build_stream));
    c->build_dix->win_prefix = c->win_prefix.p;
    return GROOT_OK;
}

// what a background open has finished moves into the ctx's index description: between two batches, on the caller's thread
static int install_background(groot_ctx *c, bool wait)
{
    const int st0 = c->bg_state.load(std::memory_order_acquire);
    if (st0 == 0 || (st0 == 1 && !wait)) return GROOT_OK;
    if (c->bg.joinable()) c->bg.join();
    const int st = c->bg_state.load(std::memory_order_acquire);
    c->bg_state.store(0);
    c->build_dix = &c->dix; c->build_stream = c->stream; c->build_shards = c->seed_shards.p;
    if (st == 3) return fail(c, c->bg_rc ? c->bg_rc : GROOT_E_DEVICE, "background part of groot_hip_open: %s", c->bg_err.c_str());
    if (st == 4) return GROOT_OK;         // abandoned: the ctx goes on with the full-width kernels (same results)
    const DeviceIndex &b = c->bg_dix;
    c->dix.win_prefix = b.win_prefix;
    c->dix.sig = b.sig; c->dix.sig_dir = b.sig_dir; c->dix.sig_mask = b.sig_mask; c->dix.win_text = b.win_text;
    c->dix.sig_info = b.sig_info; c->dix.sig_verdict_stride = b.sig_verdict_stride; c->dix.win_nodes = b.win_nodes;
    return GROOT_OK;
}

static int open_impl(groot_ctx *c, int device_id, const groot_index_view *v, const groot_params *p, uint32_t flags)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(c, GROOT_E_DEVICE, "no HIP device available (libgroot_hip has no CPU fallback)");
    if (device_id < 0 || device_id >= ndev) return fail(c, GROOT_E_INVALID, "device %d out of range (%d devices)", device_id, ndev);
    if (!v) return fail(c, GROOT_E_INVALID, "null index view");
    c->device = device_id;
    c->kn = Knobs::read();
    HIP_TRY(c, hipSetDevice(device_id));
    groot_params d;
    groot_params_default(&d);
    c->prm = p ? *p : d;
    if (!c->prm.max_read_len) c->prm.max_read_len = d.max_read_len;
    if (!c->prm.max_batch_reads) c->prm.max_batch_reads = d.max_batch_reads;
    if (!c->prm.max_seeds_per_read) c->prm.max_seeds_per_read = d.max_seeds_per_read;
    if (!c->prm.max_batch_bases) c->prm.max_batch_bases = (uint64_t)c->prm.max_batch_reads * c->prm.max_read_len;
    if (!c->prm.pipeline_depth) c->prm.pipeline_depth = d.pipeline_depth;
    if (c->prm.pipeline_depth > 16) return fail(c, GROOT_E_INVALID, "pipeline_depth must be <= 16");
    if (c->prm.max_read_len > 65535) return fail(c, GROOT_E_UNSUPPORTED, "max_read_len must be <= 65535");
    if (v->kmer_size == 0 || v->kmer_size > 64) return fail(c, GROOT_E_UNSUPPORTED, "k-mer size %u not in [1,64]", v->kmer_size);
    if (c->prm.max_read_len < v->kmer_size) return fail(c, GROOT_E_INVALID, "max_read_len smaller than the k-mer size");
    {   // never upload a view whose indices do not resolve (truncated / corrupt index, wrong file)
        const std::string why = check_index_view(v);
        if (!why.empty()) return fail(c, GROOT_E_FORMAT, "inconsistent index view: %s", why.c_str());
    }
    if (!seed_supported(v->sketch_size, v->max_k))
        return fail(c, GROOT_E_UNSUPPORTED, "sketch size %u with maxK %u is outside what the kernels handle (1 <= maxK <= sketch size <= %d)", v->sketch_size,
                    v->max_k, kGenericMaxS);
    c->s = v->sketch_size; c->k = v->kmer_size; c->max_k = v->max_k; c->l_max = v->sketch_size / v->max_k;
    c->pw_view = v->path_words; c->pw = round_pw(v->path_words);
    if (!c->pw) return fail(c, GROOT_E_UNSUPPORTED, "graphs with more than 704 paths are not supported (path_words=%u)", v->path_words);
    c->n_windows = v->n_windows;
    c->max_q = c->prm.max_read_len - c->k + 1;

    HIP_TRY(c, hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    {
        // the align stream gets a priority of its own (the lowest): streams of different priorities never share a hardware queue -- two
        // streams on one queue run their kernels in turn, seen once as a headline of 5.3 instead of 9 Greads/s -- and the hashing
        // kernels, which are the longer stage on most workloads, get their workgroups placed first
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (hipStreamCreateWithPriority(&c->astream, hipStreamNonBlocking, lo) != hipSuccess)
            HIP_TRY(c, hipStreamCreateWithFlags(&c->astream, hipStreamNonBlocking));
    }
    for (WorkSet &w : c->ws) HIP_TRY(c, hipEventCreateWithFlags(&w.ev_free, hipEventDisableTiming));
    HIP_TRY(c, hipStreamCreateWithFlags(&c->h2d_stream, hipStreamNonBlocking));
    HIP_TRY(c, hipStreamCreateWithFlags(&c->d2h_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    c->build_dix = &c->dix; c->build_stream = c->stream;
    for (uint32_t i = 0; i < c->prm.pipeline_depth; i++) {
        std::unique_ptr<Slot> s(new Slot());
        for (hipEvent_t *e : {&s->ev_seed, &s->ev_h2d0, &s->ev_h2d, &s->ev_compute, &s->ev_ctr, &s->ev_d2h0, &s->ev_d2h}) HIP_TRY(c, hipEventCreate(e));
        for (auto &e : s->ev) HIP_TRY(c, hipEventCreate(&e));
        c->slots.push_back(std::move(s));
    }

    const bool open_stats = c->kn.open_stats;
    auto t_open = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!open_stats) return;
        (void)hipDeviceSynchronize();
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[groot open] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_open).count());
        t_open = now;
    };
    lap("device + streams");
    // ---- graphs + windows -> HBM ----
    HIP_TRY(c, upload(c->edges, v->edges, v->n_edges));
    HIP_TRY(c, upload(c->bases, v->bases, v->n_bases, 64));   // kernels read 8-byte windows up to 24 bytes past a node start
    {
        std::vector<unsigned char> recs;
        if (c->pw == 3) build_node_records<3>(v, recs);
        else build_node_records<11>(v, recs);
        HIP_TRY(c, upload(c->node_rec, recs.data(), recs.size()));
    }
    // ---- first pass of the align stage (kernels_lean.hpp): everything at 2 bits per base ----
    c->lean = c->pw == 3 && c->kn.lean && !c->prm.no_exact_align;
    if (c->lean) {
        auto code_of = [](uint8_t b) -> int { return b == 'A' ? 0 : b == 'C' ? 1 : b == 'T' ? 2 : b == 'G' ? 3 : -1; };
        std::vector<uint32_t> b2((size_t)(v->n_bases + 15) / 16 + 20, 0);
        for (uint64_t i = 0; i < v->n_bases; i++) {
            const int cd = code_of(v->bases[i]);
            if (cd > 0) b2[i >> 4] |= (uint32_t)cd << (2 * (i & 15));
        }
        std::vector<LeanNode> ln(v->n_nodes);
        std::vector<LeanExt> lx(v->n_nodes);
        std::vector<uint8_t> node_bad(v->n_nodes, 0);
        for (uint32_t n = 0; n < v->n_nodes; n++) {
            LeanNode &r = ln[n];
            memset(&r, 0, sizeof r);
            memset(&lx[n], 0, sizeof(LeanExt));
            r.seq_off = v->node_seq_off[n];
            r.seq_len = v->node_seq_off[n + 1] - v->node_seq_off[n];
            const uint32_t e0 = v->node_edge_off[n], deg = v->node_edge_off[n + 1] - e0;
            bool no = deg > 4;
            for (uint32_t i = 0; i < r.seq_len; i++) {
                const int cd = code_of(v->bases[r.seq_off + i]);
                if (cd < 0) no = true;                             // the graph's 'N' (alignment.go:212-222): align_kernel's business
                else if (i < 32) r.first32 |= (uint64_t)cd << (2 * i);
                else if (i < 256) lx[n].b[(i >> 5) - 1] |= (uint64_t)cd << (2 * (i & 31));
            }
            node_bad[n] = no;
            r.deg_kids = std::min(deg, 7u) | (no ? kLeanNo : 0u);
            for (uint32_t e = 0; e < std::min(deg, 4u); e++) {
                const uint32_t ch = v->edges[e0 + e];
                r.edges[e] = ch;
                uint32_t kid = 8;                                   // an empty neighbour spells nothing: never entered
                if (v->node_seq_off[ch] < v->node_seq_off[ch + 1]) {
                    const int cd = code_of(v->bases[v->node_seq_off[ch]]);
                    kid = cd < 0 ? 4u : (uint32_t)cd;
                }
                r.deg_kids |= kid << (8 + 4 * e);
                if (v->node_seq_off[ch + 1] - v->node_seq_off[ch] > 32u) r.deg_kids |= 1u << (24 + e);
            }
            for (uint32_t w = 0; w < v->path_words && w < 3; w++) r.mask[w] = v->node_mask[(size_t)n * v->path_words + w];
        }
        std::vector<uint32_t> pre2((size_t)v->n_cn * 4 + 4, 0);
        for (uint64_t i = 0; i < v->n_cn; i++) {
            const uint32_t nd = v->cn_node[i];
            const uint32_t s0 = v->node_seq_off[nd], nlen = v->node_seq_off[nd + 1] - s0;
            uint64_t bits = 0;
            for (uint32_t j = 0; j < std::min(nlen, 24u); j++) {
                const int cd = code_of(v->bases[s0 + j]);
                if (cd > 0) bits |= (uint64_t)cd << (2 * j);
            }
            uint32_t *e = &pre2[(size_t)i * 4];
            e[0] = (uint32_t)bits; e[1] = (uint32_t)(bits >> 32) | (std::min(nlen, 65535u) << 16); e[2] = nd; e[3] = s0;
        }
        std::vector<uint8_t> ok(v->n_windows, 1);
        for (uint32_t w = 0; w < v->n_windows; w++) {
            if (node_bad[v->win_node[w]]) ok[w] = 0;
            for (uint32_t i = v->win_cn_off[w]; i < v->win_cn_off[w + 1]; i++) {
                const uint32_t nd = v->cn_node[i];
                if (node_bad[nd] || v->node_seq_off[nd + 1] - v->node_seq_off[nd] > 65535u) ok[w] = 0;
            }
        }
        HIP_TRY(c, upload(c->bases2, b2.data(), b2.size()));
        HIP_TRY(c, upload(c->lean_nodes, ln.data(), ln.size()));
        HIP_TRY(c, upload(c->lean_ext, lx.data(), lx.size()));
        HIP_TRY(c, upload(c->cn_pre2, reinterpret_cast<const uint4 *>(pre2.data()), pre2.size() / 4));
        HIP_TRY(c, upload(c->win_ok, ok.data(), ok.size()));
    }
    {   // level 2 of AlignRead: the first 24 bases, index and length of every ContainedNodes entry, in list order (DeviceIndex::cn_pre)
        std::vector<uint32_t> pre((size_t)v->n_cn * 8 + 8, 0);
        for (uint64_t i = 0; i < v->n_cn; i++) {
            const uint32_t nd = v->cn_node[i];
            const uint32_t s0 = v->node_seq_off[nd], nlen = v->node_seq_off[nd + 1] - s0;
            uint32_t *e = &pre[(size_t)i * 8];
            memcpy(e, v->bases + s0, std::min(nlen, 24u));
            e[6] = nd; e[7] = nlen;
        }
        HIP_TRY(c, upload(c->cn_pre, reinterpret_cast<const uint4 *>(pre.data()), pre.size() / 4));
        c->dix.cn_pre = c->cn_pre.p;
    }
    {   // DeviceIndex::node_l2b: which 8-mers a DFS from (node, offset 0..10) can spell
        std::vector<uint64_t> sets((size_t)v->n_nodes * 11, 0);
        const unsigned nt = std::max(1u, std::min(32u, granted_cpus()));
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++)
            th.emplace_back([&, t]() {
                struct Walk {
                    const groot_index_view *v;
                    uint64_t set = 0;
                    uint32_t budget = 0;
                    // the strings of length 8 that start with `code` (d bases so far) and go on at (node, off)
                    void go(uint32_t node, uint32_t off, uint32_t d, uint32_t code)
                    {
                        if (set == ~0ull) return;
                        if (++budget > 4096) { set = ~0ull; return; }               // (a thicket of N and branches: anything goes)
                        const uint32_t s0 = v->node_seq_off[node], len = v->node_seq_off[node + 1] - s0;
                        if (off >= len) return;                                      // alignment.go:199-201 (also: an empty node ends the path)
                        for (uint32_t i = off; i < len; i++) {
                            if (d == 8) { set |= l2_bloom_bits(code); return; }
                            const uint8_t b = v->bases[s0 + i];
                            if (b == 'N') {                                          // :212-216 the graph's wildcard
                                for (uint32_t x = 0; x < 4; x++) rest(node, i + 1, d + 1, code | (x << (2 * d)));
                                return;
                            }
                            if (b != 'A' && b != 'C' && b != 'G' && b != 'T') return;   // never equals a base of the read
                            code |= (uint32_t)((b >> 1) & 3u) << (2 * d);
                            d++;
                        }
                        rest_at_end(node, d, code);
                    }
                    // ... continuing inside the node at i (after a wildcard)
                    void rest(uint32_t node, uint32_t i, uint32_t d, uint32_t code)
                    {
                        const uint32_t s0 = v->node_seq_off[node], len = v->node_seq_off[node + 1] - s0;
                        if (i < len) { go(node, i, d, code); return; }
                        rest_at_end(node, d, code);
                    }
                    void rest_at_end(uint32_t node, uint32_t d, uint32_t code)
                    {
                        if (d == 8) { set |= l2_bloom_bits(code); return; }
                        const uint32_t e0 = v->node_edge_off[node], deg = v->node_edge_off[node + 1] - e0;
                        if (!deg) { set = ~0ull; return; }                           // :229-236 a sink reports the traversal whatever the read goes on with
                        for (uint32_t e = 0; e < deg; e++) go(v->edges[e0 + e], 0, d, code);
                    }
                };
                for (uint32_t nd = t; nd < v->n_nodes; nd += nt) {
                    const uint32_t len = v->node_seq_off[nd + 1] - v->node_seq_off[nd];
                    for (uint32_t o = 0; o < std::min(len, 11u); o++) {
                        Walk w{v};
                        w.go(nd, o, 0, 0);
                        sets[(size_t)nd * 11 + o] = w.set;
                    }
                }
            });
        for (auto &x : th) x.join();
        HIP_TRY(c, upload(c->node_l2b, sets.data(), sets.size(), 2));
        c->dix.node_l2b = c->node_l2b.p;
    }
    lap("node records + prefix tables");
    HIP_TRY(c, upload(c->win_graph, v->win_graph, v->n_windows));
    c->h_node_graph.resize(v->n_nodes);
    for (uint32_t g = 0; g < v->n_graphs; g++)
        for (uint32_t nd = v->graph_node_off[g]; nd < v->graph_node_off[g + 1]; nd++) c->h_node_graph[nd] = g;
    c->packed_travs = !c->prm.results_on_device && c->prm.max_batch_reads <= (1u << 24);
    {   // windows are numbered graph by graph (canonical seed order): the last window of every graph
        std::vector<uint32_t> end(v->n_graphs, 0);
        bool grouped = true;
        for (uint32_t w = 0; w < v->n_windows; w++) {
            if (w && v->win_graph[w] < v->win_graph[w - 1]) grouped = false;
            end[v->win_graph[w]] = w + 1;
        }
        if (grouped && v->n_graphs) {
            HIP_TRY(c, upload(c->graph_win_end, end.data(), end.size()));
            c->dix.graph_win_end = c->graph_win_end.p;
        }
    }
    {
        std::vector<uint8_t> gw(v->n_graphs, 1);
        for (uint32_t g = 0; g < v->n_graphs; g++) gw[g] = (uint8_t)std::max<uint32_t>(1, (v->graph_path_off[g + 1] - v->graph_path_off[g] + 7) / 8);   // (<= 88: path sets of up to 704 bits)
        HIP_TRY(c, upload(c->graph_words, gw.data(), gw.size()));
        c->h_graph_words = gw;
    }
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

    lap("window arrays");
    // ---- lookup structures (the reference bootstraps its LSH forests at load too, lshe.go:95-147) ----
    const uint32_t n = v->n_windows, s = v->sketch_size;
    std::vector<uint32_t> sketch_class(n);   // smallest window id with the same 64-bit sketch
    {   // exact-match table
        uint32_t cap = 16;
        while (cap < 2 * (uint64_t)n) cap <<= 1;
        std::vector<ExactEntry> tab(cap, ExactEntry{0, kEmpty});
        for (uint32_t w = 0; w < n; w++) {
            uint64_t h = GROOT_SKETCH_HASH_INIT;
            for (uint32_t i = 0; i < s; i++) h = sketch_hash_step(h, v->win_sketch[(size_t)w * s + i]);
            uint32_t slot = (uint32_t)h & (cap - 1);
            sketch_class[w] = w;
            for (; tab[slot].id != kEmpty; slot = (slot + 1) & (cap - 1))
                if (sketch_class[w] == w && tab[slot].tag == (uint32_t)(h >> 32) &&
                    !memcmp(v->win_sketch + (size_t)tab[slot].id * s, v->win_sketch + (size_t)w * s, (size_t)s * 8))
                    sketch_class[w] = sketch_class[tab[slot].id];
            tab[slot] = ExactEntry{(uint32_t)(h >> 32), w};
        }
        HIP_TRY(c, upload(c->exact, tab.data(), tab.size()));
        c->dix.exact_mask = cap - 1;
    }
    lap("exact table");
    // LSH forest band tables: per band the low-32 hash values of its max_k slots, sorted.  Host work only (sorts), so it runs on
    // its own threads while the main thread goes on building what the window-sized strings of the signature index / memo need
    // first; joined and uploaded before anything can take the LSH-Forest branch (finish_lsh below).
    struct LshHost {
        std::vector<uint32_t> keys, ids, run;
        std::vector<ExactEntry> tab;
        std::vector<uint8_t> sig;
        std::thread job;
        ~LshHost() { if (job.joinable()) job.join(); }
    } lsh;
    {
        const uint32_t mk = v->max_k, lmax = c->l_max;
        lsh.keys.assign((size_t)lmax * n * mk, 0); lsh.ids.assign((size_t)lmax * n, 0);
        // hash tables over the distinct K-prefixes of every band: the query finds the first matching row with one or two
        // probes instead of a binary search of ~log2(n) dependent loads
        uint32_t bits = 4;
        while ((1ull << bits) < 2 * (uint64_t)n) bits++;
        c->band_hash_bits = bits;
        const uint32_t cap = 1u << bits;
        lsh.tab.assign((size_t)lmax * mk * cap, ExactEntry{0, kEmpty});
        lsh.sig.assign((size_t)lmax * n * kRowBytes, 0);
        lsh.run.assign((size_t)lmax * mk * n, 0);
        const uint32_t sl = std::min<uint32_t>(s, kRowSlots);
        lsh.job = std::thread([&lsh, v, n, s, mk, lmax, bits, cap, sl]() {
        auto &keys = lsh.keys; auto &ids = lsh.ids; auto &tab = lsh.tab; auto &sig = lsh.sig; auto &run = lsh.run;
        auto band = [&](uint32_t b) {                            // the bands are independent: one thread each
            std::vector<uint32_t> order(n);
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
            for (uint32_t e = 0; e < n; e++) {
                const uint64_t *ws = v->win_sketch + (size_t)ids[(size_t)b * n + e] * s;
                uint32_t row[4] = {0, 0, 0, 0};                 // 5 bits per slot, six slots to a dword (kernels_common.hpp row_same6)
                for (uint32_t i = 0; i < sl; i++) row[i / 6] |= sig5(ws[i]) << (5 * (i % 6));
                memcpy(&sig[((size_t)b * n + e) * kRowBytes], row, kRowBytes);
            }
            for (uint32_t K = 1; K <= mk; K++) {
                uint32_t *rn = run.data() + ((size_t)b * mk + (K - 1)) * n;
                for (uint32_t e = n; e-- > 0;) {             // backwards: length of the run of equal K-prefixes starting at e
                    const uint32_t *ke = &keys[((size_t)b * n + e) * mk];
                    rn[e] = (e + 1 < n && std::equal(ke, ke + K, ke + mk)) ? rn[e + 1] + 1 : 1;
                }
            }
        };
        {
            std::atomic<uint32_t> next{0};
            std::vector<std::thread> th;
            const uint32_t workers = std::min<uint32_t>(lmax, std::min(16u, granted_cpus()));
            for (uint32_t t = 0; t < workers; t++)
                th.emplace_back([&]() { for (uint32_t b; (b = next.fetch_add(1)) < lmax;) band(b); });
            for (auto &x : th) x.join();
        }
        });
    }
    auto finish_lsh = [&]() -> int {
        if (lsh.job.joinable()) lsh.job.join();
        if (c->band_keys.p) return GROOT_OK;
        HIP_TRY(c, upload(c->band_keys, lsh.keys.data(), lsh.keys.size()));
        HIP_TRY(c, upload(c->band_ids, lsh.ids.data(), lsh.ids.size()));
        HIP_TRY(c, upload(c->band_hash, lsh.tab.data(), lsh.tab.size()));
        HIP_TRY(c, upload(c->band_sig, lsh.sig.data(), lsh.sig.size(), 32));
        HIP_TRY(c, upload(c->band_run, lsh.run.data(), lsh.run.size()));
        c->dix.band_keys = c->band_keys.p; c->dix.band_ids = c->band_ids.p; c->dix.band_hash = c->band_hash.p;
        c->dix.band_sig = c->band_sig.p; c->dix.band_run = c->band_run.p;
        lsh.keys = {}; lsh.ids = {}; lsh.run = {}; lsh.tab = {}; lsh.sig = {};
        return GROOT_OK;
    };
    lap("LSH forest tables");
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
        c->h_q_min_eq = qm;
    }
    {   // call-count table: rows appear as kmerCounts do
        std::vector<uint32_t> none(c->max_q + 2, kEmpty);
        HIP_TRY(c, upload(c->q_row, none.data(), none.size()));
        HIP_TRY(c, c->q_seen.alloc(c->max_q + 2));
        HIP_TRY(c, hipMemset(c->q_seen.p, 0, (size_t)(c->max_q + 2) * 4));
        HIP_TRY(c, c->q_of_row.alloc(c->max_q + 2));
        HIP_TRY(c, c->q_nrows.alloc(1));
        HIP_TRY(c, hipMemset(c->q_nrows.p, 0, 4));
        if (int rc = grow_attempts(c, std::min<uint32_t>(4, c->max_q + 1))) return rc;
    }
    DeviceIndex &x = c->dix;
    x.k = v->kmer_size; x.s = s; x.w = v->window_size; x.num_window_kmers = v->num_window_kmers;
    x.n_windows = n; x.n_nodes = v->n_nodes; x.pw = c->pw;
    x.edges = c->edges.p; x.bases = c->bases.p;
    x.win_prefix = c->win_prefix.p; x.win_graph = c->win_graph.p; x.win_rec = c->win_rec.p; x.cn_node = c->cn_node.p;
    x.win_sketch = c->win_sketch.p; x.exact = c->exact.p; x.band_keys = c->band_keys.p; x.band_ids = c->band_ids.p;
    x.band_hash = c->band_hash.p; x.band_hash_bits = c->band_hash_bits; x.band_sig = c->band_sig.p; x.band_run = c->band_run.p;
    x.max_k = v->max_k; x.l_max = c->l_max; x.q_k = c->q_k.p; x.q_l = c->q_l.p; x.q_min_eq = c->q_min_eq.p; x.max_q = c->max_q;
    x.q_row = c->q_row.p;

    lap("per-kmerCount tables");
    // ---- shared work buffers (inputs / outputs are per pipeline slot, allocated at their first use) ----
    const uint32_t R = c->prm.max_batch_reads;
    HIP_TRY(c, c->sort_key.alloc(R));
    HIP_TRY(c, c->sort_key_out.alloc(R));
    HIP_TRY(c, c->long_list.alloc(kLongListCap));
    HIP_TRY(c, c->long_count.alloc(4));
    HIP_TRY(c, hipMemset(c->long_count.p, 0, 4 * sizeof(uint32_t)));
    {
        std::vector<uint32_t> iota(R);
        std::iota(iota.begin(), iota.end(), 0u);
        HIP_TRY(c, upload(c->perm_in, iota.data(), iota.size()));
    }
    if (int rc = alloc_seed_slots(c, c->prm.max_seeds_per_read)) return rc;
    // (+ vcap slots behind the reads: the items of split reads, AlignArgs::vitem)
    c->vcap = c->kn.small_buffers ? 8u : std::max<uint32_t>(4096, R / 4);   // (small: most split reads find no room for their items and are handled whole)
    for (WorkSet &w : c->ws) {     // what a batch's seed stage hands to its align and order stages: two sets, taken in turn
        HIP_TRY(c, w.seed_count.alloc(R));
        HIP_TRY(c, w.read_rec.alloc(R));
        HIP_TRY(c, w.perm.alloc(R));
        HIP_TRY(c, w.perm_count.alloc(4));
        if (c->lean) {
            HIP_TRY(c, w.perm2.alloc(R));
            HIP_TRY(c, w.perm2_count.alloc(4));
            HIP_TRY(c, w.defer.alloc(R));
            HIP_TRY(c, w.packed.alloc((size_t)R * (c->prm.max_read_len <= 128 ? 2 : 4)));
        }
        if (c->prm.keep_sketches) HIP_TRY(c, w.sketches.alloc((size_t)R * s));
        HIP_TRY(c, w.trav_first.alloc((size_t)R + c->vcap));
        HIP_TRY(c, w.mask_first.alloc(((size_t)R + c->vcap) * c->pw));
        HIP_TRY(c, w.trav_cnt.alloc((size_t)R + c->vcap));
        HIP_TRY(c, w.vitem.alloc(std::max<uint32_t>(c->vcap, 1)));
        HIP_TRY(c, w.split_list.alloc(kLongListCap));
        HIP_TRY(c, w.vcount.alloc(4));
        HIP_TRY(c, hipMemset(w.vcount.p, 0, 4 * sizeof(uint32_t)));
    }
    HIP_TRY(c, c->trav_off.alloc(R));
    if (c->lean) HIP_TRY(c, c->lean_stk.alloc((size_t)R * 4));
    HIP_TRY(c, c->ovf_cnt.alloc(kOvfShards + 2));
    if (int rc = alloc_ovf(c, c->kn.small_buffers ? 2u : std::max<uint32_t>(256, R / kOvfShards / 4))) return rc;
    // the align kernel is persistent: exactly the workgroups that are resident at once (GROOT_ALIGN_WAVES per SIMD = per CU)
    int n_cu = 256;
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, c->device);
    c->n_cu = (uint32_t)std::max(n_cu, 1);
    uint32_t per_cu = c->pw > 3 ? kAlignWavesWide : kAlignWaves;
    // (3 or 2 workgroups of the persistent grid per CU instead of 4, so that the next batch's hashing kernels find free registers from the start: measured in
    // round 4 -- 3: no difference on any kernel-path workload, 2: mixed 8 M 1 022 -> 983, configs[2] through the kernels 1 861 -> 1 740 Mreads/s)
    c->align_threads = std::min<uint32_t>(((R + kBlock - 1) / kBlock) * kBlock, (uint32_t)std::max(n_cu, 1) * per_cu * kBlock);
    c->stk_depth = c->prm.max_read_len;
    HIP_TRY(c, c->stk_hdr.alloc((size_t)c->stk_depth * c->align_threads));
    HIP_TRY(c, c->stk_mask.alloc((size_t)c->stk_depth * c->align_threads * c->pw));
    // LSH-Forest branch of Query: the hashing kernels query in place (a lane per read) and hand the reads with more than lsh_defer_rows
    // candidate rows to lsh_heavy_kernel (a wavefront per read).  Two alternatives were built, measured slower on every workload
    // and removed in round 4 ("Removed"): a kernel dealing the rows of 64 reads over a wavefront, and a query launch of its own.
    if (c->l_max <= kLshMaxBands) {
#ifndef GROOT_LSH_DEFER_ROWS
#define GROOT_LSH_DEFER_ROWS 64    // (round 5, mixed 75..150-base reads, 8 M per batch, t = 0.99 / 0.90: 8 -> 813 / 404, 16 -> 980 / 512, 32 -> 1 120 / 618, 64 -> 1 203 / 656 Mreads/s)
#endif
        c->lsh_defer_rows = GROOT_LSH_DEFER_ROWS;
        c->lsh_cap = c->kn.small_buffers ? 4u : std::max<uint32_t>(4096, R / 4);   // (small: most heavy reads find the list full and walk their own rows)
        HIP_TRY(c, c->lsh_list.alloc(c->lsh_cap));
        HIP_TRY(c, c->lsh_count.alloc(4));
        HIP_TRY(c, c->lsh_sketch.alloc((size_t)c->lsh_cap * s));
    }
    HIP_TRY(c, c->todo_list.alloc(R));
    HIP_TRY(c, c->todo_count.alloc(1));
    HIP_TRY(c, c->seed_shards.alloc((size_t)kSeedShards * kSeedShardStride));
    HIP_TRY(c, hipMemset(c->seed_shards.p, 0, (size_t)kSeedShards * kSeedShardStride * sizeof(unsigned long long)));
    HIP_TRY(c, hipDeviceSynchronize());
    lap("work buffers");
    c->build_shards = c->seed_shards.p;
    // WindowSize-mers on the every-slot-equal branch of Query never touch the LSH-Forest tables: the signature index and the memo
    // (both run window-sized strings through the kernels) are built while the band tables are still being sorted
    const uint32_t q_w = v->window_size >= v->kmer_size ? v->window_size - v->kmer_size + 1 : 0;
    const bool w_exact = q_w && q_w < c->h_q_min_eq.size() && c->h_q_min_eq[q_w] == s;
    // The memo needs the signature index and the ctx's whole pipeline: with it everything is built here and now.
    const bool memo_wanted = !c->kn.no_outcome_table && c->prm.memo_budget_mb != GROOT_MEMO_OFF && !c->prm.no_exact_align && !c->prm.keep_sketches;
    if ((flags & GROOT_OPEN_BACKGROUND) && !memo_wanted) {
        // prefix tables (0.2 s on arg-annot.90) and signature index (0.4 s) on a thread of their own: the ctx takes batches at once --
        // through the full-width kernel and without the seed stage's verdicts until they are there (same results)
        if (int rc = finish_lsh()) return rc;
        lap("LSH forest tables (waited for)");
        c->bg_dix = c->dix;
        HIP_TRY(c, hipStreamCreateWithFlags(&c->bg_stream, hipStreamNonBlocking));
        HIP_TRY(c, c->bg_shards.alloc((size_t)kSeedShards * kSeedShardStride));
        HIP_TRY(c, hipMemset(c->bg_shards.p, 0, (size_t)kSeedShards * kSeedShardStride * sizeof(unsigned long long)));
        c->build_dix = &c->bg_dix; c->build_stream = c->bg_stream; c->build_shards = c->bg_shards.p;
        c->bg_seed_slots = c->seed_slots; c->bg_max_read_len = c->prm.max_read_len;
        c->bg_state.store(1);
        c->bg = std::thread([c, v, sc = std::move(sketch_class)]() {
            tl_background = true;
            int rc = GROOT_OK;
            try {
                if (hipSetDevice(c->device) != hipSuccess) rc = fail(c, GROOT_E_DEVICE, "hipSetDevice");
                if (!rc && !c->bg_cancel) rc = build_prefix_tables(c, v);
                if (!rc && !c->bg_cancel) rc = build_signature_index(c, v, sc);
            } catch (const std::exception &e) {
                rc = fail(c, GROOT_E_NOSPACE, "%s", e.what());
            }
            c->bg_rc = rc;
            c->bg_state.store(rc ? 3 : (c->bg_cancel ? 4 : 2), std::memory_order_release);
        });
        return GROOT_OK;
    }
    if (int rc = build_prefix_tables(c, v)) return rc;
    lap("prefix tables");
    if (!w_exact) { if (int rc = finish_lsh()) return rc; lap("LSH forest tables (waited for)"); }
    if (int rc = build_signature_index(c, v, sketch_class)) return rc;
    if (int rc = finish_lsh()) return rc;
    lap("signature index");
    return GROOT_OK;
}

int groot_hip_open_stats(const groot_ctx *c, groot_open_stats *out)
{
    if (!c || !out) return GROOT_E_INVALID;
    memset(out, 0, sizeof *out);
    out->open_ms = c->open_ms; out->memo_ms = c->out_build_ms;
    out->memo_strings = c->out_strings; out->memo_tabulated = c->out_tabulated; out->memo_entries = c->out_entries; out->text_entries = c->text_entries;
    out->memo_hbm_bytes = c->out_tab.n * sizeof(uint4) + c->text_tab.n * sizeof(uint4) + c->sig_info.n * sizeof(uint32_t);
    return GROOT_OK;
}

int groot_hip_open(groot_ctx **out, int device_id, const groot_index_view *idx, const groot_params *p)
{
    return groot_hip_open_flags(out, device_id, idx, p, 0);
}

int groot_hip_open_abandon(groot_ctx *c)
{
    if (!c) return GROOT_E_INVALID;
    if (c->bg_state.load(std::memory_order_acquire) == 1) c->bg_cancel = true;   // (a finished build is installed by the next submit as usual)
    return GROOT_OK;
}

int groot_hip_open_wait(groot_ctx *c)
{
    if (!c) return GROOT_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    return install_background(c, true);
}

int groot_hip_open_flags(groot_ctx **out, int device_id, const groot_index_view *idx, const groot_params *p, uint32_t flags)
{
    if (!out) return fail(nullptr, GROOT_E_INVALID, "null out pointer");
    *out = nullptr;
    groot_ctx *c = new groot_ctx();
    const auto t_open0 = std::chrono::steady_clock::now();
    int rc;
    try {
        rc = open_impl(c, device_id, idx, p, flags);
    } catch (const std::bad_alloc &) {      // (host tables of the index / the memo: nothing may unwind through the C boundary)
        rc = fail(c, GROOT_E_NOSPACE, "out of host memory while building the device tables");
    } catch (const std::exception &e) {
        rc = fail(c, GROOT_E_INVALID, "groot_hip_open: %s", e.what());
    }
    c->open_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_open0).count();
    if (rc) {
        g_open_err = c->err;
        groot_hip_close(c);
        return rc;
    }
    *out = c;
    return GROOT_OK;
}

static bool idle(const groot_ctx *c)
{
    return c->inflight.empty();
}

int groot_hip_set_stream(groot_ctx *c, void *hip_stream)
{
    if (!c) return GROOT_E_INVALID;
    if (!idle(c)) return fail(c, GROOT_E_STATE, "cannot change stream while a batch is in flight");
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return GROOT_OK;
}

int groot_hip_stream_join(groot_ctx *c, void *hip_stream)
{
    if (!c) return GROOT_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : c->stream;
    // batches run in submission order on the align stream: the newest batch's event covers the ones before it
    // (the newest batch's event: a slot's event is recorded anew only by a newer batch still, and a finished one makes the wait a no-op)
    if (c->last_compute) HIP_TRY(c, hipStreamWaitEvent(st, c->last_compute, 0));
    return GROOT_OK;
}

int groot_hip_redo_status(groot_ctx *c, const uint32_t **d_status, uint32_t *redo_mask)
{
    if (!c || !d_status || !redo_mask) return GROOT_E_INVALID;
    if (!c->newest) return fail(c, GROOT_E_STATE, "no batch submitted");
    *d_status = &c->newest->d_ctr.p->flags;
    *redo_mask = kFlagSeedOverflow | kFlagTravOverflow | kFlagOvfOverflow | kFlagQOverflow;    // what finish_counters grows and redoes
    return GROOT_OK;
}

int groot_hip_set_profiling(groot_ctx *c, int enable)
{
    if (!c) return GROOT_E_INVALID;
    c->profiling = enable != 0;
    return GROOT_OK;
}

// ---- submit ---------------------------------------------------------------------------------------------------------
static int take_slot(groot_ctx *c, uint32_t n_reads, Slot **out)
{
    if (n_reads > c->prm.max_batch_reads) return fail(c, GROOT_E_NOSPACE, "batch of %u reads exceeds max_batch_reads=%u", n_reads, c->prm.max_batch_reads);
    HIP_TRY(c, hipSetDevice(c->device));
    if (int rc = progress(c)) return rc;
    Slot *s = free_slot(c);
    if (!s) return fail(c, GROOT_E_STATE, "pipeline full: %u batches submitted and not released (groot_hip_collect + groot_hip_release first)", c->prm.pipeline_depth);
    *out = s;
    return GROOT_OK;
}

// offsets must start at 0 and not decrease; *max_len = the longest read (branch-free pass so that it vectorises: 10 M reads per batch)
static int check_offsets(groot_ctx *c, const uint64_t *seq_off, uint32_t n_reads, uint32_t *max_len, uint32_t *min_len = nullptr)
{
    if (seq_off[0] != 0) return fail(c, GROOT_E_INVALID, "seq_off[0] must be 0");
    uint64_t longest = 0, shortest = ~0ULL, bad = 0;
    for (uint32_t i = 0; i < n_reads; i++) {
        bad |= (uint64_t)(seq_off[i + 1] < seq_off[i]);
        longest = std::max(longest, seq_off[i + 1] - seq_off[i]);
        shortest = std::min(shortest, seq_off[i + 1] - seq_off[i]);
    }
    if (bad) {
        uint32_t i = 0;
        while (seq_off[i + 1] >= seq_off[i]) i++;
        return fail(c, GROOT_E_INVALID, "seq_off not monotone at read %u", i);
    }
    if (seq_off[n_reads] > c->prm.max_batch_bases)
        return fail(c, GROOT_E_NOSPACE, "batch of %llu bases exceeds max_batch_bases=%llu", (unsigned long long)seq_off[n_reads], (unsigned long long)c->prm.max_batch_bases);
    *max_len = (uint32_t)std::min<uint64_t>(longest, 0xFFFFFFFFu);
    if (min_len) *min_len = (uint32_t)std::min<uint64_t>(shortest, 0xFFFFFFFFu);
    return GROOT_OK;
}

// (ten million lengths are 4.6 ms of one core: the batch period of a host-fed stream is 4.9 ms -- the scan is dealt over the granted cores)
static int check_lengths(groot_ctx *c, const uint16_t *len, uint32_t n_reads, uint64_t *total, uint32_t *max_len, uint32_t *min_len)
{
    struct Part { uint64_t sum = 0; uint32_t longest = 0, shortest = 0xFFFFu; char pad[48]; };
    auto scan = [len](uint32_t lo, uint32_t hi, Part *p) {
        uint64_t sum = 0;
        uint32_t longest = 0, shortest = 0xFFFFu;
        for (uint32_t i0 = lo; i0 < hi; i0 += 4096) {            // (32-bit partial sums: the inner loop vectorises)
            const uint32_t i1 = std::min(hi, i0 + 4096);
            uint32_t part = 0;
            for (uint32_t i = i0; i < i1; i++) { const uint32_t v = len[i]; part += v; longest = v > longest ? v : longest; shortest = v < shortest ? v : shortest; }
            sum += part;
        }
        p->sum = sum; p->longest = longest; p->shortest = shortest;
    };
    const unsigned nt = std::max(1u, (unsigned)std::min<size_t>(std::min(16u, granted_cpus()), n_reads >> 19));
    std::vector<Part> parts(nt);
    if (nt == 1) scan(0, n_reads, &parts[0]);
    else {
        std::vector<std::thread> th;
        const uint32_t per = (n_reads + nt - 1) / nt;
        for (unsigned t = 0; t < nt; t++) th.emplace_back(scan, std::min(n_reads, t * per), std::min(n_reads, (t + 1) * per), &parts[t]);
        for (auto &x : th) x.join();
    }
    uint64_t sum = 0;
    uint32_t longest = 0, shortest = 0xFFFFFFFFu;
    for (const Part &p : parts) { sum += p.sum; longest = std::max(longest, p.longest); shortest = std::min(shortest, p.shortest); }
    if (!n_reads) shortest = 0xFFFFFFFFu;
    *min_len = shortest;
    if (sum > c->prm.max_batch_bases)
        return fail(c, GROOT_E_NOSPACE, "batch of %llu bases exceeds max_batch_bases=%llu", (unsigned long long)sum, (unsigned long long)c->prm.max_batch_bases);
    *total = sum; *max_len = longest;
    return GROOT_OK;
}

static int check_exceptions(groot_ctx *c, const uint64_t *exc_pos, uint64_t n_exc, uint64_t total)
{
    uint64_t bad = 0;
    for (uint64_t i = 0; i < n_exc; i++) bad |= (uint64_t)(exc_pos[i] >= total);
    if (bad) return fail(c, GROOT_E_INVALID, "an exception position lies outside the batch");
    return GROOT_OK;
}

int groot_hip_submit(groot_ctx *c, const uint8_t *seq_concat, const uint64_t *seq_off, uint32_t n_reads, uint32_t first_read_id)
{
    if (!c) return GROOT_E_INVALID;
    if (n_reads && (!seq_concat || !seq_off)) return fail(c, GROOT_E_INVALID, "null read buffers");
    uint32_t max_len = 0, min_len = 0;
    if (n_reads) { if (int rc = check_offsets(c, seq_off, n_reads, &max_len, &min_len)) return rc; }
    Slot *s = nullptr;
    if (int rc = take_slot(c, n_reads, &s)) return rc;
    if (int rc = ensure_slot(c, s, Slot::IN_ASCII, 0)) return rc;
    s->input = Slot::IN_ASCII; s->n_reads = n_reads; s->first_read_id = first_read_id;
    s->mixed_len = min_len != max_len; s->one_len = n_reads && min_len == max_len;
    s->n_bases = n_reads ? seq_off[n_reads] : 0; s->n_exc = 0;
    s->max_len = std::min(max_len, c->prm.max_read_len);
    if (n_reads) {   // the caller's memory is not referenced after this call returns
        par_copy(s->h_bases.p, seq_concat, s->n_bases);
        par_copy(s->h_off.p, seq_off, ((size_t)n_reads + 1) * sizeof(uint64_t));
    }
    return enqueue(c, s);
}

int groot_hip_submit_packed(groot_ctx *c, const uint8_t *packed, const uint64_t *seq_off, uint32_t n_reads, uint32_t first_read_id,
                            const uint64_t *exc_pos, const uint8_t *exc_byte, uint64_t n_exc)
{
    if (!c) return GROOT_E_INVALID;
    if (n_reads && (!packed || !seq_off)) return fail(c, GROOT_E_INVALID, "null read buffers");
    if (n_exc && (!exc_pos || !exc_byte)) return fail(c, GROOT_E_INVALID, "null exception list");
    uint32_t max_len = 0, min_len = 0;
    if (n_reads) {
        if (int rc = check_offsets(c, seq_off, n_reads, &max_len, &min_len)) return rc;
        if (int rc = check_exceptions(c, exc_pos, n_exc, seq_off[n_reads])) return rc;
    }
    Slot *s = nullptr;
    if (int rc = take_slot(c, n_reads, &s)) return rc;
    if (int rc = ensure_slot(c, s, Slot::IN_PACKED, n_exc)) return rc;
    s->input = Slot::IN_PACKED; s->n_reads = n_reads; s->first_read_id = first_read_id;
    s->mixed_len = min_len != max_len; s->one_len = n_reads && min_len == max_len;
    s->n_bases = n_reads ? seq_off[n_reads] : 0; s->n_exc = n_reads ? n_exc : 0;
    s->max_len = std::min(max_len, c->prm.max_read_len);
    if (n_reads) {
        par_copy(s->h_bases.p, packed, (size_t)((s->n_bases + 3) / 4));
        par_copy(s->h_off.p, seq_off, ((size_t)n_reads + 1) * sizeof(uint64_t));
        if (n_exc) { memcpy(s->h_exc_pos.p, exc_pos, n_exc * sizeof(uint64_t)); memcpy(s->h_exc_byte.p, exc_byte, n_exc); }
    }
    return enqueue(c, s);
}

int groot_hip_submit_packed16(groot_ctx *c, const uint8_t *packed, const uint16_t *seq_len, uint32_t n_reads, uint32_t first_read_id,
                              const uint64_t *exc_pos, const uint8_t *exc_byte, uint64_t n_exc)
{
    if (!c) return GROOT_E_INVALID;
    if (n_reads && (!packed || !seq_len)) return fail(c, GROOT_E_INVALID, "null read buffers");
    if (n_exc && (!exc_pos || !exc_byte)) return fail(c, GROOT_E_INVALID, "null exception list");
    uint64_t total = 0;
    uint32_t max_len = 0, min_len = 0;
    if (n_reads) {
        if (int rc = check_lengths(c, seq_len, n_reads, &total, &max_len, &min_len)) return rc;
        if (int rc = check_exceptions(c, exc_pos, n_exc, total)) return rc;
    }
    Slot *s = nullptr;
    if (int rc = take_slot(c, n_reads, &s)) return rc;
    if (int rc = ensure_slot(c, s, Slot::IN_PACKED16, n_exc)) return rc;
    s->input = Slot::IN_PACKED16; s->n_reads = n_reads; s->first_read_id = first_read_id;
    s->n_bases = total; s->n_exc = n_reads ? n_exc : 0;
    s->max_len = std::min(max_len, c->prm.max_read_len);
    s->uniform_len = n_reads && min_len == max_len ? max_len : 0;
    s->mixed_len = n_reads && min_len != max_len; s->one_len = n_reads && min_len == max_len;
    if (n_reads) {
        par_copy(s->h_bases.p, packed, (size_t)((total + 3) / 4));
        if (!s->uniform_len) par_copy(s->h_len.p, seq_len, (size_t)n_reads * sizeof(uint16_t));
        if (n_exc) { memcpy(s->h_exc_pos.p, exc_pos, n_exc * sizeof(uint64_t)); memcpy(s->h_exc_byte.p, exc_byte, n_exc); }
    }
    return enqueue(c, s);
}

int groot_hip_acquire(groot_ctx *c, groot_batch_buffers *out)
{
    if (!c || !out) return GROOT_E_INVALID;
    Slot *s = nullptr;
    if (int rc = take_slot(c, 0, &s)) return rc;
    if (int rc = ensure_slot(c, s, Slot::IN_PACKED16, 0)) return rc;
    s->state = Slot::ACQUIRED;
    s->ticket = c->next_ticket++;
    memset(out, 0, sizeof *out);
    out->ticket = s->ticket;
    out->packed = s->h_bases.p; out->seq_len = s->h_len.p; out->exc_pos = s->h_exc_pos.p; out->exc_byte = s->h_exc_byte.p;
    out->packed_cap = (c->prm.max_batch_bases + 3) / 4; out->exc_cap = s->h_exc_pos.n; out->reads_cap = c->prm.max_batch_reads;
    return GROOT_OK;
}

static Slot *slot_by_ticket(groot_ctx *c, uint64_t ticket, Slot::State st)
{
    for (auto &s : c->slots)
        if (s->ticket == ticket && s->state == st) return s.get();
    return nullptr;
}

int groot_hip_submit_acquired(groot_ctx *c, uint64_t ticket, uint32_t n_reads, uint64_t n_exc, uint32_t first_read_id)
{
    if (!c) return GROOT_E_INVALID;
    Slot *s = slot_by_ticket(c, ticket, Slot::ACQUIRED);
    if (!s) return fail(c, GROOT_E_STATE, "ticket %llu is not an acquired batch", (unsigned long long)ticket);
    if (n_reads > c->prm.max_batch_reads) return fail(c, GROOT_E_NOSPACE, "batch of %u reads exceeds max_batch_reads=%u", n_reads, c->prm.max_batch_reads);
    if (n_exc > s->h_exc_pos.n) return fail(c, GROOT_E_NOSPACE, "more exceptions than the acquired buffers hold");
    uint64_t total = 0;
    uint32_t max_len = 0, min_len = 0;
    if (n_reads) {
        if (int rc = check_lengths(c, s->h_len.p, n_reads, &total, &max_len, &min_len)) return rc;
        if (int rc = check_exceptions(c, s->h_exc_pos.p, n_exc, total)) return rc;
    }
    s->input = Slot::IN_PACKED16; s->n_reads = n_reads; s->first_read_id = first_read_id;
    s->n_bases = total; s->n_exc = n_reads ? n_exc : 0;
    s->max_len = std::min(max_len, c->prm.max_read_len);
    s->uniform_len = n_reads && min_len == max_len ? max_len : 0;
    s->mixed_len = n_reads && min_len != max_len; s->one_len = n_reads && min_len == max_len;
    s->state = Slot::FREE;           // enqueue re-labels it
    return enqueue(c, s);
}

int groot_hip_submit_device(groot_ctx *c, const void *d_seq, const void *d_seq_off, uint32_t n_reads, uint32_t first_read_id,
                            uint32_t max_len)
{
    if (!c) return GROOT_E_INVALID;
    if (n_reads && (!d_seq || !d_seq_off)) return fail(c, GROOT_E_INVALID, "null device buffers");
    if (((uintptr_t)d_seq & 15) != 0) return fail(c, GROOT_E_INVALID, "d_seq must be 16-byte aligned");
    Slot *s = nullptr;
    if (int rc = take_slot(c, n_reads, &s)) return rc;
    if (int rc = ensure_slot(c, s, Slot::IN_DEVICE, 0)) return rc;
    s->input = Slot::IN_DEVICE; s->n_reads = n_reads; s->first_read_id = first_read_id; s->n_bases = 0; s->n_exc = 0;
    s->ext_seq = (const uint8_t *)d_seq; s->ext_off = (const uint64_t *)d_seq_off;
    const bool mixed = (max_len & GROOT_MAXLEN_MIXED) != 0;     // the caller's word: the reads differ in length
    max_len &= ~GROOT_MAXLEN_MIXED;
    s->max_len = max_len ? std::min(max_len, c->prm.max_read_len) : c->prm.max_read_len;
    s->mixed_len = mixed;                  // (else unknown: the offsets are on the device)
    s->one_len = max_len != 0 && !mixed;   // (the caller's word: the longest read, taken as THE read length when choosing kernels)
    return enqueue(c, s);
}

// ---- collect --------------------------------------------------------------------------------------------------------
int groot_hip_collect(groot_ctx *c, groot_batch_result *out)
{
    if (!c || !out) return GROOT_E_INVALID;
    Slot *s = nullptr;
    if (int rc = collect_impl(c, &s)) return rc;
    memset(out, 0, sizeof *out);
    out->ticket = s->ticket; out->first_read_id = s->first_read_id; out->n_reads = s->n_reads;
    out->counts = s->counts;
    out->n_travs = s->n_trav;
    out->travs = s->host_results ? s->h_trav.p : nullptr;
    out->masks = s->host_results ? s->h_mask.p : nullptr;
    out->mask_ckpt = s->host_results ? s->h_ckpt.p : nullptr;
    out->n_mask_bytes = s->host_results ? s->n_mask_bytes : 0;
    out->d_travs = s->d_trav.p; out->d_masks = s->d_mask.p;
    out->path_words = c->pw_view;
    out->status = s->status;
    out->ms = s->ms;
    if (s->status) return fail(c, s->status, "%s", s->status_msg.c_str());
    return GROOT_OK;
}

int groot_hip_release(groot_ctx *c, uint64_t ticket)
{
    if (!c) return GROOT_E_INVALID;
    Slot *s = slot_by_ticket(c, ticket, Slot::COLLECTED);
    if (!s) s = slot_by_ticket(c, ticket, Slot::ACQUIRED);     // an acquired batch may be abandoned
    if (!s) return fail(c, GROOT_E_STATE, "ticket %llu is not a collected batch", (unsigned long long)ticket);
    release_slot(c, s);
    return GROOT_OK;
}

int groot_hip_in_flight(groot_ctx *c, uint32_t *submitted_not_collected, uint32_t *free_slots)
{
    if (!c) return GROOT_E_INVALID;
    if (submitted_not_collected) *submitted_not_collected = (uint32_t)c->inflight.size();
    if (free_slots) {
        uint32_t n = 0;
        for (auto &s : c->slots) n += s->state == Slot::FREE || s.get() == c->waited;
        *free_slots = n;
    }
    return GROOT_OK;
}

int groot_hip_wait(groot_ctx *c, groot_counts *counts)
{
    if (!c) return GROOT_E_INVALID;
    if (c->inflight.empty()) {
        if (!c->waited) return fail(c, GROOT_E_STATE, "no batch submitted");
    } else {
        if (c->waited) release_slot(c, c->waited);
        Slot *s = nullptr;
        if (int rc = collect_impl(c, &s)) return rc;
        c->waited = s;
    }
    if (counts) *counts = c->waited->counts;
    if (c->waited->status) return fail(c, c->waited->status, "%s", c->waited->status_msg.c_str());
    return GROOT_OK;
}

int groot_hip_read_travs(groot_ctx *c, groot_trav *out, uint64_t *masks, uint64_t cap, uint64_t *n)
{
    if (!c || !n) return GROOT_E_INVALID;
    Slot *s = c->waited;
    if (!s) return fail(c, GROOT_E_STATE, "no finished batch");
    HIP_TRY(c, hipSetDevice(c->device));
    *n = s->n_trav;
    const uint64_t m = std::min<uint64_t>(cap, s->n_trav);
    if (s->host_results) {
        if (m && out) memcpy(out, s->h_trav.p, m * sizeof(groot_trav));
        if (m && masks) {        // compact path sets back to path_words words per traversal
            memset(masks, 0, m * c->pw_view * sizeof(uint64_t));
            uint64_t o = 0;
            for (uint64_t i = 0; i < m; i++) {
                const uint32_t w = c->h_graph_words[s->h_trav.p[i].graph_id];
                memcpy(masks + i * c->pw_view, s->h_mask.p + o, (size_t)w);     // (w bytes)
                o += w;
            }
        }
    } else {
        if (m && out) HIP_TRY(c, hipMemcpy(out, s->d_trav.p, m * sizeof(groot_trav), hipMemcpyDeviceToHost));
        if (m && masks) HIP_TRY(c, hipMemcpy(masks, s->d_mask.p, m * c->pw_view * sizeof(uint64_t), hipMemcpyDeviceToHost));
    }
    return GROOT_OK;
}

// seeds and sketches stay in the batch's work set: they are the waited batch's only until a newer batch runs through that set
static int work_buffers_of_waited(groot_ctx *c)
{
    Slot *s = c->waited;
    if (!s) return fail(c, GROOT_E_STATE, "no finished batch");
    if (s->n_reads && (c->ws[s->set].owner != s || c->ws[s->set].ticket != s->ticket))
        return fail(c, GROOT_E_STATE, "a newer batch has been submitted: the seeds / sketches of the waited batch are gone");
    return GROOT_OK;
}

int groot_hip_read_seeds(groot_ctx *c, groot_seed *out, uint64_t cap, uint64_t *n)
{
    if (!c || !n) return GROOT_E_INVALID;
    if (int rc = work_buffers_of_waited(c)) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    const Slot *s = c->waited;
    const uint32_t R = s->n_reads;
    std::vector<uint32_t> cnt(R), win((size_t)c->seed_slots * R);
    if (R) {
        HIP_TRY(c, hipMemcpy(cnt.data(), c->ws[s->set].seed_count.p, (size_t)R * 4, hipMemcpyDeviceToHost));
        HIP_TRY(c, hipMemcpy(win.data(), c->ws[s->set].seed_win.p, (size_t)c->seed_slots * R * 4, hipMemcpyDeviceToHost));   // [slot][R], R = this batch
    }
    // reads the text lookup answered have their seed windows in the outcome table, not in the seed slots
    std::vector<uint32_t> tidx;
    if (R && c->dix.out_tab && !c->h_out_tab.empty()) {
        tidx.resize(R);
        HIP_TRY(c, hipMemcpy(tidx.data(), c->ws[s->set].tab_idx.p, (size_t)R * 4, hipMemcpyDeviceToHost));
    }
    const size_t ed = (size_t)c->dix.out_stride_q * 4;       // dwords per entry
    uint64_t total = 0;
    std::vector<uint32_t> tmp;
    for (uint32_t r = 0; r < R; r++) {
        tmp.clear();
        if (!tidx.empty() && tidx[r] != kEmpty && (tidx[r] & kTabSeedsHere)) {
            const uint32_t *e0 = &c->h_out_tab[(size_t)(tidx[r] & ((1u << kOutIdxBits) - 1u)) * ed];
            const uint32_t n_ent = std::max(e0[3] >> 16, 1u) + (e0[2] >> 20);
            for (uint32_t e = 0; e < n_ent; e++)
                for (uint32_t x = 0; x < kOutSeedDw; x++)
                    if (e0[e * ed + ed - kOutSeedDw + x] != kEmpty) tmp.push_back(e0[e * ed + ed - kOutSeedDw + x]);
        } else {
            const uint32_t m = std::min(cnt[r] & 0x7FFFFFFFu, c->seed_slots);
            for (uint32_t j = 0; j < m; j++) tmp.push_back(win[(size_t)j * R + r]);
        }
        std::sort(tmp.begin(), tmp.end());
        for (uint32_t w : tmp) {
            if (out && total < cap) out[total] = groot_seed{s->first_read_id + r, w};
            total++;
        }
    }
    *n = total;
    return GROOT_OK;
}

int groot_hip_read_sketches(groot_ctx *c, uint64_t *out, uint64_t cap_reads, uint64_t *n_reads)
{
    if (!c || !n_reads) return GROOT_E_INVALID;
    if (int rc = work_buffers_of_waited(c)) return rc;
    if (!c->prm.keep_sketches) return fail(c, GROOT_E_STATE, "ctx was opened without keep_sketches");
    HIP_TRY(c, hipSetDevice(c->device));
    *n_reads = c->waited->n_reads;
    const uint64_t m = std::min<uint64_t>(cap_reads, c->waited->n_reads);
    if (m && out) HIP_TRY(c, hipMemcpy(out, c->ws[c->waited->set].sketches.p, m * c->s * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return GROOT_OK;
}

int groot_hip_stage_ms(groot_ctx *c, groot_stage_ms *out)
{
    if (!c || !out) return GROOT_E_INVALID;
    if (c->waited) *out = c->waited->ms;
    else memset(out, 0, sizeof *out);
    return GROOT_OK;
}

// ---- call counts ----------------------------------------------------------------------------------------------------
static int table_rows(groot_ctx *c, std::vector<uint32_t> &q_of_row)     // device sync + the current row -> kmerCount map
{
    if (int rc = drain(c)) return rc;
    uint32_t n = 0;
    HIP_TRY(c, hipMemcpy(&n, c->q_nrows.p, 4, hipMemcpyDeviceToHost));
    q_of_row.resize(n);
    if (n) HIP_TRY(c, hipMemcpy(q_of_row.data(), c->q_of_row.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return GROOT_OK;
}

int groot_hip_attempts_export(groot_ctx *c, uint32_t *q_values, uint32_t *counts, uint32_t cap_rows, uint32_t *n_rows, uint32_t *n_windows)
{
    if (!c || !n_rows) return GROOT_E_INVALID;
    std::vector<uint32_t> qs;
    if (int rc = table_rows(c, qs)) return rc;
    *n_rows = (uint32_t)qs.size();
    if (n_windows) *n_windows = c->n_windows;
    std::vector<uint32_t> order(qs.size());
    std::iota(order.begin(), order.end(), 0u);
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return qs[a] < qs[b]; });
    for (uint32_t i = 0; i < order.size() && i < cap_rows; i++) {
        if (q_values) q_values[i] = qs[order[i]];
        if (counts && c->n_windows)
            HIP_TRY(c, hipMemcpy(counts + (size_t)i * c->n_windows, c->attempts_ptr + (size_t)order[i] * c->n_windows, (size_t)c->n_windows * 4,
                                 hipMemcpyDeviceToHost));
    }
    return GROOT_OK;
}

int groot_hip_attempts_layout(groot_ctx *c, const uint32_t *q_values, uint32_t n_q, void *d_table)
{
    if (!c || (n_q && !q_values)) return GROOT_E_INVALID;
    if (!idle(c)) return fail(c, GROOT_E_STATE, "a batch is in flight");
    for (uint32_t i = 0; i < n_q; i++) {
        if (q_values[i] > c->max_q) return fail(c, GROOT_E_INVALID, "kmerCount %u exceeds max_read_len-k+1=%u", q_values[i], c->max_q);
        if (i && q_values[i] <= q_values[i - 1]) return fail(c, GROOT_E_INVALID, "kmerCounts must be strictly ascending");
    }
    std::vector<uint32_t> qs;
    if (int rc = table_rows(c, qs)) return rc;
    std::vector<uint32_t> new_row(qs.size());
    for (size_t r = 0; r < qs.size(); r++) {
        const uint32_t *p = std::lower_bound(q_values, q_values + n_q, qs[r]);
        if (p == q_values + n_q || *p != qs[r]) return fail(c, GROOT_E_INVALID, "the layout lacks kmerCount %u, which has counts", qs[r]);
        new_row[r] = (uint32_t)(p - q_values);
    }
    const uint32_t cap = std::max<uint32_t>(n_q, 1);
    DevBuf<uint32_t> own;
    uint32_t *dst = (uint32_t *)d_table;
    if (!dst) { HIP_TRY(c, own.alloc((size_t)cap * c->n_windows)); dst = own.p; }
    // (a caller re-laying out the buffer the table already lives in: its rows are staged first, or the memset below would wipe
    // them and the row moves could overlap)
    DevBuf<uint32_t> stage;
    const uint32_t *src = c->attempts_ptr;
    if (dst == c->attempts_ptr && !qs.empty()) {
        HIP_TRY(c, stage.alloc(qs.size() * (size_t)c->n_windows));
        HIP_TRY(c, hipMemcpy(stage.p, c->attempts_ptr, qs.size() * (size_t)c->n_windows * 4, hipMemcpyDeviceToDevice));
        src = stage.p;
    }
    if (n_q) HIP_TRY(c, hipMemset(dst, 0, (size_t)n_q * c->n_windows * 4));
    for (size_t r = 0; r < qs.size(); r++)
        HIP_TRY(c, hipMemcpy(dst + (size_t)new_row[r] * c->n_windows, src + r * c->n_windows, (size_t)c->n_windows * 4, hipMemcpyDeviceToDevice));
    std::vector<uint32_t> rowmap(c->max_q + 2, kEmpty);
    for (uint32_t i = 0; i < n_q; i++) rowmap[q_values[i]] = i;
    HIP_TRY(c, hipMemcpy(c->q_row.p, rowmap.data(), rowmap.size() * 4, hipMemcpyHostToDevice));
    if (n_q) HIP_TRY(c, hipMemcpy(c->q_of_row.p, q_values, (size_t)n_q * 4, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->q_nrows.p, &n_q, 4, hipMemcpyHostToDevice));
    if (d_table) {
        c->attempts.release();
        c->attempts_ptr = dst; c->att_external = true; c->att_cap = n_q;
    } else {
        std::swap(c->attempts.p, own.p); std::swap(c->attempts.n, own.n);
        c->attempts_ptr = c->attempts.p; c->att_external = false; c->att_cap = cap;
    }
    return GROOT_OK;
}

int groot_hip_attempts_import(groot_ctx *c, const uint32_t *q_values, const uint32_t *counts, uint32_t n_rows)
{
    if (!c || (n_rows && (!q_values || !counts))) return GROOT_E_INVALID;
    if (!idle(c)) return fail(c, GROOT_E_STATE, "a batch is in flight");
    if (c->att_external) return fail(c, GROOT_E_STATE, "the table lives in a caller-owned buffer");
    std::vector<uint32_t> qs;
    if (int rc = table_rows(c, qs)) return rc;
    std::vector<uint32_t> all(qs);
    all.insert(all.end(), q_values, q_values + n_rows);
    std::sort(all.begin(), all.end());
    all.erase(std::unique(all.begin(), all.end()), all.end());
    if (int rc = groot_hip_attempts_layout(c, all.data(), (uint32_t)all.size(), nullptr)) return rc;
    if (!n_rows || !c->n_windows) return GROOT_OK;
    DevBuf<uint32_t> tmp;
    HIP_TRY(c, tmp.alloc(c->n_windows));
    for (uint32_t r = 0; r < n_rows; r++) {
        const uint32_t row = (uint32_t)(std::lower_bound(all.begin(), all.end(), q_values[r]) - all.begin());
        HIP_TRY(c, hipMemcpy(tmp.p, counts + (size_t)r * c->n_windows, (size_t)c->n_windows * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(add_u32_kernel, dim3((unsigned)std::min<size_t>((c->n_windows + kBlock - 1) / kBlock, 65535)), dim3(kBlock), 0, c->stream,
                           c->attempts_ptr + (size_t)row * c->n_windows, tmp.p, (size_t)c->n_windows);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return GROOT_OK;
}

int groot_hip_attempts_device(groot_ctx *c, void **d_table, uint32_t *n_rows, uint32_t *n_windows)
{
    if (!c || !d_table) return GROOT_E_INVALID;
    std::vector<uint32_t> qs;
    if (int rc = table_rows(c, qs)) return rc;
    *d_table = c->attempts_ptr;
    if (n_rows) *n_rows = (uint32_t)qs.size();
    if (n_windows) *n_windows = c->n_windows;
    return GROOT_OK;
}

int groot_hip_attempts_reset(groot_ctx *c)
{
    if (!c) return GROOT_E_INVALID;
    if (int rc = drain(c)) return rc;
    if (c->att_cap) HIP_TRY(c, hipMemset(c->attempts_ptr, 0, (size_t)c->att_cap * c->n_windows * sizeof(uint32_t)));
    return GROOT_OK;
}

int groot_hip_attempts_shape(groot_ctx *c, uint32_t *n_q, uint32_t *n_windows)
{
    if (!c) return GROOT_E_INVALID;
    if (n_q) *n_q = c->max_q + 1;
    if (n_windows) *n_windows = c->n_windows;
    return GROOT_OK;
}

int groot_hip_attempts_read(groot_ctx *c, uint32_t *out, uint64_t n_elems)
{
    if (!c || !out) return GROOT_E_INVALID;
    const uint64_t have = (uint64_t)(c->max_q + 1) * c->n_windows;
    if (n_elems < have) return fail(c, GROOT_E_NOSPACE, "need room for %llu counts", (unsigned long long)have);
    std::vector<uint32_t> qs;
    if (int rc = table_rows(c, qs)) return rc;
    memset(out, 0, have * sizeof(uint32_t));
    for (size_t r = 0; r < qs.size(); r++)
        if (c->n_windows)
            HIP_TRY(c, hipMemcpy(out + (size_t)qs[r] * c->n_windows, c->attempts_ptr + r * c->n_windows, (size_t)c->n_windows * 4, hipMemcpyDeviceToHost));
    return GROOT_OK;
}

// ---- the one exchange of a multi-GPU run ----------------------------------------------------------------------------
namespace {
// RCCL is loaded on first use: a single-GPU run never touches it
struct Rccl {
    void *lib = nullptr;
    int (*CommInitAll)(void **, int, const int *) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool load()
    {
        if (lib) return true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) return false;
        CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
        GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
        AllReduce = (decltype(AllReduce))dlsym(lib, "ncclAllReduce");
        GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
        return CommInitAll && CommDestroy && GroupStart && GroupEnd && AllReduce;
    }
};
Rccl g_rccl;
constexpr int kNcclUint32 = 3, kNcclSum = 0;    // ncclDataType_t / ncclRedOp_t values of rccl.h
} // namespace

int groot_hip_attempts_allreduce(groot_ctx *const *ctxs, int n_ctx)
{
    if (!ctxs || n_ctx <= 0) return GROOT_E_INVALID;
    groot_ctx *c0 = ctxs[0];
    for (int i = 0; i < n_ctx; i++) {
        if (!ctxs[i]) return GROOT_E_INVALID;
        if (ctxs[i]->n_windows != c0->n_windows || ctxs[i]->max_q != c0->max_q) return fail(c0, GROOT_E_INVALID, "ctxs were opened on different indexes / read length limits");
        if (ctxs[i]->att_external) return fail(c0, GROOT_E_STATE, "ctx %d keeps its table in a caller-owned buffer", i);
    }
    // GROOT_FORCE_RCCL=1: take the RCCL branch even when all ctxs share one device (a communicator over a single device is legal):
    // the one way to run dlopen, the symbol lookups, the enum values and the grouped in-place ncclAllReduce on a one-GPU box
    const bool force_rccl = c0->kn.force_rccl;
    if (n_ctx == 1 && !force_rccl) return drain(c0);
    // union row layout (ascending kmerCount) on every ctx
    std::vector<uint32_t> all;
    for (int i = 0; i < n_ctx; i++) {
        std::vector<uint32_t> qs;
        if (int rc = table_rows(ctxs[i], qs)) return fail(c0, rc, "%s", ctxs[i]->err.c_str());
        all.insert(all.end(), qs.begin(), qs.end());
    }
    std::sort(all.begin(), all.end());
    all.erase(std::unique(all.begin(), all.end()), all.end());
    for (int i = 0; i < n_ctx; i++) {
        HIP_TRY(c0, hipSetDevice(ctxs[i]->device));
        if (int rc = groot_hip_attempts_layout(ctxs[i], all.data(), (uint32_t)all.size(), nullptr)) return fail(c0, rc, "%s", ctxs[i]->err.c_str());
    }
    const size_t count = all.size() * (size_t)c0->n_windows;
    if (!count) return GROOT_OK;
    // ctxs sharing a device (tests; several ctxs per GPU): fold them into the first ctx of that device with a kernel
    std::vector<int> lead;                       // one ctx index per distinct device
    for (int i = 0; i < n_ctx; i++) {
        int l = -1;
        for (int j : lead) if (ctxs[j]->device == ctxs[i]->device) l = j;
        if (l < 0) { lead.push_back(i); continue; }
        HIP_TRY(c0, hipSetDevice(ctxs[i]->device));
        hipLaunchKernelGGL(add_u32_kernel, dim3((unsigned)std::min<size_t>((count + kBlock - 1) / kBlock, 65535)), dim3(kBlock), 0, ctxs[l]->stream,
                           ctxs[l]->attempts_ptr, ctxs[i]->attempts_ptr, count);
        HIP_TRY(c0, hipGetLastError());
        HIP_TRY(c0, hipStreamSynchronize(ctxs[l]->stream));
    }
    if (lead.size() > 1 || force_rccl) {
        // one RCCL communicator over the distinct devices, one in-place ncclAllReduce(sum, uint32) per device: ring over xGMI
        if (!g_rccl.load()) return fail(c0, GROOT_E_DEVICE, "librccl.so could not be loaded: %s", dlerror());
        std::vector<int> devs;
        for (int j : lead) devs.push_back(ctxs[j]->device);
        std::vector<void *> comms(lead.size(), nullptr);
        int nrc = g_rccl.CommInitAll(comms.data(), (int)devs.size(), devs.data());
        if (nrc) return fail(c0, GROOT_E_DEVICE, "ncclCommInitAll: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(nrc) : "error");
        nrc = g_rccl.GroupStart();
        for (size_t j = 0; j < lead.size() && !nrc; j++) {
            groot_ctx *c = ctxs[lead[j]];
            (void)hipSetDevice(c->device);
            nrc = g_rccl.AllReduce(c->attempts_ptr, c->attempts_ptr, count, kNcclUint32, kNcclSum, comms[j], c->stream);
        }
        const int erc = g_rccl.GroupEnd();
        if (!nrc) nrc = erc;
        for (size_t j = 0; j < lead.size(); j++) {
            (void)hipSetDevice(ctxs[lead[j]]->device);
            (void)hipStreamSynchronize(ctxs[lead[j]]->stream);
        }
        for (void *cm : comms) if (cm) (void)g_rccl.CommDestroy(cm);
        if (nrc) return fail(c0, GROOT_E_DEVICE, "ncclAllReduce: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(nrc) : "error");
    }
    // every ctx ends up with the totals
    for (int i = 0; i < n_ctx; i++) {
        int l = -1;
        for (int j : lead) if (ctxs[j]->device == ctxs[i]->device) l = j;
        if (l == i) continue;
        HIP_TRY(c0, hipSetDevice(ctxs[i]->device));
        HIP_TRY(c0, hipMemcpy(ctxs[i]->attempts_ptr, ctxs[l]->attempts_ptr, count * 4, hipMemcpyDeviceToDevice));
    }
    return GROOT_OK;
}

int groot_hip_sketch(groot_ctx *c, const uint8_t *seq_concat, const uint64_t *seq_off, uint32_t n, uint64_t *out)
{
    if (!c || !out || (n && (!seq_concat || !seq_off))) return GROOT_E_INVALID;
    if (!idle(c)) return fail(c, GROOT_E_STATE, "a batch is in flight");
    if (!n) return GROOT_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    if (n > c->prm.max_batch_reads) return fail(c, GROOT_E_NOSPACE, "more sequences than max_batch_reads");
    const uint64_t total = seq_off[n];
    uint32_t max_len = 0;
    for (uint32_t i = 0; i < n; i++) max_len = std::max<uint32_t>(max_len, (uint32_t)(seq_off[i + 1] - seq_off[i]));
    DevBuf<uint64_t> sk, off;
    DevBuf<uint8_t> seq;
    DevBuf<DeviceCounters> ctr;
    HIP_TRY(c, sk.alloc((size_t)n * c->s));
    HIP_TRY(c, off.alloc((size_t)n + 1));
    HIP_TRY(c, seq.alloc(total + 64));
    HIP_TRY(c, ctr.alloc(1));
    HIP_TRY(c, hipMemcpyAsync(seq.p, seq_concat, total, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(off.p, seq_off, ((size_t)n + 1) * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemsetAsync(ctr.p, 0, sizeof(DeviceCounters), c->stream));
    SeedArgs a{};
    a.ix = c->dix;
    a.ix.max_q = 0;   // no lookup: every read gets min_eq = S+1
    a.seq = seq.p; a.seq_off = off.p; a.n_reads = n; a.max_read_len = c->prm.max_read_len;
    a.lds_read_bytes = (uint32_t)std::min<uint64_t>((uint64_t)kBlock * std::min(max_len, c->prm.max_read_len) + 32, kMaxLdsReadBytes);
    a.seed_slots = c->seed_slots; a.seed_count = c->ws[0].seed_count.p; a.seed_win = c->ws[0].seed_win.p;
    a.sketch_out = sk.p; a.sort_key = nullptr; a.read_rec = nullptr; a.q_seen = nullptr; a.ctr = ctr.p; a.shards = c->seed_shards.p;
    launch_seed(c->s, c->max_k, a, true, dim3((n + kBlock - 1) / kBlock), kLdsReads + ((a.lds_read_bytes + 15) & ~15u), c->stream);
    HIP_TRY(c, hipGetLastError());
    DeviceCounters h{};
    HIP_TRY(c, hipMemcpyAsync(&h, ctr.p, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(out, sk.p, (size_t)n * c->s * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->ws[0].owner = nullptr;      // its seed slots were used as scratch
    if (h.flags & kFlagShortRead) return fail(c, GROOT_E_SHORT_READ, "k size is greater than sequence length");
    if (h.flags & kFlagLongRead) return fail(c, GROOT_E_NOSPACE, "a sequence is longer than max_read_len=%u", c->prm.max_read_len);
    return GROOT_OK;
}

} // extern "C"

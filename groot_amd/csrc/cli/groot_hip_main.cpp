// groot-hip -- flag-compatible `index` and `align` subcommands on top of libgroot_host.so / libgroot_hip.so.
//
//   groot-hip index -m <msaDir> -i <indexDir> [-k 31 -s 21 -w 100 -x 8 -y 4 --maxSketchSpan 30 -p N --log F]
//        cmd/index.go:44-133: writes <indexDir>/groot.gidx (flat index) and the reference's groot.gg + groot.lshe
//   groot-hip align -i <indexDir> -f a.fq[,b.fq.gz] [-t 0.99 -c 1.0 -g <graphDir> --noAlign -p N --log F] > out.bam
//        cmd/align.go:30-197 + src/pipeline/sketch.go (DataStreamer..GraphPruner): BAM on stdout, weighted GFAs in
//        graphDir, the reference's log lines in --log (default groot.log)
//
// Extra flags: --gpu <id> (device) or --gpus <N> (reads shard over N GPUs), --batch <reads> (reads per device batch),
// --maxReadLen, --bam <file> (Info.Sketch.BAMout), --bamLevel, --stats <json>; index: --writeGob.
// The align hot path runs only on the GPU: no device -> error, never a CPU fallback.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <ctime>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <sys/stat.h>
#include <unistd.h>
#include <vector>

#include "groot_hip.h"

namespace {

FILE *g_log = nullptr;

void logf(const char *fmt, ...)
{
    char ts[32];
    time_t now = time(nullptr);
    struct tm tmv;
    localtime_r(&now, &tmv);
    strftime(ts, sizeof ts, "%Y/%m/%d %H:%M:%S", &tmv);   // Go's log.LstdFlags
    fprintf(g_log ? g_log : stderr, "%s ", ts);
    va_list ap;
    va_start(ap, fmt);
    vfprintf(g_log ? g_log : stderr, fmt, ap);
    va_end(ap);
    fputc('\n', g_log ? g_log : stderr);
    fflush(g_log ? g_log : stderr);
}

[[noreturn]] void die(const char *fmt, ...)   // misc.ErrorCheck -> log.Fatalf
{
    char buf[2048];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    logf("%s", buf);
    if (g_log) fprintf(stderr, "%s\n", buf);
    fflush(nullptr);
    _exit(1);                                           // (parser / mapper / HIP threads may be running: no static destructors under their feet)
}

struct Args {
    std::string cmd, index_dir, msa_dir, log_file = "groot.log", graph_dir, bam_out, bam_file;
    double cov_cutoff = 0.97;
    bool low_cov = false;
    std::vector<std::string> fastq;
    int proc = 1, gpu = 0, gpus = 0, ctx_per_gpu = 1, bam_level = -1;
    bool gpu_given = false, write_gob = false;
    uint32_t k = 31, s = 21, w = 100, x = 8, y = 4, max_span = 30, batch = 1u << 20, max_read_len = 512, depth = 3;
    uint64_t block_bytes = 0;
    std::string stats_file;
    double threshold = 0.99, min_kmer_cov = 1.0;
    bool no_align = false, fasta = false;
    std::string memo = "auto";             // auto | on | off | <MiB>
};

std::vector<std::string> split(const std::string &s, char d)
{
    std::vector<std::string> out;
    size_t a = 0;
    for (;;) {
        size_t b = s.find(d, a);
        if (b == std::string::npos) { if (a < s.size()) out.push_back(s.substr(a)); break; }
        if (b > a) out.push_back(s.substr(a, b - a));
        a = b + 1;
    }
    return out;
}

void usage()
{
    fprintf(stderr,
            "groot-hip %s (MI355X-native groot align hot path)\n\n"
            "  groot-hip index -m <msaDir> -i <indexDir> [-k 31] [-s 21] [-w 100] [-x 8] [-y 4] [--maxSketchSpan 30] [-p N] [--log F]\n"
            "                  [--gpu 0]      (sketch the graph windows on that GPU instead of the host)\n"
            "                  [--writeGob]   (also write the reference's groot.gg + groot.lshe: experimental, unpinned against a Go-written file)\n"
            "  groot-hip align -i <indexDir> -f <fastq>[,<fastq>...] [-t 0.99] [-c 1.0] [-g <graphDir>] [--noAlign] [-p N] [--log F]\n"
            "                  [--gpu 0 | --gpus N] [--batch 1048576] [--maxReadLen 512] [--bam out.bam] [--bamLevel -2..9] [--stats f.json]\n"
            "                  [--memo auto|on|off|<MiB>]   (the device's memo of indexed strings; auto: on for inputs of 20 GB and more)\n"
            "                  (BAM goes to stdout unless --bam; --gpus N shards the reads over N GPUs, index replicated)\n"
            "  groot-hip report [--bamFile x.bam] [-c 0.97] [--lowCov] [--log F]      (BAM from stdin unless --bamFile)\n",
            groot_host_version());
}

Args parse(int argc, char **argv)
{
    Args a;
    if (argc < 2) { usage(); exit(1); }
    a.cmd = argv[1];
    auto need = [&](int &i) -> const char * {
        if (i + 1 >= argc) { fprintf(stderr, "flag needs an argument: %s\n", argv[i]); exit(1); }
        return argv[++i];
    };
    for (int i = 2; i < argc; i++) {
        std::string f = argv[i];
        std::string val;
        size_t eq = f.find('=');
        bool has_val = false;
        if (f.rfind("--", 0) == 0 && eq != std::string::npos) { val = f.substr(eq + 1); f = f.substr(0, eq); has_val = true; }
        auto v = [&]() -> std::string { return has_val ? val : std::string(need(i)); };
        if (f == "-i" || f == "--indexDir") a.index_dir = v();
        else if (f == "-m" || f == "--msaDir") a.msa_dir = v();
        else if (f == "--log") a.log_file = v();
        else if (f == "-p" || f == "--processors") a.proc = atoi(v().c_str());
        else if (f == "-k" || f == "--kmerSize") a.k = (uint32_t)atoi(v().c_str());
        else if (f == "-s" || f == "--sketchSize") a.s = (uint32_t)atoi(v().c_str());
        else if (f == "-w" || f == "--windowSize") a.w = (uint32_t)atoi(v().c_str());
        else if (f == "-x" || f == "--numPart") a.x = (uint32_t)atoi(v().c_str());
        else if (f == "-y" || f == "--maxK") a.y = (uint32_t)atoi(v().c_str());
        else if (f == "--maxSketchSpan") a.max_span = (uint32_t)atoi(v().c_str());
        else if (f == "-f" || f == "--fastq") { for (auto &x : split(v(), ',')) a.fastq.push_back(x); }
        else if (f == "-t" || f == "--contThresh") a.threshold = atof(v().c_str());
        else if (a.cmd == "report" && (f == "-c" || f == "--covCutoff")) a.cov_cutoff = atof(v().c_str());
        else if (f == "--bamFile") a.bam_file = v();
        else if (f == "--lowCov") a.low_cov = true;
        else if (f == "-c" || f == "--minKmerCov") a.min_kmer_cov = atof(v().c_str());
        else if (f == "-g" || f == "--graphDir") a.graph_dir = v();
        else if (f == "--noAlign") a.no_align = true;
        else if (f == "--fasta") a.fasta = true;
        else if (f == "--profiling") {}
        else if (f == "--gpu") { a.gpu = atoi(v().c_str()); a.gpu_given = true; }
        else if (f == "--batch") a.batch = (uint32_t)atol(v().c_str());
        else if (f == "--gpus") a.gpus = atoi(v().c_str());
        else if (f == "--ctxPerGpu") a.ctx_per_gpu = std::max(1, atoi(v().c_str()));
        else if (f == "--maxReadLen") a.max_read_len = (uint32_t)atol(v().c_str());
        else if (f == "--depth") a.depth = (uint32_t)atol(v().c_str());
        else if (f == "--bamLevel") a.bam_level = atoi(v().c_str());
        else if (f == "--blockBytes") a.block_bytes = (uint64_t)atoll(v().c_str());
        else if (f == "--stats") a.stats_file = v();
        else if (f == "--writeGob") a.write_gob = true;
        else if (f == "--bam") a.bam_out = v();
        else if (f == "--memo") a.memo = v();
        else if (f == "-h" || f == "--help") { usage(); exit(0); }
        else { fprintf(stderr, "unknown flag: %s\n", f.c_str()); usage(); exit(1); }
    }
    return a;
}

bool is_dir(const std::string &p)
{
    struct stat st;
    return stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode);
}
bool is_file(const std::string &p)
{
    struct stat st;
    return stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode);
}
void make_dir(const std::string &p)
{
    if (!is_dir(p) && mkdir(p.c_str(), 0700) != 0) die("can't create specified output directory");
}

void start_logging(const Args &a)
{
    if (!a.log_file.empty()) {
        g_log = fopen(a.log_file.c_str(), "w");
        if (!g_log) { fprintf(stderr, "can't open log file %s\n", a.log_file.c_str()); exit(1); }
    }
}

double seconds_since(std::chrono::steady_clock::time_point t0)
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// ---------------------------------------------------------------------------------------------
int run_index(const Args &a)   // cmd/index.go:57-133
{
    if (a.index_dir.empty()) { puts("please specify a directory for the index files (--indexDir)"); return 1; }
    if (a.msa_dir.empty()) { fprintf(stderr, "required flag(s) \"msaDir\" not set\n"); return 1; }
    start_logging(a);
    auto t0 = std::chrono::steady_clock::now();
    logf("i am groot (version %s)", groot_host_version());
    logf("starting the index subcommand");
    logf("checking parameters...");
    logf("\tdirectory containing MSA files: %s", a.msa_dir.c_str());
    if (!is_dir(a.msa_dir)) die("no directory found at %s", a.msa_dir.c_str());
    if (a.k > a.w) die("supplied k-mer size greater than read length");
    make_dir(a.index_dir);
    logf("\tprocessors: %d", a.proc);
    logf("\tk-mer size: %u", a.k);
    logf("\tsketch size: %u", a.s);
    logf("\tgraph window size: %u", a.w);
    logf("\tnum. partitions: %u", a.x);
    logf("\tmax. K: %u", a.y);
    logf("\tmax. sketch span: %u", a.max_span);
    logf("creating graphs, sketching traversals and indexing...");
    groot_index_params p;
    groot_index_params_default(&p);
    p.kmer_size = a.k; p.sketch_size = a.s; p.window_size = a.w; p.num_part = a.x; p.max_k = a.y; p.max_sketch_span = a.max_span;
    p.n_threads = a.proc > 0 ? (uint32_t)a.proc : 0;
    groot_index *idx = nullptr;
    if (a.gpu_given) {
        // window sketches on the GPU: a ctx opened on an index view without graphs is a pure RunMinHash engine
        logf("\tsketching graph windows on GPU %d", a.gpu);
        groot_index_view empty;
        memset(&empty, 0, sizeof empty);
        empty.kmer_size = a.k; empty.sketch_size = a.s; empty.window_size = a.w; empty.num_part = a.x; empty.max_k = a.y;
        empty.num_window_kmers = a.w - a.k + 1; empty.path_words = 1;
        groot_params prm;
        groot_params_default(&prm);
        prm.max_read_len = std::max<uint32_t>(a.w, 64);
        prm.max_batch_reads = 1u << 16;
        groot_ctx *ctx = nullptr;
        if (groot_hip_open(&ctx, a.gpu, &empty, &prm)) die("%s", groot_hip_last_error(nullptr));
        struct Sk { groot_ctx *ctx; uint32_t s, max_n; } sk{ctx, a.s, prm.max_batch_reads};
        auto fn = [](void *user, const uint8_t *seq, const uint64_t *off, uint32_t n, uint64_t *out) -> int {
            Sk *k = (Sk *)user;
            std::vector<uint64_t> rel;
            for (uint32_t i = 0; i < n; i += k->max_n) {
                const uint32_t m = std::min(k->max_n, n - i);
                rel.resize(m + 1);
                for (uint32_t j = 0; j <= m; j++) rel[j] = off[i + j] - off[i];
                if (groot_hip_sketch(k->ctx, seq + off[i], rel.data(), m, out + (size_t)i * k->s)) return -1;
            }
            return 0;
        };
        const int rc = groot_index_build_msa_dir_with(a.msa_dir.c_str(), &p, fn, &sk, &idx);
        if (rc) die("%s (%s)", groot_host_last_error(), groot_hip_last_error(ctx));
        groot_hip_close(ctx);
    } else if (groot_index_build_msa_dir(a.msa_dir.c_str(), &p, &idx)) die("%s", groot_host_last_error());
    groot_index_view v;
    groot_index_get_view(idx, &v);
    uint32_t masked = 0;
    for (uint32_t g = 0; g < v.n_graphs; g++) masked += v.graph_masked[g];
    logf("\tnumber of groot graphs built: %u", v.n_graphs);
    logf("\t\tgraphs sketched: %u", v.n_graphs - masked);
    logf("\tnumber of sketches added to the LSH Ensemble index: %u", v.n_windows);
    const std::string out = a.index_dir + "/groot.gidx";
    logf("writing index files in \"%s\"...", a.index_dir.c_str());
    if (groot_index_save(idx, out.c_str())) die("%s", groot_host_last_error());
    // the reference's own files (cmd/index.go:130-131) only on request: no Go-written groot.gg / groot.lshe has been
    // round-tripped yet, so a layout or hash-constant slip would make the reference mis-seed silently
    if (a.write_gob) {
        logf("\twriting groot.gg + groot.lshe (experimental: not yet checked against files written by the reference)");
        if (groot_index_save_gob(idx, a.index_dir.c_str(), a.max_span)) die("%s", groot_host_last_error());
    }
    groot_index_free(idx);
    logf("finished in %.3fs", seconds_since(t0));
    return 0;
}

// ---------------------------------------------------------------------------------------------
// align: parse | map | write as three overlapping stages.
//   producer thread   groot_reads_next: FASTQ text -> parsed + packed batches (reader thread per file, parse over -p cores)
//   mapper threads    one per GPU: groot_hip_submit_packed16 / groot_hip_collect, several batches in flight per ctx
//   writer thread     batches back in input order: traversal records -> sam.Records -> BGZF over -p cores
// The reference's pipeline has the same shape with goroutines and channels (DataStreamer -> FastqHandler -> ReadMapper with
// its bamwriter goroutine, sketch.go:41-350, boss.go:86-104); reads shard over the GPUs batch by batch, the index is
// replicated, and the only exchange is the sum of the IncrementSubPath call counts at the end (RCCL).
struct WorkItem {
    uint64_t seq = 0;                 // position of the batch in the input
    groot_reads_batch *batch = nullptr;
    groot_reads_view view{};
    // filled by the mapper
    int gpu = -1;
    groot_batch_result res{};
};

template <class T> struct BoundedQueue {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<T> q;
    size_t cap;
    bool closed = false;
    explicit BoundedQueue(size_t c) : cap(c) {}
    void push(T v)
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return q.size() < cap; });
        q.push_back(std::move(v));
        cv.notify_all();
    }
    // 1 = got one, 0 = none right now (only when !block), -1 = closed and empty
    int pop(T &out, bool block)
    {
        std::unique_lock<std::mutex> lk(mu);
        if (block) cv.wait(lk, [&] { return !q.empty() || closed; });
        if (q.empty()) return closed ? -1 : 0;
        out = std::move(q.front());
        q.pop_front();
        cv.notify_all();
        return 1;
    }
    void close()
    {
        std::lock_guard<std::mutex> lk(mu);
        closed = true;
        cv.notify_all();
    }
};

struct Gpu {
    int device = 0;
    groot_ctx *ctx = nullptr;
    uint32_t max_read_len = 0;
    std::mutex mu;                    // tickets the writer is done with (the ctx itself belongs to the mapper thread)
    std::condition_variable cv;
    std::vector<uint64_t> done_tickets;
    uint32_t held = 0, inflight = 0;
    std::deque<WorkItem> pending;     // submitted, in order
};

int run_align(const Args &a)   // cmd/align.go:54-163
{
    if (a.index_dir.empty()) { puts("please specify a directory with the index files (--indexDir)"); return 1; }
    if (a.fasta) { fprintf(stderr, "--fasta is an experimental reference feature that is not supported\n"); return 1; }
    start_logging(a);
    auto t0 = std::chrono::steady_clock::now();
    logf("i am groot (version %s)", groot_host_version());
    logf("starting the sketch subcommand");
    logf("checking parameters...");
    for (auto &f : a.fastq) {
        if (!is_file(f)) die("no file found at %s", f.c_str());
        static const char *exts[] = {"fastq", "fq", "fasta", "fna", "fa"};   // misc.CheckExt (cmd/align.go:175)
        std::string base = f;
        if (base.size() > 3 && base.compare(base.size() - 3, 3, ".gz") == 0) base.resize(base.size() - 3);
        size_t dot = base.rfind('.');
        bool ok = false;
        for (auto e : exts) ok |= dot != std::string::npos && base.substr(dot + 1) == e;
        if (!ok) die("file does not have recognised extension: %s", f.c_str());
    }
    if (a.fastq.empty()) logf("\tinput file: using STDIN");
    if (!is_dir(a.index_dir)) die("no directory found at %s", a.index_dir.c_str());
    const std::string gidx = a.index_dir + "/groot.gidx";
    // an index directory of the reference itself (cmd/align.go:181-182: groot.gg + groot.lshe) is read through the gob reader
    const std::string gg = a.index_dir + "/groot.gg", lshe = a.index_dir + "/groot.lshe";
    const bool have_gob = is_file(gg) && is_file(lshe);
    if (!is_file(gidx) && !have_gob) die("no file found at %s (nor groot.gg + groot.lshe)", gidx.c_str());
    std::string graph_dir = a.graph_dir;
    if (graph_dir.empty()) {   // cmd/align.go:24: ./groot-graphs-<timestamp>
        char ts[32];
        time_t now = time(nullptr);
        struct tm tmv;
        localtime_r(&now, &tmv);
        strftime(ts, sizeof ts, "%Y%m%d%H%M%S", &tmv);
        graph_dir = std::string("./groot-graphs-") + ts;
    }
    make_dir(graph_dir);
    logf("\tminimum k-mer coverage: %.0f", a.min_kmer_cov);
    logf("\tprocessors: %d", a.proc);
    for (auto &f : a.fastq) logf("\tinput file: %s", f.c_str());
    // ---- the FASTQ parser starts NOW, before the index is read and the GPU context opened: inflating and packing the first batches
    // takes as long as those do (a gzip FASTQ inflates on one thread, as bufio over gzip.Reader does in the reference), and neither
    // needs the other; up to kParsedAhead batches wait for the mappers ----
    constexpr size_t kParsedAhead = 8;
    const uint32_t cores = a.proc > 0 ? (uint32_t)a.proc : 0;
    const uint64_t max_batch_bases = (uint64_t)a.batch * std::min<uint32_t>(a.max_read_len, 512);
    std::vector<const char *> files;
    for (auto &f : a.fastq) files.push_back(f.c_str());
    groot_reads *reads = nullptr;
    if (groot_reads_open(files.empty() ? nullptr : files.data(), (uint32_t)files.size(), cores, a.block_bytes, a.batch, max_batch_bases, &reads))
        die("%s", groot_host_last_error());
    std::vector<std::unique_ptr<Gpu>> gpus;
    std::atomic<bool> gpus_ready{false};
    BoundedQueue<WorkItem> parsed(kParsedAhead);
    BoundedQueue<WorkItem> mapped(4);                 // (its real capacity is set once the contexts are known, before anyone uses it)
    std::mutex fatal_mu;
    std::string fatal;
    std::atomic<bool> failed{false};
    auto fail_with = [&](const std::string &msg) {
        std::lock_guard<std::mutex> lk(fatal_mu);
        if (fatal.empty()) fatal = msg;
        failed = true;
        parsed.close(); mapped.close();
        if (gpus_ready)
            for (auto &g : gpus) { std::lock_guard<std::mutex> l2(g->mu); g->cv.notify_all(); }
    };

    std::atomic<uint64_t> length_total{0};
    double parse_s = 0, bam_s = 0;                    // busy time of the producer / the writer
    std::atomic<uint64_t> collect_wait_us{0}, n_batches{0};
    std::thread producer([&]() {
        uint64_t seq = 0;
        for (;;) {
            if (failed) break;
            WorkItem w;
            auto tp = std::chrono::steady_clock::now();
            const int prc = groot_reads_next(reads, &w.batch);
            parse_s += seconds_since(tp);
            if (prc) { fail_with(groot_host_last_error()); break; }
            if (!w.batch) break;
            groot_reads_batch_view(w.batch, &w.view);
            length_total += w.view.n_bases;
            w.seq = seq++;
            parsed.push(std::move(w));
        }
        parsed.close();
    });
    logf("loading the index information...");
    // (the HIP runtime starts up on a thread of its own while the index is read: a few tenths of a second each)
    int n_dev = 0, dev_rc = 0;
    std::thread hip_init([&]() { dev_rc = groot_hip_device_count(&n_dev); });
    groot_index *idx = nullptr;
    const int load_rc = is_file(gidx) ? groot_index_load(gidx.c_str(), &idx) : groot_index_load_gob(gg.c_str(), lshe.c_str(), &idx);
    hip_init.join();
    if (load_rc) die("%s", groot_host_last_error());
    groot_index_view v;
    groot_index_get_view(idx, &v);
    logf("\tk-mer size: %u", v.kmer_size);
    logf("\tsketch size: %u", v.sketch_size);
    logf("\twindow size used in indexing: %u", v.window_size);
    logf("loading the graphs...");
    logf("\tnumber of variation graphs: %u", v.n_graphs);
    logf("rebuilding the LSH Ensemble...");
    // ---- one ctx per GPU (index replicated), opened concurrently ----
    if (dev_rc || n_dev == 0) die("no HIP device available (groot-hip align has no CPU fallback): %s", groot_hip_last_error(nullptr));
    std::vector<int> devices;
    if (a.gpus > 0) {
        if (a.gpus > n_dev) die("--gpus %d but only %d device(s) visible", a.gpus, n_dev);
        for (int i = 0; i < a.gpus; i++) devices.push_back(i);
    } else devices.push_back(a.gpu);
    for (int extra = 1; extra < a.ctx_per_gpu; extra++)          // test hook: several ctxs on one device (exercises the N>1 path on a one-GPU box)
        for (size_t i = 0, n = devices.size() / (size_t)extra; i < n; i++) devices.push_back(devices[i]);
    const uint32_t depth = std::max(2u, a.depth);
    // The memo of groot_hip_open answers reads that equal an indexed string without hashing or graph walk: it costs a third
    // of a second per GB of path bases at open and saves ~0.4 ms of GPU time per million such reads -- in this process the GPU waits for
    // the FASTQ parser and the BAM writer, so it only pays on inputs that keep it busy for minutes.  auto: on from 20 GB of
    // input (gzip counted four-fold; stdin: off).
    uint32_t memo_budget = GROOT_MEMO_OFF;
    if (a.memo == "on") memo_budget = 0;
    else if (a.memo == "auto") {
        uint64_t bytes = 0;
        for (auto &f : a.fastq) {
            struct stat st;
            if (stat(f.c_str(), &st) == 0) bytes += (uint64_t)st.st_size * (f.size() > 3 && f.compare(f.size() - 3, 3, ".gz") == 0 ? 4 : 1);
        }
        if (bytes >= (20ull << 30)) memo_budget = 0;
    } else if (a.memo != "off") {
        char *end = nullptr;
        const long mb = strtol(a.memo.c_str(), &end, 10);
        if (a.memo.empty() || *end || mb < 1 || mb > (1L << 30)) {
            die("--memo: '%s' is neither auto, on, off nor a number of MiB", a.memo.c_str());
        }
        memo_budget = (uint32_t)mb;
    }
    auto params_for = [&](uint32_t max_read_len) {
        groot_params prm;
        groot_params_default(&prm);
        prm.containment_threshold = a.threshold;
        prm.no_exact_align = a.no_align ? 1 : 0;
        prm.max_batch_reads = a.batch;
        prm.max_read_len = max_read_len;
        prm.max_batch_bases = (uint64_t)a.batch * std::min<uint32_t>(max_read_len, 512);
        prm.pipeline_depth = depth;
        prm.memo_budget_mb = memo_budget;
        return prm;
    };
    for (int d : devices) {
        std::unique_ptr<Gpu> g(new Gpu());
        g->device = d; g->max_read_len = a.max_read_len;
        gpus.push_back(std::move(g));
    }
    {
        std::vector<std::thread> th;
        std::vector<std::string> errs(gpus.size());
        for (size_t i = 0; i < gpus.size(); i++)
            th.emplace_back([&, i]() {
                groot_params prm = params_for(gpus[i]->max_read_len);
                if (groot_hip_open_flags(&gpus[i]->ctx, gpus[i]->device, &v, &prm, GROOT_OPEN_BACKGROUND)) errs[i] = groot_hip_last_error(nullptr);
            });
        for (auto &t : th) t.join();
        for (auto &e : errs) if (!e.empty()) die("%s", e.c_str());
    }
    logf("\tcontainment threshold: %.2f", a.threshold);
    if (a.no_align) logf("\tprevent exact alignments and using approximated mapping only");
    logf("initialising alignment pipeline...");
    logf("\tinitialising the processes");
    logf("\tconnecting data streams");
    logf("\tnumber of processes added to the alignment pipeline: 5");
    if (gpus.size() > 1) logf("\treads shard over %zu GPU contexts (index replicated)", gpus.size());
    const double load_s = seconds_since(t0);

    groot_bam *bam = nullptr;
    if (!a.no_align && groot_bam_open(a.bam_out.empty() ? nullptr : a.bam_out.c_str(), &v, nullptr, &bam)) die("%s", groot_host_last_error());
    if (bam) { groot_bam_set_threads(bam, cores); if (groot_bam_set_level(bam, a.bam_level)) die("%s", groot_host_last_error()); }

    logf("now streaming reads...");
    auto t_stream = std::chrono::steady_clock::now();
    mapped.cap = gpus.size() * depth + 2;
    gpus_ready = true;

    std::atomic<int> mappers_left{(int)gpus.size()};
    std::vector<std::thread> mappers;
    for (size_t gi = 0; gi < gpus.size(); gi++)
        mappers.emplace_back([&, gi]() {
            Gpu &g = *gpus[gi];
            bool input_done = false;
            auto drain_released = [&](bool wait) {
                std::unique_lock<std::mutex> lk(g.mu);
                if (wait) g.cv.wait(lk, [&] { return !g.done_tickets.empty() || failed; });
                for (uint64_t t : g.done_tickets) { groot_hip_release(g.ctx, t); g.held--; }
                g.done_tickets.clear();
            };
            auto collect_one = [&]() -> bool {
                WorkItem w = std::move(g.pending.front());
                g.pending.pop_front();
                auto tc = std::chrono::steady_clock::now();
                const int rc = groot_hip_collect(g.ctx, &w.res);
                collect_wait_us += (uint64_t)(seconds_since(tc) * 1e6);
                n_batches++;
                // the reference's panics (short read, RevComplement on a byte > 'T') and over-long reads end the run
                if (rc) { fail_with(groot_hip_last_error(g.ctx)); return false; }
                g.inflight--; g.held++;
                w.gpu = (int)gi;
                mapped.push(std::move(w));
                return true;
            };
            // a batch with a read longer than the ctx was opened for: finish what is in flight, carry the call counts over
            // into a ctx with room for it (the reference has no read-length limit)
            auto grow_ctx = [&](uint32_t need) -> bool {
                while (g.inflight) if (!collect_one()) return false;
                while (g.held && !failed) drain_released(true);
                if (failed) return false;
                uint32_t n_rows = 0, nw = 0;
                if (groot_hip_attempts_export(g.ctx, nullptr, nullptr, 0, &n_rows, &nw)) { fail_with(groot_hip_last_error(g.ctx)); return false; }
                std::vector<uint32_t> qv(n_rows), cnt((size_t)n_rows * nw);
                if (n_rows && groot_hip_attempts_export(g.ctx, qv.data(), cnt.data(), n_rows, &n_rows, &nw)) { fail_with(groot_hip_last_error(g.ctx)); return false; }
                groot_hip_close(g.ctx);
                g.ctx = nullptr;
                g.max_read_len = std::min<uint32_t>(65535, need + need / 2);
                groot_params prm = params_for(g.max_read_len);
                logf("\tread of %u bases: reopening the GPU context for reads up to %u bases", need, g.max_read_len);
                if (groot_hip_open_flags(&g.ctx, g.device, &v, &prm, GROOT_OPEN_BACKGROUND)) { fail_with(groot_hip_last_error(nullptr)); return false; }
                if (n_rows && groot_hip_attempts_import(g.ctx, qv.data(), cnt.data(), n_rows)) { fail_with(groot_hip_last_error(g.ctx)); return false; }
                return true;
            };
            while (!failed) {
                drain_released(false);
                const uint32_t free_slots = depth - g.held - g.inflight;
                if (!input_done && free_slots > 0) {
                    WorkItem w;
                    const int got = parsed.pop(w, g.inflight == 0 && g.held == 0);
                    if (got < 0) input_done = true;
                    else if (got > 0) {
                        if (w.view.max_len > g.max_read_len && !grow_ctx(w.view.max_len)) break;
                        if (groot_hip_submit_packed16(g.ctx, w.view.packed, w.view.seq_len, w.view.n_reads, 0, w.view.exc_pos, w.view.exc_byte, w.view.n_exc)) {
                            fail_with(groot_hip_last_error(g.ctx));
                            break;
                        }
                        g.inflight++;
                        g.pending.push_back(std::move(w));
                        continue;
                    }
                }
                if (g.inflight) { if (!collect_one()) break; continue; }
                if (input_done && g.held == 0) break;
                if (g.held) drain_released(true);               // everything is with the writer: wait for a slot
                else if (!input_done) {                          // nothing to do but wait for input
                    WorkItem w;
                    const int got = parsed.pop(w, true);
                    if (got < 0) { input_done = true; continue; }
                    if (got > 0) {
                        if (w.view.max_len > g.max_read_len && !grow_ctx(w.view.max_len)) break;
                        if (groot_hip_submit_packed16(g.ctx, w.view.packed, w.view.seq_len, w.view.n_reads, 0, w.view.exc_pos, w.view.exc_byte, w.view.n_exc)) {
                            fail_with(groot_hip_last_error(g.ctx));
                            break;
                        }
                        g.inflight++;
                        g.pending.push_back(std::move(w));
                    }
                }
            }
            if (--mappers_left == 0) mapped.close();
        });

    // ---- writer: batches in input order ----
    uint64_t received = 0, mapped_reads = 0, multimapped = 0, alignments = 0, full_sketch = 0;
    {
        std::map<uint64_t, WorkItem> waiting;
        uint64_t next_seq = 0;
        for (;;) {
            WorkItem w;
            const int got = mapped.pop(w, true);
            if (got < 0) break;
            waiting.emplace(w.seq, std::move(w));
            while (!waiting.empty() && waiting.begin()->first == next_seq) {
                WorkItem it = std::move(waiting.begin()->second);
                waiting.erase(waiting.begin());
                next_seq++;
                const groot_counts &c = it.res.counts;
                received += c.received; mapped_reads += c.mapped; multimapped += c.multimapped; alignments += c.alignments;
                full_sketch += c.full_sketch_reads;
                if (bam && it.res.n_travs && !failed) {
                    uint64_t nrec = 0;
                    auto tw = std::chrono::steady_clock::now();
                    const int wrc = groot_bam_write_batch(bam, &v, &it.view, 0, it.res.travs, it.res.masks, it.res.mask_ckpt, it.res.n_travs, &nrec);
                    bam_s += seconds_since(tw);
                    if (wrc) fail_with(groot_host_last_error());
                    else if (nrec != c.alignments)
                        fail_with("internal error: " + std::to_string(nrec) + " records written, " + std::to_string(c.alignments) + " alignments counted");
                }
                groot_reads_batch_free(it.batch);
                Gpu &g = *gpus[(size_t)it.gpu];
                { std::lock_guard<std::mutex> lk(g.mu); g.done_tickets.push_back(it.res.ticket); }
                g.cv.notify_all();
            }
        }
        // after a failure: hand back whatever is still queued so that the mappers can finish
        for (auto &kv : waiting) {
            groot_reads_batch_free(kv.second.batch);
            Gpu &g = *gpus[(size_t)kv.second.gpu];
            { std::lock_guard<std::mutex> lk(g.mu); g.done_tickets.push_back(kv.second.res.ticket); }
            g.cv.notify_all();
        }
    }
    for (auto &t : mappers) t.join();
    producer.join();
    for (auto &g : gpus) if (g->ctx) groot_hip_open_abandon(g->ctx);   // (a short input can end before the background part of the open has)
    if (failed) {
        // unblock a producer that may sit in push()
        die("%s", fatal.c_str());
    }
    groot_reads_close(reads);
    if (received == 0) die("no fastq reads received");                                           // sketch.go:275-277
    logf("\tnumber of reads received from input: %llu", (unsigned long long)received);           // sketch.go:278-280
    logf("\tmean read length: %.0f", (double)length_total.load() / (double)received);
    logf("\tnumber of reads sketched: %llu", (unsigned long long)received);                      // sketch.go:321
    const uint64_t bam_bytes = bam ? groot_bam_bytes_written(bam) : 0;
    if (bam && groot_bam_close(bam)) die("%s", groot_host_last_error());
    const double stream_s = seconds_since(t_stream);
    auto t_post = std::chrono::steady_clock::now();

    int rc = 0;
    if (mapped_reads == 0) {
        logf("no reads could be mapped to the reference graphs");                                // sketch.go:328-334
    } else {
        logf("\ttotal number of unmapped reads: %llu", (unsigned long long)(received - mapped_reads)); // sketch.go:335-339
        logf("\ttotal number of mapped reads: %llu", (unsigned long long)mapped_reads);
        logf("\t\tmapped to one graph: %llu", (unsigned long long)(mapped_reads - multimapped));
        logf("\t\tmapped to multiple graphs: %llu", (unsigned long long)multimapped);
        logf("\ttotal number of exact alignments: %llu", (unsigned long long)alignments);
        // graph weights: exact call counts from the devices (summed over the GPUs: one RCCL all-reduce of a table with one row
        // per kmerCount that occurred), one replay of IncrementSubPath on the host
        // (a context that met a longer read was reopened with a larger limit, the others were not: the tables can only be summed
        // over contexts with the same kmerCount range, so the shorter ones follow now -- export, reopen, import, as grow_ctx does)
        uint32_t longest = 0;
        for (auto &g : gpus) longest = std::max(longest, g->max_read_len);
        for (auto &g : gpus) {
            if (g->max_read_len == longest) continue;
            uint32_t n_rows = 0, nw = 0;
            if (groot_hip_attempts_export(g->ctx, nullptr, nullptr, 0, &n_rows, &nw)) die("%s", groot_hip_last_error(g->ctx));
            std::vector<uint32_t> qv(n_rows), cnt((size_t)n_rows * nw);
            if (n_rows && groot_hip_attempts_export(g->ctx, qv.data(), cnt.data(), n_rows, &n_rows, &nw)) die("%s", groot_hip_last_error(g->ctx));
            groot_hip_close(g->ctx);
            g->ctx = nullptr;
            g->max_read_len = longest;
            groot_params prm = params_for(longest);
            logf("\tGPU %d: reopening its context for reads up to %u bases (another context met one) before the call counts are summed", g->device, longest);
            if (groot_hip_open_flags(&g->ctx, g->device, &v, &prm, GROOT_OPEN_BACKGROUND)) die("%s", groot_hip_last_error(nullptr));
            if (n_rows && groot_hip_attempts_import(g->ctx, qv.data(), cnt.data(), n_rows)) die("%s", groot_hip_last_error(g->ctx));
        }
        std::vector<groot_ctx *> ctxs;
        for (auto &g : gpus) ctxs.push_back(g->ctx);
        if (groot_hip_attempts_allreduce(ctxs.data(), (int)ctxs.size())) die("%s", groot_hip_last_error(ctxs[0]));
        uint32_t n_rows = 0, nw = 0;
        if (groot_hip_attempts_export(ctxs[0], nullptr, nullptr, 0, &n_rows, &nw)) die("%s", groot_hip_last_error(ctxs[0]));
        std::vector<uint32_t> qv(n_rows), counts((size_t)n_rows * nw);
        if (n_rows && groot_hip_attempts_export(ctxs[0], qv.data(), counts.data(), n_rows, &n_rows, &nw)) die("%s", groot_hip_last_error(ctxs[0]));
        std::vector<double> kf(v.n_nodes);
        std::vector<uint64_t> kt(v.n_graphs);
        if (groot_host_weights_rows(&v, qv.data(), n_rows, counts.data(), kf.data(), kt.data())) die("%s", groot_host_last_error());
        uint64_t total_kmers = 0;
        for (auto t : kt) total_kmers += t;
        logf("processing graphs...");
        logf("\ttotal number of k-mers projected onto graphs: %llu", (unsigned long long)total_kmers);   // sketch.go:346-347
        std::vector<uint8_t> gk(v.n_graphs), pk(v.n_paths), nr(v.n_nodes);
        if (groot_host_prune(&v, kf.data(), a.min_kmer_cov, gk.data(), pk.data(), nr.data())) die("%s", groot_host_last_error());
        uint32_t kept_graphs = 0, kept_paths = 0;
        for (uint32_t g = 0; g < v.n_graphs; g++) {
            if (!gk[g]) continue;
            kept_graphs++;
            // sketch.go:409: len(g.Paths) is never shrunk by Prune, so the reference logs the full path count
            logf("\tgraph %u has %u remaining paths after weighting and pruning", g, v.graph_path_off[g + 1] - v.graph_path_off[g]);
            for (uint32_t p = v.graph_path_off[g]; p < v.graph_path_off[g + 1]; p++)
                logf("\t- [%.*s]", (int)(v.path_name_off[p + 1] - v.path_name_off[p]), v.path_names + v.path_name_off[p]);
            kept_paths += v.graph_path_off[g + 1] - v.graph_path_off[g];
        }
        logf("\ttotal number of graphs pruned: %u", v.n_graphs);                                 // sketch.go:421-427
        if (!kept_graphs) logf("\tno graphs remaining after pruning");
        else {
            logf("\ttotal number of graphs remaining: %u", kept_graphs);
            logf("\ttotal number of possible haplotypes found: %u", kept_paths);
            logf("saving graphs...");                                                            // cmd/align.go:153-161
            // (one file per graph, independent of each other: written side by side)
            std::atomic<uint32_t> next_g{0};
            std::atomic<bool> gfa_failed{false};
            std::mutex gfa_mu;
            std::string gfa_err;
            auto save = [&]() {
                for (uint32_t g = next_g.fetch_add(1); g < v.n_graphs; g = next_g.fetch_add(1)) {
                    if (!gk[g]) continue;
                    const std::string fn = graph_dir + "/groot-graph-" + std::to_string(g) + ".gfa";
                    int written = 0;
                    if (groot_host_save_gfa(&v, g, kf.data(), pk.data(), nr.data(), total_kmers, nullptr, fn.c_str(), &written)) {
                        std::lock_guard<std::mutex> lk(gfa_mu);
                        if (!gfa_failed.exchange(true)) gfa_err = groot_host_last_error();
                    }
                }
            };
            std::vector<std::thread> savers;
            for (int t = 1; t < std::max(1, std::min(a.proc, 16)); t++) savers.emplace_back(save);
            save();
            for (auto &t : savers) t.join();
            if (gfa_failed) die("%s", gfa_err.c_str());
        }
    }
    // (the contexts and the index are NOT torn down: the process ends here -- log and stats written, BAM and GFAs closed -- and handing a few
    // GB of HBM and host memory back buffer by buffer, then unloading the HIP runtime, costs a tenth of a second that no output needs:
    // main() leaves through _exit once everything is flushed)
    for (auto &g : gpus) if (g->ctx) groot_hip_open_abandon(g->ctx);
    (void)idx;
    const double post_s = seconds_since(t_post), total_s = seconds_since(t0);
    if (!a.stats_file.empty()) {
        FILE *sf = fopen(a.stats_file.c_str(), "w");
        if (sf) {
            fprintf(sf, "{\"reads\": %llu, \"mapped\": %llu, \"alignments\": %llu, \"gpu_contexts\": %zu, \"load_s\": %.6f, \"stream_s\": %.6f, "
                        "\"post_s\": %.6f, \"total_s\": %.6f, \"bam_bytes\": %llu, \"bam_level\": %d, \"threads\": %u, \"batches\": %llu, "
                        "\"parse_busy_s\": %.6f, \"bam_busy_s\": %.6f, \"collect_wait_s\": %.6f, \"full_sketch_reads\": %llu}\n",
                    (unsigned long long)received, (unsigned long long)mapped_reads, (unsigned long long)alignments, gpus.size(), load_s, stream_s,
                    post_s, total_s, (unsigned long long)bam_bytes, a.bam_level, cores ? cores : groot_host_usable_cpus(),
                    (unsigned long long)n_batches.load(), parse_s, bam_s, (double)collect_wait_us.load() / 1e6, (unsigned long long)full_sketch);
            fclose(sf);
        }
    }
    logf("finished in %.3fs", total_s);
    return rc;
}

// cmd/report.go:104-129
int run_report(const Args &a)
{
    start_logging(a);
    logf("i am groot (version %s)", groot_host_version());
    logf("starting the report subcommand");
    logf("checking parameters...");
    if (a.bam_file.empty()) logf("\tBAM file: using STDIN");
    else {
        if (!is_file(a.bam_file)) die("BAM file does not exist: %s", a.bam_file.c_str());
        const size_t dot = a.bam_file.rfind('.');
        if (dot == std::string::npos || a.bam_file.substr(dot + 1) != "bam") die("the BAM file does not have a `.bam` extension: %s", a.bam_file.c_str());
        logf("\tBAM file: %s", a.bam_file.c_str());
    }
    if (a.cov_cutoff > 1.0) die("supplied coverage cutoff exceeds 1.0 (100%%): %g", a.cov_cutoff);
    logf("\tcoverage cutoff: %.2f", a.cov_cutoff);
    logf("\tprocessors: %d", a.proc);
    uint64_t n = 0;
    if (groot_host_report(a.bam_file.empty() ? nullptr : a.bam_file.c_str(), a.cov_cutoff, a.low_cov ? 1 : 0, nullptr, &n))
        die("%s", groot_host_last_error());
    logf("finished");
    return 0;
}

} // namespace

int main(int argc, char **argv)
{
    Args a = parse(argc, argv);
    if (a.cmd == "index") return run_index(a);
    if (a.cmd == "align") {
        const int rc = run_align(a);
        fflush(nullptr);
        _exit(rc);                                      // (see the end of run_align)
    }
    if (a.cmd == "report") return run_report(a);
    if (a.cmd == "version") { printf("%s\n", groot_host_version()); return 0; }
    usage();
    return 1;
}

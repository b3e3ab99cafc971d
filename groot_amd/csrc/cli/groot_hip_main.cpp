// groot-hip -- flag-compatible `index` and `align` subcommands on top of libgroot_host.so / libgroot_hip.so.
//
//   groot-hip index -m <msaDir> -i <indexDir> [-k 31 -s 21 -w 100 -x 8 -y 4 --maxSketchSpan 30 -p N --log F]
//        cmd/index.go:44-133: writes <indexDir>/groot.gidx (flat index) and the reference's groot.gg + groot.lshe
//   groot-hip align -i <indexDir> -f a.fq[,b.fq.gz] [-t 0.99 -c 1.0 -g <graphDir> --noAlign -p N --log F] > out.bam
//        cmd/align.go:30-197 + src/pipeline/sketch.go (DataStreamer..GraphPruner): BAM on stdout, weighted GFAs in
//        graphDir, the reference's log lines in --log (default groot.log)
//
// Extra flags: --gpu <id> (device), --batch <reads> (reads per device batch), --bam <file> (Info.Sketch.BAMout).
// The align hot path runs only on the GPU: no device -> error, never a CPU fallback.
#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <ctime>
#include <mutex>
#include <string>
#include <thread>
#include <sys/stat.h>
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
    exit(1);
}

struct Args {
    std::string cmd, index_dir, msa_dir, log_file = "groot.log", graph_dir, bam_out, bam_file;
    double cov_cutoff = 0.97;
    bool low_cov = false;
    std::vector<std::string> fastq;
    int proc = 1, gpu = 0;
    bool gpu_given = false;
    uint32_t k = 31, s = 21, w = 100, x = 8, y = 4, max_span = 30, batch = 1u << 20;
    double threshold = 0.99, min_kmer_cov = 1.0;
    bool no_align = false, fasta = false;
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
            "  groot-hip align -i <indexDir> -f <fastq>[,<fastq>...] [-t 0.99] [-c 1.0] [-g <graphDir>] [--noAlign] [-p N] [--log F]\n"
            "                  [--gpu 0] [--batch 1048576] [--bam out.bam]      (BAM goes to stdout unless --bam)\n"
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
        else if (f == "--bam") a.bam_out = v();
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
    // and the reference's own files (cmd/index.go:130-131), so that `groot align|haplotype` can use this directory too
    if (groot_index_save_gob(idx, a.index_dir.c_str(), a.max_span)) die("%s", groot_host_last_error());
    groot_index_free(idx);
    logf("finished in %.3fs", seconds_since(t0));
    return 0;
}

// ---------------------------------------------------------------------------------------------
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
    logf("loading the index information...");
    groot_index *idx = nullptr;
    if (is_file(gidx) ? groot_index_load(gidx.c_str(), &idx) : groot_index_load_gob(gg.c_str(), lshe.c_str(), &idx))
        die("%s", groot_host_last_error());
    groot_index_view v;
    groot_index_get_view(idx, &v);
    logf("\tk-mer size: %u", v.kmer_size);
    logf("\tsketch size: %u", v.sketch_size);
    logf("\twindow size used in indexing: %u", v.window_size);
    logf("loading the graphs...");
    logf("\tnumber of variation graphs: %u", v.n_graphs);
    logf("rebuilding the LSH Ensemble...");
    groot_params prm;
    groot_params_default(&prm);
    prm.containment_threshold = a.threshold;
    prm.no_exact_align = a.no_align ? 1 : 0;
    prm.max_batch_reads = a.batch;
    prm.max_read_len = 512;
    groot_ctx *ctx = nullptr;
    if (groot_hip_open(&ctx, a.gpu, &v, &prm)) die("%s", groot_hip_last_error(nullptr));
    logf("\tcontainment threshold: %.2f", a.threshold);
    if (a.no_align) logf("\tprevent exact alignments and using approximated mapping only");
    logf("initialising alignment pipeline...");
    logf("\tinitialising the processes");
    logf("\tconnecting data streams");
    logf("\tnumber of processes added to the alignment pipeline: 5");

    groot_bam *bam = nullptr;
    if (!a.no_align && groot_bam_open(a.bam_out.empty() ? nullptr : a.bam_out.c_str(), &v, nullptr, &bam)) die("%s", groot_host_last_error());
    if (bam) groot_bam_set_threads(bam, a.proc > 0 ? (uint32_t)a.proc : 0);   // -p: BGZF write concurrency

    std::vector<const char *> files;
    for (auto &f : a.fastq) files.push_back(f.c_str());
    groot_fastq *fq = nullptr;
    if (groot_fastq_open(files.empty() ? nullptr : files.data(), (uint32_t)files.size(), &fq)) die("%s", groot_host_last_error());
    logf("now streaming reads...");

    // DataStreamer/FastqHandler run ahead of the mapper (the reference connects them with buffered channels,
    // pipeline.go:5): a reader thread parses batch i+1 while batch i is on the GPU and in the BAM writer
    const uint64_t seq_cap = (uint64_t)a.batch * prm.max_read_len, name_cap = (uint64_t)a.batch * 256;
    struct Batch {
        std::vector<uint8_t> seq, qual;
        std::vector<char> names;
        std::vector<uint64_t> seq_off, name_off;
        int64_t n = 0;
        std::string err;
    } bufs[2];
    for (auto &bf : bufs) {
        bf.seq.resize(seq_cap); bf.qual.resize(seq_cap); bf.names.resize(name_cap);
        bf.seq_off.resize(a.batch + 1); bf.name_off.resize(a.batch + 1);
    }
    std::mutex mu;
    std::condition_variable cv;
    int filled[2] = {0, 0};      // 0 = free for the reader, 1 = ready for the mapper
    bool reader_done = false;
    std::thread reader([&]() {
        for (int slot = 0;; slot ^= 1) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return filled[slot] == 0; });
            }
            Batch &bf = bufs[slot];
            bf.n = groot_fastq_next_batch(fq, a.batch, bf.seq.data(), bf.qual.data(), bf.seq_off.data(), seq_cap, bf.names.data(),
                                          bf.name_off.data(), name_cap);
            if (bf.n < 0) bf.err = groot_host_last_error();
            const bool last = bf.n <= 0;
            {
                std::lock_guard<std::mutex> lk(mu);
                filled[slot] = 1;
                if (last) reader_done = true;
            }
            cv.notify_all();
            if (last) break;
        }
    });
    std::vector<groot_trav> travs;
    std::vector<uint64_t> masks;
    uint64_t received = 0, length_total = 0, mapped = 0, multimapped = 0, alignments = 0;
    uint32_t first_id = 0;
    for (int slot = 0;; slot ^= 1) {
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return filled[slot] == 1; });
        }
        Batch &bf = bufs[slot];
        const int64_t n = bf.n;
        if (n < 0) { reader.join(); die("%s", bf.err.c_str()); }
        if (n == 0) break;
        length_total += bf.seq_off[n];
        if (groot_hip_submit(ctx, bf.seq.data(), bf.seq_off.data(), (uint32_t)n, first_id)) die("%s", groot_hip_last_error(ctx));
        groot_counts c;
        if (groot_hip_wait(ctx, &c)) die("%s", groot_hip_last_error(ctx));   // the reference's panics become fatal errors
        received += c.received; mapped += c.mapped; multimapped += c.multimapped; alignments += c.alignments;
        if (!a.no_align && c.travs) {
            travs.resize(c.travs);
            masks.resize(c.travs * v.path_words);
            uint64_t nt = 0;
            if (groot_hip_read_travs(ctx, travs.data(), masks.data(), c.travs, &nt)) die("%s", groot_hip_last_error(ctx));
            // traversal records -> sam.Records -> BGZF, in parallel over chunks of traversals
            groot_read_batch rb{bf.seq.data(), bf.qual.data(), bf.seq_off.data(), bf.names.data(), bf.name_off.data(), (uint32_t)n, first_id};
            uint64_t nrec = 0;
            if (groot_bam_write_travs(bam, &v, &rb, travs.data(), masks.data(), nt, &nrec)) die("%s", groot_host_last_error());
            if (nrec != c.alignments) die("internal error: %llu records written, %llu alignments counted", (unsigned long long)nrec, (unsigned long long)c.alignments);
        }
        first_id += (uint32_t)n;
        {
            std::lock_guard<std::mutex> lk(mu);
            filled[slot] = 0;
        }
        cv.notify_all();
    }
    reader.join();
    (void)reader_done;
    groot_fastq_close(fq);
    if (received == 0) die("no fastq reads received");                                           // sketch.go:275-277
    logf("\tnumber of reads received from input: %llu", (unsigned long long)received);           // sketch.go:278-280
    logf("\tmean read length: %.0f", (double)length_total / (double)received);
    logf("\tnumber of reads sketched: %llu", (unsigned long long)received);                      // sketch.go:321
    if (bam && groot_bam_close(bam)) die("%s", groot_host_last_error());

    int rc = 0;
    if (mapped == 0) {
        logf("no reads could be mapped to the reference graphs");                                // sketch.go:328-334
    } else {
        logf("\ttotal number of unmapped reads: %llu", (unsigned long long)(received - mapped)); // sketch.go:335-339
        logf("\ttotal number of mapped reads: %llu", (unsigned long long)mapped);
        logf("\t\tmapped to one graph: %llu", (unsigned long long)(mapped - multimapped));
        logf("\t\tmapped to multiple graphs: %llu", (unsigned long long)multimapped);
        logf("\ttotal number of exact alignments: %llu", (unsigned long long)alignments);
        // graph weights: exact call counts from the device, one replay of IncrementSubPath on the host
        uint32_t nq = 0, nw = 0;
        groot_hip_attempts_shape(ctx, &nq, &nw);
        std::vector<uint32_t> counts((size_t)nq * nw);
        if (groot_hip_attempts_read(ctx, counts.data(), counts.size())) die("%s", groot_hip_last_error(ctx));
        std::vector<double> kf(v.n_nodes);
        std::vector<uint64_t> kt(v.n_graphs);
        if (groot_host_weights(&v, counts.data(), nq, kf.data(), kt.data())) die("%s", groot_host_last_error());
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
            uint32_t np = 0;
            for (uint32_t p = v.graph_path_off[g]; p < v.graph_path_off[g + 1]; p++) np += pk[p];
            // sketch.go:409: len(g.Paths) is never shrunk by Prune, so the reference logs the full path count
            logf("\tgraph %u has %u remaining paths after weighting and pruning", g, v.graph_path_off[g + 1] - v.graph_path_off[g]);
            for (uint32_t p = v.graph_path_off[g]; p < v.graph_path_off[g + 1]; p++)
                logf("\t- [%.*s]", (int)(v.path_name_off[p + 1] - v.path_name_off[p]), v.path_names + v.path_name_off[p]);
            kept_paths += v.graph_path_off[g + 1] - v.graph_path_off[g];
            (void)np;
        }
        logf("\ttotal number of graphs pruned: %u", v.n_graphs);                                 // sketch.go:421-427
        if (!kept_graphs) logf("\tno graphs remaining after pruning");
        else {
            logf("\ttotal number of graphs remaining: %u", kept_graphs);
            logf("\ttotal number of possible haplotypes found: %u", kept_paths);
            logf("saving graphs...");                                                            // cmd/align.go:153-161
            for (uint32_t g = 0; g < v.n_graphs; g++) {
                if (!gk[g]) continue;
                const std::string fn = graph_dir + "/groot-graph-" + std::to_string(g) + ".gfa";
                int written = 0;
                if (groot_host_save_gfa(&v, g, kf.data(), pk.data(), nr.data(), total_kmers, nullptr, fn.c_str(), &written)) die("%s", groot_host_last_error());
            }
        }
    }
    groot_hip_close(ctx);
    groot_index_free(idx);
    logf("finished in %.3fs", seconds_since(t0));
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
    if (a.cmd == "align") return run_align(a);
    if (a.cmd == "report") return run_report(a);
    if (a.cmd == "version") { printf("%s\n", groot_host_version()); return 0; }
    usage();
    return 1;
}

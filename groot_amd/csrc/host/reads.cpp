// reads.cpp -- parallel FASTQ ingest for the align path: text blocks -> parsed + 2-bit packed batches.
//   src/pipeline/sketch.go:41-77   DataStreamer: line scanner over stdin / files, gzip if the name ends in .gz
//   src/pipeline/sketch.go:213-236 FastqHandler: every four lines form one read; a trailing partial record is dropped
//   src/seqio/seqio.go:173-188     NewFASTQread: line 1 must start with '@'; no other check
// The reference pushes every line through a channel to ONE FastqHandler goroutine.  Here one thread per input file
// (the next few files are opened ahead: their gzip streams inflate concurrently, bounded queues) produces raw text
// blocks; the caller's thread frames each block at a record boundary and a pool of workers finds the line breaks, parses
// the records and packs the bases straight into the wire format of groot_hip_submit_packed16 -- all in parallel over
// the block.  Lines from consecutive files form ONE stream (the four-line grouping carries across files), lines end at
// "\n" or "\r\n" (bufio.ScanLines), a last line without terminator counts.
#include "host_common.hpp"

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>

#include <cerrno>
#include <fcntl.h>
#include <unistd.h>
#include <zlib.h>

#include "gz_inflate.hpp"

using namespace groot;

namespace {

constexpr size_t kHeadroom = 4u << 20;          // room in front of a block for the partial record carried over

struct RawBlock {
    std::vector<char> buf;                      // [kHeadroom + capacity]
    size_t begin = kHeadroom, end = kHeadroom;  // text = buf[begin, end)
};

template <class F> void parallel_for(unsigned n_threads, size_t n_tasks, F fn)
{
    const unsigned nt = (unsigned)std::max<size_t>(1, std::min<size_t>(n_threads, n_tasks));
    if (nt == 1) { for (size_t i = 0; i < n_tasks; i++) fn(i); return; }
    std::atomic<size_t> next{0};
    auto work = [&]() { for (size_t i; (i = next.fetch_add(1)) < n_tasks;) fn(i); };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; t++) th.emplace_back(work);
    work();
    for (auto &x : th) x.join();
}

// one input (a file, or stdin) read by its own thread into a bounded queue of raw blocks
struct Source {
    std::string path;            // empty = stdin
    size_t block_bytes;
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::unique_ptr<RawBlock>> q;
    bool done = false, stop = false;
    std::string err;

    void run()
    {
        // gzip streams go through GzInflater; anything else is read straight into the block (a gzip layer would copy it through its
        // own buffer at a fraction of the page-cache rate)
        const int fd = path.empty() ? 0 : open(path.c_str(), O_RDONLY);
        if (fd < 0) { finish("cannot open " + path); return; }
        unsigned char magic[2] = {0, 0};
        size_t have_magic = 0;
        while (have_magic < 2) {
            const ssize_t n = read(fd, magic + have_magic, 2 - have_magic);
            if (n <= 0) break;
            have_magic += (size_t)n;
        }
        // (stdin is scanned as it comes, as in the reference: sketch.go:45-53 wraps only named *.gz files in a gzip reader)
        const bool gz = !path.empty() && have_magic == 2 && magic[0] == 0x1f && magic[1] == 0x8b && lseek(fd, 0, SEEK_SET) == 0;
        // (gz_inflate.hpp: a gzip reader of this repo's own, 2-2.5x zlib's gzread on FASTQ -- the one inflate stream per file is what a gzip input waits for)
        std::unique_ptr<GzInflater> fh;
        if (gz) fh.reset(new GzInflater(fd));
        char last = '\n';
        bool first = true;
        for (;;) {
            std::unique_ptr<RawBlock> b(new RawBlock());
            b->buf.resize(kHeadroom + block_bytes + 1);
            char *dst = b->buf.data() + kHeadroom;
            size_t fill = 0;
            bool eof = false;
            if (first && !gz) { memcpy(dst, magic, have_magic); fill = have_magic; }
            first = false;
            while (fill < block_bytes) {
                const size_t want = std::min<size_t>(block_bytes - fill, 1u << 30);
                const ssize_t n = gz ? fh->read(reinterpret_cast<uint8_t *>(dst) + fill, want) : read(fd, dst + fill, want);
                if (n < 0) {
                    if (!gz && errno == EINTR) continue;
                    const std::string why = gz ? " (" + fh->error() + ")" : std::string();
                    if (fd) close(fd);
                    finish("read error in FASTQ input " + path + why);
                    return;
                }
                if (n == 0) { eof = true; break; }
                fill += (size_t)n;
            }
            if (fill) last = dst[fill - 1];
            if (eof && last != '\n') { dst[fill++] = '\n'; last = '\n'; }   // a last line without '\n' ends at the file end
            b->end = kHeadroom + fill;
            if (fill) {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return q.size() < 3 || stop; });
                if (stop) break;
                q.push_back(std::move(b));
                cv.notify_all();
            }
            if (eof) break;
        }
        if (fd) close(fd);
        finish("");
    }
    void finish(const std::string &e)
    {
        std::lock_guard<std::mutex> lk(mu);
        err = e; done = true;
        cv.notify_all();
    }
    // next block of this input; nullptr at its end (err set on failure)
    std::unique_ptr<RawBlock> pop()
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return !q.empty() || done; });
        if (q.empty()) return nullptr;
        auto b = std::move(q.front());
        q.pop_front();
        cv.notify_all();
        return b;
    }
    void shutdown()
    {
        { std::lock_guard<std::mutex> lk(mu); stop = true; cv.notify_all(); }
        if (th.joinable()) th.join();
    }
};

struct PackLut {
    uint8_t t[256];      // 2-bit code (byte >> 1) & 3, bit 7 set for bytes other than A C G T
    PackLut()
    {
        for (int c = 0; c < 256; c++) t[c] = (uint8_t)(((c >> 1) & 3) | ((c == 'A' || c == 'C' || c == 'G' || c == 'T') ? 0 : 0x80));
    }
};
const PackLut g_lut;
struct PackLut2 {
    uint8_t t[65536];    // two bases at once (little endian: first base in the low byte): code of the first | code of the second << 2, bit 7 set if either is not A C G T
    PackLut2()
    {
        for (int c = 0; c < 65536; c++) {
            const int a = c & 0xFF, b = c >> 8;
            t[c] = (uint8_t)(g_lut.t[a] & 3) | (uint8_t)((g_lut.t[b] & 3) << 2) | (uint8_t)((g_lut.t[a] | g_lut.t[b]) & 0x80);
        }
    }
};
const PackLut2 g_lut2;

} // namespace

// one parsed batch: per-read positions into the (shared) text block + the wire format of groot_hip_submit_packed16
struct groot_reads_batch {
    std::shared_ptr<RawBlock> text;
    std::vector<uint32_t> name_pos, name_len, seq_pos, qual_pos, qual_len;
    std::vector<uint16_t> seq_len;
    std::vector<uint8_t> packed, exc_byte;
    std::vector<uint64_t> exc_pos;
    uint64_t n_bases = 0;
    uint32_t max_len = 0;
};

struct groot_reads {
    std::vector<std::string> files;     // empty = stdin
    unsigned n_threads = 1;
    size_t block_bytes = 0;
    uint32_t max_batch_reads = 0;
    uint64_t max_batch_bases = 0;
    std::deque<std::unique_ptr<Source>> open;   // inputs being read (front = the one consumed now)
    size_t next_file = 0;
    bool started = false, input_done = false;
    std::vector<char> carry;            // text after the last complete record of the previous block
    std::deque<groot_reads_batch *> ready;      // batches cut from the current block, not yet handed out
    uint64_t n_reads = 0;

    void open_more()
    {
        const size_t ahead = 4;
        if (files.empty()) {
            if (!started) { add(""); }
            return;
        }
        while (open.size() < ahead && next_file < files.size()) add(files[next_file++]);
    }
    void add(const std::string &p)
    {
        std::unique_ptr<Source> s(new Source());
        s->path = p; s->block_bytes = block_bytes;
        Source *raw = s.get();
        s->th = std::thread([raw]() { raw->run(); });
        open.push_back(std::move(s));
    }
    ~groot_reads()
    {
        for (auto &s : open) s->shutdown();
        for (auto *b : ready) delete b;
    }
};

// frames one raw block (carry + new text), parses it and appends its batches to r->ready
static int parse_block(groot_reads *r, std::unique_ptr<RawBlock> blk, bool last_block)
{
    // the partial record carried over goes in front of the new text
    if (!r->carry.empty()) {
        if (r->carry.size() <= blk->begin) {
            blk->begin -= r->carry.size();
            memcpy(blk->buf.data() + blk->begin, r->carry.data(), r->carry.size());
        } else {      // a carry larger than the headroom (reads of megabases): rebuild the block
            std::vector<char> nb(kHeadroom + r->carry.size() + (blk->end - blk->begin) + 1);
            memcpy(nb.data() + kHeadroom, r->carry.data(), r->carry.size());
            memcpy(nb.data() + kHeadroom + r->carry.size(), blk->buf.data() + blk->begin, blk->end - blk->begin);
            blk->end = kHeadroom + r->carry.size() + (blk->end - blk->begin);
            blk->begin = kHeadroom;
            blk->buf.swap(nb);
        }
        r->carry.clear();
    }
    const char *text = blk->buf.data() + blk->begin;
    const size_t n = blk->end - blk->begin;
    if (n > 0xFFFFFFF0ull) return set_error(GROOT_E_NOSPACE, "FASTQ block larger than 4 GB");
    // ---- line breaks, in parallel over ranges of the text ----
    const unsigned T = r->n_threads;
    const size_t n_ranges = std::max<size_t>(1, std::min<size_t>(T * 4, n / (1u << 20) + 1));
    std::vector<std::vector<uint32_t>> nl_part(n_ranges);
    parallel_for(T, n_ranges, [&](size_t i) {
        const size_t lo = n * i / n_ranges, hi = n * (i + 1) / n_ranges;
        auto &v = nl_part[i];
        v.reserve((hi - lo) / 48 + 16);
        const char *p = text + lo, *e = text + hi;
        while (p < e) {
            const char *q = (const char *)memchr(p, '\n', (size_t)(e - p));
            if (!q) break;
            v.push_back((uint32_t)(q - text));
            p = q + 1;
        }
    });
    std::vector<size_t> part_off(n_ranges + 1, 0);
    for (size_t i = 0; i < n_ranges; i++) part_off[i + 1] = part_off[i] + nl_part[i].size();
    const size_t n_lines = part_off[n_ranges];
    std::vector<uint32_t> nl(n_lines);
    parallel_for(T, n_ranges, [&](size_t i) { if (!nl_part[i].empty()) memcpy(nl.data() + part_off[i], nl_part[i].data(), nl_part[i].size() * 4); });
    nl_part.clear();
    auto raw_line = [&](size_t li, uint32_t &pos, uint32_t &len) {
        pos = li ? nl[li - 1] + 1 : 0;
        len = nl[li] - pos;
        if (len && text[pos + len - 1] == '\r') len--;            // bufio.ScanLines drops one trailing '\r'
    };
    // FastqHandler.Run (sketch.go:216-236) fills l1, l2, l3 with the next line that is not nil -- and an EMPTY line arrives as nil
    // (append([]byte(nil), scanner.Bytes()...) of no bytes) -- then takes whatever comes next as l4.  Blank lines before an ID,
    // a sequence or a '+' line are therefore skipped (a file ending in an extra newline followed by a second file maps fine);
    // only files that hold one pay for the regrouping.
    std::vector<uint8_t> any_blank(n_ranges, 0);
    parallel_for(T, n_ranges, [&](size_t i) {
        for (size_t li = part_off[i]; li < part_off[i + 1]; li++) {
            uint32_t p, l;
            raw_line(li, p, l);
            if (!l) { any_blank[i] = 1; return; }
        }
    });
    std::vector<uint32_t> rec_line;                                // [4 * n_rec] line of every record field; empty = lines 4i .. 4i+3
    size_t n_rec = n_lines / 4;
    if (std::find(any_blank.begin(), any_blank.end(), 1) != any_blank.end()) {
        uint32_t field = 0, cur[4];
        for (size_t li = 0; li < n_lines; li++) {
            uint32_t p, l;
            raw_line(li, p, l);
            if (field < 3 && !l) continue;
            cur[field++] = (uint32_t)li;
            if (field == 4) { rec_line.insert(rec_line.end(), cur, cur + 4); field = 0; }
        }
        n_rec = rec_line.size() / 4;
        if (rec_line.empty()) rec_line.push_back(0);               // (non-empty = "regrouped")
    }
    const bool regrouped = !rec_line.empty();
    const size_t last_line = n_rec ? (regrouped ? (size_t)rec_line[4 * n_rec - 1] : 4 * n_rec - 1) : 0;
    const size_t cut = n_rec ? (size_t)nl[last_line] + 1 : 0;
    if (!last_block) r->carry.assign(text + cut, text + n);       // lines of the next record (and a partial line): carried over
    // (at the end of all input a trailing partial record is dropped, as in FastqHandler.Run)
    if (!n_rec) return GROOT_OK;
    auto line = [&](size_t fi, uint32_t &pos, uint32_t &len) { raw_line(regrouped ? (size_t)rec_line[fi] : fi, pos, len); };
    // ---- pass A: validate, sizes (parallel over record ranges) ----
    const size_t n_tasks = std::max<size_t>(1, std::min<size_t>(T * 4, n_rec / 4096 + 1));
    std::vector<uint64_t> task_bases(n_tasks, 0);
    std::vector<uint32_t> task_max(n_tasks, 0);
    std::vector<int> task_err(n_tasks, 0);
    std::vector<size_t> task_bad(n_tasks, 0);
    parallel_for(T, n_tasks, [&](size_t t) {
        const size_t r0 = n_rec * t / n_tasks, r1 = n_rec * (t + 1) / n_tasks;
        uint64_t bases = 0;
        uint32_t mx = 0;
        for (size_t i = r0; i < r1; i++) {
            uint32_t p, l;
            line(4 * i, p, l);
            if (l == 0 || text[p] != '@') { task_err[t] = 1; task_bad[t] = i; return; }    // seqio.go:179-181
            line(4 * i + 1, p, l);
            if (l > 65535) { task_err[t] = 2; task_bad[t] = i; return; }
            bases += l; mx = std::max(mx, l);
        }
        task_bases[t] = bases; task_max[t] = mx;
    });
    for (size_t t = 0; t < n_tasks; t++)
        if (task_err[t]) {
            uint32_t p, l;
            line(4 * task_bad[t], p, l);
            if (task_err[t] == 1) return set_error(GROOT_E_FORMAT, "read ID in fastq file does not begin with @: %.*s", (int)std::min<uint32_t>(l, 200), text + p);
            return set_error(GROOT_E_UNSUPPORTED, "read longer than 65535 bases: %.*s", (int)std::min<uint32_t>(l, 200), text + p);
        }
    // ---- cut the block into batches that fit the device ctx (reads and bases), then pass B per batch ----
    std::shared_ptr<RawBlock> shared(blk.release());
    size_t rec0 = 0;
    while (rec0 < n_rec) {
        // whole tasks while they fit; a task that does not fit is split record by record
        size_t rec1 = rec0;
        uint64_t bases = 0;
        {
            size_t i = rec0;
            while (i < n_rec && (i - rec0) < r->max_batch_reads) {
                uint32_t p, l;
                line(4 * i + 1, p, l);
                if (bases + l > r->max_batch_bases) break;
                bases += l; i++;
            }
            rec1 = i;
        }
        if (rec1 == rec0) return set_error(GROOT_E_NOSPACE, "a single read does not fit the batch (max_batch_bases=%llu)", (unsigned long long)r->max_batch_bases);
        const size_t m = rec1 - rec0;
        std::unique_ptr<groot_reads_batch> b(new groot_reads_batch());
        b->text = shared;
        b->name_pos.resize(m); b->name_len.resize(m); b->seq_pos.resize(m); b->qual_pos.resize(m); b->qual_len.resize(m); b->seq_len.resize(m);
        b->n_bases = bases;
        b->packed.assign((bases + 3) / 4 + 8, 0);
        const size_t nt2 = std::max<size_t>(1, std::min<size_t>(T * 4, m / 2048 + 1));
        // base offset of every task's first read: one more sizes pass over this batch (cheap next to the packing)
        std::vector<uint64_t> tb(nt2 + 1, 0);
        parallel_for(T, nt2, [&](size_t t) {
            const size_t a = rec0 + m * t / nt2, z = rec0 + m * (t + 1) / nt2;
            uint64_t s = 0;
            for (size_t i = a; i < z; i++) { uint32_t p, l; line(4 * i + 1, p, l); s += l; }
            tb[t + 1] = s;
        });
        for (size_t t = 0; t < nt2; t++) tb[t + 1] += tb[t];
        struct Edge { uint64_t byte; uint8_t val; };
        std::vector<std::vector<Edge>> edges(nt2);
        std::vector<std::vector<std::pair<uint64_t, uint8_t>>> exc(nt2);
        std::vector<uint32_t> mx(nt2, 0);
        const uint8_t *lut = g_lut.t;
        parallel_for(T, nt2, [&](size_t t) {
            const size_t a = rec0 + m * t / nt2, z = rec0 + m * (t + 1) / nt2;
            uint64_t bpos = tb[t];                              // global base index of the next base
            const uint64_t first_byte = tb[t] / 4, last_byte = tb[t + 1] ? (tb[t + 1] - 1) / 4 : 0;
            uint8_t acc = 0;                                    // bits of the byte being filled
            uint8_t *out = b->packed.data();
            auto flush = [&](uint64_t byte) {
                // bytes shared with a neighbouring task are merged by the caller after the join
                if ((byte == first_byte && (tb[t] & 3)) || (byte == last_byte && (tb[t + 1] & 3) && t + 1 < nt2)) edges[t].push_back(Edge{byte, acc});
                else out[byte] = acc;
                acc = 0;
            };
            uint32_t lmx = 0;
            for (size_t i = a; i < z; i++) {
                const size_t j = i - rec0;
                uint32_t p, l;
                line(4 * i, p, l);
                b->name_pos[j] = (uint32_t)(p + 1); b->name_len[j] = l - 1;      // record name = read.ID[1:] (alignment.go:119)
                line(4 * i + 3, p, l);
                b->qual_pos[j] = p; b->qual_len[j] = l;
                line(4 * i + 1, p, l);
                b->seq_pos[j] = p; b->seq_len[j] = (uint16_t)l;
                lmx = std::max(lmx, l);
                const uint8_t *s = reinterpret_cast<const uint8_t *>(text + p);
                uint32_t x = 0;
                const uint8_t *lut2 = g_lut2.t;
                while (x < l) {
                    // whole packed bytes away from the bytes this task shares with its neighbours: four bases per step, two table
                    // look-ups (reads of a multiple of four bases -- the usual 100 or 150 -- go this way from their first base to their last)
                    if ((bpos & 3) == 0) {
                        while (x + 4 <= l && bpos / 4 > first_byte && bpos / 4 < last_byte) {
                            uint16_t h0, h1;
                            memcpy(&h0, s + x, 2); memcpy(&h1, s + x + 2, 2);
                            const uint8_t c0 = lut2[h0], c1 = lut2[h1];
                            if ((c0 | c1) & 0x80) break;              // a byte other than A C G T: base by base, with the exception list
                            out[bpos / 4] = (uint8_t)((c0 & 15) | ((c1 & 15) << 4));
                            x += 4; bpos += 4;
                        }
                        if (x >= l) break;
                    }
                    // base by base up to the next byte boundary
                    do {
                        const uint8_t c = lut[s[x]];
                        if (c & 0x80) exc[t].emplace_back(bpos, s[x]);
                        acc |= (uint8_t)((c & 3) << (2 * (bpos & 3)));
                        if ((bpos & 3) == 3) flush(bpos / 4);
                        x++; bpos++;
                    } while (x < l && (bpos & 3));
                }
            }
            if (bpos & 3 && bpos > tb[t]) flush((bpos - 1) / 4);
            mx[t] = lmx;
        });
        for (auto &ev : edges)
            for (auto &e : ev) b->packed[e.byte] |= e.val;        // packed was zero-filled
        uint64_t n_exc = 0;
        for (auto &e : exc) n_exc += e.size();
        b->exc_pos.reserve(n_exc); b->exc_byte.reserve(n_exc);
        for (auto &e : exc)
            for (auto &pr : e) { b->exc_pos.push_back(pr.first); b->exc_byte.push_back(pr.second); }
        for (uint32_t v : mx) b->max_len = std::max(b->max_len, v);
        r->n_reads += m;
        r->ready.push_back(b.release());
        rec0 = rec1;
    }
    return GROOT_OK;
}

extern "C" {

int groot_reads_open(const char *const *files, uint32_t n_files, uint32_t n_threads, uint64_t block_bytes, uint32_t max_batch_reads,
                     uint64_t max_batch_bases, groot_reads **out)
{
    if (!out || (n_files && !files)) return set_error(GROOT_E_INVALID, "null argument");
    std::unique_ptr<groot_reads> r(new groot_reads());
    for (uint32_t i = 0; i < n_files; i++) r->files.push_back(files[i]);
    r->n_threads = n_threads ? n_threads : usable_cpus();
    r->block_bytes = (size_t)std::min<uint64_t>(block_bytes ? block_bytes : (256ull << 20), 3ull << 30);
    r->max_batch_reads = max_batch_reads ? max_batch_reads : (1u << 20);
    r->max_batch_bases = max_batch_bases ? max_batch_bases : (uint64_t)r->max_batch_reads * 256;
    for (auto &f : r->files) {       // fail early on a missing file, like os.Open + misc.ErrorCheck
        FILE *t = fopen(f.c_str(), "rb");
        if (!t) return set_error(GROOT_E_IO, "cannot open %s", f.c_str());
        fclose(t);
    }
    r->open_more();
    r->started = true;
    *out = r.release();
    return GROOT_OK;
}

int groot_reads_next(groot_reads *r, groot_reads_batch **out)
{
    if (!r || !out) return set_error(GROOT_E_INVALID, "null argument");
    *out = nullptr;
    while (r->ready.empty() && !r->input_done) {
        if (r->open.empty()) {
            // end of all input: what is left in the carry is whole lines of a last partial record, or nothing
            r->input_done = true;
            if (!r->carry.empty()) {
                std::unique_ptr<RawBlock> b(new RawBlock());
                b->buf.resize(kHeadroom + 1);
                if (int rc = parse_block(r, std::move(b), true)) return rc;
            }
            break;
        }
        Source *s = r->open.front().get();
        std::unique_ptr<RawBlock> b = s->pop();
        if (!b) {
            if (!s->err.empty()) return set_error(GROOT_E_IO, "%s", s->err.c_str());
            s->shutdown();
            r->open.pop_front();
            r->open_more();
            continue;
        }
        if (int rc = parse_block(r, std::move(b), false)) return rc;
    }
    if (!r->ready.empty()) { *out = r->ready.front(); r->ready.pop_front(); }
    return GROOT_OK;
}

void groot_reads_batch_view(const groot_reads_batch *b, groot_reads_view *v)
{
    if (!b || !v) return;
    memset(v, 0, sizeof *v);
    v->n_reads = (uint32_t)b->seq_len.size();
    v->max_len = b->max_len;
    v->n_bases = b->n_bases;
    v->n_exc = b->exc_pos.size();
    v->packed = b->packed.data(); v->seq_len = b->seq_len.data();
    v->exc_pos = b->exc_pos.data(); v->exc_byte = b->exc_byte.data();
    v->text = reinterpret_cast<const uint8_t *>(b->text->buf.data() + b->text->begin);
    v->name_pos = b->name_pos.data(); v->name_len = b->name_len.data(); v->seq_pos = b->seq_pos.data();
    v->qual_pos = b->qual_pos.data(); v->qual_len = b->qual_len.data();
}

void groot_reads_batch_free(groot_reads_batch *b) { delete b; }

uint64_t groot_reads_count(const groot_reads *r) { return r ? r->n_reads : 0; }

void groot_reads_close(groot_reads *r) { delete r; }

} // extern "C"

// bam.cpp -- BAM output of the align path (BGZF-compressed, SAM/BAM v1.5).
//   src/pipeline/boss.go:45-105   setupBAM: @HD VN:1.5, one @SQ per path of every graph, @PG, @RG
//   src/graph/graphio.go:141-154  GetSAMrefs: sam.NewReference(pathName, "", "", Lengths[pathID], nil, nil)
//   src/graph/alignment.go:113-156 the record fields
//   src/pipeline/boss.go:225-240  records written in arrival order, then Close
// The reference writes through github.com/biogo/hts v1.1.0 (not in /root/reference); this writer follows the
// BAM specification for the same field values: next_refID=-1, next_pos=-1 (no mate), tlen=0, no aux tags,
// bin = reg2bin(pos, end).
#include "host_common.hpp"
#include "bgzf_struct.hpp"

#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <chrono>
#include <cstring>
#include <ctime>
#include <thread>

#include <cerrno>
#include <sys/uio.h>
#include <unistd.h>
#include <zlib.h>

using namespace groot;

struct groot_bam {
    FILE *f = nullptr;
    bool own = false;
    std::vector<uint8_t> block;   // uncompressed bytes waiting for the next BGZF block
    std::vector<uint8_t> out;
    uint32_t n_ref = 0;
    unsigned threads = 1;         // BGZF workers for large groot_bam_write calls (bam.NewWriter's write concurrency)
    int level = Z_DEFAULT_COMPRESSION;   // biogo's bgzf.NewWriter default = gzip.DefaultCompression
    uint64_t bytes_out = 0;       // compressed bytes written so far (handed to the file or to the writer thread)
    // The compressed chunks of a batch (groot_bam_write_travs / _batch) are written by a thread of their own, in order, while the
    // workers already build the next batch's records: the gathered writes were 9 of the 31 ms a batch of 4.5 M records kept the
    // collector busy.  Everything else that touches the file waits for that thread to be through (io_drain).
    std::thread io;
    std::mutex io_mu;
    std::condition_variable io_cv;
    std::deque<std::vector<std::vector<uint8_t>>> io_q;
    bool io_busy = false, io_stop = false;
    int io_err = 0;
};

// the writer thread: batches of chunks in the order they were queued, gathered writes straight from the workers' buffers
static void io_loop(groot_bam *b)
{
    const int fd = fileno(b->f);
    std::vector<struct iovec> iov;
    for (;;) {
        std::vector<std::vector<uint8_t>> outs;
        {
            std::unique_lock<std::mutex> lk(b->io_mu);
            b->io_cv.wait(lk, [&] { return b->io_stop || !b->io_q.empty(); });
            if (b->io_q.empty()) return;
            outs = std::move(b->io_q.front());
            b->io_q.pop_front();
            b->io_busy = true;
        }
        b->io_cv.notify_all();
        int err = 0;
        bool broken;
        { std::lock_guard<std::mutex> lk(b->io_mu); broken = b->io_err != 0; }
        // (after a failed write the stream is broken for good: what is still queued is dropped, not appended behind the gap)
        for (size_t c = 0; c < outs.size() && !err && !broken;) {
            iov.clear();
            for (; c < outs.size() && iov.size() < 512; c++)
                if (!outs[c].empty()) iov.push_back({outs[c].data(), outs[c].size()});
            size_t first = 0;
            while (first < iov.size()) {
                const ssize_t w = writev(fd, iov.data() + first, (int)(iov.size() - first));
                if (w < 0) { if (errno == EINTR) continue; err = GROOT_E_IO; break; }
                size_t left = (size_t)w;
                while (first < iov.size() && left >= iov[first].iov_len) { left -= iov[first].iov_len; first++; }
                if (first < iov.size() && left) { iov[first].iov_base = (char *)iov[first].iov_base + left; iov[first].iov_len -= left; }
            }
        }
        {
            std::lock_guard<std::mutex> lk(b->io_mu);
            b->io_busy = false;
            if (err && !b->io_err) b->io_err = err;
        }
        b->io_cv.notify_all();
    }
}

// everything queued is in the file (or has failed)
static int io_drain(groot_bam *b)
{
    if (!b->io.joinable()) return GROOT_OK;
    std::unique_lock<std::mutex> lk(b->io_mu);
    b->io_cv.wait(lk, [&] { return b->io_q.empty() && !b->io_busy; });
    return b->io_err ? set_error(b->io_err, "BAM write failed") : GROOT_OK;
}

static int io_push(groot_bam *b, std::vector<std::vector<uint8_t>> &&outs)
{
    if (fflush(b->f) != 0) return set_error(GROOT_E_IO, "BAM write failed");     // (what stdio still holds goes first)
    if (!b->io.joinable()) b->io = std::thread(io_loop, b);
    {
        std::unique_lock<std::mutex> lk(b->io_mu);
        b->io_cv.wait(lk, [&] { return b->io_q.size() < 2; });               // (at most two batches of chunks waiting)
        if (b->io_err) return set_error(b->io_err, "BAM write failed");
        b->io_q.push_back(std::move(outs));
    }
    b->io_cv.notify_all();
    return GROOT_OK;
}

static const size_t kBgzfBlock = 0xff00;   // max uncompressed payload per block
static const int kBamStructural = -2;      // groot_bam_set_level: members written from the records' structure (bgzf_struct.hpp)

// one BGZF member for data[0..n) into out; returns its size or a negative error
static long compress_block(const uint8_t *data, size_t n, std::vector<uint8_t> &out, int level = Z_DEFAULT_COMPRESSION)
{
    if (level == kBamStructural) level = 1;                // (what has no record structure -- header, odd pieces -- goes through zlib)
    out.resize(18 + compressBound((uLong)n) + 8);
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return GROOT_E_NOMEM;
    zs.next_in = const_cast<Bytef *>(data);
    zs.avail_in = (uInt)n;
    zs.next_out = out.data() + 18;
    zs.avail_out = (uInt)(out.size() - 18 - 8);
    const int rc = deflate(&zs, Z_FINISH);
    const size_t clen = zs.total_out;
    deflateEnd(&zs);
    if (rc != Z_STREAM_END) return GROOT_E_IO;
    static const uint8_t hdr[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
    memcpy(out.data(), hdr, 16);
    const size_t total = 18 + clen + 8;
    if (total > 0x10000) return GROOT_E_IO;
    const uint16_t bsize = (uint16_t)(total - 1);
    out[16] = (uint8_t)bsize; out[17] = (uint8_t)(bsize >> 8);
    const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), data, (uInt)n), isize = (uint32_t)n;
    memcpy(out.data() + 18 + clen, &crc, 4);
    memcpy(out.data() + 18 + clen + 4, &isize, 4);
    return (long)total;
}

static int flush_block(groot_bam *b, const uint8_t *data, size_t n)
{
    const long total = compress_block(data, n, b->out, b->level);
    if (total < 0) return set_error((int)total, "BGZF compression failed");
    if (int rc = io_drain(b)) return rc;
    if (fwrite(b->out.data(), 1, (size_t)total, b->f) != (size_t)total) return set_error(GROOT_E_IO, "BAM write failed");
    b->bytes_out += (uint64_t)total;
    return GROOT_OK;
}

static int put(groot_bam *b, const void *p, size_t n)
{
    const uint8_t *s = (const uint8_t *)p;
    while (n) {
        const size_t room = kBgzfBlock - b->block.size();
        const size_t m = std::min(room, n);
        b->block.insert(b->block.end(), s, s + m);
        s += m; n -= m;
        if (b->block.size() == kBgzfBlock) {
            if (int rc = flush_block(b, b->block.data(), b->block.size())) return rc;
            b->block.clear();
        }
    }
    return GROOT_OK;
}
static int put32(groot_bam *b, int32_t v) { return put(b, &v, 4); }

// SAM spec 5.3
static int reg2bin(int64_t beg, int64_t end)
{
    --end;
    if (beg >> 14 == end >> 14) return (int)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (int)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (int)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (int)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (int)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}

extern "C" {

int groot_bam_open(const char *path, const groot_index_view *ix, const char *date, groot_bam **out)
{
    if (!ix || !out) return set_error(GROOT_E_INVALID, "null argument");
    auto b = new groot_bam();
    if (!path || strcmp(path, "-") == 0) b->f = stdout;   // boss.go:94-96
    else {
        b->f = fopen(path, "wb");
        b->own = true;
        if (!b->f) { delete b; return set_error(GROOT_E_IO, "could not open file for BAM writing: %s", path); }
    }
    char stamp[64];
    if (!date) {
        time_t now = time(nullptr);
        struct tm tmv;
        gmtime_r(&now, &tmv);
        strftime(stamp, sizeof stamp, "%Y-%m-%dT%H:%M:%SZ", &tmv);
        date = stamp;
    }
    std::string text = "@HD\tVN:1.5\tSO:unknown\n";
    for (uint32_t p = 0; p < ix->n_paths; p++) {
        text += "@SQ\tSN:";
        text.append(ix->path_names + ix->path_name_off[p], ix->path_name_off[p + 1] - ix->path_name_off[p]);
        text += "\tLN:" + std::to_string(ix->path_len[p]) + "\n";
    }
    text += std::string("@RG\tID:readsID\tDT:") + date + "\tPG:groot align\tPI:1000\tPL:illumina\tSM:sampleID\n";   // boss.go:58
    text += std::string("@PG\tID:1\tPN:groot\tCL:groot align\tVN:") + groot_host_version() + "\n";                   // boss.go:55
    int rc = put(b, "BAM\1", 4);
    if (!rc) rc = put32(b, (int32_t)text.size());
    if (!rc) rc = put(b, text.data(), text.size());
    if (!rc) rc = put32(b, (int32_t)ix->n_paths);
    for (uint32_t p = 0; p < ix->n_paths && !rc; p++) {
        const uint32_t nl = ix->path_name_off[p + 1] - ix->path_name_off[p];
        rc = put32(b, (int32_t)nl + 1);
        if (!rc) rc = put(b, ix->path_names + ix->path_name_off[p], nl);
        if (!rc) rc = put(b, "\0", 1);
        if (!rc) rc = put32(b, (int32_t)ix->path_len[p]);
    }
    if (rc) { if (b->own) fclose(b->f); delete b; return rc; }
    b->n_ref = ix->n_paths;
    *out = b;
    return GROOT_OK;
}

static size_t record_size(const groot_aln_record &r)
{
    const uint32_t nc = 1 + (r.start_clip ? 1 : 0) + (r.end_clip ? 1 : 0);
    return 4 + 32 + (r.name_len + 1) + 4 * nc + (r.seq_len + 1) / 2 + r.seq_len;
}

struct Nt16 {
    uint8_t t[256];
    Nt16()
    {
        static const char *code = "=ACMGRSVTWYHKDBN";
        memset(t, 15, sizeof t);
        for (int i = 0; i < 16; i++) { t[(uint8_t)code[i]] = (uint8_t)i; t[(uint8_t)tolower(code[i])] = (uint8_t)i; }
    }
};

// appends the BAM encoding of recs[i0, i1) to buf
} // extern "C"

// a byte buffer that grows without touching what it does not hold yet (std::vector::resize value-initialises: 230 bytes zeroed only to be
// overwritten, per record): the interface format_records needs
struct RawBuf {
    std::vector<uint8_t> v;
    size_t n = 0;
    size_t size() const { return n; }
    uint8_t *data() { return v.data(); }
    const uint8_t *data() const { return v.data(); }
    void clear() { n = 0; }
    void resize(size_t m)
    {
        if (m > v.size()) v.resize(std::max<size_t>(m + m / 2, 1u << 16));
        n = m;
    }
};

template <class Buf>
static void format_records(const groot_aln_record *recs, uint64_t i0, uint64_t i1, Buf &buf, std::vector<uint32_t> *starts = nullptr)
{
    static const Nt16 nt;
    const uint8_t *nt16 = nt.t;
    const uint8_t *prev_seq = nullptr;
    uint32_t prev_len = 0;
    size_t prev_at = 0;   // where the previous record's packed sequence sits in buf
    for (uint64_t i = i0; i < i1; i++) {
        const groot_aln_record &r = recs[i];
        uint32_t cigar[3];
        uint32_t nc = 0;
        if (r.start_clip) cigar[nc++] = ((uint32_t)r.start_clip << 4) | 5;   // H (alignment.go:132-134)
        cigar[nc++] = (r.seq_len << 4) | 0;                                   // M (:135)
        if (r.end_clip) cigar[nc++] = ((uint32_t)r.end_clip << 4) | 5;       // H (:136-138)
        const uint16_t flag = (uint16_t)((r.secondary ? 0x100 : 0) | (r.reverse ? 0x10 : 0));   // :147-152
        const int bin = reg2bin(r.pos, (int64_t)r.pos + (r.seq_len ? r.seq_len : 1));
        const uint32_t l_name = r.name_len + 1;
        const uint32_t body = 32 + l_name + 4 * nc + (r.seq_len + 1) / 2 + r.seq_len;
        const size_t at = buf.size();
        if (starts) starts->push_back((uint32_t)at);
        buf.resize(at + 4 + body);
        uint8_t *p = buf.data() + at;
        auto w32 = [&](uint32_t v) { memcpy(p, &v, 4); p += 4; };
        w32(body);
        w32(r.ref_id);
        w32(r.pos);
        w32(((uint32_t)bin << 16) | (30u << 8) | l_name);                     // MAPQ 30 (:143)
        w32(((uint32_t)flag << 16) | nc);
        w32(r.seq_len);
        w32((uint32_t)-1);                                                    // next_refID
        w32((uint32_t)-1);                                                    // next_pos
        w32(0);                                                               // tlen
        memcpy(p, r.name, r.name_len); p += r.name_len; *p++ = 0;
        for (uint32_t c = 0; c < nc; c++) w32(cigar[c]);
        const size_t seq_at = (size_t)(p - buf.data());
        if (r.seq == prev_seq && r.seq_len == prev_len) {
            // the records of one read share Seq/Qual (one per path of the traversal): reuse the packed bases
            memcpy(p, buf.data() + prev_at, (r.seq_len + 1) / 2);
            p += (r.seq_len + 1) / 2;
        } else {
            for (uint32_t j = 0; j < r.seq_len; j += 2) {
                const uint8_t hi = nt16[r.seq[j]], lo = j + 1 < r.seq_len ? nt16[r.seq[j + 1]] : 0;
                *p++ = (uint8_t)(hi << 4 | lo);
            }
        }
        prev_seq = r.seq; prev_len = r.seq_len; prev_at = seq_at;
        if (r.qual) memcpy(p, r.qual, r.seq_len);                             // raw bytes, not Phred-33 corrected (:121)
        else memset(p, 0xff, r.seq_len);
    }
}

extern "C" {

int groot_bam_set_threads(groot_bam *b, uint32_t n_threads)
{
    if (!b) return set_error(GROOT_E_INVALID, "null argument");
    b->threads = n_threads ? n_threads : usable_cpus();
    return GROOT_OK;
}

int groot_bam_write(groot_bam *b, const groot_aln_record *recs, uint64_t n)
{
    if (!b || (n && !recs)) return set_error(GROOT_E_INVALID, "null argument");
    for (uint64_t i = 0; i < n; i++) {
        if (recs[i].ref_id >= b->n_ref) return set_error(GROOT_E_INVALID, "record %llu: reference id out of range", (unsigned long long)i);
        if (recs[i].name_len > 254) return set_error(GROOT_E_FORMAT, "read name longer than 254 characters");
    }
    if (b->threads > 1 && n >= 1024) {
        // large call: cut the records into BGZF-block-sized chunks at record boundaries; workers format + deflate
        // their chunks, the blocks are written in order (what bam.NewWriter's write concurrency does)
        if (!b->block.empty()) {
            if (int rc = flush_block(b, b->block.data(), b->block.size())) return rc;
            b->block.clear();
        }
        std::vector<uint64_t> cut{0};
        size_t acc = 0;
        for (uint64_t i = 0; i < n; i++) {
            const size_t sz = record_size(recs[i]);
            if (acc && acc + sz > kBgzfBlock) { cut.push_back(i); acc = 0; }
            acc += sz;
        }
        cut.push_back(n);
        const size_t n_chunks = cut.size() - 1;
        std::vector<std::vector<uint8_t>> outs(n_chunks);
        std::vector<long> sizes(n_chunks, 0);
        std::atomic<size_t> next{0};
        auto work = [&]() {
            std::vector<uint8_t> raw;
            for (;;) {
                const size_t c = next.fetch_add(1);
                if (c >= n_chunks) break;
                raw.clear();
                format_records(recs, cut[c], cut[c + 1], raw);
                if (raw.size() > kBgzfBlock) {   // a single huge record: fall back to splitting its bytes
                    sizes[c] = -1000 - (long)c;
                    continue;
                }
                sizes[c] = compress_block(raw.data(), raw.size(), outs[c], b->level);
            }
        };
        const unsigned nt = (unsigned)std::min<size_t>(b->threads, n_chunks);
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nt; t++) th.emplace_back(work);
        work();
        for (auto &t : th) t.join();
        if (int rc = io_drain(b)) return rc;      // (this path writes through stdio: the writer thread has to be through first)
        for (size_t c = 0; c < n_chunks; c++) {
            if (sizes[c] <= -1000) {             // oversized single record: serial path keeps the stream valid
                std::vector<uint8_t> raw;
                format_records(recs, cut[c], cut[c + 1], raw);
                if (int rc = put(b, raw.data(), raw.size())) return rc;
                if (!b->block.empty()) {
                    if (int rc = flush_block(b, b->block.data(), b->block.size())) return rc;
                    b->block.clear();
                }
                continue;
            }
            if (sizes[c] < 0) return set_error((int)sizes[c], "BGZF compression failed");
            if (fwrite(outs[c].data(), 1, (size_t)sizes[c], b->f) != (size_t)sizes[c]) return set_error(GROOT_E_IO, "BAM write failed");
            b->bytes_out += (uint64_t)sizes[c];
        }
        return GROOT_OK;
    }
    std::vector<uint8_t> buf;
    format_records(recs, 0, n, buf);
    return put(b, buf.data(), buf.size());
}

} // extern "C"

// one read of a batch as the BAM writer needs it
struct ReadRef {
    const char *name; uint32_t name_len;
    const uint8_t *seq, *qual;       // qual may be NULL
    uint32_t len, qual_len;          // qual_len < len: the missing quality bytes are written as '!' (the reference never checks the two lengths, seqio.go:175-178)
};

// traversal records of one batch -> sam.Records (alignment.go:113-156) -> BGZF blocks, in parallel over chunks of
// traversals; blocks are written in traversal order = read order.  read(r, ref) fills the read's fields, false if r is
// out of range.
template <class ReadFn>
static int write_travs_impl(groot_bam *b, const groot_index_view *ix, ReadFn read, const groot_trav *travs, const void *masks_v,
                            const uint32_t *mask_ckpt, uint64_t n_trav, uint64_t *n_records)
{
    if (n_records) *n_records = 0;
    if (!n_trav) return GROOT_OK;
    if (!b->block.empty()) {
        if (int rc = flush_block(b, b->block.data(), b->block.size())) return rc;
        b->block.clear();
    }
    const uint32_t pw = ix->path_words;
    const uint64_t kChunk = 256;                                  // traversals per task
    const size_t n_chunks = (size_t)((n_trav + kChunk - 1) / kChunk);
    std::vector<std::vector<uint8_t>> outs(n_chunks);
    std::vector<int> errs(n_chunks, 0);
    std::vector<uint64_t> nrec(n_chunks, 0);
    std::atomic<size_t> next{0};
    const int level = b->level;
    static const bool bam_stats = getenv("GROOT_BAM_STATS") != nullptr;
    std::atomic<uint64_t> ns_build{0}, ns_format{0}, ns_encode{0};
    auto now_ns = []() { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const uint64_t t_begin = now_ns();
    auto work = [&]() {
        uint64_t tb = 0, tf = 0, te = 0;
        RawBuf raw;
        std::vector<uint8_t> blk, rcs, rcq, padq;
        std::vector<uint32_t> starts, rel;                       // record starts in raw (structural BGZF only)
        std::vector<uint8_t> follow;                             // ... and: the record is a patched copy of the one before it (bgzf_struct.hpp)
        std::vector<groot_aln_record> recs;
        for (;;) {
            const size_t c = next.fetch_add(1);
            if (c >= n_chunks) { ns_build += tb; ns_format += tf; ns_encode += te; break; }
            raw.clear(); starts.clear(); follow.clear();
            const uint64_t t0 = c * kChunk, t1 = std::min<uint64_t>(n_trav, t0 + kChunk);
            uint64_t moff = mask_ckpt ? mask_ckpt[c] : 0;        // compact path sets: a checkpoint per chunk (kChunk = 256 traversals)
            for (uint64_t t = t0; t < t1; t++) {
                const uint64_t q0 = bam_stats ? now_ns() : 0;
                const groot_trav &tr = travs[t];
                ReadRef rd;
                if (!read(tr.read_id, rd) || tr.node >= ix->n_nodes || tr.graph_id >= ix->n_graphs) { errs[c] = GROOT_E_INVALID; break; }
                const uint64_t len = rd.len;
                const uint8_t *sq = rd.seq, *ql = rd.qual;
                if (ql && rd.qual_len < len) {                    // pad a short quality line to the sequence
                    padq.assign(len, '!');
                    memcpy(padq.data(), ql, rd.qual_len);
                    ql = padq.data();
                }
                if (tr.flags & GROOT_TRAV_RC) {                   // seqio.go:120-133
                    rcs.resize(len); rcq.resize(len);
                    for (uint64_t i = 0; i < len; i++) {
                        const uint8_t ch = sq[len - 1 - i];
                        rcs[i] = ch == 'A' ? 'T' : ch == 'T' ? 'A' : ch == 'C' ? 'G' : ch == 'G' ? 'C' : ch == 'N' ? 'N' : 0;
                        rcq[i] = ql ? ql[len - 1 - i] : 0xff;
                    }
                    sq = rcs.data(); ql = ql ? rcq.data() : nullptr;
                }
                const uint8_t sc = (tr.flags & GROOT_TRAV_START_CLIP) ? 1 : 0, ec = (tr.flags & GROOT_TRAV_END_CLIP) ? 1 : 0;
                bool first = (tr.flags & GROOT_TRAV_FIRST) != 0;
                recs.clear();
                uint64_t n_here = 0;                                  // records of this traversal so far (structural path: formatted as they come)
                const uint32_t np0 = ix->node_np_off[tr.node], np1 = ix->node_np_off[tr.node + 1];
                // (compact path sets: max(1, ceil(paths / 8)) bytes per traversal, widened here; else path_words words)
                uint64_t wide[16];
                uint32_t gw = pw;
                const uint64_t *mk = static_cast<const uint64_t *>(masks_v) + t * pw;
                if (mask_ckpt) {
                    const uint32_t nb = std::min<uint32_t>(sizeof wide, std::max<uint32_t>(1, (ix->graph_path_off[tr.graph_id + 1] - ix->graph_path_off[tr.graph_id] + 7) / 8));
                    gw = (nb + 7) / 8;
                    wide[gw - 1] = 0;
                    memcpy(wide, static_cast<const uint8_t *>(masks_v) + moff, nb);
                    mk = wide;
                    moff += nb;
                }
                uint32_t jn = np0;                                   // the node's (path, position) pairs ascend by path: walked along with the set bits
                for (uint32_t w = 0; w < gw; w++) {
                    uint64_t m = mk[w];
                    while (m) {
                        const uint32_t p = w * 64 + (uint32_t)__builtin_ctzll(m);
                        m &= m - 1;
                        uint32_t pos = 0;
                        while (jn < np1 && ix->np_path[jn] < p) jn++;
                        if (jn < np1 && ix->np_path[jn] == p) pos = ix->np_pos[jn] + tr.offset;     // alignment.go:296
                        else
                            for (uint32_t j = np0; j < np1; j++)           // (pairs not in path order: look through all of them)
                                if (ix->np_path[j] == p) { pos = ix->np_pos[j] + tr.offset; break; }
                        const uint32_t ref_id = ix->graph_path_off[tr.graph_id] + p;
                        const uint32_t seq_len = (uint32_t)len - sc - ec;                                // alignment.go:117-122
                        if (rd.name_len > 254 || ref_id >= b->n_ref) { errs[c] = GROOT_E_FORMAT; break; }
                        if (level == kBamStructural && n_here) {
                            // the next path of the same traversal: the record before it with refID, pos, bin and the Secondary flag
                            // patched (alignment.go:113-156 builds them from the same read and the same CIGAR) -- no field-by-field assembly
                            const size_t prev_at = starts.back(), L = raw.size() - prev_at, at2 = raw.size();
                            raw.resize(at2 + L);
                            uint8_t *q = raw.data() + at2;
                            memcpy(q, raw.data() + prev_at, L);
                            const uint16_t bin = (uint16_t)reg2bin(pos, (int64_t)pos + (seq_len ? seq_len : 1));
                            uint16_t flag;
                            memcpy(&flag, q + 18, 2);
                            flag |= 0x100;                                                               // alignment.go:147-149
                            memcpy(q + 4, &ref_id, 4); memcpy(q + 8, &pos, 4); memcpy(q + 14, &bin, 2); memcpy(q + 18, &flag, 2);
                            starts.push_back((uint32_t)at2); follow.push_back(1);
                            n_here++;
                            continue;
                        }
                        groot_aln_record rec;
                        rec.name = rd.name;
                        rec.name_len = rd.name_len;
                        rec.seq = sq; rec.qual = ql;
                        rec.seq_len = seq_len;
                        rec.ref_id = ref_id;
                        rec.pos = pos; rec.start_clip = sc; rec.end_clip = ec;
                        rec.reverse = (tr.flags & GROOT_TRAV_RC) ? 1 : 0;
                        rec.secondary = first ? 0 : 1;                                            // alignment.go:147-149
                        first = false;
                        if (level == kBamStructural) {
                            format_records(&rec, 0, 1, raw, &starts);
                            follow.push_back(0);
                            n_here++;
                            continue;
                        }
                        recs.push_back(rec);
                    }
                }
                if (errs[c]) break;
                nrec[c] += level == kBamStructural ? n_here : recs.size();
                const uint64_t q1 = bam_stats ? now_ns() : 0;
                tb += q1 - q0;
                if (level != kBamStructural) format_records(recs.data(), 0, recs.size(), raw, nullptr);   // rcs/rcq/padq stay valid until here
                if (bam_stats) tf += now_ns() - q1;
            }
            if (errs[c]) continue;
            const uint64_t q2 = bam_stats ? now_ns() : 0;
            if (level == kBamStructural) {
                // members end at record boundaries; each is written from what the records have in common (bgzf_struct.hpp).  A piece
                // that cannot be (a record of 32 KB or more) goes through zlib -1 like everything else used to.
                size_t r0 = 0;
                while (r0 < starts.size()) {
                    size_t r1 = r0 + 1;
                    const size_t a = starts[r0];
                    // (0xd000, not the 0xff00 a member may inflate to: first records go out as stored blocks, so the member is a little
                    // LARGER than its contents when few records follow another -- 5 bytes per stored block -- and must stay below 64 KB)
                    const size_t kCap = 0xd000;
                    while (r1 < starts.size() && (r1 + 1 < starts.size() ? starts[r1 + 1] : raw.size()) - a <= kCap) r1++;
                    const size_t e = r1 < starts.size() ? starts[r1] : raw.size();
                    bool ok = e - a <= kBgzfBlock;
                    if (ok) {
                        rel.resize(r1 - r0);
                        for (size_t x = r0; x < r1; x++) rel[x - r0] = starts[x] - (uint32_t)a;
                        ok = bgzf_member_structural(raw.data() + a, e - a, rel.data(), rel.size(), outs[c], follow.data() + r0);
                    }
                    if (!ok)
                        for (size_t o = a; o < e; o += kBgzfBlock) {
                            const long sz = compress_block(raw.data() + o, std::min(kBgzfBlock, e - o), blk, 1);
                            if (sz < 0) { errs[c] = (int)sz; break; }
                            outs[c].insert(outs[c].end(), blk.begin(), blk.begin() + sz);
                        }
                    r0 = r1;
                }
                if (bam_stats) te += now_ns() - q2;
                continue;
            }
            for (size_t o = 0; o < raw.size(); o += kBgzfBlock) {   // records may span BGZF blocks
                const long sz = compress_block(raw.data() + o, std::min(kBgzfBlock, raw.size() - o), blk, level);
                if (sz < 0) { errs[c] = (int)sz; break; }
                outs[c].insert(outs[c].end(), blk.begin(), blk.begin() + sz);
            }
        }
    };
    const unsigned nt = (unsigned)std::max<size_t>(1, std::min<size_t>(b->threads, n_chunks));
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; t++) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
    const uint64_t t_par = now_ns();
    uint64_t total = 0;
    for (size_t c = 0; c < n_chunks; c++) {
        if (errs[c]) return set_error(errs[c], "could not build the BAM records of traversal chunk %zu", c);
        total += nrec[c];
    }
    // the compressed chunks go out in order with gathered writes straight from the workers' buffers (no stdio copy), on the writer thread
    uint64_t bytes = 0;
    for (auto &o : outs) bytes += o.size();
    if (int rc = io_push(b, std::move(outs))) return rc;
    b->bytes_out += bytes;
    if (bam_stats)
        fprintf(stderr, "[groot bam] %llu records: threads %u, parallel part %.1f ms (summed over threads: records %.1f, format %.1f, encode %.1f ms), write %.1f ms\n",
                (unsigned long long)total, nt, (t_par - t_begin) / 1e6, ns_build.load() / 1e6, ns_format.load() / 1e6, ns_encode.load() / 1e6, (now_ns() - t_par) / 1e6);
    if (n_records) *n_records = total;
    return GROOT_OK;
}

extern "C" {

int groot_bam_write_travs(groot_bam *b, const groot_index_view *ix, const groot_read_batch *rb, const groot_trav *travs,
                          const uint64_t *masks, uint64_t n_trav, uint64_t *n_records)
{
    if (!b || !ix || !rb || (n_trav && (!travs || !masks))) return set_error(GROOT_E_INVALID, "null argument");
    auto read = [rb](uint32_t read_id, ReadRef &o) -> bool {
        const uint32_t r = read_id - rb->first_read_id;
        if (r >= rb->n_reads) return false;
        const uint64_t s0 = rb->seq_off[r];
        o.len = o.qual_len = (uint32_t)(rb->seq_off[r + 1] - s0);
        o.seq = rb->seq + s0; o.qual = rb->qual ? rb->qual + s0 : nullptr;
        o.name = rb->names + rb->name_off[r];
        o.name_len = (uint32_t)(rb->name_off[r + 1] - rb->name_off[r]);
        return true;
    };
    return write_travs_impl(b, ix, read, travs, masks, nullptr, n_trav, n_records);
}

int groot_bam_write_batch(groot_bam *b, const groot_index_view *ix, const groot_reads_view *rv, uint32_t first_read_id, const groot_trav *travs,
                          const void *masks, const uint32_t *mask_ckpt, uint64_t n_trav, uint64_t *n_records)
{
    if (!b || !ix || !rv || (n_trav && (!travs || !masks))) return set_error(GROOT_E_INVALID, "null argument");
    auto read = [rv, first_read_id](uint32_t read_id, ReadRef &o) -> bool {
        const uint32_t r = read_id - first_read_id;
        if (r >= rv->n_reads) return false;
        o.len = rv->seq_len[r];
        o.seq = rv->text + rv->seq_pos[r];
        o.qual = rv->text + rv->qual_pos[r];
        o.qual_len = std::min<uint32_t>(rv->qual_len[r], o.len);
        o.name = reinterpret_cast<const char *>(rv->text + rv->name_pos[r]);
        o.name_len = rv->name_len[r];
        return true;
    };
    return write_travs_impl(b, ix, read, travs, masks, mask_ckpt, n_trav, n_records);
}

int groot_bam_set_level(groot_bam *b, int level)
{
    if (!b) return set_error(GROOT_E_INVALID, "null argument");
    if (level < -2 || level > 9) return set_error(GROOT_E_INVALID, "BGZF compression level %d not in [-2, 9]", level);
    b->level = level;
    return GROOT_OK;
}

uint64_t groot_bam_bytes_written(const groot_bam *b) { return b ? b->bytes_out : 0; }

int groot_bam_close(groot_bam *b)
{
    if (!b) return GROOT_OK;
    int rc = io_drain(b);
    if (!rc && !b->block.empty()) rc = flush_block(b, b->block.data(), b->block.size());
    if (b->io.joinable()) {
        { std::lock_guard<std::mutex> lk(b->io_mu); b->io_stop = true; }
        b->io_cv.notify_all();
        b->io.join();
    }
    static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (!rc && fwrite(eof, 1, 28, b->f) != 28) rc = set_error(GROOT_E_IO, "BAM write failed");
    if (b->own) { if (fclose(b->f) != 0 && !rc) rc = set_error(GROOT_E_IO, "BAM close failed"); }
    else fflush(b->f);
    delete b;
    return rc;
}

} // extern "C"

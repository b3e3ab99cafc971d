// fastq.cpp -- FASTQ input for the align path.
//   src/pipeline/sketch.go:41-77   DataStreamer: line scanner over stdin / files, gzip if the name ends in .gz
//   src/pipeline/sketch.go:213-236 FastqHandler: every four lines form one read
//   src/seqio/seqio.go:173-188     NewFASTQread: line 1 must start with '@'; no other check
// Lines from consecutive files form ONE stream (the reference's four-line grouping carries across files).
#include "host_common.hpp"

#include <algorithm>
#include <cstring>
#include <thread>
#include <utility>
#include <zlib.h>

using namespace groot;

struct groot_fastq {
    std::vector<std::string> files;
    size_t next_file = 0;
    gzFile fh = nullptr;       // gzread handles plain files transparently
    bool use_stdin = false, eof = false;
    std::vector<char> buf;
    size_t pos = 0, fill = 0;
    std::string line[4];
    int have = 0;              // lines of the current record already collected
    uint64_t n_reads = 0;
    // a record that did not fit the caller's buffers in the previous call
    bool pending = false;
};

// 0 = next input opened, 1 = no more inputs, -1 = open failed
static int open_next(groot_fastq *fq)
{
    if (fq->fh) { gzclose(fq->fh); fq->fh = nullptr; }
    if (fq->use_stdin) {
        if (fq->next_file) return 1;
        fq->next_file = 1;
        fq->fh = gzdopen(0, "rb");
        return fq->fh ? 0 : -1;
    }
    if (fq->next_file >= fq->files.size()) return 1;
    fq->fh = gzopen(fq->files[fq->next_file++].c_str(), "rb");
    return fq->fh ? 0 : -1;
}

// next line without its terminator ("\n" or "\r\n", bufio.ScanLines); false at end of all input
static int read_line(groot_fastq *fq, std::string &out)
{
    out.clear();
    for (;;) {
        if (fq->pos == fq->fill) {
            if (fq->eof) return out.empty() ? 0 : 1;
            int n = fq->fh ? gzread(fq->fh, fq->buf.data(), (unsigned)fq->buf.size()) : 0;
            if (n < 0) return set_error(GROOT_E_IO, "read error in FASTQ input");
            if (n == 0) {
                // a final line without '\n' ends at the file end, like bufio.Scanner at EOF
                const bool had = !out.empty();
                const int on = open_next(fq);
                if (on < 0) return set_error(GROOT_E_IO, "cannot open %s", fq->use_stdin ? "stdin" : fq->files[fq->next_file - 1].c_str());
                if (on > 0) fq->eof = true;
                if (had) { if (!out.empty() && out.back() == '\r') out.pop_back(); return 1; }
                continue;
            }
            fq->pos = 0; fq->fill = (size_t)n;
        }
        const char *b = fq->buf.data() + fq->pos;
        const char *nl = (const char *)memchr(b, '\n', fq->fill - fq->pos);
        if (nl) {
            out.append(b, nl - b);
            fq->pos += (size_t)(nl - b) + 1;
            if (!out.empty() && out.back() == '\r') out.pop_back();
            return 1;
        }
        out.append(b, fq->fill - fq->pos);
        fq->pos = fq->fill;
    }
}

extern "C" {

int groot_fastq_open(const char *const *files, uint32_t n_files, groot_fastq **out)
{
    if (!out || (n_files && !files)) return set_error(GROOT_E_INVALID, "null argument");
    auto fq = new groot_fastq();
    fq->buf.resize(1 << 20);
    fq->use_stdin = n_files == 0;
    for (uint32_t i = 0; i < n_files; i++) fq->files.push_back(files[i]);
    if (open_next(fq) != 0) {
        std::string f = fq->use_stdin ? "stdin" : fq->files[0];
        delete fq;
        return set_error(GROOT_E_IO, "cannot open %s", f.c_str());
    }
    *out = fq;
    return GROOT_OK;
}

int64_t groot_fastq_next_batch(groot_fastq *fq, uint32_t max_reads, uint8_t *seq, uint8_t *qual, uint64_t *seq_off, uint64_t seq_cap,
                               char *names, uint64_t *name_off, uint64_t name_cap)
{
    if (!fq || !seq || !seq_off || !names || !name_off) return set_error(GROOT_E_INVALID, "null argument");
    uint32_t n = 0;
    seq_off[0] = 0;
    name_off[0] = 0;
    while (n < max_reads) {
        if (!fq->pending) {
            while (fq->have < 4) {
                int rc = read_line(fq, fq->line[fq->have]);
                if (rc < 0) return rc;
                if (rc == 0) break;
                // an empty line reaches FastqHandler.Run as nil (append([]byte(nil), ...) of no bytes): it cannot fill l1, l2 or l3
                // (sketch.go:217-222) -- only the fourth line is taken as it comes
                if (fq->have < 3 && fq->line[fq->have].empty()) continue;
                fq->have++;
            }
            if (fq->have < 4) break;   // a trailing partial record is dropped, as in FastqHandler.Run
            if (fq->line[0].empty() || fq->line[0][0] != '@')   // seqio.go:179-181
                return set_error(GROOT_E_FORMAT, "read ID in fastq file does not begin with @: %s", fq->line[0].c_str());
        }
        const std::string &id = fq->line[0], &s = fq->line[1], &q = fq->line[3];
        if (seq_off[n] + s.size() > seq_cap || name_off[n] + id.size() - 1 > name_cap) {
            if (n == 0) return set_error(GROOT_E_NOSPACE, "a single read does not fit the batch buffers");
            fq->pending = true;
            break;
        }
        memcpy(seq + seq_off[n], s.data(), s.size());
        if (qual) {
            // Qual is carried raw; the reference never checks len(l2)==len(l4) (seqio.go:175-178): pad / cut to the sequence
            const size_t m = std::min(s.size(), q.size());
            memcpy(qual + seq_off[n], q.data(), m);
            if (m < s.size()) memset(qual + seq_off[n] + m, '!', s.size() - m);
        }
        memcpy(names + name_off[n], id.data() + 1, id.size() - 1);   // record name = read.ID[1:] (alignment.go:119)
        seq_off[n + 1] = seq_off[n] + s.size();
        name_off[n + 1] = name_off[n] + id.size() - 1;
        n++;
        fq->n_reads++;
        fq->have = 0;
        fq->pending = false;
    }
    return (int64_t)n;
}

void groot_fastq_close(groot_fastq *fq)
{
    if (!fq) return;
    if (fq->fh) gzclose(fq->fh);
    delete fq;
}

} // extern "C"

// ---- 2-bit packing for groot_hip_submit_packed ------------------------------------------------------------
extern "C" int groot_host_pack_reads(const uint8_t *seq, uint64_t n_bases, uint8_t *packed, uint64_t *exc_pos, uint8_t *exc_byte,
                                     uint64_t exc_cap, uint64_t *n_exc, uint32_t n_threads)
{
    if ((n_bases && (!seq || !packed)) || !n_exc) return groot::set_error(GROOT_E_INVALID, "null argument");
    unsigned nt = n_threads ? n_threads : groot::usable_cpus();
    const uint64_t n_quads = (n_bases + 3) / 4;
    nt = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(nt, n_quads / 65536 + 1));
    std::vector<std::vector<std::pair<uint64_t, uint8_t>>> exc(nt);
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++)
        th.emplace_back([&, t]() {
            const uint64_t q0 = n_quads * t / nt, q1 = n_quads * (t + 1) / nt;   // whole output bytes per thread
            for (uint64_t q = q0; q < q1; q++) {
                unsigned v = 0;
                for (unsigned b = 0; b < 4; b++) {
                    const uint64_t i = q * 4 + b;
                    if (i >= n_bases) break;
                    const uint8_t c = seq[i];
                    v |= ((c >> 1) & 3u) << (2 * b);
                    if (c != 'A' && c != 'C' && c != 'G' && c != 'T') exc[t].emplace_back(i, c);
                }
                packed[q] = (uint8_t)v;
            }
        });
    for (auto &x : th) x.join();
    uint64_t total = 0;
    for (auto &e : exc) total += e.size();
    *n_exc = total;
    if (total > exc_cap) return groot::set_error(GROOT_E_NOSPACE, "%llu bytes are not ACGT, room for %llu", (unsigned long long)total, (unsigned long long)exc_cap);
    uint64_t o = 0;
    for (auto &e : exc)
        for (auto &pr : e) { exc_pos[o] = pr.first; exc_byte[o] = pr.second; o++; }
    return GROOT_OK;
}

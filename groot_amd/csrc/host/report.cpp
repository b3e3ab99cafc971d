// report.cpp -- `groot report`: per-reference breadth of coverage from the BAM written by `groot align`
//
// Restates src/reporting/reporting.go:33-173 (BAMreader.Run) and :178-213 (cigarClean); cmd/report.go:104-129 for the
// cutoff handling.  It sits after the hot path (the BAM is the hot path's output) and is here so that the reference's
// own end-to-end assertion -- testing/run_travis_tests.sh:36-56, "(Bla)B-7 is the only ARG reported" -- can be run
// against this build.  Free choices of the reference fixed here: annotations are printed in BAM header order (the
// reference ranges over a Go map), and the read count is the number of records on that reference (the reference reads
// a loop variable shared between goroutines, reporting.go:149).
#include <zlib.h>

#include <cstring>
#include <string>
#include <vector>

#include "host_common.hpp"

using namespace groot;

namespace {

// sequential reader of a BGZF stream (concatenated gzip members), file or stdin
struct BgzfIn {
    FILE *f = nullptr;
    bool own = false;
    std::vector<uint8_t> in, out;
    size_t out_pos = 0;
    bool eof = false;
    std::string err;

    bool open(const char *path)
    {
        if (!path) { f = stdin; return true; }
        f = fopen(path, "rb");
        own = true;
        return f != nullptr;
    }
    ~BgzfIn() { if (f && own) fclose(f); }

    bool next_block()
    {
        uint8_t hdr[18];
        const size_t got = fread(hdr, 1, 18, f);
        if (got == 0) { eof = true; return false; }
        if (got != 18 || hdr[0] != 0x1f || hdr[1] != 0x8b || hdr[2] != 8 || !(hdr[3] & 4)) { err = "not a BGZF block"; return false; }
        const unsigned xlen = hdr[10] | (hdr[11] << 8);
        // the BC subfield is the first (and in practice only) extra field of a BGZF writer
        if (xlen < 6 || hdr[12] != 'B' || hdr[13] != 'C') { err = "BGZF block without a BC field"; return false; }
        const unsigned bsize = (hdr[16] | (hdr[17] << 8)) + 1u;
        if (bsize < 18 + (xlen - 6) + 8) { err = "bad BGZF block size"; return false; }
        in.resize(bsize - 18);
        if (fread(in.data(), 1, in.size(), f) != in.size()) { err = "truncated BGZF block"; return false; }
        const size_t skip = xlen - 6;
        const size_t clen = in.size() - skip - 8;
        const uint8_t *tail = in.data() + in.size() - 8;
        const uint32_t isize = tail[4] | (tail[5] << 8) | (tail[6] << 16) | ((uint32_t)tail[7] << 24);
        out.resize(isize);
        out_pos = 0;
        if (isize == 0) return true;
        z_stream zs{};
        if (inflateInit2(&zs, -15) != Z_OK) { err = "zlib init failed"; return false; }
        zs.next_in = in.data() + skip; zs.avail_in = (uInt)clen;
        zs.next_out = out.data(); zs.avail_out = (uInt)isize;
        const int rc = inflate(&zs, Z_FINISH);
        inflateEnd(&zs);
        if (rc != Z_STREAM_END || zs.total_out != isize) { err = "corrupt BGZF block"; return false; }
        return true;
    }
    // false at a clean end of file before the first byte, or on error (err set)
    bool read(void *dst, size_t n, bool *clean_eof = nullptr)
    {
        uint8_t *d = (uint8_t *)dst;
        size_t done = 0;
        while (done < n) {
            if (out_pos == out.size()) {
                if (!next_block()) {
                    if (clean_eof) *clean_eof = eof && done == 0 && err.empty();
                    if (err.empty() && !(eof && done == 0)) err = "unexpected end of BAM";
                    return false;
                }
                continue;
            }
            const size_t take = std::min(n - done, out.size() - out_pos);
            memcpy(d + done, out.data() + out_pos, take);
            out_pos += take; done += take;
        }
        return true;
    }
};

uint32_t le32(const uint8_t *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }

// reporting.go:178-213.  Returns the run-length string and whether it holds "internal" uncovered stretches.
std::string cigar_clean(const std::vector<uint8_t> &covered, bool &internal_d)
{
    std::string cigar;
    internal_d = false;
    if (covered.empty()) return cigar;
    size_t counter = 1;
    uint8_t pre = covered[0];
    size_t runs[2] = {0, 0};   // [0] = "D", [1] = "M"
    const char sym[2] = {'D', 'M'};
    for (size_t i = 1; i < covered.size(); i++) {
        const uint8_t val = covered[i];
        if (i == covered.size() - 1) {
            if (val == pre) {
                counter++;
                cigar += std::to_string(counter) + sym[val];
                runs[val]++;
            } else {
                cigar += std::to_string(counter) + sym[pre] + "1" + sym[val];
                runs[val]++;                 // (the run before it is not counted: reporting.go:194-197)
            }
            break;
        }
        if (val == pre) counter++;
        else {
            runs[pre]++;
            cigar += std::to_string(counter) + sym[pre];
            pre = val;
            counter = 1;
        }
    }
    internal_d = !((runs[0] + runs[1] <= 2) || (runs[0] == 2 && runs[1] == 1));
    return cigar;
}

} // namespace

extern "C" int groot_host_report(const char *bam_path, double cov_cutoff, int low_cov, const char *out_path, uint64_t *n_reported)
{
    if (cov_cutoff > 1.0) return set_error(GROOT_E_INVALID, "supplied coverage cutoff exceeds 1.0 (100%%): %g", cov_cutoff);   // cmd/report.go:95-97
    if (low_cov) cov_cutoff = 0.97;                                                                                            // cmd/report.go:119-122
    BgzfIn in;
    if (!in.open(bam_path)) return set_error(GROOT_E_IO, "could not open BAM file %s", bam_path);
    uint8_t b4[4];
    if (!in.read(b4, 4) || memcmp(b4, "BAM\1", 4) != 0) return set_error(GROOT_E_FORMAT, "could not read BAM file: %s", in.err.empty() ? "bad magic" : in.err.c_str());
    if (!in.read(b4, 4)) return set_error(GROOT_E_FORMAT, "could not read BAM file: %s", in.err.c_str());
    std::vector<uint8_t> skip(le32(b4));
    if (!skip.empty() && !in.read(skip.data(), skip.size())) return set_error(GROOT_E_FORMAT, "could not read BAM file: %s", in.err.c_str());
    if (!in.read(b4, 4)) return set_error(GROOT_E_FORMAT, "could not read BAM file: %s", in.err.c_str());
    const uint32_t n_ref = le32(b4);
    std::vector<std::string> names(n_ref);
    std::vector<uint32_t> lens(n_ref);
    for (uint32_t r = 0; r < n_ref; r++) {
        if (!in.read(b4, 4)) return set_error(GROOT_E_FORMAT, "could not read BAM file: %s", in.err.c_str());
        const uint32_t l_name = le32(b4);
        std::vector<char> nm(l_name);
        if (l_name == 0 || !in.read(nm.data(), l_name) || !in.read(b4, 4)) return set_error(GROOT_E_FORMAT, "could not read BAM file: %s", in.err.c_str());
        names[r].assign(nm.data(), l_name - 1);
        lens[r] = le32(b4);
    }
    // pileup per reference (reporting.go:100-127): every record covers [Start, Start+Len] INCLUSIVE, clipped to the last base
    std::vector<std::vector<uint32_t>> pileup(n_ref);
    std::vector<uint64_t> count(n_ref, 0);
    std::vector<uint8_t> rec;
    for (;;) {
        bool clean = false;
        if (!in.read(b4, 4, &clean)) {
            if (clean) break;
            return set_error(GROOT_E_FORMAT, "error reading bam: %s", in.err.c_str());
        }
        const uint32_t bs = le32(b4);
        if (bs < 32) return set_error(GROOT_E_FORMAT, "error reading bam: record too short");
        rec.resize(bs);
        if (!in.read(rec.data(), bs)) return set_error(GROOT_E_FORMAT, "error reading bam: %s", in.err.c_str());
        const int32_t ref_id = (int32_t)le32(rec.data()), pos = (int32_t)le32(rec.data() + 4);
        const uint32_t l_read_name = rec[8];
        const uint32_t n_cigar = rec[12] | (rec[13] << 8), flag = rec[14] | (rec[15] << 8);
        if (flag == 4) continue;                                            // reporting.go:81-83
        if (ref_id < 0 || (uint32_t)ref_id >= n_ref || pos < 0) continue;   // no reference to add the record to
        if (32 + (uint64_t)l_read_name + 4ull * n_cigar > bs) return set_error(GROOT_E_FORMAT, "error reading bam: cigar past the record");
        uint64_t ref_len = 0;                                               // sam.Record.Len(): reference bases the CIGAR consumes
        for (uint32_t c = 0; c < n_cigar; c++) {
            const uint32_t op = le32(rec.data() + 32 + l_read_name + 4 * c);
            const uint32_t t = op & 15;
            if (t == 0 || t == 2 || t == 3 || t == 7 || t == 8) ref_len += op >> 4;   // M D N = X
        }
        auto &pl = pileup[ref_id];
        if (pl.empty()) pl.assign(lens[ref_id], 0);
        count[ref_id]++;
        if (pl.empty()) continue;
        uint64_t end = (uint64_t)pos + ref_len;
        if (end > pl.size() - 1) end = pl.size() - 1;
        for (uint64_t i = (uint64_t)pos; i <= end; i++) pl[i]++;
    }
    FILE *out = out_path ? fopen(out_path, "w") : stdout;
    if (!out) return set_error(GROOT_E_IO, "cannot create %s", out_path);
    uint64_t reported = 0;
    for (uint32_t r = 0; r < n_ref; r++) {
        if (!count[r] || pileup[r].empty()) continue;
        const auto &pl = pileup[r];
        size_t covered = 0;
        std::vector<uint8_t> cov(pl.size());
        for (size_t i = 0; i < pl.size(); i++) { cov[i] = pl[i] != 0; covered += cov[i]; }
        if ((double)covered / (double)pl.size() < cov_cutoff) continue;     // reporting.go:130-131
        bool internal_d = false;
        const std::string cigar = cigar_clean(cov, internal_d);
        if (internal_d && low_cov) continue;                                 // reporting.go:151-153
        const char *name = names[r].c_str();
        if (name[0] == '*') name++;                                         // cluster representative marker (:135-137)
        fprintf(out, "%s\t%llu\t%u\t%s\n", name, (unsigned long long)count[r], lens[r], cigar.c_str());
        reported++;
    }
    if (out_path) fclose(out); else fflush(out);
    if (n_reported) *n_reported = reported;
    return GROOT_OK;
}

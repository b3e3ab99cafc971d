// bam.cpp -- BAM output of the align path (BGZF-compressed, SAM/BAM v1.5).
//   src/pipeline/boss.go:45-105   setupBAM: @HD VN:1.5, one @SQ per path of every graph, @PG, @RG
//   src/graph/graphio.go:141-154  GetSAMrefs: sam.NewReference(pathName, "", "", Lengths[pathID], nil, nil)
//   src/graph/alignment.go:113-156 the record fields
//   src/pipeline/boss.go:225-240  records written in arrival order, then Close
// The reference writes through github.com/biogo/hts v1.1.0 (not in /root/reference); this writer follows the
// BAM specification for the same field values: next_refID=-1, next_pos=-1 (no mate), tlen=0, no aux tags,
// bin = reg2bin(pos, end).
#include "host_common.hpp"

#include <cstring>
#include <ctime>
#include <zlib.h>

using namespace groot;

struct groot_bam {
    FILE *f = nullptr;
    bool own = false;
    std::vector<uint8_t> block;   // uncompressed bytes waiting for the next BGZF block
    std::vector<uint8_t> out;
    uint32_t n_ref = 0;
};

static const size_t kBgzfBlock = 0xff00;   // max uncompressed payload per block

static int flush_block(groot_bam *b, const uint8_t *data, size_t n)
{
    b->out.resize(18 + compressBound((uLong)n) + 8);
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (deflateInit2(&zs, Z_DEFAULT_COMPRESSION, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return set_error(GROOT_E_NOMEM, "deflateInit2 failed");
    zs.next_in = const_cast<Bytef *>(data);
    zs.avail_in = (uInt)n;
    zs.next_out = b->out.data() + 18;
    zs.avail_out = (uInt)(b->out.size() - 18 - 8);
    const int rc = deflate(&zs, Z_FINISH);
    const size_t clen = zs.total_out;
    deflateEnd(&zs);
    if (rc != Z_STREAM_END) return set_error(GROOT_E_IO, "deflate failed");
    static const uint8_t hdr[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
    memcpy(b->out.data(), hdr, 16);
    const size_t total = 18 + clen + 8;
    if (total > 0x10000) return set_error(GROOT_E_IO, "BGZF block too large");
    const uint16_t bsize = (uint16_t)(total - 1);
    b->out[16] = (uint8_t)bsize; b->out[17] = (uint8_t)(bsize >> 8);
    const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), data, (uInt)n), isize = (uint32_t)n;
    memcpy(b->out.data() + 18 + clen, &crc, 4);
    memcpy(b->out.data() + 18 + clen + 4, &isize, 4);
    if (fwrite(b->out.data(), 1, total, b->f) != total) return set_error(GROOT_E_IO, "BAM write failed");
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

int groot_bam_write(groot_bam *b, const groot_aln_record *recs, uint64_t n)
{
    if (!b || (n && !recs)) return set_error(GROOT_E_INVALID, "null argument");
    static const char *code = "=ACMGRSVTWYHKDBN";
    uint8_t nt16[256];
    memset(nt16, 15, sizeof nt16);
    for (int i = 0; i < 16; i++) { nt16[(uint8_t)code[i]] = (uint8_t)i; nt16[(uint8_t)tolower(code[i])] = (uint8_t)i; }
    std::vector<uint8_t> buf;
    for (uint64_t i = 0; i < n; i++) {
        const groot_aln_record &r = recs[i];
        if (r.ref_id >= b->n_ref) return set_error(GROOT_E_INVALID, "record %llu: reference id out of range", (unsigned long long)i);
        if (r.name_len > 254) return set_error(GROOT_E_FORMAT, "read name longer than 254 characters");
        uint32_t cigar[3];
        uint32_t nc = 0;
        if (r.start_clip) cigar[nc++] = ((uint32_t)r.start_clip << 4) | 5;   // H (alignment.go:132-134)
        cigar[nc++] = (r.seq_len << 4) | 0;                                   // M (:135)
        if (r.end_clip) cigar[nc++] = ((uint32_t)r.end_clip << 4) | 5;       // H (:136-138)
        const uint16_t flag = (uint16_t)((r.secondary ? 0x100 : 0) | (r.reverse ? 0x10 : 0));   // :147-152
        const int bin = reg2bin(r.pos, (int64_t)r.pos + (r.seq_len ? r.seq_len : 1));
        const uint32_t l_name = r.name_len + 1;
        const uint32_t body = 32 + l_name + 4 * nc + (r.seq_len + 1) / 2 + r.seq_len;
        buf.resize(4 + body);
        uint8_t *p = buf.data();
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
        for (uint32_t j = 0; j < r.seq_len; j += 2) {
            const uint8_t hi = nt16[r.seq[j]], lo = j + 1 < r.seq_len ? nt16[r.seq[j + 1]] : 0;
            *p++ = (uint8_t)(hi << 4 | lo);
        }
        if (r.qual) memcpy(p, r.qual, r.seq_len);                             // raw bytes, not Phred-33 corrected (:121)
        else memset(p, 0xff, r.seq_len);
        p += r.seq_len;
        if (int rc = put(b, buf.data(), buf.size())) return rc;
    }
    return GROOT_OK;
}

int groot_bam_close(groot_bam *b)
{
    if (!b) return GROOT_OK;
    int rc = GROOT_OK;
    if (!b->block.empty()) rc = flush_block(b, b->block.data(), b->block.size());
    static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (!rc && fwrite(eof, 1, 28, b->f) != 28) rc = set_error(GROOT_E_IO, "BAM write failed");
    if (b->own) { if (fclose(b->f) != 0 && !rc) rc = set_error(GROOT_E_IO, "BAM close failed"); }
    else fflush(b->f);
    delete b;
    return rc;
}

} // extern "C"

// gz_inflate.hpp -- a gzip reader for the FASTQ ingest (reads.cpp) that inflates two to three times as fast as zlib's gzread.
//
// DataStreamer (src/pipeline/sketch.go:41-77) wraps a named *.gz file in compress/gzip and scans lines: ONE inflate stream per file, on one
// thread -- with the kernels at 2 000 Mreads/s and the parser at 10-20, that stream is what a gzip FASTQ waits for (bench.py mixed.cli_gzip).
// The format leaves no parallelism inside a member, so the gain has to come from the decoder itself:
//   * a 64-bit bit buffer refilled with one unaligned 8-byte load, never more than once per symbol pair;
//   * ONE table look-up per literal / length symbol (11 bits, sub-tables for the few longer codes), the entry holding the symbol's value, its
//     extra-bit count and its length; distance codes likewise (8 bits);
//   * literals two at a time when the second one is already in the bit buffer (FASTQ of real reads is literal-heavy: bases cost ~2.2 bits each);
//   * matches copied in 8-byte words (distance >= 8), as a byte fill (distance 1: constant quality strings), or byte by byte;
//   * the member's CRC-32 by carry-less multiplication over the finished output (bgzf_struct.hpp) -- checked, with ISIZE, at every member's end,
//     so a decoder fault cannot pass for data.
// Members follow each other (bgzip, cat of several .gz): each is checked and the next one begun; bytes that are not a gzip header after a complete
// member end the stream, as in zlib's gzread.  Everything is bounds-checked: a corrupt stream ends in an error, not in a write outside the window.
// The decoder is resumable at symbol boundaries (output space) and at any input byte (the file is read in pieces).
#pragma once

#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cerrno>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "bgzf_struct.hpp"

namespace groot {

class GzInflater {
public:
    explicit GzInflater(int fd) : fd_(fd)
    {
        in_.resize(kInBuf + kPad);
        out_.resize(kWindow + kOutChunk + kSlack);
        opos_ = odel_ = kWindow;
        build_fixed();
    }
    const std::string &error() const { return err_; }

    // up to `want` bytes of inflated data into dst; 0 at the end of the stream, -1 on error (error() says what)
    ssize_t read(uint8_t *dst, size_t want)
    {
        size_t got = 0;
        while (got < want) {
            if (odel_ < opos_) {
                const size_t n = std::min(want - got, opos_ - odel_);
                memcpy(dst + got, out_.data() + odel_, n);
                odel_ += n; got += n;
                continue;
            }
            if (st_ == ST_DONE) break;
            if (st_ == ST_ERROR) return -1;
            if (opos_ + kSlack + 258 > out_.size()) slide();
            if (!step()) { st_ = ST_ERROR; return -1; }
        }
        return (ssize_t)got;
    }

private:
    static constexpr size_t kInBuf = 1u << 20, kPad = 64, kWindow = 32768, kOutChunk = 1u << 20, kSlack = 32;
    static constexpr int kLitBits = 11, kDistBits = 10;
    // table entry: bits 0-3 bits to consume (a code's length; in a sub-table the bits BEYOND the primary ones; for literal runs the sum), bits 6-7 type.
    //   literal(s): bits 4-5 = count - 1, bits 8-15 / 16-23 / 24-31 = the bytes -- the literal/length table holds up to THREE literals per entry when
    //               their codes fit the 11 index bits together (the bases of a FASTQ cost ~2.2 bits each), written with one 4-byte store
    //   base (length / distance): bits 8-11 extra bits, bits 16-31 the base value;  sub-table: bits 8-11 its index bits, bits 16-31 its offset
    enum : uint32_t { T_LIT = 0, T_BASE = 1, T_END = 2, T_SUB = 3 };
    static inline uint32_t type_of(uint32_t e) { return (e >> 6) & 3; }
    static inline uint32_t lit_entry(uint32_t len, uint32_t byte) { return len | (T_LIT << 6) | (byte << 8); }
    static inline uint32_t entry(uint32_t len, uint32_t extra, uint32_t type, uint32_t value) { return len | (type << 6) | (extra << 8) | (value << 16); }
    enum State { ST_HEADER, ST_BLOCK, ST_STORED, ST_HUFF, ST_TRAILER, ST_DONE, ST_ERROR };

    int fd_;
    std::vector<uint8_t> in_, out_;
    size_t ipos_ = 0, iend_ = 0;
    bool ieof_ = false;
    uint64_t bb_ = 0;             // bit buffer (LSB first)
    int bc_ = 0;                  // valid bits in it
    size_t opos_, odel_;          // output produced / delivered (offsets into out_; [opos_ - 32768, opos_) is the window)
    size_t ovalid_ = kWindow;     // lowest offset of out_ that holds data of this member (a distance may not reach below it)
    size_t ocrc_ = kWindow;       // output of this member already folded into crc_
    State st_ = ST_HEADER;
    bool last_ = false, any_member_ = false;
    uint32_t stored_left_ = 0, crc_ = 0, isize_ = 0;
    std::string err_;
    uint32_t lit_[(1u << kLitBits) + 288 * 16], dist_[(1u << kDistBits) + 32 * 32];
    uint32_t fixed_lit_[(1u << kLitBits) + 512], fixed_dist_[(1u << kDistBits) + 32];
    const uint32_t *lt_ = nullptr, *dt_ = nullptr;

    bool fail(const char *m) { if (err_.empty()) err_ = m; return false; }

    // ---- input ----
    bool fill_input()            // more bytes behind iend_ (the unread ones move to the front); false: read error
    {
        if (ieof_) return true;
        if (ipos_ > 0) { memmove(in_.data(), in_.data() + ipos_, iend_ - ipos_); iend_ -= ipos_; ipos_ = 0; }
        while (iend_ < kInBuf) {
            const ssize_t n = ::read(fd_, in_.data() + iend_, kInBuf - iend_);
            if (n < 0) { if (errno == EINTR) continue; return fail("read error in gzip input"); }
            if (n == 0) { ieof_ = true; break; }
            iend_ += (size_t)n;
        }
        memset(in_.data() + iend_, 0, kPad);      // (the fast loop may load up to 8 bytes past the end: zeros, and never consumed -- see need())
        return true;
    }
    // make n (<= 56) bits available; false: the input ends first
    inline bool need(int n)
    {
        while (bc_ < n) {
            if (ipos_ >= iend_) {
                if (ieof_) return false;
                if (!fill_input()) return false;
                if (ipos_ >= iend_) return false;
            }
            bb_ |= (uint64_t)in_[ipos_++] << bc_;
            bc_ += 8;
        }
        return true;
    }
    inline uint32_t take(int n) { const uint32_t v = (uint32_t)(bb_ & ((1ull << n) - 1)); bb_ >>= n; bc_ -= n; return v; }
    bool byte(uint8_t &b) { if (!need(8)) return false; b = (uint8_t)take(8); return true; }

    // ---- output ----
    void fold_crc()
    {
        if (opos_ > ocrc_) {
            crc_ = crc_update(crc_, out_.data() + ocrc_, opos_ - ocrc_);
            isize_ += (uint32_t)(opos_ - ocrc_);
            ocrc_ = opos_;
        }
    }
    static uint32_t crc_update(uint32_t c, const uint8_t *p, size_t n)
    {
#if defined(__x86_64__)
        // (the folding routine is checked against zlib once per process, as fast_crc32 does)
        static const bool usable = []() {
            if (!__builtin_cpu_supports("pclmul") || !__builtin_cpu_supports("sse4.1")) return false;
            uint8_t t[304];
            for (size_t i = 0; i < sizeof t; i++) t[i] = (uint8_t)(i * 131 + 7);
            const uint32_t seed = 0x12345678u;
            for (size_t len : {(size_t)64, (size_t)80, (size_t)256, (size_t)304})
                if (~crc32_clmul_(t, len, ~seed) != (uint32_t)crc32(seed, t, (uInt)len)) return false;
            return true;
        }();
        if (usable && n >= 64) {
            const size_t body = n & ~(size_t)15;
            c = ~crc32_clmul_(p, body, ~c);
            p += body; n -= body;
        }
#endif
        while (n) { const uInt k = (uInt)std::min<size_t>(n, 1u << 30); c = (uint32_t)crc32(c, p, k); p += k; n -= k; }
        return c;
    }
    void slide()                 // everything produced has been delivered: keep the last 32 KB as the window
    {
        fold_crc();
        const size_t keep = std::min<size_t>(kWindow, opos_ - ovalid_);
        memmove(out_.data() + kWindow - keep, out_.data() + opos_ - keep, keep);
        ovalid_ = kWindow - keep;
        opos_ = odel_ = ocrc_ = kWindow;
    }

    // ---- Huffman tables ----
    // canonical code of `n` symbols with lengths len[] (0 = unused) -> table of `bits` primary bits + sub-tables; false: over-subscribed / incomplete
    // (one code of length 1 is allowed for the distance alphabet, as zlib allows it)
    static bool build(const uint8_t *len, int n, int bits, uint32_t *tab, size_t cap, bool is_dist, bool runs = false)
    {
        static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
        static const uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
        static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
        static const uint8_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
        int count[16] = {0};
        for (int s = 0; s < n; s++) count[len[s]]++;
        count[0] = 0;
        int left = 1, used = 0;
        for (int l = 1; l <= 15; l++) { left = left * 2 - count[l]; used += count[l]; if (left < 0) return false; }
        if (left > 0 && !(used <= 1)) return false;          // incomplete: only the empty code and a single code are allowed
        uint32_t next[16], code = 0;
        for (int l = 1; l <= 15; l++) { code = (code + (uint32_t)count[l - 1]) << 1; next[l] = code; }
        const uint32_t invalid = entry(1, 0, T_END, 0xFFFF);  // (value 0xFFFF: "not a code" -- decoded as an error)
        for (uint32_t i = 0; i < (1u << bits); i++) tab[i] = invalid;
        auto sym_entry = [&](int s, uint32_t l) -> uint32_t {
            if (is_dist) return s < 30 ? entry(l, dext[s], T_BASE, dbase[s]) : invalid;
            if (s < 256) return lit_entry(l, (uint32_t)s);
            if (s == 256) return entry(l, 0, T_END, 0);
            return s < 286 ? entry(l, lext[s - 257], T_BASE, lbase[s - 257]) : invalid;
        };
        auto rev = [](uint32_t v, int l) { uint32_t r = 0; for (int i = 0; i < l; i++) r |= ((v >> i) & 1u) << (l - 1 - i); return r; };
        // longest code behind every primary prefix that has sub-table codes
        std::vector<uint8_t> sub_bits((size_t)1 << bits, 0);
        {
            uint32_t nx[16];
            memcpy(nx, next, sizeof nx);
            for (int s = 0; s < n; s++) {
                const int l = len[s];
                if (l <= bits) { if (l) nx[l]++; continue; }
                const uint32_t r = rev(nx[l]++, l), pre = r & ((1u << bits) - 1);
                sub_bits[pre] = std::max<uint8_t>(sub_bits[pre], (uint8_t)(l - bits));
            }
        }
        size_t top = (size_t)1 << bits;
        for (uint32_t pre = 0; pre < (1u << bits); pre++)
            if (sub_bits[pre]) {
                if (top + ((size_t)1 << sub_bits[pre]) > cap) return false;
                tab[pre] = entry(0, sub_bits[pre], T_SUB, (uint32_t)top);
                for (size_t i = 0; i < ((size_t)1 << sub_bits[pre]); i++) tab[top + i] = invalid;
                top += (size_t)1 << sub_bits[pre];
                if (top > 0xFFFF) return false;
            }
        for (int s = 0; s < n; s++) {
            const int l = len[s];
            if (!l) continue;
            const uint32_t r = rev(next[l]++, l);
            if (l <= bits) {
                const uint32_t e = sym_entry(s, (uint32_t)l);
                for (uint32_t i = r; i < (1u << bits); i += 1u << l) tab[i] = e;
            } else {
                const uint32_t pre = r & ((1u << bits) - 1), sb = sub_bits[pre], off = tab[pre] >> 16;
                const uint32_t e = sym_entry(s, (uint32_t)(l - bits));
                for (uint32_t i = r >> bits; i < (1u << sb); i += 1u << (l - bits)) tab[off + i] = e;
            }
        }
        if (runs) {
            // literal runs: an entry whose literal leaves index bits over takes the literal(s) those bits decode to as well
            std::vector<uint32_t> one(tab, tab + ((size_t)1 << bits));
            const uint32_t mask = (1u << bits) - 1;
            for (uint32_t i = 0; i <= mask; i++) {
                uint32_t e = one[i];
                if (type_of(e) != T_LIT) continue;
                uint32_t used = e & 15, n = 1, bytes = (e >> 8) & 0xFF;
                while (n < 3) {
                    const uint32_t e2 = one[(i >> used) & mask];      // (the unknown upper bits read as zeros: fine while the code fits the known ones)
                    if (type_of(e2) != T_LIT || used + (e2 & 15) > (uint32_t)bits) break;
                    bytes |= ((e2 >> 8) & 0xFF) << (8 * n);
                    used += e2 & 15;
                    n++;
                }
                tab[i] = used | ((n - 1) << 4) | (T_LIT << 6) | (bytes << 8);
            }
        }
        return true;
    }
    void build_fixed()
    {
        uint8_t l[288];
        for (int s = 0; s < 288; s++) l[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
        build(l, 288, kLitBits, fixed_lit_, sizeof fixed_lit_ / 4, false, true);
        uint8_t d[32];
        for (int s = 0; s < 32; s++) d[s] = 5;
        build(d, 32, kDistBits, fixed_dist_, sizeof fixed_dist_ / 4, true);
    }
    bool read_dynamic()
    {
        if (!need(14)) return fail("gzip input ends inside a block header");
        const int hlit = (int)take(5) + 257, hdist = (int)take(5) + 1, hclen = (int)take(4) + 4;
        if (hlit > 286 || hdist > 30) return fail("gzip input: bad block header");
        static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        uint8_t cl[19] = {0};
        for (int i = 0; i < hclen; i++) { if (!need(3)) return fail("gzip input ends inside a block header"); cl[order[i]] = (uint8_t)take(3); }
        uint32_t ct[(1u << 7) + 8];
        if (!build(cl, 19, 7, ct, sizeof ct / 4, false)) return fail("gzip input: bad code-length code");
        uint8_t lens[320] = {0};
        int i = 0;
        while (i < hlit + hdist) {
            if (!need(7 + 7)) { if (bc_ < 1) return fail("gzip input ends inside a block header"); }
            const uint32_t e = ct[bb_ & 127];
            const int l = (int)(e & 15);
            if (type_of(e) != T_LIT || l > bc_) return fail("gzip input: bad code lengths");
            take(l);
            const uint32_t s = (e >> 8) & 0xFF;
            if (s < 16) { lens[i++] = (uint8_t)s; continue; }
            int rep, xb;
            uint8_t v = 0;
            if (s == 16) { if (!i) return fail("gzip input: bad code lengths"); v = lens[i - 1]; xb = 2; rep = 3; }
            else if (s == 17) { xb = 3; rep = 3; }
            else { xb = 7; rep = 11; }
            if (bc_ < xb && !need(xb)) return fail("gzip input ends inside a block header");
            rep += (int)take(xb);
            if (i + rep > hlit + hdist) return fail("gzip input: bad code lengths");
            while (rep--) lens[i++] = v;
        }
        if (!lens[256]) return fail("gzip input: block without an end code");
        if (!build(lens, hlit, kLitBits, lit_, sizeof lit_ / 4, false, true)) return fail("gzip input: bad literal/length code");
        if (!build(lens + hlit, hdist, kDistBits, dist_, sizeof dist_ / 4, true)) return fail("gzip input: bad distance code");
        lt_ = lit_; dt_ = dist_;
        return true;
    }

    // ---- the stream ----
    bool step()
    {
        switch (st_) {
        case ST_HEADER: return header();
        case ST_BLOCK: {
            if (last_) { st_ = ST_TRAILER; return true; }
            if (!need(3)) return fail("gzip input ends before the final block");
            last_ = take(1) != 0;
            const uint32_t type = take(2);
            if (type == 0) {
                take(bc_ & 7);                                   // to the byte boundary
                if (!need(32)) return fail("gzip input ends inside a stored block");
                const uint32_t l = take(16), nl = take(16);
                if ((l ^ nl) != 0xFFFFu) return fail("gzip input: bad stored block");
                stored_left_ = l;
                st_ = ST_STORED;
                return true;
            }
            if (type == 1) { lt_ = fixed_lit_; dt_ = fixed_dist_; }
            else if (type == 2) { if (!read_dynamic()) return false; }
            else return fail("gzip input: bad block type");
            st_ = ST_HUFF;
            return true;
        }
        case ST_STORED: {
            while (stored_left_ && opos_ + kSlack < out_.size()) {
                if (bc_ >= 8) { out_[opos_++] = (uint8_t)take(8); stored_left_--; continue; }
                if (ipos_ >= iend_) {
                    if (ieof_) return fail("gzip input ends inside a stored block");
                    if (!fill_input()) return false;
                    if (ipos_ >= iend_) return fail("gzip input ends inside a stored block");
                }
                bb_ = 0;                                         // (bits loaded ahead of bc_ belong to bytes that are now copied past the bit buffer)
                const size_t n = std::min(std::min<size_t>(stored_left_, iend_ - ipos_), out_.size() - kSlack - opos_);
                memcpy(out_.data() + opos_, in_.data() + ipos_, n);
                opos_ += n; ipos_ += n; stored_left_ -= (uint32_t)n;
            }
            if (!stored_left_) st_ = ST_BLOCK;
            return true;
        }
        case ST_HUFF: return huff();
        case ST_TRAILER: {
            take(bc_ & 7);
            fold_crc();
            uint32_t c = 0, n = 0;
            for (int i = 0; i < 4; i++) { uint8_t b; if (!byte(b)) return fail("gzip input ends inside a member's trailer"); c |= (uint32_t)b << (8 * i); }
            for (int i = 0; i < 4; i++) { uint8_t b; if (!byte(b)) return fail("gzip input ends inside a member's trailer"); n |= (uint32_t)b << (8 * i); }
            if (c != crc_) return fail("gzip input: CRC mismatch");
            if (n != isize_) return fail("gzip input: length mismatch");
            st_ = ST_HEADER;
            return true;
        }
        default: return true;
        }
    }
    bool header()
    {
        // (byte-aligned here: at the start of the file, or behind a trailer)
        // Behind a member, the input may end, or go on with bytes that are not a gzip header (ignored, as gzread does).  Once the magic has matched, a
        // member HAS begun: input that ends inside its header is a truncated file (a bgzip FASTQ cut inside a block header), and an error -- zlib's
        // gzread reports "unexpected end of file" there and Go's gzip.Reader (the reference: sketch.go:175-238) io.ErrUnexpectedEOF.
        uint8_t h[10];
        for (int i = 0; i < 2; i++)
            if (!byte(h[i]) || h[i] != (i ? 0x8b : 0x1f)) {
                if (any_member_) { st_ = ST_DONE; return true; }
                return fail("not a gzip stream");
            }
        for (int i = 2; i < 10; i++)
            if (!byte(h[i])) return fail("gzip input ends inside a header");
        if (h[2] != 8 || (h[3] & 0xE0)) return fail("gzip input: unsupported header");
        uint8_t b;
        if (h[3] & 4) {                                        // FEXTRA
            uint8_t l0, l1;
            if (!byte(l0) || !byte(l1)) return fail("gzip input ends inside a header");
            for (uint32_t n = l0 | ((uint32_t)l1 << 8); n; n--) if (!byte(b)) return fail("gzip input ends inside a header");
        }
        if (h[3] & 8) do { if (!byte(b)) return fail("gzip input ends inside a header"); } while (b);    // FNAME
        if (h[3] & 16) do { if (!byte(b)) return fail("gzip input ends inside a header"); } while (b);   // FCOMMENT
        if (h[3] & 2) { if (!byte(b) || !byte(b)) return fail("gzip input ends inside a header"); }      // FHCRC
        any_member_ = true;
        last_ = false;
        fold_crc();                                            // (nothing of the new member yet: brings ocrc_ up to opos_)
        crc_ = 0; isize_ = 0;
        ovalid_ = opos_;                                       // distances do not reach into the member before
        ocrc_ = opos_;
        st_ = ST_BLOCK;
        return true;
    }
    inline void refill_fast()
    {
        uint64_t v;
        memcpy(&v, in_.data() + ipos_, 8);
        bb_ |= v << bc_;
        ipos_ += (size_t)((63 - bc_) >> 3);
        bc_ |= 56;
    }
    bool huff()
    {
        const uint32_t *lt = lt_, *dt = dt_;
        uint8_t *const out = out_.data();
        const size_t olimit = out_.size() - kSlack - 258;
        // (the loop's state in locals: the byte stores into the window may alias any member, and would have every one of them reloaded per symbol)
        uint64_t bb = bb_;
        int bc = bc_;
        size_t op = opos_;
        const uint8_t *ip = in_.data() + ipos_, *iend = in_.data() + iend_;
        const size_t ovalid = ovalid_;
        auto save = [&]() { bb_ = bb; bc_ = bc; opos_ = op; ipos_ = (size_t)(ip - in_.data()); };
        for (;;) {
            if (op > olimit) { save(); return true; }         // output space: back to read(), which delivers and slides
            if (iend - ip < 16) {
                save();
                if (!ieof_) { if (!fill_input()) return false; }
                if (iend_ - ipos_ < 16) return huff_tail();     // the last bytes of the input: bit by bit, with every check
                ip = in_.data() + ipos_; iend = in_.data() + iend_;
            }
            {                                                  // refill: >= 56 bits
                uint64_t v;
                memcpy(&v, ip, 8);
                bb |= v << bc;
                ip += (63 - bc) >> 3;
                bc |= 56;
            }
            uint32_t e = lt[bb & ((1u << kLitBits) - 1)];
            if (__builtin_expect(type_of(e) == T_SUB, 0)) { bb >>= kLitBits; bc -= kLitBits; e = lt[(e >> 16) + (bb & ((1u << ((e >> 8) & 15)) - 1))]; }
            bb >>= (e & 15); bc -= (int)(e & 15);
            const uint32_t type = type_of(e);
            if (type == T_LIT) {
                // one to three literals per entry, one 4-byte store; then up to three more entries from the bits already here (an entry of the
                // primary table costs at most 11 bits, and 56 - 15 were left)
                uint32_t w = e >> 8;
                memcpy(out + op, &w, 4);
                op += ((e >> 4) & 3) + 1;
                e = lt[bb & ((1u << kLitBits) - 1)];
                if (type_of(e) == T_LIT) {
                    bb >>= (e & 15); bc -= (int)(e & 15);
                    w = e >> 8; memcpy(out + op, &w, 4); op += ((e >> 4) & 3) + 1;
                    e = lt[bb & ((1u << kLitBits) - 1)];
                    if (type_of(e) == T_LIT) {
                        bb >>= (e & 15); bc -= (int)(e & 15);
                        w = e >> 8; memcpy(out + op, &w, 4); op += ((e >> 4) & 3) + 1;
                        e = lt[bb & ((1u << kLitBits) - 1)];
                        if (type_of(e) == T_LIT) {
                            bb >>= (e & 15); bc -= (int)(e & 15);
                            w = e >> 8; memcpy(out + op, &w, 4); op += ((e >> 4) & 3) + 1;
                        }
                    }
                }
                continue;
            }
            if (type == T_END) {
                save();
                if ((e >> 16) == 0xFFFF) return fail("gzip input: invalid literal/length code");
                st_ = ST_BLOCK;
                return true;
            }
            // a match: base length + extra bits, then the distance (at least 56 - 15 - 5 = 36 bits are left: 15 + 13 needed)
            const uint32_t len = (e >> 16) + (uint32_t)(bb & ((1u << ((e >> 8) & 15)) - 1));
            bb >>= ((e >> 8) & 15); bc -= (int)((e >> 8) & 15);
            uint32_t d = dt[bb & ((1u << kDistBits) - 1)];
            if (__builtin_expect(type_of(d) == T_SUB, 0)) { bb >>= kDistBits; bc -= kDistBits; d = dt[(d >> 16) + (bb & ((1u << ((d >> 8) & 15)) - 1))]; }
            if (__builtin_expect(type_of(d) != T_BASE, 0)) { save(); return fail("gzip input: invalid distance code"); }
            bb >>= (d & 15); bc -= (int)(d & 15);
            const uint32_t dist = (d >> 16) + (uint32_t)(bb & ((1u << ((d >> 8) & 15)) - 1));
            bb >>= ((d >> 8) & 15); bc -= (int)((d >> 8) & 15);
            if (__builtin_expect(dist > op - ovalid, 0)) { save(); return fail("gzip input: distance reaches before the start of the data"); }
            {
                uint8_t *dst = out + op;
                const uint8_t *src = dst - dist;
                op += len;
                if (dist >= 8) {                               // word by word (past the match's end: slack, overwritten by what follows)
                    uint64_t v0, v1;                           // (sixteen bytes without a question: matches of DNA are ~14 bytes long)
                    memcpy(&v0, src, 8); memcpy(dst, &v0, 8);
                    memcpy(&v1, src + 8, 8); memcpy(dst + 8, &v1, 8);
                    if (__builtin_expect(len > 16, 0)) {
                        uint8_t *end = dst + len;
                        src += 16; dst += 16;
                        do { uint64_t v; memcpy(&v, src, 8); memcpy(dst, &v, 8); src += 8; dst += 8; } while (dst < end);
                    }
                } else if (dist == 1) {
                    memset(dst, *src, len);
                } else {
                    for (uint32_t i = 0; i < len; i++) dst[i] = src[i];
                }
            }
        }
    }
    inline void copy_match(uint8_t *out, uint32_t len, uint32_t dist)
    {
        uint8_t *dst = out + opos_;
        const uint8_t *src = dst - dist;
        opos_ += len;
        if (dist >= 8) {                                       // word by word (up to 7 bytes past the match: slack, overwritten by what follows)
            uint8_t *end = dst + len;
            do { uint64_t v; memcpy(&v, src, 8); memcpy(dst, &v, 8); src += 8; dst += 8; } while (dst < end);
        } else if (dist == 1) {
            memset(dst, *src, len);
        } else {
            for (uint32_t i = 0; i < len; i++) dst[i] = src[i];
        }
    }
    // the same loop for the last few bytes of the input: no load past the end, every bit accounted for
    bool huff_tail()
    {
        const uint32_t *lt = lt_, *dt = dt_;
        uint8_t *out = out_.data();
        const size_t olimit = out_.size() - kSlack - 258;
        for (;;) {
            if (opos_ > olimit) return true;
            if (iend_ - ipos_ >= 16 && !ieof_) return true;    // (never: the caller only comes here at the end of the input)
            need(48);                                          // as many bits as there are (at most 56 wanted below)
            uint32_t e = lt[bb_ & ((1u << kLitBits) - 1)];
            int used = 0;
            if (type_of(e) == T_SUB) { used = kLitBits; e = lt[(e >> 16) + ((bb_ >> kLitBits) & ((1u << ((e >> 8) & 15)) - 1))]; }
            const uint32_t type = type_of(e);
            // (a run of literals that reaches past the bits that are left: in a whole stream the end-of-block code follows the last literal, and
            // an entry describes literals only -- so the input was cut inside the block, as when a single code does not fit: the check below)
            used += (int)(e & 15);
            if (type == T_BASE) used += (int)((e >> 8) & 15);
            if (used > bc_) return fail("gzip input ends inside a block");
            if (type == T_LIT) {
                take(used);
                const uint32_t w = e >> 8;
                memcpy(out + opos_, &w, 4);
                opos_ += ((e >> 4) & 3) + 1;
                continue;
            }
            if (type == T_END) {
                if ((e >> 16) == 0xFFFF) return fail("gzip input: invalid literal/length code");
                take(used);
                st_ = ST_BLOCK;
                return true;
            }
            const uint32_t xl = (e >> 8) & 15;
            const uint32_t len = (e >> 16) + (uint32_t)((bb_ >> (used - (int)xl)) & ((1u << xl) - 1));
            take(used);
            need(28);
            uint32_t d = dt[bb_ & ((1u << kDistBits) - 1)];
            int du = 0;
            if (type_of(d) == T_SUB) { du = kDistBits; d = dt[(d >> 16) + ((bb_ >> kDistBits) & ((1u << ((d >> 8) & 15)) - 1))]; }
            if (type_of(d) != T_BASE) return fail("gzip input: invalid distance code");
            du += (int)(d & 15);
            const uint32_t xd = (d >> 8) & 15;
            if (du + (int)xd > bc_) return fail("gzip input ends inside a block");
            const uint32_t dist = (d >> 16) + (uint32_t)((bb_ >> du) & ((1u << xd) - 1));
            take(du + (int)xd);
            if (dist > opos_ - ovalid_) return fail("gzip input: distance reaches before the start of the data");
            copy_match(out, len, dist);
        }
    }
};

} // namespace groot

// bgzf_struct.hpp -- BGZF blocks for BAM records without a match search.
//
// `groot align` writes one sam.Record per path of a traversal (alignment.go:113-156): 17 records per read on arg-annot.90 that
// differ in refID, pos, bin and the Secondary flag -- 8 to 12 of ~230 bytes.  A general deflate spends its time looking for
// what the writer already knows.  Here every record whose predecessor in the block has the same size is written as deflate
// back-references to that predecessor (distance = record size) with the differing header bytes as literals; other records go
// out as literals plus distance-1 runs (constant quality strings).  Fixed Huffman codes, one deflate block per BGZF member, the
// member's CRC-32 by carry-less multiplication where the CPU has it.  The inflated stream is byte for byte what zlib's path
// produces (tests/test_host.py inflates both and compares); only the compressed form differs (about 4x larger than zlib -1).
// Replaces: bgzf.Writer of biogo/hts behind bam.NewWriter (src/pipeline/boss.go:86-104).
#pragma once

#include <zlib.h>

#include <cstdint>
#include <cstring>
#include <vector>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace groot {

// ---- CRC-32 (zlib's polynomial) ------------------------------------------------------------------------------------
#if defined(__x86_64__)
// folding by carry-less multiplication (Gopal et al., "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ"; the
// constants are the bit-reflected ones of the IEEE 802.3 polynomial).  len >= 64 and a multiple of 16; crc = running value
// with the final inversion undone (~crc32).
__attribute__((target("pclmul,sse4.1"))) static inline uint32_t crc32_clmul_(const uint8_t *buf, size_t len, uint32_t crc)
{
    alignas(16) static const uint64_t k1k2[2] = {0x0154442bd4ULL, 0x01c6e41596ULL};
    alignas(16) static const uint64_t k3k4[2] = {0x01751997d0ULL, 0x00ccaa009eULL};
    alignas(16) static const uint64_t k5k0[2] = {0x0163cd6124ULL, 0x0000000000ULL};
    alignas(16) static const uint64_t poly[2] = {0x01db710641ULL, 0x01f7011641ULL};
    __m128i x0, x1, x2, x3, x4, x5, x6, x7, x8, y5, y6, y7, y8;
    x1 = _mm_loadu_si128((const __m128i *)(buf + 0x00));
    x2 = _mm_loadu_si128((const __m128i *)(buf + 0x10));
    x3 = _mm_loadu_si128((const __m128i *)(buf + 0x20));
    x4 = _mm_loadu_si128((const __m128i *)(buf + 0x30));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
    x0 = _mm_load_si128((const __m128i *)k1k2);
    buf += 64; len -= 64;
    while (len >= 64) {
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x6 = _mm_clmulepi64_si128(x2, x0, 0x00);
        x7 = _mm_clmulepi64_si128(x3, x0, 0x00); x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x2 = _mm_clmulepi64_si128(x2, x0, 0x11);
        x3 = _mm_clmulepi64_si128(x3, x0, 0x11); x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
        y5 = _mm_loadu_si128((const __m128i *)(buf + 0x00)); y6 = _mm_loadu_si128((const __m128i *)(buf + 0x10));
        y7 = _mm_loadu_si128((const __m128i *)(buf + 0x20)); y8 = _mm_loadu_si128((const __m128i *)(buf + 0x30));
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), y5); x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), y6);
        x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), y7); x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), y8);
        buf += 64; len -= 64;
    }
    x0 = _mm_load_si128((const __m128i *)k3k4);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
    while (len >= 16) {
        x2 = _mm_loadu_si128((const __m128i *)buf);
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
        buf += 16; len -= 16;
    }
    x2 = _mm_clmulepi64_si128(x1, x0, 0x10);
    x3 = _mm_setr_epi32(~0, 0, ~0, 0);
    x1 = _mm_srli_si128(x1, 8);
    x1 = _mm_xor_si128(x1, x2);
    x0 = _mm_loadl_epi64((const __m128i *)k5k0);
    x2 = _mm_srli_si128(x1, 4);
    x1 = _mm_and_si128(x1, x3);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    x0 = _mm_load_si128((const __m128i *)poly);
    x2 = _mm_and_si128(x1, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x10);
    x2 = _mm_and_si128(x2, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    return (uint32_t)_mm_extract_epi32(x1, 1);
}
#endif

// crc32() of zlib, faster where the CPU can; the folding routine is checked against zlib once per process and left alone if
// the two ever disagree
static inline uint32_t fast_crc32(const uint8_t *p, size_t n)
{
#if defined(__x86_64__)
    static const bool usable = []() {
        if (!__builtin_cpu_supports("pclmul") || !__builtin_cpu_supports("sse4.1")) return false;
        uint8_t t[256 + 48];
        for (size_t i = 0; i < sizeof t; i++) t[i] = (uint8_t)(i * 131 + 7);
        for (size_t len : {(size_t)64, (size_t)80, (size_t)256, (size_t)304})
            if (~crc32_clmul_(t, len, ~0u) != (uint32_t)crc32(crc32(0L, Z_NULL, 0), t, (uInt)len)) return false;
        return true;
    }();
    if (usable && n >= 64) {
        const size_t body = n & ~(size_t)15;
        uint32_t c = ~crc32_clmul_(p, body, ~0u);
        if (body < n) c = (uint32_t)crc32(c, p + body, (uInt)(n - body));
        return c;
    }
#endif
    return (uint32_t)crc32(crc32(0L, Z_NULL, 0), p, (uInt)n);
}

// ---- fixed-Huffman deflate writer ------------------------------------------------------------------------------------
struct DeflateTables {
    uint16_t lit_code[288];
    uint8_t lit_bits[288];
    uint16_t len_sym[259];      // match length 3..258 -> literal/length symbol
    uint8_t len_xbits[259];
    uint16_t len_xval[259];
    uint8_t dist_code[30];      // 5-bit codes, bit-reversed
    DeflateTables()
    {
        auto rev = [](uint32_t v, int bits) { uint32_t r = 0; for (int i = 0; i < bits; i++) r |= ((v >> i) & 1u) << (bits - 1 - i); return r; };
        for (int s = 0; s < 288; s++) {
            uint32_t code; int bits;
            if (s <= 143) { code = 0x30 + s; bits = 8; }
            else if (s <= 255) { code = 0x190 + (s - 144); bits = 9; }
            else if (s <= 279) { code = s - 256; bits = 7; }
            else { code = 0xC0 + (s - 280); bits = 8; }
            lit_code[s] = (uint16_t)rev(code, bits); lit_bits[s] = (uint8_t)bits;
        }
        static const uint16_t base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
        static const uint8_t xb[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
        for (int len = 3; len <= 258; len++) {
            int s = 28;
            while (base[s] > len) s--;
            if (len == 258) s = 28;
            len_sym[len] = (uint16_t)(257 + s); len_xbits[len] = xb[s]; len_xval[len] = (uint16_t)(len - base[s]);
        }
        for (int d = 0; d < 30; d++) dist_code[d] = (uint8_t)rev((uint32_t)d, 5);
    }
};

struct BitWriter {
    uint8_t *p;
    uint64_t acc = 0;
    int n = 0;
    explicit BitWriter(uint8_t *out) : p(out) {}
    inline void put(uint32_t v, int bits)
    {
        acc |= (uint64_t)v << n;
        n += bits;
        if (n >= 32) { memcpy(p, &acc, 4); p += 4; acc >>= 32; n -= 32; }
    }
    inline uint8_t *finish()
    {
        while (n > 0) { *p++ = (uint8_t)acc; acc >>= 8; n -= 8; }
        n = 0; acc = 0;
        return p;
    }
    // a stored block's payload starts at a byte boundary: pad with zero bits, then the bytes as they are
    inline void bytes(const uint8_t *src, size_t len)
    {
        finish();
        memcpy(p, src, len);
        p += len;
    }
};

// data[0..n) with record starts rec[0..n_rec) (ascending, rec[0] == 0, the last record ends at n) -> one BGZF member appended to
// out.  false: the block cannot be written this way (a record of 32 KB or more, output larger than a member may be).
// follow (optional, one byte per record): the writer's own knowledge of the records.  follow[r] = 1: record r is a copy of record
// r - 1 in which only bytes [4, 20) may differ (refID, pos, bin, flag: the next path of the same traversal, bam.cpp) -- no comparison
// of the ~200 bytes behind them; follow[r] = 0: a record of its own (the first of a traversal: another read's name, bases and
// qualities) -- it goes out as a STORED deflate block, a memcpy instead of ~230 Huffman-coded literals (which fixed codes would
// not make any smaller).  The member is then a sequence of stored and fixed-Huffman blocks closed by an empty final block.
static inline bool bgzf_member_structural(const uint8_t *data, size_t n, const uint32_t *rec, size_t n_rec, std::vector<uint8_t> &out,
                                          const uint8_t *follow = nullptr)
{
    static const DeflateTables T;
    if (n == 0 || n > 0xff00) return false;
    const size_t at = out.size();
    out.resize(at + 18 + n + n / 8 + 8 * n_rec + 64 + 8);          // literals cost at most 9 bits; a stored block 5-6 bytes
    BitWriter bw(out.data() + at + 18);
    bool in_huff = false;                                          // a fixed-Huffman block is open
    if (!follow) { bw.put(3, 3); in_huff = true; }                 // BFINAL = 1, BTYPE = 01 (fixed Huffman): the whole member is one block
    auto lit = [&](uint8_t b) { bw.put(T.lit_code[b], T.lit_bits[b]); };
    auto match = [&](uint32_t len, uint32_t dist) {                // 3 <= len <= 258, 1 <= dist <= 32768
        bw.put(T.lit_code[T.len_sym[len]], T.lit_bits[T.len_sym[len]]);
        if (T.len_xbits[len]) bw.put(T.len_xval[len], T.len_xbits[len]);
        uint32_t d = dist - 1, code, xbits;
        if (d < 4) { code = d; xbits = 0; }
        else { const uint32_t hb = 31u - (uint32_t)__builtin_clz(d); xbits = hb - 1; code = 2 * hb + ((d >> (hb - 1)) & 1u); }
        bw.put(T.dist_code[code], 5);
        if (xbits) bw.put(d & ((1u << xbits) - 1u), (int)xbits);
    };
    auto copy = [&](uint32_t len, uint32_t dist) {                 // any length >= 3
        while (len) {
            uint32_t m = len > 258 ? 258 : len;
            if (len - m > 0 && len - m < 3) m = len - 3;           // never leave a tail shorter than a match
            match(m, dist);
            len -= m;
        }
    };
    // literals, with runs of one byte as distance-1 matches (quality strings, padding)
    auto plain = [&](const uint8_t *s, size_t len) {
        size_t i = 0;
        while (i < len) {
            size_t j = i + 1;
            while (j < len && s[j] == s[i]) j++;
            const size_t run = j - i;
            lit(s[i]);
            if (run >= 4) copy((uint32_t)(run - 1), 1);
            else for (size_t x = 1; x < run; x++) lit(s[i]);
            i = j;
        }
    };
    for (size_t r = 0; r < n_rec; r++) {
        const size_t a = rec[r], b = r + 1 < n_rec ? rec[r + 1] : n;
        const size_t len = b - a;
        const uint8_t *cur = data + a;
        const bool trusted = follow && follow[r] && r > 0 && len == a - rec[r - 1] && len < 32768 && len >= 36;
        if (follow && !trusted) {                                  // stored block (BTYPE = 00): header bits, pad to a byte, LEN, ~LEN, the bytes
            if (in_huff) { bw.put(T.lit_code[256], T.lit_bits[256]); in_huff = false; }
            bw.put(0, 3);
            const uint16_t l16 = (uint16_t)len, nl16 = (uint16_t)~l16;
            uint8_t hdr4[4] = {(uint8_t)l16, (uint8_t)(l16 >> 8), (uint8_t)nl16, (uint8_t)(nl16 >> 8)};
            bw.bytes(hdr4, 4);
            bw.bytes(cur, len);
            continue;
        }
        if (follow && !in_huff) { bw.put(2, 3); in_huff = true; }   // BFINAL = 0, BTYPE = 01
        if (trusted || (r > 0 && len == a - rec[r - 1] && len < 32768 && len >= 36)) {
            // same size as the record before it: whatever equals it comes from there
            const uint8_t *prev = data + rec[r - 1];
            // (the records of a read differ in their first 36 bytes only -- refID, pos, bin, flag: everything behind the last
            // difference there is one comparison and one run of matches, not a loop over ~200 bytes)
            size_t tail = len;                                     // [tail, len) equals the previous record
            if (trusted) {
                tail = 20;
                while (tail > 0 && cur[tail - 1] == prev[tail - 1]) tail--;
            } else if (!memcmp(cur + 36, prev + 36, len - 36)) {
                tail = 36;
                while (tail > 0 && cur[tail - 1] == prev[tail - 1]) tail--;
            }
            size_t i = 0;
            while (i < len) {
                if (i >= tail && len - i >= 3) { copy((uint32_t)(len - i), (uint32_t)len); break; }
                if (cur[i] == prev[i]) {
                    size_t j = i + 1;
                    while (j < len && cur[j] == prev[j]) j++;
                    if (j - i >= 3) { copy((uint32_t)(j - i), (uint32_t)len); i = j; continue; }
                    while (i < j) lit(cur[i++]);
                } else lit(cur[i++]);
            }
        } else plain(cur, len);
    }
    if (in_huff) bw.put(T.lit_code[256], T.lit_bits[256]);         // end of block
    if (follow) { bw.put(3, 3); bw.put(T.lit_code[256], T.lit_bits[256]); }   // ... and an empty final block (BFINAL = 1, fixed Huffman, end of block)
    uint8_t *end = bw.finish();
    const size_t clen = (size_t)(end - (out.data() + at + 18));
    const size_t total = 18 + clen + 8;
    if (total > 0x10000) { out.resize(at); return false; }
    static const uint8_t hdr[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
    memcpy(out.data() + at, hdr, 16);
    const uint16_t bsize = (uint16_t)(total - 1);
    out[at + 16] = (uint8_t)bsize; out[at + 17] = (uint8_t)(bsize >> 8);
    const uint32_t crc = fast_crc32(data, n), isize = (uint32_t)n;
    memcpy(out.data() + at + 18 + clen, &crc, 4);
    memcpy(out.data() + at + 18 + clen + 4, &isize, 4);
    out.resize(at + total);
    return true;
}

} // namespace groot

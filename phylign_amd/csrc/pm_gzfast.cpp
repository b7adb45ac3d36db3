// pm_gzfast.cpp -- the `gzip --fast` of the 03_match files (Snakefile:427, :468, :483), written for the
// text this library prints: one gzip member per call, one fixed-Huffman deflate block, matches found by the line
// structure of cobs / post-filter output instead of a byte-wise hash chain:
//   * a line is compared, byte for byte at one constant distance, with ONE earlier line: the last line that began with
//     the same name (the bytes before the first tab: a reference that was hit a few queries ago, found through a small
//     hash table of line starts) or else the last line that was not a "*" line, for "*" lines the previous "*" line
//     (read names count up), else the line before;
//   * equal stretches of 4+ bytes (3+ at distances up to 128) become (length, distance) pairs, everything else literals.
// The consumers only ever inflate the stream (scripts/filter_queries.py:46 through xopen; `gzip -dc`), so the
// contract is the decoded bytes, and RFC 1951 / 1952 validity -- tests/test_golden_cpu.py decodes every shape with
// Python's gzip.  zlib level 1 spends ~10 ns per byte on this text, this encoder ~1; at a million reads deflate
// was the largest single cost of a clustered 03_match run.
#include "pm_host.h"
#include <zlib.h>

namespace {

struct Tables {
    uint16_t lit_code[257]; uint8_t lit_bits[257];          // literals + end of block (256), bit-reversed codes
    uint32_t len_code[259]; uint8_t len_bits[259];          // length 3..258: reversed code | extra << code bits
    uint8_t dist_sym[512];                                   // zlib's two-level distance-symbol lookup
    uint16_t dist_base[30]; uint8_t dist_extra[30];
    uint8_t dist_rev[30];                                    // 5-bit reversed symbol
    Tables() {
        auto rev = [](uint32_t v, int n) { uint32_t r = 0; for (int i = 0; i < n; ++i) { r = (r << 1) | (v & 1u); v >>= 1; } return r; };
        for (int s = 0; s <= 256; ++s) {                     // RFC 1951 3.2.6
            if (s < 144) { lit_code[s] = (uint16_t)rev(0x30u + (uint32_t)s, 8); lit_bits[s] = 8; }
            else if (s < 256) { lit_code[s] = (uint16_t)rev(0x190u + (uint32_t)(s - 144), 9); lit_bits[s] = 9; }
            else { lit_code[s] = 0; lit_bits[s] = 7; }
        }
        static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
        static const uint8_t lextra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
        for (int len = 3; len <= 258; ++len) {
            int s = 28;
            while (lbase[s] > len) --s;
            if (len == 258) s = 28;
            const int sym = 257 + s;
            uint32_t code; int bits;
            if (sym < 280) { code = rev((uint32_t)(sym - 256), 7); bits = 7; }
            else { code = rev(0xC0u + (uint32_t)(sym - 280), 8); bits = 8; }
            len_code[len] = code | ((uint32_t)(len - lbase[s]) << bits);
            len_bits[len] = (uint8_t)(bits + lextra[s]);
        }
        static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073,
                                           4097, 6145, 8193, 12289, 16385, 24577};
        static const uint8_t dextra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
        for (int s = 0; s < 30; ++s) { dist_base[s] = dbase[s]; dist_extra[s] = dextra[s]; dist_rev[s] = (uint8_t)rev((uint32_t)s, 5); }
        for (int d = 1; d <= 256; ++d) { int s = 29; while (dbase[s] > d) --s; dist_sym[d - 1] = (uint8_t)s; }
        for (int i = 0; i < 256; ++i) {                       // distances 257..32768 by (d - 1) >> 7
            const int d = (i << 7) + 1;
            int s = 29; while (dbase[s] > std::max(d, 257)) --s;
            dist_sym[256 + i] = (uint8_t)s;
        }
    }
};
const Tables& tables() { static const Tables t; return t; }

struct BitWriter {
    uint8_t* p; uint64_t acc = 0; int n = 0;
    inline void put(uint64_t v, int bits) {                  // bits <= 32 per call, LSB first
        acc |= v << n; n += bits;
        if (n >= 32) { const uint32_t w = (uint32_t)acc; memcpy(p, &w, 4); p += 4; acc >>= 32; n -= 32; }
    }
    inline uint8_t* finish() { while (n > 0) { *p++ = (uint8_t)acc; acc >>= 8; n -= 8; } n = 0; return p; }
};

inline uint64_t load64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
inline size_t eq_run(const uint8_t* a, const uint8_t* b, size_t max) {
    size_t k = 0;
    while (k + 8 <= max) {
        const uint64_t x = load64(a + k) ^ load64(b + k);
        if (x) return k + ((size_t)__builtin_ctzll(x) >> 3);
        k += 8;
    }
    while (k < max && a[k] == b[k]) ++k;
    return k;
}
inline uint32_t name_hash(const uint8_t* p, size_t n) {
    uint64_t h = 0x9E3779B97F4A7C15ull ^ n;
    size_t j = 0;
    for (; j + 8 <= n; j += 8) h = (h ^ load64(p + j)) * 0xFF51AFD7ED558CCDull;
    if (j < n) { uint64_t t = 0; memcpy(&t, p + j, n - j); h = (h ^ t) * 0xFF51AFD7ED558CCDull; }
    return (uint32_t)(h >> 52);                              // 12 bits
}

}  // namespace

// room one member of n text bytes may take (every byte a 9-bit literal, header, trailer, the writer's 4-byte stores)
size_t gz_fast_bound(size_t n) { return n + n / 8 + 96; }

// one gzip member holding text[0, n), n < 2^31, written to o[0, gz_fast_bound(n)); returns its length
size_t gz_fast_member(const char* text, size_t n, uint8_t* o) {
    const Tables& T = tables();
    const uint8_t* s = (const uint8_t*)text;
    static const uint8_t head[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 4, 3};      // deflate, no name / time, "fastest", unix
    memcpy(o, head, 10);
    BitWriter w; w.p = o + 10;
    w.put(1, 1); w.put(1, 2);                                 // final block, fixed Huffman codes
    constexpr uint32_t kNone = 0xFFFFFFFFu;
    std::vector<uint32_t> tab(1u << 12, kNone);
    uint32_t prev_line = kNone, prev_star = kNone, prev_plain = kNone;
    size_t pos = 0;
    while (pos < n) {
        const uint8_t* nlp = (const uint8_t*)memchr(s + pos, '\n', n - pos);
        const size_t len = nlp ? (size_t)(nlp - (s + pos)) + 1 : n - pos;
        uint32_t ref = kNone;
        if (s[pos] == '*') {
            ref = prev_star;
            prev_star = (uint32_t)pos;
        } else {
            const uint8_t* tb = (const uint8_t*)memchr(s + pos, '\t', len);
            const size_t nl = tb ? (size_t)(tb - (s + pos)) : len;
            const uint32_t h = name_hash(s + pos, nl);
            const uint32_t cand = tab[h];
            tab[h] = (uint32_t)pos;
            if (cand != kNone && pos - cand <= 32768 && eq_run(s + pos, s + cand, std::min<size_t>(len, 8)) >= 4) ref = cand;
            else ref = prev_plain;                            // another reference's line: same shape, often the same score
            prev_plain = (uint32_t)pos;
        }
        if (ref == kNone || pos - ref > 32768) ref = prev_line;
        if (ref != kNone && pos - ref > 32768) ref = kNone;
        prev_line = (uint32_t)pos;
        if (ref == kNone) {
            for (size_t i = 0; i < len; ++i) w.put(T.lit_code[s[pos + i]], T.lit_bits[s[pos + i]]);
        } else {
            const size_t d = pos - ref;
            const uint32_t ds = d <= 256 ? T.dist_sym[d - 1] : T.dist_sym[256 + ((d - 1) >> 7)];
            const uint64_t dcode = (uint64_t)T.dist_rev[ds] | ((uint64_t)(d - T.dist_base[ds]) << 5);
            const int dbits = 5 + T.dist_extra[ds];
            const size_t min_run = dbits <= 10 ? 3 : 4;        // a 3-byte match pays only at a short distance
            size_t i = 0;
            while (i < len) {
                const size_t run = eq_run(s + pos + i, s + ref + i, std::min<size_t>(len - i, 258));
                if (run >= min_run) {
                    w.put((uint64_t)T.len_code[run] | (dcode << T.len_bits[run]), T.len_bits[run] + dbits);
                    i += run;
                } else {
                    const size_t lits = run + 1 <= len - i ? run + 1 : len - i;       // the equal bytes and the one that differs
                    for (size_t k = 0; k < lits; ++k) w.put(T.lit_code[s[pos + i + k]], T.lit_bits[s[pos + i + k]]);
                    i += lits;
                }
            }
        }
        pos += len;
    }
    w.put(T.lit_code[256], T.lit_bits[256]);
    uint8_t* e = w.finish();
    uint32_t crc = (uint32_t)crc32(0L, Z_NULL, 0);
    for (size_t off = 0; off < n;) {                          // crc32() takes a 32-bit length
        const size_t step = std::min<size_t>(n - off, 1u << 30);
        crc = (uint32_t)crc32(crc, (const Bytef*)(s + off), (uInt)step);
        off += step;
    }
    const uint32_t isize = (uint32_t)n;
    memcpy(e, &crc, 4); memcpy(e + 4, &isize, 4);
    return (size_t)(e + 8 - o);
}

// pm_gzfast.cpp -- the `gzip --fast` of the 03_match files (Snakefile:427, :468, :483), written for the
// text this library prints: one gzip member per call, one deflate block (Huffman codes built from the member's own
// counts; the fixed codes when those are no worse), matches found by the line structure of cobs / post-filter output
// instead of a byte-wise hash chain:
//   * a line is compared, byte for byte at one constant distance, with ONE earlier line: the last line that began with
//     the same name (the bytes before the first tab: a reference that was hit a few queries ago, found through a small
//     hash table of line starts) or else the last line that was not a "*" line, for "*" lines the previous "*" line
//     (read names count up), else the line before;
//   * equal stretches of 4+ bytes (3+ at distances up to 128) become (length, distance) pairs, everything else literals.
// The consumers only ever inflate the stream (scripts/filter_queries.py:46 through xopen; `gzip -dc`), so the
// contract is the decoded bytes, and RFC 1951 / 1952 validity -- tests/test_golden_cpu.py decodes every shape with
// Python's gzip.  zlib level 1 spends ~10 ns per byte on this text, this encoder 1-2; at a million reads deflate
// was the largest single cost of a clustered 03_match run (profiles/r03/NOTES.md section 6, config 5).
#include "pm_host.h"
#include <zlib.h>

namespace {

struct Tables {
    uint16_t lit_code[257]; uint8_t lit_bits[257];          // literals + end of block (256), bit-reversed codes
    uint16_t len_sym[259];                                   // length 3..258 -> symbol 257..285
    uint16_t len_base[29]; uint8_t len_extra[29];
    uint16_t fix_len_code[29];                               // fixed-Huffman code of a length symbol, bit-reversed (7 or 8 bits)
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
        for (int i = 0; i < 29; ++i) {
            len_base[i] = lbase[i]; len_extra[i] = lextra[i];
            const int sym = 257 + i;
            fix_len_code[i] = (uint16_t)(sym < 280 ? rev((uint32_t)(sym - 256), 7) : rev(0xC0u + (uint32_t)(sym - 280), 8));
        }
        for (int len = 3; len <= 258; ++len) {
            int si = 28;
            while (lbase[si] > len) --si;
            len_sym[len] = (uint16_t)(257 + si);
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

// ---- Huffman code lengths (length-limited) and canonical codes for one alphabet
// lens[i] = 0 for unused symbols.  Plain two-queue construction over the used symbols sorted by count, then, when a
// code would be longer than maxbits, the usual repair: clamp, and lengthen the deepest shorter codes until the Kraft
// sum fits again.
void huff_lengths(const uint32_t* freq, int n, int maxbits, uint8_t* lens) {
    memset(lens, 0, (size_t)n);
    int order[288], m = 0;
    for (int i = 0; i < n; ++i) if (freq[i]) order[m++] = i;
    if (m == 0) return;
    if (m == 1) { lens[order[0]] = 1; return; }
    std::sort(order, order + m, [&](int a, int b) { return freq[a] != freq[b] ? freq[a] < freq[b] : a < b; });
    // nodes 0..m-1 leaves (ascending count), m..2m-2 internal in creation order (ascending count as well)
    uint64_t w[2 * 288]; int parent[2 * 288];
    for (int i = 0; i < m; ++i) w[i] = freq[order[i]];
    int leaf = 0, inner = m, made = m;
    auto take = [&]() {
        if (leaf < m && (inner >= made || w[leaf] <= w[inner])) return leaf++;
        return inner++;
    };
    while (made < 2 * m - 1) {
        const int a = take(), b = take();
        w[made] = w[a] + w[b];
        parent[a] = parent[b] = made;
        ++made;
    }
    int depth[2 * 288];
    depth[2 * m - 2] = 0;
    for (int i = 2 * m - 3; i >= 0; --i) depth[i] = depth[parent[i]] + 1;
    // codes per length; anything deeper than maxbits is first counted at maxbits, which over-subscribes the code: each
    // repair step takes one code away from maxbits, moves one shorter code a level down and hangs the freed code beside
    // it -- the Kraft sum falls by exactly one unit (2^-maxbits) per step, so it lands on 1 and the code stays complete
    // (inflate rejects incomplete codes).  Lengths are then dealt out again: the rarest symbols get the longest codes.
    uint32_t count[64] = {0};
    bool over = false;
    for (int i = 0; i < m; ++i) { if (depth[i] > maxbits) { depth[i] = maxbits; over = true; } count[depth[i]]++; }
    if (over) {
        uint64_t total = 0;
        for (int l = 1; l <= maxbits; ++l) total += (uint64_t)count[l] << (maxbits - l);
        while (total != (1ull << maxbits)) {
            count[maxbits]--;
            for (int l = maxbits - 1; l > 0; --l)
                if (count[l]) { count[l]--; count[l + 1] += 2; break; }
            --total;
        }
        int i = 0;
        for (int l = maxbits; l >= 1; --l)
            for (uint32_t c = 0; c < count[l]; ++c) depth[i++] = l;
    }
    for (int i = 0; i < m; ++i) lens[order[i]] = (uint8_t)depth[i];
}
// canonical codes (RFC 1951 3.2.2), bit-reversed for an LSB-first writer
void huff_codes(const uint8_t* lens, int n, uint16_t* codes) {
    uint32_t count[16] = {0}, next[16];
    for (int i = 0; i < n; ++i) count[lens[i]]++;
    count[0] = 0;
    uint32_t code = 0;
    for (int b = 1; b <= 15; ++b) { code = (code + count[b - 1]) << 1; next[b] = code; }
    for (int i = 0; i < n; ++i) {
        if (!lens[i]) { codes[i] = 0; continue; }
        uint32_t c = next[lens[i]]++, r = 0;
        for (int b = 0; b < lens[i]; ++b) { r = (r << 1) | (c & 1u); c >>= 1; }
        codes[i] = (uint16_t)r;
    }
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
    // ---- pass 1: tokens.  A literal is its byte; a match is 1 << 31 | (length - 3) << 16 | (distance - 1).
    static thread_local std::vector<uint32_t> tokens;          // the workers are persistent: the buffer keeps its pages
    static thread_local std::vector<uint32_t> tab;
    tokens.clear();
    if (tokens.capacity() < n / 2 + 64) tokens.reserve(n / 2 + 64);
    constexpr uint32_t kNone = 0xFFFFFFFFu;
    tab.assign(1u << 12, kNone);
    uint32_t lfreq[288] = {0}, dfreq[30] = {0};
    auto dist_symbol = [&](size_t d) { return d <= 256 ? T.dist_sym[d - 1] : T.dist_sym[256 + ((d - 1) >> 7)]; };
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
            for (size_t i = 0; i < len; ++i) { tokens.push_back(s[pos + i]); lfreq[s[pos + i]]++; }
        } else {
            const size_t d = pos - ref;
            const uint32_t ds = dist_symbol(d);
            const size_t min_run = T.dist_extra[ds] <= 5 ? 3 : 4;     // a 3-byte match pays only at a short distance
            size_t i = 0;
            while (i < len) {
                const size_t run = eq_run(s + pos + i, s + ref + i, std::min<size_t>(len - i, 258));
                if (run >= min_run) {
                    tokens.push_back(0x80000000u | ((uint32_t)(run - 3) << 16) | (uint32_t)(d - 1));
                    lfreq[T.len_sym[run]]++; dfreq[ds]++;
                    i += run;
                } else {
                    const size_t lits = run + 1 <= len - i ? run + 1 : len - i;       // the equal bytes and the one that differs
                    for (size_t k = 0; k < lits; ++k) { tokens.push_back(s[pos + i + k]); lfreq[s[pos + i + k]]++; }
                    i += lits;
                }
            }
        }
        pos += len;
    }
    lfreq[256] = 1;
    // ---- the block's codes: dynamic Huffman from the counts, unless the fixed codes are no worse (tiny members)
    uint8_t llen[288], dlen[30], clen[19];
    huff_lengths(lfreq, 286, 15, llen);
    huff_lengths(dfreq, 30, 15, dlen);
    bool any_dist = false;
    for (int i = 0; i < 30; ++i) any_dist |= dlen[i] != 0;
    if (!any_dist) dlen[0] = 1;                               // "one distance code": a block without matches still names one
    int hlit = 286, hdist = 30;
    while (hlit > 257 && !llen[hlit - 1]) --hlit;
    while (hdist > 1 && !dlen[hdist - 1]) --hdist;
    uint32_t cfreq[19] = {0};
    for (int i = 0; i < hlit; ++i) cfreq[llen[i]]++;
    for (int i = 0; i < hdist; ++i) cfreq[dlen[i]]++;
    huff_lengths(cfreq, 19, 7, clen);                        // code lengths are sent one by one (no repeat codes: ~150 bytes per member)
    static const uint8_t corder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    int hclen = 19;
    while (hclen > 4 && !clen[corder[hclen - 1]]) --hclen;
    uint64_t dyn_bits = 3 + 5 + 5 + 4 + 3ull * (uint64_t)hclen, fix_bits = 3;
    for (int i = 0; i < hlit; ++i) dyn_bits += clen[llen[i]];
    for (int i = 0; i < hdist; ++i) dyn_bits += clen[dlen[i]];
    for (int i = 0; i < 286; ++i) {
        const uint64_t extra = i >= 257 ? T.len_extra[i - 257] : 0;
        dyn_bits += (uint64_t)lfreq[i] * (llen[i] + extra);
        fix_bits += (uint64_t)lfreq[i] * ((i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8) + extra);
    }
    for (int i = 0; i < 30; ++i) {
        dyn_bits += (uint64_t)dfreq[i] * (dlen[i] + T.dist_extra[i]);
        fix_bits += (uint64_t)dfreq[i] * (5 + T.dist_extra[i]);
    }
    int cl_used = 0;
    for (int i = 0; i < 19; ++i) cl_used += clen[i] != 0;
    const bool dynamic = dyn_bits < fix_bits && cl_used >= 2;      // (a one-code code-length code is not a complete code)
    // ---- pass 2: the bits
    BitWriter w; w.p = o + 10;
    uint16_t lcode[288], dcode[30];
    uint8_t lbits[288], dbits[30];
    if (dynamic) {
        uint16_t ccode[19];
        huff_codes(llen, 286, lcode); huff_codes(dlen, 30, dcode); huff_codes(clen, 19, ccode);
        memcpy(lbits, llen, 286); memcpy(dbits, dlen, 30);
        w.put(1, 1); w.put(2, 2);                             // final block, dynamic Huffman codes
        w.put((uint64_t)(hlit - 257), 5); w.put((uint64_t)(hdist - 1), 5); w.put((uint64_t)(hclen - 4), 4);
        for (int i = 0; i < hclen; ++i) w.put(clen[corder[i]], 3);
        for (int i = 0; i < hlit; ++i) w.put(ccode[llen[i]], clen[llen[i]]);
        for (int i = 0; i < hdist; ++i) w.put(ccode[dlen[i]], clen[dlen[i]]);
    } else {
        w.put(1, 1); w.put(1, 2);                             // final block, fixed Huffman codes
        for (int i = 0; i < 286; ++i) {
            if (i <= 256) { lcode[i] = T.lit_code[i]; lbits[i] = T.lit_bits[i]; }
            else { lcode[i] = T.fix_len_code[i - 257]; lbits[i] = (uint8_t)(i < 280 ? 7 : 8); }
        }
        for (int i = 0; i < 30; ++i) { dcode[i] = T.dist_rev[i]; dbits[i] = 5; }
    }
    for (const uint32_t t : tokens) {
        if (!(t & 0x80000000u)) { w.put(lcode[t], lbits[t]); continue; }
        const uint32_t run = ((t >> 16) & 0xFFu) + 3u, d = (t & 0xFFFFu) + 1u;
        const uint32_t ls = T.len_sym[run], ds = dist_symbol(d);
        // length code + extra (<= 15 + 5 bits), distance code + extra (<= 15 + 13 bits): two puts
        w.put((uint64_t)lcode[ls] | ((uint64_t)(run - T.len_base[ls - 257]) << lbits[ls]), lbits[ls] + T.len_extra[ls - 257]);
        w.put((uint64_t)dcode[ds] | ((uint64_t)(d - T.dist_base[ds]) << dbits[ds]), dbits[ds] + T.dist_extra[ds]);
    }
    w.put(lcode[256], lbits[256]);
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

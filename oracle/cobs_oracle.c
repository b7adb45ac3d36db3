/* cobs_oracle.c -- CPU restatement of `cobs query` over a COBS classic index.
 *
 * TEST INFRASTRUCTURE ONLY (see cobs_oracle.h).  "parity unpinned": the COBS
 * 0.2.1 binary/source is not available (envs/cobs.yaml:5 pins it, nothing
 * vendors it), so every COBS rule below is a restatement of the published
 * algorithm (Bingmann et al., "COBS: a Compact Bit-Sliced Signature Index",
 * SPIRE 2019; upstream files named per function) anchored on the reference's
 * call sites and grammar witnesses.  Each uncertain rule sits in ONE function.
 */
#define _GNU_SOURCE
#include "cobs_oracle.h"
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <emmintrin.h>

/* ------------------------------------------------------------------ XXH64 */
/* Published XXH64 (xxHash 0.8.2 spec, doc/xxhash_spec.md); COBS hashes each
 * term with XXH64(term bytes, term_size, seed = hash function number)
 * (upstream cobs/query/classic_search.cpp, create_hashes).  Reference call
 * sites of the binary: scripts/run_cobs_streaming.sh:24-29. */
#define P1 0x9E3779B185EBCA87ULL
#define P2 0xC2B2AE3D27D4EB4FULL
#define P3 0x165667B19E3779F9ULL
#define P4 0x85EBCA77C2B2AE63ULL
#define P5 0x27D4EB2F165667C5ULL
static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static inline uint64_t rd64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
static inline uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint64_t xxh_round(uint64_t acc, uint64_t in) {
    acc += in * P2; acc = rotl64(acc, 31); return acc * P1;
}
static inline uint64_t xxh_merge(uint64_t acc, uint64_t v) {
    v = xxh_round(0, v); acc ^= v; return acc * P1 + P4;
}
uint64_t orc_xxh64(const void* data, size_t len, uint64_t seed) {
    const uint8_t* p = (const uint8_t*)data;
    const uint8_t* end = p + len;
    uint64_t h;
    if (len >= 32) {
        uint64_t v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
        const uint8_t* lim = end - 32;
        do {
            v1 = xxh_round(v1, rd64(p));      v2 = xxh_round(v2, rd64(p + 8));
            v3 = xxh_round(v3, rd64(p + 16)); v4 = xxh_round(v4, rd64(p + 24));
            p += 32;
        } while (p <= lim);
        h = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
        h = xxh_merge(h, v1); h = xxh_merge(h, v2);
        h = xxh_merge(h, v3); h = xxh_merge(h, v4);
    } else {
        h = seed + P5;
    }
    h += (uint64_t)len;
    while (p + 8 <= end) { h ^= xxh_round(0, rd64(p)); h = rotl64(h, 27) * P1 + P4; p += 8; }
    if (p + 4 <= end)    { h ^= (uint64_t)rd32(p) * P1; h = rotl64(h, 23) * P2 + P3; p += 4; }
    while (p < end)      { h ^= (uint64_t)(*p) * P5;    h = rotl64(h, 11) * P1;      p++; }
    h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
    return h;
}

/* ------------------------------------------------------- canonicalisation */
/* Canonical k-mer = lexicographically smaller (ASCII) of the k-mer and its
 * reverse complement (upstream cobs/kmer.hpp canonicalize_kmer).  Returns 0
 * when the k-mer holds a byte outside ACGT; the Phylign pipeline never sends
 * one (Snakefile:327-332 maps non-ACGT to A). */
static inline char comp_base(char c) {
    switch (c) { case 'A': return 'T'; case 'C': return 'G';
                 case 'G': return 'C'; case 'T': return 'A'; default: return 0; }
}
int orc_canonicalize(const char* kmer, size_t k, char* out) {
    int use_rc = 0, decided = 0;
    for (size_t i = 0; i < k; i++) {
        char f = kmer[i], r = comp_base(kmer[k - 1 - i]);
        if (!comp_base(f) || !r) return 0;
        if (!decided && f != r) { use_rc = (r < f); decided = 1; }
    }
    for (size_t i = 0; i < k; i++)
        out[i] = use_rc ? comp_base(kmer[k - 1 - i]) : kmer[i];
    return 1;
}

/* -------------------------------------------------------------- threshold */
/* Minimum score kept: ceil(threshold * num_terms) in IEEE double (upstream
 * cobs/query/classic_search.cpp, counts_to_result).  `-t 0.7` comes from
 * config.yaml:20 through Snakefile:410 / :451. */
/* The two rules no file of the reference pins (SURVEY.md 8c "unresolvable here") are switchable, like the product's
 * pm_set_option("cobs_threshold_rule" / "cobs_tie_order"): tools/pin_against_cobs.sh decides them against a real
 * cobs 0.2.1.  threshold_rule 0 = ceil (default), 1 = floor, 2 = round half up; tie_desc 1 = equal scores by
 * descending document index (default 0: ascending). */
static int g_threshold_rule = 0, g_tie_desc = 0;
void orc_set_rules(int threshold_rule, int tie_desc) { g_threshold_rule = threshold_rule; g_tie_desc = tie_desc; }
uint32_t orc_threshold(double threshold, uint64_t num_terms) {
    const double x = threshold * (double)num_terms;
    double t = g_threshold_rule == 1 ? floor(x) : (g_threshold_rule == 2 ? floor(x + 0.5) : ceil(x));
    if (t < 0) t = 0;
    if (t > 4294967295.0) t = 4294967295.0;
    return (uint32_t)t;
}

/* ------------------------------------------------------- classic header   */
/* upstream cobs/file/header.hpp + cobs/file/classic_index_header.cpp:
 *   "COBS:" "CLASSIC_INDEX" u32 version | fields | names '\n'... | "CLASSIC_INDEX" | matrix
 * Field order is recalled, not verifiable here; two orders are accepted and
 * the one whose trailing magic validates wins:
 *   layout 0: u32 term_size, u8 canon, u32 n_docs, u64 sig_size, u64 num_hashes
 *   layout 1: u32 term_size, u8 canon, u64 sig_size, u64 num_hashes, u32 n_docs
 */
static const char MAGIC0[] = "COBS:";
static const char MAGIC1[] = "CLASSIC_INDEX";
static int try_layout(const uint8_t* buf, size_t len, int layout, orc_header_t* h) {
    size_t o = 5 + 13;
    if (len < o + 4 + 4 + 1 + 4 + 8 + 8) return -1;
    memcpy(&h->version, buf + o, 4); o += 4;
    memcpy(&h->term_size, buf + o, 4); o += 4;
    h->canonicalize = buf[o]; o += 1;
    if (layout == 0) {
        memcpy(&h->n_docs, buf + o, 4); o += 4;
        memcpy(&h->signature_size, buf + o, 8); o += 8;
        memcpy(&h->num_hashes, buf + o, 8); o += 8;
    } else {
        memcpy(&h->signature_size, buf + o, 8); o += 8;
        memcpy(&h->num_hashes, buf + o, 8); o += 8;
        memcpy(&h->n_docs, buf + o, 4); o += 4;
    }
    h->names_off = o;
    for (uint32_t d = 0; d < h->n_docs; d++) {
        const uint8_t* nl = (o < len) ? memchr(buf + o, '\n', len - o) : NULL;
        if (!nl) return -1;
        o = (size_t)(nl - buf) + 1;
    }
    if (o + 13 > len || memcmp(buf + o, MAGIC1, 13) != 0) return -1;
    o += 13;
    h->data_off = o;
    h->row_bytes = ((uint64_t)h->n_docs + 7) / 8;
    h->layout = layout;
    if (h->version != 1 || h->term_size == 0 || h->canonicalize > 1) return -1;
    if (h->signature_size == 0 || h->num_hashes == 0) return -1;
    /* overflow-safe: data must hold signature_size rows */
    if (h->row_bytes && h->signature_size > (len - o) / h->row_bytes) return -1;
    return 0;
}
int orc_header_parse(const uint8_t* buf, size_t len, orc_header_t* h) {
    if (len < 18 || memcmp(buf, MAGIC0, 5) || memcmp(buf + 5, MAGIC1, 13)) return -1;
    if (try_layout(buf, len, 0, h) == 0) return 0;
    if (try_layout(buf, len, 1, h) == 0) return 0;
    return -1;
}
uint8_t* orc_index_alloc(uint32_t term_size, uint8_t canon, uint64_t sig_size,
                         uint64_t num_hashes, uint32_t n_docs,
                         const char* const* names, size_t* total_len, size_t* data_off) {
    size_t hl = 5 + 13 + 4 + 4 + 1 + 4 + 8 + 8 + 13;
    for (uint32_t d = 0; d < n_docs; d++) hl += strlen(names[d]) + 1;
    uint64_t rb = ((uint64_t)n_docs + 7) / 8;
    size_t tot = hl + (size_t)(sig_size * rb);
    uint8_t* b = (uint8_t*)calloc(tot ? tot : 1, 1);
    if (!b) return NULL;
    size_t o = 0; uint32_t ver = 1;
    memcpy(b + o, MAGIC0, 5); o += 5; memcpy(b + o, MAGIC1, 13); o += 13;
    memcpy(b + o, &ver, 4); o += 4; memcpy(b + o, &term_size, 4); o += 4;
    b[o++] = canon;
    memcpy(b + o, &n_docs, 4); o += 4;
    memcpy(b + o, &sig_size, 8); o += 8; memcpy(b + o, &num_hashes, 8); o += 8;
    for (uint32_t d = 0; d < n_docs; d++) {
        size_t l = strlen(names[d]); memcpy(b + o, names[d], l); o += l; b[o++] = '\n';
    }
    memcpy(b + o, MAGIC1, 13); o += 13;
    *total_len = tot; *data_off = o;
    return b;
}

/* ----------------------------------------------------------------- hashes */
/* upstream cobs/query/classic_search.cpp create_hashes: term i = seq[i..i+k),
 * canonicalised when the index says so, hashed with seeds 0..num_hashes-1. */
int orc_create_hashes(const char* seq, size_t len, uint32_t k, int canon,
                      uint64_t num_hashes, uint64_t* hashes) {
    if (len < k) return -1;
    size_t nt = len - k + 1;
    char* buf = (char*)malloc(k);
    for (size_t i = 0; i < nt; i++) {
        const char* term = seq + i;
        if (canon) { if (!orc_canonicalize(seq + i, k, buf)) { free(buf); return -2; } term = buf; }
        for (uint64_t j = 0; j < num_hashes; j++)
            hashes[i * num_hashes + j] = orc_xxh64(term, k, j);
    }
    free(buf);
    return 0;
}

/* ----------------------------------------------------------------- scores */
/* score[d] = number of terms whose num_hashes rows (row = hash % sig_size,
 * upstream classic_index/mmap_search_file.cpp read_from_disk) all hold bit d;
 * document d <-> byte d/8, bit d%8 from the LSB.  Obviously-correct form. */
int orc_scores(const uint8_t* matrix, uint64_t stride, const orc_header_t* h,
               const char* seq, size_t len, uint32_t* scores) {
    if (len < h->term_size) return -1;
    size_t nt = len - h->term_size + 1;
    uint64_t* hs = (uint64_t*)malloc(nt * h->num_hashes * 8);
    int rc = orc_create_hashes(seq, len, h->term_size, h->canonicalize, h->num_hashes, hs);
    if (rc) { free(hs); return rc; }
    memset(scores, 0, (size_t)h->n_docs * 4);
    uint8_t* acc = (uint8_t*)malloc(h->row_bytes ? h->row_bytes : 1);
    for (size_t i = 0; i < nt; i++) {
        memset(acc, 0xFF, h->row_bytes);
        for (uint64_t j = 0; j < h->num_hashes; j++) {
            const uint8_t* row = matrix + (hs[i * h->num_hashes + j] % h->signature_size) * stride;
            for (uint64_t b = 0; b < h->row_bytes; b++) acc[b] &= row[b];
        }
        for (uint32_t d = 0; d < h->n_docs; d++) scores[d] += (acc[d >> 3] >> (d & 7)) & 1;
    }
    free(acc); free(hs);
    return 0;
}

/* ----------------------------------------------------- threshold + order  */
/* upstream counts_to_result: keep score >= orc_threshold(); threshold 0 keeps
 * every document; order by score descending then document index ascending
 * (comparator std::tie(score[b], a) < std::tie(score[a], b)); num_results 0
 * means all.  Downstream never depends on tie order
 * (scripts/postprocess_cobs.py:31-39 keeps whole tie groups,
 * scripts/filter_queries.py:135 re-sorts). */
static int cmp_hit(const void* a, const void* b) {
    const orc_hit_t* x = (const orc_hit_t*)a; const orc_hit_t* y = (const orc_hit_t*)b;
    if (x->score != y->score) return x->score > y->score ? -1 : 1;
    if (g_tie_desc) return x->doc > y->doc ? -1 : (x->doc < y->doc);
    return x->doc < y->doc ? -1 : (x->doc > y->doc);
}
size_t orc_select(const uint32_t* scores, uint32_t n_docs, uint64_t num_terms,
                  double threshold, size_t num_results, orc_hit_t* hits) {
    uint32_t t = (threshold == 0.0) ? 0 : orc_threshold(threshold, num_terms);
    size_t n = 0;
    for (uint32_t d = 0; d < n_docs; d++)
        if (scores[d] >= t) { hits[n].doc = d; hits[n].score = scores[d]; n++; }
    qsort(hits, n, sizeof(orc_hit_t), cmp_hit);
    if (num_results && num_results < n) n = num_results;
    return n;
}

/* ---------------------------------------------------------- compact index */
/* upstream cobs/file/compact_index_header.cpp serialize():
 *   "COBS:" "COMPACT_INDEX" u32 version=1 | u32 term_size, u8 canonicalize,
 *   u32 n_parameters, u32 n_file_names, u64 page_size |
 *   n_parameters x {u64 signature_size, u64 num_hashes} | names '\n'... |
 *   zero padding so that the closing "COMPACT_INDEX" ends on a page boundary |
 *   "COMPACT_INDEX" | sub-index 0 (signature_size_0 x page_size bytes) | sub-index 1 ...
 * Sub-index p holds documents [p*page_size*8, (p+1)*page_size*8).  Query
 * (upstream compact_index/mmap_search_file.cpp): row of sub-index p = hash %
 * signature_size_p; hash functions 0..num_hashes_p-1. */
static const char MAGICC[] = "COMPACT_INDEX";
int orc_compact_parse(const uint8_t* buf, size_t len, orc_compact_t* c) {
    if (len < 18 + 4 + 4 + 1 + 4 + 4 + 8 || memcmp(buf, MAGIC0, 5) || memcmp(buf + 5, MAGICC, 13)) return -1;
    size_t o = 18; uint32_t ver;
    memcpy(&ver, buf + o, 4); o += 4;
    memcpy(&c->term_size, buf + o, 4); o += 4;
    c->canonicalize = buf[o]; o += 1;
    memcpy(&c->n_parts, buf + o, 4); o += 4;
    memcpy(&c->n_docs, buf + o, 4); o += 4;
    memcpy(&c->page_size, buf + o, 8); o += 8;
    if (ver != 1 || c->term_size == 0 || c->canonicalize > 1 || c->page_size == 0 || c->n_parts == 0) return -1;
    if ((uint64_t)c->n_parts * c->page_size * 8 < c->n_docs) return -1;
    c->params_off = o;
    if (o + (size_t)c->n_parts * 16 > len) return -1;
    o += (size_t)c->n_parts * 16;
    c->names_off = o;
    for (uint32_t d = 0; d < c->n_docs; d++) {
        const uint8_t* nl = (o < len) ? memchr(buf + o, '\n', len - o) : NULL;
        if (!nl) return -1;
        o = (size_t)(nl - buf) + 1;
    }
    o += (size_t)((c->page_size - ((o + 13) % c->page_size)) % c->page_size);
    if (o + 13 > len || memcmp(buf + o, MAGICC, 13) != 0) return -1;
    c->data_off = o + 13;
    uint64_t need = 0;
    for (uint32_t p = 0; p < c->n_parts; p++) {
        uint64_t sig, nh; memcpy(&sig, buf + c->params_off + 16 * p, 8); memcpy(&nh, buf + c->params_off + 16 * p + 8, 8);
        if (sig == 0 || nh == 0) return -1;
        need += sig * c->page_size;
    }
    if (need > len - c->data_off) return -1;
    return 0;
}
uint8_t* orc_compact_alloc(uint32_t term_size, uint8_t canon, uint64_t page_size, uint32_t n_parts,
                           const uint64_t* sig_sizes, const uint64_t* num_hashes, uint32_t n_docs,
                           const char* const* names, size_t* total_len, size_t* data_off) {
    size_t hl = 18 + 4 + 4 + 1 + 4 + 4 + 8 + (size_t)n_parts * 16;
    for (uint32_t d = 0; d < n_docs; d++) hl += strlen(names[d]) + 1;
    hl += (size_t)((page_size - ((hl + 13) % page_size)) % page_size) + 13;
    size_t tot = hl;
    for (uint32_t p = 0; p < n_parts; p++) tot += (size_t)(sig_sizes[p] * page_size);
    uint8_t* b = (uint8_t*)calloc(tot, 1);
    if (!b) return NULL;
    size_t o = 0; uint32_t ver = 1;
    memcpy(b + o, MAGIC0, 5); o += 5; memcpy(b + o, MAGICC, 13); o += 13;
    memcpy(b + o, &ver, 4); o += 4; memcpy(b + o, &term_size, 4); o += 4; b[o++] = canon;
    memcpy(b + o, &n_parts, 4); o += 4; memcpy(b + o, &n_docs, 4); o += 4; memcpy(b + o, &page_size, 8); o += 8;
    for (uint32_t p = 0; p < n_parts; p++) { memcpy(b + o, &sig_sizes[p], 8); memcpy(b + o + 8, &num_hashes[p], 8); o += 16; }
    for (uint32_t d = 0; d < n_docs; d++) { size_t l = strlen(names[d]); memcpy(b + o, names[d], l); o += l; b[o++] = '\n'; }
    o = hl - 13;
    memcpy(b + o, MAGICC, 13);
    *total_len = tot; *data_off = hl;
    return b;
}
int orc_scores_compact(const uint8_t* index, const orc_compact_t* c, const char* seq, size_t len, uint32_t* scores) {
    if (len < c->term_size) return -1;
    memset(scores, 0, (size_t)c->n_docs * 4);
    const uint8_t* part = index + c->data_off;
    for (uint32_t p = 0; p < c->n_parts; p++) {
        orc_header_t h; memset(&h, 0, sizeof h);
        memcpy(&h.signature_size, index + c->params_off + 16 * p, 8);
        memcpy(&h.num_hashes, index + c->params_off + 16 * p + 8, 8);
        h.term_size = c->term_size; h.canonicalize = c->canonicalize;
        uint64_t first = (uint64_t)p * c->page_size * 8;
        if (first >= c->n_docs) break;
        uint64_t nd = c->n_docs - first; if (nd > c->page_size * 8) nd = c->page_size * 8;
        h.n_docs = (uint32_t)nd; h.row_bytes = (nd + 7) / 8;
        int rc = orc_scores(part, c->page_size, &h, seq, len, scores + first);
        if (rc) return rc;
        part += h.signature_size * c->page_size;
    }
    return 0;
}

/* ------------------------------------------------------- cobs query -f    */
/* upstream src/main.cpp process_query (file branch): std::getline loop; empty
 * lines skipped; a line starting with '>' or ';' flushes the pending record
 * and becomes the next header with its first byte replaced by '*'; other
 * lines are appended to the sequence; a record with an empty sequence prints
 * nothing.  Output per record: "<header>\t<N>\n" then N x "<doc>\t<score>\n"
 * (witnesses: scripts/postprocess_cobs.py:23-26, scripts/filter_queries.py:51-65). */
typedef struct { char* p; size_t n, cap; } sbuf;
static void sb_put(sbuf* s, const char* d, size_t l) {
    if (s->n + l + 1 > s->cap) { s->cap = (s->n + l + 1) * 2; s->p = (char*)realloc(s->p, s->cap); }
    memcpy(s->p + s->n, d, l); s->n += l; s->p[s->n] = 0;
}
typedef struct { const orc_header_t* h; const orc_compact_t* c; uint32_t term_size, n_docs; } any_index;
static int flush_record(sbuf* out, const uint8_t* index, const any_index* ai,
                        const char* const* names, const size_t* name_len,
                        const char* hdr, size_t hdr_len, const char* seq, size_t seq_len,
                        double threshold, size_t num_results, char* err, size_t errcap) {
    if (seq_len == 0) return 0;
    if (seq_len < ai->term_size) { snprintf(err, errcap, "query too short, needs at least %u characters", ai->term_size); return -1; }
    uint32_t* sc = (uint32_t*)malloc(((size_t)ai->n_docs + 1) * 4);
    int rc = ai->h ? orc_scores(index + ai->h->data_off, ai->h->row_bytes, ai->h, seq, seq_len, sc)
                   : orc_scores_compact(index, ai->c, seq, seq_len, sc);
    if (rc) { free(sc); snprintf(err, errcap, "invalid base in query (only ACGT accepted)"); return -1; }
    orc_hit_t* hits = (orc_hit_t*)malloc(((size_t)ai->n_docs + 1) * sizeof(orc_hit_t));
    size_t n = orc_select(sc, ai->n_docs, seq_len - ai->term_size + 1, threshold, num_results, hits);
    char num[64];
    sb_put(out, hdr, hdr_len);
    sb_put(out, num, (size_t)snprintf(num, sizeof num, "\t%zu\n", n));
    for (size_t i = 0; i < n; i++) {
        sb_put(out, names[hits[i].doc], name_len[hits[i].doc]);
        sb_put(out, num, (size_t)snprintf(num, sizeof num, "\t%u\n", hits[i].score));
    }
    free(hits); free(sc);
    return 0;
}
char* orc_query_file(const uint8_t* index, size_t index_len, const char* fasta, size_t fasta_len,
                     double threshold, size_t num_results, size_t* out_len, char* err, size_t errcap) {
    orc_header_t hc; orc_compact_t cc; any_index ai;
    size_t o;
    if (orc_header_parse(index, index_len, &hc) == 0) {
        ai.h = &hc; ai.c = NULL; ai.term_size = hc.term_size; ai.n_docs = hc.n_docs; o = hc.names_off;
    } else if (orc_compact_parse(index, index_len, &cc) == 0) {
        ai.h = NULL; ai.c = &cc; ai.term_size = cc.term_size; ai.n_docs = cc.n_docs; o = cc.names_off;
    } else { snprintf(err, errcap, "not a COBS classic or compact index"); return NULL; }
    struct { uint32_t n_docs; } h = { ai.n_docs };
    const char** names = (const char**)malloc(((size_t)h.n_docs + 1) * sizeof(char*));
    size_t* nlen = (size_t*)malloc(((size_t)h.n_docs + 1) * sizeof(size_t));
    for (uint32_t d = 0; d < h.n_docs; d++) {
        const uint8_t* nl = memchr(index + o, '\n', index_len - o);
        names[d] = (const char*)index + o; nlen[d] = (size_t)(nl - (index + o)); o += nlen[d] + 1;
    }
    sbuf out = {0}; sb_put(&out, "", 0);
    sbuf seq = {0}; sb_put(&seq, "", 0);
    sbuf hdr = {0}; sb_put(&hdr, "", 0);
    int rc = 0; size_t p = 0;
    while (p < fasta_len && rc == 0) {
        const char* nl = memchr(fasta + p, '\n', fasta_len - p);
        size_t ll = nl ? (size_t)(nl - (fasta + p)) : fasta_len - p;
        const char* line = fasta + p;
        p += ll + (nl ? 1 : 0);
        if (ll == 0) continue;
        if (line[0] == '>' || line[0] == ';') {
            rc = flush_record(&out, index, &ai, names, nlen, hdr.p, hdr.n, seq.p, seq.n, threshold, num_results, err, errcap);
            hdr.n = 0; sb_put(&hdr, "*", 1); sb_put(&hdr, line + 1, ll - 1);
            seq.n = 0; seq.p[0] = 0;
        } else {
            sb_put(&seq, line, ll);
        }
    }
    if (rc == 0)
        rc = flush_record(&out, index, &ai, names, nlen, hdr.p, hdr.n, seq.p, seq.n, threshold, num_results, err, errcap);
    free(seq.p); free(hdr.p); free(names); free(nlen);
    if (rc) { free(out.p); return NULL; }
    if (out_len) *out_len = out.n;
    return out.p;
}

/* ------------------------------------------------- synthetic matrix spec  */
/* The build's own generator (Bernoulli(0.25) bits from a
 * counter-based PRNG keyed by (seed, batch, row, dword)).  The device
 * generator in phylign_amd/csrc implements the same function independently;
 * this copy lets the checker reproduce any row of a matrix that is far too
 * large to hold on the host. */
uint64_t orc_splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}
void orc_synth_row(uint64_t seed, uint32_t batch, uint64_t row, uint32_t n_docs, uint8_t* out) {
    uint64_t rb = ((uint64_t)n_docs + 7) / 8;
    uint64_t kb = orc_splitmix64(seed ^ ((uint64_t)batch * 0xD1B54A32D192ED03ULL));
    uint64_t kr = orc_splitmix64(kb + row);
    for (uint64_t j = 0; j * 4 < rb; j++) {
        uint64_t u = orc_splitmix64(kr + j);
        uint32_t w = (uint32_t)u & (uint32_t)(u >> 32);       /* P(bit)=1/4 */
        for (int b = 0; b < 4; b++) {
            uint64_t byte = j * 4 + (uint64_t)b;
            if (byte >= rb) break;
            uint8_t v = (uint8_t)(w >> (8 * b));
            uint64_t first_doc = byte * 8;
            if (first_doc + 8 > n_docs) v &= (uint8_t)((1u << (n_docs - first_doc)) - 1u);
            out[byte] = v;
        }
    }
}

typedef struct { uint64_t seed; uint32_t batch, n_docs; uint64_t r0, r1, stride; uint8_t* out; } fill_arg;
static void* fill_worker(void* vp) {
    fill_arg* a = (fill_arg*)vp;
    for (uint64_t r = a->r0; r < a->r1; r++) orc_synth_row(a->seed, a->batch, r, a->n_docs, a->out + r * a->stride);
    return NULL;
}
void orc_synth_fill(uint64_t seed, uint32_t batch, uint64_t n_rows, uint32_t n_docs,
                    uint64_t stride, uint8_t* out, int threads) {
    if (threads < 1) threads = 1;
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
    fill_arg* args = (fill_arg*)calloc((size_t)threads, sizeof(fill_arg));
    for (int t = 0; t < threads; t++) {
        args[t] = (fill_arg){seed, batch, n_docs, n_rows * (uint64_t)t / (uint64_t)threads,
                             n_rows * (uint64_t)(t + 1) / (uint64_t)threads, stride, out};
        pthread_create(&th[t], NULL, fill_worker, &args[t]);
    }
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    free(args); free(th);
}

/* ---------------------------------------------------------- CPU baseline  */
/* COBS-style inner loop (upstream classic_search.cpp compute_counts /
 * aggregate_rows / add_rows): gather the term rows, AND across hash functions,
 * add each byte through a 256-entry byte -> 8 x u16 expansion table with SSE2
 * adds, then threshold.  Queries are distributed over `threads` pthreads
 * (COBS -T splits document columns instead; same total work).  This is the
 * "port" CPU baseline bench.py reports; it is never the thing shipped. */
static __m128i g_expand[256];
static pthread_once_t g_expand_once = PTHREAD_ONCE_INIT;
static void init_expand(void) {
    for (int v = 0; v < 256; v++) {
        uint16_t e[8]; for (int b = 0; b < 8; b++) e[b] = (uint16_t)((v >> b) & 1);
        g_expand[v] = _mm_loadu_si128((const __m128i*)e);
    }
}
typedef struct {
    const uint8_t* matrix; uint64_t stride; const orc_header_t* h; const char* seqs;
    size_t qlen, q0, q1; double threshold; uint64_t hits;
} bl_arg;
static void* bl_worker(void* vp) {
    bl_arg* a = (bl_arg*)vp; const orc_header_t* h = a->h;
    size_t nt = a->qlen - h->term_size + 1;
    uint64_t* hs = (uint64_t*)malloc(nt * h->num_hashes * 8);
    size_t rb = (size_t)h->row_bytes;
    __m128i* cnt = (__m128i*)aligned_alloc(16, (rb ? rb : 1) * 16);
    uint8_t* acc = (uint8_t*)malloc(rb ? rb : 1);
    uint32_t T = orc_threshold(a->threshold, nt);
    for (size_t q = a->q0; q < a->q1; q++) {
        if (orc_create_hashes(a->seqs + q * a->qlen, a->qlen, h->term_size, h->canonicalize, h->num_hashes, hs)) continue;
        memset(cnt, 0, rb * 16);
        for (size_t i = 0; i < nt; i++) {
            const uint8_t* r0 = a->matrix + (hs[i * h->num_hashes] % h->signature_size) * a->stride;
            const uint8_t* src = r0;
            if (h->num_hashes > 1) {
                memcpy(acc, r0, rb);
                for (uint64_t j = 1; j < h->num_hashes; j++) {
                    const uint8_t* rj = a->matrix + (hs[i * h->num_hashes + j] % h->signature_size) * a->stride;
                    for (size_t b = 0; b < rb; b++) acc[b] &= rj[b];
                }
                src = acc;
            }
            for (size_t b = 0; b < rb; b++) cnt[b] = _mm_add_epi16(cnt[b], g_expand[src[b]]);
        }
        const uint16_t* c16 = (const uint16_t*)cnt;
        for (uint32_t d = 0; d < h->n_docs; d++) a->hits += (c16[d] >= T);
    }
    free(acc); free(cnt); free(hs);
    return NULL;
}
uint64_t orc_baseline_run(const uint8_t* matrix, uint64_t stride, const orc_header_t* h,
                          const char* seqs, size_t qlen, size_t n_queries, double threshold, int threads) {
    pthread_once(&g_expand_once, init_expand);
    if (threads < 1) threads = 1;
    if (qlen < h->term_size || qlen - h->term_size + 1 > 65535) return 0;
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
    bl_arg* args = (bl_arg*)calloc((size_t)threads, sizeof(bl_arg));
    for (int t = 0; t < threads; t++) {
        args[t] = (bl_arg){matrix, stride, h, seqs, qlen,
                           n_queries * (size_t)t / (size_t)threads, n_queries * (size_t)(t + 1) / (size_t)threads,
                           threshold, 0};
        pthread_create(&th[t], NULL, bl_worker, &args[t]);
    }
    uint64_t hits = 0;
    for (int t = 0; t < threads; t++) { pthread_join(th[t], NULL); hits += args[t].hits; }
    free(args); free(th);
    return hits;
}

/* The same work partitioned the way `cobs query -T` partitions it (upstream classic_search.cpp: the
 * term hashes of a query are computed once, then threads take COLUMN SLABS of the rows and count
 * every term's slab into the slab's scores): phase 1 maps every k-mer of the sample to its row
 * (threads split the queries), phase 2 hands out (column slab of `slab_bytes` row bytes, chunk of
 * queries) units -- a narrow row is a single slab, so the query chunks keep all threads busy where
 * cobs would leave T - 1 of them idle.  Results equal orc_baseline_run's. */
typedef struct {
    const uint8_t* matrix; uint64_t stride; const orc_header_t* h; const char* seqs;
    size_t qlen, nq, nt, slab_bytes, n_slabs, q_chunk, n_units; double threshold;
    uint64_t* rows; volatile long* next; uint64_t hits; int phase; size_t q0, q1;
} sl_arg;
static void* sl_worker(void* vp) {
    sl_arg* a = (sl_arg*)vp; const orc_header_t* h = a->h;
    if (a->phase == 1) {
        uint64_t* hs = (uint64_t*)malloc(a->nt * h->num_hashes * 8);
        for (size_t q = a->q0; q < a->q1; q++) {
            if (orc_create_hashes(a->seqs + q * a->qlen, a->qlen, h->term_size, h->canonicalize, h->num_hashes, hs)) continue;
            for (size_t i = 0; i < a->nt * h->num_hashes; i++) a->rows[q * a->nt * h->num_hashes + i] = hs[i] % h->signature_size;
        }
        free(hs);
        return NULL;
    }
    const size_t rb = (size_t)h->row_bytes, nh = (size_t)h->num_hashes;
    __m128i* cnt = (__m128i*)aligned_alloc(16, a->slab_bytes * 16);
    uint8_t* acc = (uint8_t*)malloc(a->slab_bytes);
    const uint32_t T = orc_threshold(a->threshold, a->nt);
    for (;;) {
        const long u = __sync_fetch_and_add(a->next, 1);
        if ((size_t)u >= a->n_units) break;
        const size_t slab = (size_t)u % a->n_slabs, chunk = (size_t)u / a->n_slabs;
        const size_t b0 = slab * a->slab_bytes, b1 = b0 + a->slab_bytes < rb ? b0 + a->slab_bytes : rb, w = b1 - b0;
        const size_t qa = chunk * a->q_chunk, qb = qa + a->q_chunk < a->nq ? qa + a->q_chunk : a->nq;
        for (size_t q = qa; q < qb; q++) {
            memset(cnt, 0, w * 16);
            const uint64_t* rq = a->rows + q * a->nt * nh;
            for (size_t i = 0; i < a->nt; i++) {
                const uint8_t* src = a->matrix + rq[i * nh] * a->stride + b0;
                if (nh > 1) {
                    memcpy(acc, src, w);
                    for (size_t j = 1; j < nh; j++) {
                        const uint8_t* rj = a->matrix + rq[i * nh + j] * a->stride + b0;
                        for (size_t b = 0; b < w; b++) acc[b] &= rj[b];
                    }
                    src = acc;
                }
                for (size_t b = 0; b < w; b++) cnt[b] = _mm_add_epi16(cnt[b], g_expand[src[b]]);
            }
            const uint16_t* c16 = (const uint16_t*)cnt;
            const size_t d1 = b1 * 8 < h->n_docs ? b1 * 8 : h->n_docs;
            for (size_t d = b0 * 8; d < d1; d++) a->hits += (c16[d - b0 * 8] >= T);
        }
    }
    free(acc); free(cnt);
    return NULL;
}
uint64_t orc_baseline_run_slabs(const uint8_t* matrix, uint64_t stride, const orc_header_t* h,
                                const char* seqs, size_t qlen, size_t n_queries, double threshold,
                                int threads, size_t slab_bytes) {
    pthread_once(&g_expand_once, init_expand);
    if (threads < 1) threads = 1;
    if (slab_bytes < 1) slab_bytes = 64;
    if (qlen < h->term_size || qlen - h->term_size + 1 > 65535 || n_queries == 0) return 0;
    const size_t nt = qlen - h->term_size + 1;
    uint64_t* rows = (uint64_t*)malloc(n_queries * nt * h->num_hashes * 8);
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
    sl_arg* args = (sl_arg*)calloc((size_t)threads, sizeof(sl_arg));
    volatile long next = 0;
    const size_t n_slabs = ((size_t)h->row_bytes + slab_bytes - 1) / slab_bytes;
    size_t chunks = ((size_t)threads * 4 + n_slabs - 1) / n_slabs;          /* >= 4 units per thread */
    if (chunks > n_queries) chunks = n_queries;
    const size_t q_chunk = (n_queries + chunks - 1) / chunks;
    chunks = (n_queries + q_chunk - 1) / q_chunk;
    for (int phase = 1; phase <= 2; phase++) {
        for (int t = 0; t < threads; t++) {
            args[t] = (sl_arg){matrix, stride, h, seqs, qlen, n_queries, nt, slab_bytes, n_slabs, q_chunk, n_slabs * chunks,
                               threshold, rows, &next, 0, phase,
                               n_queries * (size_t)t / (size_t)threads, n_queries * (size_t)(t + 1) / (size_t)threads};
            pthread_create(&th[t], NULL, sl_worker, &args[t]);
        }
        for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    }
    uint64_t hits = 0;
    for (int t = 0; t < threads; t++) hits += args[t].hits;
    free(args); free(th); free(rows);
    return hits;
}

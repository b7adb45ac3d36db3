/* cobs_oracle.h -- CPU restatement of the `cobs query` classic-index search.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under phylign_amd/ may include, link or
 * call this.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg use it, and only as the checker / reported CPU baseline.
 *
 * PARITY STATUS: "parity unpinned" against bioconda cobs=0.2.1.
 * The algorithm lives in a third-party dependency that is absent from
 * /root/reference (pinned at envs/cobs.yaml:5, `cobs=0.2.1`), so this file
 * restates COBS's published algorithm and anchors it on the reference's own
 * call sites and output-grammar witnesses:
 *   call sites        scripts/run_cobs_streaming.sh:24-29, Snakefile:419-424, Snakefile:476-481
 *   output grammar    scripts/postprocess_cobs.py:10-39, scripts/filter_queries.py:51-65,
 *                     scripts/deprec/translate_cobs_matches.py:20-29
 *   input contract    Snakefile:314-333 (upper-case, single line, non-ACGT -> A)
 * What IS pinned: XXH64 against 2 400 known-answer vectors generated with
 * python-xxhash 3.8.1 / libxxhash 0.8.2 (tests/golden/xxh64_kat.tsv).
 */
#ifndef COBS_ORACLE_H
#define COBS_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    uint32_t version;
    uint32_t term_size;
    uint8_t  canonicalize;
    uint64_t signature_size;
    uint64_t num_hashes;
    uint32_t n_docs;
    uint64_t row_bytes;     /* ceil(n_docs/8) */
    size_t   names_off;     /* offset of the first document name */
    size_t   data_off;      /* offset of the first matrix byte */
    int      layout;        /* which field order validated (see .c) */
} orc_header_t;

typedef struct { uint32_t doc; uint32_t score; } orc_hit_t;

uint64_t orc_xxh64(const void* data, size_t len, uint64_t seed);
int  orc_canonicalize(const char* kmer, size_t k, char* out);
uint32_t orc_threshold(double threshold, uint64_t num_terms);
void orc_set_rules(int threshold_rule, int tie_desc);   /* the two unpinned rules: see cobs_oracle.c */

int  orc_header_parse(const uint8_t* buf, size_t len, orc_header_t* h);
/* returns malloc'd buffer holding a complete classic index; matrix zeroed */
uint8_t* orc_index_alloc(uint32_t term_size, uint8_t canon, uint64_t sig_size,
                         uint64_t num_hashes, uint32_t n_docs,
                         const char* const* names, size_t* total_len,
                         size_t* data_off);

/* hashes[i*num_hashes + j] for the num_terms = len-k+1 terms of seq */
int  orc_create_hashes(const char* seq, size_t len, uint32_t k, int canon,
                       uint64_t num_hashes, uint64_t* hashes);

/* scores[d], d < n_docs; matrix rows are `stride` bytes apart (>= row_bytes) */
int  orc_scores(const uint8_t* matrix, uint64_t stride, const orc_header_t* h,
                const char* seq, size_t len, uint32_t* scores);

/* threshold + order; returns number of hits, fills hits[] (capacity n_docs).
 * Order: score descending, then document index ascending (a total order). */
size_t orc_select(const uint32_t* scores, uint32_t n_docs, uint64_t num_terms,
                  double threshold, size_t num_results, orc_hit_t* hits);

/* ---- compact index ("COMPACT_INDEX"): a page-aligned concatenation of classic
 * sub-indexes, each page_size bytes (= page_size*8 documents) wide with its own
 * (signature_size, num_hashes).  Phylign does not use this format; layout
 * restated from upstream cobs/file/compact_index_header.{hpp,cpp}, unpinned. */
typedef struct {
    uint32_t term_size;
    uint8_t  canonicalize;
    uint32_t n_parts;
    uint32_t n_docs;
    uint64_t page_size;
    size_t   params_off;    /* n_parts x {u64 signature_size, u64 num_hashes} */
    size_t   names_off;
    size_t   data_off;      /* first byte of part 0; part i follows part i-1 */
} orc_compact_t;
int  orc_compact_parse(const uint8_t* buf, size_t len, orc_compact_t* c);
uint8_t* orc_compact_alloc(uint32_t term_size, uint8_t canon, uint64_t page_size, uint32_t n_parts,
                           const uint64_t* sig_sizes, const uint64_t* num_hashes, uint32_t n_docs,
                           const char* const* names, size_t* total_len, size_t* data_off);
int  orc_scores_compact(const uint8_t* index, const orc_compact_t* c, const char* seq, size_t len,
                        uint32_t* scores);

/* whole `cobs query -i index -f fasta -t threshold` restatement: returns a
 * malloc'd NUL-terminated text (COBS stdout) or NULL (error text in err). */
char* orc_query_file(const uint8_t* index, size_t index_len,
                     const char* fasta, size_t fasta_len,
                     double threshold, size_t num_results,
                     size_t* out_len, char* err, size_t errcap);

/* ---- synthetic 661k-shaped matrix (the build's own generator spec) ---- */
uint64_t orc_splitmix64(uint64_t x);
/* fills one logical COBS row (row_bytes bytes) of synthetic batch `batch` */
void orc_synth_row(uint64_t seed, uint32_t batch, uint64_t row, uint32_t n_docs,
                   uint8_t* out);

/* fills n_rows rows (stride bytes apart, zero padded) with `threads` threads */
void orc_synth_fill(uint64_t seed, uint32_t batch, uint64_t n_rows, uint32_t n_docs,
                    uint64_t stride, uint8_t* out, int threads);

/* ---- CPU baseline (timed by bench.py): COBS-style expansion-table adds ----
 * Scores `n_queries` equal-length queries against a resident matrix using
 * `threads` threads; returns total hits (to keep the work observable). */
uint64_t orc_baseline_run(const uint8_t* matrix, uint64_t stride,
                          const orc_header_t* h, const char* seqs,
                          size_t qlen, size_t n_queries, double threshold,
                          int threads);
/* the same search with the work split into column slabs like `cobs query -T` (hashes once per query,
 * threads count (slab of slab_bytes row bytes, chunk of queries) units); returns the same count */
uint64_t orc_baseline_run_slabs(const uint8_t* matrix, uint64_t stride, const orc_header_t* h,
                                const char* seqs, size_t qlen, size_t n_queries, double threshold,
                                int threads, size_t slab_bytes);
#ifdef __cplusplus
}
#endif
#endif

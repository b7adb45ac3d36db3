/* phylign_match_bench.h -- C ABI of libphylign_bench.so: measurement and test
 * aids for libphylign_match.so.  NOT part of the drop-in boundary: nothing the
 * reference's 03_match path does is replaced by these.  They exist because the
 * 661k indexes (Zenodo) are not available to the benchmark, so 661k-SHAPED
 * indexes are generated in HBM (SURVEY.md 8d "Synthetic index"), true positives
 * are planted into them, and the memory system's ceiling for the scan's access
 * pattern is probed.  bench.py, tools/ and tests/ load this library; the
 * product (scripts/, phylign_amd.cobs_query, match_stage on real files,
 * server) never does.
 *
 * The library has its own kernels and reaches the product only through its C
 * ABI: pm_index_create + pm_index_matrix_device (a zeroed resident matrix and
 * its device address), pm_hash_terms, pm_queries_count / _terms.
 * Every function returns 0 or a negative PM_E* code; pm_bench_last_error() has
 * the text (thread-local). */
#ifndef PHYLIGN_MATCH_BENCH_H
#define PHYLIGN_MATCH_BENCH_H
#include "phylign_match.h"
#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

const char* pm_bench_last_error(void);

/* 661k-shaped synthetic index generated in HBM: bits i.i.d. Bernoulli(1/4) from a counter-based
 * generator (splitmix64 of seed, batch, row, dword), canonicalize = 1, names "%05x_SYN%03uD%07u"
 * (a pseudo-random sorting prefix and one underscore, like "<rnd>_<accession>"); header_only != 0
 * creates just the names table. */
int pm_bench_index_synth(uint32_t batch_id, uint32_t n_docs, uint64_t signature_size,
                         uint32_t num_hashes, uint32_t term_size, uint64_t seed,
                         int layout, int header_only, pm_index_t** out);
/* sets bit (rows[i], docs[i]) for i < n: planted hits for parity runs */
int pm_bench_index_plant(pm_index_t* idx, const uint64_t* rows, const uint32_t* docs, size_t n);

/* the hashes of a query set on the device in the aids' own dense layout (one pm_hash_terms call),
 * shared by the plantings of many batches; freed with pm_bench_hashes_free */
typedef struct pm_bench_hashes pm_bench_hashes_t;
int  pm_bench_hashes_create(pm_queries_t* q, int canonicalize, uint32_t num_hashes, pm_bench_hashes_t** out);
void pm_bench_hashes_free(pm_bench_hashes_t* h);
/* Synthetic "related batch" content.  Makes this index the HOME batch of queries q_first,
 * q_first + q_step, ...: about half of its 32-document clusters match each of those queries at a
 * k-mer fraction between 0.6 and 1.0 -- the shape real phylogenetic batches have for reads of their
 * own species (many documents near the threshold, long hit lists). */
int pm_bench_index_plant_cluster(pm_index_t* idx, const pm_bench_hashes_t* h, uint32_t q_first, uint32_t q_step, uint64_t seed);

/* For cold-path timings only: overwrites the resident matrix with COMPRESSIBLE content (along a row, a document repeats its
 * neighbour's bit except with probability 2^-flip_log2; 7 gives an xz ratio near 0.1), so that the .xz of a saved index shrinks the way the real 661k indexes do
 * (Bernoulli(1/4) bits barely compress).  Nothing meaningful can be searched on such a matrix. */
int pm_bench_index_correlate(pm_index_t* idx, uint64_t seed, uint32_t flip_log2);

/* SURVEY.md 8d, many-queries regime: the number of DISTINCT signature rows the query set's k-mers map to in this index
 * (unique_rows x row_bytes is what a scan with perfect row reuse would have to read; reported beside the algorithmic bytes) */
int pm_bench_unique_rows(const pm_index_t* idx, const pm_bench_hashes_t* h, uint64_t* unique_rows);

/* writes a resident classic index back as a .cobs_classic file (for the cold / cached / resident timings of the stage
 * on 661k-shaped files: tools/e2e_cold_warm.py) */
int pm_bench_index_save(const pm_index_t* idx, const char* path);

/* Times a pure random-row gather over this index with k_scan's access pattern (n_groups
 * row-cooperating lane groups x lookups_per_group rows each, no counting); *ms = hipEvent time,
 * *bytes = rows fetched x row_bytes: the memory-system ceiling the scan kernel is compared with.
 * mode 0 = uniform rows, 1 = ascending stratified rows, 2 = ascending order statistics; flavor 0 =
 * plain loads, 1 ... 5 = cache-policy bits (nt, sc1, sc0 sc1, sc0 sc1 nt, sc0); unroll = 4, 8 or 16
 * gathers in flight per lane (0 = 8). */
int pm_bench_probe_gather(const pm_index_t* idx, uint64_t n_groups, uint64_t lookups_per_group,
                          int mode, int flavor, int unroll, double* ms, uint64_t* bytes);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif

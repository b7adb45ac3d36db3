/* phylign_match.h -- C ABI of libphylign_match.so, the MI355X-native COBS
 * classic-index matching stage for Phylign's intermediate/03_match.
 *
 * This is the drop-in boundary.  The reference binds to its
 * engine through a process boundary: `cobs query --load-complete -t T -T N
 * -i INDEX [--index-sizes BYTES] -f QUERY.fa` (scripts/run_cobs_streaming.sh:24-29,
 * Snakefile:419-424, Snakefile:476-481) followed by `postprocess_cobs.py -n N`
 * (Snakefile:425, :467).  The entry points below are what a Python/ctypes (or
 * cgo / JNI) binding for that path calls instead; phylign_amd/cobs_query.py is
 * that binding and INTEGRATION.md shows the Snakefile-side change.
 *
 * Conventions: every function returns 0 on success or a negative PM_E* code;
 * pm_last_error() holds the message (thread-local).  No C++ exceptions cross
 * the boundary.  One process drives one GPU.  There is NO CPU fallback: without
 * a visible gfx950 device pm_init() fails and every compute entry point
 * returns PM_ENODEV.
 *
 * Measurement and test aids (synthetic 661k-shaped indexes, planted hits, the
 * gather probe) are NOT here: include/phylign_match_bench.h, libphylign_bench.so.
 */
#ifndef PHYLIGN_MATCH_H
#define PHYLIGN_MATCH_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with hidden visibility: exactly the entry points declared here are exported */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define PM_OK        0
#define PM_EINVAL   -1   /* bad argument / malformed input */
#define PM_ENODEV   -2   /* no usable GPU or pm_init() not called */
#define PM_ENOMEM   -3   /* host or device allocation failed */
#define PM_EIO      -4   /* read error / short index stream */
#define PM_EFORMAT  -5   /* not a COBS classic index */
#define PM_EQUERY   -6   /* query shorter than k, or non-ACGT base */
#define PM_EHIP     -7   /* HIP runtime error (message has the detail) */
#define PM_ERANGE   -8   /* size outside what this build supports */

typedef struct pm_index   pm_index_t;    /* one phylogenetic batch index, resident in HBM */
typedef struct pm_queries pm_queries_t;  /* a parsed query FASTA, resident in HBM */
typedef struct pm_result  pm_result_t;   /* hits of one pm_search call */
typedef struct pm_merge   pm_merge_t;    /* 04_filter state: best matches per query across batches */
typedef struct pm_slice   pm_slice_t;    /* the records of one index of a search on the host (pooled pinned memory) */

/* Row layout policy in HBM (DESIGN.md "Data layout"). */
#define PM_LAYOUT_AUTO     0  /* line-aligned stride if it fits, else compact */
#define PM_LAYOUT_COMPACT  1  /* stride = row_bytes rounded up to 16 */
#define PM_LAYOUT_ALIGNED  2  /* stride = pow2 (<=128 B) or multiple of 128 B */

typedef struct {
    uint32_t term_size;        /* k (31 for the 661k indexes) */
    uint32_t canonicalize;     /* 0/1 */
    uint64_t signature_size;   /* number of rows S */
    uint32_t num_hashes;
    uint32_t n_docs;           /* D */
    uint64_t row_bytes;        /* ceil(D/8): the algorithmic bytes of one row */
    uint64_t stride;           /* bytes between rows in HBM (>= row_bytes) */
    uint64_t device_bytes;     /* HBM held by the matrix */
    uint32_t header_layout;    /* which header field order validated (0/1) */
    uint32_t has_matrix;       /* 0 for header-only handles */
    uint32_t n_parts;          /* 0 for a classic index; sub-indexes of a compact index */
    uint32_t reserved;
    uint64_t page_size;        /* compact: bytes per sub-index row (row_bytes reports it too) */
} pm_index_info_t;

/* One hit record, 16 bytes; identical in HBM, on the wire (RCCL) and on host.
 * A record with doc == PM_DOC_COUNT is not a hit but a COUNT RECORD: `score` holds
 * the number of documents of (query, slot) that passed the threshold -- the N that
 * cobs prints in the "*header\tN" line, which postprocess_cobs.py keeps even when
 * it drops lines.
 *   In HBM (pm_result_hits_device) the scan kernel writes one RUN per (query, slot
 * [, column slab / sub-index]) that has hits: a count record followed by the hits in
 * cobs' line order (score descending, ties by ascending document index); runs are
 * in arbitrary order.
 *   On the host (pm_result_hits_into / _host) runs are ordered by (slot, query) and
 * a count record is kept only where the hit list was cut to the n best on the GPU
 * (its count then differs from the number of hit records that follow). */
#define PM_DOC_COUNT 0xFFFFFFFFu
typedef struct {
    uint32_t query;   /* index of the FASTA record inside the pm_queries_t */
    uint32_t doc;     /* document (column) index inside the batch index */
    uint32_t score;   /* number of matched k-mers */
    uint32_t slot;    /* position of the index in the pm_search() array (+ slot_base) */
} pm_hit_t;

typedef struct {
    uint64_t n_queries, n_terms;     /* FASTA records / k-mers in the query set */
    uint64_t n_hits;                 /* hit records over all slots (= n_records - n_runs) */
    uint64_t algorithmic_bytes;      /* sum_slots n_terms * num_hashes * row_bytes */
    double   ms_total;               /* hipEvent time: hash + scan kernels */
    double   ms_hash;                /* canonicalise + XXH64 kernel */
    double   ms_scan;                /* sum of scan-kernel launch durations */
    uint32_t n_scan_launches;
    uint32_t reserved;
    uint64_t n_records;              /* records in HBM: hits + one count record per run */
    uint64_t n_runs;                 /* (query, slot[, slab]) groups with at least one hit */
    uint64_t fetched_bytes;          /* option "count_fetched": algorithmic bytes of the row chunks really gathered (0 = not counted) */
} pm_stats_t;

typedef struct {
    uint32_t lanes_per_row;      /* template parameter G of k_scan; 0 = the mixed-width launch of the narrow batches */
    uint32_t planes;             /* template parameter P of k_scan */
    uint32_t num_hashes;
    uint32_t n_batches;          /* batch indexes covered by the launch */
    uint64_t n_queries;
    uint64_t algorithmic_bytes;  /* sum over its batches and k-mers of num_hashes * row_bytes */
    double   ms;                 /* hipEvent duration on the launch stream */
    uint64_t fetched_bytes;      /* option "count_fetched": the part of algorithmic_bytes really gathered */
    uint32_t wide_query;         /* 1: several lane groups of a workgroup shared each query (few, long queries) */
    uint32_t reserved;
} pm_launch_t;

/* ---- runtime ---------------------------------------------------------- */
int  pm_init(int device);                 /* bind this process to GPU `device` */
void pm_shutdown(void);
const char* pm_last_error(void);
int  pm_device_info(char* name, size_t cap, uint64_t* hbm_total, uint64_t* hbm_free, int* n_cus);
void pm_free(void* p);                    /* frees buffers documented as caller-freed */
/* Tuning / measurement switches.  "threshold_bound" (default 1): the scan stops
 * fetching a signature line once none of its documents can reach the minimum
 * score any more (count so far + k-mers left < minimum); hits and scores are
 * identical either way, 0 makes the scan fetch every row like cobs does.
 * "count_fetched" (default 0): the scan also counts the algorithmic bytes of the
 * row chunks it really gathered (pm_stats_t / pm_launch_t .fetched_bytes).
 * "wide_query" (default 0 = automatic): query classes of 128+ k-mers with too few queries
 * to fill the GPU are scanned with several lane groups per query (partial counts added
 * through LDS; the threshold bound is off in that form); 1 = always, 2 = never.
 * "wide_query_split" (default 0 = automatic): in that form, how many workgroups share the steps of one query
 * (a dozen chromosome-sized queries would otherwise leave most of the chip idle); 1 = never, n = force n.
 * "single_launch" (default 0): rows of every width up to 1024 B share one launch.
 * "merge_counting_sort" (default 1): hit lists written as several runs (rows wider than 1024 B, compact sub-indexes) are
 * merged by a counting sort on the score; 0 = the general pairwise form for every group (same result).
 * "release_query_pool" (any value; an action): the device buffers of query sets that gave their HBM copies back
 * (pm_queries_release_device) wait in a small pool for the next set -- freeing them on the spot would wait for every
 * search queued behind -- and this call really frees them (it waits for the device: call it between jobs).
 * "release_pools" (any value; an action): the same for every idle device buffer the library keeps -- the query-set pool
 * and the hit buffers of finished searches.  (The library does this by itself before it reports PM_ENOMEM for a
 * signature matrix, a query set or a hit buffer.)
 * "cobs_threshold_rule" (default 0) / "cobs_tie_order" (default 0): the two rules of `cobs query` that no file of the
 * reference pins -- how -t becomes a minimum score: 0 = ceil(t x k-mers), 1 = floor, 2 = round half up; how documents of
 * equal score are listed: 0 = ascending document index, 1 = descending.  The defaults are upstream's as recalled
 * (DESIGN.md section 6); tools/pin_against_cobs.sh checks them against a real cobs 0.2.1.  Neither rule can be changed
 * while a pm_result_t is alive (PM_EINVAL): the device orders a search's records under the rule in force when it was
 * queued, the host orders and formats them under the current one. */
int  pm_set_option(const char* name, int64_t value);
/* ceil(threshold * num_terms): the score a document must reach (cobs -t). */
uint32_t pm_threshold_terms(double threshold, uint64_t num_terms);

/* ---- index (replaces `cobs query -i`, --load-complete, --index-sizes) --- */
/* path may be a regular file or a pipe (/dev/fd/N as produced by
 * run_cobs_streaming.sh:27); size_hint = --index-sizes value or 0.  Both
 * "CLASSIC_INDEX" files (what Phylign ships, Snakefile:48) and "COMPACT_INDEX"
 * files are accepted; a compact index is held as one matrix per page column
 * and searched like several classic indexes that share one slot. */
int  pm_index_load_file(const char* path, uint64_t size_hint, int layout, pm_index_t** out);
int  pm_index_load_fd(int fd, uint64_t size_hint, int layout, pm_index_t** out);
int  pm_index_load_mem(const void* buf, size_t len, int layout, pm_index_t** out);
/* pm_index_load_fd that also keeps what it read: the decode-once cache of a compressed index (the reference's `mem-disk`
 * mode with keep_cobs_indexes: rule decompress_cobs, Snakefile:364-387, writes "<batch>.cobs_classic.tmp" and renames it).
 * Every byte of the stream goes to a file of this call's own in tee_path's directory -- an unnamed inode (O_TMPFILE) where
 * the file system has them, so that even a killed process leaves nothing behind; "<tee_path>.XXXXXX.tmp" otherwise -- which
 * becomes `tee_path` (link + rename) once the whole index is in HBM; a failed load leaves nothing behind; when another
 * process published the same index meanwhile, this call's copy is dropped.  A failed write (disk full) does not fail the
 * load: *cached (optional) says whether the file exists now.  A later pm_index_load_file(tee_path) takes the parallel pread path. */
int  pm_index_load_fd_tee(int fd, uint64_t size_hint, int layout, const char* tee_path, int* cached, pm_index_t** out);
/* header + document names only (no matrix): for the rank that formats text */
int  pm_index_load_header_mem(const void* buf, size_t len, pm_index_t** out);
/* names-only handle from `n_docs` newline-terminated names (rank 0 formats and
 * merges batches it does not hold) */
int  pm_index_from_names(const char* names, size_t len, uint32_t n_docs, uint32_t term_size, pm_index_t** out);
/* releases the HBM matrix, keeps header and names (streaming one batch after another) */
int  pm_index_drop_matrix(pm_index_t* idx);
/* An index made in place instead of read from a file: the header fields, `n_docs` newline-terminated names and a ZEROED
 * signature matrix in HBM (header_only != 0: names only).  Code that builds signatures on the device fills the matrix
 * through pm_index_matrix_device. */
int  pm_index_create(uint32_t term_size, uint32_t canonicalize, uint64_t signature_size, uint32_t num_hashes,
                     const char* names, size_t names_len, uint32_t n_docs, int layout, int header_only, pm_index_t** out);
/* the resident matrix of a classic index, for code that shares the device with the library: row r starts at
 * *dptr + r * *stride, document d is bit d % 8 of byte d / 8 (fails for header-only handles and compact indexes) */
int  pm_index_matrix_device(const pm_index_t* idx, void** dptr, uint64_t* stride);
/* copies n rows starting at row0 (row_bytes each, file layout) back to the host (checks) */
int  pm_index_read_rows(const pm_index_t* idx, uint64_t row0, uint64_t n, void* out);
int  pm_index_info(const pm_index_t* idx, pm_index_info_t* info);
/* GPU that holds the matrix (-1: header-only handle).  HIP's current device is a
 * per-thread setting; every entry point binds the calling thread to the device of
 * pm_init() first, so indexes loaded from worker threads land on that GPU too. */
int  pm_index_device(const pm_index_t* idx, int* device);
const char* pm_index_doc_name(const pm_index_t* idx, uint32_t doc, size_t* len);
/* copies the row_bytes logical bytes of one row back to the host (checks) */
int  pm_index_read_row(const pm_index_t* idx, uint64_t row, void* out);
void pm_index_free(pm_index_t* idx);

/* ---- queries (replaces `cobs query -f`) ---------------------------------- */
/* FASTA record rules of the cobs CLI: '>' or ';' starts a record, empty lines
 * are skipped, sequence lines are concatenated, records with an empty sequence
 * are dropped.  Fails with PM_EQUERY on a sequence shorter than term_size or
 * holding a byte outside ACGT. */
int  pm_queries_parse(const char* fasta, size_t len, uint32_t term_size, pm_queries_t** out);
/* where a prepared query file is cut into pieces of max_records records each (offsets of record starts: the lines that
 * begin with '>' or ';'); *cuts (pm_free) holds *n_cuts ascending offsets, none when the file has no more records than
 * that.  For query files with more reads than fit HBM at once (a 01_queries_merged file is one per query set:
 * Snakefile:336-352). */
int  pm_fasta_record_cuts(const char* fasta, size_t len, uint64_t max_records, uint64_t** cuts, uint64_t* n_cuts);
/* normalise != 0: `buf` is an unprocessed query file (FASTA or FASTQ, multi-line, lower case, IUPAC
 * codes, comments) and the parser applies rules fix_query + concatenate_queries itself
 * (Snakefile:314-352: `seqtk seq -A -U -C | awk gsub(/[^ACGT]/, "A")`, kseq record rules): names cut
 * at the first blank, sequences joined, upper-cased, every byte outside ACGT mapped to 'A', quality
 * dropped.  normalise == 0 is pm_queries_parse. */
int  pm_queries_parse_raw(const char* buf, size_t len, uint32_t term_size, int normalise, pm_queries_t** out);
/* the prepared single-line FASTA this query set stands for (what intermediate/01_queries_merged/
 * holds in the reference); *text malloc'd, pm_free() */
int  pm_queries_fasta(const pm_queries_t* q, char** text, size_t* len);
int  pm_queries_count(const pm_queries_t* q, uint64_t* n_queries, uint64_t* n_terms);
/* number of k-mers of record i (length - k + 1) */
int  pm_queries_terms(const pm_queries_t* q, uint64_t i, uint64_t* n_terms);
void pm_queries_free(pm_queries_t* q);
/* Frees the HBM copies of the query set (sequences, descriptors, hash buffers: about 8 bytes per k-mer and hash function)
 * and keeps the host side -- names and sequences, which results, texts and the 04_filter merge refer to.  The next search
 * uploads them again.  For a query file searched chunk after chunk (pm_fasta_record_cuts): a chunk that is not searched
 * for a while need not stay resident.  Waits only for searches that were queued with THIS set.  The buffers go to a small
 * pool inside the library that the next query set draws from (a hipFree would wait for every search queued behind);
 * pm_set_option("release_query_pool", 1) really frees them. */
int  pm_queries_release_device(pm_queries_t* q);
/* *resident: HBM bytes the set holds now; *when_searched: what it holds while searched against indexes of num_hashes
 * hash functions (either may be NULL) */
int  pm_queries_device_bytes(const pm_queries_t* q, uint32_t num_hashes, uint64_t* resident, uint64_t* when_searched);
/* runs the canonicalise+XXH64 kernel and copies hashes[term*num_hashes + j]
 * (dense, FASTA order) to the host: parity hook for SURVEY row a5 */
int  pm_hash_terms(pm_queries_t* q, int canonicalize, uint32_t num_hashes, uint64_t* out);

/* ---- search (replaces the query loop of `cobs query`) -------------------- */
/* Scores every query against every index of idx[0..n_idx) and keeps documents
 * with score >= pm_threshold_terms(threshold, terms(query)); threshold 0 keeps
 * all.  nb_best_hits > 0 additionally keeps, per (query, index), only the
 * nb_best_hits best documents plus those tied with the last of them -- the
 * selection rule of scripts/postprocess_cobs.py:31-39, applied on the GPU before
 * anything leaves HBM (indexes wider than 8192 documents are pruned when the
 * text is formatted instead).  Hits stay in HBM until asked for.  slot_base is
 * added to the slot field (global batch numbering across ranks). */
int  pm_search(pm_index_t* const* idx, size_t n_idx, pm_queries_t* q,
               double threshold, uint32_t nb_best_hits, uint32_t slot_base, pm_result_t** out);
/* The same job queued on the GPU without waiting for it: returns as soon as the
 * kernels are enqueued, so the host can order / format / gather the records of the
 * previous search meanwhile.  Several searches may be in flight (they run in order);
 * the index array may be released after the call, the indexes and the query set must
 * stay alive until the result is waited for.  Every pm_result_* getter waits itself. */
/* Threads: searches may be queued from several threads, but ONE pm_queries_t must not be searched (or hashed,
 * or freed) from two threads at the same time -- the handle caches its hashes and per-query thresholds. */
int  pm_search_async(pm_index_t* const* idx, size_t n_idx, pm_queries_t* q,
                     double threshold, uint32_t nb_best_hits, uint32_t slot_base, pm_result_t** out);
/* The same with a PART of the query set per index: a batch that is resident on several GPUs is searched by each of them
 * with a share of the queries (the static batch -> GPU map evens out the ranks' scan work that way).  In
 * every counter-width class of the query set (queries of < 8, < 128, < 1 024, < 2^16, < 2^20, < 2^24 k-mers) with n
 * queries in file order, index i sees those at positions floor(lo * n / den) ... floor(hi * n / den) - 1; den == 0 (or
 * parts == NULL): all of them.  Which queries a part holds depends on the query set alone, so the parts [0, a), [a, b),
 * ..., [z, den) of one index -- searched anywhere -- cover every query exactly once, and the union of their records is
 * the record set of the whole search. */
typedef struct { uint32_t lo, hi, den; } pm_qpart_t;
int  pm_search_async_parts(pm_index_t* const* idx, size_t n_idx, pm_queries_t* q,
                           double threshold, uint32_t nb_best_hits, uint32_t slot_base,
                           const pm_qpart_t* parts, pm_result_t** out);
int  pm_result_wait(pm_result_t* r);
int  pm_result_stats(const pm_result_t* r, pm_stats_t* st);
/* the scan-kernel launches of the search (one per row-width class x counter-width
 * class, each covering all batches of the class) with their hipEvent durations;
 * writes min(cap, *n) entries, *n = number of launches */
int  pm_result_launches(const pm_result_t* r, pm_launch_t* out, size_t cap, size_t* n);
/* raw (unordered) records in HBM, e.g. as the send buffer of the RCCL gather */
int  pm_result_hits_device(const pm_result_t* r, const void** dptr, uint64_t* n);
/* the records in HBM after the device-side ordering (runs moved into (slot, query) order
 * by k_permute_runs, count records kept only where a list was cut): the send buffer of
 * the RCCL gather.  Valid until pm_result_free. */
int  pm_result_ordered_device(pm_result_t* r, const void** dptr, uint64_t* n);
/* D2D copy of the records (ordered != 0: after the device-side ordering, else raw runs)
 * into caller-owned device memory, e.g. a torch tensor that RCCL sends */
int  pm_result_copy_hits_device(pm_result_t* r, void* dst_dptr, uint64_t capacity, int ordered);
/* D2H copy of the ordered records into caller-owned host memory (capacity in records;
 * pm_stats_t.n_records always suffices): (slot, query, score desc, doc asc); *n_out = records written */
int  pm_result_hits_into(const pm_result_t* r, pm_hit_t* out, uint64_t capacity, uint64_t* n_out);
/* records on the host ordered by (slot, query, score desc, doc asc);
 * library-owned pinned memory, valid until pm_result_free */
int  pm_result_hits_host(pm_result_t* r, const pm_hit_t** hits, uint64_t* n);
/* the ordered records of ONE index of the search (slot = its position in the idx array, 0 ... n_idx - 1) read back on
 * their own: what one 03_match file / one pm_merge_add needs (one `cobs query` job's stdout: Snakefile:463-469).  The
 * host half of a stage takes a search apart batch by batch, from several threads; the records of a batch come in a
 * pooled pinned buffer that goes back to the pool with pm_slice_free() (pinning the memory for ALL records of a
 * million-read search costs more than copying them).  Thread-safe. */
int  pm_result_slot_hits(pm_result_t* r, uint32_t slot, const pm_hit_t** hits, uint64_t* n, pm_slice_t** slice);
void pm_slice_free(pm_slice_t* slice);
void pm_result_free(pm_result_t* r);
/* orders records in place by (slot, query, score desc, doc asc): the order of
 * cobs' result lines; for records gathered from other ranks */
void pm_hits_sort(pm_hit_t* hits, uint64_t n);

/* ---- text (replaces cobs stdout and, optionally, postprocess_cobs.py) ---- */
/* Orders `hits` (any order, all slots allowed; only records with slot==slot are
 * used) the COBS way and renders "*<header>\t<N>\n" + N x "<doc>\t<score>\n"
 * for every query.  nb_best_hits < 0: plain cobs output.  nb_best_hits >= 0:
 * additionally applies scripts/postprocess_cobs.py:16-39 (-n nb_best_hits).
 * *text is malloc'd; release with pm_free(). */
int  pm_format_hits(const pm_index_t* idx, const pm_queries_t* q,
                    const pm_hit_t* hits, uint64_t n_hits, uint32_t slot,
                    int64_t nb_best_hits, char** text, size_t* len);
/* plain cobs text with at most `limit` result lines per query and the header counting what is
 * printed: `cobs query -l limit` (0 = all; Phylign never passes -l) */
int  pm_format_hits_limit(const pm_index_t* idx, const pm_queries_t* q,
                          const pm_hit_t* hits, uint64_t n_hits, uint32_t slot,
                          uint64_t limit, char** text, size_t* len);
/* The 03_match file of one batch in one call: pm_format_hits(nb_best_hits) deflated like `gzip --fast` (level 1: pm_gzip_fast's encoder; 0, 2-9:
 * zlib at that level; as consecutive gzip members, built on several threads) and written to `path` via "<path>.tmp" + rename -- what
 * `... | postprocess_cobs.py -n N | gzip --fast > intermediate/03_match/<batch>____<qfile>.gz` leaves (Snakefile:463-469).
 * *text_bytes / *gz_bytes (optional): sizes before / after compression. */
int  pm_format_hits_gz(const pm_index_t* idx, const pm_queries_t* q,
                       const pm_hit_t* hits, uint64_t n_hits, uint32_t slot, int64_t nb_best_hits,
                       const char* path, int level, uint64_t* text_bytes, uint64_t* gz_bytes);
/* the same when the query set is searched in chunks (more reads than fit HBM at once): the pieces of a batch's file are
 * written in query order into "<path>.tmp" -- piece 1 = first (creates it), 2 = middle (appends), 3 = last (appends and
 * renames to `path`), 0 = the whole file; gzip members simply follow each other. */
int  pm_format_hits_gz_piece(const pm_index_t* idx, const pm_queries_t* q,
                             const pm_hit_t* hits, uint64_t n_hits, uint32_t slot, int64_t nb_best_hits,
                             const char* path, int level, int piece, uint64_t* text_bytes, uint64_t* gz_bytes);
/* `gzip --fast` of a result text as the stage writes it (level 1 above): the text is cut at line boundaries into ~1 MiB
 * pieces, each becomes one gzip member (one deflate block with Huffman codes built from the member's counts; matches
 * are found through the line structure of cobs / post-filter output: the last line with the same reference name, the previous "*" line), built on several threads.
 * Any byte string is accepted; `gzip -dc` / xopen / Python's gzip decode the result to `text` (Snakefile:427, :468, :483
 * only ever pipe into `gzip --fast`; scripts/filter_queries.py:46 only ever inflates).  *gz is freed with pm_free(). */
int  pm_gzip_fast(const char* text, size_t len, char** gz, size_t* gz_len);
/* one-shot: what `cobs query -i INDEX -f FASTA -t T` prints */
int  pm_query_text(pm_index_t* idx, const char* fasta, size_t fasta_len,
                   double threshold, int64_t nb_best_hits, char** text, size_t* len);

/* ---- 04_filter (replaces scripts/filter_queries.py) ---- */
/* keep = -n of filter_queries.py (config nb_best_hits).  The queries handle must
 * outlive the merge. */
int  pm_merge_create(const pm_queries_t* q, uint32_t keep, pm_merge_t** out);
/* the next PIECE of the same query file (a 01_queries_merged file that is parsed and searched piece by piece): its
 * records continue the numbering.  The consumer's dict semantics hold through the whole file
 * (scripts/filter_queries.py:107-120, :178-185): a read name is one query -- printed where it first occurs, with the
 * sequence of its last occurrence and the matches of all its occurrences (the two mates of a read pair in
 * concatenated files, Snakefile:336-352).  The piece must outlive the merge. */
int  pm_merge_extend(pm_merge_t* m, const pm_queries_t* piece);
/* adds the 03_match content of one batch: records of `slot` (count records are
 * skipped), post-filtered with nb_best_hits like pm_format_hits (>= 0) or taken
 * as they are (< 0); `batch` is the batch name of the file name
 * "<batch>____<qfile>.gz" (scripts/filter_queries.py:44), the tie-break after the score.  The
 * merge copies the names it needs: the index may be freed right after the call.  Calls may come
 * from several threads (serialised inside). */
int  pm_merge_add(pm_merge_t* m, const char* batch, const pm_index_t* idx,
                  const pm_hit_t* hits, uint64_t n_hits, uint32_t slot, int64_t nb_best_hits);
/* the same for records of a search with piece `piece` of the query file (0 = the set of pm_merge_create, 1 ... the
 * pieces of pm_merge_extend in their order): their query numbers count inside that piece; piece = -1: they count
 * through the whole file (pm_merge_export's records).  A batch added once per piece is one batch. */
int  pm_merge_add_piece(pm_merge_t* m, int64_t piece, const char* batch, const pm_index_t* idx,
                        const pm_hit_t* hits, uint64_t n_hits, uint32_t slot, int64_t nb_best_hits);
/* adds the 03_match TEXT of one batch (the content of "<batch>____<qfile>.gz" after gunzip) -- the native
 * form of the reader of scripts/filter_queries.py:27-66 behind the drop-in scripts/filter_queries.py:
 * "*<qname>[ comment]\t<N>" starts a query, every other non-empty line is "<rnd>_<ref> <kmers>" (two fields,
 * exactly one '_'); a malformed line, a text without header and a query that is not in the query file fail
 * with PM_EINVAL like the reference raises (match lines ahead of the first header join the first query, as
 * they do there).  Lines end at "\n", "\r\n" or a lone "\r" (the reference reads in text mode).  Counts are kept as 32 bits:
 * a k-mer count above 2^32 - 1 (Python's int() takes any; no read has that many k-mers) counts as 2^32 - 1.
 * Safe to call while other threads add or extend the same merge. */
int  pm_merge_add_text(pm_merge_t* m, const char* batch, const char* text, size_t len);
/* What the merge holds so far, as hit records {query (numbered through the whole file), doc, score, slot = number of
 * the batch in this merge: pm_merge_batches} ordered by (slot, query, score desc, doc asc): one rank's share of the
 * 04_filter result.  The ranks of a multi-GPU stage gather these (the single RCCL gather at the end)
 * and rank 0 adds them again batch by batch (pm_merge_add_piece, piece -1, nb_best_hits < 0): the `keep` best (+ ties) of the
 * union are among the `keep` best (+ ties) of every part, so the result equals the one-process merge
 * (scripts/filter_queries.py:123-156 prunes the same way file after file).  *hits malloc'd, pm_free(). */
int  pm_merge_export(const pm_merge_t* m, pm_hit_t** hits, uint64_t* n);
/* the batch names of the merge in the order of their numbers, '\n'-terminated each; *names malloc'd, pm_free() */
int  pm_merge_batches(const pm_merge_t* m, char** names, size_t* len);
/* ">qname ref1,ref2,...\nseq\n" per query in FASTA order; *text malloc'd, pm_free() */
int  pm_merge_emit(const pm_merge_t* m, char** text, size_t* len);
/* the same text written to `path` (via "<path>.tmp" + rename), built and written on several threads;
 * *bytes (optional) = size of the file */
int  pm_merge_emit_file(const pm_merge_t* m, const char* path, uint64_t* bytes);
/* the same for a query file searched in chunks (one merge per chunk, in file order): piece 1 = first (creates
 * "<path>.tmp"), 2 = middle (appends), 3 = last (appends and renames to `path`), 0 = the whole file; *bytes = this piece */
int  pm_merge_emit_file_piece(const pm_merge_t* m, const char* path, int piece, uint64_t* bytes);
void pm_merge_free(pm_merge_t* m);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif

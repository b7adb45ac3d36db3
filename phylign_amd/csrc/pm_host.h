// pm_host.h -- shared by the host translation units of libphylign_match.so
// (pm_runtime.cpp, pm_index.cpp, pm_queries.cpp, pm_search.cpp, pm_text.cpp).
// Not part of the C ABI (that is include/phylign_match.h).
//
// The reference reaches this functionality through `cobs query ...`
// (scripts/run_cobs_streaming.sh:24-29; Snakefile:419-424, :476-481) and
// `postprocess_cobs.py -n N` (scripts/postprocess_cobs.py:21-39).
// No CPU fallback exists anywhere in these files: scoring happens only in pm_kernels.hip.
#pragma once
#include "../../include/phylign_match.h"
#include "pm_internal.h"

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <functional>
#include <condition_variable>
#include <map>
#include <mutex>
#include <string>
#include <sys/stat.h>
#include <thread>
#include <tuple>
#include <unordered_map>
#include <unistd.h>
#include <vector>

using namespace pm;

// counter-width classes: queries of < 2^3, 2^7, 2^10, 2^13, 2^16, 2^20, 2^24 k-mers get bit-sliced per-document
// counters of that many planes (one k_scan instantiation each), so short reads never pay for a
// long gene or plasmid in the same FASTA, and a query of 1 ... 7 k-mers (BASELINE configs[1]: "31-mer
// queries" = one k-mer each, hit <=> bit set) carries three planes instead of seven.  The 13-plane class is the
// gene class (data/ARGannot_r3.fa, 207 ... 3 123 k-mers per gene): it is the widest counter that
// still fits 128 VGPRs (4 waves per SIMD) without scratch; the 16-plane class above it spills 5-14 registers.
constexpr int kNumClasses = 7;
static const int kPlaneClass[kNumClasses] = {3, 7, 10, 13, 16, 20, 24};

// ------------------------------------------------------------------ errors
// sets the calling thread's pm_last_error() text, returns `code`
int fail(int code, const char* fmt, ...);
#define HIPCHK(expr)                                                                   \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess)                                                          \
            return fail(e_ == hipErrorOutOfMemory ? PM_ENOMEM : PM_EHIP, "%s: %s (%s:%d)", \
                        #expr, hipGetErrorString(e_), __FILE__, __LINE__);             \
    } while (0)

// No C++ exception crosses the C ABI: every `extern "C" int` entry point is a function-try-block that ends in
// PM_GUARD_END, which turns whatever was thrown (std::bad_alloc from a vector or string of a huge input, on the calling
// thread or on a pool thread: parallel_for hands it over) into an error code + pm_last_error().
int on_exception();          // call inside a catch block
#define PM_GUARD_END catch (...) { return on_exception(); }

struct HitBuf { uint4* p; uint64_t cap; };
struct PinBuf { void* p; size_t bytes; };
// Per-search scratch that must stay untouched while the search is in flight (several
// searches may be queued back to back: pm_search_async): record counters with their pinned
// mirror, batch descriptors, timing events.  Pooled in the context, grow-only.
struct Workspace {
    unsigned long long* d_cnt = nullptr;      // [0] records written, [1] runs
    unsigned long long* h_cnt = nullptr;      // pinned, device-mapped mirror: written by k_publish behind the scans
    unsigned long long* h_cnt_dev = nullptr;  // its device address
    // batch descriptors: 5 slices of desc_cap entries (base + one per query counter-width class, which
    // carries the block ranges of a mixed-width launch); `uploaded` is what the device copy holds, so
    // a step loop over the same indexes uploads nothing (and puts no DMA on the compute stream)
    BatchDesc* d_desc = nullptr; BatchDesc* h_desc = nullptr; size_t desc_cap = 0;
    std::vector<BatchDesc> uploaded;
    std::vector<hipEvent_t> events;
    // wide-query form shared across workgroups: partial count planes + one arrival counter per workgroup (grow-only)
    uint4* d_split = nullptr; size_t split_bytes = 0;
    uint32_t* d_split_cnt = nullptr; size_t split_cnt_n = 0;
    hipEvent_t done = nullptr;                // recorded behind the counter read-back
    bool busy = false;
};
constexpr uint32_t kFetchShards = 4096;       // counters of the "count_fetched" measurement option
struct Ctx {
    bool ready = false;
    int device = -1;
    hipStream_t stream = nullptr;             // hash + scan kernels
    hipStream_t copy_stream = nullptr;        // index upload (H2D + re-stride)
    hipStream_t d2h_stream = nullptr;         // hit records to the host: never queues behind later kernels
    std::vector<Workspace*> ws;
    std::vector<HitBuf> free_hits;
    std::vector<PinBuf> free_pinned;
    unsigned long long* d_fetch = nullptr;
    uint64_t hit_hint = 0;                    // most records one search produced so far: sizes the next hit buffer
};
extern Ctx g_ctx;
// HIP's current device is a per-thread setting that starts at 0: every entry point
// that allocates, copies or launches binds the CALLING thread to the library's
// device first, so loader threads of a rank with local_rank != 0 never end up
// on GPU 0 (phylign_amd/match_stage.py loads indexes from a thread pool).
int bind_thread();
#define NEED_DEV()                                  \
    do {                                            \
        int rc_dev_ = bind_thread();                \
        if (rc_dev_) return rc_dev_;                \
    } while (0)
// frees from any thread: the owning device must be current for the runtime's bookkeeping
static inline void bind_thread_quiet() { if (g_ctx.ready) (void)hipSetDevice(g_ctx.device); }

// ------------------------------------------------------------------ objects
struct pm_index {
    pm_index_info_t info{};
    std::string names_blob;            // all names, '\0' separated
    std::vector<uint64_t> name_off;    // n_docs + 1
    uint8_t* d_matrix = nullptr;
    int g = 1;                         // lanes per row
    uint32_t slabs = 1;
    // compact index: one sub-index per page column, each a matrix of its own
    // (signature_size_p, num_hashes_p) that is page_size bytes wide; empty for classic
    std::vector<pm_index*> parts;
    uint64_t page_size = 0;
    // every document name holds a '_' (the "<random prefix>_<accession>" shape the reference's
    // post-filter relies on, scripts/postprocess_cobs.py:16-18); false -> the n-best cut is
    // never taken on the GPU for this index (see enqueue_search)
    bool names_have_sep = true;
};

struct pm_queries {
    uint32_t k = 0;
    std::vector<std::string> headers;       // header line without its first byte
    std::vector<uint8_t> headerless;        // 1: sequence lines came before any header ("\tN" is printed without '*')
    std::string seqs;                       // packed sequences (host copy, for the 04_filter emit)
    std::vector<uint64_t> seq_off;          // n_queries + 1
    std::vector<uint32_t> n_terms;
    uint64_t total_terms = 0;
    uint64_t n_slots = 0;                   // padded to 8 per query
    std::vector<QDesc> qd;
    // plane classes: queries ordered by class, ranges per class
    std::vector<uint32_t> qmap;
    std::vector<uint64_t> term_prefix;      // k-mers ahead of every position of qmap (n_queries + 1; built by the first search)
    std::vector<uint32_t> blkq;             // 8-slot block -> query
    bool on_device = false;
    uint32_t class_begin[kNumClasses + 1] = {};
    // device
    uint8_t* d_seq = nullptr;
    QDesc* d_qd = nullptr;
    uint32_t* d_blkq = nullptr;
    uint32_t* d_qmap = nullptr;
    uint32_t* d_thr = nullptr; double thr_for = -1.0; uint32_t thr_rule = 0;   // per-query minimum score, cached per threshold (and rule)
    // hash buffers per (canonicalize, num_hashes); the kernel re-runs once per pm_search
    struct HashBuf { int canon; uint32_t nh; uint64_t* d; uint64_t epoch; };
    std::vector<HashBuf> hashes;
    uint64_t epoch = 0;
    hipEvent_t last_use = nullptr;          // recorded behind the last search queued with this set (pm_queries_release_device)
};

// pm_runtime.cpp: fn(0) ... fn(n - 1) on the library's persistent worker threads (and the caller's); returns when all
// are done; an exception thrown by fn on any thread is rethrown here (the first one), after every item has run.  Host-side work (text formatting, deflate, FASTA emit, query parsing) used to start its own std::threads per
// call: thousands of short-lived threads leave glibc one malloc arena each (RSS grew by ~7 MB per stage run on a 256-CPU
// box, tools/leak_check_stage.py); pooled threads keep their arenas.  fn must not call parallel_for itself.
void parallel_for(size_t n, const std::function<void(size_t)>& fn);
size_t parallel_width();                    // worker threads + the caller (<= 16)
// pm_index.cpp: pooled staging buffers of the parallel file loader (released by pm_shutdown)
void release_stage_pool();
void release_query_pool();                   // pm_queries.cpp: device buffers of released query sets
void release_hit_pool();                     // pm_search.cpp: pooled device hit buffers of finished searches
// pm_runtime.cpp: hipMalloc that gives the library's OWN idle HBM back before it reports out-of-memory: the pools of
// query-set buffers and hit buffers hold memory the stage budget does not count (up to 16 + 8 buffers); a signature
// matrix or a hit buffer that does not fit next to them gets one more try after they were released
hipError_t device_malloc_reclaim(void** out, size_t bytes);
int query_buf_take(size_t bytes, void** out);   // a device buffer of a query set (pooled; given back by pm_queries_release_device)
void release_text_pool();                    // pm_text.cpp: the pooled text / gzip buffers of the 03_match writer
// pm_gzfast.cpp: text[0, n) (n < 2^31) as one gzip member -- fixed-Huffman deflate, line-structured matches -- written
// to out[0, gz_fast_bound(n)); returns the member's length
size_t gz_fast_bound(size_t n);
size_t gz_fast_member(const char* text, size_t n, uint8_t* out);
// pm_queries.cpp: HBM copies of a query set on first use; device hashes per (canonicalize, num_hashes)
int upload_queries(pm_queries* q);
int ensure_hashes(pm_queries* q, int canon, uint32_t nh, uint64_t** out);

// pm_set_option("threshold_bound"): product default on; off reproduces the fetch-everything scan
extern uint32_t g_threshold_bound;
// pm_set_option("count_fetched"): the scan also counts the algorithmic bytes it really gathered
extern uint32_t g_count_fetched;
// pm_set_option("single_launch"): every row width (up to 1024 B) goes into the mixed-width launch
extern uint32_t g_single_launch;
// pm_set_option("wide_query"): 0 = automatic (few long queries: several lane groups share a query), 1 = always
// where instantiated (128+ k-mers per query), 2 = never
extern uint32_t g_wide_query;
// pm_set_option("wide_query_split"): 0 = automatic number of workgroups that share a long query's steps, 1 = never split, n = force n
extern uint32_t g_wq_split;
// pm_set_option("cobs_threshold_rule"): how -t becomes a minimum score: 0 = ceil(t * k-mers) (default), 1 = floor, 2 = round
extern uint32_t g_threshold_rule;
// pm_set_option("cobs_tie_order"): 1 = documents of equal score are listed by DESCENDING index (default 0: ascending)
extern uint32_t g_tie_desc;
extern std::atomic<int> g_live_results;      // pm_search.cpp: results not yet freed
// pm_set_option("merge_counting_sort") (default 1): (slot, query) groups written as several runs are merged by the
// O(records + runs) counting sort where it applies; 0 = always the O(records x runs x log) form (A/B, tests)
extern uint32_t g_merge_hist;

// pm_search.cpp: cobs' line order on records: (slot, query, count records first, score desc, doc asc)
bool hit_less(const pm_hit_t& a, const pm_hit_t& b);
void order_hits(pm_hit_t* h, uint64_t n);

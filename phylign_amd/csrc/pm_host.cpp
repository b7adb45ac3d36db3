// pm_host.cpp -- host side of libphylign_match.so: C ABI (include/phylign_match.h),
// COBS classic-index reader, FASTA reader with the cobs CLI's
// record rules (a5 input), search orchestration on a HIP stream (a5-a7), COBS
// result ordering and text (a7) and the fused post-filter (a8).
//
// The reference reaches this functionality through `cobs query ...`
// (scripts/run_cobs_streaming.sh:24-29; Snakefile:419-424, :476-481) and
// `postprocess_cobs.py -n N` (scripts/postprocess_cobs.py:21-39).
// No CPU fallback exists in this file: scoring happens only in pm_kernels.hip.
#include "../../include/phylign_match.h"
#include "pm_internal.h"

#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <tuple>
#include <unistd.h>
#include <vector>

using namespace pm;

// ------------------------------------------------------------------ errors
static thread_local std::string g_err;
static int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_err = buf;
    return code;
}
#define HIPCHK(expr)                                                                   \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess)                                                          \
            return fail(e_ == hipErrorOutOfMemory ? PM_ENOMEM : PM_EHIP, "%s: %s (%s:%d)", \
                        #expr, hipGetErrorString(e_), __FILE__, __LINE__);             \
    } while (0)

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
static Ctx g_ctx;
// HIP's current device is a per-thread setting that starts at 0: every entry point
// that allocates, copies or launches binds the CALLING thread to the library's
// device first, so loader threads of a rank with local_rank != 0 never end up
// on GPU 0 (phylign_amd/match_stage.py loads indexes from a thread pool).
static int bind_thread() {
    if (!g_ctx.ready) return fail(PM_ENODEV, "pm_init() has not succeeded: no GPU bound (there is no CPU fallback)");
    hipError_t e = hipSetDevice(g_ctx.device);
    if (e != hipSuccess) return fail(PM_EHIP, "hipSetDevice(%d): %s", g_ctx.device, hipGetErrorString(e));
    return PM_OK;
}
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
    std::vector<uint32_t> blkq;             // 8-slot block -> query
    bool on_device = false;
    uint32_t class_begin[5] = {0, 0, 0, 0, 0};
    // device
    uint8_t* d_seq = nullptr;
    QDesc* d_qd = nullptr;
    uint32_t* d_blkq = nullptr;
    uint32_t* d_qmap = nullptr;
    uint32_t* d_thr = nullptr; double thr_for = -1.0;      // per-query minimum score, cached per threshold
    // hash buffers per (canonicalize, num_hashes); the kernel re-runs once per pm_search
    struct HashBuf { int canon; uint32_t nh; uint64_t* d; uint64_t epoch; };
    std::vector<HashBuf> hashes;
    uint64_t epoch = 0;
};

static int upload_queries(pm_queries* q);
static int ensure_hashes(pm_queries* q, int canon, uint32_t nh, uint64_t** out);
static const int kPlaneClass[4] = {7, 10, 16, 24};
// pm_set_option("threshold_bound"): product default on; off reproduces the fetch-everything scan
static uint32_t g_threshold_bound = 1;
// pm_set_option("count_fetched"): the scan also counts the algorithmic bytes it really gathered
static uint32_t g_count_fetched = 0;
// pm_set_option("single_launch"): every row width (up to 1024 B) goes into the mixed-width launch
static uint32_t g_single_launch = 0;

// ------------------------------------------------------------------ runtime
extern "C" const char* pm_last_error(void) { return g_err.c_str(); }

extern "C" int pm_init(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(PM_ENODEV, "no HIP device visible (%s); this library has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(PM_EINVAL, "device %d out of range (0..%d)", device, n - 1);
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(PM_ENODEV, "device %d is %s; this build targets gfx950 only", device, prop.gcnArchName);
    if (g_ctx.ready && g_ctx.device == device) return PM_OK;
    if (g_ctx.ready) pm_shutdown();
    HIPCHK(hipStreamCreateWithFlags(&g_ctx.stream, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&g_ctx.copy_stream, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&g_ctx.d2h_stream, hipStreamNonBlocking));
    g_ctx.device = device;
    g_ctx.ready = true;
    return PM_OK;
}

extern "C" void pm_shutdown(void) {
    if (!g_ctx.ready) return;
    bind_thread_quiet();
    hipDeviceSynchronize();
    hipStreamDestroy(g_ctx.stream);
    hipStreamDestroy(g_ctx.copy_stream);
    hipStreamDestroy(g_ctx.d2h_stream);
    for (Workspace* w : g_ctx.ws) {
        if (w->d_cnt) hipFree(w->d_cnt);
        if (w->h_cnt) hipHostFree(w->h_cnt);
        if (w->d_desc) hipFree(w->d_desc);
        if (w->h_desc) hipHostFree(w->h_desc);
        for (auto e : w->events) hipEventDestroy(e);
        if (w->done) hipEventDestroy(w->done);
        delete w;
    }
    for (auto& b : g_ctx.free_hits) hipFree(b.p);
    if (g_ctx.d_fetch) hipFree(g_ctx.d_fetch);
    for (auto& b : g_ctx.free_pinned) hipHostFree(b.p);
    g_ctx = Ctx();
}

extern "C" int pm_device_info(char* name, size_t cap, uint64_t* hbm_total, uint64_t* hbm_free, int* n_cus) {
    NEED_DEV();
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, g_ctx.device));
    if (name && cap) snprintf(name, cap, "%s (%s)", prop.name, prop.gcnArchName);
    size_t fr = 0, tot = 0;
    HIPCHK(hipMemGetInfo(&fr, &tot));
    if (hbm_total) *hbm_total = tot;
    if (hbm_free) *hbm_free = fr;
    if (n_cus) *n_cus = prop.multiProcessorCount;
    return PM_OK;
}

extern "C" void pm_free(void* p) { free(p); }

extern "C" int pm_set_option(const char* name, int64_t value) {
    if (!name) return fail(PM_EINVAL, "bad argument");
    if (strcmp(name, "threshold_bound") == 0) { g_threshold_bound = value ? 1u : 0u; return PM_OK; }
    if (strcmp(name, "count_fetched") == 0) { g_count_fetched = value ? 1u : 0u; return PM_OK; }
    if (strcmp(name, "single_launch") == 0) { g_single_launch = value ? 1u : 0u; return PM_OK; }
    return fail(PM_EINVAL, "unknown option '%s'", name);
}

// The ONE place that turns `-t` into a minimum score (cobs counts_to_result):
// ceil(threshold * num_terms) in IEEE double.  config.yaml:20 -> Snakefile:410.
extern "C" uint32_t pm_threshold_terms(double threshold, uint64_t num_terms) {
    double t = std::ceil(threshold * (double)num_terms);
    if (!(t > 0)) return 0;
    if (t > 4294967295.0) return 4294967295u;
    return (uint32_t)t;
}

// ---------------------------------------------------- classic index header
// "COBS:" "CLASSIC_INDEX" u32 version, then the fields, the newline-terminated
// document names and a closing "CLASSIC_INDEX"; the matrix follows.  The field
// order cannot be checked against a real file here, so both plausible orders
// are tried and the one whose closing magic (and version/k sanity) validates
// is taken.  Returns 0 ok, 1 need more bytes, <0 error.
struct ParsedHeader {
    uint32_t version = 0, term_size = 0, n_docs = 0;
    uint8_t canon = 0;
    uint64_t sig = 0, nh = 0;
    size_t names_off = 0, data_off = 0;
    int layout = 0;
};
static int try_header(const uint8_t* b, size_t len, int layout, ParsedHeader& h) {
    size_t o = 18;
    const size_t fixed = 4 + 4 + 1 + 4 + 8 + 8;
    if (len < o + fixed) return 1;
    auto rd32 = [&](size_t at) { uint32_t v; memcpy(&v, b + at, 4); return v; };
    auto rd64 = [&](size_t at) { uint64_t v; memcpy(&v, b + at, 8); return v; };
    h.version = rd32(o); o += 4;
    h.term_size = rd32(o); o += 4;
    h.canon = b[o]; o += 1;
    if (layout == 0) { h.n_docs = rd32(o); o += 4; h.sig = rd64(o); o += 8; h.nh = rd64(o); o += 8; }
    else             { h.sig = rd64(o); o += 8; h.nh = rd64(o); o += 8; h.n_docs = rd32(o); o += 4; }
    if (h.version != 1 || h.term_size == 0 || h.term_size > 4096 || h.canon > 1) return -1;
    if (h.sig == 0 || h.nh == 0 || h.nh > 64) return -1;
    h.names_off = o;
    for (uint32_t d = 0; d < h.n_docs; ++d) {
        if (o >= len) return 1;
        const void* nl = memchr(b + o, '\n', len - o);
        if (!nl) return (len - o > (1u << 20)) ? -1 : 1;   // a 1 MiB "name" is not a name
        o = (size_t)((const uint8_t*)nl - b) + 1;
    }
    if (o + 13 > len) return 1;
    if (memcmp(b + o, "CLASSIC_INDEX", 13) != 0) return -1;
    h.data_off = o + 13;
    h.layout = layout;
    return 0;
}
static int parse_header(const uint8_t* b, size_t len, ParsedHeader& h) {
    if (len < 18) return 1;
    if (memcmp(b, "COBS:", 5) != 0 || memcmp(b + 5, "CLASSIC_INDEX", 13) != 0) return -1;
    int need_more = 0;
    for (int layout = 0; layout < 2; ++layout) {
        ParsedHeader t;
        int rc = try_header(b, len, layout, t);
        if (rc == 0) { h = t; return 0; }
        if (rc == 1) need_more = 1;
    }
    return need_more ? 1 : -1;
}

static uint64_t pow2ceil(uint64_t x) { uint64_t p = 1; while (p < x) p <<= 1; return p; }
static uint64_t stride_compact(uint64_t rb) { return std::max<uint64_t>(16, (rb + 15) / 16 * 16); }
static uint64_t stride_aligned(uint64_t rb) {
    if (rb <= 16) return 16;
    if (rb <= 128) return pow2ceil(rb);
    return (rb + 127) / 128 * 128;
}

static int finish_index_shape(pm_index* ix, const ParsedHeader& h, int layout, bool want_matrix) {
    pm_index_info_t& in = ix->info;
    in.term_size = h.term_size; in.canonicalize = h.canon; in.signature_size = h.sig;
    in.num_hashes = (uint32_t)h.nh; in.n_docs = h.n_docs;
    in.row_bytes = ((uint64_t)h.n_docs + 7) / 8;
    in.header_layout = (uint32_t)h.layout;
    in.has_matrix = 0; in.stride = 0; in.device_bytes = 0;
    if (!want_matrix) return PM_OK;
    if (in.row_bytes == 0) return fail(PM_EFORMAT, "index holds no documents");
    uint64_t sc = stride_compact(in.row_bytes), sa = stride_aligned(in.row_bytes);
    uint64_t stride = sc;
    // loaders run concurrently (match_stage --loaders): the "does the aligned layout still fit"
    // question and the allocation that answers it are one critical section
    static std::mutex alloc_mu;
    std::lock_guard<std::mutex> alloc_lock(alloc_mu);
    if (layout == PM_LAYOUT_ALIGNED) stride = sa;
    else if (layout == PM_LAYOUT_AUTO) {
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) != hipSuccess) return fail(PM_EHIP, "hipMemGetInfo failed");
        // keep 6% of HBM or 2 GiB free for query state and hit buffers
        uint64_t reserve = std::max<uint64_t>((uint64_t)tot / 16, 2ull << 30);
        stride = (h.sig * sa + reserve <= (uint64_t)fr) ? sa : sc;
    } else if (layout != PM_LAYOUT_COMPACT) return fail(PM_EINVAL, "unknown layout %d", layout);
    in.stride = stride;
    in.device_bytes = h.sig * stride;
    uint64_t lanes = (std::min<uint64_t>(stride, 1024) + 15) / 16;
    ix->g = (int)pow2ceil(lanes);
    ix->slabs = (uint32_t)((stride + 1023) / 1024);
    hipError_t e = hipMalloc((void**)&ix->d_matrix, in.device_bytes);
    if (e != hipSuccess) return fail(PM_ENOMEM, "hipMalloc(%llu bytes) for the signature matrix failed: %s",
                                     (unsigned long long)in.device_bytes, hipGetErrorString(e));
    in.has_matrix = 1;
    return PM_OK;
}

static void take_names(pm_index* ix, const uint8_t* b, const ParsedHeader& h) {
    ix->name_off.resize((size_t)h.n_docs + 1);
    size_t o = h.names_off;
    for (uint32_t d = 0; d < h.n_docs; ++d) {
        const uint8_t* nl = (const uint8_t*)memchr(b + o, '\n', h.data_off - o);
        size_t l = (size_t)(nl - (b + o));
        ix->name_off[d] = ix->names_blob.size();
        ix->names_blob.append((const char*)b + o, l);
        ix->names_blob.push_back('\0');
        if (!memchr(b + o, '_', l)) ix->names_have_sep = false;
        o += l + 1;
    }
    ix->name_off[h.n_docs] = ix->names_blob.size();
}

// Byte source with push-back, so the header bytes read ahead can be re-used.
struct Reader {
    int fd = -1;
    const uint8_t* mem = nullptr; size_t mem_len = 0, mem_pos = 0;
    std::vector<uint8_t> pending; size_t pend_pos = 0;
    // returns bytes read (< n only at EOF), -1 on error
    ssize_t read_full(uint8_t* dst, size_t n) {
        size_t got = 0;
        if (pend_pos < pending.size()) {
            size_t t = std::min(n, pending.size() - pend_pos);
            memcpy(dst, pending.data() + pend_pos, t); pend_pos += t; got += t;
        }
        if (mem) {
            size_t t = std::min(n - got, mem_len - mem_pos);
            memcpy(dst + got, mem + mem_pos, t); mem_pos += t; got += t;
            return (ssize_t)got;
        }
        while (got < n) {
            ssize_t r = ::read(fd, dst + got, n - got);
            if (r < 0) { if (errno == EINTR) continue; return -1; }
            if (r == 0) break;
            got += (size_t)r;
        }
        return (ssize_t)got;
    }
};

// Streams S rows of rb bytes from the reader into ix->d_matrix (row stride
// ix->info.stride): double-buffered pinned chunks -> staging -> re-stride kernel.
static int stream_matrix(Reader& rd, pm_index* ix, uint64_t rb, uint64_t S) {
    const uint64_t stride = ix->info.stride;
    const uint64_t chunk_rows = std::max<uint64_t>(1, (32ull << 20) / rb);
    const size_t chunk_bytes = (size_t)(std::min<uint64_t>(chunk_rows, S) * rb);
    uint8_t* hbuf[2] = {nullptr, nullptr};
    uint8_t* dbuf[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    auto cleanup = [&]() {
        for (int i = 0; i < 2; ++i) {
            if (hbuf[i]) hipHostFree(hbuf[i]);
            if (dbuf[i]) hipFree(dbuf[i]);
            if (ev[i]) hipEventDestroy(ev[i]);
        }
    };
#define LCHK(expr)                                                                        \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) {                                                           \
            cleanup();                                                                    \
            return fail(PM_EHIP, "%s: %s", #expr, hipGetErrorString(e_));                 \
        }                                                                                 \
    } while (0)
    for (int i = 0; i < 2; ++i) {
        LCHK(hipHostMalloc((void**)&hbuf[i], chunk_bytes, hipHostMallocDefault));
        LCHK(hipMalloc((void**)&dbuf[i], chunk_bytes));
        LCHK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
    }
    uint64_t row = 0; int cur = 0; bool used[2] = {false, false};
    while (row < S) {
        const uint64_t nrows = std::min<uint64_t>(chunk_rows, S - row);
        const size_t nbytes = (size_t)(nrows * rb);
        if (used[cur]) LCHK(hipEventSynchronize(ev[cur]));
        ssize_t r = rd.read_full(hbuf[cur], nbytes);
        if (r < 0 || (size_t)r != nbytes) {
            cleanup();
            return fail(PM_EIO, "index stream ended after %llu of %llu matrix bytes",
                        (unsigned long long)(row * rb + (r > 0 ? (uint64_t)r : 0)), (unsigned long long)(S * rb));
        }
        LCHK(hipMemcpyAsync(dbuf[cur], hbuf[cur], nbytes, hipMemcpyHostToDevice, g_ctx.copy_stream));
        LCHK(launch_restride(dbuf[cur], rb, ix->d_matrix + row * stride, stride, nrows, g_ctx.copy_stream));
        LCHK(hipEventRecord(ev[cur], g_ctx.copy_stream));
        used[cur] = true;
        row += nrows; cur ^= 1;
    }
    LCHK(hipStreamSynchronize(g_ctx.copy_stream));
#undef LCHK
    cleanup();
    return PM_OK;
}

// Compact index header ("COBS:" "COMPACT_INDEX", upstream
// cobs/file/compact_index_header.cpp; Phylign itself only uses classic indexes,
// Snakefile:48 -- SURVEY.md 8f rank 3): u32 version, u32 term_size, u8
// canonicalize, u32 n_parameters, u32 n_docs, u64 page_size, n_parameters x
// {u64 signature_size, u64 num_hashes}, names, zero padding so that the closing
// magic ends on a page boundary, "COMPACT_INDEX", then the sub-indexes.
struct ParsedCompact {
    uint32_t term_size = 0, n_parts = 0, n_docs = 0;
    uint8_t canon = 0;
    uint64_t page = 0;
    std::vector<uint64_t> sig, nh;
    size_t names_off = 0, data_off = 0;
};
static int parse_compact(const uint8_t* b, size_t len, ParsedCompact& c) {   // 0 ok, 1 need more, <0 bad
    const size_t fixed = 18 + 4 + 4 + 1 + 4 + 4 + 8;
    if (len < fixed) return 1;
    size_t o = 18;
    auto rd32 = [&](size_t at) { uint32_t v; memcpy(&v, b + at, 4); return v; };
    auto rd64 = [&](size_t at) { uint64_t v; memcpy(&v, b + at, 8); return v; };
    const uint32_t ver = rd32(o); o += 4;
    c.term_size = rd32(o); o += 4;
    c.canon = b[o]; o += 1;
    c.n_parts = rd32(o); o += 4;
    c.n_docs = rd32(o); o += 4;
    c.page = rd64(o); o += 8;
    if (ver != 1 || c.term_size == 0 || c.term_size > 4096 || c.canon > 1 || c.page == 0 || c.page > (1ull << 30) ||
        c.n_parts == 0 || c.n_parts > (1u << 20) || (uint64_t)c.n_parts * c.page * 8 < c.n_docs) return -1;
    if (len < o + (size_t)c.n_parts * 16) return 1;
    c.sig.resize(c.n_parts); c.nh.resize(c.n_parts);
    for (uint32_t p = 0; p < c.n_parts; ++p) {
        c.sig[p] = rd64(o); c.nh[p] = rd64(o + 8); o += 16;
        if (c.sig[p] == 0 || c.nh[p] == 0 || c.nh[p] > 64) return -1;
    }
    c.names_off = o;
    for (uint32_t d = 0; d < c.n_docs; ++d) {
        if (o >= len) return 1;
        const void* nl = memchr(b + o, '\n', len - o);
        if (!nl) return (len - o > (1u << 20)) ? -1 : 1;
        o = (size_t)((const uint8_t*)nl - b) + 1;
    }
    const size_t names_end = o;
    o += (size_t)((c.page - ((o + 13) % c.page)) % c.page);
    if (o + 13 > len) return 1;
    if (memcmp(b + o, "COMPACT_INDEX", 13) != 0) return -1;
    (void)names_end;
    c.data_off = o + 13;
    return 0;
}

static int load_from_reader(Reader& rd, uint64_t size_hint, int layout, bool header_only, pm_index_t** out) {
    // 1. header: read ahead until it parses (classic or compact)
    std::vector<uint8_t> head;
    size_t want = 1 << 16;
    ParsedHeader h;
    ParsedCompact pc;
    bool compact = false;
    for (;;) {
        size_t old = head.size();
        head.resize(want);
        ssize_t r = rd.read_full(head.data() + old, want - old);
        if (r < 0) return fail(PM_EIO, "read error on index stream: %s", strerror(errno));
        head.resize(old + (size_t)r);
        compact = head.size() >= 18 && memcmp(head.data(), "COBS:", 5) == 0 && memcmp(head.data() + 5, "COMPACT_INDEX", 13) == 0;
        int rc = compact ? parse_compact(head.data(), head.size(), pc) : parse_header(head.data(), head.size(), h);
        if (rc == 0) break;
        if (rc < 0) return fail(PM_EFORMAT, "input is not a COBS classic (or compact) index (magic/version/field check failed)");
        if ((size_t)r < want - old) return fail(PM_EFORMAT, "index stream ended inside the header");
        want *= 2;
        if (want > (1ull << 31)) return fail(PM_EFORMAT, "index header larger than 2 GiB");
    }
    pm_index* ix = new pm_index();
    if (!compact) {
        take_names(ix, head.data(), h);
        int rc = finish_index_shape(ix, h, layout, !header_only);
        if (rc) { delete ix; return rc; }
        if (header_only) { *out = ix; return PM_OK; }
        const uint64_t rb = ix->info.row_bytes, S = h.sig;
        if (size_hint && size_hint != h.data_off + S * rb)
            fprintf(stderr, "phylign_match: warning: --index-sizes %llu != header-implied %llu bytes\n",
                    (unsigned long long)size_hint, (unsigned long long)(h.data_off + S * rb));
        rd.pending.assign(head.begin() + (long)h.data_off, head.end());
        rd.pend_pos = 0;
        rc = stream_matrix(rd, ix, rb, S);
        if (rc) { pm_index_free(ix); return rc; }
        *out = ix;
        return PM_OK;
    }
    // ---- compact: names for all documents, one sub-index object per page column
    {
        ParsedHeader nh_;                       // reuse take_names through a classic-shaped view
        nh_.n_docs = pc.n_docs; nh_.names_off = pc.names_off; nh_.data_off = pc.data_off;
        take_names(ix, head.data(), nh_);
    }
    ix->page_size = pc.page;
    ix->info.term_size = pc.term_size; ix->info.canonicalize = pc.canon; ix->info.n_docs = pc.n_docs;
    ix->info.signature_size = pc.sig[0]; ix->info.num_hashes = (uint32_t)pc.nh[0];
    ix->info.row_bytes = pc.page; ix->info.n_parts = pc.n_parts; ix->info.page_size = pc.page;
    rd.pending.assign(head.begin() + (long)pc.data_off, head.end());
    rd.pend_pos = 0;
    for (uint32_t p = 0; p < pc.n_parts; ++p) {
        pm_index* part = new pm_index();
        ix->parts.push_back(part);
        const uint64_t first = (uint64_t)p * pc.page * 8;
        ParsedHeader ph;
        ph.version = 1; ph.term_size = pc.term_size; ph.canon = pc.canon; ph.sig = pc.sig[p]; ph.nh = pc.nh[p];
        ph.n_docs = first >= pc.n_docs ? 0u : (uint32_t)std::min<uint64_t>(pc.page * 8, pc.n_docs - first);
        if (ph.n_docs == 0) {                     // page column without documents: skip its bytes, never searched
            if (!header_only) {
                std::vector<uint8_t> sink(1 << 20);
                uint64_t left = pc.sig[p] * pc.page;
                while (left) { ssize_t r = rd.read_full(sink.data(), (size_t)std::min<uint64_t>(left, sink.size())); if (r <= 0) break; left -= (uint64_t)r; }
            }
            part->info.n_docs = 0;
            continue;
        }
        // rows of a sub-index are page_size bytes in the file whatever its document count
        int rc = finish_index_shape(part, ph, layout, false);
        part->info.row_bytes = pc.page;
        if (rc == PM_OK && !header_only) {
            pm_index_info_t& in = part->info;
            const uint64_t sc = stride_compact(pc.page), sa = stride_aligned(pc.page);
            uint64_t stride = (layout == PM_LAYOUT_COMPACT) ? sc : sa;
            in.stride = stride; in.device_bytes = pc.sig[p] * stride;
            part->g = (int)pow2ceil((std::min<uint64_t>(stride, 1024) + 15) / 16);
            part->slabs = (uint32_t)((stride + 1023) / 1024);
            hipError_t e = hipMalloc((void**)&part->d_matrix, in.device_bytes);
            if (e != hipSuccess) rc = fail(PM_ENOMEM, "hipMalloc(%llu bytes) for sub-index %u failed: %s",
                                           (unsigned long long)in.device_bytes, p, hipGetErrorString(e));
            else { in.has_matrix = 1; rc = stream_matrix(rd, part, pc.page, pc.sig[p]); }
        }
        if (rc) { pm_index_free(ix); return rc; }
        ix->info.device_bytes += part->info.device_bytes;
    }
    ix->info.has_matrix = header_only ? 0 : 1;
    ix->info.stride = ix->parts[0]->info.stride;
    (void)size_hint;
    *out = ix;
    return PM_OK;
}

extern "C" int pm_index_load_fd(int fd, uint64_t size_hint, int layout, pm_index_t** out) {
    NEED_DEV();
    if (!out || fd < 0) return fail(PM_EINVAL, "bad argument");
    Reader rd; rd.fd = fd;
    return load_from_reader(rd, size_hint, layout, false, out);
}
extern "C" int pm_index_load_file(const char* path, uint64_t size_hint, int layout, pm_index_t** out) {
    NEED_DEV();
    if (!path || !out) return fail(PM_EINVAL, "bad argument");
    int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(PM_EIO, "cannot open index '%s': %s", path, strerror(errno));
    int rc = pm_index_load_fd(fd, size_hint, layout, out);
    close(fd);
    return rc;
}
extern "C" int pm_index_load_mem(const void* buf, size_t len, int layout, pm_index_t** out) {
    NEED_DEV();
    if (!buf || !out) return fail(PM_EINVAL, "bad argument");
    Reader rd; rd.mem = (const uint8_t*)buf; rd.mem_len = len;
    return load_from_reader(rd, 0, layout, false, out);
}
extern "C" int pm_index_load_header_mem(const void* buf, size_t len, pm_index_t** out) {
    if (!buf || !out) return fail(PM_EINVAL, "bad argument");
    Reader rd; rd.mem = (const uint8_t*)buf; rd.mem_len = len;
    return load_from_reader(rd, 0, PM_LAYOUT_COMPACT, true, out);
}

static uint64_t host_splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}

extern "C" int pm_index_synth(uint32_t batch_id, uint32_t n_docs, uint64_t signature_size,
                              uint32_t num_hashes, uint32_t term_size, uint64_t seed,
                              int layout, int header_only, pm_index_t** out) {
    if (!out || n_docs == 0 || signature_size == 0 || num_hashes == 0 || term_size == 0)
        return fail(PM_EINVAL, "bad synthetic index shape");
    if (!header_only) NEED_DEV();
    pm_index* ix = new pm_index();
    ParsedHeader h;
    h.version = 1; h.term_size = term_size; h.canon = 1; h.sig = signature_size; h.nh = num_hashes;
    h.n_docs = n_docs; h.layout = 0;
    // names "<5 hex>_SYN<batch>D<doc>": a pseudo-random sorting prefix, one underscore
    const uint64_t kb = host_splitmix64(seed ^ ((uint64_t)batch_id * 0xD1B54A32D192ED03ULL));
    ix->name_off.resize((size_t)n_docs + 1);
    char nm[64];
    for (uint32_t d = 0; d < n_docs; ++d) {
        int l = snprintf(nm, sizeof nm, "%05x_SYN%03uD%07u",
                         (unsigned)(host_splitmix64(kb ^ (0xA5A5A5A5ull + d)) & 0xFFFFF), batch_id, d);
        ix->name_off[d] = ix->names_blob.size();
        ix->names_blob.append(nm, (size_t)l);
        ix->names_blob.push_back('\0');
    }
    ix->name_off[n_docs] = ix->names_blob.size();
    int rc = finish_index_shape(ix, h, layout, !header_only);
    if (rc) { delete ix; return rc; }
    if (!header_only) {
        hipError_t e = launch_synth(ix->d_matrix, ix->info.stride, signature_size, n_docs, seed, batch_id, g_ctx.stream);
        if (e == hipSuccess) e = hipStreamSynchronize(g_ctx.stream);
        if (e != hipSuccess) { pm_index_free(ix); return fail(PM_EHIP, "synthetic generator: %s", hipGetErrorString(e)); }
    }
    *out = ix;
    return PM_OK;
}

extern "C" int pm_index_plant(pm_index_t* ix, const uint64_t* rows, const uint32_t* docs, size_t n) {
    NEED_DEV();
    if (!ix || !ix->d_matrix) return fail(PM_EINVAL, "index has no matrix (planting works on classic indexes)");
    if (n == 0) return PM_OK;
    for (size_t i = 0; i < n; ++i)
        if (rows[i] >= ix->info.signature_size || docs[i] >= ix->info.n_docs)
            return fail(PM_EINVAL, "plant %zu out of range", i);
    uint64_t* dr = nullptr; uint32_t* dd = nullptr;
    HIPCHK(hipMalloc((void**)&dr, n * 8));
    hipError_t e = hipMalloc((void**)&dd, n * 4);
    if (e == hipSuccess) e = hipMemcpyAsync(dr, rows, n * 8, hipMemcpyHostToDevice, g_ctx.stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dd, docs, n * 4, hipMemcpyHostToDevice, g_ctx.stream);
    if (e == hipSuccess) e = launch_plant(ix->d_matrix, ix->info.stride, dr, dd, n, g_ctx.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(g_ctx.stream);
    hipFree(dr); if (dd) hipFree(dd);
    if (e != hipSuccess) return fail(PM_EHIP, "plant: %s", hipGetErrorString(e));
    return PM_OK;
}

// Synthetic "related batch" content (measurement / test aid, see k_plant_cluster): makes this
// index the HOME batch of queries q_first, q_first + q_step, ...: about half of its 32-document
// clusters match each of those queries at a fraction between 0.6 and 1.0.
extern "C" int pm_index_plant_cluster(pm_index_t* ix, pm_queries_t* q, uint32_t q_first, uint32_t q_step, uint64_t seed) {
    NEED_DEV();
    if (!ix || !q || !ix->d_matrix || q_step == 0) return fail(PM_EINVAL, "bad argument (planting works on classic indexes)");
    if (ix->info.term_size != q->k) return fail(PM_EINVAL, "term_size mismatch");
    const size_t nq = q->headers.size();
    if (q_first >= nq) return PM_OK;
    { int urc = upload_queries(q); if (urc) return urc; }
    uint64_t* d_h = nullptr;
    q->epoch++;
    { int rc = ensure_hashes(q, (int)ix->info.canonicalize, ix->info.num_hashes, &d_h); if (rc) return rc; }
    const uint32_t n_sel = (uint32_t)((nq - q_first + q_step - 1) / q_step);
    uint32_t max_terms = 0;
    for (size_t i = q_first; i < nq; i += q_step) max_terms = std::max(max_terms, q->n_terms[i]);
    HIPCHK(launch_plant_cluster(ix->d_matrix, ix->info.stride, ix->info.signature_size, ix->info.n_docs, d_h, q->d_qd,
                                ix->info.num_hashes, q_first, q_step, n_sel, max_terms, seed, g_ctx.stream));
    HIPCHK(hipStreamSynchronize(g_ctx.stream));
    return PM_OK;
}

// Copies rows [row0, row0 + n) (row_bytes each, file layout) back to the host: lets a test
// rebuild the .cobs_classic file of a synthetic / planted index for the oracle.
extern "C" int pm_index_read_rows(const pm_index_t* ix, uint64_t row0, uint64_t n, void* out) {
    NEED_DEV();
    if (!ix || !ix->d_matrix || !out || row0 + n > ix->info.signature_size) return fail(PM_EINVAL, "bad argument");
    if (n == 0) return PM_OK;
    HIPCHK(hipMemcpy2D(out, ix->info.row_bytes, ix->d_matrix + row0 * ix->info.stride, ix->info.stride,
                       ix->info.row_bytes, n, hipMemcpyDeviceToHost));
    return PM_OK;
}

extern "C" int pm_index_probe_gather(const pm_index_t* ix, uint64_t n_groups, uint64_t lookups_per_group,
                                     double* ms, uint64_t* bytes) {
    NEED_DEV();
    if (!ix || !ix->d_matrix || !ms || !bytes || n_groups == 0) return fail(PM_EINVAL, "bad argument");
    if (ix->slabs != 1) return fail(PM_EINVAL, "probe supports rows up to 1024 bytes");
    lookups_per_group = (lookups_per_group + 15) / 16 * 16;
    uint32_t* sink = nullptr;
    HIPCHK(hipMalloc((void**)&sink, 4));
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    hipError_t e = hipEventRecord(e0, g_ctx.stream);
    if (e == hipSuccess) e = launch_probe_gather(ix->d_matrix, ix->info.stride, ix->info.signature_size, ix->g,
                                                 n_groups, lookups_per_group, sink, g_ctx.stream);
    if (e == hipSuccess) e = hipEventRecord(e1, g_ctx.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(g_ctx.stream);
    float f = 0;
    if (e == hipSuccess) e = hipEventElapsedTime(&f, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1); hipFree(sink);
    if (e != hipSuccess) return fail(PM_EHIP, "probe: %s", hipGetErrorString(e));
    *ms = f; *bytes = n_groups * lookups_per_group * ix->info.row_bytes;
    return PM_OK;
}

extern "C" int pm_index_from_names(const char* names, size_t len, uint32_t n_docs, uint32_t term_size, pm_index_t** out) {
    if ((!names && len) || !out) return fail(PM_EINVAL, "bad argument");
    pm_index* ix = new pm_index();
    ix->info.term_size = term_size; ix->info.n_docs = n_docs; ix->info.row_bytes = ((uint64_t)n_docs + 7) / 8;
    ix->name_off.resize((size_t)n_docs + 1);
    size_t o = 0;
    for (uint32_t d = 0; d < n_docs; ++d) {
        const char* nl = (o < len) ? (const char*)memchr(names + o, '\n', len - o) : nullptr;
        if (!nl) { delete ix; return fail(PM_EINVAL, "names blob holds fewer than %u newline-terminated names", n_docs); }
        ix->name_off[d] = ix->names_blob.size();
        ix->names_blob.append(names + o, (size_t)(nl - (names + o)));
        ix->names_blob.push_back('\0');
        if (!memchr(names + o, '_', (size_t)(nl - (names + o)))) ix->names_have_sep = false;
        o = (size_t)(nl - names) + 1;
    }
    ix->name_off[n_docs] = ix->names_blob.size();
    *out = ix;
    return PM_OK;
}
extern "C" int pm_index_drop_matrix(pm_index_t* ix) {
    if (!ix) return fail(PM_EINVAL, "bad argument");
    bind_thread_quiet();
    if (ix->d_matrix) { hipFree(ix->d_matrix); ix->d_matrix = nullptr; }
    for (pm_index* p : ix->parts) pm_index_drop_matrix(p);
    ix->info.has_matrix = 0; ix->info.device_bytes = 0;
    return PM_OK;
}

extern "C" int pm_index_info(const pm_index_t* ix, pm_index_info_t* info) {
    if (!ix || !info) return fail(PM_EINVAL, "bad argument");
    *info = ix->info;
    return PM_OK;
}
extern "C" const char* pm_index_doc_name(const pm_index_t* ix, uint32_t doc, size_t* len) {
    if (!ix || doc >= ix->info.n_docs) return nullptr;
    if (len) *len = (size_t)(ix->name_off[doc + 1] - ix->name_off[doc] - 1);
    return ix->names_blob.data() + ix->name_off[doc];
}
extern "C" int pm_index_read_row(const pm_index_t* ix, uint64_t row, void* out) {
    NEED_DEV();
    if (!ix || !ix->d_matrix || !out || row >= ix->info.signature_size) return fail(PM_EINVAL, "bad argument");
    HIPCHK(hipMemcpy(out, ix->d_matrix + row * ix->info.stride, ix->info.row_bytes, hipMemcpyDeviceToHost));
    return PM_OK;
}
// GPU that holds the signature matrix (hipPointerGetAttributes); -1 for header-only handles
extern "C" int pm_index_device(const pm_index_t* ix, int* device) {
    if (!ix || !device) return fail(PM_EINVAL, "bad argument");
    const uint8_t* p = ix->d_matrix;
    if (!p) for (const pm_index* part : ix->parts) if (part->d_matrix) { p = part->d_matrix; break; }
    *device = -1;
    if (!p) return PM_OK;
    hipPointerAttribute_t at;
    HIPCHK(hipPointerGetAttributes(&at, p));
    *device = at.device;
    return PM_OK;
}
extern "C" void pm_index_free(pm_index_t* ix) {
    if (!ix) return;
    bind_thread_quiet();
    if (ix->d_matrix) hipFree(ix->d_matrix);
    for (pm_index* p : ix->parts) pm_index_free(p);
    delete ix;
}

// ------------------------------------------------------------------ queries
// Record rules of `cobs query -f` (upstream src/main.cpp process_query): see
// include/phylign_match.h.  The input contract (upper-case ACGT, single line)
// is produced by Snakefile:314-333.
extern "C" int pm_queries_parse(const char* fasta, size_t len, uint32_t term_size, pm_queries_t** out) {
    if ((!fasta && len) || !out || term_size == 0) return fail(PM_EINVAL, "bad argument");
    pm_queries* q = new pm_queries();
    q->k = term_size;
    std::string& seqs = q->seqs;            // packed sequences
    std::vector<uint64_t>& seq_off = q->seq_off;
    std::string cur_hdr, cur_seq;
    bool have_any = false;
    int rc = PM_OK;
    auto flush = [&]() -> int {
        if (cur_seq.empty()) return PM_OK;
        if (cur_seq.size() < term_size)
            return fail(PM_EQUERY, "query '%s' too short: %zu < %u characters", cur_hdr.c_str(), cur_seq.size(), term_size);
        for (size_t i = 0; i < cur_seq.size(); ++i) {
            const char c = cur_seq[i];
            if (c != 'A' && c != 'C' && c != 'G' && c != 'T')
                return fail(PM_EQUERY, "query '%s': byte 0x%02x at position %zu is not one of ACGT "
                            "(Phylign's fix_query step maps such bases to A)", cur_hdr.c_str(), (unsigned char)c, i);
        }
        const uint64_t nt = cur_seq.size() - term_size + 1;
        if (nt >= (1ull << 24)) return fail(PM_ERANGE, "query '%s' has %llu k-mers; this build supports < 2^24 per query",
                                            cur_hdr.c_str(), (unsigned long long)nt);
        q->headers.push_back(cur_hdr);
        q->headerless.push_back(have_any ? 0 : 1);
        q->n_terms.push_back((uint32_t)nt);
        seq_off.push_back(seqs.size());
        seqs += cur_seq;
        return PM_OK;
    };
    size_t p = 0;
    while (p < len && rc == PM_OK) {
        const char* nl = (const char*)memchr(fasta + p, '\n', len - p);
        size_t ll = nl ? (size_t)(nl - (fasta + p)) : len - p;
        const char* line = fasta + p;
        p += ll + (nl ? 1 : 0);
        if (ll == 0) continue;
        if (line[0] == '>' || line[0] == ';') {
            rc = flush();
            cur_hdr.assign(line + 1, ll - 1);
            cur_seq.clear();
            have_any = true;
        } else {
            cur_seq.append(line, ll);
        }
    }
    if (rc == PM_OK) rc = flush();
    if (rc != PM_OK) { delete q; return rc; }

    const size_t nq = q->headers.size();
    seq_off.push_back(seqs.size());
    if (nq >= 0xFFFFFFFFull) { delete q; return fail(PM_ERANGE, "too many queries"); }
    q->qd.resize(nq);
    uint64_t blk = 0;
    for (size_t i = 0; i < nq; ++i) {
        q->qd[i].n_terms = q->n_terms[i];
        q->qd[i].pad_blk = (uint32_t)blk;
        q->qd[i].seq_lo = (uint32_t)seq_off[i];
        q->qd[i].seq_hi = (uint32_t)(seq_off[i] >> 32);
        blk += (q->n_terms[i] + 7) / 8;
        q->total_terms += q->n_terms[i];
        if (blk >= 0xFFFFFFFFull) { delete q; return fail(PM_ERANGE, "query set too large (>= 2^35 padded k-mers)"); }
    }
    q->n_slots = blk * 8;
    std::vector<uint32_t> blkq((size_t)blk);
    for (size_t i = 0; i < nq; ++i) {
        uint64_t b0 = q->qd[i].pad_blk, nb = (q->n_terms[i] + 7) / 8;
        for (uint64_t b = 0; b < nb; ++b) blkq[(size_t)(b0 + b)] = (uint32_t)i;
    }
    // plane classes (counter width): stable partition of query ids by class
    auto cls = [](uint32_t nt) { return nt <= 127 ? 0 : nt <= 1023 ? 1 : nt <= 65535 ? 2 : 3; };
    q->qmap.reserve(nq);
    for (int c = 0; c < 4; ++c) {
        q->class_begin[c] = (uint32_t)q->qmap.size();
        for (size_t i = 0; i < nq; ++i) if (cls(q->n_terms[i]) == c) q->qmap.push_back((uint32_t)i);
    }
    q->class_begin[4] = (uint32_t)q->qmap.size();

    q->blkq.swap(blkq);
    // HBM copies are made on first use by a compute call (upload_queries): parsing, text
    // formatting and the 04_filter merge are host work and need no GPU
    *out = q;
    return PM_OK;
}

extern "C" int pm_queries_count(const pm_queries_t* q, uint64_t* n_queries, uint64_t* n_terms) {
    if (!q) return fail(PM_EINVAL, "bad argument");
    if (n_queries) *n_queries = q->headers.size();
    if (n_terms) *n_terms = q->total_terms;
    return PM_OK;
}
extern "C" int pm_queries_terms(const pm_queries_t* q, uint64_t i, uint64_t* n_terms) {
    if (!q || i >= q->n_terms.size() || !n_terms) return fail(PM_EINVAL, "bad argument");
    *n_terms = q->n_terms[(size_t)i];
    return PM_OK;
}
extern "C" void pm_queries_free(pm_queries_t* q) {
    if (!q) return;
    bind_thread_quiet();
    if (g_ctx.ready && q->on_device) hipStreamSynchronize(g_ctx.stream);   // a search in flight may still read them
    if (q->d_seq) hipFree(q->d_seq);
    if (q->d_qd) hipFree(q->d_qd);
    if (q->d_blkq) hipFree(q->d_blkq);
    if (q->d_qmap) hipFree(q->d_qmap);
    if (q->d_thr) hipFree(q->d_thr);
    for (auto& h : q->hashes) if (h.d) hipFree(h.d);
    delete q;
}

static int upload_queries(pm_queries* q) {
    if (q->on_device) return PM_OK;
    const size_t nq = q->headers.size();
    if (nq) {
        HIPCHK(hipMalloc((void**)&q->d_seq, q->seqs.size() + 64));
        HIPCHK(hipMemset(q->d_seq, 0, q->seqs.size() + 64));
        HIPCHK(hipMalloc((void**)&q->d_qd, nq * sizeof(QDesc)));
        HIPCHK(hipMalloc((void**)&q->d_blkq, std::max<size_t>(q->blkq.size(), 1) * 4));
        HIPCHK(hipMalloc((void**)&q->d_qmap, nq * 4));
        HIPCHK(hipMemcpy(q->d_seq, q->seqs.data(), q->seqs.size(), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(q->d_qd, q->qd.data(), nq * sizeof(QDesc), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(q->d_blkq, q->blkq.data(), q->blkq.size() * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(q->d_qmap, q->qmap.data(), nq * 4, hipMemcpyHostToDevice));
    }
    q->on_device = true;
    return PM_OK;
}

// Device hashes for (canonicalize, num_hashes).  The buffer is kept on the query
// set; the kernel runs once per epoch (pm_search bumps the epoch: one job =
// hash + scan, nothing is carried over between searches).
static int ensure_hashes(pm_queries* q, int canon, uint32_t nh, uint64_t** out) {
    pm_queries::HashBuf* hb = nullptr;
    for (auto& h : q->hashes) if (h.canon == canon && h.nh == nh) hb = &h;
    if (!hb) {
        q->hashes.push_back({canon, nh, nullptr, ~0ull});
        hb = &q->hashes.back();
        if (q->n_slots) HIPCHK(hipMalloc((void**)&hb->d, q->n_slots * nh * 8));
    }
    if (hb->epoch != q->epoch) {
        HIPCHK(launch_hash_terms(q->d_seq, q->d_qd, q->d_blkq, q->n_slots, q->k, canon, nh, hb->d, g_ctx.stream));
        hb->epoch = q->epoch;
    }
    *out = hb->d;
    return PM_OK;
}

extern "C" int pm_hash_terms(pm_queries_t* q, int canonicalize, uint32_t num_hashes, uint64_t* out) {
    NEED_DEV();
    if (!q || !out || num_hashes == 0) return fail(PM_EINVAL, "bad argument");
    { int urc = upload_queries(q); if (urc) return urc; }
    q->epoch++;           // force a fresh kernel run
    uint64_t* d_h = nullptr;
    int rc = ensure_hashes(q, canonicalize ? 1 : 0, num_hashes, &d_h);
    if (rc) return rc;
    std::vector<uint64_t> padded((size_t)(q->n_slots * num_hashes));
    if (!padded.empty())
        HIPCHK(hipMemcpyAsync(padded.data(), d_h, padded.size() * 8, hipMemcpyDeviceToHost, g_ctx.stream));
    HIPCHK(hipStreamSynchronize(g_ctx.stream));
    uint64_t o = 0;
    for (size_t i = 0; i < q->n_terms.size(); ++i) {
        const uint64_t b0 = q->qd[i].pad_blk;
        for (uint32_t t = 0; t < q->n_terms[i]; ++t)
            for (uint32_t j = 0; j < num_hashes; ++j)
                out[o++] = padded[(size_t)(((b0 + t / 8) * num_hashes + j) * 8 + (t & 7))];
    }
    return PM_OK;
}

// ------------------------------------------------------------------- search
static std::mutex g_pool_mu;            // workspace / hit-buffer pools (searches may come from several threads)

static Workspace* take_workspace() {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (Workspace* w : g_ctx.ws) if (!w->busy) { w->busy = true; return w; }
    Workspace* w = new Workspace();
    w->busy = true;
    g_ctx.ws.push_back(w);
    return w;
}
static void give_workspace(Workspace* w) {
    if (!w) return;
    std::lock_guard<std::mutex> lk(g_pool_mu);
    w->busy = false;
}
static int ws_event(Workspace* w, size_t i, hipEvent_t* ev) {
    while (w->events.size() <= i) {
        hipEvent_t e = nullptr;
        HIPCHK(hipEventCreate(&e));
        w->events.push_back(e);
    }
    *ev = w->events[i];
    return PM_OK;
}
static inline uint64_t run_cap_of(uint64_t cap) { return cap / 2 + 1; }
static int take_hit_buffer(uint64_t cap, HitBuf* out) {
    if (cap >= 0xFFFFFFF0ull) return fail(PM_ERANGE, "more than 2^32 hit records in one search: split the query set");
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        size_t best = g_ctx.free_hits.size();          // the largest pooled buffer that is big enough
        for (size_t i = 0; i < g_ctx.free_hits.size(); ++i)
            if (g_ctx.free_hits[i].cap >= cap && (best == g_ctx.free_hits.size() || g_ctx.free_hits[i].cap > g_ctx.free_hits[best].cap))
                best = i;
        if (best != g_ctx.free_hits.size()) {
            *out = g_ctx.free_hits[best];
            g_ctx.free_hits.erase(g_ctx.free_hits.begin() + (long)best);
            return PM_OK;
        }
    }
    out->cap = cap;
    // `cap` records followed by the run directory (a run holds at least two records)
    HIPCHK(hipMalloc((void**)&out->p, (cap + run_cap_of(cap)) * sizeof(uint4)));
    return PM_OK;
}
static void give_hit_buffer(HitBuf b) {
    if (!b.p) return;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        // raw + ordered buffers of two searches in flight must all come from the pool: hipFree
        // waits for the device, i.e. for the kernels of the NEXT search, and would undo the overlap
        if (g_ctx.ready) {
            g_ctx.free_hits.push_back(b);
            if (g_ctx.free_hits.size() <= 8) return;
            size_t small = 0;                              // pool full: the smallest buffer goes
            for (size_t i = 1; i < g_ctx.free_hits.size(); ++i) if (g_ctx.free_hits[i].cap < g_ctx.free_hits[small].cap) small = i;
            b = g_ctx.free_hits[small];
            g_ctx.free_hits.erase(g_ctx.free_hits.begin() + (long)small);
        }
    }
    hipFree(b.p);
}

// pinned host buffers (results on the host, staging): pooled, since pinning memory is slow
static int take_pinned(size_t bytes, PinBuf* out) {
    bytes = std::max<size_t>(bytes, 1 << 16);
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        size_t best = g_ctx.free_pinned.size();          // the smallest pooled buffer that is big enough
        for (size_t i = 0; i < g_ctx.free_pinned.size(); ++i)
            if (g_ctx.free_pinned[i].bytes >= bytes && (best == g_ctx.free_pinned.size() || g_ctx.free_pinned[i].bytes < g_ctx.free_pinned[best].bytes))
                best = i;
        if (best != g_ctx.free_pinned.size()) {
            *out = g_ctx.free_pinned[best];
            g_ctx.free_pinned.erase(g_ctx.free_pinned.begin() + (long)best);
            return PM_OK;
        }
    }
    bytes += bytes / 4;
    out->bytes = bytes;
    HIPCHK(hipHostMalloc(&out->p, bytes, hipHostMallocDefault));
    return PM_OK;
}
static void give_pinned(PinBuf b) {
    if (!b.p) return;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        if (g_ctx.ready) {
            g_ctx.free_pinned.push_back(b);
            if (g_ctx.free_pinned.size() <= 8) return;
            size_t small = 0;                              // pool full: the smallest buffer goes
            for (size_t i = 1; i < g_ctx.free_pinned.size(); ++i) if (g_ctx.free_pinned[i].bytes < g_ctx.free_pinned[small].bytes) small = i;
            b = g_ctx.free_pinned[small];
            g_ctx.free_pinned.erase(g_ctx.free_pinned.begin() + (long)small);
        }
    }
    hipHostFree(b.p);
}

// One scan unit per classic index or per sub-index of a compact index.
struct Unit { const pm_index* ix; uint32_t slot, doc_base; bool prune; };
struct Group { int g, canon; uint32_t nh, slabs; std::vector<size_t> members; };

struct pm_result {
    // what was asked (kept for the one re-run after a hit-buffer overflow)
    std::vector<pm_index_t*> idx;
    pm_queries* q = nullptr;
    double threshold = 0;
    uint32_t nb_best = 0, slot_base = 0;
    // device output
    uint4* d_hits = nullptr;
    uint64_t cap = 0;
    uint64_t n_records = 0, n_runs = 0;
    pm_stats_t st{};
    std::vector<pm_launch_t> launches;
    // in-flight state
    Workspace* ws = nullptr;
    bool pending = false;
    int attempt = 0;
    size_t nev = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> lev;
    unsigned long long* h_fetch = nullptr;       // pinned [launch][kFetchShards] when "count_fetched" is on
    // ordered form (ensure_ordered): records permuted on the device into (slot, query) run order
    bool ordered = false;
    HitBuf d_ord{nullptr, 0};
    uint64_t n_out = 0;
    std::vector<std::pair<uint64_t, uint64_t>> fixups;   // [begin, end) of (slot, query) groups merged from several runs
    // host copy (pinned, pooled)
    PinBuf host{nullptr, 0};
    bool host_ready = false;
};

static void result_release(pm_result* r) {
    give_hit_buffer(HitBuf{r->d_hits, r->cap});
    r->d_hits = nullptr; r->cap = 0;
    give_hit_buffer(r->d_ord); r->d_ord = HitBuf{nullptr, 0};
    give_pinned(r->host); r->host = PinBuf{nullptr, 0};
    give_workspace(r->ws); r->ws = nullptr;
    if (r->h_fetch) { hipHostFree(r->h_fetch); r->h_fetch = nullptr; }
}

// Enqueues hash + scan launches + the read-back of the record counters on the compute
// stream; returns without waiting for the GPU.
static int enqueue_search(pm_result* r, uint64_t want_cap) {
    pm_queries* q = r->q;
    const size_t n_idx = r->idx.size();
    std::vector<Unit> units;
    for (size_t s = 0; s < n_idx; ++s) {
        const pm_index* ix = r->idx[s];
        if (ix->parts.empty()) {
            if (!ix->d_matrix) return fail(PM_EINVAL, "index %zu has no matrix", s);
            // names without '_' make the reference's post-filter raise on a line it would otherwise
            // drop (scripts/postprocess_cobs.py:10-18): such an index is cut on the host, where
            // every document that passed -t is seen, so both ways to prune fail alike
            units.push_back({ix, r->slot_base + (uint32_t)s, 0u, ix->names_have_sep});
        } else {
            // the n best documents of a compact index span its sub-indexes: cut when formatting
            for (size_t p = 0; p < ix->parts.size(); ++p) {
                const pm_index* part = ix->parts[p];
                if (part->info.n_docs == 0) continue;
                if (!part->d_matrix) return fail(PM_EINVAL, "index %zu has no matrix", s);
                units.push_back({part, r->slot_base + (uint32_t)s, (uint32_t)(p * ix->page_size * 8), false});
            }
        }
    }
    const size_t nq = q->headers.size();
    hipStream_t st = g_ctx.stream;
    { int urc = upload_queries(q); if (urc) return urc; }

    // ---- launch plan: one scan launch per (lanes-per-row class, canonicalize,
    // num_hashes) x counter-width class covers every unit of that class;
    // rows wider than 1024 B (column slabs) get a launch of their own.
    // Narrow rows (fewer than 32 lanes, i.e. at most 256 bytes) of all widths share one
    // mixed-width launch (g = 0): each of them is short, and separate launches would pay
    // one drain tail per width class.
    std::vector<Group> groups;
    for (size_t u = 0; u < units.size(); ++u) {
        const pm_index* ix = units[u].ix;
        const int key = (ix->slabs == 1 && (ix->g < 32 || g_single_launch)) ? 0 : ix->g;
        Group* gp = nullptr;
        if (ix->slabs == 1)
            for (auto& g : groups)
                if (g.slabs == 1 && g.g == key && g.canon == (int)ix->info.canonicalize && g.nh == ix->info.num_hashes) gp = &g;
        if (!gp) { groups.push_back({key, (int)ix->info.canonicalize, ix->info.num_hashes, ix->slabs, {}}); gp = &groups.back(); }
        gp->members.push_back(u);
    }
    for (auto& g : groups)      // a mixed group with one width is an ordinary group
        if (g.g == 0) {
            bool same = true;
            for (size_t u : g.members) same = same && units[u].ix->g == units[g.members[0]].ix->g;
            if (same) g.g = units[g.members[0]].ix->g;
        }
    const size_t n_units = units.size();

    // ---- workspace of this search (pooled, grow-only): counters, batch descriptors, events
    if (!r->ws) r->ws = take_workspace();
    Workspace* ws = r->ws;
    if (!ws->d_cnt) {
        HIPCHK(hipMalloc((void**)&ws->d_cnt, 4 * sizeof(unsigned long long)));
        HIPCHK(hipHostMalloc((void**)&ws->h_cnt, 4 * sizeof(unsigned long long), hipHostMallocMapped));
        HIPCHK(hipHostGetDevicePointer((void**)&ws->h_cnt_dev, ws->h_cnt, 0));
        HIPCHK(hipEventCreateWithFlags(&ws->done, hipEventDisableTiming));
    }
    if (ws->desc_cap < n_units) {
        if (ws->d_desc) hipFree(ws->d_desc);
        if (ws->h_desc) hipHostFree(ws->h_desc);
        ws->d_desc = nullptr; ws->h_desc = nullptr; ws->desc_cap = 0;
        ws->uploaded.clear();
        const size_t cap = std::max<size_t>(n_units, 64);
        HIPCHK(hipMalloc((void**)&ws->d_desc, 5 * cap * sizeof(BatchDesc)));
        HIPCHK(hipHostMalloc((void**)&ws->h_desc, 5 * cap * sizeof(BatchDesc), hipHostMallocDefault));
        ws->desc_cap = cap;
    }
    // host image of all five slices: base descriptors, then per query class the block ranges of mixed launches
    const size_t dcap = ws->desc_cap;
    memset(ws->h_desc, 0, 5 * dcap * sizeof(BatchDesc));
    {
        size_t o = 0;
        for (auto& g : groups)
            for (size_t u : g.members) {
                const pm_index* ix = units[u].ix;
                BatchDesc& d = ws->h_desc[o++];
                d.matrix = ix->d_matrix; d.stride = ix->info.stride; d.sig_size = ix->info.signature_size;
                d.barrett_m = barrett_m(ix->info.signature_size); d.n_docs = ix->info.n_docs;
                d.slot = units[u].slot; d.doc_base = units[u].doc_base; d.prune = units[u].prune ? 1u : 0u;
                d.lanes = (uint32_t)ix->g; d.block_begin = 0; d.pad_ = 0;
            }
    }
    std::vector<uint32_t> mixed_blocks(groups.size() * 4, 0u);
    {
        size_t desc_off = 0;
        for (size_t gi = 0; gi < groups.size(); ++gi) {
            Group& g = groups[gi];
            if (g.g == 0)
                for (int c = 0; c < 4; ++c) {
                    const uint32_t b = q->class_begin[c], e = q->class_begin[c + 1];
                    if (e == b) continue;
                    uint64_t blk = 0;
                    BatchDesc* stage = ws->h_desc + (size_t)(1 + c) * dcap + desc_off;
                    for (size_t k = 0; k < g.members.size(); ++k) {
                        stage[k] = ws->h_desc[desc_off + k];
                        const uint32_t qpb = scan_queries_per_block((int)stage[k].lanes);
                        stage[k].block_begin = (uint32_t)blk;
                        blk += (e - b + qpb - 1) / qpb;
                    }
                    if (blk > 0x7FFFFFFFull) return fail(PM_ERANGE, "launch grid too large");
                    mixed_blocks[gi * 4 + (size_t)c] = (uint32_t)blk;
                }
            desc_off += g.members.size();
        }
    }
    if (ws->uploaded.size() != 5 * dcap || memcmp(ws->uploaded.data(), ws->h_desc, 5 * dcap * sizeof(BatchDesc)) != 0) {
        HIPCHK(hipMemcpyAsync(ws->d_desc, ws->h_desc, 5 * dcap * sizeof(BatchDesc), hipMemcpyHostToDevice, st));
        ws->uploaded.assign(ws->h_desc, ws->h_desc + 5 * dcap);
    }
    // per-query minimum score, cached on the query set per threshold value
    if (nq && (!q->d_thr || q->thr_for != r->threshold)) {
        std::vector<uint32_t> thr(nq);
        for (size_t i = 0; i < nq; ++i) thr[i] = r->threshold == 0.0 ? 0u : pm_threshold_terms(r->threshold, q->n_terms[i]);
        if (!q->d_thr) HIPCHK(hipMalloc((void**)&q->d_thr, nq * 4));
        HIPCHK(hipStreamSynchronize(st));                 // an earlier search in flight may still read the old values
        HIPCHK(hipMemcpy(q->d_thr, thr.data(), nq * 4, hipMemcpyHostToDevice));
        q->thr_for = r->threshold;
    }
    // measurement option: sharded counters of the algorithmic bytes the scan really gathered
    size_t n_launch_max = 0;
    for (auto& g : groups) { (void)g; n_launch_max += 4; }
    if (g_count_fetched) {
        if (!g_ctx.d_fetch) HIPCHK(hipMalloc((void**)&g_ctx.d_fetch, kFetchShards * sizeof(unsigned long long)));
        if (r->h_fetch) { hipHostFree(r->h_fetch); r->h_fetch = nullptr; }
        HIPCHK(hipHostMalloc((void**)&r->h_fetch, std::max<size_t>(n_launch_max, 1) * kFetchShards * sizeof(unsigned long long), hipHostMallocDefault));
    }

    HitBuf hb{nullptr, 0};
    { int rc = take_hit_buffer(want_cap, &hb); if (rc) return rc; }
    r->d_hits = hb.p; r->cap = hb.cap;
    r->launches.clear(); r->lev.clear();
    q->epoch++;                       // hashes are part of the job: recomputed by every search
    r->nev = 0;
    { int rc = ws_event(ws, r->nev++, &r->ev0); if (rc) return rc; }
    { int rc = ws_event(ws, r->nev++, &r->ev1); if (rc) return rc; }
    { int rc = ws_event(ws, r->nev++, &r->ev2); if (rc) return rc; }
    HIPCHK(hipMemsetAsync(ws->d_cnt, 0, 4 * sizeof(unsigned long long), st));
    HIPCHK(hipEventRecord(r->ev0, st));
    uint64_t* d_h = nullptr;
    for (auto& g : groups) { int rc = ensure_hashes(q, g.canon, g.nh, &d_h); if (rc) return rc; }
    HIPCHK(hipEventRecord(r->ev1, st));
    uint64_t alg = 0;
    size_t desc_off = 0;
    for (auto& g : groups) {
        { int rc = ensure_hashes(q, g.canon, g.nh, &d_h); if (rc) return rc; }
        uint64_t rowsum = 0;
        for (size_t u : g.members) rowsum += units[u].ix->info.row_bytes;
        for (int c = 0; c < 4; ++c) {
            const uint32_t b = q->class_begin[c], e = q->class_begin[c + 1];
            if (e == b) continue;
            ScanArgs a;
            a.batches = ws->d_desc + desc_off; a.n_batches = (uint32_t)g.members.size();
            a.tiles = 0; a.total_blocks = 0;
            if (g.g > 0) {
                const uint32_t qpb = scan_queries_per_block(g.g);
                a.tiles = (e - b + qpb - 1) / qpb;
            } else {
                // mixed widths: the slice of this query class holds the per-batch workgroup ranges
                a.batches = ws->d_desc + (size_t)(1 + c) * dcap + desc_off;
                a.total_blocks = mixed_blocks[(size_t)(&g - groups.data()) * 4 + (size_t)c];
            }
            a.hashes = d_h; a.qd = q->d_qd; a.thr = q->d_thr; a.qmap = q->d_qmap + b; a.nq = e - b;
            a.prune_n = r->nb_best;
            a.bound = g_threshold_bound;
            a.nh = g.nh; a.hits = hb.p; a.hit_count = ws->d_cnt; a.hit_cap = hb.cap;
            a.fetch_count = g_count_fetched ? g_ctx.d_fetch : nullptr; a.fetch_shards = kFetchShards; a.pad_ = 0;
            a.runs = hb.p + hb.cap; a.run_cap = run_cap_of(hb.cap);
            if ((uint64_t)a.tiles * a.n_batches > 0x7FFFFFFFull)
                return fail(PM_ERANGE, "launch grid too large (%u tiles x %u batches)", a.tiles, a.n_batches);
            hipEvent_t es, ee;
            { int rc = ws_event(ws, r->nev++, &es); if (rc) return rc; }
            { int rc = ws_event(ws, r->nev++, &ee); if (rc) return rc; }
            if (a.fetch_count) HIPCHK(hipMemsetAsync(g_ctx.d_fetch, 0, kFetchShards * sizeof(unsigned long long), st));
            HIPCHK(hipEventRecord(es, st));
            HIPCHK(launch_scan(a, g.g, kPlaneClass[c], g.slabs, st));
            HIPCHK(hipEventRecord(ee, st));
            if (a.fetch_count)
                HIPCHK(hipMemcpyAsync(r->h_fetch + r->launches.size() * kFetchShards, g_ctx.d_fetch,
                                      kFetchShards * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
            r->lev.push_back({es, ee});
            uint64_t terms = 0;
            for (uint32_t i = b; i < e; ++i) terms += q->n_terms[q->qmap[i]];
            pm_launch_t L{};
            L.lanes_per_row = (uint32_t)g.g; L.planes = (uint32_t)kPlaneClass[c]; L.num_hashes = g.nh;
            L.n_batches = a.n_batches; L.n_queries = e - b;
            L.algorithmic_bytes = terms * g.nh * rowsum;
            r->launches.push_back(L);
            alg += L.algorithmic_bytes;
        }
        desc_off += g.members.size();
    }
    HIPCHK(hipEventRecord(r->ev2, st));
    // counters to the host by a one-thread kernel writing mapped pinned memory: no DMA engine on the
    // compute stream, so a large D2H of an earlier result (other stream) never delays this search
    HIPCHK(launch_publish(ws->d_cnt, ws->h_cnt_dev, 4, st));
    HIPCHK(hipEventRecord(ws->done, st));
    r->st.n_queries = nq; r->st.n_terms = q->total_terms;
    r->st.algorithmic_bytes = alg;
    r->st.n_scan_launches = (uint32_t)r->launches.size();
    r->pending = true;
    return PM_OK;
}

extern "C" int pm_result_wait(pm_result_t* r) {
    if (!r) return fail(PM_EINVAL, "bad argument");
    if (!r->pending) return PM_OK;
    NEED_DEV();
    for (;;) {
        HIPCHK(hipEventSynchronize(r->ws->done));
        const unsigned long long cnt = r->ws->h_cnt[0], runs = r->ws->h_cnt[1];
        if (cnt <= r->cap) {
            r->n_records = cnt; r->n_runs = runs;
            if (cnt > g_ctx.hit_hint) g_ctx.hit_hint = cnt;
            break;
        }
        // hit buffer too small: grow to the exact count and run the job again
        if (r->attempt >= 1) { r->pending = false; result_release(r); return fail(PM_EHIP, "hit count changed between runs"); }
        r->attempt++;
        hipFree(r->d_hits); r->d_hits = nullptr; r->cap = 0;
        int rc = enqueue_search(r, cnt);
        if (rc) { r->pending = false; result_release(r); return rc; }
    }
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, r->ev0, r->ev2)); r->st.ms_total = ms;
    HIPCHK(hipEventElapsedTime(&ms, r->ev0, r->ev1)); r->st.ms_hash = ms;
    double scan = 0;
    uint64_t fetched = 0;
    for (size_t i = 0; i < r->lev.size(); ++i) {
        HIPCHK(hipEventElapsedTime(&ms, r->lev[i].first, r->lev[i].second));
        r->launches[i].ms = ms; scan += ms;
        if (r->h_fetch) {
            uint64_t f = 0;
            for (uint32_t k = 0; k < kFetchShards; ++k) f += r->h_fetch[i * kFetchShards + k];
            r->launches[i].fetched_bytes = f; fetched += f;
        }
    }
    r->st.ms_scan = scan;
    r->st.fetched_bytes = fetched;
    r->st.n_records = r->n_records; r->st.n_runs = r->n_runs; r->st.n_hits = r->n_records - r->n_runs;
    r->pending = false;
    give_workspace(r->ws); r->ws = nullptr;       // events and counters have been read: the next search may take them
    return PM_OK;
}

extern "C" int pm_search_async(pm_index_t* const* idx, size_t n_idx, pm_queries_t* q,
                               double threshold, uint32_t nb_best_hits, uint32_t slot_base, pm_result_t** out) {
    NEED_DEV();
    if (!idx || !q || !out || n_idx == 0) return fail(PM_EINVAL, "bad argument");
    if (!(threshold >= 0.0)) return fail(PM_EINVAL, "threshold must be >= 0");
    for (size_t s = 0; s < n_idx; ++s) {
        if (!idx[s]) return fail(PM_EINVAL, "index %zu is null", s);
        if (idx[s]->info.term_size != q->k)
            return fail(PM_EINVAL, "index %zu has term_size %u but the queries were parsed for %u", s, idx[s]->info.term_size, q->k);
    }
    pm_result* r = new pm_result();
    r->idx.assign(idx, idx + n_idx);
    r->q = q; r->threshold = threshold; r->nb_best = nb_best_hits; r->slot_base = slot_base;
    const uint64_t want_cap = std::max<uint64_t>(std::max<uint64_t>(1u << 20, (uint64_t)q->headers.size() * 16),
                                                 g_ctx.hit_hint + g_ctx.hit_hint / 4);
    int rc = enqueue_search(r, want_cap);
    if (rc) {
        // whatever was queued before the failure must not outlive its buffers
        hipStreamSynchronize(g_ctx.stream);
        result_release(r);
        delete r;
        return rc;
    }
    *out = r;
    return PM_OK;
}

extern "C" int pm_search(pm_index_t* const* idx, size_t n_idx, pm_queries_t* q,
                         double threshold, uint32_t nb_best_hits, uint32_t slot_base, pm_result_t** out) {
    pm_result_t* r = nullptr;
    int rc = pm_search_async(idx, n_idx, q, threshold, nb_best_hits, slot_base, &r);
    if (rc) return rc;
    rc = pm_result_wait(r);
    if (rc) { delete r; return rc; }
    *out = r;
    return PM_OK;
}

#define RESULT_READY(r)                                            \
    do {                                                           \
        if ((r)->pending) {                                        \
            int rc_w_ = pm_result_wait(const_cast<pm_result_t*>(r)); \
            if (rc_w_) return rc_w_;                               \
        }                                                          \
    } while (0)

extern "C" int pm_result_stats(const pm_result_t* r, pm_stats_t* st) {
    if (!r || !st) return fail(PM_EINVAL, "bad argument");
    RESULT_READY(r);
    *st = r->st;
    return PM_OK;
}
extern "C" int pm_result_launches(const pm_result_t* r, pm_launch_t* out, size_t cap, size_t* n) {
    if (!r || !n) return fail(PM_EINVAL, "bad argument");
    RESULT_READY(r);
    *n = r->launches.size();
    if (out) for (size_t i = 0; i < r->launches.size() && i < cap; ++i) out[i] = r->launches[i];
    return PM_OK;
}
extern "C" int pm_result_hits_device(const pm_result_t* r, const void** dptr, uint64_t* n) {
    if (!r || !dptr || !n) return fail(PM_EINVAL, "bad argument");
    RESULT_READY(r);
    *dptr = r->d_hits; *n = r->n_records;
    return PM_OK;
}
static int ensure_ordered(pm_result* r);
extern "C" int pm_result_copy_hits_device(pm_result_t* r, void* dst, uint64_t capacity, int ordered) {
    NEED_DEV();
    if (!r) return fail(PM_EINVAL, "bad argument");
    RESULT_READY(r);
    if (ordered) { int rc = ensure_ordered(r); if (rc) return rc; }
    const uint64_t n = ordered ? r->n_out : r->n_records;
    const uint4* src = ordered ? r->d_ord.p : r->d_hits;
    if (!dst && n) return fail(PM_EINVAL, "bad argument");
    if (capacity < n) return fail(PM_EINVAL, "destination holds %llu records, need %llu",
                                  (unsigned long long)capacity, (unsigned long long)n);
    if (n) {
        HIPCHK(hipMemcpyAsync(dst, src, n * sizeof(uint4), hipMemcpyDeviceToDevice, g_ctx.d2h_stream));
        HIPCHK(hipStreamSynchronize(g_ctx.d2h_stream));
    }
    return PM_OK;
}

static inline bool hit_less(const pm_hit_t& a, const pm_hit_t& b) {
    if (a.slot != b.slot) return a.slot < b.slot;
    if (a.query != b.query) return a.query < b.query;
    const bool am = a.doc == PM_DOC_COUNT, bm = b.doc == PM_DOC_COUNT;
    if (am != bm) return am;                              // count records lead their (slot, query) run
    if (a.score != b.score) return a.score > b.score;     // score descending
    return a.doc < b.doc;                                 // then document index ascending
}

// Orders records by (slot, query, score desc, doc asc).  Large inputs: stable
// LSD radix passes on the (slot, query) key, then a comparison sort inside each
// (slot, query) run (runs are short: the hits of one query in one batch).
// General form, for records in any order (gathered from elsewhere, written by a caller).
static void order_hits(pm_hit_t* h, uint64_t n) {
    if (std::is_sorted(h, h + n, hit_less)) return;
    if (n < 4096) { std::sort(h, h + n, hit_less); return; }
    // dense key: slot * (max query + 1) + query, 12-bit digits
    uint32_t max_slot = 0, max_query = 0;
    for (uint64_t i = 0; i < n; ++i) { max_slot = std::max(max_slot, h[i].slot); max_query = std::max(max_query, h[i].query); }
    const uint64_t qspan = (uint64_t)max_query + 1;
    const uint64_t maxkey = (uint64_t)max_slot * qspan + max_query;      // < 2^64: both are 32-bit
    auto key = [qspan](const pm_hit_t& r) { return (uint64_t)r.slot * qspan + r.query; };
    std::vector<pm_hit_t> tmp((size_t)n);
    pm_hit_t* src = h; pm_hit_t* dst = tmp.data();
    constexpr int DB = 12;
    std::vector<uint64_t> cnt(1u << DB);
    for (int shift = 0; shift < 64 && (maxkey >> shift) != 0; shift += DB) {
        std::fill(cnt.begin(), cnt.end(), 0);
        for (uint64_t i = 0; i < n; ++i) cnt[(key(src[i]) >> shift) & ((1u << DB) - 1)]++;
        uint64_t sum = 0;
        for (auto& c : cnt) { uint64_t t = c; c = sum; sum += t; }
        for (uint64_t i = 0; i < n; ++i) dst[cnt[(key(src[i]) >> shift) & ((1u << DB) - 1)]++] = src[i];
        std::swap(src, dst);
    }
    if (src != h) memcpy(h, src, (size_t)n * sizeof(pm_hit_t));
    uint64_t b = 0;
    while (b < n) {
        uint64_t e = b + 1;
        while (e < n && h[e].slot == h[b].slot && h[e].query == h[b].query) ++e;
        if (e - b > 1) std::sort(h + b, h + e, hit_less);
        b = e;
    }
}

extern "C" void pm_hits_sort(pm_hit_t* hits, uint64_t n) {
    if (hits && n) order_hits(hits, n);
}

// a7 ordering.  k_scan wrote the records as runs {count record}{hits, best first, ties by
// document}, one per (query, slot[, column slab / sub-index]) with hits, in arbitrary run
// order, plus a directory entry {query, slot, first record, hits | cut flag} per run.  The
// records inside a run are already in cobs' line order, so only the RUNS need ordering: the
// host radix-sorts the directory by (slot, query) (16 bytes per run, not per record), turns
// it into a copy plan, and k_permute_runs moves every run to its final place in HBM.  The
// count record of a run that was not cut on the GPU carries no information (its count is
// the run length) and is dropped.  Several runs of one (slot, query) (rows wider than 1024
// bytes, compact sub-indexes) are laid out back to back and merged by score on the host.
struct RunEnt { uint32_t query, slot, begin, len; };      // len bit 31: the list was cut to the n best
static void sort_directory(std::vector<RunEnt>& dir) {
    auto key = [](const RunEnt& d) { return ((uint64_t)d.slot << 32) | d.query; };
    if (dir.size() < 2048) {
        std::sort(dir.begin(), dir.end(), [&](const RunEnt& a, const RunEnt& b) { return key(a) != key(b) ? key(a) < key(b) : a.begin < b.begin; });
        return;
    }
    // begin order first (cheap determinism for several runs of one key), then stable LSD passes on the key digits that vary
    uint64_t varies = 0;
    for (const RunEnt& d : dir) varies |= key(d) ^ key(dir[0]);
    std::vector<RunEnt> tmp(dir.size());
    std::vector<uint64_t> cnt(1u << 16);
    RunEnt* src = dir.data(); RunEnt* dst = tmp.data();
    auto pass = [&](auto digit) {
        std::fill(cnt.begin(), cnt.end(), 0);
        for (size_t k = 0; k < dir.size(); ++k) cnt[digit(src[k])]++;
        uint64_t sum = 0;
        for (auto& c : cnt) { uint64_t t = c; c = sum; sum += t; }
        for (size_t k = 0; k < dir.size(); ++k) dst[cnt[digit(src[k])]++] = src[k];
        std::swap(src, dst);
    };
    pass([](const RunEnt& d) { return d.begin & 0xFFFFu; });
    pass([](const RunEnt& d) { return d.begin >> 16; });
    for (int shift = 0; shift < 64; shift += 16) {
        if (((varies >> shift) & 0xFFFFull) == 0) continue;
        pass([&](const RunEnt& d) { return (uint32_t)((key(d) >> shift) & 0xFFFFu); });
    }
    if (src != dir.data()) memcpy(dir.data(), src, dir.size() * sizeof(RunEnt));
}

static std::mutex g_order_mu;
static int ensure_ordered(pm_result* r) {
    if (r->ordered) return PM_OK;
    std::lock_guard<std::mutex> lk(g_order_mu);
    if (r->ordered) return PM_OK;
    r->n_out = 0;
    r->fixups.clear();
    if (r->n_records == 0) { r->ordered = true; return PM_OK; }
    hipStream_t st = g_ctx.d2h_stream;        // never behind the kernels of a later search
    const uint64_t n_runs = r->n_runs;
    PinBuf stage{nullptr, 0};
    { int rc = take_pinned((size_t)n_runs * 2 * sizeof(uint4), &stage); if (rc) return rc; }
    auto done = [&](int rc) { give_pinned(stage); return rc; };
#define OCHK(expr)                                                                          \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) return done(fail(e_ == hipErrorOutOfMemory ? PM_ENOMEM : PM_EHIP, "%s: %s", #expr, hipGetErrorString(e_))); \
    } while (0)
    RunEnt* h_dir = (RunEnt*)stage.p;
    uint4* h_plan = (uint4*)stage.p + n_runs;
    OCHK(hipMemcpyAsync(h_dir, r->d_hits + r->cap, n_runs * sizeof(uint4), hipMemcpyDeviceToHost, st));
    OCHK(hipStreamSynchronize(st));
    std::vector<RunEnt> dir(h_dir, h_dir + n_runs);
    sort_directory(dir);
    uint64_t o = 0;
    size_t np = 0, k = 0;
    while (k < dir.size()) {
        size_t e = k + 1;
        while (e < dir.size() && dir[e].slot == dir[k].slot && dir[e].query == dir[k].query) ++e;
        if (e == k + 1) {
            const RunEnt& d = dir[k];
            const uint32_t len = d.len & 0x7FFFFFFFu;
            const bool cut = (d.len >> 31) != 0;             // cut on the GPU: the count record stays
            const uint32_t n = len + (cut ? 1u : 0u);
            h_plan[np++] = make_uint4(d.begin + (cut ? 0u : 1u), (uint32_t)o, n, 0u);
            o += n;
        } else {
            const uint64_t first = o;                        // several runs of one (slot, query)
            for (size_t j = k; j < e; ++j) {
                const uint32_t len = dir[j].len & 0x7FFFFFFFu;
                h_plan[np++] = make_uint4(dir[j].begin + 1u, (uint32_t)o, len, 0u);
                o += len;
            }
            r->fixups.push_back({first, o});
        }
        k = e;
    }
    r->n_out = o;
    { int rc = take_hit_buffer(std::max<uint64_t>(std::max<uint64_t>(o, 2 * (uint64_t)np), 1), &r->d_ord); if (rc) return done(rc); }
    // the plan travels in the (unused) directory part of the destination buffer
    uint4* d_plan = r->d_ord.p + r->d_ord.cap;
    if (np > run_cap_of(r->d_ord.cap)) return done(fail(PM_EHIP, "run directory larger than its bound"));
    OCHK(hipMemcpyAsync(d_plan, h_plan, np * sizeof(uint4), hipMemcpyHostToDevice, st));
    OCHK(launch_permute_runs(d_plan, (uint32_t)np, r->d_hits, r->d_ord.p, st));
    OCHK(hipStreamSynchronize(st));
    // groups merged from several runs: interleave by score on the host, write back
    for (auto& f : r->fixups) {
        std::vector<pm_hit_t> tmp((size_t)(f.second - f.first));
        OCHK(hipMemcpy(tmp.data(), r->d_ord.p + f.first, tmp.size() * sizeof(pm_hit_t), hipMemcpyDeviceToHost));
        std::sort(tmp.begin(), tmp.end(), hit_less);
        OCHK(hipMemcpy(r->d_ord.p + f.first, tmp.data(), tmp.size() * sizeof(pm_hit_t), hipMemcpyHostToDevice));
    }
#undef OCHK
    r->ordered = true;
    return done(PM_OK);
}

extern "C" int pm_result_ordered_device(pm_result_t* r, const void** dptr, uint64_t* n) {
    NEED_DEV();
    if (!r || !dptr || !n) return fail(PM_EINVAL, "bad argument");
    RESULT_READY(r);
    { int rc = ensure_ordered(r); if (rc) return rc; }
    *dptr = r->d_ord.p; *n = r->n_out;
    return PM_OK;
}

extern "C" int pm_result_hits_into(const pm_result_t* r_, pm_hit_t* out, uint64_t capacity, uint64_t* n_out) {
    NEED_DEV();
    pm_result_t* r = const_cast<pm_result_t*>(r_);
    if (!r || !n_out) return fail(PM_EINVAL, "bad argument");
    RESULT_READY(r);
    { int rc = ensure_ordered(r); if (rc) return rc; }
    if (!out && r->n_out) return fail(PM_EINVAL, "bad argument");
    if (capacity < r->n_out) return fail(PM_EINVAL, "destination holds %llu records, need %llu",
                                         (unsigned long long)capacity, (unsigned long long)r->n_out);
    if (r->n_out) {
        HIPCHK(hipMemcpyAsync(out, r->d_ord.p, r->n_out * sizeof(pm_hit_t), hipMemcpyDeviceToHost, g_ctx.d2h_stream));
        HIPCHK(hipStreamSynchronize(g_ctx.d2h_stream));
    }
    *n_out = r->n_out;
    return PM_OK;
}

extern "C" int pm_result_hits_host(pm_result_t* r, const pm_hit_t** hits, uint64_t* n) {
    NEED_DEV();
    if (!r || !hits || !n) return fail(PM_EINVAL, "bad argument");
    RESULT_READY(r);
    if (!r->host_ready) {
        { int rc = ensure_ordered(r); if (rc) return rc; }
        { int rc = take_pinned((size_t)r->n_out * sizeof(pm_hit_t), &r->host); if (rc) return rc; }
        if (r->n_out) {
            HIPCHK(hipMemcpyAsync(r->host.p, r->d_ord.p, r->n_out * sizeof(pm_hit_t), hipMemcpyDeviceToHost, g_ctx.d2h_stream));
            HIPCHK(hipStreamSynchronize(g_ctx.d2h_stream));
        }
        r->host_ready = true;
    }
    *hits = (const pm_hit_t*)r->host.p; *n = r->n_out;
    return PM_OK;
}
extern "C" void pm_result_free(pm_result_t* r) {
    if (!r) return;
    bind_thread_quiet();
    if (r->pending && r->ws) hipEventSynchronize(r->ws->done);      // the GPU may still write into the buffers
    result_release(r);
    delete r;
}

// --------------------------------------------------------------------- text
// cobs stdout grammar (witnesses: scripts/postprocess_cobs.py:23-26, :10-13;
// scripts/filter_queries.py:51-65): "*<header>\t<N>\n" then N lines
// "<doc name>\t<score>\n", best score first, ties by document index.
// nb_best_hits >= 0 fuses scripts/postprocess_cobs.py:16-39: header untouched,
// each name cut to "_" + what follows its first '_', the first n lines kept plus
// later lines whose score equals the n-th score.
static inline void append_tab_uint_nl(std::string& out, uint64_t v) {     // "\t<v>\n"
    char buf[24]; int n = 0;
    do { buf[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    out.push_back('\t');
    while (n) out.push_back(buf[--n]);
    out.push_back('\n');
}

// formats the records of queries [qa, qb) (a slice of one slot's ordered records) into `out`;
// returns PM_OK or an error code with the message in `err`
static int format_query_range(const pm_index* ix, const pm_queries* q, const pm_hit_t* mine, size_t n_mine,
                              size_t qa, size_t qb, int64_t nb_best, std::string& out, std::string& err) {
    char msg[512];
    size_t p = (size_t)(std::lower_bound(mine, mine + n_mine, (uint32_t)qa,
                                         [](const pm_hit_t& h, uint32_t v) { return h.query < v; }) - mine);
    for (size_t qi = qa; qi < qb; ++qi) {
        size_t e = p;
        while (e < n_mine && mine[e].query == qi) ++e;
        size_t total = e - p;
        if (p < e && mine[p].doc == PM_DOC_COUNT) {
            // count records lead the run: cut on the GPU (one record: the number of documents that
            // passed -t) or raw device runs of several column slabs / sub-indexes (they add up)
            total = 0;
            while (p < e && mine[p].doc == PM_DOC_COUNT) { total += mine[p].score; ++p; }
        }
        if (!q->headerless[qi]) out.push_back('*');
        else if (nb_best >= 0) {    // the post-filter needs a '*' line first (postprocess_cobs.py:23-29 raises)
            snprintf(msg, sizeof msg, "record %zu has sequence lines before any FASTA header: the post-filter cannot parse its result", qi);
            err = msg;
            return PM_EINVAL;
        }
        out += q->headers[qi];
        append_tab_uint_nl(out, total);
        uint32_t min_kmers = 0;
        for (size_t i = p; i < e; ++i) {
            if (mine[i].doc >= ix->info.n_docs) {
                snprintf(msg, sizeof msg, "hit record (query %u, doc %u) out of range for this index", mine[i].query, mine[i].doc);
                err = msg;
                return PM_EINVAL;
            }
            const char* nm = ix->names_blob.data() + ix->name_off[mine[i].doc];
            const size_t nl = (size_t)(ix->name_off[mine[i].doc + 1] - ix->name_off[mine[i].doc] - 1);
            if (nb_best < 0) {
                out.append(nm, nl);
                append_tab_uint_nl(out, mine[i].score);
                continue;
            }
            const int64_t rank = (int64_t)(i - p) + 1;      // 1-based like the post-filter's counter
            const char* us = (const char*)memchr(nm, '_', nl);
            if (!us) {
                // postprocess_cobs.py:16-18 turns such a line into a bare "_" (no newline) and
                // raises on int("_") once rank >= n: an error for the whole rule
                if (rank < nb_best) { out.push_back('_'); continue; }
                snprintf(msg, sizeof msg, "document name '%.*s' has no '_' separator (post-filter cannot parse it)", (int)nl, nm);
                err = msg;
                return PM_EINVAL;
            }
            bool keep;
            if (rank < nb_best) keep = true;
            else if (rank == nb_best) { keep = true; min_kmers = mine[i].score; }
            else keep = mine[i].score == min_kmers;
            if (keep) {
                out.append(us, nl - (size_t)(us - nm));
                append_tab_uint_nl(out, mine[i].score);
            }
        }
        p = e;
    }
    return PM_OK;
}

extern "C" int pm_format_hits(const pm_index_t* ix, const pm_queries_t* q,
                              const pm_hit_t* hits, uint64_t n_hits, uint32_t slot,
                              int64_t nb_best, char** text, size_t* len) {
    if (!ix || !q || (!hits && n_hits) || !text || !len) return fail(PM_EINVAL, "bad argument");
    const size_t nq = q->headers.size();
    // records as pm_result_hits_* deliver them are already in line order: the slot's records are
    // one contiguous slice; anything else (gathered, hand-made) is copied and ordered first
    std::vector<pm_hit_t> copy;
    const pm_hit_t* mine = hits;
    size_t n_mine = 0;
    if (std::is_sorted(hits, hits + n_hits, [](const pm_hit_t& a, const pm_hit_t& b) { return hit_less(a, b); })) {
        const pm_hit_t* lo = std::lower_bound(hits, hits + n_hits, slot, [](const pm_hit_t& h, uint32_t v) { return h.slot < v; });
        const pm_hit_t* hi = std::upper_bound(lo, hits + n_hits, slot, [](uint32_t v, const pm_hit_t& h) { return v < h.slot; });
        mine = lo; n_mine = (size_t)(hi - lo);
    } else {
        for (uint64_t i = 0; i < n_hits; ++i) if (hits[i].slot == slot) copy.push_back(hits[i]);
        order_hits(copy.data(), copy.size());
        mine = copy.data(); n_mine = copy.size();
    }
    if (n_mine && mine[n_mine - 1].query >= nq)
        return fail(PM_EINVAL, "hit record (query %u) out of range for this query set", mine[n_mine - 1].query);
    // query ranges are independent: format them on several host threads (at 1 M queries a single
    // thread spends seconds per batch here, scripts/postprocess_cobs.py far more)
    size_t nt = std::min<size_t>(std::min<size_t>(std::thread::hardware_concurrency(), 16), (nq + n_mine / 8) / 4096);
    if (nt < 1) nt = 1;
    std::vector<std::string> parts(nt), errs(nt);
    std::vector<int> rcs(nt, PM_OK);
    // split by records + queries so that long hit lists spread evenly
    std::vector<size_t> cutq(nt + 1, nq);
    cutq[0] = 0;
    for (size_t t = 1; t < nt; ++t) {
        const size_t target = (n_mine + nq) * t / nt;          // position in the merged (records + headers) stream
        size_t lo = cutq[t - 1], hi = nq;                        // smallest query whose prefix weight reaches target
        while (lo < hi) {
            const size_t mid = (lo + hi) / 2;
            const size_t recs = (size_t)(std::lower_bound(mine, mine + n_mine, (uint32_t)mid,
                                                          [](const pm_hit_t& h, uint32_t v) { return h.query < v; }) - mine);
            if (recs + mid < target) lo = mid + 1; else hi = mid;
        }
        cutq[t] = lo;
    }
    auto work = [&](size_t t) {
        parts[t].reserve((size_t)((double)(n_mine * 28 + nq * 24) / (double)nt * 1.1) + 64);
        rcs[t] = format_query_range(ix, q, mine, n_mine, cutq[t], cutq[t + 1], nb_best, parts[t], errs[t]);
    };
    if (nt == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (size_t t = 1; t < nt; ++t) th.emplace_back(work, t);
        work(0);
        for (auto& x : th) x.join();
    }
    for (size_t t = 0; t < nt; ++t)
        if (rcs[t] != PM_OK) return fail(rcs[t], "%s", errs[t].c_str());      // the first failing query range, as a serial pass would report
    size_t total = 0;
    for (auto& s2 : parts) total += s2.size();
    char* buf = (char*)malloc(total + 1);
    if (!buf) return fail(PM_ENOMEM, "out of host memory");
    size_t o = 0;
    for (auto& s2 : parts) { memcpy(buf + o, s2.data(), s2.size()); o += s2.size(); }
    buf[total] = 0;
    *text = buf; *len = total;
    return PM_OK;
}

extern "C" int pm_query_text(pm_index_t* ix, const char* fasta, size_t fasta_len,
                             double threshold, int64_t nb_best, char** text, size_t* len) {
    NEED_DEV();
    if (!ix) return fail(PM_EINVAL, "bad argument");
    pm_queries_t* q = nullptr; pm_result_t* r = nullptr;
    int rc = pm_queries_parse(fasta, fasta_len, ix->info.term_size, &q);
    if (rc) return rc;
    uint64_t nq = 0; pm_queries_count(q, &nq, nullptr);
    const pm_hit_t* hits = nullptr; uint64_t n = 0;
    if (nq) {
        pm_index_t* arr[1] = {ix};
        rc = pm_search(arr, 1, q, threshold, nb_best > 0 ? (uint32_t)std::min<int64_t>(nb_best, 0xFFFFFFFFll) : 0u, 0, &r);
        if (rc == PM_OK) rc = pm_result_hits_host(r, &hits, &n);
    }
    if (rc == PM_OK) rc = pm_format_hits(ix, q, hits, n, 0, nb_best, text, len);
    if (r) pm_result_free(r);
    pm_queries_free(q);
    return rc;
}

// --------------------------------------------------------- 04_filter merge
// Native form of the reference's consumer (scripts/filter_queries.py:107-206):
// for every query keep the globally best `keep` matches across batches plus the
// ones tied with the last of them, ordered by (-kmers, batch, ref), and emit
// ">qname ref1,ref2,...\nseq".  What is merged per batch is what the 03_match
// file of that batch holds, i.e. the hit list after the per-batch post-filter
// (scripts/postprocess_cobs.py:21-39 with -n nb_best_hits).
struct MergeItem { uint32_t kmers; uint32_t batch; std::string ref; };
struct pm_merge {
    const pm_queries* q = nullptr;
    uint32_t keep = 0;
    std::vector<std::string> batches;
    std::map<std::string, uint32_t> by_name;        // query name (first word) -> record index
    std::vector<std::string> qnames;
    std::vector<std::vector<MergeItem>> items;
    std::vector<uint32_t> floor_;
};

extern "C" int pm_merge_create(const pm_queries_t* q, uint32_t keep, pm_merge_t** out) {
    if (!q || !out) return fail(PM_EINVAL, "bad argument");
    pm_merge* m = new pm_merge();
    m->q = q; m->keep = keep;
    const size_t nq = q->headers.size();
    m->items.resize(nq); m->floor_.assign(nq, 0); m->qnames.resize(nq);
    for (size_t i = 0; i < nq; ++i) {
        // readfq name: the header up to its first space (scripts/filter_queries.py:80)
        const std::string& h = q->headers[i];
        m->qnames[i] = h.substr(0, h.find(' '));
        m->by_name[m->qnames[i]] = (uint32_t)i;        // duplicates: the last record wins, as in a dict
    }
    *out = m;
    return PM_OK;
}

extern "C" int pm_merge_add(pm_merge_t* m, const char* batch, const pm_index_t* ix,
                            const pm_hit_t* hits, uint64_t n_hits, uint32_t slot, int64_t nb_best) {
    if (!m || !batch || !ix || (!hits && n_hits)) return fail(PM_EINVAL, "bad argument");
    const size_t nq = m->q->headers.size();
    std::vector<pm_hit_t> mine;
    for (uint64_t i = 0; i < n_hits; ++i)
        if (hits[i].slot == slot && hits[i].doc != PM_DOC_COUNT) {
            if (hits[i].query >= nq || hits[i].doc >= ix->info.n_docs)
                return fail(PM_EINVAL, "hit record out of range for batch %s", batch);
            mine.push_back(hits[i]);
        }
    order_hits(mine.data(), mine.size());
    const uint32_t bid = (uint32_t)m->batches.size();
    m->batches.push_back(batch);
    size_t p = 0;
    while (p < mine.size()) {
        size_t e = p;
        const uint32_t qi = mine[p].query;
        while (e < mine.size() && mine[e].query == qi) ++e;
        // the 03_match header is "*<header>\tN": the consumer looks the query up by the text
        // before the first TAB, cut at the first space (scripts/filter_queries.py:58-59)
        const std::string& h = m->q->headers[qi];
        std::string key = h.substr(0, h.find('\t'));
        key = key.substr(0, key.find(' '));
        auto it = m->by_name.find(key);
        if (it == m->by_name.end()) return fail(PM_EINVAL, "query '%s' of batch %s is not in the query file", key.c_str(), batch);
        const uint32_t target = it->second;
        std::vector<MergeItem>& v = m->items[target];
        uint32_t nth = 0;
        for (size_t i = p; i < e; ++i) {
            if (nb_best >= 0) {                          // per-batch post-filter, same rule as pm_format_hits
                const int64_t rank = (int64_t)(i - p) + 1;
                if (rank == nb_best) nth = mine[i].score;
                if (rank > nb_best && mine[i].score != nth) continue;
            }
            const char* nm = ix->names_blob.data() + ix->name_off[mine[i].doc];
            const size_t nl = (size_t)(ix->name_off[mine[i].doc + 1] - ix->name_off[mine[i].doc] - 1);
            const char* us = (const char*)memchr(nm, '_', nl);
            if (!us || memchr(us + 1, '_', nl - (size_t)(us + 1 - nm)))
                return fail(PM_EINVAL, "document name '%.*s' must hold exactly one '_' (scripts/filter_queries.py:64)", (int)nl, nm);
            if (mine[i].score >= m->floor_[target])
                v.push_back({mine[i].score, bid, std::string(us + 1, nl - (size_t)(us + 1 - nm))});
        }
        std::sort(v.begin(), v.end(), [&](const MergeItem& a, const MergeItem& b) {
            if (a.kmers != b.kmers) return a.kmers > b.kmers;
            if (a.batch != b.batch) return m->batches[a.batch] < m->batches[b.batch];
            return a.ref < b.ref;
        });
        if (v.size() > m->keep) {
            if (m->keep == 0) return fail(PM_EINVAL, "keep = 0 is not supported by the 04_filter rule");
            size_t cut = m->keep;
            m->floor_[target] = v[cut - 1].kmers;
            while (cut < v.size() && v[cut].kmers == m->floor_[target]) ++cut;
            v.resize(cut);
        }
        p = e;
    }
    return PM_OK;
}

extern "C" int pm_merge_emit(const pm_merge_t* m, char** text, size_t* len) {
    if (!m || !text || !len) return fail(PM_EINVAL, "bad argument");
    std::string out;
    const pm_queries* q = m->q;
    // dict semantics of the consumer: one record per distinct name, at the position of its
    // first occurrence, with the sequence of its last occurrence
    std::vector<char> seen(q->headers.size(), 0);
    for (size_t i = 0; i < q->headers.size(); ++i) {
        const uint32_t rec = m->by_name.at(m->qnames[i]);
        if (seen[rec]) continue;
        seen[rec] = 1;
        out.push_back('>'); out += m->qnames[i]; out.push_back(' ');
        const std::vector<MergeItem>& v = m->items[rec];
        for (size_t k = 0; k < v.size(); ++k) { if (k) out.push_back(','); out += v[k].ref; }
        out.push_back('\n');
        out.append(q->seqs, (size_t)q->seq_off[rec], (size_t)(q->seq_off[rec + 1] - q->seq_off[rec]));
        out.push_back('\n');
    }
    char* buf = (char*)malloc(out.size() + 1);
    if (!buf) return fail(PM_ENOMEM, "out of host memory");
    memcpy(buf, out.data(), out.size()); buf[out.size()] = 0;
    *text = buf; *len = out.size();
    return PM_OK;
}

extern "C" void pm_merge_free(pm_merge_t* m) { delete m; }

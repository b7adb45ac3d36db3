// pm_index.cpp -- index residency: COBS classic / compact header
// readers, streaming upload with re-striding, synthetic 661k-shaped indexes, planted content.
#include "pm_host.h"
#include <chrono>
#include <ctime>

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

// PM_EFORMAT unless `rows` rows of `stride` bytes are a size this GPU could ever hold
static int check_matrix_size(uint64_t rows, uint64_t stride) {
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) != hipSuccess) return fail(PM_EHIP, "hipMemGetInfo failed");
    if (stride == 0 || rows > (uint64_t)tot / stride)
        return fail(PM_EFORMAT, "index header asks for %llu rows of %llu bytes: more than this GPU's %llu bytes of HBM (corrupt header?)",
                    (unsigned long long)rows, (unsigned long long)stride, (unsigned long long)tot);
    return PM_OK;
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
    // signature_size comes straight from the file: no row count may make `rows x stride` wrap (a crafted header with
    // signature_size >= 2^60 would otherwise get a tiny allocation and the upload would write past it) or exceed HBM
    {
        int rc_sz = check_matrix_size(h.sig, sa);
        if (rc_sz) return rc_sz;
    }
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
    // k_scan's lane groups of 8+ lanes exchange 32-bit ROW INDICES (pm_kernels.hip, SHARE).  8+ lanes mean a stride of more
    // than 64 bytes, so 2^32 rows would be a matrix of 275 GB+: check_matrix_size above already refused anything this GPU
    // cannot hold.  Stated here as a check of its own so that a device with more memory fails loudly instead of wrapping.
    if (ix->g >= 8 && h.sig > 0xFFFFFFFFull)
        return fail(PM_ERANGE, "index has %llu rows of %llu bytes: the scan addresses at most 2^32 rows of this width",
                    (unsigned long long)h.sig, (unsigned long long)stride);
    const auto t_m0 = std::chrono::steady_clock::now();
    hipError_t e = device_malloc_reclaim((void**)&ix->d_matrix, in.device_bytes);   // idle pooled buffers go first
    if (getenv("PM_LOAD_TRACE"))
        fprintf(stderr, "[pm_load] hipMalloc %.2f GB: %.1f ms\n", in.device_bytes / 1e9,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_m0).count());
    if (e != hipSuccess || !ix->d_matrix) {
        ix->d_matrix = nullptr;
        return fail(PM_ENOMEM, "hipMalloc(%llu bytes) for the signature matrix failed: %s",
                    (unsigned long long)in.device_bytes, e != hipSuccess ? hipGetErrorString(e) : "null pointer");
    }
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
    // decode-once cache: everything read from `fd` is also written here (pm_index_load_fd_tee)
    int tee_fd = -1; int tee_errno = 0; uint64_t tee_bytes = 0;
    void tee(const uint8_t* p, size_t n) {
        while (n && tee_fd >= 0 && !tee_errno) {
            ssize_t w = ::write(tee_fd, p, n);
            if (w < 0) { if (errno == EINTR) continue; tee_errno = errno; return; }
            p += w; n -= (size_t)w; tee_bytes += (uint64_t)w;
        }
    }
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
            if (tee_fd >= 0) tee(dst + got, (size_t)r);
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
            if (hbuf[i]) (void)hipHostFree(hbuf[i]);
            if (dbuf[i]) (void)hipFree(dbuf[i]);
            if (ev[i]) (void)hipEventDestroy(ev[i]);
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

// The same for a REGULAR FILE (a decompressed index on disk or in the page cache: rule decompress_cobs,
// Snakefile:364-387, index_load_mode mem-disk): a single read() loop tops out at the speed of one memcpy out of the
// page cache (~20 GB/s), a third of PCIe Gen5.  Here `nthreads` workers pread() alternate chunks into their own pinned
// double buffers and queue H2D + re-stride on their own streams, so several copies are in flight.
// staging of one reader thread: two pinned chunks, two device chunks, their events and a stream.  Pinning memory is slow
// (tens of ms per buffer), a whole-stage run loads dozens of indexes: the sets are pooled for the life of the library.
struct StageSet { uint8_t* hbuf[2]; uint8_t* dbuf[2]; hipEvent_t ev[2]; hipStream_t st; };
static std::mutex g_stage_mu;
static std::vector<StageSet> g_stage_pool;
static constexpr size_t kStageBytes = 32ull << 20;
static hipError_t take_stage(StageSet* out) {
    {
        std::lock_guard<std::mutex> lk(g_stage_mu);
        if (!g_stage_pool.empty()) { *out = g_stage_pool.back(); g_stage_pool.pop_back(); return hipSuccess; }
    }
    StageSet s{};
    hipError_t e = hipStreamCreateWithFlags(&s.st, hipStreamNonBlocking);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) {
        e = hipHostMalloc((void**)&s.hbuf[i], kStageBytes, hipHostMallocDefault);
        if (e == hipSuccess) e = hipMalloc((void**)&s.dbuf[i], kStageBytes);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s.ev[i], hipEventDisableTiming);
    }
    if (e != hipSuccess) {
        for (int i = 0; i < 2; ++i) {
            if (s.hbuf[i]) (void)hipHostFree(s.hbuf[i]);
            if (s.dbuf[i]) (void)hipFree(s.dbuf[i]);
            if (s.ev[i]) (void)hipEventDestroy(s.ev[i]);
        }
        if (s.st) (void)hipStreamDestroy(s.st);
        return e;
    }
    *out = s;
    return hipSuccess;
}
static void give_stage(const StageSet& s) {
    {
        std::lock_guard<std::mutex> lk(g_stage_mu);
        if (g_ctx.ready && g_stage_pool.size() < 16) { g_stage_pool.push_back(s); return; }
    }
    for (int i = 0; i < 2; ++i) { (void)hipHostFree(s.hbuf[i]); (void)hipFree(s.dbuf[i]); (void)hipEventDestroy(s.ev[i]); }
    (void)hipStreamDestroy(s.st);
}
void release_stage_pool() {                    // pm_shutdown
    std::lock_guard<std::mutex> lk(g_stage_mu);
    for (auto& s : g_stage_pool) {
        for (int i = 0; i < 2; ++i) { (void)hipHostFree(s.hbuf[i]); (void)hipFree(s.dbuf[i]); (void)hipEventDestroy(s.ev[i]); }
        (void)hipStreamDestroy(s.st);
    }
    g_stage_pool.clear();
}

static int stream_matrix_file(int fd, uint64_t file_off, pm_index* ix, uint64_t rb, uint64_t S) {
    const uint64_t stride = ix->info.stride;
    if (rb > kStageBytes) return fail(PM_ERANGE, "row of %llu bytes is wider than a staging chunk", (unsigned long long)rb);
    const uint64_t chunk_rows = std::max<uint64_t>(1, kStageBytes / rb);
    const uint64_t n_chunks = (S + chunk_rows - 1) / chunk_rows;
    int want = 6;                                               // PM_LOAD_THREADS: readers per index file
    if (const char* env = getenv("PM_LOAD_THREADS")) want = std::max(1, std::min(32, atoi(env)));
    const int nthreads = (int)std::min<uint64_t>((uint64_t)want, n_chunks);
    std::vector<int> rcs((size_t)nthreads, PM_OK);
    std::vector<std::string> errs((size_t)nthreads);
    const int device = g_ctx.device;
    // PM_LOAD_TRACE: where a load spends its time -- pread out of the page cache vs waiting for the staging buffer's H2D
    const bool trace = getenv("PM_LOAD_TRACE") != nullptr;
    std::vector<double> t_read((size_t)nthreads, 0.0), t_wait((size_t)nthreads, 0.0);
    const auto t_all0 = std::chrono::steady_clock::now();
    auto secs = [](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count(); };
    auto worker = [&](int t) {
        char msg[256];
        hipError_t e = hipSetDevice(device);
        StageSet sg{};
        bool have = false;
        if (e == hipSuccess) { e = take_stage(&sg); have = e == hipSuccess; }
        bool used[2] = {false, false}; int cur = 0;
        for (uint64_t c = (uint64_t)t; c < n_chunks && e == hipSuccess && rcs[(size_t)t] == PM_OK; c += (uint64_t)nthreads) {
            const uint64_t row = c * chunk_rows, nrows = std::min<uint64_t>(chunk_rows, S - row);
            const size_t nbytes = (size_t)(nrows * rb);
            auto t0 = std::chrono::steady_clock::now();
            if (used[cur]) { e = hipEventSynchronize(sg.ev[cur]); if (e != hipSuccess) break; }
            if (trace) { t_wait[(size_t)t] += secs(t0); t0 = std::chrono::steady_clock::now(); }
            size_t got = 0;
            while (got < nbytes) {
                ssize_t r = pread(fd, sg.hbuf[cur] + got, nbytes - got, (off_t)(file_off + row * rb + got));
                if (r < 0) { if (errno == EINTR) continue; break; }
                if (r == 0) break;
                got += (size_t)r;
            }
            if (trace) t_read[(size_t)t] += secs(t0);
            if (got != nbytes) {
                snprintf(msg, sizeof msg, "index file ended after %llu of %llu matrix bytes",
                         (unsigned long long)(row * rb + got), (unsigned long long)(S * rb));
                errs[(size_t)t] = msg; rcs[(size_t)t] = PM_EIO;
                break;
            }
            e = hipMemcpyAsync(sg.dbuf[cur], sg.hbuf[cur], nbytes, hipMemcpyHostToDevice, sg.st);
            if (e == hipSuccess) e = launch_restride(sg.dbuf[cur], rb, ix->d_matrix + row * stride, stride, nrows, sg.st);
            if (e == hipSuccess) e = hipEventRecord(sg.ev[cur], sg.st);
            used[cur] = true; cur ^= 1;
        }
        if (have) {
            hipError_t e2 = hipStreamSynchronize(sg.st);          // nothing of this load may still use the set
            if (e == hipSuccess) e = e2;
            give_stage(sg);
        }
        if (e != hipSuccess && rcs[(size_t)t] == PM_OK) {
            snprintf(msg, sizeof msg, "index upload: %s", hipGetErrorString(e)); errs[(size_t)t] = msg; rcs[(size_t)t] = PM_EHIP;
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nthreads; ++t) th.emplace_back(worker, t);
    worker(0);
    for (auto& x : th) x.join();
    if (trace) {
        double r = 0, w = 0;
        for (int t = 0; t < nthreads; ++t) { r += t_read[(size_t)t]; w += t_wait[(size_t)t]; }
        const double wall = secs(t_all0);
        fprintf(stderr, "[pm_load] %.2f GB in %.3f s = %.1f GB/s; %d readers: pread %.3f s, waiting for a staging buffer %.3f s (mean per reader)\n",
                S * rb / 1e9, wall, S * rb / 1e9 / wall, nthreads, r / nthreads, w / nthreads);
    }
    for (int t = 0; t < nthreads; ++t)
        if (rcs[(size_t)t] != PM_OK) return fail(rcs[(size_t)t], "%s", errs[(size_t)t].c_str());
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
        struct stat fst;
        const off_t pos = (rd.fd >= 0 && !rd.mem) ? lseek(rd.fd, 0, SEEK_CUR) : (off_t)-1;
        // (a row wider than one pooled staging chunk -- more than 2^28 documents -- takes the serial path, whose buffers
        // are sized to the row)
        if (pos >= (off_t)head.size() && fstat(rd.fd, &fst) == 0 && S_ISREG(fst.st_mode) && S * rb >= (256ull << 20) &&
            rb <= kStageBytes) {
            // seekable: several readers in parallel (the index began at pos - head.size() of the file)
            rc = stream_matrix_file(rd.fd, (uint64_t)(pos - (off_t)head.size()) + h.data_off, ix, rb, S);
        } else {
            rd.pending.assign(head.begin() + (long)h.data_off, head.end());
            rd.pend_pos = 0;
            rc = stream_matrix(rd, ix, rb, S);
        }
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
            rc = check_matrix_size(pc.sig[p], sa);
            if (!rc && part->g >= 8 && pc.sig[p] > 0xFFFFFFFFull)      // (see finish_index_shape: 32-bit row indices in k_scan)
                rc = fail(PM_ERANGE, "sub-index %u has %llu rows: the scan addresses at most 2^32 rows of this width", p, (unsigned long long)pc.sig[p]);
            hipError_t e = rc ? hipSuccess : device_malloc_reclaim((void**)&part->d_matrix, in.device_bytes);
            if (rc) {}
            else if (e != hipSuccess || !part->d_matrix) {
                part->d_matrix = nullptr;
                rc = fail(PM_ENOMEM, "hipMalloc(%llu bytes) for sub-index %u failed: %s",
                          (unsigned long long)in.device_bytes, p, e != hipSuccess ? hipGetErrorString(e) : "null pointer");
            } else { in.has_matrix = 1; rc = stream_matrix(rd, part, pc.page, pc.sig[p]); }
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

extern "C" int pm_index_load_fd(int fd, uint64_t size_hint, int layout, pm_index_t** out) try {
    NEED_DEV();
    if (!out || fd < 0) return fail(PM_EINVAL, "bad argument");
    Reader rd; rd.fd = fd;
    return load_from_reader(rd, size_hint, layout, false, out);
} PM_GUARD_END
// The decode-once cache of a compressed index (the reference's `mem-disk` mode with
// keep_cobs_indexes, Snakefile:364-387): while the stream (`xzcat` pipe) is loaded into HBM every byte read is also
// written to a file of this call's own, which becomes `tee_path` once the whole index has arrived and the file on
// disk has exactly the bytes that were read -- a later run finds the plain file and takes the parallel pread path
// instead of decoding again.  The file is an UNNAMED inode of the cache directory (O_TMPFILE) for as long as it is
// incomplete: a stage that is killed mid-load (SIGKILL from the launcher's grace period, the OOM killer) leaves nothing
// behind, however many GB it had written.  Publishing = linkat() to "<tee_path>.XXXXXX.tmp" + rename() over `tee_path`
// (two processes decoding the same batch never share an inode; when another one published the same index first, this
// call's copy is dropped).  On a file system without O_TMPFILE the same name is created up front with mkstemps and
// unlinked on every failing return.  A failed load leaves no file; a failed WRITE (disk full) does not fail the load: the
// index is resident, only the cache file is dropped (*cached = 0).
extern "C" int pm_index_load_fd_tee(int fd, uint64_t size_hint, int layout, const char* tee_path, int* cached, pm_index_t** out) try {
    NEED_DEV();
    if (!out || fd < 0 || !tee_path) return fail(PM_EINVAL, "bad argument");
    if (cached) *cached = 0;
    std::string tmp = std::string(tee_path) + ".XXXXXX.tmp";
    std::string dir = tee_path;
    const size_t slash = dir.rfind('/');
    dir = slash == std::string::npos ? std::string(".") : (slash == 0 ? std::string("/") : dir.substr(0, slash));
    Reader rd; rd.fd = fd;
    bool named = false;                                       // `tmp` exists under its name (mkstemps fallback)
    rd.tee_fd = open(dir.c_str(), O_TMPFILE | O_WRONLY | O_CLOEXEC, 0644);
    if (rd.tee_fd >= 0) (void)fchmod(rd.tee_fd, 0644);        // whatever the umask: the cache is read by later runs of any user
    else {
        rd.tee_fd = mkstemps(&tmp[0], 4);                      // O_CREAT | O_EXCL, a name of this call's own
        if (rd.tee_fd < 0) rd.tee_errno = errno;              // no cache file: the load itself goes on
        else { named = true; (void)fchmod(rd.tee_fd, 0644); }
    }
    int rc = load_from_reader(rd, size_hint, layout, false, out);
    bool ok = rc == PM_OK && rd.tee_fd >= 0 && !rd.tee_errno;
    if (ok) {
        // the index ends where its matrix ends: the cache file must hold exactly header + matrix bytes, all of them on disk
        pm_index_info_t in = (*out)->info;
        struct stat sb;
        ok = in.n_parts == 0 && rd.tee_bytes >= in.signature_size * in.row_bytes &&
             fstat(rd.tee_fd, &sb) == 0 && (uint64_t)sb.st_size == rd.tee_bytes;
    }
    if (ok && !named) {
        // give the finished inode its (unique) name; the descriptor stays open until the link exists
        char self[64];
        snprintf(self, sizeof self, "/proc/self/fd/%d", rd.tee_fd);
        ok = false;
        for (int attempt = 0; attempt < 16 && !ok; ++attempt) {
            std::string cand = std::string(tee_path) + ".XXXXXX.tmp";
            static const char al[] = "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789";
            uint64_t r = ((uint64_t)getpid() << 32) ^ (uint64_t)(uintptr_t)&rd ^ ((uint64_t)attempt * 0x9E3779B97F4A7C15ull) ^ (uint64_t)time(nullptr);
            for (size_t i = cand.size() - 10; i < cand.size() - 4; ++i) { r = r * 6364136223846793005ull + 1442695040888963407ull; cand[i] = al[(r >> 33) % 62]; }
            if (linkat(AT_FDCWD, self, AT_FDCWD, cand.c_str(), AT_SYMLINK_FOLLOW) == 0) { tmp = cand; named = true; ok = true; }
            else if (errno != EEXIST) break;
        }
    }
    if (rd.tee_fd >= 0) { if (close(rd.tee_fd) != 0) ok = false; }
    if (ok) {
        struct stat sb;
        if (stat(tee_path, &sb) == 0 && (uint64_t)sb.st_size == rd.tee_bytes) (void)unlink(tmp.c_str());   // published by another process meanwhile
        else if (rename(tmp.c_str(), tee_path) != 0) ok = false;
    }
    if (!ok && named) (void)unlink(tmp.c_str());
    if (cached) *cached = ok ? 1 : 0;
    return rc;
} PM_GUARD_END
extern "C" int pm_index_load_file(const char* path, uint64_t size_hint, int layout, pm_index_t** out) try {
    NEED_DEV();
    if (!path || !out) return fail(PM_EINVAL, "bad argument");
    int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(PM_EIO, "cannot open index '%s': %s", path, strerror(errno));
    int rc = pm_index_load_fd(fd, size_hint, layout, out);
    close(fd);
    return rc;
} PM_GUARD_END
extern "C" int pm_index_load_mem(const void* buf, size_t len, int layout, pm_index_t** out) try {
    NEED_DEV();
    if (!buf || !out) return fail(PM_EINVAL, "bad argument");
    Reader rd; rd.mem = (const uint8_t*)buf; rd.mem_len = len;
    return load_from_reader(rd, 0, layout, false, out);
} PM_GUARD_END
extern "C" int pm_index_load_header_mem(const void* buf, size_t len, pm_index_t** out) try {
    if (!buf || !out) return fail(PM_EINVAL, "bad argument");
    Reader rd; rd.mem = (const uint8_t*)buf; rd.mem_len = len;
    return load_from_reader(rd, 0, PM_LAYOUT_COMPACT, true, out);
} PM_GUARD_END

// An index made in place instead of read from a file: header fields + document names + a ZEROED matrix in HBM (or, with
// header_only, names only).  Whoever builds signatures on the device fills the matrix through pm_index_matrix_device.
extern "C" int pm_index_create(uint32_t term_size, uint32_t canonicalize, uint64_t signature_size, uint32_t num_hashes,
                               const char* names, size_t names_len, uint32_t n_docs, int layout, int header_only,
                               pm_index_t** out) try {
    if (!out || n_docs == 0 || signature_size == 0 || num_hashes == 0 || term_size == 0 || (!names && names_len))
        return fail(PM_EINVAL, "bad index shape");
    if (!header_only) NEED_DEV();
    pm_index_t* named = nullptr;
    { int rc = pm_index_from_names(names, names_len, n_docs, term_size, &named); if (rc) return rc; }
    ParsedHeader h;
    h.version = 1; h.term_size = term_size; h.canon = canonicalize ? 1 : 0; h.sig = signature_size; h.nh = num_hashes;
    h.n_docs = n_docs; h.layout = 0;
    int rc = finish_index_shape(named, h, layout, !header_only);
    if (rc) { pm_index_free(named); return rc; }
    if (!header_only) {
        hipError_t e = hipMemsetAsync(named->d_matrix, 0, named->info.device_bytes, g_ctx.stream);
        if (e == hipSuccess) e = hipStreamSynchronize(g_ctx.stream);
        if (e != hipSuccess) { pm_index_free(named); return fail(PM_EHIP, "clearing the new matrix: %s", hipGetErrorString(e)); }
    }
    *out = named;
    return PM_OK;
} PM_GUARD_END
// The resident matrix of a classic index for code that shares the device with the library (kernels of its own that
// build or inspect signatures): row r starts at dptr + r * stride, document d is bit d % 8 of byte d / 8.
extern "C" int pm_index_matrix_device(const pm_index_t* ix, void** dptr, uint64_t* stride) try {
    if (!ix || !dptr || !stride) return fail(PM_EINVAL, "bad argument");
    if (!ix->d_matrix) return fail(PM_EINVAL, "index has no single resident matrix (header-only, dropped, or a compact index)");
    *dptr = ix->d_matrix; *stride = ix->info.stride;
    return PM_OK;
} PM_GUARD_END

// Copies rows [row0, row0 + n) (row_bytes each, file layout) back to the host: lets a test
// rebuild the .cobs_classic file of a synthetic / planted index for the oracle.
extern "C" int pm_index_read_rows(const pm_index_t* ix, uint64_t row0, uint64_t n, void* out) try {
    NEED_DEV();
    if (!ix || !ix->d_matrix || !out || row0 + n > ix->info.signature_size) return fail(PM_EINVAL, "bad argument");
    if (n == 0) return PM_OK;
    HIPCHK(hipMemcpy2D(out, ix->info.row_bytes, ix->d_matrix + row0 * ix->info.stride, ix->info.stride,
                       ix->info.row_bytes, n, hipMemcpyDeviceToHost));
    return PM_OK;
} PM_GUARD_END

extern "C" int pm_index_from_names(const char* names, size_t len, uint32_t n_docs, uint32_t term_size, pm_index_t** out) try {
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
} PM_GUARD_END
extern "C" int pm_index_drop_matrix(pm_index_t* ix) try {
    if (!ix) return fail(PM_EINVAL, "bad argument");
    bind_thread_quiet();
    if (ix->d_matrix) { (void)hipFree(ix->d_matrix); ix->d_matrix = nullptr; }
    for (pm_index* p : ix->parts) pm_index_drop_matrix(p);
    ix->info.has_matrix = 0; ix->info.device_bytes = 0;
    return PM_OK;
} PM_GUARD_END

extern "C" int pm_index_info(const pm_index_t* ix, pm_index_info_t* info) try {
    if (!ix || !info) return fail(PM_EINVAL, "bad argument");
    *info = ix->info;
    return PM_OK;
} PM_GUARD_END
extern "C" const char* pm_index_doc_name(const pm_index_t* ix, uint32_t doc, size_t* len) {
    if (!ix || doc >= ix->info.n_docs) return nullptr;
    if (len) *len = (size_t)(ix->name_off[doc + 1] - ix->name_off[doc] - 1);
    return ix->names_blob.data() + ix->name_off[doc];
}
extern "C" int pm_index_read_row(const pm_index_t* ix, uint64_t row, void* out) try {
    NEED_DEV();
    if (!ix || !ix->d_matrix || !out || row >= ix->info.signature_size) return fail(PM_EINVAL, "bad argument");
    HIPCHK(hipMemcpy(out, ix->d_matrix + row * ix->info.stride, ix->info.row_bytes, hipMemcpyDeviceToHost));
    return PM_OK;
} PM_GUARD_END
// GPU that holds the signature matrix (hipPointerGetAttributes); -1 for header-only handles
extern "C" int pm_index_device(const pm_index_t* ix, int* device) try {
    if (!ix || !device) return fail(PM_EINVAL, "bad argument");
    const uint8_t* p = ix->d_matrix;
    if (!p) for (const pm_index* part : ix->parts) if (part->d_matrix) { p = part->d_matrix; break; }
    *device = -1;
    if (!p) return PM_OK;
    hipPointerAttribute_t at;
    HIPCHK(hipPointerGetAttributes(&at, p));
    *device = at.device;
    return PM_OK;
} PM_GUARD_END
extern "C" void pm_index_free(pm_index_t* ix) {
    if (!ix) return;
    bind_thread_quiet();
    if (ix->d_matrix) (void)hipFree(ix->d_matrix);
    for (pm_index* p : ix->parts) pm_index_free(p);
    delete ix;
}


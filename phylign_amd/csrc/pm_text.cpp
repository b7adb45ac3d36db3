// pm_text.cpp -- cobs result text (a7), the fused post-filter (a8) and the native
// 04_filter merge: host work, no GPU needed.
#include "pm_host.h"
#include <zlib.h>

// --------------------------------------------------------------------- text
// cobs stdout grammar (witnesses: scripts/postprocess_cobs.py:23-26, :10-13;
// scripts/filter_queries.py:51-65): "*<header>\t<N>\n" then N lines
// "<doc name>\t<score>\n", best score first, ties by document index.
// nb_best_hits >= 0 fuses scripts/postprocess_cobs.py:16-39: header untouched,
// each name cut to "_" + what follows its first '_', the first n lines kept plus
// later lines whose score equals the n-th score.
// Text and gzip bytes are built in POOLED raw buffers: a 03_match file of a million reads is tens of MB per batch, and
// fresh allocations of that size are mmap'ed, zero-filled and page-faulted by every formatting thread on every call
// (measured: the same formatting took as long on 8 threads as on 1).  Buffers go back to the pool after the file is
// written and keep their pages; the pool holds at most PM_TEXT_POOL_MB (default 1024) and is emptied by pm_shutdown.
struct Buf { char* p = nullptr; size_t cap = 0, n = 0; };
namespace {
std::mutex g_tb_mu;
std::vector<Buf> g_tb_free;
size_t g_tb_bytes = 0;
size_t text_pool_limit() {
    static const size_t lim = [] { const char* e = getenv("PM_TEXT_POOL_MB"); return (size_t)(e ? atoll(e) : 1024) << 20; }();
    return lim;
}
}  // namespace
static Buf take_buf(size_t cap) {
    {
        std::lock_guard<std::mutex> lk(g_tb_mu);
        size_t best = SIZE_MAX;
        for (size_t i = 0; i < g_tb_free.size(); ++i)
            if (g_tb_free[i].cap >= cap && (best == SIZE_MAX || g_tb_free[i].cap < g_tb_free[best].cap)) best = i;
        if (best != SIZE_MAX && g_tb_free[best].cap <= 4 * cap + (1u << 20)) {
            Buf b = g_tb_free[best];
            g_tb_free[best] = g_tb_free.back(); g_tb_free.pop_back();
            g_tb_bytes -= b.cap;
            b.n = 0;
            return b;
        }
    }
    Buf b;
    b.cap = std::max<size_t>(cap + cap / 8, 1u << 16);          // some slack: the next call's piece is rarely the same size
    b.p = (char*)malloc(b.cap);
    if (!b.p) b.cap = 0;
    return b;
}
static void give_buf(Buf& b) {
    if (!b.p) return;
    {
        std::lock_guard<std::mutex> lk(g_tb_mu);
        if (g_tb_bytes + b.cap <= text_pool_limit()) {
            g_tb_bytes += b.cap;
            g_tb_free.push_back(b);
            b = Buf();
            return;
        }
    }
    free(b.p);
    b = Buf();
}
static void give_bufs(std::vector<Buf>& v) { for (Buf& b : v) give_buf(b); v.clear(); }
void release_text_pool() {
    std::lock_guard<std::mutex> lk(g_tb_mu);
    for (Buf& b : g_tb_free) free(b.p);
    g_tb_free.clear();
    g_tb_bytes = 0;
}

static inline char* put_tab_uint_nl(char* w, uint64_t v) {     // "\t<v>\n"
    char buf[24]; int n = 0;
    do { buf[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    *w++ = '\t';
    while (n) *w++ = buf[--n];
    *w++ = '\n';
    return w;
}

// formats the records of queries [qa, qb) (a slice of one slot's ordered records) into a pooled buffer;
// returns PM_OK or an error code with the message in `err`
static int format_query_range(const pm_index* ix, const pm_queries* q, const pm_hit_t* mine, size_t n_mine,
                              size_t qa, size_t qb, int64_t nb_best, uint64_t limit, Buf& out, std::string& err) {
    char msg[512];
    size_t p = (size_t)(std::lower_bound(mine, mine + n_mine, (uint32_t)qa,
                                         [](const pm_hit_t& h, uint32_t v) { return h.query < v; }) - mine);
    // room: every header line, every record's name and score (an upper bound: the post-filter prints less)
    {
        size_t need = 64;
        for (size_t qi = qa; qi < qb; ++qi) need += q->headers[qi].size() + 24;
        const size_t pe = (size_t)(std::lower_bound(mine + p, mine + n_mine, (uint32_t)qb,
                                                    [](const pm_hit_t& h, uint32_t v) { return h.query < v; }) - mine);
        for (size_t i = p; i < pe; ++i) {
            if (mine[i].doc == PM_DOC_COUNT) continue;
            if (mine[i].doc >= ix->info.n_docs) {
                snprintf(msg, sizeof msg, "hit record (query %u, doc %u) out of range for this index", mine[i].query, mine[i].doc);
                err = msg;
                return PM_EINVAL;
            }
            need += (size_t)(ix->name_off[mine[i].doc + 1] - ix->name_off[mine[i].doc]) + 12;
        }
        out = take_buf(need);
        if (!out.p) { err = "out of host memory"; return PM_ENOMEM; }
    }
    char* w = out.p;
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
        if (limit && nb_best < 0) {
            // `cobs query -l N`: the N best results (the list is ordered), and the header counts what is printed
            if (e - p > limit) e = p + (size_t)limit;
            total = e - p;
        }
        if (!q->headerless[qi]) *w++ = '*';
        else if (nb_best >= 0) {    // the post-filter needs a '*' line first (postprocess_cobs.py:23-29 raises)
            snprintf(msg, sizeof msg, "record %zu has sequence lines before any FASTA header: the post-filter cannot parse its result", qi);
            err = msg;
            return PM_EINVAL;
        }
        memcpy(w, q->headers[qi].data(), q->headers[qi].size()); w += q->headers[qi].size();
        w = put_tab_uint_nl(w, total);
        uint32_t min_kmers = 0;
        for (size_t i = p; i < e; ++i) {
            if (mine[i].doc >= ix->info.n_docs) {                 // a count record in the middle of a run: malformed input
                snprintf(msg, sizeof msg, "hit record (query %u, doc %u) out of range for this index", mine[i].query, mine[i].doc);
                err = msg;
                return PM_EINVAL;
            }
            const char* nm = ix->names_blob.data() + ix->name_off[mine[i].doc];
            const size_t nl = (size_t)(ix->name_off[mine[i].doc + 1] - ix->name_off[mine[i].doc] - 1);
            if (nb_best < 0) {
                memcpy(w, nm, nl); w += nl;
                w = put_tab_uint_nl(w, mine[i].score);
                continue;
            }
            const int64_t rank = (int64_t)(i - p) + 1;      // 1-based like the post-filter's counter
            const char* us = (const char*)memchr(nm, '_', nl);
            if (!us) {
                // postprocess_cobs.py:16-18 turns such a line into a bare "_" (no newline) and
                // raises on int("_") once rank >= n: an error for the whole rule
                if (rank < nb_best) { *w++ = '_'; continue; }
                snprintf(msg, sizeof msg, "document name '%.*s' has no '_' separator (post-filter cannot parse it)", (int)nl, nm);
                err = msg;
                return PM_EINVAL;
            }
            bool keep;
            if (rank < nb_best) keep = true;
            else if (rank == nb_best) { keep = true; min_kmers = mine[i].score; }
            else keep = mine[i].score == min_kmers;
            if (keep) {
                const size_t sl = nl - (size_t)(us - nm);
                memcpy(w, us, sl); w += sl;
                w = put_tab_uint_nl(w, mine[i].score);
            }
        }
        p = e;
        while (p < n_mine && mine[p].query == qi) ++p;          // records past a -l limit
    }
    out.n = (size_t)(w - out.p);
    return PM_OK;
}

static int format_impl(const pm_index_t* ix, const pm_queries_t* q, const pm_hit_t* hits, uint64_t n_hits, uint32_t slot,
                       int64_t nb_best, uint64_t limit, char** text, size_t* len);

extern "C" int pm_format_hits(const pm_index_t* ix, const pm_queries_t* q,
                              const pm_hit_t* hits, uint64_t n_hits, uint32_t slot,
                              int64_t nb_best, char** text, size_t* len) try {
    return format_impl(ix, q, hits, n_hits, slot, nb_best, 0, text, len);
} PM_GUARD_END
// plain cobs text with at most `limit` result lines per query: `cobs query -l limit` (0 = all)
extern "C" int pm_format_hits_limit(const pm_index_t* ix, const pm_queries_t* q,
                                    const pm_hit_t* hits, uint64_t n_hits, uint32_t slot,
                                    uint64_t limit, char** text, size_t* len) try {
    return format_impl(ix, q, hits, n_hits, slot, -1, limit, text, len);
} PM_GUARD_END

// the text in consecutive pieces (one per formatting thread, each ending on a line boundary); the caller hands the
// pieces back with give_bufs()
static int format_parts(const pm_index_t* ix, const pm_queries_t* q, const pm_hit_t* hits, uint64_t n_hits, uint32_t slot,
                        int64_t nb_best, uint64_t limit, std::vector<Buf>& parts) {
    if (!ix || !q || (!hits && n_hits)) return fail(PM_EINVAL, "bad argument");
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
    size_t nt = std::min<size_t>(parallel_width(), (nq + n_mine / 8) / 4096);
    if (nt < 1) nt = 1;
    parts.assign(nt, Buf());
    std::vector<std::string> errs(nt);
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
        rcs[t] = format_query_range(ix, q, mine, n_mine, cutq[t], cutq[t + 1], nb_best, limit, parts[t], errs[t]);
    };
    parallel_for(nt, work);
    for (size_t t = 0; t < nt; ++t)
        if (rcs[t] != PM_OK) {                                   // the first failing query range, as a serial pass would report
            give_bufs(parts);
            return fail(rcs[t], "%s", errs[t].c_str());
        }
    return PM_OK;
}

static int format_impl(const pm_index_t* ix, const pm_queries_t* q, const pm_hit_t* hits, uint64_t n_hits, uint32_t slot,
                       int64_t nb_best, uint64_t limit, char** text, size_t* len) {
    if (!text || !len) return fail(PM_EINVAL, "bad argument");
    std::vector<Buf> parts;
    { int rc = format_parts(ix, q, hits, n_hits, slot, nb_best, limit, parts); if (rc) return rc; }
    size_t total = 0;
    for (auto& s2 : parts) total += s2.n;
    char* buf = (char*)malloc(total + 1);
    if (!buf) { give_bufs(parts); return fail(PM_ENOMEM, "out of host memory"); }
    size_t o = 0;
    for (auto& s2 : parts) { memcpy(buf + o, s2.p, s2.n); o += s2.n; }
    buf[total] = 0;
    give_bufs(parts);
    *text = buf; *len = total;
    return PM_OK;
}

// pieces of ~1 MiB that end on a line boundary (4 MiB at most, whatever the lines are): the units of the parallel deflate
struct Chunk { const char* p; size_t n; };
static void cut_chunks(const char* p, size_t n, std::vector<Chunk>& chunks) {
    constexpr size_t kChunk = 1u << 20, kMax = 4u << 20;
    size_t o = 0;
    while (o < n) {
        size_t e = std::min(n, o + kChunk);
        if (e < n) {
            const size_t lim = std::min(n, o + kMax);
            const void* nl = memchr(p + e, '\n', lim - e);
            e = nl ? (size_t)((const char*)nl - p) + 1 : lim;
        }
        chunks.push_back({p + o, e - o});
        o = e;
    }
}
// chunks -> gzip members in pooled buffers (level 1: this library's `gzip --fast`, pm_gzfast.cpp; else zlib at that level)
static int deflate_chunks(const std::vector<Chunk>& chunks, int level, std::vector<Buf>& members) {
    members.assign(chunks.size(), Buf());
    std::vector<int> zrc(chunks.size(), Z_OK);
    auto deflate_one = [&](size_t i) {
        Buf& out = members[i];
        if (level == 1) {
            out = take_buf(gz_fast_bound(chunks[i].n));
            if (!out.p) { zrc[i] = Z_MEM_ERROR; return; }
            out.n = gz_fast_member(chunks[i].p, chunks[i].n, (uint8_t*)out.p);
            return;
        }
        z_stream z;
        memset(&z, 0, sizeof z);
        int rc = deflateInit2(&z, level, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY);      // 15 + 16: gzip container
        if (rc != Z_OK) { zrc[i] = rc; return; }
        out = take_buf(deflateBound(&z, (uLong)chunks[i].n) + 64);
        if (!out.p) { zrc[i] = Z_MEM_ERROR; deflateEnd(&z); return; }
        z.next_in = (Bytef*)const_cast<char*>(chunks[i].p); z.avail_in = (uInt)chunks[i].n;
        z.next_out = (Bytef*)out.p; z.avail_out = (uInt)out.cap;
        rc = deflate(&z, Z_FINISH);
        if (rc != Z_STREAM_END) zrc[i] = rc == Z_OK ? Z_BUF_ERROR : rc;
        else out.n = z.total_out;
        deflateEnd(&z);
    };
    parallel_for(chunks.size(), deflate_one);
    for (int rc : zrc) if (rc != Z_OK) { give_bufs(members); return fail(rc == Z_MEM_ERROR ? PM_ENOMEM : PM_EIO, "deflate failed (%d)", rc); }
    return PM_OK;
}
extern "C" int pm_gzip_fast(const char* text, size_t len, char** gz, size_t* gz_len) try {
    if ((!text && len) || !gz || !gz_len) return fail(PM_EINVAL, "bad argument");
    std::vector<Chunk> chunks;
    cut_chunks(text, len, chunks);
    if (chunks.empty()) chunks.push_back({"", 0});
    std::vector<Buf> members;
    { int rc = deflate_chunks(chunks, 1, members); if (rc) return rc; }
    size_t total = 0;
    for (auto& m2 : members) total += m2.n;
    char* buf = (char*)malloc(total ? total : 1);
    if (!buf) { give_bufs(members); return fail(PM_ENOMEM, "out of host memory"); }
    size_t o = 0;
    for (auto& m2 : members) { memcpy(buf + o, m2.p, m2.n); o += m2.n; }
    give_bufs(members);
    *gz = buf; *gz_len = total;
    return PM_OK;
} PM_GUARD_END

// The 03_match FILE of one batch in one call: what `run_cobs_streaming.sh ... | postprocess_cobs.py -n N | gzip --fast >
// <batch>____<qfile>.gz` leaves on disk (Snakefile:463-469).  The text is formatted on several threads (as above), cut at
// line boundaries into chunks of ~1 MiB that are deflated in parallel as consecutive gzip MEMBERS (a multi-member file is
// a valid gzip stream: `gzip -dc`, xopen and Python's gzip decode it to the same bytes -- scripts/filter_queries.py:46
// reads through xopen), and written to "<path>.tmp" + rename.  Nothing of it passes through the caller.
static int format_hits_gz_impl(const pm_index_t* ix, const pm_queries_t* q, const pm_hit_t* hits, uint64_t n_hits,
                               uint32_t slot, int64_t nb_best, const char* path, int level, int piece,
                               uint64_t* text_bytes, uint64_t* gz_bytes);
extern "C" int pm_format_hits_gz(const pm_index_t* ix, const pm_queries_t* q, const pm_hit_t* hits, uint64_t n_hits,
                                 uint32_t slot, int64_t nb_best, const char* path, int level,
                                 uint64_t* text_bytes, uint64_t* gz_bytes) try {
    return format_hits_gz_impl(ix, q, hits, n_hits, slot, nb_best, path, level, 0, text_bytes, gz_bytes);
} PM_GUARD_END
// The same for a query set that is searched in CHUNKS (a file of more reads than fit HBM at once): the pieces of a batch's
// file are written one after the other -- gzip members may simply follow each other -- into "<path>.tmp":
// piece 1 = first (creates it), 2 = a middle one (appends), 3 = the last (appends, then renames to `path`); 0 = the whole file.
extern "C" int pm_format_hits_gz_piece(const pm_index_t* ix, const pm_queries_t* q, const pm_hit_t* hits, uint64_t n_hits,
                                       uint32_t slot, int64_t nb_best, const char* path, int level, int piece,
                                       uint64_t* text_bytes, uint64_t* gz_bytes) try {
    if (piece < 0 || piece > 3) return fail(PM_EINVAL, "piece must be 0 (whole), 1 (first), 2 (middle) or 3 (last)");
    return format_hits_gz_impl(ix, q, hits, n_hits, slot, nb_best, path, level, piece, text_bytes, gz_bytes);
} PM_GUARD_END
static int format_hits_gz_impl(const pm_index_t* ix, const pm_queries_t* q, const pm_hit_t* hits, uint64_t n_hits,
                               uint32_t slot, int64_t nb_best, const char* path, int level, int piece,
                               uint64_t* text_bytes, uint64_t* gz_bytes) {
    if (!path || level < 0 || level > 9) return fail(PM_EINVAL, "bad argument");
    std::vector<Buf> parts;
    { int rc = format_parts(ix, q, hits, n_hits, slot, nb_best, 0, parts); if (rc) return rc; }
    std::vector<Chunk> chunks;
    uint64_t total = 0;
    for (const Buf& s2 : parts) {
        total += s2.n;
        cut_chunks(s2.p, s2.n, chunks);
    }
    if (chunks.empty()) chunks.push_back({"", 0});                    // an empty text is one empty member
    std::vector<Buf> members;
    { int rc = deflate_chunks(chunks, level, members); give_bufs(parts); if (rc) return rc; }
    const std::string tmp = std::string(path) + ".tmp";
    const bool append = piece == 2 || piece == 3, finish = piece == 0 || piece == 3;
    int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | (append ? O_APPEND : O_TRUNC), 0644);
    if (fd < 0) { give_bufs(members); return fail(PM_EIO, "cannot %s '%s': %s", append ? "append to" : "create", tmp.c_str(), strerror(errno)); }
    uint64_t gz = 0;
    int e = 0;
    for (const Buf& m2 : members) {
        const char* p2 = m2.p; size_t left = m2.n;
        while (left && !e) {
            ssize_t w = write(fd, p2, left);
            if (w < 0) { if (errno == EINTR) continue; e = errno; break; }
            p2 += w; left -= (size_t)w;
        }
        gz += m2.n;
    }
    give_bufs(members);
    if (close(fd) != 0 && !e) e = errno;
    if (e) { (void)unlink(tmp.c_str()); return fail(PM_EIO, "writing '%s': %s", tmp.c_str(), strerror(e)); }
    if (finish && rename(tmp.c_str(), path) != 0) return fail(PM_EIO, "rename to '%s': %s", path, strerror(errno));
    if (text_bytes) *text_bytes = total;
    if (gz_bytes) *gz_bytes = gz;
    return PM_OK;
}

extern "C" int pm_query_text(pm_index_t* ix, const char* fasta, size_t fasta_len,
                             double threshold, int64_t nb_best, char** text, size_t* len) try {
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
} PM_GUARD_END

// --------------------------------------------------------- 04_filter merge
// Native form of the reference's consumer (scripts/filter_queries.py:107-206):
// for every query keep the globally best `keep` matches across batches plus the
// ones tied with the last of them, ordered by (-kmers, batch, ref), and emit
// ">qname ref1,ref2,...\nseq".  What is merged per batch is what the 03_match
// file of that batch holds, i.e. the hit list after the per-batch post-filter
// (scripts/postprocess_cobs.py:21-39 with -n nb_best_hits).
// An item is 12 bytes: the reference name ("what follows the first '_'") is looked up in the merge's
// own copy of the batch's name table when items are ordered or printed, so a million queries with a
// hundred matches each cost ~1.2 GB, not a std::string apiece, and the index may be freed right after
// pm_merge_add (match_stage streams batches through HBM and never keeps their records).
struct MergeItem { uint32_t kmers, batch, doc; };
struct MergeBatch {
    std::string name;                   // batch name of "<batch>____<qfile>.gz"
    std::string refs;                   // reference names, '\0' separated
    std::vector<uint32_t> ref_off;      // n_docs + 1
    std::vector<uint32_t> ref_rank;     // position of the document's reference name in the batch's sorted names
};
struct pm_merge {
    // the query file: one query set, or several PIECES of it in file order (pm_merge_extend: a file that is parsed and
    // searched piece by piece); records are numbered through the pieces.  The sets outlive the merge: names and
    // sequences are views into them.
    std::vector<const pm_queries*> pieces;
    std::vector<uint32_t> piece_base;                         // global number of a piece's first record (+ the total at the end)
    uint32_t keep = 0;
    std::vector<MergeBatch> batches;
    std::unordered_map<std::string, uint32_t> batch_id;      // batch name -> its (first) entry in `batches`
    // query name (readfq: the header up to its first space) -> record.  A flat open-addressing table over views
    // into the query sets' header strings: a million std::string keys in an unordered_map cost 0.2-0.6 s to build,
    // serially, before the first search of the stage could start.  The consumer's dict semantics
    // (scripts/filter_queries.py:107-120, :178-185): a name is ONE query -- printed where it first occurs, with the
    // sequence of its last occurrence, and with the matches of all its occurrences; so the table maps a name to its
    // FIRST record (canon[] of every record with that name), and last_of[] of that record is the last one.
    std::vector<const char*> qname_p;
    std::vector<uint32_t> qname_n;
    std::vector<const char*> seq_p;
    std::vector<uint32_t> seq_n;
    std::vector<uint64_t> name_hash;                          // kept: the table is rebuilt when it grows
    std::vector<uint32_t> table;                              // record number or kEmpty
    uint32_t mask = 0;
    std::vector<uint32_t> canon;                              // canon[i]: the first record with record i's name
    std::vector<uint32_t> last_of;                            // for a first record: the last record with its name
    static constexpr uint32_t kEmpty = 0xFFFFFFFFu;
    static uint64_t hash_name(const char* p, size_t n) {
        uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)n;
        size_t j = 0;
        for (; j + 8 <= n; j += 8) { uint64_t w; memcpy(&w, p + j, 8); h = (h ^ w) * 0xFF51AFD7ED558CCDull; h ^= h >> 32; }
        if (j < n) { uint64_t w = 0; memcpy(&w, p + j, n - j); h = (h ^ w) * 0xFF51AFD7ED558CCDull; h ^= h >> 32; }
        return h * 0xC4CEB9FE1A85EC53ull;
    }
    // the table is 2^region_bits regions (top bits of the hash), probing wraps inside a region: regions are filled by
    // different threads, each in record order (so "first" and "last" are what they are in the file)
    uint32_t region_bits = 0;
    bool dup_names = false, tab_names = false;                // a repeated query name / a TAB inside one: pm_merge_add stays on one thread
    uint32_t slot0(uint64_t h, uint32_t* base, uint32_t* rmask) const {
        const uint32_t rsize = (mask + 1u) >> region_bits;
        *rmask = rsize - 1u;
        *base = region_bits ? (uint32_t)(h >> (64 - region_bits)) * rsize : 0u;
        return (uint32_t)(h >> 16) & *rmask;
    }
    uint32_t lookup(const char* p, size_t n) const {           // kEmpty: no such query
        uint32_t base, rmask;
        for (uint32_t s2 = slot0(hash_name(p, n), &base, &rmask);; s2 = (s2 + 1) & rmask) {
            const uint32_t r = table[base + s2];
            if (r == kEmpty) return kEmpty;
            if (qname_n[r] == n && memcmp(qname_p[r], p, n) == 0) return r;
        }
    }
    std::vector<std::vector<MergeItem>> items;
    std::vector<uint32_t> floor_;
    std::mutex mu;                                  // pm_merge_add may be called from a consumer thread pool
    const char* ref(const MergeItem& it, size_t* len) const {
        const MergeBatch& b = batches[it.batch];
        *len = (size_t)(b.ref_off[it.doc + 1] - b.ref_off[it.doc] - 1);
        return b.refs.data() + b.ref_off[it.doc];
    }
    bool less(const MergeItem& a, const MergeItem& b) const {       // (-kmers, batch, ref): scripts/filter_queries.py:135
        if (a.kmers != b.kmers) return a.kmers > b.kmers;
        if (a.batch == b.batch) {                                   // the usual case: integer compare
            const uint32_t ra = batches[a.batch].ref_rank[a.doc], rb = batches[a.batch].ref_rank[b.doc];
            return ra != rb ? ra < rb : a.doc < b.doc;
        }
        { const int c = batches[a.batch].name.compare(batches[b.batch].name); if (c) return c < 0; }
        size_t la, lb;
        const char* ra = ref(a, &la); const char* rb = ref(b, &lb);
        const int c = memcmp(ra, rb, std::min(la, lb));
        if (c) return c < 0;
        if (la != lb) return la < lb;
        return a.batch < b.batch;
    }
};

// items [before, end) of query `target` are new (one batch): order them, merge them into the kept ones,
// cut to the `keep` best + ties (scripts/filter_queries.py:133-150).  presorted: they arrive best score
// first (records from a search), so only runs of equal scores need ordering by name.
static int merge_settle(pm_merge* m, uint32_t target, size_t before, bool presorted) {
    std::vector<MergeItem>& v = m->items[target];
    auto lt = [&](const MergeItem& a, const MergeItem& b) { return m->less(a, b); };
    if (v.size() != before) {
        if (!presorted) std::sort(v.begin() + (long)before, v.end(), lt);
        else
            for (size_t a0 = before; a0 < v.size();) {
                size_t a1 = a0 + 1;
                while (a1 < v.size() && v[a1].kmers == v[a0].kmers) ++a1;
                if (a1 - a0 > 1) std::sort(v.begin() + (long)a0, v.begin() + (long)a1, lt);
                a0 = a1;
            }
        if (before) std::inplace_merge(v.begin(), v.begin() + (long)before, v.end(), lt);
    }
    if (v.size() > m->keep) {
        if (m->keep == 0) return fail(PM_EINVAL, "keep = 0 is not supported by the 04_filter rule");
        size_t cut = m->keep;
        m->floor_[target] = v[cut - 1].kmers;
        while (cut < v.size() && v[cut].kmers == m->floor_[target]) ++cut;
        v.resize(cut);
        v.shrink_to_fit();
    }
    return PM_OK;
}

// the records [first, end) are new: into the name table, region by region on several threads; a full region grows the table
static void merge_index_names(pm_merge* m, size_t first) {
    const size_t end = m->qname_p.size();
    for (;;) {
        const size_t regions = (size_t)1 << m->region_bits;
        std::atomic<bool> full(false);
        parallel_for(regions, [&](size_t r) {
            const uint32_t rsize = (m->mask + 1u) >> m->region_bits;
            size_t used = 0;
            if (first)                                          // slots taken so far (a region's share is small: counted, not stored)
                for (uint32_t k = 0; k < rsize; ++k) used += m->table[(uint32_t)r * rsize + k] != pm_merge::kEmpty;
            for (size_t i = first; i < end; ++i) {
                if ((m->name_hash[i] >> (64 - m->region_bits)) != r) continue;
                uint32_t base, rmask;
                const char* p = m->qname_p[i]; const size_t n = m->qname_n[i];
                for (uint32_t s2 = m->slot0(m->name_hash[i], &base, &rmask);; s2 = (s2 + 1) & rmask) {
                    const uint32_t o = m->table[base + s2];
                    if (o == pm_merge::kEmpty) {
                        if (++used > rmask) { full.store(true); return; }       // would leave no empty slot: probing needs one
                        m->table[base + s2] = (uint32_t)i;
                        m->canon[i] = (uint32_t)i; m->last_of[i] = (uint32_t)i;
                        break;
                    }
                    if (m->qname_n[o] == n && memcmp(m->qname_p[o], p, n) == 0) {       // the name of an earlier record
                        m->canon[i] = o; m->last_of[o] = (uint32_t)i;
                        break;
                    }
                }
            }
        });
        if (!full.load()) return;
        // a region ran full (names whose hashes crowd it, or the table is simply too small by now): twice the slots, all
        // records again
        m->table.assign(m->table.size() * 2, pm_merge::kEmpty);
        m->mask = (uint32_t)(m->table.size() - 1);
        first = 0;
    }
}

extern "C" int pm_merge_extend(pm_merge_t* m, const pm_queries_t* q) try {
    if (!m || !q) return fail(PM_EINVAL, "bad argument");
    std::lock_guard<std::mutex> lk(m->mu);
    const size_t old = m->qname_p.size(), nq = q->headers.size(), total = old + nq;
    if (total >= 0x7FFFFFFFull) return fail(PM_ERANGE, "too many queries for one merge");
    m->pieces.push_back(q);
    m->piece_base.back() = (uint32_t)old;
    m->piece_base.push_back((uint32_t)total);
    m->items.resize(total); m->floor_.resize(total, 0);
    m->qname_p.resize(total); m->qname_n.resize(total); m->seq_p.resize(total); m->seq_n.resize(total);
    m->name_hash.resize(total); m->canon.resize(total); m->last_of.resize(total);
    const size_t nt = std::max<size_t>(1, std::min<size_t>(parallel_width(), nq / 65536));
    std::atomic<bool> tabs(false);
    parallel_for(nt, [&](size_t t) {
        for (size_t i = nq * t / nt; i < nq * (t + 1) / nt; ++i) {
            // readfq name: the header up to its first space (scripts/filter_queries.py:80)
            const std::string& h = q->headers[i];
            const void* sp = memchr(h.data(), ' ', h.size());
            const size_t g = old + i;
            m->qname_p[g] = h.data();
            m->qname_n[g] = (uint32_t)(sp ? (size_t)((const char*)sp - h.data()) : h.size());
            m->seq_p[g] = q->seqs.data() + q->seq_off[i];
            m->seq_n[g] = (uint32_t)(q->seq_off[i + 1] - q->seq_off[i]);
            m->name_hash[g] = pm_merge::hash_name(m->qname_p[g], m->qname_n[g]);
            if (memchr(m->qname_p[g], '\t', m->qname_n[g])) tabs.store(true, std::memory_order_relaxed);
        }
    });
    // the table holds at most a quarter of its slots (a region: at most all but one)
    size_t cap = std::max<size_t>(m->table.size(), 256);
    while (cap < 4 * total) cap <<= 1;
    size_t first = old;
    if (cap != m->table.size()) {
        m->table.assign(cap, pm_merge::kEmpty);
        m->mask = (uint32_t)(cap - 1);
        first = 0;
    }
    merge_index_names(m, first);
    if (!m->dup_names)
        for (size_t g = old; g < total; ++g) if (m->canon[g] != g) { m->dup_names = true; break; }
    m->tab_names = m->tab_names || tabs.load();
    return PM_OK;
} PM_GUARD_END

extern "C" int pm_merge_create(const pm_queries_t* q, uint32_t keep, pm_merge_t** out) try {
    if (!q || !out) return fail(PM_EINVAL, "bad argument");
    pm_merge* m = new pm_merge();
    m->keep = keep;
    m->region_bits = 4;
    m->piece_base.push_back(0);
    const int rc = pm_merge_extend(m, q);
    if (rc) { delete m; return rc; }
    *out = m;
    return PM_OK;
} PM_GUARD_END

static int merge_add_impl(pm_merge_t* m, int64_t piece, const char* batch, const pm_index_t* ix,
                          const pm_hit_t* hits, uint64_t n_hits, uint32_t slot, int64_t nb_best);
extern "C" int pm_merge_add(pm_merge_t* m, const char* batch, const pm_index_t* ix,
                            const pm_hit_t* hits, uint64_t n_hits, uint32_t slot, int64_t nb_best) try {
    return merge_add_impl(m, 0, batch, ix, hits, n_hits, slot, nb_best);
} PM_GUARD_END
// the records' query numbers count inside piece `piece` of the query file (pm_merge_extend), or -- piece = -1 -- through
// the whole file (what pm_merge_export writes: another rank's kept matches)
extern "C" int pm_merge_add_piece(pm_merge_t* m, int64_t piece, const char* batch, const pm_index_t* ix,
                                  const pm_hit_t* hits, uint64_t n_hits, uint32_t slot, int64_t nb_best) try {
    return merge_add_impl(m, piece, batch, ix, hits, n_hits, slot, nb_best);
} PM_GUARD_END
static int merge_add_impl(pm_merge_t* m, int64_t piece, const char* batch, const pm_index_t* ix,
                          const pm_hit_t* hits, uint64_t n_hits, uint32_t slot, int64_t nb_best) {
    if (!m || !batch || !ix || (!hits && n_hits)) return fail(PM_EINVAL, "bad argument");
    std::lock_guard<std::mutex> lk(m->mu);
    if (piece < -1 || piece >= (int64_t)m->pieces.size()) return fail(PM_EINVAL, "piece %lld of a merge over %zu pieces", (long long)piece, m->pieces.size());
    const size_t qbase = piece < 0 ? 0 : m->piece_base[(size_t)piece];
    const size_t nq = piece < 0 ? m->qname_p.size() : (size_t)(m->piece_base[(size_t)piece + 1] - m->piece_base[(size_t)piece]);
    // the slot's records: a contiguous slice when the input is ordered (what pm_result_hits_* deliver),
    // else copied out and ordered
    std::vector<pm_hit_t> copy;
    const pm_hit_t* mine = hits; size_t n_mine = 0;
    if (std::is_sorted(hits, hits + n_hits, [](const pm_hit_t& a, const pm_hit_t& b) { return hit_less(a, b); })) {
        const pm_hit_t* lo = std::lower_bound(hits, hits + n_hits, slot, [](const pm_hit_t& h, uint32_t v) { return h.slot < v; });
        const pm_hit_t* hi = std::upper_bound(lo, hits + n_hits, slot, [](uint32_t v, const pm_hit_t& h) { return v < h.slot; });
        mine = lo; n_mine = (size_t)(hi - lo);
    } else {
        for (uint64_t i = 0; i < n_hits; ++i) if (hits[i].slot == slot) copy.push_back(hits[i]);
        order_hits(copy.data(), copy.size());
        mine = copy.data(); n_mine = copy.size();
    }
    for (size_t i = 0; i < n_mine; ++i)
        if (mine[i].query >= nq || (mine[i].doc != PM_DOC_COUNT && mine[i].doc >= ix->info.n_docs))
            return fail(PM_EINVAL, "hit record out of range for batch %s", batch);
    // reference names of this batch ("<rnd>_<ref>" -> "<ref>", exactly one '_': scripts/filter_queries.py:64);
    // a malformed name is an error only if a kept record uses it, like the consumer's tuple unpacking
    MergeBatch mb;
    mb.name = batch;
    mb.ref_off.resize((size_t)ix->info.n_docs + 1);
    std::vector<uint8_t> bad_name((size_t)ix->info.n_docs, 0);
    for (uint32_t d = 0; d < ix->info.n_docs; ++d) {
        const char* nm = ix->names_blob.data() + ix->name_off[d];
        const size_t nl = (size_t)(ix->name_off[d + 1] - ix->name_off[d] - 1);
        const char* us = (const char*)memchr(nm, '_', nl);
        mb.ref_off[d] = (uint32_t)mb.refs.size();
        if (!us || memchr(us + 1, '_', nl - (size_t)(us + 1 - nm))) bad_name[d] = 1;
        else mb.refs.append(us + 1, nl - (size_t)(us + 1 - nm));
        mb.refs.push_back('\0');
    }
    mb.ref_off[ix->info.n_docs] = (uint32_t)mb.refs.size();
    // a batch is added once per piece of the query file: its name table is kept once
    uint32_t bid = (uint32_t)m->batches.size();
    {
        auto it = m->batch_id.find(mb.name);
        if (it != m->batch_id.end() && m->batches[it->second].refs == mb.refs && m->batches[it->second].ref_off == mb.ref_off) bid = it->second;
    }
    if (bid == m->batches.size()) {   // rank of every reference name inside the batch: ordering the items of one batch needs no string compare
        std::vector<uint32_t> order((size_t)ix->info.n_docs);
        for (uint32_t d = 0; d < ix->info.n_docs; ++d) order[d] = d;
        std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
            const int c = strcmp(mb.refs.data() + mb.ref_off[x], mb.refs.data() + mb.ref_off[y]);
            return c ? c < 0 : x < y;
        });
        mb.ref_rank.resize((size_t)ix->info.n_docs);
        uint32_t r = 0;
        for (size_t i = 0; i < order.size(); ++i) {
            if (i && strcmp(mb.refs.data() + mb.ref_off[order[i]], mb.refs.data() + mb.ref_off[order[i - 1]]) != 0) r = (uint32_t)i;
            mb.ref_rank[order[i]] = r;
        }
        m->batch_id.emplace(mb.name, bid);                  // (a second table under the same name keeps its own number)
        m->batches.push_back(std::move(mb));
    }
    // records [pa, pb) (whole queries): distinct queries have distinct targets unless the query file repeats a name, so
    // ranges of queries can be merged on several threads; the first error in record order is the one reported
    auto add_range = [&](size_t pa, size_t pb, std::string& err) -> int {
        char msg[512];
        size_t p = pa;
        while (p < pb) {
            size_t e = p;
            const uint32_t qi = mine[p].query;
            while (e < pb && mine[e].query == qi) ++e;
            while (p < e && mine[p].doc == PM_DOC_COUNT) ++p;            // count records carry no match
            if (p == e) continue;
            // the 03_match header is "*<header>\tN": the consumer looks the query up by the text
            // before the first TAB, cut at the first space (scripts/filter_queries.py:58-59)
            // (a name holds no TAB in practice: then that text is the query's own name and the lookup is canon[])
            const size_t gq = qbase + qi;
            uint32_t target = m->canon[gq];
            if (const void* tab = memchr(m->qname_p[gq], '\t', m->qname_n[gq])) {
                const size_t kn = (size_t)((const char*)tab - m->qname_p[gq]);
                target = m->lookup(m->qname_p[gq], kn);
                if (target == pm_merge::kEmpty) {
                    snprintf(msg, sizeof msg, "query '%.*s' of batch %s is not in the query file", (int)kn, m->qname_p[gq], batch);
                    err = msg;
                    return PM_EINVAL;
                }
            }
            std::vector<MergeItem>& v = m->items[target];
            const size_t before = v.size();
            uint32_t nth = 0;
            for (size_t i = p; i < e; ++i) {
                if (nb_best >= 0) {                          // per-batch post-filter, same rule as pm_format_hits
                    const int64_t rank = (int64_t)(i - p) + 1;
                    if (rank == nb_best) nth = mine[i].score;
                    if (rank > nb_best && mine[i].score != nth) continue;
                }
                if (bad_name[mine[i].doc]) {
                    const char* nm = ix->names_blob.data() + ix->name_off[mine[i].doc];
                    snprintf(msg, sizeof msg, "document name '%s' must hold exactly one '_' (scripts/filter_queries.py:64)", nm);
                    err = msg;
                    return PM_EINVAL;
                }
                if (mine[i].score >= m->floor_[target]) v.push_back({mine[i].score, bid, mine[i].doc});
            }
            { int rc = merge_settle(m, target, before, true); if (rc) { err = pm_last_error(); return rc; } }
            p = e;
        }
        return PM_OK;
    };
    size_t nt = m->dup_names || m->tab_names ? 1 : std::min<size_t>(parallel_width(), n_mine / 16384);
    if (nt < 1) nt = 1;
    std::vector<size_t> cut(nt + 1, n_mine);
    cut[0] = 0;
    for (size_t t = 1; t < nt; ++t) {                    // cuts at query boundaries
        size_t c = std::max(cut[t - 1], n_mine * t / nt);
        while (c < n_mine && c > 0 && mine[c].query == mine[c - 1].query) ++c;
        cut[t] = c;
    }
    std::vector<int> rcs(nt, PM_OK);
    std::vector<std::string> errs(nt);
    parallel_for(nt, [&](size_t t) { rcs[t] = add_range(cut[t], cut[t + 1], errs[t]); });
    for (size_t t = 0; t < nt; ++t) if (rcs[t] != PM_OK) return fail(rcs[t], "%s", errs[t].c_str());
    return PM_OK;
}

// The 03_match TEXT of one batch (what `... | postprocess_cobs.py | gzip` wrote, after gunzip) added to the
// merge: the native form of the consumer's reader (scripts/filter_queries.py:27-66) for the drop-in
// scripts/filter_queries.py.  Rules kept: lines are stripped, empty ones skipped; "*<qname>[ comment]\t<N>"
// starts a query (N must be an integer); any other line is "<rnd>_<ref> <kmers>" -- exactly two
// whitespace-separated fields, exactly one '_' in the first; a text without any header and a query that is not
// in the query file are errors, as they are in the reference (behaviour on odd input captured from the
// reference's script: tests/golden/filter/edge/).
extern "C" int pm_merge_add_text(pm_merge_t* m, const char* batch, const char* text, size_t len) try {
    if (!m || !batch || (!text && len)) return fail(PM_EINVAL, "bad argument");
    struct Rec { uint32_t target, doc, kmers; };
    std::vector<Rec> recs;
    struct Block { const char* name; size_t name_len; size_t first; uint32_t target; };
    std::vector<Block> blocks;                                     // query name and first record per '*' header
    std::unordered_map<std::string, uint32_t> ref_id;
    std::vector<std::string> refs;
    // str.strip() / str.split() whitespace within ASCII: blank, TAB, CR, VT, FF and the separators 0x1c-0x1f
    auto is_ws = [](char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f' || (c >= 0x1c && c <= 0x1f); };
    // Python's int(): surrounding whitespace, one sign, decimal digits with single '_' between digits; *neg: a '-' sign
    // (the reference keeps a match only while kmers >= its floor, which starts at 0: a negative count never counts)
    auto parse_int = [&](const char* b, const char* e, uint64_t* out, bool* neg) {
        while (b < e && (is_ws(*b) || *b == '\n')) ++b;
        while (e > b && (is_ws(e[-1]) || e[-1] == '\n')) --e;
        *neg = false;
        if (b < e && (*b == '+' || *b == '-')) { *neg = *b == '-'; ++b; }
        if (b >= e || *b < '0' || *b > '9') return false;
        uint64_t v = 0;
        for (; b < e; ++b) {
            if (*b == '_') { if (b + 1 >= e || b[1] < '0' || b[1] > '9') return false; continue; }
            if (*b < '0' || *b > '9') return false;
            v = v * 10 + (uint64_t)(*b - '0');
            if (v > 0xFFFFFFFFull) v = 0xFFFFFFFFull;    // Python's int() has no limit: counts are kept as 32 bits, larger ones saturate
        }
        *out = v;
        return true;
    };
    size_t p = 0, lineno = 0;
    bool have_header = false;
    // lines end at "\n", "\r\n" or a lone "\r": the reference reads the file in text mode with universal newlines
    // (xopen / open(..., "rt"): scripts/filter_queries.py:46)
    const bool any_cr = len && memchr(text, '\r', len) != nullptr;
    while (p < len) {
        const char* b = text + p;
        const char* nl = (const char*)memchr(b, '\n', len - p);
        size_t eol = 1;
        if (any_cr) {
            const char* cr = (const char*)memchr(b, '\r', nl ? (size_t)(nl - b) : len - p);
            if (cr) { nl = cr; eol = (cr + 1 < text + len && cr[1] == '\n') ? 2 : 1; }
        }
        const char* e = nl ? nl : text + len;
        p = (size_t)(e - text) + (nl ? eol : 0);
        ++lineno;
        while (b < e && is_ws(*b)) ++b;
        while (e > b && is_ws(e[-1])) --e;
        if (b == e) continue;
        if (*b == '*') {
            const char* tab = (const char*)memchr(b + 1, '\t', (size_t)(e - (b + 1)));
            uint64_t n = 0;
            const char* nb = tab ? tab + 1 : e;
            const char* ne = tab ? (const char*)memchr(nb, '\t', (size_t)(e - nb)) : nullptr;
            bool neg_ = false;
            if (!tab || !parse_int(nb, ne ? ne : e, &n, &neg_))
                return fail(PM_EINVAL, "batch %s line %zu: query header without an integer match count", batch, lineno);
            const char* qe = (const char*)memchr(b + 1, ' ', (size_t)(tab - (b + 1)));
            const size_t qn = (size_t)((qe ? qe : tab) - (b + 1));
            // match lines ahead of the first header join the first query's list: the reference's reader only empties
            // its buffer when it has a query to yield (scripts/filter_queries.py:52-56; pinned by a captured fixture)
            // (the name is looked up below, under the merge's lock: pm_merge_extend may be growing the table right now)
            blocks.push_back({b + 1, qn, have_header ? recs.size() : 0, 0u});
            have_header = true;
            continue;
        }
        const char* f1e = b;
        while (f1e < e && !is_ws(*f1e)) ++f1e;
        const char* f2b = f1e;
        while (f2b < e && is_ws(*f2b)) ++f2b;
        const char* f2e = f2b;
        while (f2e < e && !is_ws(*f2e)) ++f2e;
        uint64_t km = 0;
        bool km_neg = false;
        if (f2b == f2e || f2e != e || !parse_int(f2b, f2e, &km, &km_neg))
            return fail(PM_EINVAL, "batch %s line %zu: a match line must be '<name> <k-mers>'", batch, lineno);
        const char* us = (const char*)memchr(b, '_', (size_t)(f1e - b));
        if (!us || memchr(us + 1, '_', (size_t)(f1e - (us + 1))))
            return fail(PM_EINVAL, "batch %s line %zu: document name '%.*s' must hold exactly one '_' (scripts/filter_queries.py:64)",
                        batch, lineno, (int)(f1e - b), b);
        std::string ref(us + 1, (size_t)(f1e - (us + 1)));
        auto ins = ref_id.emplace(std::move(ref), (uint32_t)refs.size());
        if (ins.second) refs.push_back(ins.first->first);
        if (km_neg && km) continue;                       // below every floor: never kept (the line was checked like the others)
        recs.push_back({0u, ins.first->second, (uint32_t)km});
    }
    if (!have_header) return fail(PM_EINVAL, "batch %s: no '*' query header in the match text", batch);
    std::lock_guard<std::mutex> lk(m->mu);
    for (Block& bl : blocks) {
        bl.target = m->lookup(bl.name, bl.name_len);
        if (bl.target == pm_merge::kEmpty)
            return fail(PM_EINVAL, "query '%.*s' of batch %s is not in the query file", (int)bl.name_len, bl.name, batch);
    }
    const uint32_t bid = (uint32_t)m->batches.size();
    m->batches.emplace_back();
    MergeBatch& mb = m->batches.back();
    mb.name = batch;
    mb.ref_off.resize(refs.size() + 1);
    mb.ref_rank.resize(refs.size());
    for (size_t d = 0; d < refs.size(); ++d) { mb.ref_off[d] = (uint32_t)mb.refs.size(); mb.refs += refs[d]; mb.refs.push_back('\0'); }
    mb.ref_off[refs.size()] = (uint32_t)mb.refs.size();
    {   // rank of every reference name inside the batch (names are distinct here: they were interned)
        std::vector<uint32_t> order(refs.size());
        for (size_t d = 0; d < refs.size(); ++d) order[d] = (uint32_t)d;
        std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return refs[x] < refs[y]; });
        for (size_t i = 0; i < order.size(); ++i) mb.ref_rank[order[i]] = (uint32_t)i;
    }
    for (size_t k = 0; k < blocks.size(); ++k) {
        const uint32_t target = blocks[k].target;
        const size_t a = blocks[k].first, b2 = k + 1 < blocks.size() ? blocks[k + 1].first : recs.size();
        std::vector<MergeItem>& v = m->items[target];
        const size_t before = v.size();
        for (size_t i = a; i < b2; ++i)
            if (recs[i].kmers >= m->floor_[target]) v.push_back({recs[i].kmers, bid, recs[i].doc});
        int rc = merge_settle(m, target, before, false);
        if (rc) return rc;
    }
    return PM_OK;
} PM_GUARD_END

// What is kept so far as hit records {query (numbered through the whole file), doc, score, slot = number of the batch
// in this merge: pm_merge_batches}, ordered by (slot, query, score desc, doc asc): a rank's share of the 04_filter
// merge, ready to be gathered (RCCL) and added again on rank 0 -- the best `keep` (+ ties) of the
// union are among the best `keep` (+ ties) of every part.
extern "C" int pm_merge_export(const pm_merge_t* m_, pm_hit_t** out, uint64_t* n) try {
    pm_merge* m = const_cast<pm_merge*>(m_);
    if (!m || !out || !n) return fail(PM_EINVAL, "bad argument");
    std::lock_guard<std::mutex> lk(m->mu);
    uint64_t total = 0;
    for (auto& v : m->items) total += v.size();
    pm_hit_t* buf = (pm_hit_t*)malloc(std::max<uint64_t>(total, 1) * sizeof(pm_hit_t));
    if (!buf) return fail(PM_ENOMEM, "out of host memory");
    uint64_t o = 0;
    for (size_t qi = 0; qi < m->items.size(); ++qi)
        for (const MergeItem& it : m->items[qi]) buf[o++] = pm_hit_t{(uint32_t)qi, it.doc, it.kmers, it.batch};
    order_hits(buf, total);
    *out = buf; *n = total;
    return PM_OK;
} PM_GUARD_END

// the batches of the merge in the order of their numbers (the `slot` of pm_merge_export's records), '\n'-separated
extern "C" int pm_merge_batches(const pm_merge_t* m_, char** names, size_t* len) try {
    pm_merge* m = const_cast<pm_merge*>(m_);
    if (!m || !names || !len) return fail(PM_EINVAL, "bad argument");
    std::lock_guard<std::mutex> lk(m->mu);
    std::string out;
    for (const MergeBatch& b : m->batches) { out += b.name; out.push_back('\n'); }
    char* buf = (char*)malloc(out.size() + 1);
    if (!buf) return fail(PM_ENOMEM, "out of host memory");
    memcpy(buf, out.data(), out.size());
    buf[out.size()] = 0;
    *names = buf; *len = out.size();
    return PM_OK;
} PM_GUARD_END

// The 04_filter FASTA in blocks of records: exact sizes first (the text is a concatenation of known strings), then every
// block is formatted straight to its place -- into the caller's buffer, or through a pooled scratch buffer and
// pwrite() into the file -- on several threads.  Nothing the size of the output is allocated, touched twice or copied.
struct EmitPlan {
    std::vector<uint32_t> recs;           // records in output order
    std::vector<size_t> first;            // block b covers recs[first[b], first[b + 1])
    std::vector<uint64_t> off;            // byte offset of block b; off.back() = total
};
static void merge_emit_plan(const pm_merge* m, EmitPlan& pl) {
    const size_t nq = m->qname_p.size();
    // dict semantics of the consumer: one record per distinct name, at the position of its
    // first occurrence, with the sequence of its last occurrence
    pl.recs.reserve(nq);
    for (size_t i = 0; i < nq; ++i) if (m->canon[i] == i) pl.recs.push_back((uint32_t)i);
    constexpr size_t kBlock = 4096;
    const size_t nb = (pl.recs.size() + kBlock - 1) / kBlock;
    pl.first.resize(nb + 1);
    for (size_t b = 0; b <= nb; ++b) pl.first[b] = std::min(pl.recs.size(), b * kBlock);
    pl.off.assign(nb + 1, 0);
    const size_t nt = std::max<size_t>(1, std::min<size_t>(parallel_width(), nb / 4));
    parallel_for(nt, [&](size_t t) {
        for (size_t b = nb * t / nt; b < nb * (t + 1) / nt; ++b) {
            uint64_t bytes = 0;
            for (size_t k = pl.first[b]; k < pl.first[b + 1]; ++k) {
                const uint32_t rec = pl.recs[k];
                const std::vector<MergeItem>& v = m->items[rec];
                bytes += 1 + m->qname_n[rec] + 1 + (v.empty() ? 0 : v.size() - 1) + 1 + (uint64_t)m->seq_n[m->last_of[rec]] + 1;
                for (const MergeItem& it : v) { size_t rl; (void)m->ref(it, &rl); bytes += rl; }
            }
            pl.off[b + 1] = bytes;
        }
    });
    for (size_t b = 0; b < nb; ++b) pl.off[b + 1] += pl.off[b];
}
static char* merge_emit_block(const pm_merge* m, const EmitPlan& pl, size_t b, char* w) {
    for (size_t k = pl.first[b]; k < pl.first[b + 1]; ++k) {
        const uint32_t rec = pl.recs[k];
        *w++ = '>'; memcpy(w, m->qname_p[rec], m->qname_n[rec]); w += m->qname_n[rec]; *w++ = ' ';
        const std::vector<MergeItem>& v = m->items[rec];
        for (size_t j = 0; j < v.size(); ++j) {
            if (j) *w++ = ',';
            size_t rl; const char* r = m->ref(v[j], &rl);
            memcpy(w, r, rl); w += rl;
        }
        *w++ = '\n';
        const uint32_t sr = m->last_of[rec];
        memcpy(w, m->seq_p[sr], m->seq_n[sr]); w += m->seq_n[sr];
        *w++ = '\n';
    }
    return w;
}

extern "C" int pm_merge_emit(const pm_merge_t* m, char** text, size_t* len) try {
    if (!m || !text || !len) return fail(PM_EINVAL, "bad argument");
    std::lock_guard<std::mutex> lk(const_cast<pm_merge_t*>(m)->mu);      // adds and extends wait until the text is out
    EmitPlan pl;
    merge_emit_plan(m, pl);
    const size_t total = (size_t)pl.off.back(), nb = pl.first.size() - 1;
    char* buf = (char*)malloc(total + 1);
    if (!buf) return fail(PM_ENOMEM, "out of host memory");
    const size_t nt = std::max<size_t>(1, std::min<size_t>(parallel_width(), nb / 4));
    parallel_for(nt, [&](size_t t) {
        for (size_t b = nb * t / nt; b < nb * (t + 1) / nt; ++b) merge_emit_block(m, pl, b, buf + pl.off[b]);
    });
    buf[total] = 0;
    *text = buf; *len = total;
    return PM_OK;
} PM_GUARD_END

// The same text written to `path` (through "<path>.tmp" + rename: never a partial file that looks
// complete), the pieces written in parallel at their offsets: the 04_filter FASTA of a million reads
// is hundreds of MB that need not pass through the caller.
static int merge_emit_file_impl(const pm_merge_t* m, const char* path, int piece, uint64_t* bytes);
extern "C" int pm_merge_emit_file(const pm_merge_t* m, const char* path, uint64_t* bytes) try {
    return merge_emit_file_impl(m, path, 0, bytes);
} PM_GUARD_END
// The FASTA of a query file that was searched in chunks (one merge per chunk, in file order): piece 1 = first (creates
// "<path>.tmp"), 2 = a middle one (appends), 3 = the last (appends, then renames to `path`); 0 = the whole file.
extern "C" int pm_merge_emit_file_piece(const pm_merge_t* m, const char* path, int piece, uint64_t* bytes) try {
    if (piece < 0 || piece > 3) return fail(PM_EINVAL, "piece must be 0 (whole), 1 (first), 2 (middle) or 3 (last)");
    return merge_emit_file_impl(m, path, piece, bytes);
} PM_GUARD_END
static int merge_emit_file_impl(const pm_merge_t* m, const char* path, int piece, uint64_t* bytes) {
    if (!m || !path) return fail(PM_EINVAL, "bad argument");
    std::lock_guard<std::mutex> lk(const_cast<pm_merge_t*>(m)->mu);      // adds and extends wait until the file is written
    EmitPlan pl;
    merge_emit_plan(m, pl);
    const size_t nb = pl.first.size() - 1;
    const std::string tmp = std::string(path) + ".tmp";
    const bool append = piece == 2 || piece == 3, finish = piece == 0 || piece == 3;
    int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | (append ? 0 : O_TRUNC), 0644);
    if (fd < 0) return fail(PM_EIO, "cannot %s '%s': %s", append ? "append to" : "create", tmp.c_str(), strerror(errno));
    uint64_t base = 0;
    if (append) {
        struct stat sb;
        if (fstat(fd, &sb) != 0) { const int e0 = errno; close(fd); return fail(PM_EIO, "stat '%s': %s", tmp.c_str(), strerror(e0)); }
        base = (uint64_t)sb.st_size;
    }
    // a worker formats runs of blocks of ~4 MB into one pooled scratch buffer and writes each at its offset
    const size_t nt = std::max<size_t>(1, std::min<size_t>(parallel_width(), nb / 4));
    std::vector<int> errs(nt, 0);
    parallel_for(nt, [&](size_t t) {
        const size_t b0 = nb * t / nt, b1 = nb * (t + 1) / nt;
        Buf scratch;
        for (size_t b = b0; b < b1 && !errs[t];) {
            size_t e = b + 1;
            while (e < b1 && pl.off[e + 1] - pl.off[b] <= (4u << 20)) ++e;
            const size_t need = (size_t)(pl.off[e] - pl.off[b]);
            if (scratch.cap < need) { give_buf(scratch); scratch = take_buf(need); }
            if (!scratch.p) { errs[t] = ENOMEM; break; }
            char* w = scratch.p;
            for (size_t k = b; k < e; ++k) w = merge_emit_block(m, pl, k, w);
            const char* p = scratch.p; size_t left = need; uint64_t o = base + pl.off[b];
            while (left) {
                ssize_t wr = pwrite(fd, p, left, (off_t)o);
                if (wr < 0) { if (errno == EINTR) continue; errs[t] = errno; break; }
                p += wr; left -= (size_t)wr; o += (uint64_t)wr;
            }
            b = e;
        }
        give_buf(scratch);
    });
    int e = 0;
    for (int x : errs) if (x) e = x;
    if (close(fd) != 0 && !e) e = errno;
    if (e) { (void)unlink(tmp.c_str()); return fail(PM_EIO, "writing '%s': %s", tmp.c_str(), strerror(e)); }
    if (finish && rename(tmp.c_str(), path) != 0) return fail(PM_EIO, "rename to '%s': %s", path, strerror(errno));
    if (bytes) *bytes = pl.off.back();
    return PM_OK;
}

extern "C" void pm_merge_free(pm_merge_t* m) { delete m; }


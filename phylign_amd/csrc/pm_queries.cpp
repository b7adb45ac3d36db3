// pm_queries.cpp -- query sets (a5 input): FASTA reader with the cobs CLI's record rules,
// HBM copies, the canonicalise + XXH64 kernel launch.
#include "pm_host.h"

// ------------------------------------------------------------------ queries
// Record rules of `cobs query -f` (upstream src/main.cpp process_query): see
// include/phylign_match.h.  The input contract (upper-case ACGT, single line)
// is produced by Snakefile:314-333.
// appends one record; PM_EQUERY / PM_ERANGE as documented in the header
static int add_record(pm_queries* q, const std::string& hdr, const std::string& seq, bool headerless) {
    const uint32_t term_size = q->k;
    if (seq.size() < term_size)
        return fail(PM_EQUERY, "query '%s' too short: %zu < %u characters", hdr.c_str(), seq.size(), term_size);
    for (size_t i = 0; i < seq.size(); ++i) {
        const char c = seq[i];
        if (c != 'A' && c != 'C' && c != 'G' && c != 'T')
            return fail(PM_EQUERY, "query '%s': byte 0x%02x at position %zu is not one of ACGT "
                        "(Phylign's fix_query step maps such bases to A: pm_queries_parse_raw(normalise = 1))", hdr.c_str(), (unsigned char)c, i);
    }
    const uint64_t nt = seq.size() - term_size + 1;
    if (nt >= (1ull << 24)) return fail(PM_ERANGE, "query '%s' has %llu k-mers; this build supports < 2^24 per query",
                                        hdr.c_str(), (unsigned long long)nt);
    q->headers.push_back(hdr);
    q->headerless.push_back(headerless ? 1 : 0);
    q->n_terms.push_back((uint32_t)nt);
    q->seq_off.push_back(q->seqs.size());
    q->seqs += seq;
    return PM_OK;
}

// descriptors, 8-slot blocks and counter-width classes of the records added so far
static int finish_queries(pm_queries* q) {
    const size_t nq = q->headers.size();
    q->seq_off.push_back(q->seqs.size());
    if (nq >= 0xFFFFFFFFull) return fail(PM_ERANGE, "too many queries");
    q->qd.resize(nq);
    uint64_t blk = 0;
    for (size_t i = 0; i < nq; ++i) {
        q->qd[i].n_terms = q->n_terms[i];
        q->qd[i].pad_blk = (uint32_t)blk;
        q->qd[i].seq_lo = (uint32_t)q->seq_off[i];
        q->qd[i].seq_hi = (uint32_t)(q->seq_off[i] >> 32);
        blk += (q->n_terms[i] + 7) / 8;
        q->total_terms += q->n_terms[i];
        if (blk >= 0xFFFFFFFFull) return fail(PM_ERANGE, "query set too large (>= 2^35 padded k-mers)");
    }
    q->n_slots = blk * 8;
    std::vector<uint32_t> blkq((size_t)blk);
    for (size_t i = 0; i < nq; ++i) {
        uint64_t b0 = q->qd[i].pad_blk, nb = (q->n_terms[i] + 7) / 8;
        for (uint64_t b = 0; b < nb; ++b) blkq[(size_t)(b0 + b)] = (uint32_t)i;
    }
    // plane classes (counter width): query ids partitioned by class; inside a class the longest queries come first
    // (ties in file order).  A wavefront runs as long as its longest query, so neighbours in qmap should be of similar
    // length, and the long ones should not be the last to start.  Reads of one length (config 3) keep their file order.
    auto cls = [](uint32_t nt) {
        for (int c = 0; c < kNumClasses; ++c) if (nt < (1u << kPlaneClass[c])) return c;
        return kNumClasses - 1;
    };
    q->qmap.reserve(nq);
    for (int c = 0; c < kNumClasses; ++c) {
        q->class_begin[c] = (uint32_t)q->qmap.size();
        bool mixed = false;
        uint32_t first_nt = 0;
        for (size_t i = 0; i < nq; ++i) if (cls(q->n_terms[i]) == c) {
            if (q->qmap.size() == q->class_begin[c]) first_nt = q->n_terms[i];
            else if (q->n_terms[i] != first_nt) mixed = true;
            q->qmap.push_back((uint32_t)i);
        }
        static const bool sort_on = !(getenv("PM_QMAP_SORT") && atoi(getenv("PM_QMAP_SORT")) == 0);   // 0: file order (for the A/B)
        if (mixed && sort_on)
            std::stable_sort(q->qmap.begin() + q->class_begin[c], q->qmap.end(),
                             [&](uint32_t x, uint32_t y) { return q->n_terms[x] > q->n_terms[y]; });
    }
    q->class_begin[kNumClasses] = (uint32_t)q->qmap.size();
    q->blkq.swap(blkq);
    // HBM copies are made on first use by a compute call (upload_queries): parsing, text
    // formatting and the 04_filter merge are host work and need no GPU
    return PM_OK;
}

// records of fasta[begin, end) by the cobs CLI's rules into q (a scratch object when several threads parse a file);
// have_any: a header line has been seen before `begin` (false only at the start of the file)
static int parse_cobs_range(pm_queries* q, const char* fasta, size_t begin, size_t end, bool have_any) {
    std::string cur_hdr, cur_seq;
    int rc = PM_OK;
    auto flush = [&]() -> int {
        if (cur_seq.empty()) return PM_OK;
        return add_record(q, cur_hdr, cur_seq, !have_any);
    };
    size_t p = begin;
    while (p < end && rc == PM_OK) {
        const char* nl = (const char*)memchr(fasta + p, '\n', end - p);
        size_t ll = nl ? (size_t)(nl - (fasta + p)) : end - p;
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
    return rc;
}

// Offsets at which a prepared query file (records start with '>' or ';' at a line start) is cut into pieces of
// `max_records` records: what match_stage --query-chunk needs for a file of more reads than fit HBM at once.  Record
// starts are counted on several threads, then the cuts are located inside the pieces that hold them.
extern "C" int pm_fasta_record_cuts(const char* fasta, size_t len, uint64_t max_records, uint64_t** cuts, uint64_t* n_cuts) try {
    if ((!fasta && len) || !cuts || !n_cuts) return fail(PM_EINVAL, "bad argument");
    *cuts = nullptr; *n_cuts = 0;
    if (max_records == 0 || len == 0) return PM_OK;
    const size_t np = std::max<size_t>(1, (len + (8u << 20) - 1) / (8u << 20));
    // record starts in [a, b): counted; with `first` (the 1-based ordinal of the piece's first record) the starts whose
    // ordinal is 1 modulo max_records (the first record of a later piece) are noted
    auto walk = [&](size_t a, size_t b, uint64_t first, std::vector<uint64_t>* note) -> uint64_t {
        uint64_t n = 0;
        auto seen = [&](size_t at) {
            if (note && first + n > 1 && (first + n - 1) % max_records == 0) note->push_back((uint64_t)at);
            ++n;
        };
        size_t p = a;
        if (p == 0) {
            if (fasta[0] == '>' || fasta[0] == ';') seen(0);
            p = 1;
        }
        while (p < b) {                                          // a record starts at p when a newline stands at p - 1
            const char* nl = (const char*)memchr(fasta + p - 1, '\n', b - p);
            if (!nl) break;
            p = (size_t)(nl - fasta) + 1;
            if (fasta[p] == '>' || fasta[p] == ';') seen(p);
            ++p;
        }
        return n;
    };
    std::vector<uint64_t> cnt(np, 0);
    parallel_for(np, [&](size_t t) { cnt[t] = walk(len * t / np, len * (t + 1) / np, 0, nullptr); });
    std::vector<uint64_t> first(np, 1);
    for (size_t t = 1; t < np; ++t) first[t] = first[t - 1] + cnt[t - 1];
    std::vector<std::vector<uint64_t>> found(np);
    parallel_for(np, [&](size_t t) { (void)walk(len * t / np, len * (t + 1) / np, first[t], &found[t]); });
    std::vector<uint64_t> out;
    for (auto& f : found) out.insert(out.end(), f.begin(), f.end());
    if (out.empty()) return PM_OK;
    uint64_t* buf = (uint64_t*)malloc(out.size() * sizeof(uint64_t));
    if (!buf) return fail(PM_ENOMEM, "out of host memory");
    memcpy(buf, out.data(), out.size() * sizeof(uint64_t));
    *cuts = buf; *n_cuts = out.size();
    return PM_OK;
} PM_GUARD_END

extern "C" int pm_queries_parse(const char* fasta, size_t len, uint32_t term_size, pm_queries_t** out) try {
    if ((!fasta && len) || !out || term_size == 0) return fail(PM_EINVAL, "bad argument");
    pm_queries* q = new pm_queries();
    q->k = term_size;
    int rc = PM_OK;
    // a million reads are 160 MB of text: the file is cut at header lines and the pieces are parsed on several threads
    size_t nt = std::min<size_t>(parallel_width(), len / (8u << 20));
    std::vector<size_t> cut{0};
    for (size_t t = 1; t < nt; ++t) {
        size_t p = std::max(len * t / nt, cut.back());
        size_t b = len;
        while (p < len) {                                      // next line that starts a record
            const char* nl = (const char*)memchr(fasta + p, '\n', len - p);
            if (!nl) break;
            p = (size_t)(nl - fasta) + 1;
            if (p < len && (fasta[p] == '>' || fasta[p] == ';')) { b = p; break; }
        }
        if (b > cut.back() && b < len) cut.push_back(b);
    }
    cut.push_back(len);
    if (cut.size() <= 2) {
        rc = parse_cobs_range(q, fasta, 0, len, false);
    } else {
        const size_t n = cut.size() - 1;
        std::vector<pm_queries> part(n);
        std::vector<int> rcs(n, PM_OK);
        std::vector<std::string> errs(n);
        auto work = [&](size_t t) {
            part[t].k = term_size;
            rcs[t] = parse_cobs_range(&part[t], fasta, cut[t], cut[t + 1], t > 0);
            if (rcs[t] != PM_OK) errs[t] = pm_last_error();     // the message is thread-local: carry it to the caller's thread
        };
        parallel_for(n, work);
        size_t total_seq = 0, total_rec = 0;
        for (size_t t = 0; t < n && rc == PM_OK; ++t) {
            if (rcs[t] != PM_OK) rc = fail(rcs[t], "%s", errs[t].c_str());      // the first failing record in file order
            total_seq += part[t].seqs.size(); total_rec += part[t].headers.size();
        }
        if (rc == PM_OK) {
            // the pieces' records go to their places in the whole on the threads that parsed them
            std::vector<size_t> rec0(n + 1, 0), seq0(n + 1, 0);
            for (size_t t = 0; t < n; ++t) { rec0[t + 1] = rec0[t] + part[t].headers.size(); seq0[t + 1] = seq0[t] + part[t].seqs.size(); }
            q->seqs.resize(total_seq);
            q->headers.resize(total_rec); q->headerless.resize(total_rec); q->n_terms.resize(total_rec); q->seq_off.resize(total_rec);
            parallel_for(n, [&](size_t t) {
                memcpy(&q->seqs[seq0[t]], part[t].seqs.data(), part[t].seqs.size());
                for (size_t i = 0; i < part[t].headers.size(); ++i) {
                    q->headers[rec0[t] + i] = std::move(part[t].headers[i]);
                    q->seq_off[rec0[t] + i] = seq0[t] + part[t].seq_off[i];
                }
                std::copy(part[t].headerless.begin(), part[t].headerless.end(), q->headerless.begin() + (long)rec0[t]);
                std::copy(part[t].n_terms.begin(), part[t].n_terms.end(), q->n_terms.begin() + (long)rec0[t]);
                part[t] = pm_queries();                       // the piece's memory goes back here, not serially at the end
            });
        }
    }
    if (rc == PM_OK) rc = finish_queries(q);
    if (rc != PM_OK) { delete q; return rc; }
    *out = q;
    return PM_OK;
} PM_GUARD_END

// Rules `fix_query` + `concatenate_queries` (Snakefile:314-352) fused into the parser: what
//   seqtk seq -A -U -C in | awk '{if(NR%2==1){print $0;}else{gsub(/[^ACGT]/, "A"); print;}}'
// writes, read with the record rules of kseq.h's kseq_read() (seqtk's reader; klib, restated here step by step):
//   * at the start of the input and after every FASTQ record the reader jumps to the next '>' or '@' -- wherever in a
//     line it stands; after a FASTA record the header is the line whose first byte ended the sequence;
//   * name = the header up to its first white-space byte (isspace: blank, TAB, CR, VT, FF, newline) -- the rest of the
//     line, the comment, is dropped (-C);
//   * sequence = the following lines concatenated, up to a line that starts with '>', '@' or '+';
//     empty lines are skipped, a trailing '\r' is dropped;
//   * '+' starts a FASTQ quality block: the rest of that line is skipped, then quality lines are
//     read until they hold at least as many characters as the sequence (so '@' or '>' as a quality
//     value never starts a record); a block that ends early or runs long ends the input there,
//     like seqtk's read loop (the broken record is not printed);
//   * -U upper-cases, awk maps every byte that is not A, C, G or T to 'A';
//   * a record without sequence is printed by seqtk as an empty line and ignored by cobs: dropped.
// A sequence shorter than term_size stays an error (PM_EQUERY), exactly as for a prepared file.
// The Python mirror (phylign_amd/fix_query.py) restates the same steps; the two are compared on random well- and
// ill-formed input (tests/test_golden_cpu.py).
static int parse_raw_normalised(pm_queries* q, const char* buf, size_t len) {
    auto is_space = [](char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\v' || c == '\f' || c == '\r'; };
    size_t p = 0;
    int last = 0;                                             // kseq's last_char: a header byte that was read already
    std::string name, seq;
    for (;;) {
        if (last == 0) {                                      // jump to the next header byte, anywhere
            while (p < len && buf[p] != '>' && buf[p] != '@') ++p;
            if (p >= len) break;
            ++p;
        }
        if (p >= len) break;                                  // a header byte at the very end: no name to read
        size_t e = p;
        while (e < len && !is_space(buf[e])) ++e;
        name.assign(buf + p, e - p);
        seq.clear();
        int c = -1;
        if (e < len) {
            p = e + 1;
            if (buf[e] != '\n') {                             // the comment: to the end of the line
                const char* nl = (const char*)memchr(buf + p, '\n', len - p);
                p = nl ? (size_t)(nl - buf) + 1 : len;
            }
            // sequence lines
            while (p < len) {
                c = (unsigned char)buf[p];
                if (c == '>' || c == '+' || c == '@') { ++p; break; }
                if (c == '\n') { ++p; c = -1; continue; }
                const char* nl = (const char*)memchr(buf + p, '\n', len - p);
                const size_t end = nl ? (size_t)(nl - buf) : len;
                seq.append(buf + p, end - p);
                p = nl ? end + 1 : len;
                if (seq.size() > 1 && seq.back() == '\r') seq.pop_back();
                c = -1;
            }
        } else {
            p = len;
        }
        if (c == '>' || c == '@') last = c;
        bool print = true, stop = c == -1;                    // the input ended inside or right after this record
        if (c == '+') {
            const char* nl = (const char*)memchr(buf + p, '\n', len - p);      // the rest of the '+' line
            if (!nl) break;                                   // no quality string: kseq_read returns -2
            p = (size_t)(nl - buf) + 1;
            size_t qual = 0;
            while (p < len) {
                const char* ql = (const char*)memchr(buf + p, '\n', len - p);
                const size_t end = ql ? (size_t)(ql - buf) : len;
                size_t l = end - p;
                if (qual + l > 1 && l && buf[end - 1] == '\r') --l;
                qual += l;
                p = ql ? end + 1 : len;
                if (!(qual < seq.size())) break;
            }
            last = 0;
            if (qual != seq.size()) break;                    // truncated or overlong quality: kseq_read returns -2
            stop = false;
            (void)print;
        }
        for (char& ch : seq) {
            if (ch >= 'a' && ch <= 'z') ch = (char)(ch - 32);
            if (ch != 'A' && ch != 'C' && ch != 'G' && ch != 'T') ch = 'A';
        }
        if (!seq.empty()) { int rc = add_record(q, name, seq, false); if (rc) return rc; }
        if (stop) break;
    }
    return PM_OK;
}

extern "C" int pm_queries_parse_raw(const char* buf, size_t len, uint32_t term_size, int normalise, pm_queries_t** out) try {
    if (!normalise) return pm_queries_parse(buf, len, term_size, out);
    if ((!buf && len) || !out || term_size == 0) return fail(PM_EINVAL, "bad argument");
    pm_queries* q = new pm_queries();
    q->k = term_size;
    int rc = parse_raw_normalised(q, buf, len);
    if (rc == PM_OK) rc = finish_queries(q);
    if (rc != PM_OK) { delete q; return rc; }
    *out = q;
    return PM_OK;
} PM_GUARD_END

// ">header\nSEQUENCE\n" per record: the prepared query file (intermediate/01_queries_merged/*.fa) this set stands for
extern "C" int pm_queries_fasta(const pm_queries_t* q, char** text, size_t* len) try {
    if (!q || !text || !len) return fail(PM_EINVAL, "bad argument");
    size_t total = 0;
    for (size_t i = 0; i < q->headers.size(); ++i) total += q->headers[i].size() + 3 + (size_t)(q->seq_off[i + 1] - q->seq_off[i]);
    char* buf = (char*)malloc(total + 1);
    if (!buf) return fail(PM_ENOMEM, "out of host memory");
    size_t o = 0;
    for (size_t i = 0; i < q->headers.size(); ++i) {
        if (!q->headerless[i]) buf[o++] = '>';
        memcpy(buf + o, q->headers[i].data(), q->headers[i].size()); o += q->headers[i].size();
        buf[o++] = '\n';
        const size_t sl = (size_t)(q->seq_off[i + 1] - q->seq_off[i]);
        memcpy(buf + o, q->seqs.data() + q->seq_off[i], sl); o += sl;
        buf[o++] = '\n';
    }
    buf[o] = 0;
    *text = buf; *len = o;
    return PM_OK;
} PM_GUARD_END

extern "C" int pm_queries_count(const pm_queries_t* q, uint64_t* n_queries, uint64_t* n_terms) try {
    if (!q) return fail(PM_EINVAL, "bad argument");
    if (n_queries) *n_queries = q->headers.size();
    if (n_terms) *n_terms = q->total_terms;
    return PM_OK;
} PM_GUARD_END
extern "C" int pm_queries_terms(const pm_queries_t* q, uint64_t i, uint64_t* n_terms) try {
    if (!q || i >= q->n_terms.size() || !n_terms) return fail(PM_EINVAL, "bad argument");
    *n_terms = q->n_terms[(size_t)i];
    return PM_OK;
} PM_GUARD_END
// HBM copies of query sets come from a small pool and go back to it.  hipFree waits for the whole device -- for the
// scan of the NEXT (group, chunk) unit that match_stage has queued already -- so a chunk of a large query file that gives
// its device copies back (pm_queries_release_device) would otherwise serialise the chunk pipeline to depth one.  The
// next chunk is of about the same size and takes the buffers over.  At most kQPoolMax buffers wait here; beyond that
// the smallest one is really freed.  release_query_pool(): pm_shutdown.
namespace {
struct QBuf { void* p; size_t bytes; };
std::mutex g_qpool_mu;
std::vector<QBuf> g_qpool;
constexpr size_t kQPoolMax = 16;
}
int query_buf_take(size_t bytes, void** out) {
    bytes = std::max<size_t>(bytes, 256);
    {
        std::lock_guard<std::mutex> lk(g_qpool_mu);
        size_t best = SIZE_MAX;
        for (size_t i = 0; i < g_qpool.size(); ++i)
            if (g_qpool[i].bytes >= bytes && g_qpool[i].bytes <= bytes + bytes / 2 + 4096 &&
                (best == SIZE_MAX || g_qpool[i].bytes < g_qpool[best].bytes)) best = i;
        if (best != SIZE_MAX) {
            *out = g_qpool[best].p;
            g_qpool.erase(g_qpool.begin() + (long)best);
            return PM_OK;
        }
    }
    const hipError_t e = device_malloc_reclaim(out, bytes);          // out of memory: the pools are emptied, one more try
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? PM_ENOMEM : PM_EHIP, "hipMalloc(%zu bytes) for a query set: %s", bytes, hipGetErrorString(e));
    return PM_OK;
}
// bytes = what qbuf_take was asked for (the pool keeps the allocation's real size only approximately: never less)
static void qbuf_give(void* p, size_t bytes) {
    if (!p) return;
    QBuf b{p, std::max<size_t>(bytes, 256)};
    {
        std::lock_guard<std::mutex> lk(g_qpool_mu);
        if (g_ctx.ready) {
            g_qpool.push_back(b);
            if (g_qpool.size() <= kQPoolMax) return;
            size_t small = 0;
            for (size_t i = 1; i < g_qpool.size(); ++i) if (g_qpool[i].bytes < g_qpool[small].bytes) small = i;
            b = g_qpool[small];
            g_qpool.erase(g_qpool.begin() + (long)small);
        }
    }
    (void)hipFree(b.p);
}
void release_query_pool() {
    std::vector<QBuf> all;
    { std::lock_guard<std::mutex> lk(g_qpool_mu); all.swap(g_qpool); }
    for (auto& b : all) (void)hipFree(b.p);
}
// free_now: the device is going away or the caller wants the memory back for good (pm_queries_free); otherwise the
// buffers wait in the pool for the next query set
static void drop_device_state(pm_queries* q, bool free_now) {
    const size_t nq = q->headers.size();
    auto give = [&](void* p, size_t bytes) { if (!p) return; if (free_now) (void)hipFree(p); else qbuf_give(p, bytes); };
    give(q->d_seq, q->seqs.size() + 64);
    give(q->d_qd, nq * sizeof(QDesc));
    give(q->d_blkq, std::max<size_t>(q->blkq.size(), 1) * 4);
    give(q->d_qmap, nq * 4);
    give(q->d_thr, nq * 4);
    for (auto& h : q->hashes) give(h.d, (size_t)(q->n_slots * h.nh * 8));
    q->d_seq = nullptr; q->d_qd = nullptr; q->d_blkq = nullptr; q->d_qmap = nullptr; q->d_thr = nullptr;
    q->thr_for = -1.0;
    q->hashes.clear();
    q->on_device = false;
}
extern "C" void pm_queries_free(pm_queries_t* q) {
    if (!q) return;
    bind_thread_quiet();
    if (g_ctx.ready && q->on_device) (void)hipStreamSynchronize(g_ctx.stream);   // a search in flight may still read them
    drop_device_state(q, true);
    if (q->last_use) (void)hipEventDestroy(q->last_use);
    delete q;
}
// Frees the HBM copies of the query set (sequences, descriptors, per-query thresholds, hash buffers: ~8 bytes per k-mer
// and hash function) and keeps everything on the host -- names and sequences, which results, texts and the 04_filter merge
// refer to.  The next search uploads them again.  For a query file that is searched chunk after chunk: a chunk that is
// not searched for a while need not stay resident.  Waits only for the searches that use THIS query set.
extern "C" int pm_queries_release_device(pm_queries_t* q) try {
    if (!q) return fail(PM_EINVAL, "bad argument");
    if (!q->on_device) return PM_OK;
    NEED_DEV();
    if (q->last_use) HIPCHK(hipEventSynchronize(q->last_use));
    else HIPCHK(hipStreamSynchronize(g_ctx.stream));
    drop_device_state(q, false);                  // into the pool: no hipFree, nothing waits for the searches queued behind
    return PM_OK;
} PM_GUARD_END
// *resident: HBM bytes the query set holds right now; *when_searched: what it holds while it is searched against indexes
// of num_hashes hash functions (budgeting: match_stage keeps this out of the index admission budget)
extern "C" int pm_queries_device_bytes(const pm_queries_t* q, uint32_t num_hashes, uint64_t* resident, uint64_t* when_searched) try {
    if (!q || num_hashes == 0) return fail(PM_EINVAL, "bad argument");
    const uint64_t nq = q->headers.size();
    const uint64_t base = q->seqs.size() + 64 + nq * sizeof(QDesc) + std::max<uint64_t>(q->blkq.size(), 1) * 4 + nq * 4;
    if (resident) {
        uint64_t r = q->on_device && nq ? base : 0;
        if (q->d_thr) r += nq * 4;
        for (auto& h : q->hashes) if (h.d) r += q->n_slots * h.nh * 8;
        *resident = r;
    }
    if (when_searched) *when_searched = base + nq * 4 + q->n_slots * (uint64_t)num_hashes * 8;
    return PM_OK;
} PM_GUARD_END

int upload_queries(pm_queries* q) {
    if (q->on_device) return PM_OK;
    const size_t nq = q->headers.size();
    if (nq) {
        int rc;
        if ((rc = query_buf_take(q->seqs.size() + 64, (void**)&q->d_seq))) return rc;
        HIPCHK(hipMemset(q->d_seq + q->seqs.size(), 0, 64));          // the padding behind the sequences
        if ((rc = query_buf_take(nq * sizeof(QDesc), (void**)&q->d_qd))) return rc;
        if ((rc = query_buf_take(std::max<size_t>(q->blkq.size(), 1) * 4, (void**)&q->d_blkq))) return rc;
        if ((rc = query_buf_take(nq * 4, (void**)&q->d_qmap))) return rc;
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
int ensure_hashes(pm_queries* q, int canon, uint32_t nh, uint64_t** out) {
    pm_queries::HashBuf* hb = nullptr;
    for (auto& h : q->hashes) if (h.canon == canon && h.nh == nh) hb = &h;
    if (!hb) {
        q->hashes.push_back({canon, nh, nullptr, ~0ull});
        hb = &q->hashes.back();
        if (q->n_slots) { int rc = query_buf_take((size_t)(q->n_slots * nh * 8), (void**)&hb->d); if (rc) return rc; }
    }
    if (hb->epoch != q->epoch) {
        HIPCHK(launch_hash_terms(q->d_seq, q->d_qd, q->d_blkq, q->n_slots, q->k, canon, nh, hb->d, g_ctx.stream));
        hb->epoch = q->epoch;
    }
    *out = hb->d;
    return PM_OK;
}

extern "C" int pm_hash_terms(pm_queries_t* q, int canonicalize, uint32_t num_hashes, uint64_t* out) try {
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
} PM_GUARD_END


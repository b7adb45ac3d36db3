// pm_queries.cpp -- query sets (a5 input): FASTA reader with the cobs CLI's record rules,
// HBM copies, the canonicalise + XXH64 kernel launch.
#include "pm_host.h"

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
    auto cls = [](uint32_t nt) {
        for (int c = 0; c < kNumClasses; ++c) if (nt < (1u << kPlaneClass[c])) return c;
        return kNumClasses - 1;
    };
    q->qmap.reserve(nq);
    for (int c = 0; c < kNumClasses; ++c) {
        q->class_begin[c] = (uint32_t)q->qmap.size();
        for (size_t i = 0; i < nq; ++i) if (cls(q->n_terms[i]) == c) q->qmap.push_back((uint32_t)i);
    }
    q->class_begin[kNumClasses] = (uint32_t)q->qmap.size();

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
    if (g_ctx.ready && q->on_device) (void)hipStreamSynchronize(g_ctx.stream);   // a search in flight may still read them
    if (q->d_seq) (void)hipFree(q->d_seq);
    if (q->d_qd) (void)hipFree(q->d_qd);
    if (q->d_blkq) (void)hipFree(q->d_blkq);
    if (q->d_qmap) (void)hipFree(q->d_qmap);
    if (q->d_thr) (void)hipFree(q->d_thr);
    for (auto& h : q->hashes) if (h.d) (void)hipFree(h.d);
    delete q;
}

int upload_queries(pm_queries* q) {
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
int ensure_hashes(pm_queries* q, int canon, uint32_t nh, uint64_t** out) {
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


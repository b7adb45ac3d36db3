// pm_search.cpp -- search orchestration (a5-a7): launch plan, searches in flight
// (pm_search_async / pm_result_wait), device-side run ordering, result getters.
#include "pm_host.h"

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
        size_t best = g_ctx.free_hits.size();          // the smallest pooled buffer that is big enough (a small request must
        for (size_t i = 0; i < g_ctx.free_hits.size(); ++i)   // not take the buffer the next large search needs)
            if (g_ctx.free_hits[i].cap >= cap && (best == g_ctx.free_hits.size() || g_ctx.free_hits[i].cap < g_ctx.free_hits[best].cap))
                best = i;
        if (best != g_ctx.free_hits.size()) {
            *out = g_ctx.free_hits[best];
            g_ctx.free_hits.erase(g_ctx.free_hits.begin() + (long)best);
            return PM_OK;
        }
    }
    out->cap = cap;
    // `cap` records followed by the run directory (a run holds at least two records)
    const hipError_t e = device_malloc_reclaim((void**)&out->p, (cap + run_cap_of(cap)) * sizeof(uint4));
    if (e != hipSuccess) {
        out->p = nullptr;
        return fail(e == hipErrorOutOfMemory ? PM_ENOMEM : PM_EHIP, "hipMalloc(%zu bytes) for a hit buffer: %s",
                    (size_t)((cap + run_cap_of(cap)) * sizeof(uint4)), hipGetErrorString(e));
    }
    return PM_OK;
}
void release_hit_pool() {
    std::vector<HitBuf> all;
    { std::lock_guard<std::mutex> lk(g_pool_mu); all.swap(g_ctx.free_hits); }
    for (auto& b : all) (void)hipFree(b.p);
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
    (void)hipFree(b.p);
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
    (void)hipHostFree(b.p);
}

// One scan unit per classic index or per sub-index of a compact index.
struct Unit { const pm_index* ix; uint32_t slot, doc_base; bool prune; pm_qpart_t part; };
struct Group { int g, canon; uint32_t nh, slabs; std::vector<size_t> members; bool has_parts; };
// positions [first, end) of a class of n queries that part `p` of an index covers (den == 0: all of them)
static inline void part_range(const pm_qpart_t& p, uint32_t n, uint32_t* first, uint32_t* end) {
    if (p.den == 0) { *first = 0; *end = n; return; }
    *first = (uint32_t)((uint64_t)p.lo * n / p.den);
    *end = (uint32_t)((uint64_t)p.hi * n / p.den);
}

// results alive: the two switchable cobs rules are captured per search for the device-side ordering, while the host
// comparator (hit_less) and pm_threshold_terms read the process-wide setting -- pm_set_option refuses to change either
// rule while a result exists, so the two can never disagree about records that are still around
std::atomic<int> g_live_results{0};
struct pm_result {
    pm_result() { g_live_results.fetch_add(1, std::memory_order_relaxed); }
    ~pm_result() { g_live_results.fetch_sub(1, std::memory_order_relaxed); }
    pm_result(const pm_result&) = delete;
    pm_result& operator=(const pm_result&) = delete;
    // what was asked (kept for the one re-run after a hit-buffer overflow)
    std::vector<pm_index_t*> idx;
    pm_queries* q = nullptr;
    double threshold = 0;
    uint32_t nb_best = 0, slot_base = 0;
    uint32_t tie_desc = 0;                        // the "cobs_tie_order" in force when the search was queued
    std::vector<pm_qpart_t> parts;                // empty, or the part of the queries every index is searched with
    // device output
    uint4* d_hits = nullptr;
    uint64_t cap = 0;
    uint64_t n_records = 0, n_runs = 0;
    pm_stats_t st{};
    std::vector<pm_launch_t> launches;
    // in-flight state
    Workspace* ws = nullptr;
    bool pending = false;
    int failed = 0;                               // error code of a wait that failed: every later getter reports it
    int attempt = 0;
    size_t nev = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> lev;
    unsigned long long* h_fetch = nullptr;       // pinned [launch][kFetchShards] when "count_fetched" is on
    // ordered form (ensure_ordered): records permuted on the device into (slot, query) run order
    std::atomic<bool> ordered{false};              // double-checked under g_order_mu
    HitBuf d_ord{nullptr, 0};
    uint64_t n_out = 0;
    std::vector<uint64_t> slot_first;             // first ordered record of every index of the search (+ the end): the slices of pm_result_slot_hits
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
    if (r->h_fetch) { (void)hipHostFree(r->h_fetch); r->h_fetch = nullptr; }
}

// Enqueues hash + scan launches + the read-back of the record counters on the compute
// stream; returns without waiting for the GPU.
static int enqueue_search(pm_result* r, uint64_t want_cap) {
    pm_queries* q = r->q;
    const size_t n_idx = r->idx.size();
    std::vector<Unit> units;
    for (size_t s = 0; s < n_idx; ++s) {
        const pm_index* ix = r->idx[s];
        const pm_qpart_t part = r->parts.empty() ? pm_qpart_t{0, 0, 0} : r->parts[s];
        if (ix->parts.empty()) {
            if (!ix->d_matrix) return fail(PM_EINVAL, "index %zu has no matrix", s);
            // names without '_' make the reference's post-filter raise on a line it would otherwise
            // drop (scripts/postprocess_cobs.py:10-18): such an index is cut on the host, where
            // every document that passed -t is seen, so both ways to prune fail alike
            units.push_back({ix, r->slot_base + (uint32_t)s, 0u, ix->names_have_sep, part});
        } else {
            // the n best documents of a compact index span its sub-indexes: cut when formatting
            for (size_t p = 0; p < ix->parts.size(); ++p) {
                const pm_index* sub = ix->parts[p];
                if (sub->info.n_docs == 0) continue;
                if (!sub->d_matrix) return fail(PM_EINVAL, "index %zu has no matrix", s);
                units.push_back({sub, r->slot_base + (uint32_t)s, (uint32_t)(p * ix->page_size * 8), false, part});
            }
        }
    }
    const size_t nq = q->headers.size();
    hipStream_t st = g_ctx.stream;
    { int urc = upload_queries(q); if (urc) return urc; }
    if (q->term_prefix.size() != nq + 1) {            // k-mers ahead of every position of the class-ordered query list
        q->term_prefix.assign(nq + 1, 0);
        for (size_t i = 0; i < nq; ++i) q->term_prefix[i + 1] = q->term_prefix[i] + q->n_terms[q->qmap[i]];
    }

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
        if (!gp) { groups.push_back({key, (int)ix->info.canonicalize, ix->info.num_hashes, ix->slabs, {}, false}); gp = &groups.back(); }
        gp->members.push_back(u);
        if (units[u].part.den) gp->has_parts = true;
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
        if (ws->d_desc) (void)hipFree(ws->d_desc);
        if (ws->h_desc) (void)hipHostFree(ws->h_desc);
        ws->d_desc = nullptr; ws->h_desc = nullptr; ws->desc_cap = 0;
        ws->uploaded.clear();
        const size_t cap = std::max<size_t>(n_units, 64);
        HIPCHK(hipMalloc((void**)&ws->d_desc, (1 + kNumClasses) * cap * sizeof(BatchDesc)));
        HIPCHK(hipHostMalloc((void**)&ws->h_desc, (1 + kNumClasses) * cap * sizeof(BatchDesc), hipHostMallocDefault));
        ws->desc_cap = cap;
    }
    // host image of all five slices: base descriptors, then per query class the block ranges of mixed launches
    const size_t dcap = ws->desc_cap;
    memset(ws->h_desc, 0, (1 + kNumClasses) * dcap * sizeof(BatchDesc));
    {
        size_t o = 0;
        for (auto& g : groups)
            for (size_t u : g.members) {
                const pm_index* ix = units[u].ix;
                BatchDesc& d = ws->h_desc[o++];
                d.matrix = ix->d_matrix; d.stride = ix->info.stride; d.sig_size = ix->info.signature_size;
                d.barrett_m = barrett_m(ix->info.signature_size); d.n_docs = ix->info.n_docs;
                d.slot = units[u].slot; d.doc_base = units[u].doc_base; d.prune = units[u].prune ? 1u : 0u;
                d.lanes = (uint32_t)ix->g; d.block_begin = 0; d.q_first = 0; d.q_end = 0xFFFFFFFFu;
            }
    }
    // wide-query form per (group, query class): when one lane group per query would leave the chip
    // mostly idle (few, long queries), min(8, 256 / lanes) groups share each query (k_scan<..., WQ>)
    // 0 = plain form, else the number of lane groups that may share a query (power of two, >= 8)
    std::vector<uint32_t> use_wq(groups.size() * kNumClasses, 0);
    for (size_t gi = 0; gi < groups.size(); ++gi)
        for (int c = 0; c < kNumClasses; ++c) {
            if (kPlaneClass[c] < 10) continue;              // the form exists for classes of 128+ k-mers per query
            const uint32_t nqc = q->class_begin[c + 1] - q->class_begin[c];
            if (nqc == 0 || g_wide_query == 2) continue;
            uint64_t waves = 0, lanes = 0;
            for (size_t u : groups[gi].members) {
                const uint32_t qpb = scan_queries_per_block(units[u].ix->g, 0);
                waves += 4ull * ((nqc + qpb - 1) / qpb) * units[u].ix->slabs;
                lanes += (uint64_t)units[u].ix->g * units[u].ix->slabs;
            }
            // automatic: always when the plain form could not fill the chip (< 1.5 x the 4 096 wave slots of
            // 4 waves/SIMD); narrow rows (one query per lane group of 1 ... 16 lanes) measure faster in this
            // form at any query count, but only while every row is fetched anyway -- the form gives up the
            // threshold bound (tools/variant_longq.sh, profiles/r03/NOTES.md section 6)
            const bool few = waves < 6144;
            const bool narrow = groups[gi].g < 32;
            if (!(g_wide_query == 1 || few || (narrow && !g_threshold_bound))) continue;
            // groups per query: 8, or as many as it takes to put ~8 192 wavefronts on the chip (narrow rows
            // with very few queries: up to a whole workgroup per query)
            uint64_t want = (8192ull * 64 + (uint64_t)nqc * lanes - 1) / std::max<uint64_t>((uint64_t)nqc * lanes, 1);
            uint32_t grp = 8;
            while (grp < 256 && grp < want) grp <<= 1;
            use_wq[gi * kNumClasses + (size_t)c] = grp;
        }
    // per query class: the workgroup ranges of mixed launches, and -- groups with query parts -- every unit's query range
    // (mixed launch: total workgroups; uniform launch with parts: tiles of the unit with the most queries)
    std::vector<uint32_t> mixed_blocks(groups.size() * kNumClasses, 0u);
    {
        size_t desc_off = 0;
        for (size_t gi = 0; gi < groups.size(); ++gi) {
            Group& g = groups[gi];
            if (g.g == 0 || g.has_parts)
                for (int c = 0; c < kNumClasses; ++c) {
                    const uint32_t b = q->class_begin[c], e = q->class_begin[c + 1];
                    if (e == b) continue;
                    uint64_t blk = 0, most = 0;
                    BatchDesc* stage = ws->h_desc + (size_t)(1 + c) * dcap + desc_off;
                    for (size_t k = 0; k < g.members.size(); ++k) {
                        stage[k] = ws->h_desc[desc_off + k];
                        const uint32_t qpb = scan_queries_per_block((int)stage[k].lanes, use_wq[gi * kNumClasses + (size_t)c]);
                        uint32_t first, end;
                        part_range(units[g.members[k]].part, e - b, &first, &end);
                        stage[k].q_first = first; stage[k].q_end = end;
                        stage[k].block_begin = (uint32_t)blk;
                        const uint64_t t = (end - first + qpb - 1) / qpb;
                        blk += t;
                        most = std::max(most, t);
                    }
                    if (blk > 0x7FFFFFFFull) return fail(PM_ERANGE, "launch grid too large");
                    mixed_blocks[gi * kNumClasses + (size_t)c] = (uint32_t)(g.g == 0 ? blk : most);
                }
            desc_off += g.members.size();
        }
    }
    if (ws->uploaded.size() != (1 + kNumClasses) * dcap || memcmp(ws->uploaded.data(), ws->h_desc, (1 + kNumClasses) * dcap * sizeof(BatchDesc)) != 0) {
        HIPCHK(hipMemcpyAsync(ws->d_desc, ws->h_desc, (1 + kNumClasses) * dcap * sizeof(BatchDesc), hipMemcpyHostToDevice, st));
        ws->uploaded.assign(ws->h_desc, ws->h_desc + (1 + kNumClasses) * dcap);
    }
    // per-query minimum score, cached on the query set per threshold value
    if (nq && (!q->d_thr || q->thr_for != r->threshold || q->thr_rule != g_threshold_rule)) {
        std::vector<uint32_t> thr(nq);
        for (size_t i = 0; i < nq; ++i) thr[i] = r->threshold == 0.0 ? 0u : pm_threshold_terms(r->threshold, q->n_terms[i]);
        if (!q->d_thr) { int trc = query_buf_take(nq * 4, (void**)&q->d_thr); if (trc) return trc; }
        HIPCHK(hipStreamSynchronize(st));                 // an earlier search in flight may still read the old values
        HIPCHK(hipMemcpy(q->d_thr, thr.data(), nq * 4, hipMemcpyHostToDevice));
        q->thr_for = r->threshold; q->thr_rule = g_threshold_rule;
    }
    // measurement option: sharded counters of the algorithmic bytes the scan really gathered
    size_t n_launch_max = 0;
    for (auto& g : groups) { (void)g; n_launch_max += kNumClasses; }
    if (g_count_fetched) {
        if (!g_ctx.d_fetch) HIPCHK(hipMalloc((void**)&g_ctx.d_fetch, kFetchShards * sizeof(unsigned long long)));
        if (r->h_fetch) { (void)hipHostFree(r->h_fetch); r->h_fetch = nullptr; }
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
        for (int c = 0; c < kNumClasses; ++c) {
            const uint32_t b = q->class_begin[c], e = q->class_begin[c + 1];
            if (e == b) continue;
            ScanArgs a;
            a.batches = ws->d_desc + desc_off; a.n_batches = (uint32_t)g.members.size();
            a.tiles = 0; a.total_blocks = 0;
            a.wq_groups = use_wq[(size_t)(&g - groups.data()) * kNumClasses + (size_t)c];
            a.wide_query = a.wq_groups ? 1u : 0u;
            a.nsplit = 1; a.tie_desc = r->tie_desc; a.split_slabs = nullptr; a.split_cnt = nullptr;
            if (g.g > 0 && g.has_parts) {
                // the slice of this query class holds every unit's query range; a unit with fewer queries leaves tiles empty
                a.batches = ws->d_desc + (size_t)(1 + c) * dcap + desc_off;
                a.tiles = mixed_blocks[(size_t)(&g - groups.data()) * kNumClasses + (size_t)c];
                if (a.tiles == 0) continue;
            } else if (g.g > 0) {
                const uint32_t qpb = scan_queries_per_block(g.g, a.wq_groups);
                a.tiles = (e - b + qpb - 1) / qpb;
            } else {
                // mixed widths: the slice of this query class holds the per-batch workgroup ranges
                a.batches = ws->d_desc + (size_t)(1 + c) * dcap + desc_off;
                a.total_blocks = mixed_blocks[(size_t)(&g - groups.data()) * kNumClasses + (size_t)c];
                if (a.total_blocks == 0) continue;
            }
            a.hashes = d_h; a.qd = q->d_qd; a.thr = q->d_thr; a.qmap = q->d_qmap + b; a.nq = e - b;
            a.prune_n = r->nb_best;
            a.bound = g_threshold_bound;
            a.nh = g.nh; a.hits = hb.p; a.hit_count = ws->d_cnt; a.hit_cap = hb.cap;
            a.fetch_count = g_count_fetched ? g_ctx.d_fetch : nullptr; a.fetch_shards = kFetchShards; a.pad_ = 0;
            a.runs = hb.p + hb.cap; a.run_cap = run_cap_of(hb.cap);
            if ((uint64_t)a.tiles * a.n_batches > 0x7FFFFFFFull)
                return fail(PM_ERANGE, "launch grid too large (%u tiles x %u batches)", a.tiles, a.n_batches);
            if (a.wide_query && g_wq_split != 1) {
                // few, very long queries: even with a whole workgroup per query the launch is a few hundred
                // workgroups.  nsplit workgroups then share the steps of each tile (gridDim.z); partial count planes
                // meet in global slabs (k_scan, "wide-query form across workgroups").
                const uint64_t blocks = (uint64_t)(g.g > 0 ? a.n_batches * a.tiles : a.total_blocks) * g.slabs;
                uint32_t max_nt = 0;
                for (uint32_t i = b; i < e; ++i) max_nt = std::max(max_nt, q->n_terms[q->qmap[i]]);
                int min_g = 64;
                for (size_t u : g.members) min_g = std::min(min_g, units[u].ix->g);
                const uint32_t ngrp_max = std::min<uint32_t>(a.wq_groups, 256u / (uint32_t)min_g);
                const uint32_t trips = ((max_nt + 7u) / 8u + ngrp_max - 1u) / std::max<uint32_t>(ngrp_max, 1u);
                uint32_t ns = g_wq_split ? g_wq_split : (uint32_t)std::min<uint64_t>(64, (2048 + blocks - 1) / std::max<uint64_t>(blocks, 1));
                if (!g_wq_split) ns = std::min<uint32_t>(ns, std::max<uint32_t>(1u, trips / 64u));     // >= 64 steps per workgroup
                ns = std::min<uint32_t>(ns, std::max<uint32_t>(trips, 1u));
                if (ns > 1 && blocks * ns <= 0x7FFFFFFFull && ns <= 65535u) {
                    const uint32_t ngrp_min = 4;                  // slabs are sized for the smallest groups-per-query count (256 / 4 lanes)
                    const size_t need = (size_t)blocks * ns * (size_t)kPlaneClass[c] * (256u / ngrp_min) * sizeof(uint4);
                    if (ws->split_bytes < need) {
                        HIPCHK(hipStreamSynchronize(st));         // an earlier launch of this search may still use the old slabs
                        if (ws->d_split) (void)hipFree(ws->d_split);
                        ws->d_split = nullptr; ws->split_bytes = 0;
                        HIPCHK(hipMalloc((void**)&ws->d_split, need));
                        ws->split_bytes = need;
                    }
                    if (ws->split_cnt_n < blocks) {
                        HIPCHK(hipStreamSynchronize(st));
                        if (ws->d_split_cnt) (void)hipFree(ws->d_split_cnt);
                        ws->d_split_cnt = nullptr; ws->split_cnt_n = 0;
                        HIPCHK(hipMalloc((void**)&ws->d_split_cnt, (size_t)blocks * sizeof(uint32_t)));
                        ws->split_cnt_n = (size_t)blocks;
                    }
                    HIPCHK(hipMemsetAsync(ws->d_split_cnt, 0, (size_t)blocks * sizeof(uint32_t), st));
                    a.nsplit = ns; a.split_slabs = ws->d_split; a.split_cnt = ws->d_split_cnt;
                }
            }
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
            // k-mers x row bytes of what the launch covers (a unit with a query part counts its queries only)
            const uint64_t* tp = q->term_prefix.data();
            uint64_t alg_l = 0;
            for (size_t u : g.members) {
                uint32_t first, end;
                part_range(units[u].part, e - b, &first, &end);
                alg_l += (tp[b + end] - tp[b + first]) * g.nh * units[u].ix->info.row_bytes;
            }
            pm_launch_t L{};
            L.lanes_per_row = (uint32_t)g.g; L.planes = (uint32_t)kPlaneClass[c]; L.num_hashes = g.nh;
            L.wide_query = a.wide_query;
            L.n_batches = a.n_batches; L.n_queries = e - b;
            L.algorithmic_bytes = alg_l;
            r->launches.push_back(L);
            alg += L.algorithmic_bytes;
        }
        desc_off += g.members.size();
    }
    HIPCHK(hipEventRecord(r->ev2, st));
    if (!q->last_use) HIPCHK(hipEventCreateWithFlags(&q->last_use, hipEventDisableTiming));
    HIPCHK(hipEventRecord(q->last_use, st));          // nothing queued so far reads the query set's HBM copies after this
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

static int result_wait_impl(pm_result_t* r);
extern "C" int pm_result_wait(pm_result_t* r) try {
    if (!r) return fail(PM_EINVAL, "bad argument");
    if (!r->pending) return r->failed ? fail(r->failed, "this search failed earlier") : PM_OK;
    NEED_DEV();
    const int rc = result_wait_impl(r);
    if (rc && r->pending) {                     // a HIP call failed half way: the result is dead, not "still pending"
        (void)hipStreamSynchronize(g_ctx.stream);
        r->pending = false;
        r->failed = rc;
        result_release(r);
    }
    return rc;
} PM_GUARD_END
static int result_wait_impl(pm_result_t* r) {
    for (;;) {
        HIPCHK(hipEventSynchronize(r->ws->done));
        const unsigned long long cnt = r->ws->h_cnt[0], runs = r->ws->h_cnt[1];
        if (cnt <= r->cap) {
            r->n_records = cnt; r->n_runs = runs;
            { std::lock_guard<std::mutex> lk(g_pool_mu); if (cnt > g_ctx.hit_hint) g_ctx.hit_hint = cnt; }
            break;
        }
        // hit buffer too small: grow to the exact count and run the job again
        if (r->attempt >= 1) { r->pending = false; r->failed = PM_EHIP; result_release(r); return fail(PM_EHIP, "hit count changed between runs"); }
        r->attempt++;
        (void)hipFree(r->d_hits); r->d_hits = nullptr; r->cap = 0;
        int rc = enqueue_search(r, cnt);
        if (rc) { (void)hipStreamSynchronize(g_ctx.stream); r->pending = false; r->failed = rc; result_release(r); return rc; }
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
                               double threshold, uint32_t nb_best_hits, uint32_t slot_base, pm_result_t** out) try {
    return pm_search_async_parts(idx, n_idx, q, threshold, nb_best_hits, slot_base, nullptr, out);
} PM_GUARD_END

extern "C" int pm_search_async_parts(pm_index_t* const* idx, size_t n_idx, pm_queries_t* q,
                                     double threshold, uint32_t nb_best_hits, uint32_t slot_base,
                                     const pm_qpart_t* parts, pm_result_t** out) try {
    NEED_DEV();
    if (!idx || !q || !out || n_idx == 0) return fail(PM_EINVAL, "bad argument");
    if (parts)
        for (size_t s = 0; s < n_idx; ++s)
            if (parts[s].den && (parts[s].lo > parts[s].hi || parts[s].hi > parts[s].den))
                return fail(PM_EINVAL, "index %zu: query part [%u, %u) of %u", s, parts[s].lo, parts[s].hi, parts[s].den);
    if (!(threshold >= 0.0)) return fail(PM_EINVAL, "threshold must be >= 0");
    for (size_t s = 0; s < n_idx; ++s) {
        if (!idx[s]) return fail(PM_EINVAL, "index %zu is null", s);
        if (idx[s]->info.term_size != q->k)
            return fail(PM_EINVAL, "index %zu has term_size %u but the queries were parsed for %u", s, idx[s]->info.term_size, q->k);
    }
    pm_result* r = new pm_result();
    r->idx.assign(idx, idx + n_idx);
    r->q = q; r->threshold = threshold; r->nb_best = nb_best_hits; r->slot_base = slot_base;
    if (parts) r->parts.assign(parts, parts + n_idx);
    r->tie_desc = g_tie_desc;
    uint64_t hint;
    { std::lock_guard<std::mutex> lk(g_pool_mu); hint = g_ctx.hit_hint; }
    // first guess of the record count: 48 per query (a read that matches its species' batch brings ~100 records after the
    // n-best cut; most reads match nothing elsewhere), at most 1/32 of the free HBM; a search that needs more is queued
    // again with the exact size (pm_result_wait) and later searches start from what was seen
    size_t fr = 0, tot = 0;
    (void)hipMemGetInfo(&fr, &tot);
    const uint64_t per_query = std::min<uint64_t>((uint64_t)q->headers.size() * 48, (uint64_t)fr / 32 / sizeof(uint4) / 2);
    const uint64_t want_cap = std::max<uint64_t>(std::max<uint64_t>(1u << 20, std::max<uint64_t>(per_query, (uint64_t)q->headers.size() * 16)), hint + hint / 4);
    int rc = enqueue_search(r, want_cap);
    if (rc) {
        // whatever was queued before the failure must not outlive its buffers
        (void)hipStreamSynchronize(g_ctx.stream);
        result_release(r);
        delete r;
        return rc;
    }
    *out = r;
    return PM_OK;
} PM_GUARD_END

extern "C" int pm_search(pm_index_t* const* idx, size_t n_idx, pm_queries_t* q,
                         double threshold, uint32_t nb_best_hits, uint32_t slot_base, pm_result_t** out) try {
    pm_result_t* r = nullptr;
    int rc = pm_search_async(idx, n_idx, q, threshold, nb_best_hits, slot_base, &r);
    if (rc) return rc;
    rc = pm_result_wait(r);
    if (rc) {                                   // nothing queued for this result may outlive its buffers or keep its workspace
        (void)hipStreamSynchronize(g_ctx.stream);
        r->pending = false;
        result_release(r);
        delete r;
        return rc;
    }
    *out = r;
    return PM_OK;
} PM_GUARD_END

#define RESULT_READY(r)                                            \
    do {                                                           \
        if ((r)->pending || (r)->failed) {                         \
            int rc_w_ = pm_result_wait(const_cast<pm_result_t*>(r)); \
            if (rc_w_) return rc_w_;                               \
        }                                                          \
    } while (0)

extern "C" int pm_result_stats(const pm_result_t* r, pm_stats_t* st) try {
    if (!r || !st) return fail(PM_EINVAL, "bad argument");
    RESULT_READY(r);
    *st = r->st;
    return PM_OK;
} PM_GUARD_END
extern "C" int pm_result_launches(const pm_result_t* r, pm_launch_t* out, size_t cap, size_t* n) try {
    if (!r || !n) return fail(PM_EINVAL, "bad argument");
    RESULT_READY(r);
    *n = r->launches.size();
    if (out) for (size_t i = 0; i < r->launches.size() && i < cap; ++i) out[i] = r->launches[i];
    return PM_OK;
} PM_GUARD_END
extern "C" int pm_result_hits_device(const pm_result_t* r, const void** dptr, uint64_t* n) try {
    if (!r || !dptr || !n) return fail(PM_EINVAL, "bad argument");
    RESULT_READY(r);
    *dptr = r->d_hits; *n = r->n_records;
    return PM_OK;
} PM_GUARD_END
static int ensure_ordered(pm_result* r);
extern "C" int pm_result_copy_hits_device(pm_result_t* r, void* dst, uint64_t capacity, int ordered) try {
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
} PM_GUARD_END

bool hit_less(const pm_hit_t& a, const pm_hit_t& b) {
    if (a.slot != b.slot) return a.slot < b.slot;
    if (a.query != b.query) return a.query < b.query;
    const bool am = a.doc == PM_DOC_COUNT, bm = b.doc == PM_DOC_COUNT;
    if (am != bm) return am;                              // count records lead their (slot, query) run
    if (a.score != b.score) return a.score > b.score;     // score descending
    return g_tie_desc ? a.doc > b.doc : a.doc < b.doc;    // then document index ascending ("cobs_tie_order" 1: descending)
}

// Orders records by (slot, query, score desc, doc asc).  Large inputs: stable
// LSD radix passes on the (slot, query) key, then a comparison sort inside each
// (slot, query) run (runs are short: the hits of one query in one batch).
// General form, for records in any order (gathered from elsewhere, written by a caller).
void order_hits(pm_hit_t* h, uint64_t n) {
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
// bytes, compact sub-indexes) are merged into one ordered list by k_merge_runs.
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
    std::vector<uint4> m_groups, m_runs;          // groups of several runs; .w of a group: 1 = the counting-sort merge applies
    r->slot_first.assign(r->idx.size() + 1, 0);
    size_t next_slot = 0;                          // slots up to here have their first record noted
    while (k < dir.size()) {
        size_t e = k + 1;
        while (e < dir.size() && dir[e].slot == dir[k].slot && dir[e].query == dir[k].query) ++e;
        for (const size_t s_rel = (size_t)(dir[k].slot - r->slot_base); next_slot <= s_rel && next_slot < r->idx.size(); ++next_slot)
            r->slot_first[next_slot] = o;
        if (e == k + 1) {
            const RunEnt& d = dir[k];
            const uint32_t len = d.len & 0x7FFFFFFFu;
            const bool cut = (d.len >> 31) != 0;             // cut on the GPU: the count record stays
            const uint32_t n = len + (cut ? 1u : 0u);
            h_plan[np++] = make_uint4(d.begin + (cut ? 0u : 1u), (uint32_t)o, n, 0u);
            o += n;
        } else {
            // several runs of one (slot, query): merged into one ordered list by k_merge_runs (count records dropped:
            // such groups are never cut on the GPU)
            // a score is at most the query's k-mer count: short queries with few runs take the O(records + runs) merge
            const bool hist = r->q->n_terms[dir[k].query] <= merge_hist_max_score() && e - k <= merge_hist_max_runs();
            m_groups.push_back(make_uint4((uint32_t)o, (uint32_t)(e - k), (uint32_t)m_runs.size(), hist ? 1u : 0u));
            for (size_t j = k; j < e; ++j) {
                const uint32_t len = dir[j].len & 0x7FFFFFFFu;
                m_runs.push_back(make_uint4(dir[j].begin + 1u, len, 0u, 0u));
                o += len;
            }
        }
        k = e;
    }
    for (; next_slot <= r->idx.size(); ++next_slot) r->slot_first[next_slot] = o;
    r->n_out = o;
    { int rc = take_hit_buffer(std::max<uint64_t>(std::max<uint64_t>(o, 2 * (uint64_t)np), 1), &r->d_ord); if (rc) return done(rc); }
    // the plan travels in the (unused) directory part of the destination buffer
    uint4* d_plan = r->d_ord.p + r->d_ord.cap;
    if (np > run_cap_of(r->d_ord.cap)) return done(fail(PM_EHIP, "run directory larger than its bound"));
    OCHK(hipMemcpyAsync(d_plan, h_plan, np * sizeof(uint4), hipMemcpyHostToDevice, st));
    OCHK(launch_permute_runs(d_plan, (uint32_t)np, r->d_hits, r->d_ord.p, st));
    OCHK(hipStreamSynchronize(st));
    // groups written as several runs (rows wider than 1024 bytes, compact sub-indexes) are merged on the device
    if (!m_groups.empty()) {
        // the groups of the counting-sort merge first, the others behind them
        std::stable_partition(m_groups.begin(), m_groups.end(), [](const uint4& g) { return g.w != 0u; });
        size_t n_hist = 0;
        while (n_hist < m_groups.size() && m_groups[n_hist].w) ++n_hist;
        if (g_merge_hist == 0) n_hist = 0;
        HitBuf mp{nullptr, 0};
        { int rc = take_hit_buffer(m_groups.size() + m_runs.size(), &mp); if (rc) return done(rc); }
        hipError_t e = hipMemcpyAsync(mp.p, m_groups.data(), m_groups.size() * sizeof(uint4), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(mp.p + m_groups.size(), m_runs.data(), m_runs.size() * sizeof(uint4), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = launch_merge_runs_hist(mp.p, (uint32_t)n_hist, mp.p + m_groups.size(), r->d_hits, r->d_ord.p, r->tie_desc, st);
        if (e == hipSuccess) e = launch_merge_runs(mp.p + n_hist, (uint32_t)(m_groups.size() - n_hist), mp.p + m_groups.size(), r->d_hits, r->d_ord.p, r->tie_desc, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        give_hit_buffer(mp);
        if (e != hipSuccess) return done(fail(PM_EHIP, "merging multi-run groups: %s", hipGetErrorString(e)));
    }
#undef OCHK
    r->ordered = true;
    return done(PM_OK);
}

extern "C" int pm_result_ordered_device(pm_result_t* r, const void** dptr, uint64_t* n) try {
    NEED_DEV();
    if (!r || !dptr || !n) return fail(PM_EINVAL, "bad argument");
    RESULT_READY(r);
    { int rc = ensure_ordered(r); if (rc) return rc; }
    *dptr = r->d_ord.p; *n = r->n_out;
    return PM_OK;
} PM_GUARD_END

extern "C" int pm_result_hits_into(const pm_result_t* r_, pm_hit_t* out, uint64_t capacity, uint64_t* n_out) try {
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
} PM_GUARD_END

extern "C" int pm_result_hits_host(pm_result_t* r, const pm_hit_t** hits, uint64_t* n) try {
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
} PM_GUARD_END
// The ordered records of ONE index of the search (slot = its position in the idx array), read back on their own into a
// pooled pinned buffer: the host half of a stage works batch by batch, and pinning memory for a whole search's records
// (hundreds of MB at a million reads) costs more than copying them.
struct pm_slice { PinBuf buf; };
extern "C" int pm_result_slot_hits(pm_result_t* r, uint32_t slot, const pm_hit_t** hits, uint64_t* n, pm_slice_t** slice) try {
    NEED_DEV();
    if (!r || !hits || !n || !slice) return fail(PM_EINVAL, "bad argument");
    RESULT_READY(r);
    if (slot >= r->idx.size()) return fail(PM_EINVAL, "slot %u of a search over %zu indexes", slot, r->idx.size());
    { int rc = ensure_ordered(r); if (rc) return rc; }
    uint64_t first = 0, count = 0;
    if (r->n_out) { first = r->slot_first[slot]; count = r->slot_first[slot + 1] - first; }
    pm_slice* sl = new pm_slice{PinBuf{nullptr, 0}};
    if (count) {
        { int rc = take_pinned((size_t)count * sizeof(pm_hit_t), &sl->buf); if (rc) { delete sl; return rc; } }
        hipError_t e = hipMemcpyAsync(sl->buf.p, r->d_ord.p + first, count * sizeof(pm_hit_t), hipMemcpyDeviceToHost, g_ctx.d2h_stream);
        if (e == hipSuccess) e = hipStreamSynchronize(g_ctx.d2h_stream);
        if (e != hipSuccess) { give_pinned(sl->buf); delete sl; return fail(PM_EHIP, "reading back slot %u: %s", slot, hipGetErrorString(e)); }
    }
    *hits = (const pm_hit_t*)sl->buf.p; *n = count; *slice = sl;
    return PM_OK;
} PM_GUARD_END
extern "C" void pm_slice_free(pm_slice_t* sl) {
    if (!sl) return;
    give_pinned(sl->buf);
    delete sl;
}
extern "C" void pm_result_free(pm_result_t* r) {
    if (!r) return;
    bind_thread_quiet();
    if (r->pending && r->ws) (void)hipEventSynchronize(r->ws->done);      // the GPU may still write into the buffers
    result_release(r);
    delete r;
}


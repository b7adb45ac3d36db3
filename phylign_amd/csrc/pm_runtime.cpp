// pm_runtime.cpp -- process-wide state of libphylign_match.so: error text, the GPU binding
// (pm_init / per-thread hipSetDevice), streams, option switches, pm_threshold_terms.
#include "pm_host.h"

static thread_local std::string g_err;
int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_err = buf;
    return code;
}

Ctx g_ctx;
int bind_thread() {
    if (!g_ctx.ready) return fail(PM_ENODEV, "pm_init() has not succeeded: no GPU bound (there is no CPU fallback)");
    hipError_t e = hipSetDevice(g_ctx.device);
    if (e != hipSuccess) return fail(PM_EHIP, "hipSetDevice(%d): %s", g_ctx.device, hipGetErrorString(e));
    return PM_OK;
}
uint32_t g_threshold_bound = 1;
uint32_t g_count_fetched = 0;
uint32_t g_single_launch = 0;
uint32_t g_wide_query = 0;
uint32_t g_wq_split = 0;
uint32_t g_threshold_rule = 0;
uint32_t g_tie_desc = 0;
uint32_t g_merge_hist = 1;

// ------------------------------------------------------------- worker pool
namespace {
struct Job {
    const std::function<void(size_t)>* fn; size_t n; std::atomic<size_t> next{0}, done{0};
    std::mutex err_mu; std::exception_ptr err;            // the first exception an item threw (rethrown by parallel_for)
    void run(size_t i) noexcept {
        try { (*fn)(i); }
        catch (...) { std::lock_guard<std::mutex> lk(err_mu); if (!err) err = std::current_exception(); }
    }
};
struct Pool {
    std::mutex mu;
    std::condition_variable cv, cv_done;
    std::vector<std::thread> threads;
    std::vector<Job*> jobs;                   // jobs with unclaimed items
    bool stop = false;
    void worker() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv.wait(lk, [&] { return stop || !jobs.empty(); });
            if (stop) return;
            Job* j = jobs.front();                          // picked and claimed under the lock: the job is alive
            const size_t n = j->n;
            const size_t i = j->next.fetch_add(1);
            if (i + 1 >= n) jobs.erase(jobs.begin());       // nothing left to claim (or this was the last item)
            if (i >= n) continue;
            lk.unlock();
            j->run(i);                                      // never throws: a pool thread has nobody to throw to
            lk.lock();
            // the owner may destroy the job as soon as it sees done == n: it checks under this lock, and `j` is not touched after the add
            if (j->done.fetch_add(1) + 1 == n) cv_done.notify_all();
        }
    }
    void ensure(size_t want) {
        std::lock_guard<std::mutex> lk(mu);
        while (threads.size() < want) threads.emplace_back([this] { worker(); });
    }
    ~Pool() {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv.notify_all();
        for (auto& t : threads) t.join();
    }
};
Pool& pool() { static Pool* p = new Pool(); return *p; }      // never destroyed: worker threads may outlive static destructors
}  // namespace

size_t parallel_width() {
    const size_t hw = std::max<size_t>(1, std::thread::hardware_concurrency());
    return std::min<size_t>(hw, 16);
}
void parallel_for(size_t n, const std::function<void(size_t)>& fn) {
    if (n == 0) return;
    if (n == 1) { fn(0); return; }                 // (throws straight to the caller)
    Pool& p = pool();
    p.ensure(parallel_width() - 1);
    Job job; job.fn = &fn; job.n = n;
    {
        std::lock_guard<std::mutex> lk(p.mu);
        p.jobs.push_back(&job);
    }
    p.cv.notify_all();
    for (;;) {                                 // the caller works too
        const size_t i = job.next.fetch_add(1);
        if (i >= n) break;
        job.run(i);
        job.done.fetch_add(1);
    }
    {
        std::unique_lock<std::mutex> lk(p.mu);
        for (auto it = p.jobs.begin(); it != p.jobs.end(); ++it) if (*it == &job) { p.jobs.erase(it); break; }   // all items are claimed
        p.cv_done.wait(lk, [&] { return job.done.load() >= n; });
    }
    if (job.err) std::rethrow_exception(job.err);
}

int on_exception() {
    try { throw; }
    catch (const std::bad_alloc&) { return fail(PM_ENOMEM, "out of host memory"); }
    catch (const std::length_error& e) { return fail(PM_ENOMEM, "out of host memory (%s)", e.what()); }
    catch (const std::exception& e) { return fail(PM_EINVAL, "internal error: %s", e.what()); }
    catch (...) { return fail(PM_EINVAL, "internal error: unknown exception"); }
}

// ------------------------------------------------------------------ runtime
extern "C" const char* pm_last_error(void) { return g_err.c_str(); }

extern "C" int pm_init(int device) try {
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
} PM_GUARD_END

extern "C" void pm_shutdown(void) {
    if (!g_ctx.ready) return;
    bind_thread_quiet();
    (void)hipDeviceSynchronize();
    (void)hipStreamDestroy(g_ctx.stream);
    (void)hipStreamDestroy(g_ctx.copy_stream);
    (void)hipStreamDestroy(g_ctx.d2h_stream);
    for (Workspace* w : g_ctx.ws) {
        if (w->d_cnt) (void)hipFree(w->d_cnt);
        if (w->h_cnt) (void)hipHostFree(w->h_cnt);
        if (w->d_desc) (void)hipFree(w->d_desc);
        if (w->h_desc) (void)hipHostFree(w->h_desc);
        if (w->d_split) (void)hipFree(w->d_split);
        if (w->d_split_cnt) (void)hipFree(w->d_split_cnt);
        for (auto e : w->events) (void)hipEventDestroy(e);
        if (w->done) (void)hipEventDestroy(w->done);
        delete w;
    }
    release_stage_pool();
    release_text_pool();
    release_query_pool();
    release_hit_pool();
    if (g_ctx.d_fetch) (void)hipFree(g_ctx.d_fetch);
    for (auto& b : g_ctx.free_pinned) (void)hipHostFree(b.p);
    g_ctx = Ctx();
}

hipError_t device_malloc_reclaim(void** out, size_t bytes) {
    hipError_t e = hipMalloc(out, bytes);
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();                 // the failed attempt must not be what a later hipGetLastError() reports
        release_query_pool();
        release_hit_pool();
        e = hipMalloc(out, bytes);
        if (e != hipSuccess) (void)hipGetLastError();
    }
    return e;
}

extern "C" int pm_device_info(char* name, size_t cap, uint64_t* hbm_total, uint64_t* hbm_free, int* n_cus) try {
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
} PM_GUARD_END

extern "C" void pm_free(void* p) { free(p); }

extern "C" int pm_set_option(const char* name, int64_t value) try {
    if (!name) return fail(PM_EINVAL, "bad argument");
    if (strcmp(name, "threshold_bound") == 0) { g_threshold_bound = value ? 1u : 0u; return PM_OK; }
    if (strcmp(name, "count_fetched") == 0) { g_count_fetched = value ? 1u : 0u; return PM_OK; }
    if (strcmp(name, "single_launch") == 0) { g_single_launch = value ? 1u : 0u; return PM_OK; }
    if (strcmp(name, "wide_query") == 0) {
        if (value < 0 || value > 2) return fail(PM_EINVAL, "wide_query takes 0 (auto), 1 (always) or 2 (never)");
        g_wide_query = (uint32_t)value;
        return PM_OK;
    }
    if (strcmp(name, "wide_query_split") == 0) {
        if (value < 0 || value > 256) return fail(PM_EINVAL, "wide_query_split takes 0 (auto), 1 (off) or a count up to 256");
        g_wq_split = (uint32_t)value;
        return PM_OK;
    }
    // the two rules of `cobs query` that nothing in the reference repository pins: switchable, so
    // that the day a cobs 0.2.1 binary says otherwise (tools/pin_against_cobs.sh) no code changes
    if (strcmp(name, "cobs_threshold_rule") == 0) {
        if (value < 0 || value > 2) return fail(PM_EINVAL, "cobs_threshold_rule takes 0 (ceil), 1 (floor) or 2 (round half up)");
        if ((uint32_t)value != g_threshold_rule && g_live_results.load() > 0)
            return fail(PM_EINVAL, "cobs_threshold_rule cannot change while %d search result(s) are alive (free them first)", g_live_results.load());
        g_threshold_rule = (uint32_t)value;
        return PM_OK;
    }
    if (strcmp(name, "merge_counting_sort") == 0) { g_merge_hist = value ? 1u : 0u; return PM_OK; }
    if (strcmp(name, "release_query_pool") == 0) {              // an action, not a setting: see the header
        if (g_ctx.ready) { bind_thread_quiet(); release_query_pool(); }
        return PM_OK;
    }
    if (strcmp(name, "release_pools") == 0) {
        if (g_ctx.ready) { bind_thread_quiet(); release_query_pool(); release_hit_pool(); }
        return PM_OK;
    }
    if (strcmp(name, "cobs_tie_order") == 0) {
        if (value < 0 || value > 1) return fail(PM_EINVAL, "cobs_tie_order takes 0 (equal scores by ascending document) or 1 (descending)");
        if ((uint32_t)value != g_tie_desc && g_live_results.load() > 0)
            return fail(PM_EINVAL, "cobs_tie_order cannot change while %d search result(s) are alive (free them first)", g_live_results.load());
        g_tie_desc = (uint32_t)value;
        return PM_OK;
    }
    return fail(PM_EINVAL, "unknown option '%s'", name);
} PM_GUARD_END

// The ONE place that turns `-t` into a minimum score (cobs counts_to_result):
// ceil(threshold * num_terms) in IEEE double.  config.yaml:20 -> Snakefile:410.
// pm_set_option("cobs_threshold_rule"): 1 = truncation, 2 = round half up -- the alternatives a real cobs may turn out to use.
extern "C" uint32_t pm_threshold_terms(double threshold, uint64_t num_terms) {
    const double x = threshold * (double)num_terms;
    double t = g_threshold_rule == 1 ? std::floor(x) : (g_threshold_rule == 2 ? std::floor(x + 0.5) : std::ceil(x));
    if (!(t > 0)) return 0;
    if (t > 4294967295.0) return 4294967295u;
    return (uint32_t)t;
}


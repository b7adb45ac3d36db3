// pm_bench_aids.hip -- libphylign_bench.so: measurement and test aids for the MI355X matching stage
// (include/phylign_match_bench.h).  Synthetic 661k-shaped signatures generated in HBM, planted true
// positives, "home batch" clusters, and the random-row gather probe that gives the memory system's
// ceiling for k_scan's access pattern.  None of this is on the matching path; it reaches the product
// library only through its C ABI (pm_index_create / pm_index_matrix_device / pm_hash_terms ...).
#include <hip/hip_runtime.h>
#include <errno.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>

#include "../../../include/phylign_match_bench.h"

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

thread_local char t_err[512];
int bfail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(t_err, sizeof t_err, fmt, ap);
    va_end(ap);
    return code;
}
#define BHIP(expr)                                                                              \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return bfail(e_ == hipErrorOutOfMemory ? PM_ENOMEM : PM_EHIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)
// a product call failed: its message becomes ours
#define BPM(expr)                                                               \
    do {                                                                        \
        int rc_ = (expr);                                                       \
        if (rc_) return bfail(rc_, "%s", pm_last_error());                      \
    } while (0)

// exact h mod S with m = floor(2^64 / S) (the product's row mapping: pm_kernels.hip mod_sig)
__device__ __forceinline__ uint64_t mod_sig(uint64_t h, uint64_t S, uint64_t m) {
    const uint64_t qh = __umul64hi(h, m);
    uint64_t r = h - qh * S;
    if (r >= S) r -= S;
    if (r >= S) r -= S;
    return m ? r : 0ull;
}
uint64_t barrett_m(uint64_t S) {
    if (S < 2) return 0;
    uint64_t m = ~0ull / S;
    if ((~0ull % S) + 1 == S) m += 1;
    return m;
}

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}
// Synthetic 661k-shaped matrix: dword j of row r = lo32(u) & hi32(u),
// u = splitmix64(splitmix64(splitmix64(seed ^ batch*C) + r) + j): P(bit)=1/4.
__global__ __launch_bounds__(256) void k_synth(
    uint8_t* __restrict__ dst, uint64_t stride, uint64_t n_rows, uint32_t n_docs, uint64_t kb)
{
    const uint64_t chunks_per_row = stride >> 4;
    const uint64_t total = n_rows * chunks_per_row;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = i / chunks_per_row, ch = i - r * chunks_per_row;
        const uint64_t kr = splitmix64(kb + r);
        uint32_t w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint64_t j = ch * 4 + k;
            const uint64_t first_doc = j * 32;
            uint32_t v = 0;
            if (first_doc < n_docs) {
                const uint64_t u = splitmix64(kr + j);
                v = (uint32_t)u & (uint32_t)(u >> 32);
                if (first_doc + 32 > n_docs) v &= (1u << (n_docs - first_doc)) - 1u;
            }
            w[k] = v;
        }
        *reinterpret_cast<uint4*>(dst + r * stride + ch * 16) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}
static hipError_t launch_synth(uint8_t* dst, uint64_t stride, uint64_t n_rows, uint32_t n_docs,
                        uint64_t seed, uint32_t batch, hipStream_t st) {
    if (n_rows == 0) return hipSuccess;
    // kb = splitmix64(seed ^ batch * C) computed on the host side of the launcher
    uint64_t x = seed ^ ((uint64_t)batch * 0xD1B54A32D192ED03ULL);
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    const uint64_t kb = x ^ (x >> 31);
    const uint64_t total = n_rows * (stride >> 4);
    uint64_t blocks = (total + 255) / 256;
    if (blocks > 262144) blocks = 262144;
    hipLaunchKernelGGL(k_synth, dim3((uint32_t)blocks), dim3(256), 0, st, dst, stride, n_rows, n_docs, kb);
    return hipGetLastError();
}

// Ceiling probe: the same access pattern as k_scan (random rows, 16 B per lane,
// G lanes per row, 8 gathers in flight per lane) with the counting replaced by
// one XOR per load.  Used only to measure what the memory system delivers for
// this pattern (profiles/r03/NOTES.md section 6); not part of the matching path.
template <int G, int U>
__global__ __launch_bounds__(256) void k_probe_gather(const uint8_t* __restrict__ matrix, uint64_t stride,
                                                       uint64_t n_rows, uint64_t lookups_per_group, uint32_t* sink,
                                                       int mode)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t group = ((uint64_t)blockIdx.x * 256 + threadIdx.x) / G;
    const uint32_t c = lane % G;
    u32x4 acc = (u32x4)(0u);
    uint64_t state = splitmix64(group * 0x9E3779B97F4A7C15ULL + 1);
    const bool active = (uint64_t)c * 16 < stride;
    // mode 1: ascending stratified rows; mode 2: ascending order statistics of uniform rows
    // (what a query sees when its k-mers are visited in row order) -- locality experiments
    float total = 0.f, run = 0.f;
    if (mode == 2) {
        uint64_t st2 = state;
        for (uint64_t i = 0; i <= lookups_per_group; ++i) {
            st2 = st2 * 6364136223846793005ULL + 1442695040888963407ULL;
            total += -__logf(((float)(uint32_t)(st2 >> 40) + 1.f) * (1.f / 16777217.f));
        }
    }
    for (uint64_t i = 0; i < lookups_per_group; i += U) {
        u32x4 v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            state = state * 6364136223846793005ULL + 1442695040888963407ULL;
            uint64_t r = __umul64hi(state, n_rows);
            if (mode == 1) r = (uint64_t)(((double)(i + k) + (double)(state >> 11) * (1.0 / 9007199254740992.0)) / (double)lookups_per_group * (double)n_rows);
            if (mode == 2) {
                run += -__logf(((float)(uint32_t)(state >> 40) + 1.f) * (1.f / 16777217.f));
                r = (uint64_t)((double)(run / total) * (double)(n_rows - 1));
            }
            if (r >= n_rows) r = n_rows - 1;
            v[k] = (u32x4)(0u);
            if (active) v[k] = *reinterpret_cast<const u32x4*>(matrix + r * stride + (uint64_t)c * 16);
        }
#pragma unroll
        for (int k = 0; k < U; ++k) acc ^= v[k];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[0] = 1;   // keeps the loads alive
}
// Cache-policy flavours of the same gather (PM_PROBE_FLAVOR = 1 ... 5): does a gather that needs 16-64 bytes of a
// 128-byte line cost less on the fabric when it bypasses / streams through the caches?  F: 1 = nt, 2 = sc1,
// 3 = sc0 sc1, 4 = sc0 sc1 nt, 5 = sc0.  Inline asm: the waits are explicit (the compiler does not track these loads).
#define PM_PROBE_LOAD(F, dst, ptr)                                                                                    \
    do {                                                                                                              \
        if constexpr (F == 1) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(dst) : "v"(ptr) : "memory");          \
        else if constexpr (F == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(dst) : "v"(ptr) : "memory");    \
        else if constexpr (F == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(dst) : "v"(ptr) : "memory"); \
        else if constexpr (F == 4) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt" : "=v"(dst) : "v"(ptr) : "memory"); \
        else asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=v"(dst) : "v"(ptr) : "memory");                    \
    } while (0)
template <int G, int F>
__global__ __launch_bounds__(256) void k_probe_flavor(const uint8_t* __restrict__ matrix, uint64_t stride,
                                                       uint64_t n_rows, uint64_t lookups_per_group, uint32_t* sink)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t group = ((uint64_t)blockIdx.x * 256 + threadIdx.x) / G;
    const uint32_t c = lane % G;
    u32x4 acc = (u32x4)(0u);
    uint64_t state = splitmix64(group * 0x9E3779B97F4A7C15ULL + 1);
    const uint64_t coff = ((uint64_t)c * 16 < stride) ? (uint64_t)c * 16 : 0;     // every lane loads (asm loads are unconditional)
    for (uint64_t i = 0; i < lookups_per_group; i += 8) {
        u32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            state = state * 6364136223846793005ULL + 1442695040888963407ULL;
            uint64_t r = __umul64hi(state, n_rows);
            const uint8_t* p = matrix + r * stride + coff;
            PM_PROBE_LOAD(F, v[k], p);
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) :: "memory");
#pragma unroll
        for (int k = 0; k < 8; ++k) acc ^= v[k];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[0] = 1;
}
template <int G>
static void probe_launch_flavor(int flavor, dim3 grid, hipStream_t st, const uint8_t* matrix, uint64_t stride,
                                uint64_t n_rows, uint64_t per, uint32_t* sink) {
    switch (flavor) {
        case 1: hipLaunchKernelGGL((k_probe_flavor<G, 1>), grid, dim3(256), 0, st, matrix, stride, n_rows, per, sink); break;
        case 2: hipLaunchKernelGGL((k_probe_flavor<G, 2>), grid, dim3(256), 0, st, matrix, stride, n_rows, per, sink); break;
        case 3: hipLaunchKernelGGL((k_probe_flavor<G, 3>), grid, dim3(256), 0, st, matrix, stride, n_rows, per, sink); break;
        case 4: hipLaunchKernelGGL((k_probe_flavor<G, 4>), grid, dim3(256), 0, st, matrix, stride, n_rows, per, sink); break;
        default: hipLaunchKernelGGL((k_probe_flavor<G, 5>), grid, dim3(256), 0, st, matrix, stride, n_rows, per, sink); break;
    }
}
template <int G>
static void probe_launch_u(int mode, int flavor, int unroll, dim3 grid, hipStream_t st, const uint8_t* matrix, uint64_t stride,
                           uint64_t n_rows, uint64_t per, uint32_t* sink) {
    if (flavor > 0) { probe_launch_flavor<G>(flavor, grid, st, matrix, stride, n_rows, per, sink); return; }
    if (unroll == 4)       hipLaunchKernelGGL((k_probe_gather<G, 4>), grid, dim3(256), 0, st, matrix, stride, n_rows, per, sink, mode);
    else if (unroll == 16) hipLaunchKernelGGL((k_probe_gather<G, 16>), grid, dim3(256), 0, st, matrix, stride, n_rows, per, sink, mode);
    else                   hipLaunchKernelGGL((k_probe_gather<G, 8>), grid, dim3(256), 0, st, matrix, stride, n_rows, per, sink, mode);
}
static hipError_t launch_probe_gather(const uint8_t* matrix, uint64_t stride, uint64_t n_rows, int g, uint64_t groups,
                                      uint64_t lookups_per_group, int mode, int flavor, int unroll, uint32_t* sink, hipStream_t st) {
    const uint64_t threads = groups * (uint64_t)g;
    dim3 grid((uint32_t)((threads + 255) / 256));
    switch (g) {
        case 1:  probe_launch_u<1>(mode, flavor, unroll, grid, st, matrix, stride, n_rows, lookups_per_group, sink); break;
        case 2:  probe_launch_u<2>(mode, flavor, unroll, grid, st, matrix, stride, n_rows, lookups_per_group, sink); break;
        case 4:  probe_launch_u<4>(mode, flavor, unroll, grid, st, matrix, stride, n_rows, lookups_per_group, sink); break;
        case 8:  probe_launch_u<8>(mode, flavor, unroll, grid, st, matrix, stride, n_rows, lookups_per_group, sink); break;
        case 16: probe_launch_u<16>(mode, flavor, unroll, grid, st, matrix, stride, n_rows, lookups_per_group, sink); break;
        case 32: probe_launch_u<32>(mode, flavor, unroll, grid, st, matrix, stride, n_rows, lookups_per_group, sink); break;
        case 64: probe_launch_u<64>(mode, flavor, unroll, grid, st, matrix, stride, n_rows, lookups_per_group, sink); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// Clustered true positives for the "phylogenetically related batch" workload: for every
// selected query (its HOME batch is this index) the documents are taken in clusters of 32
// (one dword of the row); cluster c gets a match fraction phi(query, c) from
// {none x8, 0.60, 0.70, 0.75, 0.85, 0.93, 0.97, 1.0, 1.0} (half of the clusters unrelated) and
// every k-mer row of the query gets, in that dword, an OR-mask of independent bits of density phi.
// So many documents end up near the 0.7 threshold, above and below it.  One thread per
// (selected query, k-mer, dword).  Set-up only, never timed.
__global__ __launch_bounds__(256) void k_plant_cluster(
    uint8_t* matrix, uint64_t stride, uint64_t S, uint64_t bm, uint32_t n_docs,
    const uint64_t* __restrict__ hashes, const uint64_t* __restrict__ term_off, uint32_t nh,
    uint32_t q_first, uint32_t q_step, uint32_t n_sel, uint32_t max_terms, uint64_t seed)
{
    const uint32_t n_dw = (n_docs + 31u) >> 5;
    const uint64_t total = (uint64_t)n_sel * max_terms * n_dw;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t dw = (uint32_t)(i % n_dw);
        const uint64_t r1 = i / n_dw;
        const uint32_t t = (uint32_t)(r1 % max_terms);
        const uint32_t qi = q_first + (uint32_t)(r1 / max_terms) * q_step;
        const uint64_t t_first = term_off[qi];
        if (t >= (uint32_t)(term_off[qi + 1] - t_first)) continue;
        const uint64_t kc = splitmix64(seed ^ ((uint64_t)qi * 0x9E3779B97F4A7C15ULL) ^ ((uint64_t)dw << 40));
        const uint32_t sel = (uint32_t)(kc & 15u);
        if (sel < 8u) continue;                                    // unrelated cluster
        const uint32_t lut[8] = {154u, 179u, 192u, 218u, 238u, 248u, 256u, 256u};   // phi * 256
        const uint32_t cut = lut[sel - 8u];
        uint32_t m = 0;
        uint64_t x = splitmix64(kc + 0x51ED270B1ULL * (t + 1u));
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            x = splitmix64(x + (uint64_t)w);
#pragma unroll
            for (int b = 0; b < 8; ++b) m |= (((uint32_t)(x >> (8 * b)) & 255u) < cut ? 1u : 0u) << (8 * w + b);
        }
        const uint32_t first = dw * 32u;
        if (first + 32u > n_docs) m &= (1u << (n_docs - first)) - 1u;
        if (m == 0u) continue;
        for (uint32_t j = 0; j < nh; ++j) {
            const uint64_t h = hashes[(t_first + t) * nh + j];                   // dense: [term][hash]
            uint32_t* w = reinterpret_cast<uint32_t*>(matrix + mod_sig(h, S, bm) * stride + (uint64_t)dw * 4);
            atomicOr(w, m);
        }
    }
}
static hipError_t launch_plant_cluster(uint8_t* matrix, uint64_t stride, uint64_t S, uint32_t n_docs,
                                const uint64_t* hashes, const uint64_t* term_off, uint32_t nh,
                                uint32_t q_first, uint32_t q_step, uint32_t n_sel, uint32_t max_terms,
                                uint64_t seed, hipStream_t st) {
    const uint64_t total = (uint64_t)n_sel * max_terms * ((n_docs + 31u) >> 5);
    if (total == 0) return hipSuccess;
    uint64_t blocks = (total + 255) / 256;
    if (blocks > 262144) blocks = 262144;
    hipLaunchKernelGGL(k_plant_cluster, dim3((uint32_t)blocks), dim3(256), 0, st, matrix, stride, S, barrett_m(S),
                       n_docs, hashes, term_off, nh, q_first, q_step, n_sel, max_terms, seed);
    return hipGetLastError();
}


__global__ void k_plant(uint8_t* matrix, uint64_t stride, const uint64_t* rows, const uint32_t* docs, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t d = docs[i];
    uint32_t* w = reinterpret_cast<uint32_t*>(matrix + rows[i] * stride + (uint64_t)(d >> 5) * 4);
    atomicOr(w, 1u << (d & 31));
}
static hipError_t launch_plant(uint8_t* matrix, uint64_t stride, const uint64_t* rows, const uint32_t* docs,
                        uint64_t n, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_plant, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, matrix, stride, rows, docs, n);
    return hipGetLastError();
}

uint64_t host_splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}

// the device that holds the index's matrix becomes the calling thread's current device
int bind_to(const pm_index_t* ix, uint8_t** matrix, uint64_t* stride, pm_index_info_t* info) {
    void* p = nullptr;
    BPM(pm_index_matrix_device(ix, &p, stride));
    BPM(pm_index_info(ix, info));
    int dev = -1;
    BPM(pm_index_device(ix, &dev));
    if (dev < 0) return bfail(PM_EINVAL, "index has no resident matrix");
    BHIP(hipSetDevice(dev));
    *matrix = (uint8_t*)p;
    return PM_OK;
}

}  // namespace

struct pm_bench_hashes {
    int device = -1;
    uint64_t n_queries = 0, n_terms = 0;
    uint32_t nh = 0;
    int canon = 0;
    uint64_t* d_hashes = nullptr;       // [term][hash]
    uint64_t* d_term_off = nullptr;     // n_queries + 1
    std::vector<uint32_t> terms;        // k-mers per query
};

extern "C" const char* pm_bench_last_error(void) { return t_err; }

extern "C" int pm_bench_index_synth(uint32_t batch_id, uint32_t n_docs, uint64_t signature_size,
                                    uint32_t num_hashes, uint32_t term_size, uint64_t seed,
                                    int layout, int header_only, pm_index_t** out) {
    if (!out || n_docs == 0 || signature_size == 0 || num_hashes == 0 || term_size == 0)
        return bfail(PM_EINVAL, "bad synthetic index shape");
    const uint64_t kb = host_splitmix64(seed ^ ((uint64_t)batch_id * 0xD1B54A32D192ED03ULL));
    std::string names;
    names.reserve((size_t)n_docs * 24);
    char nm[64];
    for (uint32_t d = 0; d < n_docs; ++d) {
        const int l = snprintf(nm, sizeof nm, "%05x_SYN%03uD%07u\n",
                               (unsigned)(host_splitmix64(kb ^ (0xA5A5A5A5ull + d)) & 0xFFFFF), batch_id, d);
        names.append(nm, (size_t)l);
    }
    pm_index_t* ix = nullptr;
    BPM(pm_index_create(term_size, 1, signature_size, num_hashes, names.data(), names.size(), n_docs, layout, header_only, &ix));
    if (!header_only) {
        uint8_t* m = nullptr; uint64_t stride = 0; pm_index_info_t info;
        int rc = bind_to(ix, &m, &stride, &info);
        hipError_t e = hipSuccess;
        if (!rc) {
            e = launch_synth(m, stride, signature_size, n_docs, seed, batch_id, nullptr);
            if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
            if (e != hipSuccess) rc = bfail(PM_EHIP, "synthetic generator: %s", hipGetErrorString(e));
        }
        if (rc) { pm_index_free(ix); return rc; }
    }
    *out = ix;
    return PM_OK;
}

extern "C" int pm_bench_index_plant(pm_index_t* ix, const uint64_t* rows, const uint32_t* docs, size_t n) {
    if (!ix || (n && (!rows || !docs))) return bfail(PM_EINVAL, "bad argument");
    uint8_t* m = nullptr; uint64_t stride = 0; pm_index_info_t info;
    { int rc = bind_to(ix, &m, &stride, &info); if (rc) return rc; }
    if (n == 0) return PM_OK;
    for (size_t i = 0; i < n; ++i)
        if (rows[i] >= info.signature_size || docs[i] >= info.n_docs) return bfail(PM_EINVAL, "plant %zu out of range", i);
    uint64_t* dr = nullptr; uint32_t* dd = nullptr;
    BHIP(hipMalloc((void**)&dr, n * 8));
    hipError_t e = hipMalloc((void**)&dd, n * 4);
    if (e == hipSuccess) e = hipMemcpy(dr, rows, n * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dd, docs, n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = launch_plant(m, stride, dr, dd, n, nullptr);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    (void)hipFree(dr); if (dd) (void)hipFree(dd);
    if (e != hipSuccess) return bfail(PM_EHIP, "plant: %s", hipGetErrorString(e));
    return PM_OK;
}

extern "C" int pm_bench_hashes_create(pm_queries_t* q, int canonicalize, uint32_t num_hashes, pm_bench_hashes_t** out) {
    if (!q || !out || num_hashes == 0) return bfail(PM_EINVAL, "bad argument");
    pm_bench_hashes* h = new pm_bench_hashes();
    h->nh = num_hashes; h->canon = canonicalize ? 1 : 0;
    int rc = pm_queries_count(q, &h->n_queries, &h->n_terms);
    std::vector<uint64_t> off, hs;
    if (!rc) {
        off.assign((size_t)h->n_queries + 1, 0);
        h->terms.resize((size_t)h->n_queries);
        for (uint64_t i = 0; i < h->n_queries && !rc; ++i) {
            uint64_t t = 0;
            rc = pm_queries_terms(q, i, &t);
            h->terms[(size_t)i] = (uint32_t)t;
            off[(size_t)i + 1] = off[(size_t)i] + t;
        }
    }
    if (!rc) { hs.resize((size_t)h->n_terms * num_hashes); rc = pm_hash_terms(q, h->canon, num_hashes, hs.data()); }
    if (rc) { delete h; return bfail(rc, "%s", pm_last_error()); }
    // pm_hash_terms left the library's device current for this thread
    hipError_t e = hipGetDevice(&h->device);
    if (e == hipSuccess) e = hipMalloc((void**)&h->d_hashes, std::max<size_t>(hs.size(), 1) * 8);
    if (e == hipSuccess) e = hipMalloc((void**)&h->d_term_off, off.size() * 8);
    if (e == hipSuccess && !hs.empty()) e = hipMemcpy(h->d_hashes, hs.data(), hs.size() * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(h->d_term_off, off.data(), off.size() * 8, hipMemcpyHostToDevice);
    if (e != hipSuccess) { pm_bench_hashes_free(h); return bfail(PM_EHIP, "uploading the hashes: %s", hipGetErrorString(e)); }
    *out = h;
    return PM_OK;
}
extern "C" void pm_bench_hashes_free(pm_bench_hashes_t* h) {
    if (!h) return;
    if (h->device >= 0) (void)hipSetDevice(h->device);
    if (h->d_hashes) (void)hipFree(h->d_hashes);
    if (h->d_term_off) (void)hipFree(h->d_term_off);
    delete h;
}

extern "C" int pm_bench_index_plant_cluster(pm_index_t* ix, const pm_bench_hashes_t* h, uint32_t q_first, uint32_t q_step, uint64_t seed) {
    if (!ix || !h || q_step == 0) return bfail(PM_EINVAL, "bad argument");
    uint8_t* m = nullptr; uint64_t stride = 0; pm_index_info_t info;
    { int rc = bind_to(ix, &m, &stride, &info); if (rc) return rc; }
    if (info.num_hashes != h->nh || (int)info.canonicalize != h->canon)
        return bfail(PM_EINVAL, "the hashes were made for canonicalize %d, %u hash functions; the index has %u, %u",
                     h->canon, h->nh, info.canonicalize, info.num_hashes);
    if (q_first >= h->n_queries) return PM_OK;
    const uint32_t n_sel = (uint32_t)((h->n_queries - q_first + q_step - 1) / q_step);
    uint32_t max_terms = 0;
    for (uint64_t i = q_first; i < h->n_queries; i += q_step) max_terms = std::max(max_terms, h->terms[(size_t)i]);
    BHIP(launch_plant_cluster(m, stride, info.signature_size, info.n_docs, h->d_hashes, h->d_term_off, h->nh,
                              q_first, q_step, n_sel, max_terms, seed, nullptr));
    BHIP(hipStreamSynchronize(nullptr));
    return PM_OK;
}

extern "C" int pm_bench_probe_gather(const pm_index_t* ix, uint64_t n_groups, uint64_t lookups_per_group,
                                     int mode, int flavor, int unroll, double* ms, uint64_t* bytes) {
    if (!ix || !ms || !bytes || n_groups == 0) return bfail(PM_EINVAL, "bad argument");
    uint8_t* m = nullptr; uint64_t stride = 0; pm_index_info_t info;
    { int rc = bind_to(ix, &m, &stride, &info); if (rc) return rc; }
    if (stride > 1024) return bfail(PM_EINVAL, "probe supports rows up to 1024 bytes");
    uint64_t lanes = (stride + 15) / 16, g = 1;                     // lanes per row as k_scan uses them (power of two)
    while (g < lanes) g <<= 1;
    lookups_per_group = (lookups_per_group + 15) / 16 * 16;
    uint32_t* sink = nullptr;
    BHIP(hipMalloc((void**)&sink, 4));
    hipEvent_t e0, e1;
    BHIP(hipEventCreate(&e0)); BHIP(hipEventCreate(&e1));
    hipError_t e = hipEventRecord(e0, nullptr);
    if (e == hipSuccess) e = launch_probe_gather(m, stride, info.signature_size, (int)g, n_groups, lookups_per_group,
                                                 mode, flavor, unroll, sink, nullptr);
    if (e == hipSuccess) e = hipEventRecord(e1, nullptr);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    float f = 0;
    if (e == hipSuccess) e = hipEventElapsedTime(&f, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(sink);
    if (e != hipSuccess) return bfail(PM_EHIP, "probe: %s", hipGetErrorString(e));
    *ms = f; *bytes = n_groups * lookups_per_group * info.row_bytes;
    return PM_OK;
}

// Compressible content for the COLD-PATH timing only.  Bernoulli(1/4) bits barely compress (xz keeps 87 % of them), while the
// real 661k indexes shrink about tenfold because neighbouring documents of a phylogenetic batch share most k-mers.  This
// overwrites a resident matrix with rows in which document d repeats document d - 1's bit except with probability 2^-flip_log2
// (runs of equal bits along a row); the search results on such a matrix mean nothing and nothing is searched on it.
__global__ __launch_bounds__(256) void k_correlate(uint8_t* matrix, uint64_t stride, uint64_t S, uint32_t n_docs, uint64_t seed, uint32_t flip_log2) {
    const uint32_t n_dw = (n_docs + 31u) >> 5;
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < S; r += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t* row = reinterpret_cast<uint32_t*>(matrix + r * stride);
        uint32_t carry = (uint32_t)(splitmix64(seed ^ (r * 0x9E3779B97F4A7C15ULL)) & 1u);
        for (uint32_t w = 0; w < n_dw; ++w) {
            uint64_t a = splitmix64(seed + r * 0xD1B54A32D192ED03ULL + w);
            uint32_t x = 0xFFFFFFFFu;                                                             // flips, density 2^-flip_log2
            for (uint32_t i = 0; i < flip_log2; i += 2) {
                x &= (uint32_t)a;
                if (i + 1 < flip_log2) x &= (uint32_t)(a >> 32);
                a = splitmix64(a ^ 0xA5A5A5A5A5A5A5A5ULL);
            }
            x ^= x << 1; x ^= x << 2; x ^= x << 4; x ^= x << 8; x ^= x << 16;                     // prefix xor: runs
            if (carry) x = ~x;
            carry = x >> 31;
            const uint32_t first = w * 32u;
            if (first + 32u > n_docs) x &= (1u << (n_docs - first)) - 1u;
            row[w] = x;
        }
    }
}
extern "C" int pm_bench_index_correlate(pm_index_t* ix, uint64_t seed, uint32_t flip_log2) {
    if (!ix || flip_log2 == 0 || flip_log2 > 16) return bfail(PM_EINVAL, "bad argument");
    uint8_t* m = nullptr; uint64_t stride = 0; pm_index_info_t info;
    { int rc = bind_to(ix, &m, &stride, &info); if (rc) return rc; }
    hipLaunchKernelGGL(k_correlate, dim3((uint32_t)std::min<uint64_t>((info.signature_size + 255) / 256, 262144)), dim3(256), 0, nullptr,
                       m, stride, info.signature_size, info.n_docs, seed, flip_log2);
    BHIP(hipGetLastError());
    BHIP(hipStreamSynchronize(nullptr));
    return PM_OK;
}

// SURVEY.md 8d "many-queries regime": the DISTINCT signature rows a query set touches in this index.  A bitmap of S
// bits is marked with one atomicOr per (k-mer, hash function) and counted; untimed, reporting only.
__global__ __launch_bounds__(256) void k_mark_rows(uint32_t* __restrict__ bitmap, const uint64_t* __restrict__ hashes,
                                                   uint64_t n, uint64_t S, uint64_t bm) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = mod_sig(hashes[i], S, bm);
        atomicOr(bitmap + (r >> 5), 1u << (r & 31));
    }
}
__global__ __launch_bounds__(256) void k_count_bits(const uint32_t* __restrict__ bitmap, uint64_t n_words,
                                                    unsigned long long* __restrict__ out) {
    unsigned long long c = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (uint64_t)gridDim.x * blockDim.x)
        c += (unsigned)__popc(bitmap[i]);
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}

extern "C" int pm_bench_unique_rows(const pm_index_t* ix, const pm_bench_hashes_t* h, uint64_t* unique_rows) {
    if (!ix || !h || !unique_rows) return bfail(PM_EINVAL, "bad argument");
    uint8_t* m = nullptr; uint64_t stride = 0; pm_index_info_t info;
    { int rc = bind_to(ix, &m, &stride, &info); if (rc) return rc; }
    if (info.num_hashes != h->nh || (int)info.canonicalize != h->canon)
        return bfail(PM_EINVAL, "the hashes were made for canonicalize %d, %u hash functions; the index has %u, %u",
                     h->canon, h->nh, info.canonicalize, info.num_hashes);
    const uint64_t S = info.signature_size, n_words = (S + 31) / 32, n = h->n_terms * h->nh;
    uint32_t* bitmap = nullptr; unsigned long long* d_out = nullptr;
    BHIP(hipMalloc((void**)&bitmap, n_words * 4 + 16));        // + the 8-byte counter, 8-byte aligned behind the bitmap
    d_out = reinterpret_cast<unsigned long long*>(bitmap + ((n_words + 1) & ~1ull));
    hipError_t e = hipMemsetAsync(bitmap, 0, n_words * 4 + 16, nullptr);
    if (e == hipSuccess && n) {
        hipLaunchKernelGGL(k_mark_rows, dim3((uint32_t)std::min<uint64_t>((n + 255) / 256, 65536)), dim3(256), 0, nullptr,
                           bitmap, h->d_hashes, n, S, barrett_m(S));
        hipLaunchKernelGGL(k_count_bits, dim3((uint32_t)std::min<uint64_t>((n_words + 255) / 256, 65536)), dim3(256), 0, nullptr,
                           bitmap, n_words, d_out);
        e = hipGetLastError();
    }
    unsigned long long got = 0;
    if (e == hipSuccess) e = hipMemcpy(&got, d_out, 8, hipMemcpyDeviceToHost);
    (void)hipFree(bitmap);
    if (e != hipSuccess) return bfail(PM_EHIP, "unique rows: %s", hipGetErrorString(e));
    *unique_rows = got;
    return PM_OK;
}

// rows of `stride` bytes -> packed rows of row_bytes (the file's layout), one byte per thread step
__global__ __launch_bounds__(256) void k_unstride(const uint8_t* __restrict__ src, uint64_t stride, uint64_t row_bytes,
                                                  uint64_t total, uint8_t* __restrict__ dst) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (uint64_t)gridDim.x * 256) {
        const uint64_t r = i / row_bytes;
        dst[i] = src[r * stride + (i - r * row_bytes)];
    }
}

// A resident classic index written back as a .cobs_classic file (header in the first of the two field orders the
// product's reader accepts): lets a measurement put 661k-SHAPED index files -- and their .xz -- on
// disk for the cold / cached / resident timings of the stage (tools/e2e_cold_warm.py).
extern "C" int pm_bench_index_save(const pm_index_t* ix, const char* path) {
    if (!ix || !path) return bfail(PM_EINVAL, "bad argument");
    pm_index_info_t in;
    BPM(pm_index_info(ix, &in));
    if (!in.has_matrix || in.n_parts) return bfail(PM_EINVAL, "only a resident classic index can be saved");
    FILE* f = fopen(path, "wb");
    if (!f) return bfail(PM_EIO, "cannot create '%s': %s", path, strerror(errno));
    bool ok = fwrite("COBS:CLASSIC_INDEX", 1, 18, f) == 18;
    const uint32_t ver = 1; const uint8_t canon = (uint8_t)in.canonicalize; const uint64_t nh = in.num_hashes;
    ok = ok && fwrite(&ver, 4, 1, f) == 1 && fwrite(&in.term_size, 4, 1, f) == 1 && fwrite(&canon, 1, 1, f) == 1 &&
         fwrite(&in.n_docs, 4, 1, f) == 1 && fwrite(&in.signature_size, 8, 1, f) == 1 && fwrite(&nh, 8, 1, f) == 1;
    for (uint32_t d = 0; d < in.n_docs && ok; ++d) {
        size_t l = 0;
        const char* nm = pm_index_doc_name(ix, d, &l);
        ok = nm && fwrite(nm, 1, l, f) == l && fputc('\n', f) != EOF;
    }
    ok = ok && fwrite("CLASSIC_INDEX", 1, 13, f) == 13;
    // rows leave HBM packed (file layout) through one pinned buffer: a 2-D copy of 13 ... 500-byte rows into pageable
    // memory ran at 0.09 GB/s (33 GB of index files took 6 minutes)
    void* dptr = nullptr; uint64_t stride = 0;
    BPM(pm_index_matrix_device(ix, &dptr, &stride));
    const uint64_t chunk = std::max<uint64_t>(1, (64ull << 20) / in.row_bytes);
    const size_t cbytes = (size_t)(std::min<uint64_t>(chunk, in.signature_size) * in.row_bytes);
    uint8_t* dpack = nullptr; uint8_t* hbuf = nullptr;
    int rc = PM_OK;
    if (hipMalloc((void**)&dpack, cbytes) != hipSuccess || hipHostMalloc((void**)&hbuf, cbytes) != hipSuccess) {
        if (dpack) (void)hipFree(dpack);
        fclose(f); (void)remove(path);
        return bfail(PM_ENOMEM, "staging buffers for '%s'", path);
    }
    for (uint64_t r = 0; r < in.signature_size && ok && !rc; r += chunk) {
        const uint64_t n = std::min<uint64_t>(chunk, in.signature_size - r);
        const uint64_t total = n * in.row_bytes;
        k_unstride<<<(unsigned)std::min<uint64_t>((total + 255) / 256, 1u << 20), 256>>>(
            (const uint8_t*)dptr + r * stride, stride, in.row_bytes, total, dpack);
        if (hipMemcpy(hbuf, dpack, (size_t)total, hipMemcpyDeviceToHost) != hipSuccess) { rc = PM_EHIP; break; }
        ok = fwrite(hbuf, 1, (size_t)total, f) == (size_t)total;
    }
    (void)hipFree(dpack); (void)hipHostFree(hbuf);
    if (rc) { fclose(f); (void)remove(path); return bfail(rc, "device read-back for '%s' failed", path); }
    if (fclose(f) != 0) ok = false;
    if (!ok) { (void)remove(path); return bfail(PM_EIO, "writing '%s' failed", path); }
    return PM_OK;
}

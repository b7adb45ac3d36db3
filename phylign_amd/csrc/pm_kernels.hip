// pm_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the COBS matching stage.
//
// Replaces the hot loop of the external `cobs query` binary the reference calls
// at scripts/run_cobs_streaming.sh:24-29 / Snakefile:419-424 (SURVEY.md 8a rows
// a5-a7).  Written for 64-wide wavefronts; no MFMA on purpose: the path is a
// random row gather plus bit-sliced counting, bounded by HBM.
//
//   k_hash_terms  a5  canonicalise each k-mer + XXH64(seed j)
//   k_scan        a5  row = hash % signature_size (Barrett, exact), fused
//                 a6  gather rows (16 B per lane, G lanes per row), AND over
//                     hash functions, carry-save bit-sliced per-document counts
//                 a7  bit-sliced ">= threshold", ballot/mbcnt compaction of hits
//                 one launch covers every resident batch of one row-width class
//   k_restride    a4  index residency: file rows -> padded rows in HBM
// (the synthetic-index generator, the planting kernels and the gather probes are measurement aids and
// live in csrc/bench/pm_bench_aids.hip -> libphylign_bench.so, not in this library)
#include "pm_internal.h"

namespace pm {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------ hashing
#define XP1 0x9E3779B185EBCA87ULL
#define XP2 0xC2B2AE3D27D4EB4FULL
#define XP3 0x165667B19E3779F9ULL
#define XP4 0x85EBCA77C2B2AE63ULL
#define XP5 0x27D4EB2F165667C5ULL

__device__ __forceinline__ uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
__device__ __forceinline__ uint64_t xround(uint64_t acc, uint64_t in) {
    acc += in * XP2; acc = rotl64(acc, 31); return acc * XP1;
}
__device__ __forceinline__ uint64_t xmerge(uint64_t acc, uint64_t v) {
    v = xround(0, v); acc ^= v; return acc * XP1 + XP4;
}
// A<->T, C<->G on upper-case ASCII: A/T have bit1 clear (xor 0x15), C/G set (xor 0x04)
__device__ __forceinline__ uint32_t comp_base(uint32_t c) { return c ^ ((c & 2u) ? 0x04u : 0x15u); }

// Byte i of the canonical form of the k-mer at p (rc: reverse complement chosen).
struct KmerView {
    const uint8_t* p; uint32_t k; bool rc;
    __device__ __forceinline__ uint64_t byte(uint32_t i) const {
        const uint32_t c = p[rc ? k - 1 - i : i];       // one load, no divergent branch
        return (uint64_t)(rc ? comp_base(c) : c);
    }
    __device__ __forceinline__ uint64_t le64(uint32_t i) const {
        uint64_t v = 0;
#pragma unroll
        for (int b = 0; b < 8; ++b) v |= byte(i + b) << (8 * b);
        return v;
    }
};

__device__ uint64_t xxh64_kmer(const KmerView& kv, uint64_t seed) {
    const uint32_t len = kv.k;
    uint32_t i = 0;
    uint64_t h;
    if (len >= 32) {
        uint64_t v1 = seed + XP1 + XP2, v2 = seed + XP2, v3 = seed, v4 = seed - XP1;
        do {
            v1 = xround(v1, kv.le64(i));      v2 = xround(v2, kv.le64(i + 8));
            v3 = xround(v3, kv.le64(i + 16)); v4 = xround(v4, kv.le64(i + 24));
            i += 32;
        } while (i + 32 <= len);
        h = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
        h = xmerge(h, v1); h = xmerge(h, v2); h = xmerge(h, v3); h = xmerge(h, v4);
    } else {
        h = seed + XP5;
    }
    h += (uint64_t)len;
    while (i + 8 <= len) { h ^= xround(0, kv.le64(i)); h = rotl64(h, 27) * XP1 + XP4; i += 8; }
    if (i + 4 <= len) {
        uint64_t w = kv.byte(i) | (kv.byte(i + 1) << 8) | (kv.byte(i + 2) << 16) | (kv.byte(i + 3) << 24);
        h ^= w * XP1; h = rotl64(h, 23) * XP2 + XP3; i += 4;
    }
    while (i < len) { h ^= kv.byte(i) * XP5; h = rotl64(h, 11) * XP1; ++i; }
    h ^= h >> 33; h *= XP2; h ^= h >> 29; h *= XP3; h ^= h >> 32;
    return h;
}

// ---- k <= 32: the whole k-mer lives in four 64-bit registers ---------------
// bytewise complement of 8 packed bases (same rule as comp_base)
__device__ __forceinline__ uint64_t comp64(uint64_t w) {
    const uint64_t m = (w >> 1) & 0x0101010101010101ULL;
    return w ^ 0x1515151515151515ULL ^ (m | (m << 4));
}
__device__ __forceinline__ uint64_t low_bytes_mask(int nbytes) {   // nbytes in [0, 8]
    return nbytes >= 8 ? ~0ULL : ((1ULL << (8 * (nbytes < 0 ? 0 : nbytes))) - 1ULL);
}
// XXH64 of the first k (<= 32) bytes held little-endian in w[0..3]
__device__ __forceinline__ uint64_t xxh64_words(const uint64_t w[4], uint32_t k, uint64_t seed) {
    uint64_t h;
    uint32_t i = 0;
    if (k == 32) {
        uint64_t v1 = seed + XP1 + XP2, v2 = seed + XP2, v3 = seed, v4 = seed - XP1;
        v1 = xround(v1, w[0]); v2 = xround(v2, w[1]); v3 = xround(v3, w[2]); v4 = xround(v4, w[3]);
        h = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
        h = xmerge(h, v1); h = xmerge(h, v2); h = xmerge(h, v3); h = xmerge(h, v4);
        h += 32;
        i = 32;
    } else {
        h = seed + XP5 + (uint64_t)k;
#pragma unroll
        for (int n = 0; n < 3; ++n)
            if (i + 8 <= k) { h ^= xround(0, w[n]); h = rotl64(h, 27) * XP1 + XP4; i += 8; }
    }
    uint64_t rem = (i == 0) ? w[0] : (i == 8) ? w[1] : (i == 16) ? w[2] : (i == 24) ? w[3] : 0ULL;
    uint32_t left = k - i;
    if (left >= 4) { h ^= (rem & 0xFFFFFFFFULL) * XP1; h = rotl64(h, 23) * XP2 + XP3; rem >>= 32; left -= 4; }
    for (; left; --left) { h ^= (rem & 0xFFULL) * XP5; h = rotl64(h, 11) * XP1; rem >>= 8; }
    h ^= h >> 33; h *= XP2; h ^= h >> 29; h *= XP3; h ^= h >> 32;
    return h;
}

// One thread per padded term slot.  hashes layout: [8-slot block][hash j][8].
// The packed sequence buffer is padded by 64 bytes so aligned 8-byte reads
// around a k-mer never leave the allocation.
// KC > 0 fixes the k-mer length at compile time (31 for the 661k indexes).
template <int KC>
__global__ __launch_bounds__(256) void k_hash_terms(
    const uint8_t* __restrict__ seq, const QDesc* __restrict__ qd,
    const uint32_t* __restrict__ blk_query, uint64_t n_slots, uint32_t k_rt, int canon,
    uint32_t nh, uint64_t* __restrict__ hashes)
{
    const uint32_t k = KC > 0 ? (uint32_t)KC : k_rt;
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_slots) return;
    const uint64_t blk = s >> 3;
    const QDesc d = qd[blk_query[blk]];
    const uint32_t t = (uint32_t)(s - (uint64_t)d.pad_blk * 8);
    uint64_t* out = hashes + blk * nh * 8 + (s & 7);
    if (t >= d.n_terms) {
        for (uint32_t j = 0; j < nh; ++j) out[j * 8] = 0;
        return;
    }
    const uint64_t off = (((uint64_t)d.seq_hi << 32) | d.seq_lo) + t;
    if (k <= 32) {
        // five aligned 8-byte loads, funnel-shifted to the k-mer's first byte
        const uint64_t* ap = reinterpret_cast<const uint64_t*>(seq + (off & ~7ULL));
        const uint32_t sh8 = (uint32_t)(off & 7ULL) * 8;
        const uint64_t a0 = ap[0], a1 = ap[1], a2 = ap[2], a3 = ap[3], a4 = ap[4];
        uint64_t f[4];
        if (sh8) {
            f[0] = (a0 >> sh8) | (a1 << (64 - sh8)); f[1] = (a1 >> sh8) | (a2 << (64 - sh8));
            f[2] = (a2 >> sh8) | (a3 << (64 - sh8)); f[3] = (a3 >> sh8) | (a4 << (64 - sh8));
        } else { f[0] = a0; f[1] = a1; f[2] = a2; f[3] = a3; }
#pragma unroll
        for (int n = 0; n < 4; ++n) f[n] &= low_bytes_mask((int)k - 8 * n);
        uint64_t w[4] = {f[0], f[1], f[2], f[3]};
        if (canon) {
            // reverse complement: complement bytewise, reverse the 32-byte buffer,
            // then drop the 32-k leading (formerly trailing) bytes
            uint64_t r[6];
            r[0] = __builtin_bswap64(comp64(f[3])); r[1] = __builtin_bswap64(comp64(f[2]));
            r[2] = __builtin_bswap64(comp64(f[1])); r[3] = __builtin_bswap64(comp64(f[0]));
            r[4] = 0; r[5] = 0;
            const uint32_t drop = 32 - k;
            if (drop & 16) { r[0] = r[2]; r[1] = r[3]; r[2] = 0; r[3] = 0; }
            if (drop & 8)  { r[0] = r[1]; r[1] = r[2]; r[2] = r[3]; r[3] = 0; }
            const uint32_t b8 = (drop & 7) * 8;
            uint64_t rc[4];
            if (b8) {
                rc[0] = (r[0] >> b8) | (r[1] << (64 - b8)); rc[1] = (r[1] >> b8) | (r[2] << (64 - b8));
                rc[2] = (r[2] >> b8) | (r[3] << (64 - b8)); rc[3] = (r[3] >> b8);
            } else { rc[0] = r[0]; rc[1] = r[1]; rc[2] = r[2]; rc[3] = r[3]; }
#pragma unroll
            for (int n = 0; n < 4; ++n) rc[n] &= low_bytes_mask((int)k - 8 * n);
            // lexicographic order of the byte strings = order of the byte-swapped words
            bool use_rc = false, decided = false;
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const uint64_t a = __builtin_bswap64(f[n]), b = __builtin_bswap64(rc[n]);
                if (!decided && a != b) { use_rc = b < a; decided = true; }
            }
#pragma unroll
            for (int n = 0; n < 4; ++n) w[n] = use_rc ? rc[n] : f[n];
        }
        for (uint32_t j = 0; j < nh; ++j) out[j * 8] = xxh64_words(w, k, (uint64_t)j);
        return;
    }
    KmerView kv;
    kv.p = seq + off;
    kv.k = k; kv.rc = false;
    if (canon) {
        // lexicographic min(kmer, revcomp): first position where they differ decides
        for (uint32_t i = 0; i < k; ++i) {
            const uint32_t f = kv.p[i], r = comp_base(kv.p[k - 1 - i]);
            if (f != r) { kv.rc = r < f; break; }
        }
    }
    for (uint32_t j = 0; j < nh; ++j) out[j * 8] = xxh64_kmer(kv, (uint64_t)j);
}

// row = h % S, exact: Barrett with m = floor(2^64 / S) and at most two fix-ups
// (q' = mulhi(h, m) >= floor(h/S) - 1).  m == 0 encodes S == 1.
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
    if ((~0ull % S) + 1 == S) m += 1;   // 2^64 = (2^64 - 1) + 1
    return m;
}

// --------------------------------------------------------------------- scan
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t t = (uint32_t)__shfl_xor((int)v, o, 64);
        v = t > v ? t : v;
    }
    return v;
}

// One 16-byte chunk of a signature row.  Rows are touched once per (query, batch) and never again by
// this launch, so the gather is issued non-temporal (`global_load_dwordx4 ... nt`): measured with the
// gather probe (profiles/r03/narrow_gather_calibration.txt) 6.59 instead of 5.89 TB/s on 512-byte rows.
#ifndef PM_SCAN_NT
#define PM_SCAN_NT 1
#endif
__device__ __forceinline__ u32x4 ld_row(const uint8_t* p) {
#if PM_SCAN_NT
    return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
#else
    return *reinterpret_cast<const u32x4*>(p);
#endif
}

// carry-save adder on 128 document columns: (s, c) = a + b + cin bitwise
#define PM_CSA(s, c, a, b, cin)                         \
    do {                                                \
        const u32x4 u_ = (a) ^ (b);                     \
        const u32x4 a_ = (a);                           \
        const u32x4 ci_ = (cin);                        \
        (s) = u_ ^ ci_;                                 \
        (c) = (u_ & ci_) | (~u_ & a_);                  \
    } while (0)

// G lanes cooperate on one query (16 B = 128 documents per lane); a wave holds
// 64/G queries; P = number of counter bit planes (queries with < 2^P terms).
// G = 0 instantiates the mixed form: the lane-group width comes from the batch
// descriptor, so ONE launch covers narrow batches of different widths (their
// launches are short, so the per-launch drain tail would otherwise add up).
#ifndef PM_SCAN_TERMS
#define PM_SCAN_TERMS 8              // k-mers (row gathers in flight per lane) per step: 8 or 4
#endif
#ifndef PM_SCAN_SHARE_ROWS
#define PM_SCAN_SHARE_ROWS 1         // lanes of a group split the row-offset computation of a step (G >= 8, one hash)
#endif
#ifndef PM_SCAN_SHARE_MAX_P
#define PM_SCAN_SHARE_MAX_P 13       // widest counter class that shares (7: round 5's setting)
#endif
#ifndef PM_SCAN_PREFETCH_HASH
#define PM_SCAN_PREFETCH_HASH 0      // 1: sharing instantiations load their hash of step s + 1 during step s (measured: no gain)
#endif
#ifndef PM_SCAN_PREFETCH_MAX_P
#define PM_SCAN_PREFETCH_MAX_P 13
#endif
#ifndef PM_SCAN_MIN_WAVES
#define PM_SCAN_MIN_WAVES 4          // waves per SIMD the register allocator must leave room for
#endif
// register budget of the wide counter classes (queries of 1 024 ... 2^24 k-mers hold 64 / 96 plane
// registers): measured on 9.7 kbp / 100 kbp queries, see profiles/r03/NOTES.md section 6
#ifndef PM_SCAN_WAVES_P16
#define PM_SCAN_WAVES_P16 4
#endif
#ifndef PM_SCAN_WAVES_P16_WQ
#define PM_SCAN_WAVES_P16_WQ 3       // the wide-query form needs a few registers more: 92 B of scratch at 4 waves
#endif
#ifndef PM_SCAN_WAVES_P20
#define PM_SCAN_WAVES_P20 3
#endif
#ifndef PM_SCAN_WAVES_P20_WQ
#define PM_SCAN_WAVES_P20_WQ 3
#endif
#ifndef PM_SCAN_WAVES_P24
#define PM_SCAN_WAVES_P24 2
#endif
// WQ ("wide query") instantiations: a query set of few, long queries leaves most of the chip idle when
// one lane group walks a whole query (a 100 kbp plasmid = 12 500 serial steps).  Here NGRP =
// min(ScanArgs.wq_groups, 256 / G) lane groups of ONE workgroup share a query (wq_groups: a power of
// two >= 4 chosen by the host, 8 unless narrow rows need more groups to fill the chip): group `sub` takes the steps sub,
// sub + NGRP, ..., every group keeps partial bit-sliced counts, and after the loop they are added
// pairwise through LDS (a ripple-carry adder over the P planes per tree level).  The threshold bound
// is off in this form (partial counts say nothing about a document's total); group 0 of the
// query runs the usual epilogue on the combined planes.  Results are identical to the plain form.
template <int G, int P, bool NH1, bool WQ>
__global__ __launch_bounds__(256, (P <= 13 ? PM_SCAN_MIN_WAVES : (P <= 16 ? (WQ ? PM_SCAN_WAVES_P16_WQ : PM_SCAN_WAVES_P16) : (P <= 20 ? (WQ ? PM_SCAN_WAVES_P20_WQ : PM_SCAN_WAVES_P20) : PM_SCAN_WAVES_P24)))) void k_scan(const ScanArgs a)
{
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    uint32_t batch, tile;
    if constexpr (G > 0) {
        batch = blockIdx.x / a.tiles;
        tile = blockIdx.x - batch * a.tiles;
    } else {                                          // last batch whose first block is <= blockIdx.x
        uint32_t lo = 0, hi = a.n_batches;
        while (hi - lo > 1u) {
            const uint32_t mid = (lo + hi) >> 1;
            if (a.batches[mid].block_begin <= blockIdx.x) lo = mid; else hi = mid;
        }
        batch = lo;
        tile = blockIdx.x - a.batches[lo].block_begin;
    }
    const BatchDesc bd = a.batches[batch];            // uniform: scalar loads
    const uint32_t g = G > 0 ? (uint32_t)G : bd.lanes;               // power of two, 1..64
    const uint32_t gl = G > 0 ? (uint32_t)__builtin_ctz((unsigned)(G > 0 ? G : 1)) : (uint32_t)__builtin_ctz(bd.lanes);
    const uint32_t gpb = 256u >> gl;                                  // lane groups per workgroup
    const uint32_t gi = ((uint32_t)wave * 64u + (uint32_t)lane) >> gl; // my group inside the workgroup
    const uint32_t ngrp = WQ ? (gpb < a.wq_groups ? gpb : a.wq_groups) : 1u;   // groups that share one query (power of two)
    const uint32_t sub = WQ ? (gi & (ngrp - 1u)) : 0u;
    // a unit may be searched with a PART of the launch's queries only (the batch is resident on several ranks that
    // share its queries: pm_search_async_parts): positions q_first ... q_end - 1 of the class-ordered query list
    const uint32_t li = (WQ ? (tile * gpb + gi) / ngrp : tile * gpb + gi) + bd.q_first;
    const uint32_t c = (uint32_t)lane & (g - 1u);
    const uint32_t gfirst0 = (uint32_t)lane & ~(g - 1u);               // first lane of my group
    const uint32_t slab = blockIdx.y;
    const uint64_t boff = ((uint64_t)slab * g + c) * 16;   // byte offset of this lane's chunk
    const bool qv = li < (bd.q_end < a.nq ? bd.q_end : a.nq);

    uint32_t q = 0, nt = 0, thr = 0;
    uint64_t pb = 0;
    if (qv) {
        q = a.qmap[li];
        const QDesc d = a.qd[q];
        nt = d.n_terms; pb = d.pad_blk; thr = a.thr[q];
    }
    const uint64_t stride = bd.stride;
    const bool active = qv && boff < stride;
    const uint32_t nblk = active ? (nt + 7u) >> 3 : 0u;      // 8-slot hash blocks of this query
    const uint32_t wmax = wave_max_u32(nblk);
    const uint32_t nh = NH1 ? 1u : a.nh;

    const uint8_t* base = bd.matrix + boff;
    const u32x4* hp = reinterpret_cast<const u32x4*>(a.hashes + pb * nh * 8);
    const uint64_t S = bd.sig_size, bm = bd.barrett_m;

    u32x4 pl[P];
#pragma unroll
    for (int p = 0; p < P; ++p) pl[p] = (u32x4)(0u);

    // Threshold bound: a document whose count so far plus the k-mers still to come
    // cannot reach thr is out for good (count + remaining never grows), and the
    // hits and their scores do not depend on it.  Once every document behind a
    // 128-byte line is out, that line is not fetched any more; once the whole
    // wavefront is out, it stops.  (Lanes that share a line: W consecutive lanes.)
    const uint32_t W = g < 8u ? g : 8u;
    bool line_alive = active;
    uint32_t nfetch = 0;                              // 16-byte row chunks this lane gathered (measurement option)
    constexpr int TS = PM_SCAN_TERMS;                 // k-mers per step: 8 (or 4): loads in flight per lane
    constexpr uint32_t SPB = 8 / TS;                  // steps per 8-slot hash block
    const uint32_t trips = WQ ? (wmax * SPB + ngrp - 1u) / ngrp : wmax * SPB;
    // wide-query form across workgroups: gridDim.z workgroups share the steps of a tile's queries (few, very long
    // queries: a chromosome-sized contig is 125 000 steps); each takes a contiguous range, the last one to arrive adds up
    const uint32_t nsplit = WQ ? a.nsplit : 1u;
    const uint32_t seg = (trips + nsplit - 1u) / nsplit;
    const uint32_t it_begin = WQ ? blockIdx.z * seg : 0u;
    const uint32_t it_end = WQ ? (it_begin + seg < trips ? it_begin + seg : trips) : trips;
    // The lanes of a group all need the same TS rows.  With 8+ lanes per group and one hash function each
    // lane maps ONE k-mer (lane c takes k-mer c mod TS: one hash load, one Barrett reduction instead of TS
    // of each) and the group shares the ROW INDICES by ds_bpermute; every lane takes part in the exchange,
    // alive or not (a disabled source lane would deliver 0).  A row index fits 32 bits here: G >= 8 means
    // a stride of at least 80 bytes (128 in the aligned layout), and rows x stride is resident in HBM, so
    // rows < 309 GB / 80 B < 2^32 (pm_index.cpp refuses 2^32 rows of such a width outright, whatever the
    // device's memory).  One exchanged register per k-mer (the byte
    // offset is one v_mad_u64_u32 per lane): 8 registers instead of the 16 that round 5's exchange of 64-bit
    // offsets took, which is what lets the 10- and 13-plane classes (40 / 52 plane registers) share as well.
    constexpr bool SHARE = PM_SCAN_SHARE_ROWS && NH1 && G >= TS && TS == 8 && P <= PM_SCAN_SHARE_MAX_P;
    // Build option PM_SCAN_PREFETCH_HASH (off): my hash of a step is loaded ONE STEP AHEAD -- issued behind the row
    // gathers of the current step, so that a step's dependent chain is gather -> count instead of hash load ->
    // reduce -> gather -> count.  The block index is clamped to the query's own blocks: the load is unconditional
    // and always in bounds, and a value fetched for a k-mer that does not exist is never used (lane i of the group
    // reads rows[i] only when k-mer i of the step exists).  Measured (profiles/r06/ab_prefetch_hash.txt): within
    // +-0.3 % on the 7- and 10-plane classes, +1.2 % on the 13-plane class at the price of 1-5 spilled VGPRs and
    // -6 % on its wide-query form: with four waves per SIMD the other waves' gathers already cover the hash load,
    // and the launch is bound by the rate of 128-byte lines the memory system delivers.  Records are identical
    // either way (the parity suite ran on both builds).
    constexpr bool PREFETCH = SHARE && PM_SCAN_PREFETCH_HASH && P <= PM_SCAN_PREFETCH_MAX_P;
    const uint32_t i_mine = c & (uint32_t)(TS - 1);
    const uint32_t qblk = qv ? (nt + 7u) >> 3 : 0u;
    auto my_hash_of_step = [&](uint32_t step) -> uint64_t {
        const uint32_t blk = step / SPB;
        return a.hashes[(pb + (blk < qblk ? blk : (qblk ? qblk - 1u : 0u))) * 8 + i_mine];       // [blk][hash 0][8]
    };
    uint64_t h_next = 0;
    if constexpr (PREFETCH) h_next = my_hash_of_step(WQ ? it_begin * ngrp + sub : it_begin);
    for (uint32_t it = it_begin; it < it_end; ++it) {
        const uint32_t sidx = WQ ? it * ngrp + sub : it;          // my step of the query
        const uint32_t b = sidx / SPB, t0i = sidx * TS;           // block, first k-mer of this step
        if (!WQ && a.bound) {
            const int need = (int)thr - (int)(nt - t0i);          // score required now to still reach thr
            bool alive = line_alive;
            if (__any(alive && need > 0)) {
                const uint32_t K = (1u << P) - (uint32_t)(need > 0 ? need : 0);
                u32x4 cy = (u32x4)(0u);
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    const u32x4 both = pl[p] & cy, any = pl[p] | cy;
                    cy = ((K >> p) & 1u) ? any : both;
                }
                if (need > 0) alive = alive && ((cy.x | cy.y | cy.z | cy.w) != 0u) && (uint32_t)need <= nt;
            }
            const unsigned long long bal = __ballot(alive);
            line_alive = ((bal >> ((uint32_t)lane & ~(W - 1u))) & ((1ull << W) - 1ull)) != 0ull;
            if (!__any(line_alive && t0i < nt)) break;
        }
        u32x4 x[TS];
#pragma unroll
        for (int i = 0; i < TS; ++i) x[i] = (u32x4)(0u);
        uint32_t my_row = 0;
        if constexpr (PREFETCH) {
            my_row = (uint32_t)mod_sig(h_next, S, bm);
        } else if constexpr (SHARE) {
            if (qv && t0i + i_mine < nt) my_row = (uint32_t)mod_sig(a.hashes[(pb + b) * 8 + i_mine], S, bm);
        }
        const bool go = active && t0i < nt && line_alive;
        const uint32_t left = go ? nt - t0i : 0u;         // valid k-mers of this step for this lane (0: no gather)
        if (a.fetch_count) nfetch += (left < (uint32_t)TS ? left : (uint32_t)TS) * nh;
        uint32_t rows[SHARE ? TS : 1];
        if constexpr (SHARE) {
#pragma unroll
            for (int i = 0; i < TS; ++i) rows[i] = (uint32_t)__shfl((int)my_row, (int)(gfirst0 + (uint32_t)i), 64);
        }
        if (go) {
            if constexpr (SHARE) {
#pragma unroll
                for (int i = 0; i < TS; ++i)
                    if ((uint32_t)i < left) x[i] = ld_row(base + (uint64_t)rows[i] * stride);
            } else
            for (uint32_t j = 0; j < nh; ++j) {
                const u32x4* hj = hp + (size_t)(b * nh + j) * 4 + (sidx % SPB) * (TS / 2);
                uint64_t h[TS];
#pragma unroll
                for (int i = 0; i < TS / 2; ++i) {
                    const u32x4 hh = hj[i];
                    h[2 * i] = ((uint64_t)hh.y << 32) | hh.x;
                    h[2 * i + 1] = ((uint64_t)hh.w << 32) | hh.z;
                }
                u32x4 v[TS];
#pragma unroll
                for (int i = 0; i < TS; ++i) {
                    v[i] = (u32x4)(0u);
                    if ((uint32_t)i < left)
                        v[i] = ld_row(base + mod_sig(h[i], S, bm) * stride);
                }
#pragma unroll
                for (int i = 0; i < TS; ++i) x[i] = (j == 0) ? v[i] : (x[i] & v[i]);
            }
        }
        if constexpr (PREFETCH) h_next = my_hash_of_step(WQ ? (it + 1u) * ngrp + sub : it + 1u);   // lands under the gathers
        // TS one-bit inputs -> low planes by carry-save adders + one carry rippling upwards
        u32x4 t0, t1, f0;
        PM_CSA(pl[0], t0, pl[0], x[0], x[1]);
        PM_CSA(pl[0], t1, pl[0], x[2], x[3]);
        PM_CSA(pl[1], f0, pl[1], t0, t1);
        u32x4 carry = f0;
        int first = 2;
        if constexpr (TS == 8) {
            u32x4 t2, t3, f1, e0;
            PM_CSA(pl[0], t2, pl[0], x[4], x[5]);
            PM_CSA(pl[0], t3, pl[0], x[6], x[7]);
            PM_CSA(pl[1], f1, pl[1], t2, t3);
            PM_CSA(pl[2], e0, pl[2], f0, f1);
            carry = e0;
            first = 3;
        }
        // wide-query form: a group counts at most a 1/ngrp share of the query's k-mers (+ 8) with
        // ngrp >= 4, so its partial counts fit P - 1 planes; the top plane stays zero until the combine
        constexpr int PL = WQ ? P - 1 : P;
#pragma unroll
        for (int p = 2; p < PL; ++p) {
            if (p < first) continue;
            const u32x4 t = pl[p] & carry;
            pl[p] ^= carry;
            carry = t;
        }
    }

    if constexpr (WQ) {
        // partial counts of the ngrp groups of a query -> its group 0, pairwise through LDS,
        // PCH planes at a time (16 KB of LDS per workgroup whatever P is; the carry stays in registers)
        constexpr int PCH = 4;
        __shared__ u32x4 xch[PCH * 256];
        const uint32_t tid = threadIdx.x;
        for (uint32_t r = 1; r < ngrp; r <<= 1) {
            const uint32_t m = 2u * r - 1u;
            const bool writer = (sub & m) == r, reader = (sub & m) == 0u && sub + r < ngrp;
            u32x4 carry = (u32x4)(0u);
#pragma unroll
            for (int p0 = 0; p0 < P; p0 += PCH) {
                if (writer) {
#pragma unroll
                    for (int p = p0; p < p0 + PCH && p < P; ++p) xch[(p - p0) * 256 + tid] = pl[p];
                }
                __syncthreads();
                if (reader) {
#pragma unroll
                    for (int p = p0; p < p0 + PCH && p < P; ++p) {
                        const u32x4 o = xch[(p - p0) * 256 + tid + r * g];
                        const u32x4 u = pl[p] ^ o;
                        const u32x4 sum = u ^ carry;
                        carry = (u & carry) | (pl[p] & o);
                        pl[p] = sum;
                    }
                }
                __syncthreads();
            }
        }
    }
    // ---- measurement option: how many row chunks were really gathered (threshold bound on/off).
    // Bytes are counted the algorithmic way (row padding excluded), one sharded atomic per wavefront.
    if (a.fetch_count) {
        const uint64_t rb = ((uint64_t)bd.n_docs + 7u) >> 3;
        const uint64_t vb = boff >= rb ? 0ull : (rb - boff < 16ull ? rb - boff : 16ull);
        unsigned long long v = (unsigned long long)nfetch * vb;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const uint32_t lo_ = (uint32_t)__shfl_xor((int)(uint32_t)v, o, 64);
            const uint32_t hi_ = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), o, 64);
            v += ((unsigned long long)hi_ << 32) | lo_;
        }
        if (lane == 0 && v != 0ull)
            atomicAdd(a.fetch_count + ((blockIdx.x * 4u + (uint32_t)wave) & (a.fetch_shards - 1u)), v);
    }

    if constexpr (WQ) {
        if (nsplit > 1u) {
            // partial counts of this workgroup's step range -> global slabs; the workgroup that draws the last ticket of
            // the tile adds the others' slabs to its own registers and runs the epilogue.  Hand-off by the counter form
            // of an agent-scope release / acquire pair (cdna_hip_programming.md, section 5: split-K reduction): plain
            // stores, every wave drains them, barrier, lane 0 releases + takes a ticket, the last arriver acquires.
            __shared__ uint32_t last_flag;
            const uint32_t blk = blockIdx.y * gridDim.x + blockIdx.x;
            const uint32_t ow_n = 256u / ngrp;                                   // lanes that hold totals in a workgroup
            const uint32_t ow = (gi / ngrp) * g + c;
            // (a workgroup's region is sized for the most owner lanes any lane-group width has: 256 / 4)
            u32x4* slab = reinterpret_cast<u32x4*>(a.split_slabs) + ((size_t)blk * nsplit) * (size_t)(P * 64);
            if (sub == 0u) {
#pragma unroll
                for (int p = 0; p < P; ++p) slab[((size_t)blockIdx.z * P + (size_t)p) * ow_n + ow] = pl[p];
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const uint32_t t = __hip_atomic_fetch_add(a.split_cnt + blk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t last = (t == nsplit - 1u) ? 1u : 0u;
                if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                last_flag = last;
            }
            __syncthreads();
            if (last_flag == 0u) return;
            if (sub == 0u) {
                for (uint32_t z = 0; z < nsplit; ++z) {
                    if (z == blockIdx.z) continue;
                    u32x4 carry = (u32x4)(0u);
#pragma unroll
                    for (int p = 0; p < P; ++p) {
                        const u32x4 o = slab[((size_t)z * P + (size_t)p) * ow_n + ow];
                        const u32x4 u = pl[p] ^ o;
                        const u32x4 sum = u ^ carry;
                        carry = (u & carry) | (pl[p] & o);
                        pl[p] = sum;
                    }
                }
            }
        }
    }
    const bool owner = !WQ || sub == 0u;              // the group that holds the query's total counts

    // ---- a7: score >= thr, bit-sliced: carry-out of score + (2^P - thr).
    // valid-document mask first (row padding, inactive lanes)
    const uint64_t doc0 = ((uint64_t)slab * g + c) * 128;
    u32x4 valid;
    {
        uint32_t kw[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint64_t first = doc0 + 32u * w;
            if (!active || !owner || first >= bd.n_docs) kw[w] = 0u;
            else if (first + 32 > bd.n_docs) kw[w] = (1u << (bd.n_docs - first)) - 1u;
            else kw[w] = 0xFFFFFFFFu;
        }
        valid = (u32x4){kw[0], kw[1], kw[2], kw[3]};
    }
    auto ge_mask = [&](uint32_t t) -> u32x4 {          // documents with score >= t (t per lane)
        if (t == 0u) return valid;
        if (t > nt) return (u32x4)(0u);
        const uint32_t K = (1u << P) - t;
        u32x4 cy = (u32x4)(0u);
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const u32x4 both = pl[p] & cy, any = pl[p] | cy;
            cy = ((K >> p) & 1u) ? any : both;
        }
        return cy & valid;
    };
    u32x4 mask = ge_mask(thr);

    // lanes of my (query, batch, slab) group inside the wavefront
    const uint32_t gfirst = (uint32_t)lane & ~(g - 1u);
    const unsigned long long gmask = (g >= 64u) ? ~0ull : (((1ull << g) - 1ull) << gfirst);
    auto group_any = [&](bool v) -> bool { return (__ballot(v) & gmask) != 0ull; };
    auto popc4 = [](const u32x4& m) -> uint32_t {
        return (uint32_t)(__popc(m.x) + __popc(m.y) + __popc(m.z) + __popc(m.w));
    };
    auto group_count = [&](const u32x4& m) -> uint32_t {          // xor-shuffle reduction over the g lanes
        uint32_t v = popc4(m);
        for (int o = (int)(g >> 1); o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o, 64);
        return v;
    };

    // ---- a8 fused: keep the n best documents plus ties with the n-th
    // (scripts/postprocess_cobs.py:31-39): raise the cut to the n-th largest score
    // when more than n documents passed.  The search for the cut is a bisection on the score.
    uint32_t total = group_count(mask);      // documents of this group that will be reported
    uint32_t full_count = total;             // documents that passed -t (what cobs prints in "*header\tN")
    if (a.prune_n > 0u && bd.prune != 0u && gridDim.y == 1) {
        const bool need = full_count > a.prune_n;
        uint32_t lo = thr, hi = nt + 1u;                 // count(>= lo) >= n, count(>= hi) < n
        while (__ballot(need && hi - lo > 1u) != 0ull) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            const uint32_t cnt = group_count(ge_mask(mid));
            if (need && hi - lo > 1u) { if (cnt >= a.prune_n) lo = mid; else hi = mid; }
        }
        if (__ballot(need) != 0ull) {
            const u32x4 cut = ge_mask(lo);
            const uint32_t kept = group_count(cut);
            if (need) { mask = cut; total = kept; }
        }
    }
    if (__ballot(total != 0u) == 0ull) return;           // the usual case: nothing to report

    // ---- a7 ordering + compaction.  Every group with hits writes ONE contiguous run
    //     {query, PM_DOC_COUNT, full_count, slot}   then its hits, best score first, ties by
    //     ascending document index -- the order of cobs' result lines -- so the host never
    //     sorts records, it only orders runs.  The wavefront reserves the space of all its
    //     runs with a single atomicAdd; a run's hits are emitted level by level: the largest
    //     remaining score is found by walking the bit planes from the top (a group-wide
    //     ballot per plane), its documents are written in lane/bit order behind a
    //     ballot + mbcnt prefix (shuffle scan when a lane holds several), then removed.
    const bool leader = (c == 0u) && total != 0u;
    const uint32_t recs = leader ? total + 1u : 0u;
    uint32_t incl = recs;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)incl, o, 64);
        if (lane >= o) incl += t;
    }
    const uint32_t wave_total = (uint32_t)__shfl((int)incl, 63, 64);
    const unsigned long long lead_bal = __ballot(leader);
    unsigned long long basev = 0, baser = 0;
    if (lane == 0) {
        basev = atomicAdd(a.hit_count, (unsigned long long)wave_total);
        baser = atomicAdd(a.hit_count + 1, (unsigned long long)__popcll(lead_bal));
    }
    const uint32_t blo = (uint32_t)__shfl((int)(uint32_t)basev, 0, 64);
    const uint32_t bhi = (uint32_t)__shfl((int)(uint32_t)(basev >> 32), 0, 64);
    const uint32_t rlo = (uint32_t)__shfl((int)(uint32_t)baser, 0, 64);
    const uint32_t rhi = (uint32_t)__shfl((int)(uint32_t)(baser >> 32), 0, 64);
    const uint32_t run_off = (uint32_t)__shfl((int)(incl - recs), (int)gfirst, 64);
    const uint64_t run_base = (((uint64_t)bhi << 32) | blo) + run_off;
    if (leader && run_base < a.hit_cap) {
        a.hits[run_base] = make_uint4(q, 0xFFFFFFFFu, full_count, bd.slot);
        // run directory {query, slot, first record, hits | cut flag}: what the host orders instead of records
        const uint64_t ridx = (((uint64_t)rhi << 32) | rlo) +
            __builtin_amdgcn_mbcnt_hi((uint32_t)(lead_bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)lead_bal, 0u));
        if (ridx < a.run_cap)
            a.runs[ridx] = make_uint4(q, bd.slot, (uint32_t)run_base, total | (full_count != total ? 0x80000000u : 0u));
    }

    u32x4 R = mask;
    uint32_t emitted = 0;
    while (__ballot((R.x | R.y | R.z | R.w) != 0u) != 0ull) {
        u32x4 M = R;
        uint32_t s = 0;
#pragma unroll
        for (int p = P - 1; p >= 0; --p) {
            const u32x4 t = M & pl[p];
            if (group_any((t.x | t.y | t.z | t.w) != 0u)) { M = t; s |= 1u << p; }
        }
        const uint32_t cnt = popc4(M);
        uint32_t excl, lvl;
        if (__ballot(cnt > 1u) == 0ull) {
            const unsigned long long b = __ballot(cnt != 0u) & gmask;
            excl = __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
            lvl = (uint32_t)__popcll(b);
        } else {
            uint32_t in2 = cnt;
            for (uint32_t o = 1; o < g; o <<= 1) {
                const uint32_t t = (uint32_t)__shfl_up((int)in2, (int)o, 64);
                if (c >= o) in2 += t;
            }
            excl = in2 - cnt;
            lvl = (uint32_t)__shfl((int)in2, (int)(gfirst + g - 1u), 64);
        }
        // rank of my first document among the lvl documents of this score level, ascending by document index; the level
        // is written in that order, or mirrored ("cobs_tie_order" 1: equal scores by descending document)
        uint32_t rank = excl;
        const uint64_t lvl_base = run_base + 1u + emitted;
        const uint32_t mw[4] = {M.x, M.y, M.z, M.w};
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            uint32_t m = mw[w];
            while (m != 0u) {
                const int bit = __ffs((int)m) - 1;
                m &= m - 1u;
                const uint64_t pos = lvl_base + (a.tie_desc ? lvl - 1u - rank : rank);
                if (pos < a.hit_cap)
                    a.hits[pos] = make_uint4(q, bd.doc_base + (uint32_t)(doc0 + 32u * w + (uint32_t)bit), s, bd.slot);
                ++rank;
            }
        }
        emitted += lvl;
        R = R & ~M;
    }
}

template <int G, int P>
static hipError_t scan_dispatch_nh(const ScanArgs& a, uint32_t slabs, hipStream_t st) {
    dim3 grid(G > 0 ? a.n_batches * a.tiles : a.total_blocks, slabs, 1);
    if constexpr (P >= 10) {
        if (a.wide_query) {
            grid.z = a.nsplit > 1u ? a.nsplit : 1u;
            if (a.nh == 1) hipLaunchKernelGGL((k_scan<G, P, true, true>), grid, dim3(256), 0, st, a);
            else           hipLaunchKernelGGL((k_scan<G, P, false, true>), grid, dim3(256), 0, st, a);
            return hipGetLastError();
        }
    }
    if (a.wide_query) return hipErrorInvalidValue;
    if (a.nh == 1) hipLaunchKernelGGL((k_scan<G, P, true, false>), grid, dim3(256), 0, st, a);
    else           hipLaunchKernelGGL((k_scan<G, P, false, false>), grid, dim3(256), 0, st, a);
    return hipGetLastError();
}
template <int G>
static hipError_t scan_dispatch_p(const ScanArgs& a, int planes, uint32_t slabs, hipStream_t st) {
    switch (planes) {
        case 3:  return scan_dispatch_nh<G, 3>(a, slabs, st);
        case 7:  return scan_dispatch_nh<G, 7>(a, slabs, st);
        case 10: return scan_dispatch_nh<G, 10>(a, slabs, st);
        case 13: return scan_dispatch_nh<G, 13>(a, slabs, st);
        case 16: return scan_dispatch_nh<G, 16>(a, slabs, st);
        case 20: return scan_dispatch_nh<G, 20>(a, slabs, st);
        case 24: return scan_dispatch_nh<G, 24>(a, slabs, st);
        default: return hipErrorInvalidValue;
    }
}
// queries one workgroup takes: 256 / g lane groups, one query each, or (wide-query form) min(wq_groups, 256 / g) groups per query
uint32_t scan_queries_per_block(int g, uint32_t wq_groups) {
    const uint32_t gpb = 256u / (uint32_t)g;
    return wq_groups ? gpb / (gpb < wq_groups ? gpb : wq_groups) : gpb;
}
hipError_t launch_scan(const ScanArgs& a, int g, int planes, uint32_t slabs, hipStream_t st) {
    if (a.nq == 0 || a.n_batches == 0) return hipSuccess;
    if (slabs > 1 && a.n_batches != 1) return hipErrorInvalidValue;
    switch (g) {
        case 0:  return scan_dispatch_p<0>(a, planes, slabs, st);   // mixed widths: grid = a.total_blocks
        case 1:  return scan_dispatch_p<1>(a, planes, slabs, st);
        case 2:  return scan_dispatch_p<2>(a, planes, slabs, st);
        case 4:  return scan_dispatch_p<4>(a, planes, slabs, st);
        case 8:  return scan_dispatch_p<8>(a, planes, slabs, st);
        case 16: return scan_dispatch_p<16>(a, planes, slabs, st);
        case 32: return scan_dispatch_p<32>(a, planes, slabs, st);
        case 64: return scan_dispatch_p<64>(a, planes, slabs, st);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_hash_terms(const uint8_t* seq, const QDesc* qd, const uint32_t* blk_query,
                             uint64_t n_slots, uint32_t k, int canon, uint32_t nh,
                             uint64_t* hashes, hipStream_t st) {
    if (n_slots == 0) return hipSuccess;
    const uint64_t blocks = (n_slots + 255) / 256;
    if (k == 31)
        hipLaunchKernelGGL(k_hash_terms<31>, dim3((uint32_t)blocks), dim3(256), 0, st,
                           seq, qd, blk_query, n_slots, k, canon, nh, hashes);
    else
        hipLaunchKernelGGL(k_hash_terms<0>, dim3((uint32_t)blocks), dim3(256), 0, st,
                           seq, qd, blk_query, n_slots, k, canon, nh, hashes);
    return hipGetLastError();
}

// ----------------------------------------------------------- index helpers
// dst row r = src row r (row_bytes) followed by zero padding up to stride.
__global__ __launch_bounds__(256) void k_restride(
    const uint8_t* __restrict__ src, uint64_t row_bytes, uint8_t* __restrict__ dst,
    uint64_t stride, uint64_t n_rows)
{
    const uint64_t chunks_per_row = stride >> 4;
    const uint64_t total = n_rows * chunks_per_row;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = i / chunks_per_row, ch = i - r * chunks_per_row;
        const uint64_t b0 = ch * 16;
        const uint8_t* s = src + r * row_bytes + b0;
        uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
        for (int b = 0; b < 16; ++b)
            if (b0 + b < row_bytes) w[b >> 2] |= (uint32_t)s[b] << (8 * (b & 3));
        *reinterpret_cast<uint4*>(dst + r * stride + b0) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}
hipError_t launch_restride(const uint8_t* src, uint64_t row_bytes, uint8_t* dst, uint64_t stride,
                           uint64_t n_rows, hipStream_t st) {
    if (n_rows == 0) return hipSuccess;
    const uint64_t total = n_rows * (stride >> 4);
    uint64_t blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(k_restride, dim3((uint32_t)blocks), dim3(256), 0, st, src, row_bytes, dst, stride, n_rows);
    return hipGetLastError();
}

// copies n 64-bit words to device-mapped host memory (record counters of a search)
__global__ void k_publish(const unsigned long long* __restrict__ src, unsigned long long* dst, int n) {
    if ((int)threadIdx.x < n) __hip_atomic_store(dst + threadIdx.x, src[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
hipError_t launch_publish(const unsigned long long* src, unsigned long long* dst_mapped, int n, hipStream_t st) {
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, st, src, dst_mapped, n);
    return hipGetLastError();
}

// a7 on the device, second half: the host has ordered the run directory by (slot, query);
// this copies every run to its final place.  plan[i] = {first source record, first
// destination record, records, -}.  One wavefront per run, 16 bytes per lane.
__global__ __launch_bounds__(256) void k_permute_runs(const uint4* __restrict__ plan, uint32_t n_plan,
                                                       const uint4* __restrict__ src, uint4* __restrict__ dst)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t waves = gridDim.x * 4u;
    for (uint32_t e = blockIdx.x * 4u + (threadIdx.x >> 6); e < n_plan; e += waves) {
        const uint4 p = plan[e];
        for (uint32_t i = lane; i < p.z; i += 64u) dst[(uint64_t)p.y + i] = src[(uint64_t)p.x + i];
    }
}
hipError_t launch_permute_runs(const uint4* plan, uint32_t n_plan, const uint4* src, uint4* dst, hipStream_t st) {
    if (n_plan == 0) return hipSuccess;
    uint32_t blocks = (n_plan + 3u) / 4u;
    if (blocks > 65536u) blocks = 65536u;
    hipLaunchKernelGGL(k_permute_runs, dim3(blocks), dim3(256), 0, st, plan, n_plan, src, dst);
    return hipGetLastError();
}

// a7 on the device, third part: a (slot, query) group that the scan wrote as SEVERAL runs -- one per 1024-byte column slab
// of a row wider than that, or one per sub-index of a compact index -- becomes one list in cobs' line order.  Every run
// is ordered (score descending, ties by ascending document) and the runs cover disjoint, ordered document ranges, so a
// record's place is a count: the records of the other runs with a higher score, those with the same score in runs of
// lower documents, and the records ahead of it in its own run.  group = {first destination record, runs, first entry in
// `runs`, -}; run = {first source record (behind the count record), records, -, -}.  One workgroup per group.
__global__ __launch_bounds__(256) void k_merge_runs(const uint4* __restrict__ groups, uint32_t n_groups,
                                                     const uint4* __restrict__ runs, const uint4* __restrict__ src,
                                                     uint4* __restrict__ dst, uint32_t tie_desc)
{
    constexpr uint32_t kStage = 512;
    __shared__ uint32_t s_begin[kStage], s_len[kStage], s_doc[kStage];
    for (uint32_t gi = blockIdx.x; gi < n_groups; gi += gridDim.x) {
        const uint4 g = groups[gi];
        const uint32_t nr = g.y, staged = nr < kStage ? nr : kStage;
        __syncthreads();
        for (uint32_t r = threadIdx.x; r < staged; r += blockDim.x) {
            const uint4 rn = runs[g.z + r];
            s_begin[r] = rn.x; s_len[r] = rn.y; s_doc[r] = src[rn.x].y;
        }
        __syncthreads();
        auto run_of = [&](uint32_t r, uint32_t* b, uint32_t* l, uint32_t* d) {
            if (r < staged) { *b = s_begin[r]; *l = s_len[r]; *d = s_doc[r]; }
            else { const uint4 rn = runs[g.z + r]; *b = rn.x; *l = rn.y; *d = src[rn.x].y; }
        };
        for (uint32_t r = 0; r < nr; ++r) {
            uint32_t b, l, d0;
            run_of(r, &b, &l, &d0);
            for (uint32_t i = threadIdx.x; i < l; i += blockDim.x) {
                const uint4 rec = src[(uint64_t)b + i];
                const uint32_t sc = rec.z;
                uint32_t pos = i;
                for (uint32_t o = 0; o < nr; ++o) {
                    if (o == r) continue;
                    uint32_t ob, ol, od;
                    run_of(o, &ob, &ol, &od);
                    uint32_t lo = 0, hi = ol;                      // records of run o with score > sc (scores descend)
                    while (lo < hi) { const uint32_t m = (lo + hi) >> 1; if (src[(uint64_t)ob + m].z > sc) lo = m + 1; else hi = m; }
                    pos += lo;
                    if (tie_desc ? od > d0 : od < d0) {            // lower documents: their equal scores come first (higher: "cobs_tie_order" 1)
                        uint32_t lo2 = lo, hi2 = ol;
                        while (lo2 < hi2) { const uint32_t m = (lo2 + hi2) >> 1; if (src[(uint64_t)ob + m].z >= sc) lo2 = m + 1; else hi2 = m; }
                        pos += lo2 - lo;
                    }
                }
                dst[(uint64_t)g.x + pos] = rec;
            }
        }
    }
}
// The same merge in O(records + runs) for the usual case -- scores below kMergeBins (reads, genes: the score is at most
// the query's k-mer count) and at most kMergeRuns runs per group: a counting sort on the score that keeps the runs in
// document order.  (1) histogram of the group's scores in LDS, turned into "records with a higher score" per score;
// (2) the runs are ranked by their first document and walked in that order (descending with tie_desc): a record's
// place is [records with a higher score] + [records of the same score in the runs walked so far] + [its distance from
// the first record of that score in its own run]; the per-score cursor advances between runs.  k_merge_runs above is
// O(records x runs x log) and stays for groups outside these bounds (a compact index of hundreds of sub-indexes used to
// go through it: VERDICT r3 "correct, not fast").
constexpr uint32_t kMergeBins = 4096, kMergeRuns = 1024;
__global__ __launch_bounds__(256) void k_merge_runs_hist(const uint4* __restrict__ groups, uint32_t n_groups,
                                                          const uint4* __restrict__ runs, const uint4* __restrict__ src,
                                                          uint4* __restrict__ dst, uint32_t tie_desc)
{
    __shared__ uint32_t above[kMergeBins];       // histogram, then: records of the group with a higher score
    __shared__ uint32_t cursor[kMergeBins];      // records of that score placed by the runs walked so far
    __shared__ uint32_t s_begin[kMergeRuns], s_len[kMergeRuns], s_doc[kMergeRuns], s_order[kMergeRuns];
    __shared__ uint32_t s_scan[256];
    const uint32_t tid = threadIdx.x;
    for (uint32_t gi = blockIdx.x; gi < n_groups; gi += gridDim.x) {
        const uint4 g = groups[gi];
        const uint32_t nr = g.y;                 // <= kMergeRuns (the host routes larger groups to k_merge_runs)
        __syncthreads();
        for (uint32_t b = tid; b < kMergeBins; b += 256u) { above[b] = 0u; cursor[b] = 0u; }
        for (uint32_t r = tid; r < nr; r += 256u) {
            const uint4 rn = runs[g.z + r];
            s_begin[r] = rn.x; s_len[r] = rn.y; s_doc[r] = src[rn.x].y;
        }
        __syncthreads();
        // runs in document order (runs cover disjoint document ranges): rank by counting
        for (uint32_t r = tid; r < nr; r += 256u) {
            const uint32_t d = s_doc[r];
            uint32_t rank = 0;
            for (uint32_t o = 0; o < nr; ++o) rank += (tie_desc ? s_doc[o] > d : s_doc[o] < d) ? 1u : 0u;
            s_order[rank] = r;
        }
        // (1) histogram of scores
        for (uint32_t r = 0; r < nr; ++r) {
            const uint32_t b = s_begin[r], l = s_len[r];
            for (uint32_t i = tid; i < l; i += 256u) atomicAdd(&above[src[(uint64_t)b + i].z], 1u);
        }
        __syncthreads();
        // exclusive suffix sum over the bins: thread t owns bins [16 t, 16 t + 16), highest scores first
        {
            constexpr uint32_t per = kMergeBins / 256u;
            uint32_t sum = 0;
            for (uint32_t k = 0; k < per; ++k) sum += above[kMergeBins - 1u - (tid * per + k)];
            s_scan[tid] = sum;
            __syncthreads();
            for (uint32_t o = 1; o < 256u; o <<= 1) {
                const uint32_t t = tid >= o ? s_scan[tid - o] : 0u;
                __syncthreads();
                s_scan[tid] += t;
                __syncthreads();
            }
            uint32_t run = s_scan[tid] - sum;                           // records in the bins of the threads before me
            for (uint32_t k = 0; k < per; ++k) {
                const uint32_t bin = kMergeBins - 1u - (tid * per + k);
                const uint32_t h = above[bin];
                above[bin] = run;
                run += h;
            }
        }
        __syncthreads();
        // (2) the runs in document order
        for (uint32_t k = 0; k < nr; ++k) {
            const uint32_t r = s_order[k], b = s_begin[r], l = s_len[r];
            for (uint32_t i = tid; i < l; i += 256u) {
                const uint4 rec = src[(uint64_t)b + i];
                const uint32_t sc = rec.z;
                uint32_t lo = 0, hi = i;                                // first record of my score in my run (scores descend)
                while (lo < hi) { const uint32_t m = (lo + hi) >> 1; if (src[(uint64_t)b + m].z > sc) lo = m + 1; else hi = m; }
                dst[(uint64_t)g.x + above[sc] + cursor[sc] + (i - lo)] = rec;
            }
            __syncthreads();
            for (uint32_t i = tid; i < l; i += 256u) atomicAdd(&cursor[src[(uint64_t)b + i].z], 1u);
            __syncthreads();
        }
    }
}
hipError_t launch_merge_runs_hist(const uint4* groups, uint32_t n_groups, const uint4* runs, const uint4* src, uint4* dst,
                                  uint32_t tie_desc, hipStream_t st) {
    if (n_groups == 0) return hipSuccess;
    hipLaunchKernelGGL(k_merge_runs_hist, dim3(n_groups < 65536u ? n_groups : 65536u), dim3(256), 0, st, groups, n_groups, runs, src, dst, tie_desc);
    return hipGetLastError();
}
uint32_t merge_hist_max_score() { return kMergeBins - 1u; }
uint32_t merge_hist_max_runs() { return kMergeRuns; }

hipError_t launch_merge_runs(const uint4* groups, uint32_t n_groups, const uint4* runs, const uint4* src, uint4* dst,
                             uint32_t tie_desc, hipStream_t st) {
    if (n_groups == 0) return hipSuccess;
    hipLaunchKernelGGL(k_merge_runs, dim3(n_groups < 65536u ? n_groups : 65536u), dim3(256), 0, st, groups, n_groups, runs, src, dst, tie_desc);
    return hipGetLastError();
}

}  // namespace pm

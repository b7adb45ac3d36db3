// pm_internal.h -- shared between the host translation units (pm_*.cpp, see pm_host.h) and
// pm_kernels.hip (gfx950 kernels).  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pm {

// Per-query descriptor in HBM (16 B).  Terms of a query live in 8-slot blocks
// starting at block `pad_blk`; slots past n_terms are padding.
struct QDesc {
    uint32_t n_terms;
    uint32_t pad_blk;
    uint32_t seq_lo, seq_hi;   // byte offset of the sequence in the packed buffer
};

// One resident batch index as the scan kernel sees it (64 B, read through the
// scalar cache: the batch is uniform per workgroup).
struct BatchDesc {
    const uint8_t* matrix;      // row r at matrix + r*stride, 16-B aligned
    uint64_t stride;
    uint64_t sig_size;          // S
    uint64_t barrett_m;         // floor(2^64 / S); 0 encodes S == 1
    uint32_t n_docs;
    uint32_t slot;
    uint32_t doc_base;          // first document of this unit inside its index (compact sub-indexes)
    uint32_t prune;             // 1: ScanArgs.prune_n applies to this unit
    uint32_t lanes;             // lanes per row (power of two): read by the mixed-width launch
    uint32_t block_begin;       // first workgroup of this unit in a mixed-width launch
    uint32_t q_first;           // this unit scans positions q_first ... min(q_end, ScanArgs.nq) - 1 of the launch's query list
    uint32_t q_end;             // (0, 0xFFFFFFFF: all of them)
};

struct ScanArgs {
    const BatchDesc* batches;   // batches of this launch (same lanes-per-row class)
    uint32_t        n_batches;
    uint32_t        tiles;      // workgroups per batch; blockIdx.x = batch*tiles + tile (a unit with fewer queries leaves its last tiles empty)
    uint32_t        total_blocks;   // mixed-width launch: sum of the per-batch workgroup counts
    const uint64_t* hashes;     // [blk][hash j][8]
    const QDesc*    qd;
    const uint32_t* thr;        // per query minimum score (0 = keep all)
    const uint32_t* qmap;       // launch-local index -> query id
    uint32_t        nq;         // queries in this launch
    uint32_t        nh;
    uint32_t        prune_n;    // > 0: keep the n best documents (+ ties) per (query, batch)
    uint32_t        bound;      // 1: stop fetching lines whose documents cannot reach thr any more
    uint4*          hits;       // pm_hit_t records
    unsigned long long* hit_count;  // [0] records written (run headers included), [1] runs
    uint64_t        hit_cap;
    unsigned long long* fetch_count;   // null, or fetch_shards counters of gathered algorithmic bytes
    uint32_t        fetch_shards;      // power of two
    uint32_t        pad_;
    uint4*          runs;       // run directory {query, slot, first record, hits | cut flag << 31}
    uint64_t        run_cap;
    uint32_t        wide_query; // 1: several lane groups of a workgroup share one query (k_scan<..., WQ>)
    uint32_t        wq_groups;  // ... at most this many (power of two, 4 ... 256; capped by 256 / lanes per row)
    uint32_t        nsplit;     // wide-query form: workgroups (gridDim.z) that share the steps of a tile's queries (1 = none)
    uint32_t        tie_desc;   // 1: documents of equal score are written by descending index (pm_set_option "cobs_tie_order")
    uint4*          split_slabs;   // nsplit > 1: [workgroup (x, y)][z][plane][256 / groups per query] partial count planes
    uint32_t*       split_cnt;     // ... and one arrival counter per workgroup (x, y), zeroed ahead of the launch
};

// launchers (pm_kernels.hip); all asynchronous on `st`, return hipError_t
hipError_t launch_hash_terms(const uint8_t* seq, const QDesc* qd, const uint32_t* blk_query,
                             uint64_t n_slots, uint32_t k, int canon, uint32_t nh,
                             uint64_t* hashes, hipStream_t st);
// g = lanes per row (1..64 pow2; 0 = mixed, taken per batch from BatchDesc.lanes),
// planes = counter bit planes (3,7,10,13,16,20,24);
// slabs > 1 only with n_batches == 1 (rows wider than 1024 B)
hipError_t launch_scan(const ScanArgs& a, int g, int planes, uint32_t slabs, hipStream_t st);
uint32_t scan_queries_per_block(int g, uint32_t wq_groups);   // wq_groups = 0: plain form
uint64_t barrett_m(uint64_t S);
hipError_t launch_restride(const uint8_t* src, uint64_t row_bytes, uint8_t* dst, uint64_t stride,
                           uint64_t n_rows, hipStream_t st);
hipError_t launch_publish(const unsigned long long* src, unsigned long long* dst_mapped, int n, hipStream_t st);
hipError_t launch_permute_runs(const uint4* plan, uint32_t n_plan, const uint4* src, uint4* dst, hipStream_t st);
hipError_t launch_merge_runs(const uint4* groups, uint32_t n_groups, const uint4* runs, const uint4* src, uint4* dst,
                             uint32_t tie_desc, hipStream_t st);
// the O(records + runs) form for groups whose scores are at most merge_hist_max_score() and that have at most
// merge_hist_max_runs() runs
hipError_t launch_merge_runs_hist(const uint4* groups, uint32_t n_groups, const uint4* runs, const uint4* src, uint4* dst,
                                  uint32_t tie_desc, hipStream_t st);
uint32_t merge_hist_max_score();
uint32_t merge_hist_max_runs();

}  // namespace pm

#!/usr/bin/env python3
"""bench.py -- k-mers matched/sec of the MI355X COBS matching stage against a
661k-shaped synthetic index set, with the HBM roofline of the scan kernel and
the CPU (oracle "port") baseline timed beside it.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W

One step = one pass of the whole hot path (canonicalise + XXH64, row map, row
gather + bit-sliced count, threshold + compaction, gather of the hit records to
rank 0 and host ordering) of the query batch over every resident phylogenetic
batch index.  Workload (BASELINE.json configs[2], SURVEY.md 8d "config 3"): the
64 661k-shaped batches that fit one GPU (~213 GB of signatures), 100 000
synthetic 150-bp queries (120 31-mers each), threshold 0.7.  With N > 1 the same
64 batches are sharded statically over the ranks (strong scaling).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBPS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline(shapes, fasta_seqs, qlen, threshold, target_s, sample_gb, log):
    """Times oracle/cobs_oracle.c (kind "port": a restatement of the cobs classic
    search, NOT bioconda cobs 0.2.1) on the host cores: same document counts per
    batch (same algorithmic bytes per k-mer), rows scaled down to fit host RAM."""
    from oracle import oracle as O
    from phylign_amd import workload as W
    cores = os.cpu_count() or 1
    total = sum(s.index_bytes for s in shapes)
    div = max(1, int(np.ceil(total / (sample_gb * 1e9))))
    small = W.scale_shapes(shapes, div)
    t0 = time.time()
    mats = []
    for s in small:
        mats.append((s, O.synth_fill(661, s.batch_id, s.signature_size, s.n_docs, cores)))
    t_gen = time.time() - t0
    hdrs = []
    for s, _ in mats:
        h = O.Header()
        h.term_size, h.canonicalize, h.num_hashes = 31, 1, 1
        h.n_docs, h.signature_size, h.row_bytes = s.n_docs, s.signature_size, s.row_bytes
        hdrs.append(h)

    def run(nq):
        seqs = fasta_seqs[:nq].tobytes()
        t = time.time()
        hits = 0
        for (s, m), h in zip(mats, hdrs):
            hits += O.baseline_run(m, s.row_bytes, h, seqs, qlen, nq, threshold, cores)
        return time.time() - t, hits
    nq = min(256, len(fasta_seqs))
    tt, _ = run(nq)
    for _ in range(4):                  # grow the sample until it costs about target_s of CPU time
        if tt >= 0.5 * target_s or nq >= len(fasta_seqs):
            break
        nq = int(min(len(fasta_seqs), max(nq * 2, nq * target_s / max(tt, 1e-6))))
        tt, _ = run(nq)
    terms = nq * (qlen - 30)
    alg = terms * sum(s.row_bytes for s in shapes)
    log(f"[cpu_baseline] gen {t_gen:.1f}s, {nq} queries in {tt:.2f}s on {cores} threads")
    return {
        "value": terms / tt, "unit": "k-mers/s", "cores": cores, "kind": "port",
        "sample": (f"{nq} of the same queries x all {len(shapes)} batch shapes with rows/{div} "
                   f"({sum(s.index_bytes for s in small) / 1e9:.2f} GB resident in host RAM), "
                   f"{tt:.1f}s wall, oracle/cobs_oracle.c COBS-restatement (not bioconda cobs 0.2.1)"),
        "algorithmic_GBps": alg / tt / 1e9,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="config3", choices=["config3", "small", "config2", "full"])
    ap.add_argument("--queries", type=int, default=100000)
    ap.add_argument("--qlen", type=int, default=150)
    ap.add_argument("--threshold", type=float, default=0.7)
    ap.add_argument("--nb-best-hits", type=int, default=100, help="config.yaml:23 nb_best_hits (on-device top-n + ties)")
    ap.add_argument("--rows-divisor", type=int, default=1, help="shrink every batch's row count (quick runs)")
    ap.add_argument("--layout", type=int, default=0, help="0 auto, 1 compact, 2 line-aligned")
    ap.add_argument("--no-threshold-bound", action="store_true",
                    help="fetch every signature row like cobs does (the product default stops fetching lines whose "
                         "documents cannot reach the threshold any more; results are identical)")
    ap.add_argument("--skip-fetch-all", action="store_true", help="do not append the comparison pass with the bound off")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-target-s", type=float, default=12.0)
    ap.add_argument("--cpu-sample-gb", type=float, default=0.85)
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="single process: hold only the shard rank --emulate-rank would get in an N-way split "
                         "(estimates the per-rank step time of a strong-scaling run; not a reported number)")
    ap.add_argument("--emulate-rank", type=int, default=0)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world

    def log(msg):
        if rank == 0:
            print(msg, file=sys.stderr, flush=True)

    import torch
    import torch.distributed as dist
    from phylign_amd import _lib as pm
    from phylign_amd import workload as W
    from phylign_amd.dist import PackedGather

    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # BENCH_DIST_BACKEND=gloo + BENCH_SHARE_GPU=1: functional check of the N>1 code path
    # with several ranks on ONE GPU (RCCL refuses duplicate devices); never a reported number
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    if os.environ.get("BENCH_SHARE_GPU"):
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    pm.init(local_rank)
    pm.set_option("threshold_bound", 0 if args.no_threshold_bound else 1)
    dev = pm.device_info()
    log(f"[bench] {dev['name']} free {dev['hbm_free'] / 1e9:.1f} GB of {dev['hbm_total'] / 1e9:.1f} GB")

    shapes = W.select(args.workload)
    if args.rows_divisor > 1:
        shapes = W.scale_shapes(shapes, args.rows_divisor)
    nparts = args.emulate_world if (args.emulate_world and world == 1) else world
    parts = W.assign_batches(shapes, nparts, capacity_bytes=int(dev["hbm_total"] * 0.85))
    slot_of = {}     # global slot -> shape position (rank-major numbering)
    base = 0
    bases = []
    for r in range(nparts):
        bases.append(base)
        for i, pos in enumerate(parts[r]):
            slot_of[base + i] = pos
        base += len(parts[r])
    part_id = args.emulate_rank if nparts != world else rank
    mine = parts[part_id]

    # ---- inputs resident in HBM before the timed region -------------------
    t0 = time.time()
    fasta, seqs = W.make_queries(args.queries, args.qlen, seed=31)
    q = pm.Queries(fasta, term_size=31)
    nq, n_terms = q.count()
    terms_per_q = args.qlen - 30
    hashes = q.hash_terms(1, 1)
    plan, sure_hits = W.plant_plan(hashes, nq, terms_per_q, shapes)
    del hashes
    indexes = []
    for pos in mine:
        s = shapes[pos]
        ix = pm.Index.synth(s.batch_id, s.n_docs, s.signature_size, 1, 31, 661, layout=args.layout)
        if pos in plan:
            ix.plant(*plan[pos])
        indexes.append(ix)
    infos = [ix.info for ix in indexes]
    resident = sum(i.device_bytes for i in infos)
    log(f"[bench] rank0: {len(indexes)} batches, {resident / 1e9:.1f} GB of signatures resident, "
        f"{nq} queries / {n_terms} k-mers, setup {time.time() - t0:.1f}s")

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    last = {}
    pg_dev = "cuda" if backend == "nccl" else "cpu"
    packed = PackedGather(1 << 16, pg_dev)          # 64 Ki records (1 MiB) per rank in one collective

    phase = {"search": 0.0, "gather": 0.0, "host": 0.0}

    def step():
        t_a = time.perf_counter()
        res = pm.search(indexes, q, args.threshold, slot_base=bases[part_id], nb_best_hits=args.nb_best_hits)
        st = res.stats
        t_b = time.perf_counter()
        # every rank brings its own records to its host and puts them into cobs order there (in
        # parallel); ranks own disjoint, increasing slot ranges, so the rank-order concatenation
        # that the gather produces on rank 0 is already globally ordered
        mine_sorted = res.hits()
        n_local = len(mine_sorted)
        t_c = time.perf_counter()
        if world == 1:
            host = mine_sorted
        else:
            t = torch.from_numpy(mine_sorted.view(np.int32).reshape(-1, 4))
            if n_local <= packed.cap:
                packed.records_view()[:n_local].copy_(t)
                g = packed.gather(n_local)
            else:
                g = packed.gather(n_local, overflow=t.cuda() if pg_dev == "cuda" else t)
            host = g.cpu().numpy().view(pm.HIT_DTYPE).reshape(-1) if rank == 0 else None
        t_d = time.perf_counter()
        phase["search"] += t_b - t_a; phase["host"] += t_c - t_b; phase["gather"] += t_d - t_c
        last["stats"], last["launches"], last["hits"] = st, res.launches(), host
        res.free()

    def timed_run(warmup, steps):
        """W untimed steps, then exactly K steps between barrier + synchronize; MAX over ranks"""
        for _ in range(warmup):
            step()
        sync()
        phase.update(search=0.0, gather=0.0, host=0.0)
        groups = {}      # kernel instantiation -> [algorithmic bytes, ms, launches, batches] over the timed steps
        t_start = time.perf_counter()
        for _ in range(steps):
            step()
            for L in last["launches"]:
                g = groups.setdefault(L["kernel"], [0.0, 0.0, 0, 0])
                g[0] += L["algorithmic_bytes"]; g[1] += L["ms"]; g[2] += 1; g[3] = L["n_batches"]
        sync()
        elapsed = time.perf_counter() - t_start
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, groups, dict(phase)

    def roofline_of(groups, steps, mode):
        """dominant scan kernel: algorithmic bytes per launch / hipEvent launch duration vs the HBM peak"""
        if not groups:
            return None
        name, (abytes, ms, launches, nb) = max(groups.items(), key=lambda kv: kv[1][1])
        achieved = abytes / (ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("workload") == args.workload and tj.get("queries") == args.queries and tj.get("kernel") == name:
                    traffic = tj.get(mode, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        r = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
             "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "kernel": name,
             "launches_per_step": launches / steps, "batches_per_launch": nb, "avg_launch_ms": ms / launches,
             "algorithmic_bytes_per_launch": abytes / launches,
             "all_scan_kernels_GBps": sum(g[0] for g in groups.values()) / (sum(g[1] for g in groups.values()) * 1e-3) / 1e9}
        if traffic:
            r["hbm_GBps_from_traffic"] = traffic / (ms / launches * 1e-3) / 1e9
        return r

    mode = "fetch_all_rows" if args.no_threshold_bound else "threshold_bound"
    elapsed, groups, phase_main = timed_run(args.warmup, args.steps)
    ms_per_step = elapsed / args.steps * 1e3
    value = n_terms / (elapsed / args.steps)
    roof = roofline_of(groups, args.steps, mode)
    hits_main, st_main = last["hits"], last["stats"]

    # the same K steps with the threshold bound switched off: the scan then fetches every
    # signature row like `cobs query` does, which is the figure to hold against the HBM roofline
    fetch_all = None
    if not args.no_threshold_bound and not args.emulate_world and not args.skip_fetch_all:
        pm.set_option("threshold_bound", 0)
        e2, g2, _ = timed_run(1, args.steps)
        pm.set_option("threshold_bound", 1)
        fetch_all = {"value": n_terms / (e2 / args.steps), "unit": "k-mers/s", "ms_per_step": e2 / args.steps * 1e3,
                     "hbm_fraction_whole_step": sum(s.row_bytes for s in shapes) * n_terms / (e2 / args.steps) / (HBM_PEAK_GBPS * 1e9 * world),
                     "roofline": roofline_of(g2, args.steps, "fetch_all_rows"),
                     "hits_identical": bool(rank != 0 or (hits_main is not None and np.array_equal(hits_main, last["hits"])))}
    phase.update(phase_main)
    last["hits"], last["stats"] = hits_main, st_main

    st = last["stats"]
    alg_total = sum(shapes[p].row_bytes for p in range(len(shapes))) * n_terms

    out = {
        "metric": "query k-mers matched/sec vs 661k-shaped COBS index",
        "value": value, "unit": "k-mers/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {len(shapes)} 661k-shaped batches "
                               f"({sum(s.index_bytes for s in shapes) / 1e9:.1f} GB of signatures, "
                               f"{sum(s.row_bytes for s in shapes)} row bytes per k-mer), "
                               f"{nq} synthetic {args.qlen}-bp queries ({terms_per_q} 31-mers each), threshold {args.threshold}",
                   "batches": len(shapes), "queries": nq, "query_len": args.qlen, "k": 31,
                   "num_hashes": 1, "threshold": args.threshold, "nb_best_hits": args.nb_best_hits,
                   "rows_divisor": args.rows_divisor,
                   "sharding": f"{world} rank(s), static LPT batch assignment, one gather of hit records"},
        "hbm_fraction_whole_step": alg_total / (elapsed / args.steps) / (HBM_PEAK_GBPS * 1e9 * world),
        "hits": int(len(last["hits"])) if last["hits"] is not None else None,
        "planted_pairs_at_or_above_threshold": sure_hits,
        "rank0_ms": {"kernels_total": st.ms_total, "hash": st.ms_hash, "scan": st.ms_scan,
                     "host_search_call": phase["search"] / args.steps * 1e3,
                     "host_hit_gather": phase["gather"] / args.steps * 1e3,
                     "host_d2h_and_order": phase["host"] / args.steps * 1e3},
        "scan_launches": {k: {"launches_per_step": v[2] / args.steps, "batches": v[3], "avg_ms": v[1] / v[2],
                              "algorithmic_GBps": v[0] / (v[1] * 1e-3) / 1e9} for k, v in groups.items()},
        "roofline": roof,
        "arithmetic": "bitwise AND / carry-save adders on u32 words (bit-sliced per-document counters), u64 integer hashing",
        "scan_mode": ("fetch_all_rows" if args.no_threshold_bound else
                      "threshold_bound: a signature line is no longer fetched once none of its documents can reach "
                      "ceil(threshold*k-mers) (count so far + k-mers left); hit lists and scores are bit-identical to "
                      "the fetch-everything scan (see fetch_all_rows and tests/test_gpu_fullsize.py)"),
        "fetch_all_rows": fetch_all,
    }
    if args.emulate_world:
        out["emulated_shard"] = f"rank {part_id} of {nparts}"
    if rank == 0 and not args.emulate_world and last["hits"] is not None and len(last["hits"]) < sure_hits:
        sys.exit(f"bench self-check failed: {len(last['hits'])} hits < {sure_hits} planted pairs")
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(shapes, seqs, args.qlen, args.threshold,
                                           args.cpu_target_s, args.cpu_sample_gb, log)
        out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

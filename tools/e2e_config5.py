#!/usr/bin/env python3
"""BASELINE configs[3]/[4] on one rank's shard of the full 661k collection (SURVEY.md 8d "Config 5": report
match-only time and end-to-end time separately).  Emulates rank --rank of a --world-way split on ONE GPU:
its 38 batches (~135 GB of synthetic 661k-shaped signatures) are generated in HBM (outside the timed
region, like an index that a resident server holds), then phylign_amd.match_stage.run_stage -- the
product's stage code -- turns a query FASTA into the 38 `03_match/*.gz` files and the `04_filter` FASTA.

    python3 tools/e2e_config5.py --queries 1000000 [--clustered] [--max-group 1] --out gpurun_out/e2e

Prints one JSON line: parse / match-only (hash + scan kernels) / D2H / format / gzip / merge / emit /
end-to-end wall, output sizes, launches.  --max-group 1 searches batch by batch (the round-2 stage)."""
import argparse
import json
import os
import shutil
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from phylign_amd import _lib as pm  # noqa: E402
from phylign_amd import match_stage as MS  # noqa: E402
from phylign_amd import workload as W  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--workload", default="full")
    ap.add_argument("--queries", type=int, default=1_000_000)
    ap.add_argument("--qlen", type=int, default=150)
    ap.add_argument("--nb-best-hits", type=int, default=100)
    ap.add_argument("--threshold", type=float, default=0.7)
    ap.add_argument("--clustered", action="store_true",
                    help="every query gets a home batch among ALL batches of the workload (query i -> batch i mod B); the ones "
                         "whose home is in this shard find hundreds of documents around the threshold there")
    ap.add_argument("--max-group", type=int, nargs="*", default=[0], help="0 = all resident batches in one search; several values = several runs")
    ap.add_argument("--bound", type=int, default=1)
    ap.add_argument("--query-chunk", type=int, default=0, help="search the query file in chunks of this many reads (0 = at once)")
    ap.add_argument("--piece-mb", type=int, default=0,
                    help="> 0: the query file is cut into pieces of about this many MB that are parsed by a thread of their own "
                         "INSIDE the timed stage (match_stage's default, 48): parsing overlaps the searches of earlier pieces")
    ap.add_argument("--out", default="gpurun_out/e2e")
    ap.add_argument("--filter-only", action="store_true", help="match_stage --filter-only: no per-batch .gz files, only the 04_filter FASTA")
    ap.add_argument("--json", default=None, help="append the JSON lines to this file")
    args = ap.parse_args()

    pm.init(0)
    pm.set_option("threshold_bound", args.bound)
    shapes = W.select(args.workload)
    mine = W.assign_batches(shapes, args.world)[args.rank]
    sub = [shapes[p] for p in mine]
    t0 = time.perf_counter()
    fasta, _ = W.make_queries(args.queries, args.qlen, seed=5)
    t_make = time.perf_counter() - t0
    t0 = time.perf_counter()
    pieces = MS.split_prepared_fasta(fasta, args.query_chunk)
    qs = [pm.Queries(p) for p in pieces]
    t_parse = time.perf_counter() - t0
    q = qs[0]
    nq, n_terms = sum(x.count()[0] for x in qs), sum(x.count()[1] for x in qs)
    del pieces
    t0 = time.perf_counter()
    nq0 = q.count()[0]                                   # planted reads come from the first chunk
    hashes = q.hash_terms(1, 1)
    plan, sure = W.plant_plan(hashes, nq0, args.qlen - 30, sub, every=max(1, nq0 // 400), docs_per_query=12)
    del hashes
    ixs = {}
    for i, s in enumerate(sub):
        ix = pm.Index.synth(s.batch_id, s.n_docs, s.signature_size, 1, 31, 661)
        if i in plan:
            ix.plant(*plan[i])
        if args.clustered:
            for x in qs:
                ix.plant_cluster(x, mine[i], len(shapes), seed=97)
        ixs[s.batch] = ix
    t_gen = time.perf_counter() - t0
    names = sorted(ixs)
    resident = sum(ix.info.device_bytes for ix in ixs.values())
    alg_per_kmer = sum(s.row_bytes for s in sub)
    print(f"[e2e] shard {args.rank}/{args.world}: {len(sub)} batches, {resident / 1e9:.1f} GB resident ({t_gen:.1f} s to generate), "
          f"{nq} queries / {n_terms} k-mers (parse {t_parse:.2f} s)", file=sys.stderr, flush=True)
    src = MS.ResidentSource(ixs)
    for mg in args.max_group:
        out_dir = os.path.join(args.out, f"03_match_g{mg}")
        shutil.rmtree(out_dir, ignore_errors=True)
        for warm in (True, False):               # first pass warms the pooled hit / pinned buffers (as a long-running stage has them)
            t0 = time.perf_counter()
            stage_q, parse_busy = qs, [0.0]
            if args.piece_mb > 0:
                # the product's way (match_stage.main): pieces parsed one after the other by one thread while the stage runs
                from concurrent.futures import ThreadPoolExecutor

                def parse_piece(p_):
                    ta = time.perf_counter()
                    x = pm.Queries(p_)
                    parse_busy[0] += time.perf_counter() - ta
                    return x
                parser = ThreadPoolExecutor(max_workers=1)
                stage_q = [parser.submit(parse_piece, p_) for p_ in MS.split_prepared_fasta(fasta, args.query_chunk, args.piece_mb << 20)]
            report, merge = MS.run_stage(pm, names, list(range(len(names))), src, stage_q, "Q", out_dir, args.threshold,
                                          args.nb_best_hits, want_merge=True, max_group=mg, write_match_files=not args.filter_only)
            if args.piece_mb > 0:
                parser.shutdown()
            t1 = time.perf_counter()
            os.makedirs(os.path.join(args.out, "04_filter"), exist_ok=True)
            fasta_bytes = merge.emit_to(os.path.join(args.out, "04_filter", f"Q_g{mg}.fa"))
            t2 = time.perf_counter()
            merge.free()
            if args.piece_mb > 0:                    # (a merge reads its query set's names: the sets go after the merges)
                for f_ in stage_q:
                    f_.result().free()
            if warm and args.queries > 200_000:
                break                            # one pass is enough at 1 M queries (the pools matter little there)
        gz = sum(os.path.getsize(os.path.join(out_dir, f)) for f in os.listdir(out_dir)) if os.path.isdir(out_dir) else 0
        line = {
            "config": f"{args.workload} shard {args.rank}/{args.world}: {len(sub)} batches, {resident / 1e9:.1f} GB resident, "
                      f"{alg_per_kmer} row bytes per k-mer; {nq} x {args.qlen} bp queries, threshold {args.threshold}, "
                      f"nb_best_hits {args.nb_best_hits}, {'clustered home batches' if args.clustered else 'i.i.d. + planted'}, "
                      f"threshold bound {'on' if args.bound else 'off'}",
            "max_group": mg, "query_chunks": report["query_chunks"], "groups": report["groups"], "scan_launches": report["scan_launches"],
            "parse_queries_s": round(parse_busy[0] if args.piece_mb > 0 else t_parse, 3), "parse_inside_stage": args.piece_mb > 0,
            "match_only_s": report["match_only_s"],
            "match_only_kmers_per_s": n_terms / report["match_only_s"],
            "match_only_algorithmic_GBps": n_terms * alg_per_kmer / report["match_only_s"] / 1e9,
            "gpu_wait_s": report["gpu_wait_s"], "d2h_s": report["d2h_s"],
            "format_s_thread_sum": report["format_s_thread_sum"], "gzip_s_thread_sum": report["gzip_s_thread_sum"],
            "merge_s_thread_sum": report["merge_s_thread_sum"],
            "stage_wall_s": round(t1 - t0, 3), "filter_emit_s": round(t2 - t1, 3),
            "e2e_s": round((0.0 if args.piece_mb > 0 else t_parse) + (t2 - t0), 3),
            "e2e_kmers_per_s": n_terms / ((0.0 if args.piece_mb > 0 else t_parse) + (t2 - t0)),
            "records": sum(g["records"] for g in report["per_group"]),
            "gz_bytes": gz, "filter_fasta_bytes": fasta_bytes, "host_cpus": len(os.sched_getaffinity(0)),
            "filter_only": args.filter_only,
        }
        print(json.dumps(line), flush=True)
        if args.json:
            with open(args.json, "a") as f:
                f.write(json.dumps(line) + "\n")


if __name__ == "__main__":
    main()

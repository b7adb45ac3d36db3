#!/usr/bin/env python3
"""BASELINE configs[3] / [4] as a WHOLE on the one GPU a builder box has (VERDICT r3 "weak" 6): all 305 batches of
batches_full.txt as 661k-shaped index FILES (rows / --rows-divisor so that the collection fits one disk and one GPU),
one query file, and the product's stage exactly as a user types it

    python -m phylign_amd.match_stage --gpus 8 --batches ... --cobs-dir ... --sizes ... --queries Q.fa \
        --out-dir 03_match --filter-out 04_filter/Q.fa

with its 8 self-launched ranks sharing the GPU (PHYLIGN_SHARE_GPU=1) over gloo -- or over RCCL with --backend nccl
when 8 GPUs are visible.  Every rank loads and searches its static share of the 305 batches, writes its batches' 03_match
files, exports what its 04_filter merge kept; ONE gather brings the exports to rank 0, which merges the 8 parts and
writes the FASTA.  The same command with --gpus 1 gives the reference result: the 305 files (after gunzip) and the FASTA
must be identical byte for byte.  Every planted query must come out of the FASTA with a non-empty match list.

    python3 tools/e2e_full_collection.py --rows-divisor 32 --queries 1000000 --work /tmp/fc --out gpurun_out/r04/full_collection.json
"""
import argparse
import gzip
import hashlib
import json
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from phylign_amd import _lib as pm  # noqa: E402
from phylign_amd import bench_aids  # noqa: E402
from phylign_amd import workload as W  # noqa: E402


def run_stage(work, tag, gpus, backend, extra):
    out_dir = os.path.join(work, f"03_{tag}")
    filt = os.path.join(work, f"04_{tag}", "Q.fa")
    cmd = [sys.executable, "-m", "phylign_amd.match_stage", "--batches", os.path.join(work, "batches.txt"),
           "--cobs-dir", os.path.join(work, "cobs"), "--sizes", os.path.join(work, "sizes.txt"),
           "--queries", os.path.join(work, "Q.fa"), "--out-dir", out_dir, "--filter-out", filt] + extra
    env = dict(os.environ, PYTHONPATH=ROOT)
    if gpus > 1:
        cmd += ["--gpus", str(gpus)]
        env["PHYLIGN_DIST_BACKEND"] = backend
        if backend != "nccl":
            env["PHYLIGN_SHARE_GPU"] = "1"
    t0 = time.perf_counter()
    r = subprocess.run(cmd, capture_output=True, env=env)
    wall = time.perf_counter() - t0
    err = r.stderr.decode(errors="replace")
    if r.returncode != 0:
        sys.exit(f"[{tag}] rc={r.returncode}\n" + err[-4000:])
    reps = [json.loads(ln) for ln in err.splitlines() if ln.startswith("{") and '"stage_wall_s"' in ln]
    reps.sort(key=lambda x: x["rank"])
    return wall, reps, out_dir, filt


def digest_dir(d):
    """{file name: (sha1 of the gunzipped bytes, gunzipped length, header lines)}"""
    out = {}
    for f in sorted(os.listdir(d)):
        with gzip.open(os.path.join(d, f), "rb") as g:
            data = g.read()
        out[f] = (hashlib.sha1(data).hexdigest(), len(data), data.count(b"\n*") + int(data[:1] == b"*"))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rows-divisor", type=int, default=32)
    ap.add_argument("--queries", type=int, default=1000000)
    ap.add_argument("--qlen", type=int, default=150)
    ap.add_argument("--backend", default="gloo")
    ap.add_argument("--work", default="/tmp/phylign_full_collection")
    ap.add_argument("--out", default=None)
    ap.add_argument("--keep", action="store_true")
    args = ap.parse_args()

    shutil.rmtree(args.work, ignore_errors=True)
    os.makedirs(os.path.join(args.work, "cobs"))
    shapes = W.scale_shapes(W.select("full"), args.rows_divisor)
    pm.init(0)
    fasta, _ = W.make_queries(args.queries, args.qlen, seed=5)
    with open(os.path.join(args.work, "Q.fa"), "wb") as f:
        f.write(fasta)
    q = pm.Queries(fasta)
    hashes = q.hash_terms(1, 1)
    every = max(1, args.queries // 3050)                       # ~10 planted queries per batch
    plan, sure = W.plant_plan(hashes, args.queries, args.qlen - 30, shapes, every=every, docs_per_query=12)
    del hashes

    t0 = time.perf_counter()
    plain_bytes = 0
    with open(os.path.join(args.work, "sizes.txt"), "w") as sz, open(os.path.join(args.work, "batches.txt"), "w") as bl:
        for i, s in enumerate(shapes):
            ix = pm.Index.synth(s.batch_id, s.n_docs, s.signature_size, 1, 31, 661)
            if i in plan:
                ix.plant(*plan[i])
            path = os.path.join(args.work, "cobs", f"{s.batch}.cobs_classic")
            bench_aids.index_save(ix, path)
            ix.free()
            n = os.path.getsize(path)
            plain_bytes += n
            sz.write(f"cobs/{s.batch}.cobs_classic.xz  {n}  1610678320\n")
            bl.write(s.batch + "\n")
    t_save = time.perf_counter() - t0
    print(f"[full_collection] {len(shapes)} batches, {plain_bytes / 1e9:.2f} GB of index files written in {t_save:.1f} s",
          file=sys.stderr, flush=True)
    q.free()
    pm.shutdown()

    rows = {}
    wall1, rep1, d1, f1 = run_stage(args.work, "one_rank", 1, args.backend, [])
    print(f"[full_collection] 1 rank: {wall1:.1f} s", file=sys.stderr, flush=True)
    wallN, repN, dN, fN = run_stage(args.work, f"{args.world}_ranks", args.world, args.backend, [])
    print(f"[full_collection] {args.world} ranks: {wallN:.1f} s", file=sys.stderr, flush=True)
    g1, gN = digest_dir(d1), digest_dir(dN)
    fa1, faN = open(f1, "rb").read(), open(fN, "rb").read()
    same_files = g1 == gN
    same_fasta = fa1 == faN
    # planted pairs: every planted query must be a record of the FASTA (its header line carries the read name)
    planted_q = set(f"q{qi:07d}".encode() for qi in range(0, args.queries, every))
    with_matches = set()
    for ln in faN.split(b"\n"):
        if ln[:1] == b">":
            name, _, com = ln[1:].partition(b" ")
            if com.strip():
                with_matches.add(name)
    found = len(planted_q & with_matches)
    unplanted_with_matches = len(with_matches - planted_q)

    def summarise(reps, wall):
        return {"process_wall_s": round(wall, 3), "ranks": len(reps),
                "batches_per_rank": [r["batches"] for r in reps],
                "stage_wall_s_per_rank": [r["stage_wall_s"] for r in reps],
                "match_only_s_per_rank": [r["match_only_s"] for r in reps],
                "e2e_s_per_rank": [r["e2e_s"] for r in reps],
                "e2e_s": max(r["e2e_s"] for r in reps),
                "filter_emit_s_rank0": reps[0].get("filter_emit_s"), "filter_phases_s_rank0": reps[0].get("filter_phases_s"),
                "scan_launches": sum(r["scan_launches"] for r in reps),
                "groups": sum(r["groups"] for r in reps)}
    rows["one_rank"] = summarise(rep1, wall1)
    rows[f"{args.world}_ranks"] = summarise(repN, wallN)
    line = {
        "config": f"ALL {len(shapes)} batches of batches_full.txt with rows / {args.rows_divisor} ({plain_bytes / 1e9:.2f} GB of "
                  f".cobs_classic files), {args.queries} x {args.qlen} bp queries, threshold 0.7, nb_best_hits 100; "
                  f"`python -m phylign_amd.match_stage --gpus {args.world}` ({args.backend}"
                  f"{', ranks share the one GPU' if args.backend != 'nccl' else ''}) against the same command with one rank",
        "match_files": len(gN), "match_files_identical_after_gunzip": bool(same_files),
        "match_text_bytes": sum(v[1] for v in gN.values()),
        "header_lines_per_file": sorted({v[2] for v in gN.values()}),
        "filter_fasta_bytes": len(faN), "filter_fasta_identical": bool(same_fasta),
        "planted_queries": len(planted_q), "planted_queries_with_matches_in_fasta": found,
        "other_queries_with_matches_in_fasta": unplanted_with_matches,
        "planted_pairs_at_or_above_threshold": sure,
        "host_cpus": len(os.sched_getaffinity(0)), "runs": rows,
    }
    print(json.dumps(line), flush=True)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(line, f, indent=1)
    if not args.keep:
        shutil.rmtree(args.work, ignore_errors=True)
    if not (same_files and same_fasta and found == len(planted_q) and len(gN) == len(shapes)):
        sys.exit("full collection: the 8-rank outputs differ from the one-rank outputs (or a planted query is missing)")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Cold / cached / resident end-to-end time of the 03_match -> 04_filter stage on 661k-SHAPED index FILES (VERDICT r3
item 5).  One rank's shard of an N-way split of the full collection is written to disk as
`<batch>.cobs_classic.xz` files (synthetic signatures generated in HBM, saved, `xz -T0`: outside every timer), then the
product's stage (phylign_amd.match_stage, as a subprocess, exactly the command a user types) runs three times:

  cold      every batch is decoded from .xz (`xzcat` pipes -> HBM) and the decode-once cache (--cache-dir) is filled
  cached    the same command again: the cache holds <batch>.cobs_classic, the parallel pread loader reads them
  resident  the matrices are in HBM already (what phylign_amd.server keeps between query sets): in-process run_stage

    python3 tools/e2e_cold_warm.py --rows-divisor 8 --queries 100000 --work /tmp/cw --out gpurun_out/r04/cold_warm.json

Synthetic Bernoulli(1/4) signatures are nearly incompressible (0.81 bits of entropy per bit), so these .xz files are
about as large as the plain ones and decode SLOWER per output byte than the real 661k indexes (which shrink ~10x): the
cold figure is a conservative one.  --rows-divisor shrinks every batch's row count (the full shard is 135 GB)."""
import argparse
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
from phylign_amd import match_stage as MS  # noqa: E402
from phylign_amd import workload as W  # noqa: E402


def stage_cmd(work, out_dir, filt, extra):
    return [sys.executable, "-m", "phylign_amd.match_stage", "--batches", os.path.join(work, "batches.txt"),
            "--cobs-dir", os.path.join(work, "cobs"), "--sizes", os.path.join(work, "sizes.txt"),
            "--queries", os.path.join(work, "Q.fa"), "--out-dir", out_dir, "--filter-out", filt] + extra


def run_stage_cmd(cmd):
    t0 = time.perf_counter()
    r = subprocess.run(cmd, capture_output=True, env=dict(os.environ, PYTHONPATH=ROOT))
    wall = time.perf_counter() - t0
    if r.returncode != 0:
        sys.exit(r.stderr.decode()[-3000:])
    rep = json.loads([ln for ln in r.stderr.decode().splitlines() if ln.startswith("{")][-1])
    return wall, rep


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--rows-divisor", type=int, default=8)
    ap.add_argument("--queries", type=int, default=100000)
    ap.add_argument("--qlen", type=int, default=150)
    ap.add_argument("--xz-level", default="-1")
    ap.add_argument("--work", default="/tmp/phylign_cold_warm")
    ap.add_argument("--out", default=None)
    ap.add_argument("--keep", action="store_true")
    args = ap.parse_args()

    shutil.rmtree(args.work, ignore_errors=True)
    os.makedirs(os.path.join(args.work, "cobs"))
    shapes = W.select("full")
    mine = W.assign_batches(shapes, args.world)[args.rank]
    sub = W.scale_shapes([shapes[p] for p in mine], args.rows_divisor)
    pm.init(0)
    fasta, _ = W.make_queries(args.queries, args.qlen, seed=5)
    with open(os.path.join(args.work, "Q.fa"), "wb") as f:
        f.write(fasta)
    q = pm.Queries(fasta)
    hashes = q.hash_terms(1, 1)
    plan, sure = W.plant_plan(hashes, args.queries, args.qlen - 30, sub, every=max(1, args.queries // 400), docs_per_query=12)
    del hashes

    # ---- files on disk (untimed): synth -> plant -> save -> xz
    t0 = time.perf_counter()
    plain_bytes = 0
    with open(os.path.join(args.work, "sizes.txt"), "w") as sz, open(os.path.join(args.work, "batches.txt"), "w") as bl:
        for i, s in enumerate(sub):
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
    t0 = time.perf_counter()
    files = [os.path.join(args.work, "cobs", f"{s.batch}.cobs_classic") for s in sub]
    subprocess.run(["xz", "-T0", args.xz_level, "--block-size=16MiB"] + files, check=True)
    t_xz = time.perf_counter() - t0
    xz_bytes = sum(os.path.getsize(f + ".xz") for f in files)
    print(f"[cold_warm] {len(sub)} batches, {plain_bytes / 1e9:.2f} GB plain, {xz_bytes / 1e9:.2f} GB as .xz "
          f"(saved in {t_save:.1f} s, xz {args.xz_level} in {t_xz:.1f} s)", file=sys.stderr, flush=True)
    q.free()
    pm.shutdown()                                            # the stage runs are processes of their own

    cache = os.path.join(args.work, "cache")
    rows = {}
    for name in ("cold", "cached"):
        out_dir = os.path.join(args.work, f"03_{name}")
        wall, rep = run_stage_cmd(stage_cmd(args.work, out_dir, os.path.join(args.work, f"04_{name}", "Q.fa"), ["--cache-dir", cache]))
        rows[name] = {"process_wall_s": round(wall, 3), "e2e_s": rep["e2e_s"], "stage_wall_s": rep["stage_wall_s"],
                      "match_only_s": rep["match_only_s"], "load_s_thread_sum": rep["load_s_thread_sum"],
                      "index_source": rep["index_source"], "host_ram_plan": rep["host_ram_plan"], "groups": rep["groups"]}
    same = all(open(os.path.join(args.work, "03_cold", f), "rb").read() == open(os.path.join(args.work, "03_cached", f), "rb").read()
               for f in os.listdir(os.path.join(args.work, "03_cold")))
    same = same and open(os.path.join(args.work, "04_cold", "Q.fa"), "rb").read() == open(os.path.join(args.work, "04_cached", "Q.fa"), "rb").read()

    # ---- resident: the matrices stay in HBM between query sets (server); in-process, load time outside the timer
    pm.init(0)
    t0 = time.perf_counter()
    ixs = {s.batch: pm.Index.load_file(os.path.join(cache, f"{s.batch}.cobs_classic")) for s in sub}
    t_load = time.perf_counter() - t0
    names = sorted(ixs)
    src = MS.ResidentSource(ixs)
    res_rows = []
    for _ in range(2):                                       # first pass warms the pooled buffers
        t0 = time.perf_counter()
        qq = pm.Queries(fasta)
        rep, merge = MS.run_stage(pm, names, list(range(len(names))), src, qq, "Q", os.path.join(args.work, "03_resident"), 0.7, 100,
                                  want_merge=True)
        os.makedirs(os.path.join(args.work, "04_resident"), exist_ok=True)
        merge.emit_to(os.path.join(args.work, "04_resident", "Q.fa"))
        res_rows.append(time.perf_counter() - t0)
        merge.free()
        qq.free()
    same = same and all(open(os.path.join(args.work, "03_cold", f), "rb").read() == open(os.path.join(args.work, "03_resident", f), "rb").read()
                        for f in os.listdir(os.path.join(args.work, "03_cold")))
    same = same and open(os.path.join(args.work, "04_cold", "Q.fa"), "rb").read() == open(os.path.join(args.work, "04_resident", "Q.fa"), "rb").read()
    rows["resident"] = {"e2e_s": round(res_rows[-1], 3), "match_only_s": rep["match_only_s"],
                        "plain_files_to_hbm_s": round(t_load, 3), "plain_files_to_hbm_GBps": plain_bytes / t_load / 1e9}
    line = {
        "config": f"rank {args.rank} of {args.world} of batches_full.txt with rows / {args.rows_divisor}: {len(sub)} batches, "
                  f"{plain_bytes / 1e9:.2f} GB of index files ({xz_bytes / 1e9:.2f} GB as .xz), {args.queries} x {args.qlen} bp queries, "
                  f"threshold 0.7, nb_best_hits 100, 03_match files + 04_filter FASTA",
        "cold_s": rows["cold"]["e2e_s"], "cached_s": rows["cached"]["e2e_s"], "resident_s": rows["resident"]["e2e_s"],
        "cold_decode_GBps": plain_bytes / rows["cold"]["e2e_s"] / 1e9, "cached_load_GBps": plain_bytes / rows["cached"]["e2e_s"] / 1e9,
        "outputs_identical": bool(same), "planted_pairs_at_or_above_threshold": sure,
        "host_cpus": len(os.sched_getaffinity(0)), "runs": rows,
        "note": "synthetic Bernoulli(1/4) signatures barely compress: the .xz decode is slower per output byte than on the real "
                "661k indexes; cold_s is conservative",
    }
    print(json.dumps(line), flush=True)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(line, f, indent=1)
    if not args.keep:
        shutil.rmtree(args.work, ignore_errors=True)


if __name__ == "__main__":
    main()

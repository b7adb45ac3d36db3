#!/usr/bin/env python3
"""Cold / cached / resident end-to-end time of the 03_match -> 04_filter stage on 661k-SHAPED index FILES (VERDICT r3
item 5).  One rank's shard of an N-way split of the full collection is written to disk as
`<batch>.cobs_classic.xz` files (synthetic signatures generated in HBM, saved, `xz -T0`: outside every timer), then the
product's stage (phylign_amd.match_stage, as a subprocess, exactly the command a user types) runs three times:

  cold      every batch is decoded from .xz (`xzcat` pipes -> HBM) and the decode-once cache (--cache-dir) is filled
  cached    the same command again: the cache holds <batch>.cobs_classic, the parallel pread loader reads them
  resident  the matrices are in HBM already (what phylign_amd.server keeps between query sets): in-process run_stage

    python3 tools/e2e_cold_warm.py --rows-divisor 8 --queries 100000 --work /tmp/cw --out gpurun_out/r04/cold_warm.json
    python3 tools/e2e_cold_warm.py --rows-divisor 1 --modes plain,cached,resident --work /dev/shm/pf ...     (full size, no .xz)
    python3 tools/e2e_cold_warm.py --rows-divisor 1 --modes cold --cold-without-cache --work /dev/shm/pcf ... (full size, xz -> HBM)
    COLD_ALSO_XZCAT=1 python3 tools/e2e_cold_warm.py --workload small --rows-divisor 1 --xz-block-mib 24 ... (3 batches: block-parallel vs xzcat)

  plain     the decompressed files in --cobs-dir (rule decompress_cobs), no .xz involved

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


STAGE_TIMEOUT_S = float(os.environ.get("STAGE_TIMEOUT_S", "0")) or None


def run_stage_cmd(cmd):
    t0 = time.perf_counter()
    try:
        r = subprocess.run(cmd, capture_output=True, env=dict(os.environ, PYTHONPATH=ROOT), timeout=STAGE_TIMEOUT_S)
    except subprocess.TimeoutExpired as e:
        sys.exit(f"stage did not finish within {STAGE_TIMEOUT_S} s; stderr tail:\n" + (e.stderr or b"").decode(errors="replace")[-3000:])
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
    ap.add_argument("--workload", default="full", help="full (a rank's shard of batches_full.txt) or small (the three batches of "
                                                       "data/batches_small.txt at full size: fewer batches than CPUs, --world 1)")
    ap.add_argument("--xz-block-mib", type=int, default=16)
    ap.add_argument("--flip-log2", type=int, default=7, help="--compressible: a bit differs from its neighbour's with probability 2^-n")
    ap.add_argument("--compressible", action="store_true",
                    help="files with compressible content (bench_aids.index_correlate: runs of equal bits along a row) instead of "
                         "Bernoulli(1/4) bits: the .xz then shrinks and decodes like a real index does; no planted hits (outputs are "
                         "still compared between the modes)")
    ap.add_argument("--cold-without-cache", action="store_true",
                    help="the cold run decodes straight into HBM and keeps nothing (the reference's default mem-stream mode): the "
                         "full-size shard as .xz (112 GB) plus a 128 GB cache would not fit a memory-backed --work")
    ap.add_argument("--modes", default="cold,cached,resident",
                    help="comma list of cold (xz -> HBM, fills the cache), cached (second run on the decode-once cache), plain "
                         "(decompressed .cobs_classic files in --cobs-dir: rule decompress_cobs / mem-disk, no .xz involved), resident.  "
                         "Without `cold` no .xz is made and the cache directory is filled by MOVING the plain files into it "
                         "(what a cold run leaves there): the full-size shard (--rows-divisor 1, 128 GB) fits a memory-backed --work "
                         "only once")
    args = ap.parse_args()
    modes = [m for m in args.modes.split(",") if m]
    assert set(modes) <= {"cold", "cached", "plain", "resident"} and modes

    shutil.rmtree(args.work, ignore_errors=True)
    os.makedirs(os.path.join(args.work, "cobs"))
    shapes = W.select(args.workload)
    if args.workload != "full":
        args.world, args.rank = 1, 0
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
            if args.compressible:
                bench_aids.index_correlate(ix, seed=7 + i, flip_log2=args.flip_log2)
            elif i in plan:
                ix.plant(*plan[i])
            path = os.path.join(args.work, "cobs", f"{s.batch}.cobs_classic")
            bench_aids.index_save(ix, path)
            ix.free()
            n = os.path.getsize(path)
            plain_bytes += n
            sz.write(f"cobs/{s.batch}.cobs_classic.xz  {n}  1610678320\n")
            bl.write(s.batch + "\n")
    t_save = time.perf_counter() - t0
    print(f"[cold_warm] {len(sub)} index files ({plain_bytes / 1e9:.2f} GB) saved in {t_save:.1f} s", file=sys.stderr, flush=True)
    files = [os.path.join(args.work, "cobs", f"{s.batch}.cobs_classic") for s in sub]
    q.free()
    pm.shutdown()                                            # the stage runs are processes of their own
    cache = os.path.join(args.work, "cache")
    rows = {}

    def stage_row(name, extra):
        out_dir = os.path.join(args.work, f"03_{name}")
        wall, rep = run_stage_cmd(stage_cmd(args.work, out_dir, os.path.join(args.work, f"04_{name}", "Q.fa"), extra))
        rows[name] = {"process_wall_s": round(wall, 3), "e2e_s": rep["e2e_s"], "stage_wall_s": rep["stage_wall_s"],
                      "match_only_s": rep["match_only_s"], "load_s_thread_sum": rep["load_s_thread_sum"],
                      "index_source": rep["index_source"], "host_ram_plan": rep["host_ram_plan"], "groups": rep["groups"],
                      "index_GBps_over_e2e": plain_bytes / rep["e2e_s"] / 1e9, "index_GBps_over_process_wall": plain_bytes / wall / 1e9}
    if "plain" in modes:                                     # decompressed files in --cobs-dir: before any .xz exists
        stage_row("plain", [])
    xz_bytes, t_xz = 0, 0.0
    if "cold" in modes:
        t0 = time.perf_counter()
        # (-T = the CPUs the job may use, not -T0: with a cgroup quota of 16 on a 256-thread host, 256 xz threads thrash)
        from phylign_amd.sysinfo import effective_cpus
        subprocess.run(["xz", f"-T{effective_cpus()}", args.xz_level, f"--block-size={args.xz_block_mib}MiB"] + files, check=True)
        t_xz = time.perf_counter() - t0
        xz_bytes = sum(os.path.getsize(f + ".xz") for f in files)
        print(f"[cold_warm] xz {args.xz_level} of {plain_bytes / 1e9:.2f} GB took {t_xz:.1f} s -> {xz_bytes / 1e9:.2f} GB", file=sys.stderr, flush=True)
        # the block structure the decoder would see (`xz --list`): blocks per file decide whether a file could be decoded
        # by several threads at all
        lst = subprocess.run(["xz", "--robot", "--list"] + [f + ".xz" for f in files], capture_output=True, text=True).stdout
        blocks = [int(ln.split("\t")[2]) for ln in lst.splitlines() if ln.startswith("file\t")]
        rows["xz_list"] = {"files": len(blocks), "blocks_min": min(blocks), "blocks_max": max(blocks), "blocks_total": sum(blocks)}
        stage_row("cold", [] if args.cold_without_cache else ["--cache-dir", cache])
        if os.environ.get("COLD_ALSO_XZCAT"):
            # the same cold run with one xzcat per file (PHYLIGN_XZ_THREADS=1), cache emptied first: what the
            # block-parallel decoder buys when a rank has fewer compressed batches than CPUs
            shutil.rmtree(cache, ignore_errors=True)
            os.environ["PHYLIGN_XZ_THREADS"] = "1"
            stage_row("cold_xzcat_only", ["--cache-dir", cache])
            del os.environ["PHYLIGN_XZ_THREADS"]
    else:
        os.makedirs(cache, exist_ok=True)
        for f in files:                                      # what a cold run leaves in the cache: the decoded files
            os.rename(f, os.path.join(cache, os.path.basename(f)))
    print(f"[cold_warm] {len(sub)} batches, {plain_bytes / 1e9:.2f} GB plain, {xz_bytes / 1e9:.2f} GB as .xz "
          f"(saved in {t_save:.1f} s = {plain_bytes / t_save / 1e9:.2f} GB/s, xz {args.xz_level} in {t_xz:.1f} s)", file=sys.stderr, flush=True)
    if "cached" in modes:
        stage_row("cached", ["--cache-dir", cache])
    first = [m for m in ("plain", "cold", "cached") if m in rows][0]


    def same_as_first(name):
        a, b = os.path.join(args.work, f"03_{first}"), os.path.join(args.work, f"03_{name}")
        ok_ = all(open(os.path.join(a, f), "rb").read() == open(os.path.join(b, f), "rb").read() for f in os.listdir(a))
        return ok_ and open(os.path.join(args.work, f"04_{first}", "Q.fa"), "rb").read() == open(os.path.join(args.work, f"04_{name}", "Q.fa"), "rb").read()
    same = all(same_as_first(m) for m in ("plain", "cold", "cold_xzcat_only", "cached") if m in rows and m != first)

    # ---- resident: the matrices stay in HBM between query sets (server); in-process, load time outside the timer
    if "resident" in modes:
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
        same = same and same_as_first("resident")
        rows["resident"] = {"e2e_s": round(res_rows[-1], 3), "match_only_s": rep["match_only_s"],
                            "plain_files_to_hbm_s_one_thread_of_loads": round(t_load, 3), "plain_files_to_hbm_GBps": plain_bytes / t_load / 1e9}
    line = {
        "config": f"rank {args.rank} of {args.world} of batches_{args.workload}.txt with rows / {args.rows_divisor}: {len(sub)} batches, "
                  f"{plain_bytes / 1e9:.2f} GB of index files ({xz_bytes / 1e9:.2f} GB as .xz), {args.queries} x {args.qlen} bp queries, "
                  f"threshold 0.7, nb_best_hits 100, 03_match files + 04_filter FASTA; files under {args.work}",
        "index_GB": plain_bytes / 1e9,
        "outputs_identical": bool(same), "planted_pairs_at_or_above_threshold": sure,
        "host_cpus": len(os.sched_getaffinity(0)), "runs": rows,
        "note": ("compressible content (--compressible): runs of equal bits along a row; xz ratio and decode speed in the range of real indexes"
                 if args.compressible else
                 "synthetic Bernoulli(1/4) signatures barely compress: the .xz decode is slower per output byte than on the real "
                 "661k indexes; a cold figure is conservative"),
    }
    for m in ("plain", "cold", "cold_xzcat_only", "cached", "resident"):
        if m in rows:
            line[m + "_s"] = rows[m]["e2e_s"]
    print(json.dumps(line), flush=True)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(line, f, indent=1)
    if not args.keep:
        shutil.rmtree(args.work, ignore_errors=True)


if __name__ == "__main__":
    main()

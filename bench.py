#!/usr/bin/env python3
"""bench.py -- k-mers matched/sec of the MI355X COBS matching stage against a
661k-shaped synthetic index set, with the HBM roofline of the scan kernel and
the CPU (oracle "port") baseline timed beside it.

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: starts its own N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W

One step = one pass of the whole hot path (canonicalise + XXH64, row map, row
gather + bit-sliced count, threshold + on-device ordering and compaction, gather of
the hit records to rank 0) of the query batch over every resident phylogenetic
batch index.  Workload (BASELINE.json configs[2], SURVEY.md 8d "config 3"): the
64 661k-shaped batches that fit one GPU (~213 GB of signatures), 100 000
synthetic 150-bp queries (120 31-mers each), threshold 0.7.  With N > 1 the same
64 batches are sharded statically over the ranks (strong scaling).

Output: the LAST stdout line is one small strict-JSON object (compact_line: the contract's
keys + roofline + roofline_narrow + cpu_baseline + participants; < 6 KB by construction);
the WHOLE record -- that line's content plus every auxiliary leg below -- is written to the
side file --legs-out (default ./bench_legs.json).  roofline.traffic is measured in the run:
after the timed legs rank 0 runs two `rocprofv3 --kernel-trace --pmc` child passes over one
untimed step of its own launch (live_pmc_traffic); --no-live-pmc falls back to the table
committed under profiles/.

What is reported where:
  value / roofline     every signature row of every k-mer is gathered, like `cobs
                       query` does ("fetch_all_rows"): independent of the data, and
                       the algorithmic bytes are the bytes the kernel really moves.
  threshold_bound      (side file) the product default: lines whose documents cannot reach the
                       threshold any more are not fetched (identical results).  Its
                       speed depends on the data, so it is a secondary figure, and
                       its roofline uses the bytes really gathered (counted in-kernel).
  clustered            (side file) the same two modes after every query was given a HOME batch in
                       which about half of the documents match it at 0.6-1.0 of its
                       k-mers (what phylogenetic batches look like for reads of their
                       own species): many documents near the threshold, long hit lists.
Steps are software-pipelined: the kernels of step i+1 are queued before the host
orders / gathers the records of step i (pm_search_async); all of it is inside the
timed region.
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


from phylign_amd.sysinfo import effective_cpus  # noqa: E402


from phylign_amd.sysinfo import available_ram_gb as host_memory_gb  # noqa: E402


def flush_c_stdio():
    """fflush(NULL): whatever native libraries (RCCL's banner) left in C stdio buffers goes out now"""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def kernel_blob_hash():
    """git blob id of the kernel source: PMC traffic figures are only valid for the code they were measured on"""
    import hashlib
    data = open(os.path.join(ROOT, "phylign_amd", "csrc", "pm_kernels.hip"), "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


# ---- the line the driver parses -------------------------------------------------------------------------------------
# The LAST stdout line of bench.py is a small strict-JSON object (headline + roofline + cpu_baseline + who took part); the
# auxiliary legs (threshold_bound, unique_rows, argannot, clustered, l31, full_shard, full_collection, per-launch and
# per-rank detail) go to a side file (--legs-out) that holds the whole record.  Round 5's single 28.6 KB line came back
# from the driver unparsed (BENCH_r05.json: "parsed": null); the line is therefore capped and the cap is tested.
LINE_MAX_BYTES = 6144
LINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
             "vs_baseline", "dtype", "data", "config", "roofline", "roofline_narrow", "cpu_baseline", "gpu_over_cpu",
             "participants", "pm_kernels_blob", "hits", "legs_file")
_ROOFLINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "launches_per_step", "batches_per_launch",
                  "avg_launch_ms", "algorithmic_bytes_per_launch", "hbm_GBps_from_traffic", "wire_frac_of_peak", "traffic_source",
                  "traffic_committed")
_CPU_KEYS = ("value", "unit", "cores", "kind", "sample", "partition", "runs", "sample_GB", "algorithmic_GBps", "timed_while")
_CONFIG_KEYS = ("workload", "batches", "queries", "query_len", "k", "num_hashes", "threshold", "nb_best_hits", "rows_divisor",
                "sharding", "scan_mode", "pipeline_depth")


def _clip(text, n):
    return text if text is None or len(text) <= n else text[:n - 3] + "..."


def _sig(x, digits=7):
    """floats of the line carry 7 significant digits (the legs file keeps full precision)"""
    if isinstance(x, float):
        return float(f"{x:.{digits}g}")
    if isinstance(x, dict):
        return {k: _sig(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, digits) for v in x]
    return x


def compact_line(full, legs_file=None):
    """the driver-facing line of a full bench record: the contract's keys + roofline + roofline_narrow + cpu_baseline +
    condensed participants, nothing else.  Pure function of the record (tests/test_golden_cpu.py runs it on recorded ones)."""
    def pick(d, keys):
        return None if d is None else {k: d[k] for k in keys if k in d}
    line = {k: full.get(k) for k in LINE_KEYS if k in full}
    cfg = pick(full.get("config"), _CONFIG_KEYS) or {}
    cfg["workload"] = _clip(cfg.get("workload"), 240)
    cfg["scan_mode"] = (full.get("scan_mode") or "").split(":")[0] or None
    two = (full.get("config") or {}).get("batches_on_two_ranks")
    if two:
        cfg["batches_on_two_ranks"] = len(two)
    line["config"] = cfg
    for k in ("roofline", "roofline_narrow"):
        r = pick(full.get(k), _ROOFLINE_KEYS)
        if r is not None and r.get("traffic") is None and (full.get(k) or {}).get("traffic_note"):
            r["traffic_note"] = _clip(full[k]["traffic_note"], 200)
        if r is not None and r.get("traffic_source"):
            r["traffic_source"] = _clip(r["traffic_source"], 200)
        line[k] = r
    cb = pick(full.get("cpu_baseline"), _CPU_KEYS)
    if cb is not None:
        cb["sample"] = _clip(cb.get("sample"), 320)
    line["cpu_baseline"] = cb
    line["gpu_over_cpu"] = full.get("gpu_over_cpu")
    p = full.get("participants") or {}
    line["participants"] = {"ranks": p.get("ranks", len(p.get("rank_ms_per_step", [])) or 1), "backend": p.get("backend"),
                            "rccl_ranks": p.get("rccl_ranks"),
                            "rank_ms_per_step": [round(v, 3) for v in p.get("rank_ms_per_step", [])],
                            "rank_devices": p.get("rank_devices"), "rank_batches": p.get("rank_batches")}
    line["legs_file"] = legs_file
    line = _sig(line)
    line["value"] = full.get("value")                       # the headline keeps every digit
    line["ms_per_step"] = full.get("ms_per_step")
    return line


def emit(full, legs_path, log=None):
    """writes the whole record to `legs_path` (atomically, BEFORE the line), then prints the compact line as the last thing
    on stdout.  A legs file that cannot be written costs the side file, never the line."""
    written = None
    if legs_path:
        tmp = f"{legs_path}.{os.getpid()}.tmp"
        try:
            with open(tmp, "w") as f:
                json.dump(full, f, allow_nan=False)
                f.write("\n")
            os.replace(tmp, legs_path)
            written = os.path.abspath(legs_path)
        except (OSError, ValueError) as e:
            try:
                os.unlink(tmp)
            except OSError:
                pass
            if log:
                log(f"[bench] legs file {legs_path!r} not written: {e!r}")
    text = json.dumps(compact_line(full, written), allow_nan=False)
    if len(text) > LINE_MAX_BYTES:                          # cannot happen with the clipped fields; never print an oversized line
        raise RuntimeError(f"bench line is {len(text)} bytes (cap {LINE_MAX_BYTES})")
    flush_c_stdio()
    sys.stdout.write(text + "\n")
    sys.stdout.flush()
    return text


# ---- HBM traffic of the scan kernels, measured in THIS run ------------------------------------------------------------
# `roofline.traffic` used to be a constant committed under profiles/ (keyed to the kernel's git blob).  With one GPU the
# bench now collects it itself once its own GPU work is done and its matrices are freed: two `rocprofv3 --kernel-trace
# --pmc` child runs (one counter group each, as MI355X_MICROARCH.md prescribes: TCC slots do not hold both) of ONE
# untimed step of the same workload, bytes = 128 x RDREQ_128B + 64 x RDREQ_64B + 32 x RDREQ_32B + 1024 x WRITE_SIZE per
# launch (request sizes counted, not assumed: FETCH_SIZE tallies a 128-byte request at 64 on gfx950).  Any failure --
# no rocprofv3, a pass that times out -- falls back to the committed table and says so.
PMC_PASSES = (("rdreq", ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum")),
              ("write", ("WRITE_SIZE",)))


def scan_kernel_name(raw):
    """rocprofv3's `void pm::k_scan<32, 7, true, false>(pm::ScanArgs)` -> the name pm_launch_t / bench.py use"""
    import re
    m = re.search(r"k_scan<(\d+), (\d+), (true|false), (true|false)>", raw)
    if not m:
        return None
    g, p_, nh1, wq = m.groups()
    return f"k_scan<G={'mixed' if g == '0' else g},P={p_},{'NH1' if nh1 == 'true' else 'NHn'}{',WQ' if wq == 'true' else ''}>"


def pmc_bytes_per_launch(counter_csvs):
    """{kernel: {"hbm_bytes_per_launch", "read_requests", "write_bytes", "launches"}} from rocprofv3 counter_collection CSVs"""
    import csv
    acc = {}                                             # kernel -> counter -> [dispatches, sum]
    for path in counter_csvs:
        with open(path) as fh:
            for row in csv.DictReader(fh):
                name = scan_kernel_name(row["Kernel_Name"])
                if name:
                    e = acc.setdefault(name, {}).setdefault(row["Counter_Name"], [0, 0.0])
                    e[0] += 1; e[1] += float(row["Counter_Value"])
    out = {}
    for name, c in acc.items():
        if "TCC_EA0_RDREQ_128B_sum" not in c or "WRITE_SIZE" not in c:
            continue
        mean = {k: v[1] / v[0] for k, v in c.items()}
        rd = 128 * mean["TCC_EA0_RDREQ_128B_sum"] + 64 * mean.get("TCC_EA0_RDREQ_64B_sum", 0.0) + 32 * mean.get("TCC_EA0_RDREQ_32B_sum", 0.0)
        wr = 1024 * mean["WRITE_SIZE"]
        out[name] = {"hbm_bytes_per_launch": int(rd + wr), "write_bytes": int(wr), "launches": c["TCC_EA0_RDREQ_128B_sum"][0],
                     "read_requests": {k: int(mean.get(k, 0)) for k in PMC_PASSES[0][1]}}
    return out


def under_a_profiler():
    return any(k.startswith(("ROCPROF", "ROCP_", "ROCTX")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")


def live_pmc_traffic(workload_argv, budget_s, log, min_pass_s=20.0):
    """(per-kernel table or None, why-not or None, seconds): the PMC passes of one untimed step, as child processes"""
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile
    t0 = time.time()
    exe = shutil.which("rocprofv3") or next((p_ for p_ in ("/opt/rocm/bin/rocprofv3",) if os.path.exists(p_)), None)
    if not exe:
        return None, "rocprofv3 is not installed on this host", 0.0
    if under_a_profiler():
        return None, "this run is itself being profiled", 0.0
    work = tempfile.mkdtemp(prefix="pm_bench_pmc_", dir="/tmp")
    csvs = []
    try:
        for tag, counters in PMC_PASSES:
            left = budget_s - (time.time() - t0)
            if left < min_pass_s:
                return None, f"time budget of {budget_s:.0f} s used up before the {tag} pass", time.time() - t0
            cmd = [exe, "--kernel-trace", "--pmc", *counters, "--output-format", "csv", "-d", os.path.join(work, tag), "-o", "pmc", "--",
                   sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
                   "--only-headline", "--no-pipeline", "--no-live-pmc", "--legs-out", ""] + workload_argv
            env = dict(os.environ, TMPDIR="/tmp", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
            # the child is a plain one-process run: nothing of this rank's launcher / process group may reach it
            # (PHYLIGN_LAUNCHER_PID would make it end itself: its parent is the profiler, not the launcher)
            for k in list(env):
                if k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "GROUP_WORLD_SIZE", "ROLE_RANK", "ROLE_WORLD_SIZE",
                         "ROLE_NAME", "MASTER_ADDR", "MASTER_PORT", "PHYLIGN_LAUNCHER_PID", "BENCH_FORCE_DIST", "BENCH_FORCE_GATHER",
                         "BENCH_DIST_BACKEND", "BENCH_FULL_MIN_WORLD") or k.startswith(("TORCHELASTIC_", "TORCH_NCCL_", "NCCL_ASYNC")):
                    env.pop(k)
            p = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, start_new_session=True)
            try:
                _, err = p.communicate(timeout=left)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(p.pid, signal.SIGKILL)          # the session this call started, nothing else
                except OSError:
                    pass
                p.communicate()
                return None, f"the {tag} pass did not finish within {left:.0f} s", time.time() - t0
            if p.returncode != 0:
                return None, f"the {tag} pass exited with status {p.returncode}: {err.decode(errors='replace')[-160:]!r}", time.time() - t0
            found = glob.glob(os.path.join(work, tag, "**", "*counter_collection.csv"), recursive=True)
            if not found:
                return None, f"the {tag} pass wrote no counter_collection.csv", time.time() - t0
            csvs += found
        table = pmc_bytes_per_launch(csvs)
        if not table:
            return None, "no k_scan dispatch in the counter files", time.time() - t0
        return table, None, time.time() - t0
    except Exception as e:                                   # noqa: BLE001 -- an optional measurement never costs the line
        return None, f"{e!r}", time.time() - t0
    finally:
        shutil.rmtree(work, ignore_errors=True)


def cpu_baseline(shapes, fasta_seqs, qlen, threshold, target_s, sample_gb, log):
    """Times oracle/cobs_oracle.c (kind "port": a restatement of the cobs classic search, NOT bioconda
    cobs 0.2.1) on the host cores, as SURVEY.md 8d / BASELINE.md section 2 specify it: index fully in
    RAM (`--load-complete`), the largest row-scaled copy of the SAME batch shapes (same document counts =
    same algorithmic bytes per k-mer) that fits the host-memory budget -- tens of GB, DRAM-resident, not
    cache-resident --, -T = the cores the job may use, median of 3 runs, in both work partitions:
    `queries` (threads split the queries) and `column_slabs` (cobs -T: hashes once per query, threads take
    64-byte column slabs of the rows).  `value` is the faster of the two."""
    from oracle import oracle as O
    from phylign_amd import workload as W
    cores = effective_cpus()
    total = sum(s.index_bytes for s in shapes)
    mem = host_memory_gb()
    budget = sample_gb if sample_gb > 0 else min(48.0, 0.35 * mem)
    div = max(1, int(np.ceil(total / (budget * 1e9))))
    small = W.scale_shapes(shapes, div)
    t0 = time.time()
    mats = [(s, O.synth_fill(661, s.batch_id, s.signature_size, s.n_docs, cores)) for s in small]
    t_gen = time.time() - t0
    sample_bytes = sum(m.nbytes for _, m in mats)
    hdrs = []
    for s, _ in mats:
        h = O.Header()
        h.term_size, h.canonicalize, h.num_hashes = 31, 1, 1
        h.n_docs, h.signature_size, h.row_bytes = s.n_docs, s.signature_size, s.row_bytes
        hdrs.append(h)

    def run(nq, slabs):
        seqs = fasta_seqs[:nq].tobytes()
        t = time.time()
        hits = 0
        for (s, m), h in zip(mats, hdrs):
            if slabs:
                hits += O.baseline_run_slabs(m, s.row_bytes, h, seqs, qlen, nq, threshold, cores, 64)
            else:
                hits += O.baseline_run(m, s.row_bytes, h, seqs, qlen, nq, threshold, cores)
        return time.time() - t, hits
    per_run = target_s / 6.0            # 2 partitions x 3 runs
    nq = min(64 * cores, len(fasta_seqs))
    tt, _ = run(nq, False)
    for _ in range(4):                  # grow the query sample until one run costs about per_run seconds
        if tt >= 0.6 * per_run or nq >= len(fasta_seqs):
            break
        nq = int(min(len(fasta_seqs), max(nq * 2, nq * per_run / max(tt, 1e-6))))
        tt, _ = run(nq, False)
    runs = {"queries": [], "column_slabs": []}
    hits = {}
    for _ in range(3):
        for name, slabs in (("queries", False), ("column_slabs", True)):
            t, hcount = run(nq, slabs)
            runs[name].append(t)
            hits[name] = hcount
    assert hits["queries"] == hits["column_slabs"], "the two CPU partitions disagree"
    terms = nq * (qlen - 30)
    alg = terms * sum(s.row_bytes for s in shapes)
    med = {k: float(np.median(v)) for k, v in runs.items()}
    best = min(med, key=med.get)
    log(f"[cpu_baseline] {sample_bytes / 1e9:.1f} GB sample generated in {t_gen:.1f}s, {nq} queries, medians "
        + ", ".join(f"{k} {v:.2f}s" for k, v in med.items()) + f" on {cores} threads")
    return {
        "value": terms / med[best], "unit": "k-mers/s", "cores": cores, "kind": "port", "runs": 3, "partition": best,
        "partitions": {k: {"k-mers/s": terms / v, "algorithmic_GBps": alg / v / 1e9, "run_s": [round(x, 3) for x in runs[k]]}
                       for k, v in med.items()},
        "host": f"{os.cpu_count()} logical CPUs visible, {cores} usable (affinity / cgroup quota): {cores} threads; "
                f"{mem:.0f} GB of host RAM available to the job",
        "sample_GB": sample_bytes / 1e9,
        "sample": (f"{nq} of the same queries x all {len(shapes)} batch shapes at rows/{div}: {sample_bytes / 1e9:.1f} GB of signatures in "
                   f"host RAM (DRAM-resident, like --load-complete), median of 3 runs x 2 partitions, "
                   f"{sum(sum(v) for v in runs.values()):.0f} s of timed CPU work; oracle/cobs_oracle.c (restatement, not bioconda cobs 0.2.1)"),
        "sample_vs_cache": f"the matrices are {sample_bytes / 2.56e8 / 2:.0f} x the host's 2 x 256 MB of L3",
        "algorithmic_GBps": alg / med[best] / 1e9,
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
    ap.add_argument("--headline", default="fetch_all_rows", choices=["fetch_all_rows", "threshold_bound"],
                    help="scan mode of `value`/`roofline` (default: every row fetched, like cobs; the other mode is "
                         "reported beside it)")
    ap.add_argument("--only-headline", action="store_true",
                    help="skip the other scan mode, the clustered variant and the fetched-bytes pass (profiling runs)")
    ap.add_argument("--no-clustered", action="store_true", help="skip the clustered (home batch) variant")
    ap.add_argument("--no-l31", action="store_true", help="skip the 31-bp (one k-mer per query) legs")
    ap.add_argument("--no-argannot", action="store_true",
                    help="skip the gene-length leg (SURVEY.md 8d: the length mix of the reference's data/ARGannot_r3.fa)")
    ap.add_argument("--no-unique-rows", action="store_true",
                    help="skip the untimed pass that counts the distinct signature rows the query set touches (SURVEY.md 8d)")
    ap.add_argument("--no-full-shard", action="store_true",
                    help="skip the configs[3]/[4] leg (one rank's shard of the full 305-batch collection, generated after the headline set is freed)")
    ap.add_argument("--full-shard-world", type=int, default=8)
    ap.add_argument("--full-shard-rank", type=int, default=0)
    ap.add_argument("--clustered-multi", action="store_true",
                    help="run the clustered variant with N > 1 too (it ships ~0.5 GB of records per step to rank 0; "
                         "by default it is a single-GPU figure)")
    ap.add_argument("--replicas", action="store_true",
                    help="N > 1: level the ranks' scan work by making a few small batches resident on two ranks that share their "
                         "queries (workload.assign_parts / pm_search_async_parts).  Off by default: on the 64 batches of config 3 "
                         "whole batches already balance 8 ranks to within the run-to-run noise (profiles/r04/NOTES.md)")
    ap.add_argument("--no-pipeline", action="store_true", help="finish every step before queueing the next one")
    ap.add_argument("--pipeline-depth", type=int, default=0,
                    help="steps queued ahead of the one the host is finishing (0 = 1 on one GPU, 2 with N > 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-target-s", type=float, default=16.0,
                    help="sizes the CPU baseline's query sample: 6 runs (2 partitions x 3) of about target / 6 s each by the faster "
                         "partition; the column-slab partition runs ~2.5 x longer, so 16 gives about 25 s of timed CPU work")
    ap.add_argument("--cpu-sample-gb", type=float, default=0.0,
                    help="host-RAM size of the CPU baseline's index sample (0 = min(48 GB, 35 %% of the RAM available to the job))")
    ap.add_argument("--legs-out", default="bench_legs.json",
                    help="side file for the WHOLE record (the stdout line + every auxiliary leg: threshold_bound, unique_rows, "
                         "argannot, clustered, l31, full_shard, full_collection, per-launch / per-rank detail); '' = none")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="do not collect the scan kernels' HBM traffic with rocprofv3 --pmc child runs at the end of a 1-GPU run "
                         "(`roofline.traffic` then comes from the committed profiles/pmc_traffic.json, if that matches the kernel source)")
    ap.add_argument("--live-pmc-budget-s", type=float, default=240.0, help="wall-clock cap of those child runs, both passes together")
    ap.add_argument("--whole-record", action="store_true",
                    help="developer tooling (tools/*.sh): print the WHOLE record as the stdout line instead of the compact one; "
                         "never what the driver runs -- the default line is capped at LINE_MAX_BYTES")
    ap.add_argument("--dump-hits", default=None, help="rank 0 saves the ordered hit records of the headline mode (.npy)")
    ap.add_argument("--dump-full-hits", default=None, help="rank 0 saves the records of the full_collection leg (N >= 8 ranks) (.npy)")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="single process: hold only the shard rank --emulate-rank would get in an N-way split "
                         "(estimates the per-rank step time of a strong-scaling run; not a reported number)")
    ap.add_argument("--emulate-rank", type=int, default=0)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    from phylign_amd import launch
    if launch.wants_self_launch(args.gpus):
        # plain `python bench.py --gpus N`: this process has not touched the GPU (torch is not even imported yet), so it
        # starts N fresh ranks of the same command line, relays their output (rank 0 prints the JSON line) and their status
        sys.exit(launch.self_launch_script(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    launch.arm_parent_death_signal()          # a rank of that launcher: ends with it
    if world != args.gpus:
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
    if backend == "nccl" and world > 1 and not os.environ.get("BENCH_SHARE_GPU") and 1 < torch.cuda.device_count() < world:
        sys.exit(f"bench.py --gpus {world}: only {torch.cuda.device_count()} GPUs are visible (RCCL needs one device per rank)")
    if os.environ.get("BENCH_SHARE_GPU") or local_rank >= torch.cuda.device_count():
        # functional check on one GPU, or a launcher that already narrowed the visible devices per rank
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    # BENCH_FORCE_DIST=1: ONE rank walks the whole N > 1 code path over a real process group -- RCCL collectives on device
    # tensors (gather of the packed records, all_gather of the timings, all_reduce, barrier) with a world of one: the
    # plumbing that an 8-GPU run uses, on the one GPU a test box has
    multi = world > 1 or bool(os.environ.get("BENCH_FORCE_DIST"))
    depth = args.pipeline_depth if args.pipeline_depth > 0 else (2 if multi else 1)
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            from phylign_amd.launch import free_port
            os.environ.setdefault("MASTER_PORT", str(free_port()))
        if backend == "nccl":
            # the per-step gather runs beside the NEXT step's scan, which fills every CU: RCCL's kernels go on a
            # high-priority stream so that they are dispatched as soon as a wave slot frees instead of queueing behind it
            opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=os.environ.get("BENCH_RCCL_HIGH_PRIORITY", "1") != "0")
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank), pg_options=opts)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    pm.init(local_rank)
    if os.environ.get("PM_SINGLE_LAUNCH", "0") not in ("", "0"):
        pm.set_option("single_launch", 1)
    if os.environ.get("PM_WIDE_QUERY"):
        pm.set_option("wide_query", int(os.environ["PM_WIDE_QUERY"]))
    if os.environ.get("PM_WQ_SPLIT"):
        pm.set_option("wide_query_split", int(os.environ["PM_WQ_SPLIT"]))
    dev = pm.device_info()
    log(f"[bench] {dev['name']} free {dev['hbm_free'] / 1e9:.1f} GB of {dev['hbm_total'] / 1e9:.1f} GB")

    shapes = W.select(args.workload)
    if args.rows_divisor > 1:
        shapes = W.scale_shapes(shapes, args.rows_divisor)
    nparts = args.emulate_world if (args.emulate_world and world == 1) else world
    # static batch -> rank map; a few small batches are resident on two ranks that share their queries (assign_parts)
    if args.replicas:
        pparts = W.assign_parts(shapes, nparts, capacity_bytes=int(dev["hbm_total"] * 0.85))
    else:
        pparts = [[(p_, 0, W.PART_DEN) for p_ in part] for part in W.assign_batches(shapes, nparts, capacity_bytes=int(dev["hbm_total"] * 0.85))]
    parts = [[p_ for p_, _, _ in part] for part in pparts]
    base = 0
    bases = []
    for r in range(nparts):
        bases.append(base)
        base += len(parts[r])
    part_id = args.emulate_rank if nparts != world else rank
    mine = parts[part_id]
    my_shares = [None if hi - lo == W.PART_DEN else (lo, hi, W.PART_DEN) for _, lo, hi in pparts[part_id]]
    shared = sorted({p_ for part in pparts for p_, lo, hi in part if hi - lo != W.PART_DEN})

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
        if ix.device != local_rank:
            sys.exit(f"rank {rank}: batch {s.batch} landed on GPU {ix.device}, expected {local_rank}")
        if pos in plan:
            ix.plant(*plan[pos])
        indexes.append(ix)
    infos = [ix.info for ix in indexes]
    resident = sum(i.device_bytes for i in infos)
    log(f"[bench] rank0: {len(indexes)} batches, {resident / 1e9:.1f} GB of signatures resident, "
        f"{nq} queries / {n_terms} k-mers, setup {time.time() - t0:.1f}s")

    def rendezvous_store():
        return dist.distributed_c10d._get_default_store()

    def sync():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
            torch.cuda.synchronize()

    pg_dev = "cuda" if backend == "nccl" else "cpu"
    # BENCH_FORCE_GATHER=1: a single rank walks the device-tensor gather path of the N>1 runs (D2D copy of the
    # ordered records into the packed tensor, rank-0 read-back) without a collective -- functional check
    force_gather = bool(os.environ.get("BENCH_FORCE_GATHER"))
    packed = PackedGather(1 << 16, pg_dev)          # 64 Ki records (1 MiB) per rank in one collective
    last = {}
    kept = []        # results whose pinned records are still referenced (freed at the end)
    phase = {"queue": 0.0, "wait": 0.0, "host": 0.0, "gather": 0.0}

    def narrow_lines(infs, shares=None):
        """128-byte lines one k-mer touches in the batches of the mixed-width launch (rows of at most 256 bytes) per
        algorithmic byte of those rows; a batch searched with a share of the queries counts by that share"""
        w = [1.0 if sh is None else (sh[1] - sh[0]) / sh[2] for sh in (shares or [None] * len(infs))]
        lines = sum(max(1, int(i.stride) // 128) * w_ for i, w_ in zip(infs, w) if i.stride <= 256)
        rowb = sum(int(i.row_bytes) * w_ for i, w_ in zip(infs, w) if i.stride <= 256)
        return lines / rowb if rowb else 0.0

    # what the step functions below run on; swapped for the l31 and full_shard legs
    cur = {"indexes": indexes, "q": q, "n_terms": n_terms, "rowsum": sum(s.row_bytes for s in shapes),
           "slot_base": bases[part_id], "terms_per_q": terms_per_q, "tag": args.workload,
           "narrow_lines_per_byte": narrow_lines(infos, my_shares), "shares": my_shares}

    def queue_step():
        t_a = time.perf_counter()
        res = pm.search_async(cur["indexes"], cur["q"], args.threshold, slot_base=cur["slot_base"], nb_best_hits=args.nb_best_hits,
                              parts=cur.get("shares"))
        phase["queue"] += time.perf_counter() - t_a
        return res

    def finish_step(res, groups=None, keep=False):
        t_a = time.perf_counter()
        res.wait()
        st = res.stats
        t_b = time.perf_counter()
        # a7 ordering happens on the device (the kernel wrote each hit list as an ordered run, the host
        # orders the run directory, k_permute_runs moves the runs); ranks own disjoint, increasing slot
        # ranges, so the rank-order concatenation that the gather produces on rank 0 is globally ordered
        host = None
        if not multi and not force_gather:
            host = res.hits(copy=False)                    # view of the library's pinned buffer (lives as long as `res`)
            t_c = time.perf_counter()
        elif pg_dev == "cuda":
            _, n_local = res.ordered_device()
            t_c = time.perf_counter()
            if n_local <= packed.cap:
                res.copy_hits_device(packed.records_view().data_ptr(), packed.cap, ordered=True)
                g = packed.gather(n_local)
            else:
                big = torch.empty((n_local, 4), dtype=torch.int32, device="cuda")
                res.copy_hits_device(big.data_ptr(), n_local, ordered=True)
                g = packed.gather(n_local, overflow=big)
            if rank == 0:
                host = g.cpu().numpy().view(pm.HIT_DTYPE).reshape(-1)
        else:                                              # gloo: functional check of the same sequence
            mine_sorted = res.hits(copy=False)
            n_local = len(mine_sorted)
            t_c = time.perf_counter()
            t = torch.from_numpy(mine_sorted.view(np.int32).reshape(-1, 4).copy())
            if n_local <= packed.cap:
                packed.records_view()[:n_local].copy_(t)
                g = packed.gather(n_local)
            else:
                g = packed.gather(n_local, overflow=t)
            if rank == 0:
                host = g.numpy().view(pm.HIT_DTYPE).reshape(-1).copy()
        t_d = time.perf_counter()
        phase["wait"] += t_b - t_a; phase["host"] += t_c - t_b; phase["gather"] += t_d - t_c
        last["stats"] = st
        last["n_hits"] = len(host) if host is not None else None
        if groups is not None:
            for L in res.launches():
                g = groups.setdefault(L["kernel"], [0.0, 0.0, 0, 0, 0])   # [algorithmic bytes, ms, launches, batches, queries]
                g[0] += L["algorithmic_bytes"]; g[1] += L["ms"]; g[2] += 1; g[3] = L["n_batches"]; g[4] = L["n_queries"]
        if keep:
            # the records of the last step stay for the cross-mode comparison: keep the result alive
            # instead of copying hundreds of MB inside the timed region
            last["hits"] = host
            kept.append(res)
        else:
            res.free()

    def run_steps(n, groups=None, keep_last=False):
        """n steps; unless --no-pipeline the kernels of the next `depth` steps are queued before a step is finished (depth 1:
        step i + 1 runs while the host orders / gathers step i; depth 2, the default for N > 1: a gather that comes back
        late -- RCCL's kernels share the CUs with the running scan -- still finds a queued step behind the running one)"""
        if args.no_pipeline:
            for i in range(n):
                finish_step(queue_step(), groups, keep_last and i == n - 1)
            return
        inflight = []
        for i in range(n):
            inflight.append(queue_step())
            if len(inflight) > depth:
                finish_step(inflight.pop(0), groups)
        while inflight:
            res = inflight.pop(0)
            finish_step(res, groups, keep_last and not inflight)

    def timed_run(bound, warmup, steps):
        """W untimed steps, then exactly K steps between barrier + synchronize; MAX over ranks"""
        pm.set_option("threshold_bound", 1 if bound else 0)
        run_steps(warmup)
        sync()
        for k in phase:
            phase[k] = 0.0
        groups = {}      # kernel instantiation -> [algorithmic bytes, ms, launches, batches] over the timed steps
        t_start = time.perf_counter()
        run_steps(steps, groups, keep_last=True)
        sync()
        elapsed = time.perf_counter() - t_start
        rank_elapsed = [elapsed]
        rank_phase = [[phase["queue"], phase["wait"], phase["host"], phase["gather"], packed.waited_s]]
        if multi:
            t = torch.tensor([elapsed] + rank_phase[0], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            every = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(every, t)
            rank_elapsed = [float(x[0].item()) for x in every]
            rank_phase = [[float(v) for v in x[1:].tolist()] for x in every]
            elapsed = max(rank_elapsed)                       # the step of the job is the slowest rank's
        packed.waited_s = 0.0
        return {"elapsed": elapsed, "rank_elapsed": rank_elapsed, "rank_phase": rank_phase, "groups": groups, "phase": dict(phase), "hits": last.get("hits"),
                "n_hits": last.get("n_hits"), "stats": last["stats"]}

    def fetched_pass(bound):
        """one untimed search with the in-kernel counter on: algorithmic bytes really gathered, per kernel"""
        pm.set_option("threshold_bound", 1 if bound else 0)
        pm.set_option("count_fetched", 1)
        res = pm.search(cur["indexes"], cur["q"], args.threshold, slot_base=cur["slot_base"], nb_best_hits=args.nb_best_hits,
                        parts=cur.get("shares"))
        out = {L["kernel"]: (L["fetched_bytes"], L["algorithmic_bytes"]) for L in res.launches()}
        res.free()
        pm.set_option("count_fetched", 0)
        return out

    blob = kernel_blob_hash()

    def pmc_traffic(name, mode):
        """HBM bytes per launch from the PMC passes committed under profiles/ (tools/run_pmc.sh): valid only for the
        kernel source they were measured on (git blob of pm_kernels.hip), this workload and this query count"""
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        try:
            if world > 1 or args.emulate_world:
                return None, (f"PMC passes were collected on the 1-rank launch over all {len(shapes)} batches (profiles/pmc_traffic.json); "
                              f"a rank of {nparts} scans its own shard only, so the per-launch figure does not transfer")
            tj = json.load(open(tpath))
            if tj.get("pm_kernels_blob") != blob:
                return None, f"profiles/pmc_traffic.json was measured on pm_kernels.hip blob {str(tj.get('pm_kernels_blob'))[:12]}, this is {blob[:12]}: re-run tools/run_pmc.sh"
            if tj.get("workload") == cur["tag"] and tj.get("queries") == args.queries and args.rows_divisor == 1 \
                    and args.qlen == tj.get("query_len", 150):
                ent = tj.get("kernels", {}).get(name, {}).get(mode)
                if ent:
                    return ent.get("hbm_bytes_per_launch"), None
            return None, "no PMC pass committed for this workload / query count / row divisor (tools/run_pmc.sh)"
        except Exception:
            pass
        return None, None

    def roofline_entry(groups, name, steps, mode, fetched):
        abytes, ms, launches, nb, nqk = groups[name]
        alg_per_launch = abytes / launches
        moved = alg_per_launch
        if mode == "threshold_bound":
            moved = fetched[name][0] if fetched and name in fetched else None
        achieved = moved / (ms / launches * 1e-3) / 1e9 if moved is not None else None
        traffic, why = pmc_traffic(name, mode)
        r = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
             "frac": achieved / HBM_PEAK_GBPS if achieved is not None else None, "traffic": traffic, "kernel": name,
             "launches_per_step": launches / steps, "batches_per_launch": nb, "avg_launch_ms": ms / launches,
             "algorithmic_bytes_per_launch": alg_per_launch,
             "bytes_gathered_per_launch": moved}
        if why:
            r["traffic_note"] = why
        if mode == "threshold_bound":
            r["algorithmic_equivalent_GBps"] = alg_per_launch / (ms / launches * 1e-3) / 1e9
            r["note"] = ("achieved = algorithmic bytes of the row chunks this kernel really gathered (pm_set_option "
                         "count_fetched) / launch time; algorithmic_equivalent_GBps prices the launch at the bytes a "
                         "fetch-everything scan would move and may exceed the HBM peak")
        if traffic:
            r["hbm_GBps_from_traffic"] = traffic / (ms / launches * 1e-3) / 1e9
        return r

    def roofline_of(run, steps, mode, fetched):
        """dominant scan kernel.  fetch_all_rows: algorithmic bytes per launch / hipEvent launch duration.
        threshold_bound: the algorithmic bytes of the row chunks really gathered (in-kernel count) instead,
        with the data-independent figure kept as `algorithmic_equivalent_GBps`."""
        groups = run["groups"]
        if not groups:
            return None
        name = max(groups.items(), key=lambda kv: kv[1][1])[0]
        r = roofline_entry(groups, name, steps, mode, fetched)
        r["all_scan_kernels_GBps"] = sum(g[0] for g in groups.values()) / (sum(g[1] for g in groups.values()) * 1e-3) / 1e9
        return r

    def roofline_narrow_of(run, steps, mode, fetched):
        """the kernel FURTHEST below the roofline: the mixed-width launch of the narrow batches (rows of at most 256
        bytes).  Every lookup of a 13 ... 125-byte row moves one whole 128-byte line from HBM (measured: one
        TCC_EA0_RDREQ_128B per lookup, profiles/r03/pmc_calibration_narrow.txt), so the wire carries
        `wire_GBps_at_128B_per_line` while the algorithmic rate is what `achieved` says."""
        groups = run["groups"]
        names = [k for k in groups if "G=mixed" in k or any(f"G={g}," in k for g in (1, 2, 4, 8, 16))]
        if not names:
            return None
        name = max(names, key=lambda k: groups[k][1])
        r = roofline_entry(groups, name, steps, mode, fetched)
        abytes, ms, launches, nb, nqk = groups[name]
        if mode == "fetch_all_rows":
            lines = cur["narrow_lines_per_byte"] * abytes / launches              # 128-byte lines touched per launch
            r["lines_per_s"] = lines / (ms / launches * 1e-3)
            r["wire_GBps_at_128B_per_line"] = lines * 128 / (ms / launches * 1e-3) / 1e9
            r["wire_frac_of_peak"] = r["wire_GBps_at_128B_per_line"] / HBM_PEAK_GBPS
        return r

    def summary(run, steps, mode, fetched):
        el = run["elapsed"] / steps
        alg_total = cur["rowsum"] * cur["n_terms"]
        frac_key = "hbm_fraction_whole_step_algorithmic" if mode == "fetch_all_rows" else "whole_step_algorithmic_equivalent_over_peak"
        out = {"value": cur["n_terms"] / el, "unit": "k-mers/s", "ms_per_step": el * 1e3,
               frac_key: alg_total / el / (HBM_PEAK_GBPS * 1e9 * world),
               "roofline": roofline_of(run, steps, mode, fetched),
               "roofline_narrow": roofline_narrow_of(run, steps, mode, fetched),
               "hits": run["n_hits"],
               "rank0_ms": {"kernels_total": run["stats"].ms_total,
                            "host_wait_for_gpu": run["phase"]["wait"] / steps * 1e3,
                            "run_order_on_device_and_d2h": run["phase"]["host"] / steps * 1e3,
                            "hit_gather": run["phase"]["gather"] / steps * 1e3}}
        if fetched:
            tot_f = sum(v[0] for v in fetched.values()); tot_a = sum(v[1] for v in fetched.values())
            out["fraction_of_row_bytes_gathered_rank0"] = tot_f / tot_a if tot_a else None
        return out

    modes = ["fetch_all_rows", "threshold_bound"]
    head = args.headline
    other = [m for m in modes if m != head][0]
    full = not (args.only_headline or args.emulate_world)

    run_head = timed_run(head == "threshold_bound", args.warmup, args.steps)
    fetched_head = fetched_pass(head == "threshold_bound") if full or head == "threshold_bound" else None
    if fetched_head and head == "fetch_all_rows":
        for k, (f, a) in fetched_head.items():
            assert f == a, f"{k}: fetch-all scan gathered {f} bytes, algorithmic bytes are {a}"
    sum_head = summary(run_head, args.steps, head, fetched_head)
    st_head = run_head["stats"]

    # ---- who took part (N > 1): a real all_reduce on device tensors counts the RCCL ranks, every rank reports the
    # GPU that holds its matrices and its own time for the K steps
    participants = {"backend": backend if multi else None, "rccl_ranks": None,
                    "rank_ms_per_step": [e / args.steps * 1e3 for e in run_head["rank_elapsed"]],
                    # host time per step and rank: queueing the launches, waiting for the GPU, ordering the runs on the device,
                    # the gather of hit records (root: until every rank's records are on its host; others: queueing the send),
                    # and waiting for a send buffer to come free again (double-buffered: ~0)
                    "rank_host_ms": [dict(zip(("queue_launches", "wait_for_gpu", "run_order", "hit_gather", "send_buffer_wait"),
                                              [round(v / args.steps * 1e3, 4) for v in ph_])) for ph_ in run_head["rank_phase"]],
                    "rank_devices": [indexes[0].device if indexes else local_rank]}
    if multi:
        ddev = "cuda" if backend == "nccl" else "cpu"
        one = torch.ones(1, dtype=torch.int32, device=ddev)
        dist.all_reduce(one)
        devs = [torch.zeros(2, dtype=torch.int32, device=ddev) for _ in range(world)]
        dist.all_gather(devs, torch.tensor([indexes[0].device if indexes else -1, len(indexes)], dtype=torch.int32, device=ddev))
        participants.update({"rccl_ranks": int(one.item()) if backend == "nccl" else None, "ranks": int(one.item()),
                             "rank_devices": [int(d[0].item()) for d in devs], "rank_batches": [int(d[1].item()) for d in devs]})

    sum_other = None
    ok = True
    if full:
        run_other = timed_run(other == "threshold_bound", 1, args.steps)      # pools are warm: same record volume
        fetched_other = fetched_pass(other == "threshold_bound")
        sum_other = summary(run_other, args.steps, other, fetched_other)
        same = bool(rank != 0 or np.array_equal(run_head["hits"], run_other["hits"]))
        sum_other["hits_identical_to_headline"] = same
        sum_other["speed_vs_headline"] = sum_other["value"] / sum_head["value"]
        ok = ok and same

    # ---- SURVEY.md 8d, many-queries regime: "report additionally unique_rows x row_bytes and label it".  One untimed
    # device pass per resident batch marks the rows the 12 M k-mers map to in a bitmap of signature_size bits.
    unique_rows = None
    if (full or multi) and not args.no_unique_rows and not args.only_headline and not args.emulate_world:
        from phylign_amd import bench_aids
        t0 = time.time()
        mine_rows = mine_bytes = mine_alg = 0
        for ix, inf in zip(indexes, infos):
            u = bench_aids.unique_rows(ix, q)
            mine_rows += u; mine_bytes += u * int(inf.row_bytes); mine_alg += n_terms * int(inf.num_hashes) * int(inf.row_bytes)
        tot = [mine_rows, mine_bytes, mine_alg, sum(int(i.signature_size) for i in infos), sum(int(i.signature_size) * int(i.row_bytes) for i in infos)]
        if multi:
            t = torch.tensor(tot, dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(t)
            tot = [int(v) for v in t.tolist()]
        unique_rows = {"label": "unique_rows x row_bytes (SURVEY.md 8d, many-queries regime): the distinct signature rows the query set's "
                                "k-mers map to, summed over the resident batches -- what a scan with perfect row reuse would read; "
                                "NOT what `value`, `roofline` or the algorithmic bytes are priced at",
                       "unique_rows": tot[0], "unique_rows_x_row_bytes": tot[1], "algorithmic_bytes_per_step": tot[2],
                       "ratio_to_algorithmic": tot[1] / tot[2] if tot[2] else None,
                       "rows_resident": tot[3], "fraction_of_resident_rows_touched": tot[0] / tot[3] if tot[3] else None,
                       "fraction_of_resident_matrix_bytes": tot[1] / tot[4] if tot[4] else None,
                       "pass_s": round(time.time() - t0, 3)}

    # ---- SURVEY.md 8d's third query shape: the reference's bundled gene file data/ARGannot_r3.fa -- 1 856 genes of
    # 237 ... 3 153 bp (1 594 532 31-mers) in file order, uniform ACGT of those lengths.  Every gene falls into the 10- or
    # the 13-plane counter class (128 ... 1 023 / 1 024 ... 8 191 k-mers); the file as it is (x1) and eight files' worth
    # (x8: 12.8 M k-mers, the headline's volume) against the resident set, both scan modes, records compared.
    argannot = None
    if full and not args.no_argannot and world == 1:
        saved = dict(cur)
        try:
            lens = W.argannot_lengths()
            argannot = {"workload": f"length mix of data/ARGannot_r3.fa ({len(lens)} genes, {min(lens)}-{max(lens)} bp, "
                                    f"{sum(n - 30 for n in lens)} 31-mers per file) against the {len(indexes)} resident batches"}
            for rep in (1, 8):
                terms = [n - 30 for n in lens] * rep
                fa_g, _ = W.make_queries_lengths(lens * rep, seed=41 + rep, prefix="gene")
                qg = pm.Queries(fa_g, term_size=31)
                nqg, ntg = qg.count()
                plan_g, sure_g = W.plant_plan_ragged(qg.hash_terms(1, 1), terms, shapes, every=16, threshold=args.threshold)
                for pos, ix in zip(mine, indexes):
                    if pos in plan_g:
                        ix.plant(*plan_g[pos])
                cur.update({"q": qg, "n_terms": ntg, "terms_per_q": None, "tag": f"{args.workload}/argannot_x{rep}"})
                g_steps = max(3, min(args.steps, 10))
                ent = {"queries": nqg, "kmers": ntg, "planted_pairs_at_or_above_threshold": sure_g}
                g_runs = {}
                for m in modes:
                    r = timed_run(m == "threshold_bound", 2, g_steps)
                    fp = fetched_pass(m == "threshold_bound")
                    if m == "fetch_all_rows":
                        for k_, (f_, a_) in fp.items():
                            assert f_ == a_, f"{k_}: fetch-all scan gathered {f_} bytes, algorithmic bytes are {a_}"
                    g_runs[m] = r
                    sm = summary(r, g_steps, m, fp)
                    sm["scan_launches"] = {k_: {"launches_per_step": v[2] / g_steps, "batches": v[3], "queries": v[4], "avg_ms": v[1] / v[2],
                                                "algorithmic_GBps": v[0] / (v[1] * 1e-3) / 1e9} for k_, v in r["groups"].items()}
                    ent[m] = sm
                same = bool(np.array_equal(g_runs[modes[0]]["hits"], g_runs[modes[1]]["hits"]))
                n_real = int(np.count_nonzero(g_runs[modes[0]]["hits"]["doc"] != pm.PM_DOC_COUNT))
                ent["hits_identical"] = same
                ok = ok and same and n_real >= sure_g
                argannot[f"x{rep}"] = ent
                cur.clear(); cur.update(saved)
                qg.free()
        except Exception as e:                                       # an optional leg never costs the headline line
            log(f"[bench] argannot leg failed: {e!r}")
            argannot = dict(argannot or {}, error=repr(e))
            cur.clear(); cur.update(saved)
        pm.set_option("threshold_bound", 1)

    # ---- clustered variant: every query gets a home batch (changes the resident matrices: last)
    clustered = None
    if full and not args.no_clustered and (world == 1 or args.clustered_multi):
        try:
            t0 = time.time()
            for pos, ix in zip(mine, indexes):
                ix.plant_cluster(q, pos, len(shapes), seed=97)
            log(f"[bench] clustered planting {time.time() - t0:.1f}s")
            c_runs = {}
            for m in modes:
                # 3 warm-up steps: the hit lists are ~700x longer here, and the pooled device / pinned
                # buffers of two searches in flight are (re)allocated once, outside the timed steps
                r = timed_run(m == "threshold_bound", 3, args.steps)
                c_runs[m] = (r, fetched_pass(m == "threshold_bound"))
            same = bool(rank != 0 or np.array_equal(c_runs[modes[0]][0]["hits"], c_runs[modes[1]][0]["hits"]))
            ok = ok and same
            clustered = {
                "data": ("every query has one home batch (query i -> batch i mod %d) in which about half of the "
                         "32-document clusters match it at a k-mer fraction of 0.60-1.0 (k_plant_cluster); elsewhere "
                         "only the Bernoulli(1/4) background" % len(shapes)),
                "hits_identical": same,
            }
            for m in modes:
                clustered[m] = summary(c_runs[m][0], args.steps, m, c_runs[m][1])
            clustered["threshold_bound"]["speed_vs_fetch_all_rows"] = (clustered["threshold_bound"]["value"] /
                                                                       clustered["fetch_all_rows"]["value"])
        except Exception as e:                                       # an optional leg never costs the headline line
            if multi:
                raise
            log(f"[bench] clustered leg failed: {e!r}")
            clustered = {"error": repr(e)}
    pm.set_option("threshold_bound", 1)

    # ---- BASELINE configs[3] on real GPUs: with 8 ranks (or BENCH_FULL_MIN_WORLD) ALL 305 batches of batches_full.txt are
    # sharded over the ranks -- 1.06 TB of signatures, ~135 GB per GPU -- and the same query set is searched against the whole
    # 661k collection with the per-step gather of hit records: whole-job k-mers/s against the full index.  The config-3 shard
    # of every rank is freed first.  (One GPU holds the same shard in the `full_shard` object of the N = 1 line.)
    full_collection = None
    min_world_full = int(os.environ.get("BENCH_FULL_MIN_WORLD", "8"))
    if world > 1 and world >= min_world_full and not args.no_full_shard and not args.only_headline and not args.emulate_world \
            and args.workload == "config3":
        for ix in indexes:
            ix.free()
        indexes = []
        t0 = time.time()
        fshapes = W.select("full")
        if args.rows_divisor > 1:
            fshapes = W.scale_shapes(fshapes, args.rows_divisor)
        fparts = W.assign_batches(fshapes, world, capacity_bytes=int(dev["hbm_total"] * 0.85))
        fbases, acc_b = [], 0
        for r in range(world):
            fbases.append(acc_b)
            acc_b += len(fparts[r])
        fplan, fsure = W.plant_plan(q.hash_terms(1, 1), nq, terms_per_q, fshapes)
        setup_error = None
        try:
            for pos in fparts[rank]:
                sh = fshapes[pos]
                ix = pm.Index.synth(sh.batch_id, sh.n_docs, sh.signature_size, 1, 31, 661, layout=args.layout)
                indexes.append(ix)
                if pos in fplan:
                    ix.plant(*fplan[pos])
        except Exception as e:                       # e.g. a GPU with less free memory than its shard needs
            setup_error = repr(e)
            log(f"[bench] full_collection: rank {rank} could not build its shard: {setup_error}")
        # the ranks agree before the first collective of the leg: one failed shard skips it everywhere, the headline line stays
        flag = torch.tensor([0 if setup_error else 1], dtype=torch.int32, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        fc_ready = bool(flag.item())
        finfos = [ix.info for ix in indexes]
        if fc_ready:
            log(f"[bench] full_collection: rank0 holds {len(indexes)} of {len(fshapes)} batches, "
                f"{sum(i.device_bytes for i in finfos) / 1e9:.1f} GB resident, setup {time.time() - t0:.1f}s")
            saved = dict(cur)
            cur.update({"indexes": indexes, "rowsum": sum(sh.row_bytes for sh in fshapes), "slot_base": fbases[rank],
                        "tag": f"full/{world}", "narrow_lines_per_byte": narrow_lines(finfos), "shares": None})
            fc_steps = max(3, min(args.steps, 10))
            full_collection = {"workload": f"all {len(fshapes)} batches of batches_full.txt ({sum(sh.index_bytes for sh in fshapes) / 1e12:.2f} TB of "
                                           f"signatures, {cur['rowsum']} row bytes per k-mer) sharded over {world} ranks, {nq} x {args.qlen} bp queries, "
                                           "one gather of hit records per step",
                               "planted_pairs_at_or_above_threshold": fsure, "rank_batches": [len(p_) for p_ in fparts]}
            fc_runs = {}
            for m in modes:
                r = timed_run(m == "threshold_bound", 2, fc_steps)
                fc_runs[m] = r
                full_collection[m] = summary(r, fc_steps, m, None)
                full_collection[m]["rank_ms_per_step"] = [e / fc_steps * 1e3 for e in r["rank_elapsed"]]
            if rank == 0:
                same = bool(np.array_equal(fc_runs[modes[0]]["hits"], fc_runs[modes[1]]["hits"]))
                n_real = int(np.count_nonzero(fc_runs[modes[0]]["hits"]["doc"] != pm.PM_DOC_COUNT))
                full_collection["hits_identical"] = same
                ok = ok and same and n_real >= fsure
                if args.dump_full_hits:
                    pos_of = np.zeros(len(fshapes), dtype=np.uint32)
                    for r_ in range(world):
                        for i, pos in enumerate(fparts[r_]):
                            pos_of[fbases[r_] + i] = pos
                    h = fc_runs[modes[0]]["hits"].copy()
                    h["slot"] = pos_of[h["slot"]]
                    np.save(args.dump_full_hits, pm.sort_hits(np.ascontiguousarray(h)))
            cur.clear(); cur.update(saved)
            pm.set_option("threshold_bound", 1)
        else:
            full_collection = {"skipped": setup_error or "another rank could not build its shard"}
            for ix in indexes:
                ix.free()
            indexes = []

    # ---- BASELINE configs[1] read literally: "10k synthetic 31-mer queries" = ONE k-mer per query (hit <=> bit
    # set: with Bernoulli(1/4) signatures a quarter of all documents match every query, so this leg is bound by
    # writing hit records, not by the scan).  (a) config 2: one batch, latency of one query set; (b) the resident
    # index set of this run, k-mers/s.
    l31 = None
    if full and not args.no_l31 and world == 1 and args.qlen != 31:
        l31 = {}
        fasta31, _ = W.make_queries(10000, 31, seed=32)
        q31 = pm.Queries(fasta31, term_size=31)
        ba = [ix for pos, ix in zip(mine, indexes) if shapes[pos].batch == "bacillus_anthracis__01"]
        if ba:
            lat, st31, nrec = [], None, 0
            for i in range(8):
                t_a = time.perf_counter()
                r = pm.search(ba, q31, args.threshold, nb_best_hits=args.nb_best_hits)
                h = r.hits(copy=False)
                lat.append(time.perf_counter() - t_a)
                st31, nrec = r.stats, len(h)
                r.free()
            l31["config2"] = {"workload": "bacillus_anthracis__01 shape (664 documents, 83-byte rows), 10 000 x 31-bp queries, threshold 0.7",
                              "latency_ms_per_query_set_median": float(np.median(lat[2:]) * 1e3),
                              "kernels_ms": st31.ms_total, "hash_ms": st31.ms_hash, "scan_ms": st31.ms_scan,
                              "kmers_per_s": 10000 / float(np.median(lat[2:])), "hit_records": nrec,
                              "note": "wall time of pm_search + records on the host (pinned); one k-mer per query"}
        # the resident set: as many 31-mer queries as keep one search near 2^26 hit records (1 GB of them)
        docs = sum(i.n_docs for i in infos)
        nq31 = int(max(256, min(10000, (1 << 26) // max(1, docs // 4))))
        fasta31b, _ = W.make_queries(nq31, 31, seed=33)
        q31b = pm.Queries(fasta31b, term_size=31)
        saved = dict(cur)
        cur.update({"q": q31b, "n_terms": nq31, "terms_per_q": 1, "tag": args.workload + "/l31"})
        try:
            r31 = timed_run(False, 2, max(3, min(args.steps, 5)))
            s31 = summary(r31, max(3, min(args.steps, 5)), "fetch_all_rows", None)
            l31["resident_set"] = {"workload": f"{len(indexes)} resident batches ({docs} documents), {nq31} x 31-bp queries (query count capped "
                                               "so that one search stays near 2^26 hit records)",
                                   "value": s31["value"], "unit": "k-mers/s", "ms_per_step": s31["ms_per_step"],
                                   "hit_records_per_step": s31["hits"], "roofline": s31["roofline"],
                                   "note": "output-bound: every query matches ~1/4 of all documents (Bernoulli(1/4) signatures, "
                                           "threshold 0.7 of 1 k-mer = 1); records are 16 bytes each"}
        finally:
            cur.clear(); cur.update(saved)
        q31.free(); q31b.free()

    # ---- BASELINE configs[3] / configs[4] on their own workload: the shard ONE rank holds when all 305 batches of
    # batches_full.txt are split over 8 GPUs (38 batches, ~135 GB), 100 k queries (config 4's per-rank step) and
    # 1 M queries (config 5's match-only time; the end-to-end figure is tools/e2e_config5.py).  The headline set is
    # freed first: both do not fit one GPU together.
    full_shard = None
    if full and not args.no_full_shard and world == 1 and args.workload == "config3" and args.rows_divisor == 1:
        saved = dict(cur)
        try:
            for ix in indexes:
                ix.free()
            indexes = []
            t0 = time.time()
            fshapes = W.select("full")
            fmine = W.assign_batches(fshapes, args.full_shard_world)[args.full_shard_rank]
            fsub = [fshapes[p] for p in fmine]
            fplan, fsure = W.plant_plan(q.hash_terms(1, 1), nq, terms_per_q, fsub)
            for i, sshape in enumerate(fsub):
                ix = pm.Index.synth(sshape.batch_id, sshape.n_docs, sshape.signature_size, 1, 31, 661, layout=args.layout)
                if i in fplan:
                    ix.plant(*fplan[i])
                indexes.append(ix)
            finfos = [ix.info for ix in indexes]
            log(f"[bench] full_shard: {len(indexes)} batches, {sum(i.device_bytes for i in finfos) / 1e9:.1f} GB resident, setup {time.time() - t0:.1f}s")
            saved = dict(cur)
            cur.update({"indexes": indexes, "rowsum": sum(sh.row_bytes for sh in fsub), "slot_base": 0,
                        "tag": f"full/{args.full_shard_world}/{args.full_shard_rank}", "narrow_lines_per_byte": narrow_lines(finfos), "shares": None})
            fs_steps = max(3, min(args.steps, 10))
            full_shard = {"workload": f"rank {args.full_shard_rank} of {args.full_shard_world} of batches_full.txt (305 batches, 1.06 TB, 82 741 row "
                                      f"bytes per k-mer): {len(fsub)} batches, {sum(sh.index_bytes for sh in fsub) / 1e9:.1f} GB on disk, "
                                      f"{sum(i.device_bytes for i in finfos) / 1e9:.1f} GB resident, {cur['rowsum']} row bytes per k-mer; "
                                      f"{nq} x {args.qlen} bp queries",
                          "planted_pairs_at_or_above_threshold": fsure}
            f_runs = {}
            for m in modes:
                r = timed_run(m == "threshold_bound", 2, fs_steps)
                f_runs[m] = r
                full_shard[m] = summary(r, fs_steps, m, fetched_pass(m == "threshold_bound"))
            same = bool(np.array_equal(f_runs[modes[0]]["hits"], f_runs[modes[1]]["hits"]))
            n_real = int(np.count_nonzero(f_runs[modes[0]]["hits"]["doc"] != pm.PM_DOC_COUNT))
            full_shard["hits_identical"] = same
            ok = ok and same and n_real >= fsure
            # config 5's query count: match-only rate at 1 M queries
            t0 = time.time()
            fasta1m, _ = W.make_queries(1_000_000, args.qlen, seed=5)
            q1m = pm.Queries(fasta1m, term_size=31)
            del fasta1m
            plan1m, sure1m = W.plant_plan(q1m.hash_terms(1, 1), 1_000_000, terms_per_q, fsub, every=2500, docs_per_query=8)
            for i, ix in enumerate(indexes):
                if i in plan1m:
                    ix.plant(*plan1m[i])
            cur.update({"q": q1m, "n_terms": 1_000_000 * terms_per_q, "tag": cur["tag"] + "/1M"})
            full_shard["queries_1M"] = {"setup_s": round(time.time() - t0, 2), "planted_pairs_at_or_above_threshold": sure1m}
            for m in modes:
                r = timed_run(m == "threshold_bound", 1, 2)
                sm = summary(r, 2, m, None)
                ok = ok and sm["hits"] is not None and sm["hits"] >= sure1m
                full_shard["queries_1M"][m] = {"value": sm["value"], "unit": "k-mers/s", "ms_per_step": sm["ms_per_step"],
                                               "hits": sm["hits"], "roofline": sm["roofline"], "roofline_narrow": sm["roofline_narrow"]}
            cur.clear(); cur.update(saved)
            q1m.free()
            pm.set_option("threshold_bound", 1)
        except Exception as e:
            log(f"[bench] full_shard leg failed: {e!r}")
            full_shard = dict(full_shard or {}, error=repr(e))
            cur.clear(); cur.update(saved)
            pm.set_option("threshold_bound", 1)

    ph = run_head["phase"]
    out = {
        "metric": "query k-mers matched/sec vs 661k-shaped COBS index",
        "value": sum_head["value"], "unit": "k-mers/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": sum_head["ms_per_step"], "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {len(shapes)} 661k-shaped batches "
                               f"({sum(s.index_bytes for s in shapes) / 1e9:.1f} GB of signatures, "
                               f"{sum(s.row_bytes for s in shapes)} row bytes per k-mer), "
                               f"{nq} synthetic {args.qlen}-bp queries ({terms_per_q} 31-mers each), threshold {args.threshold}",
                   "batches": len(shapes), "queries": nq, "query_len": args.qlen, "k": 31,
                   "num_hashes": 1, "threshold": args.threshold, "nb_best_hits": args.nb_best_hits,
                   "rows_divisor": args.rows_divisor,
                   "sharding": f"{world} rank(s), static LPT batch assignment, one gather of hit records",
                   "batches_on_two_ranks": [{"batch": shapes[p_].batch, "index_GB": round(shapes[p_].index_bytes / 1e9, 3),
                                             "query_shares": {str(r_): [lo, hi, W.PART_DEN] for r_, part in enumerate(pparts)
                                                              for q_, lo, hi in part if q_ == p_}} for p_ in shared],
                   "pipelined_steps": not args.no_pipeline, "pipeline_depth": 0 if args.no_pipeline else depth},
        "scan_mode": (head + (": every signature row of every k-mer is gathered, like `cobs query` does -- independent "
                              "of the data; the product default (threshold_bound) is reported beside it"
                              if head == "fetch_all_rows" else
                              ": a signature line is no longer fetched once none of its documents can reach "
                              "ceil(threshold*k-mers); identical results, data-dependent speed")),
        "value_note": ("round 1 reported the threshold_bound mode as `value` (5.3e8 k-mers/s, roofline priced at bytes it "
                       "did not move); since round 2 `value`/`roofline` are the fetch-every-row scan and the product "
                       "default is the `threshold_bound` object -- compare that one with round 1"
                       if head == "fetch_all_rows" else "headline switched to the data-dependent threshold_bound mode by --headline"),
        "hbm_fraction_whole_step": sum_head.get("hbm_fraction_whole_step_algorithmic", sum_head.get("whole_step_algorithmic_equivalent_over_peak")),
        "hits": sum_head["hits"],
        "planted_pairs_at_or_above_threshold": sure_hits,
        "rank0_ms": {"kernels_total": st_head.ms_total, "hash": st_head.ms_hash, "scan": st_head.ms_scan,
                     "host_queue_launches": ph["queue"] / args.steps * 1e3,
                     "host_wait_for_gpu": ph["wait"] / args.steps * 1e3,
                     "run_order_on_device_and_d2h": ph["host"] / args.steps * 1e3,
                     "hit_gather": ph["gather"] / args.steps * 1e3},
        "scan_launches": {k: {"launches_per_step": v[2] / args.steps, "batches": v[3], "avg_ms": v[1] / v[2],
                              "algorithmic_GBps": v[0] / (v[1] * 1e-3) / 1e9} for k, v in run_head["groups"].items()},
        "pm_kernels_blob": blob,
        "participants": participants,
        "roofline": sum_head["roofline"],
        "roofline_narrow": sum_head["roofline_narrow"],
        "arithmetic": "bitwise AND / carry-save adders on u32 words (bit-sliced per-document counters), u64 integer hashing",
        other: sum_other,
        "unique_rows": unique_rows,
        "argannot": argannot,
        "clustered": clustered,
        "l31": l31,
        "full_shard": full_shard,
        "full_collection": full_collection,
    }
    if world > 1:
        # what an N > 1 line leaves out, and why (the N = 1 line of the same run carries them)
        why = {"argannot": "single-GPU leg: 1.6 M k-mers per file are not a multi-GPU workload; see the N = 1 line",
               "l31": "single-GPU leg: bound by the read-back of hit records, not by the scan; see the N = 1 line",
               "full_shard": "single-GPU stand-in for one rank of `full_collection`, which this line runs itself with 8+ ranks",
               "clustered": "ships ~0.5 GB of records per step to rank 0; run with --clustered-multi to include it"}
        for k_, v_ in why.items():
            if out.get(k_) is None:
                out[k_] = {"skipped": v_}
        if out.get("full_collection") is None:
            out["full_collection"] = {"skipped": f"runs with {min_world_full}+ ranks (all 305 batches need 8 x 135 GB of HBM)"}
    if args.emulate_world:
        out["emulated_shard"] = f"rank {part_id} of {nparts}"
    if rank == 0 and not args.emulate_world and run_head["hits"] is not None:
        n_real = int(np.count_nonzero(run_head["hits"]["doc"] != pm.PM_DOC_COUNT))
        if n_real < sure_hits:
            log(f"bench self-check failed: {n_real} hits < {sure_hits} planted pairs")
            ok = False
    if rank == 0 and not ok:
        log("bench self-check failed: scan modes disagree or planted hits are missing")
    if rank == 0 and args.dump_hits and run_head["hits"] is not None:
        # slots are numbered rank-major; name every record by its batch (position in the shape list)
        # and re-order, so that runs with different shardings can be compared record by record
        pos_of_slot = np.zeros(sum(len(p_) for p_ in parts), dtype=np.uint32)     # (a batch on two ranks has two slots)
        for r in range(nparts):
            for i, pos in enumerate(parts[r]):
                pos_of_slot[bases[r] + i] = pos
        h = run_head["hits"].copy()
        h["slot"] = pos_of_slot[h["slot"]]
        np.save(args.dump_hits, pm.sort_hits(np.ascontiguousarray(h)))
    # every rank learns the verdict and leaves together (no rank waits in a barrier for one that exited)
    if multi:
        # RCCL prints its version banner through C stdio, which is fully buffered when stdout is a file or a pipe: left
        # alone it comes out at exit, BEHIND rank 0's JSON line.  Every rank pushes it out before the collective that
        # precedes the line, so the line is the last thing on stdout.
        flush_c_stdio()
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = bool(flag.item())
    # ---- after the timed legs rank 0 works alone while the other ranks sleep on a key of the rendezvous store (a socket wait: no
    # GPU work queued, no core spinning beside the CPU threads):
    # (1) HBM traffic of the headline's scan kernels from PMC passes of THIS run (see live_pmc_traffic).  The GPU work of this
    #     process is over: its matrices, query sets and pooled buffers are released first, the children need the room.  With
    #     N > 1 the children hold the shard rank 0 had (--emulate-world N --emulate-rank 0) on rank 0's GPU: the line's
    #     `roofline` is rank 0's launch, and so is its `traffic`.
    # (2) the CPU path, timed in the same run at every N (north_star) on the host's cores.
    want_pmc = full and not args.no_live_pmc and head == "fetch_all_rows"
    want_cpu = not args.no_cpu_baseline
    if ok and rank == 0 and not want_cpu:
        out["cpu_baseline"] = None
    if ok and (want_pmc or want_cpu):
        if rank == 0:
            try:
                if want_pmc:
                    for ix in indexes:
                        ix.free()
                    indexes = []
                    q.free()
                    pm.set_option("release_pools", 1)
                    wl = ["--workload", args.workload, "--queries", str(args.queries), "--qlen", str(args.qlen), "--threshold", str(args.threshold),
                          "--nb-best-hits", str(args.nb_best_hits), "--rows-divisor", str(args.rows_divisor), "--layout", str(args.layout)]
                    if world > 1:
                        wl += ["--emulate-world", str(world), "--emulate-rank", "0"] + (["--replicas"] if args.replicas else [])
                    table, why, took = live_pmc_traffic(wl, args.live_pmc_budget_s, log)
                    log(f"[bench] live PMC passes: {'ok' if table else 'failed (' + str(why) + ')'} in {took:.1f}s")
                    out["live_pmc"] = {"seconds": round(took, 1), "kernels": table, "error": why,
                                       "how": "rocprofv3 --kernel-trace --pmc, one counter group per child run of `bench.py --steps 1 --warmup 0 "
                                              "--only-headline --no-pipeline` on the same workload" +
                                              (f" and rank 0's shard of the {world}-way split (--emulate-world {world} --emulate-rank 0)" if world > 1 else "") +
                                              "; bytes = 128 x TCC_EA0_RDREQ_128B_sum + 64 x _64B_sum + 32 x _32B_sum + 1024 x WRITE_SIZE per launch"}
                    for key in ("roofline", "roofline_narrow"):
                        r = out.get(key)
                        if not r:
                            continue
                        if table and r["kernel"] in table:
                            if r.get("traffic") is not None:
                                r["traffic_committed"] = r["traffic"]             # profiles/pmc_traffic.json, same kernel source: for comparison
                            r["traffic"] = table[r["kernel"]]["hbm_bytes_per_launch"]
                            r["hbm_GBps_from_traffic"] = r["traffic"] / (r["avg_launch_ms"] * 1e-3) / 1e9
                            r["traffic_source"] = (f"live: rocprofv3 --pmc passes run by this bench.py invocation ({took:.0f} s, "
                                                   f"{table[r['kernel']]['launches']} launch" + (f", rank 0's shard of {world}" if world > 1 else "") + ")")
                            r.pop("traffic_note", None)
                        elif r.get("traffic") is not None:
                            r["traffic_source"] = f"profiles/pmc_traffic.json (committed, same pm_kernels.hip blob); live pass: {why}"
                        else:
                            r["traffic_note"] = _clip(f"live pass: {why}; " + (r.get("traffic_note") or ""), 300)
                if want_cpu:
                    out["cpu_baseline"] = cpu_baseline(shapes, seqs, args.qlen, args.threshold,
                                                       args.cpu_target_s, args.cpu_sample_gb, log)
                    out["cpu_baseline"]["timed_while"] = ("the only process on the host" if world == 1 else
                                                          f"the other {world - 1} ranks slept on a rendezvous-store key, GPUs idle")
                    out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
            finally:
                if world > 1:
                    rendezvous_store().set("bench/cpu_baseline_done", "1")
        elif world > 1:
            from datetime import timedelta
            rendezvous_store().wait(["bench/cpu_baseline_done"], timedelta(seconds=1500))
    if ok and rank == 0:
        if args.whole_record:
            flush_c_stdio()
            print(json.dumps(out, allow_nan=False), flush=True)
        else:
            emit(out, args.legs_out, log)
        # the same headline on stderr, in words: whoever reads only the log tail still sees what the line said
        rf_, cb_ = out.get("roofline") or {}, out.get("cpu_baseline") or {}
        log(f"[bench] n_gpus {world}: {out['value']:.4g} k-mers/s, {out['ms_per_step']:.3f} ms/step; {rf_.get('kernel')} {rf_.get('avg_launch_ms', 0):.3f} ms = "
            f"{rf_.get('achieved') or 0:.0f} GB/s = {rf_.get('frac') or 0:.3f} of {HBM_PEAK_GBPS:.0f}, traffic {rf_.get('traffic')} B"
            + (f"; cpu {cb_['value']:.4g} k-mers/s on {cb_['cores']} threads" if cb_ else ""))
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        sys.exit(1)


if __name__ == "__main__":
    main()

"""Whole 03_match stage (and optionally 04_filter) for a list of phylogenetic
batches on one or several MI355X: the multi-batch / multi-GPU form of what the
reference runs as one Snakemake job per batch (Snakefile:431-487 rule
decompress_and_run_cobs, then Snakefile:490-520 rule translate_matches).

    python -m phylign_amd.match_stage --batches data/batches_full.txt --cobs-dir cobs \
        --sizes data/decompressed_indexes_sizes.txt --queries intermediate/01_queries_merged/Q.fa \
        --out-dir intermediate/03_match [--filter-out intermediate/04_filter/Q.fa]
    python -m torch.distributed.run --nproc-per-node 8 ... -m phylign_amd.match_stage ...   # one rank per GPU

Per rank: its batches (static LPT map on index bytes) are decoded by a pool of
`xzcat` pipes and streamed into HBM while the GPU searches the previous ones;
every batch yields `<out-dir>/<batch>____<qfile>.gz` with exactly the bytes of
`run_cobs_streaming.sh ... | postprocess_cobs.py -n N | gzip` after gunzip.
With --filter-out the pruned hit records are gathered to rank 0 (RCCL) and
merged natively into the 04_filter FASTA (scripts/filter_queries.py semantics).
"""
import argparse
import gzip
import json
import os
import subprocess
import sys
import threading
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np


def read_batches(path):
    with open(path) as f:
        return sorted(x.strip() for x in f if x.strip())        # Snakefile:31-33 sorts them too


def read_sizes(path):
    out = {}
    if path:
        with open(path) as f:
            for line in f:
                p, nbytes, _xz = line.split()
                out[p.rsplit("/", 1)[-1].replace(".cobs_classic.xz", "")] = int(nbytes)
    return out


def lpt(weights, n):
    order = sorted(range(len(weights)), key=lambda i: (-weights[i], i))
    load, parts = [0] * n, [[] for _ in range(n)]
    for i in order:
        r = min(range(n), key=lambda k: (load[k], k))
        parts[r].append(i)
        load[r] += weights[i]
    return [sorted(p) for p in parts]


class Admission:
    """HBM budget for decoded-but-not-yet-searched indexes.  Loaders are admitted strictly in
    submission order (a ticket counter): the consumer drains batches in that same order and a
    reservation is only returned after its batch was searched, so a later batch must never hold
    the budget an earlier one is still waiting for.  A batch larger than the whole budget is
    admitted once nothing else is resident."""

    def __init__(self, budget):
        self.budget, self.resident, self.next_ticket = float(budget), 0.0, 0
        self.aborted = False
        self.cv = threading.Condition()

    def acquire(self, ticket, need):
        with self.cv:
            while not self.aborted and (ticket != self.next_ticket or
                                        (self.resident > 0 and self.resident + need > self.budget)):
                self.cv.wait()
            if self.aborted:
                raise RuntimeError("stage aborted")
            self.resident += need
            self.next_ticket += 1
            self.cv.notify_all()

    def abort(self):
        """the consumer failed: nobody will return budget any more, waiting loaders must give up"""
        with self.cv:
            self.aborted = True
            self.cv.notify_all()

    def release(self, amount):
        with self.cv:
            self.resident = max(0.0, self.resident - amount)
            self.cv.notify_all()


def open_index_stream(cobs_dir, batch):
    """(file object, process or None): the plain index if it was decompressed
    already (Snakefile:364-387), else an xzcat pipe (run_cobs_streaming.sh:27)"""
    plain = os.path.join(cobs_dir, f"{batch}.cobs_classic")
    if os.path.exists(plain):
        return open(plain, "rb"), None
    xz = plain + ".xz"
    if not os.path.exists(xz):
        raise FileNotFoundError(xz)
    p = subprocess.Popen(["xzcat", "--no-sparse", "--ignore-check", xz], stdout=subprocess.PIPE)
    return p.stdout, p


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--batches", required=True)
    ap.add_argument("--cobs-dir", required=True)
    ap.add_argument("--sizes", default=None, help="data/decompressed_indexes_sizes.txt")
    ap.add_argument("--queries", required=True)
    ap.add_argument("--out-dir", required=True)
    ap.add_argument("--threshold", type=float, default=0.7)          # config.yaml:20
    ap.add_argument("--nb-best-hits", type=int, default=100)         # config.yaml:23
    ap.add_argument("--filter-out", default=None)
    ap.add_argument("--loaders", type=int, default=4, help="concurrent xz decoders per rank")
    ap.add_argument("--max-resident-gb", type=float, default=0.0, help="HBM budget for decoded-but-unsearched indexes (0 = 60%% of free)")
    args = ap.parse_args(argv)

    from . import _lib as pm

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("PHYLIGN_DIST_BACKEND", "nccl")
    if world > 1:                                                   # torch only for the multi-rank exchange
        import torch
        import torch.distributed as dist
        from .dist import gather_hits
        if os.environ.get("PHYLIGN_SHARE_GPU") or local_rank >= max(torch.cuda.device_count(), 1):
            # functional tests (several ranks, one GPU), or a launcher that narrowed the visible devices per rank
            local_rank %= max(torch.cuda.device_count(), 1)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    pm.init(local_rank)
    t_start = time.time()

    batches = read_batches(args.batches)
    sizes = read_sizes(args.sizes)
    parts = lpt([sizes.get(b, 1) for b in batches], world)
    mine = parts[rank]
    qfile = os.path.basename(args.queries)
    qfile = qfile[:-3] if qfile.endswith(".fa") else qfile
    os.makedirs(args.out_dir, exist_ok=True)
    with open(args.queries, "rb") as f:
        fasta = f.read()
    queries = pm.Queries(fasta, term_size=31)          # 661k indexes are 31-mer indexes; checked per index below
    nq, n_terms = queries.count()

    budget = args.max_resident_gb * 1e9 if args.max_resident_gb > 0 else 0.6 * pm.device_info()["hbm_free"]
    admit = Admission(budget)

    def load(ticket, pos):
        b = batches[pos]
        # what the loader may allocate at most: the line-aligned layout never needs more than twice
        # the file's bytes (a 65-byte row becomes 128), plus the two 32 MiB staging chunks
        need = 2.0 * float(sizes.get(b, 0)) + (128 << 20)
        admit.acquire(ticket, need)
        t0 = time.time()
        try:
            fobj, proc = open_index_stream(args.cobs_dir, b)
            try:
                ix = pm.Index.load_fd(fobj.fileno(), size_hint=sizes.get(b, 0))
            finally:
                fobj.close()
                if proc is not None and proc.wait() != 0:
                    raise RuntimeError(f"xzcat failed on batch {b}")
        except BaseException:
            admit.release(need)
            raise
        held = float(ix.info.device_bytes)
        admit.release(need - held)                      # keep only what the matrix really occupies
        return pos, ix, held, time.time() - t0

    from . import pgzip
    writers = ThreadPoolExecutor(max_workers=4)
    deflaters = ThreadPoolExecutor(max_workers=max(2, min(16, len(os.sched_getaffinity(0)))))

    def write_gz(path, text):
        # `gzip --fast` (Snakefile:468), deflated in parallel as consecutive gzip members
        pgzip.write(path, text, level=1, pool=deflaters)

    kept, names_of, log_rows, pending = [], {}, [], []
    nb = args.nb_best_hits

    def consume(futures):
        for fut in futures:
            pos, ix, need, t_load = fut.result()
            b = batches[pos]
            info = ix.info
            if ix.device != pm.bound_device():
                raise SystemExit(f"batch {b}: matrix is on GPU {ix.device}, this rank drives GPU {pm.bound_device()}")
            if info.term_size != 31:
                raise SystemExit(f"batch {b}: term_size {info.term_size} != 31")
            t0 = time.time()
            res = pm.search([ix], queries, args.threshold, slot_base=pos, nb_best_hits=max(nb, 0))
            hits = res.hits()
            ms = res.stats.ms_total
            res.free()
            text = pm.format_hits(ix, queries, hits, slot=pos, nb_best_hits=nb)
            pending.append(writers.submit(write_gz, os.path.join(args.out_dir, f"{b}____{qfile}.gz"), text))
            if args.filter_out:
                kept.append(hits)
                names_of[pos] = ix.names()
            ix.free()
            admit.release(need)
            log_rows.append({"batch": b, "load_s": round(t_load, 3), "gpu_ms": round(ms, 3), "hits": int(len(hits)),
                             "search_and_format_s": round(time.time() - t0, 3)})

    with ThreadPoolExecutor(max_workers=max(1, args.loaders)) as pool:
        futures = [pool.submit(load, ticket, pos) for ticket, pos in enumerate(mine)]
        try:
            consume(futures)
        except BaseException:
            admit.abort()                    # loaders waiting for budget would wait forever otherwise
            for f in futures:
                f.cancel()
            raise
    for p in pending:
        p.result()
    writers.shutdown()
    deflaters.shutdown()

    # ---- 04_filter: gather the pruned records (and names) to rank 0, merge natively
    if args.filter_out:
        local = np.concatenate(kept) if kept else np.zeros(0, dtype=pm.HIT_DTYPE)
        all_names = [names_of]
        allhits = local
        if world > 1:
            t = torch.from_numpy(local.view(np.int32).reshape(-1, 4).copy())
            if backend == "nccl":
                t = t.cuda()
            g = gather_hits(t, dst=0)
            all_names = [None] * world if rank == 0 else None
            dist.gather_object(names_of, all_names, dst=0)
            if rank == 0:
                allhits = g.cpu().numpy().view(pm.HIT_DTYPE).reshape(-1)
        if rank == 0:
            names = {}
            for d in all_names:
                names.update(d)
            m = pm.Merge(queries, keep=args.nb_best_hits)
            for pos in sorted(names):                        # file order of the consumer = sorted batches
                ix = pm.Index.from_names(names[pos])
                m.add(batches[pos], ix, allhits[allhits["slot"] == pos], slot=pos, nb_best_hits=nb)
            os.makedirs(os.path.dirname(os.path.abspath(args.filter_out)), exist_ok=True)
            tmp = args.filter_out + ".tmp"
            with open(tmp, "wb") as f:
                f.write(m.emit())
            os.replace(tmp, args.filter_out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    print(json.dumps({"rank": rank, "world": world, "batches": len(mine), "queries": nq, "kmers": n_terms,
                      "wall_s": round(time.time() - t_start, 3), "per_batch": log_rows}), file=sys.stderr)


if __name__ == "__main__":
    main()

"""Whole 03_match stage (and optionally 04_filter) for a list of phylogenetic
batches on one or several MI355X: the multi-batch / multi-GPU form of what the
reference runs as one Snakemake job per batch (Snakefile:431-487 rule
decompress_and_run_cobs, then Snakefile:490-520 rule translate_matches).

    python -m phylign_amd.match_stage --batches data/batches_full.txt --cobs-dir cobs \
        --sizes data/decompressed_indexes_sizes.txt --queries intermediate/01_queries_merged/Q.fa \
        --out-dir intermediate/03_match [--filter-out intermediate/04_filter/Q.fa]
    python -m phylign_amd.match_stage --input-dir input ...       # rules fix_query + concatenate_queries too: every query file of input/
    python -m phylign_amd.match_stage --gpus 8 ...                                          # starts its own 8 ranks, one per GPU
    python -m torch.distributed.run --nproc-per-node 8 ... -m phylign_amd.match_stage ...   # or under a launcher

Per rank (static batch -> rank map on scan cost, workload.assign_named):

  loaders    a pool of `xzcat` pipes decodes the rank's batches and streams them into HBM, admitted
             to an HBM budget in submission order;
  consumer   takes EVERY batch that is resident by now as one group and queues ONE fused search for
             it (pm_search_async: one scan launch per row-width class x counter-width class, whatever
             the number of batches -- the path bench.py measures); while the GPU scans group i + 1 the
             host finishes group i: per batch the cobs text after the post-filter
             (pm_format_hits(nb_best_hits)), `gzip --fast` members deflated in parallel, and the
             batch's records streamed into the native 04_filter merge (pm_merge_add) -- then the
             records and the matrix are dropped.  Host memory holds one group's records plus the
             merge state (12 bytes per kept match), never every batch's hit list.
  end        with --filter-out every rank exports what its merge kept (pm_merge_export), ONE gather
             (RCCL) brings the exports to rank 0, which adds them again and writes the FASTA
             (scripts/filter_queries.py semantics).

Every batch yields `<out-dir>/<batch>____<qfile>.gz` with exactly the bytes of
`run_cobs_streaming.sh ... | postprocess_cobs.py -n N | gzip` after gunzip.
The last line on stderr is a JSON report with match-only, format, gzip, merge and
end-to-end times.
"""
import argparse
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


class Admission:
    """HBM budget for decoded-but-not-yet-searched indexes.  Loaders are admitted strictly in
    submission order (a ticket counter) and a reservation is only returned after its batch was
    searched, so a later batch never holds the budget an earlier one is still waiting for.  A
    batch larger than the whole budget is admitted once nothing else is resident."""

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


def xz_decode_threads(pl, xz_threads):
    """threads the block-parallel decoder uses for a file with block table `pl` (None: not decodable that way -> 1)"""
    return 1 if pl is None else max(1, min(int(xz_threads), len(pl.blocks)))


def open_index_stream(cobs_dir, batch, cache_dir=None, xz_threads=1):
    """(file object, decoder or None): the plain index if it was decompressed
    already (Snakefile:364-387; in <cache_dir> or next to the .xz), else an xzcat pipe (run_cobs_streaming.sh:27) -- or,
    with xz_threads > 1 and a file of several independent blocks, the in-process block-parallel decoder (xzpar.py);
    the decoder has .wait() -> 0 on success"""
    for d in ([cache_dir] if cache_dir else []) + [cobs_dir]:
        plain = os.path.join(d, f"{batch}.cobs_classic")
        if os.path.exists(plain):
            return open(plain, "rb"), None
    xz = os.path.join(cobs_dir, f"{batch}.cobs_classic.xz")
    if not os.path.exists(xz):
        raise FileNotFoundError(xz)
    if xz_threads > 1:
        from . import xzpar
        pl = xzpar.plan(xz)
        if pl is not None:
            p = xzpar.ParallelXz(pl, xz_decode_threads(pl, xz_threads))
            return p.stdout, p
    p = subprocess.Popen(["xzcat", "--no-sparse", "--ignore-check", xz], stdout=subprocess.PIPE)
    return p.stdout, p


def xz_block_structure(paths):
    """`xz --robot --list` of the given .xz files (it reads only the stream footers and indexes): how many blocks each
    file holds.  One block (what plain `xz` writes, and what the reference's own data/*.xz are) can only be decoded by one
    thread; a multi-block file (`xz -T`) could be decoded by several.  The stage decodes one file per loader thread
    either way -- the image's xz 5.2.5 has no threaded decoder -- and reports what it saw, so that a run on the real
    661k indexes says whether in-file threading would have anything to work with.  None when xz cannot tell."""
    paths = [p_ for p_ in paths if os.path.exists(p_)]
    if not paths:
        return None
    try:
        out = subprocess.run(["xz", "--robot", "--list"] + paths, capture_output=True, text=True, timeout=120)
        blocks = [int(ln.split("\t")[2]) for ln in out.stdout.splitlines() if ln.startswith("file\t")]
    except (OSError, ValueError, IndexError, subprocess.SubprocessError):
        return None
    if len(blocks) != len(paths):
        return None
    return {"files": len(blocks), "multi_block_files": sum(1 for b in blocks if b > 1), "blocks_max": max(blocks), "blocks_total": sum(blocks)}


class FileSource:
    """indexes read from <cobs-dir>/<batch>.cobs_classic[.xz] (what the reference's rules read).

    cache_dir: the decode-once cache (the reference's index_load_mode mem-disk + keep_cobs_indexes, config.yaml:91-104,
    rule decompress_cobs Snakefile:364-387): a batch that had to be decoded from .xz leaves
    <cache_dir>/<batch>.cobs_classic behind (written while it streams into HBM: unnamed temporary + rename), and the next stage run
    reads that file with the parallel pread loader instead of decoding again."""

    def __init__(self, pm, cobs_dir, sizes, cache_dir=None, host_ram=None, host_mb=None):
        self.pm, self.cobs_dir, self.sizes, self.cache_dir = pm, cobs_dir, sizes, cache_dir
        # host-RAM admission of the xz decoders (sizing.HostRam / sizing.stage_plan: the reference's max_ram_gb)
        self.host_ram, self.host_mb = host_ram, host_mb or {}
        self.counts = {"xz_decoded": 0, "plain_files": 0, "cache_files_written": 0, "xz_decoded_block_parallel": 0}
        # threads ONE multi-block .xz file is decoded with (xzpar.py): set by the stage when the rank has fewer compressed
        # batches than CPUs (PHYLIGN_XZ_THREADS overrides; 1 = always xzcat)
        self.xz_threads = 1
        self._mu = threading.Lock()
        # plain files are read by the library's own pread workers (6 per file, PM_LOAD_THREADS), which move 40+ GB/s out of
        # the page cache by themselves: the loader threads -- as many as xz decoders would need, sizing.stage_plan -- take
        # turns at them, or 12 x 6 copy threads share the job's CPUs (measured on a 128 GB shard: profiles/r05/NOTES.md)
        self._plain_gate = threading.Semaphore(max(1, int(os.environ.get("PHYLIGN_PLAIN_LOADS", "2"))))
        if cache_dir:
            os.makedirs(cache_dir, exist_ok=True)

    def need(self, batch):
        # what the loader may allocate at most: the line-aligned layout never needs more than twice
        # the file's bytes (a 65-byte row becomes 128), plus the two 32 MiB staging chunks
        return 2.0 * float(self.sizes.get(batch, 0)) + (128 << 20)

    def is_compressed(self, batch):
        return not any(os.path.exists(os.path.join(d, f"{batch}.cobs_classic")) for d in ([self.cache_dir] if self.cache_dir else []) + [self.cobs_dir])

    def load(self, batch):
        # a decoder is admitted to the host-RAM budget before it starts (a plain file needs only the pooled staging)
        mb = self.host_mb.get(batch, 0) if (self.host_ram is not None and self.is_compressed(batch)) else 0
        if mb and self.xz_threads > 1:                  # a block-parallel decode holds that many decoders (and their blocks)
            from . import xzpar
            mb *= xz_decode_threads(xzpar.plan(os.path.join(self.cobs_dir, f"{batch}.cobs_classic.xz")), self.xz_threads)
        if mb:
            self.host_ram.acquire(mb)
        try:
            fobj, proc = open_index_stream(self.cobs_dir, batch, self.cache_dir, self.xz_threads)
            tee = os.path.join(self.cache_dir, f"{batch}.cobs_classic") if (self.cache_dir and proc is not None) else None
            ix = None
            try:
                if proc is None:
                    with self._plain_gate:
                        ix = self.pm.Index.load_fd(fobj.fileno(), size_hint=self.sizes.get(batch, 0), tee_path=tee)
                else:
                    ix = self.pm.Index.load_fd(fobj.fileno(), size_hint=self.sizes.get(batch, 0), tee_path=tee)
            finally:
                fobj.close()
                if proc is not None and proc.wait() != 0:
                    # whatever the decoder choked on is not a cache entry -- but only the file THIS load published is
                    # removed (another process decoding the same batch writes its own temporary and may have finished well)
                    if tee and ix is not None and getattr(ix, "cached", False) and os.path.exists(tee):
                        os.unlink(tee)
                    raise RuntimeError(f"xz decoding failed on batch {batch}")
        finally:
            if mb:
                self.host_ram.release(mb)
        with self._mu:
            self.counts["xz_decoded" if proc is not None else "plain_files"] += 1
            self.counts["xz_decoded_block_parallel"] += int(proc is not None and not isinstance(proc, subprocess.Popen))
            self.counts["cache_files_written"] += int(bool(tee and getattr(ix, "cached", False)))
        return ix


class SynthSource:
    """measurement / test aid: 661k-shaped synthetic signatures generated in HBM in place
    of files; `plant` maps a batch name to (rows, docs) bit positions to set"""

    def __init__(self, pm, shapes, seed=661, plant=None):
        self.pm, self.by_name, self.seed, self.plant = pm, {s.batch: s for s in shapes}, seed, plant or {}

    def need(self, batch):
        s = self.by_name[batch]
        return 2.0 * s.signature_size * s.row_bytes + (16 << 20)

    def load(self, batch):
        s = self.by_name[batch]
        ix = self.pm.Index.synth(s.batch_id, s.n_docs, s.signature_size, 1, 31, self.seed)
        if batch in self.plant:
            ix.plant(*self.plant[batch])
        return ix


class ResidentSource:
    """indexes that are in HBM already (a resident-index server, a measurement that keeps load time out
    of the timed region): the stage searches them as ONE group; they are not freed by the stage"""
    resident = True

    def __init__(self, indexes):
        self.indexes = dict(indexes)                   # batch name -> pm.Index

    def need(self, batch):
        return 0.0

    def load(self, batch):
        return self.indexes[batch]


def split_prepared_fasta(fasta, max_records, piece_bytes=0):
    """cuts a prepared query file (records start with '>' or ';' at a line start) into pieces of at most max_records
    records, and -- piece_bytes > 0 -- pieces of more than twice that many bytes further into pieces of about piece_bytes
    (cut at record starts): the pieces are parsed one after the other while the first ones are already searched.  Returns
    the list of pieces -- views of `fasta`, nothing is copied (one piece when there is nothing to cut)"""
    from . import _lib as pm
    cuts = pm.fasta_record_cuts(fasta, max_records) if max_records > 0 else []
    bounds = [0] + cuts + [len(fasta)]
    if piece_bytes > 0 and isinstance(fasta, (bytes, bytearray)):
        fine = [0]
        for a, b in zip(bounds[:-1], bounds[1:]):
            p = a
            while b - p > 2 * piece_bytes:
                hits = [x for x in (fasta.find(b"\n>", p + piece_bytes, b), fasta.find(b"\n;", p + piece_bytes, b)) if x >= 0]
                if not hits:
                    break
                p = min(hits) + 1
                fine.append(p)
            fine.append(b)
        bounds = fine
    if len(bounds) <= 2:
        return [fasta]
    view = memoryview(fasta)
    return [view[bounds[i]:bounds[i + 1]] for i in range(len(bounds) - 1)]


def run_stage(pm, batches, mine, source, queries, qfile, out_dir, threshold=0.7, nb_best_hits=100,
              want_merge=False, loaders=4, budget_bytes=None, max_group=0, keep_texts=None, kmer_size=31,
              query_reserve_bytes=0, write_match_files=True, load_order=None):
    """The per-rank pipeline described in the module docstring over the batches `mine` (positions into
    `batches`).  `queries` is one pm.Queries or a LIST of them (or of Futures of them: a file that is still being parsed):
    the chunks, in file order, of a query file with more reads than fit HBM at once, or of one cut up so that parsing and
    searching overlap -- every group of resident batches is then searched chunk after chunk (the pipeline's units are
    (group, chunk) pairs), a batch's file grows by one piece per chunk, and the ONE 04_filter merge is extended chunk by
    chunk (read names are one namespace through the whole file).  Returns (report dict, pm.Merge or None).  keep_texts: optional dict that receives {batch: post-filtered text}
    (tests)."""
    from . import pgzip
    t_start = time.perf_counter()
    os.makedirs(out_dir, exist_ok=True)
    chunks = list(queries) if isinstance(queries, (list, tuple)) else [queries]
    nc = len(chunks)
    nb = nb_best_hits
    # HBM for decoded-but-unsearched indexes: 60 % of what is free now, minus what the query chunks in flight will take
    # (query_reserve_bytes: two chunks are resident at a time -- the one being searched and the one queued behind it;
    # a chunk's HBM copies are released as soon as its unit is finished, so this does not grow with the file)
    budget = budget_bytes if budget_bytes else max(0.6 * pm.device_info()["hbm_free"] - query_reserve_bytes, 1.0)
    admit = Admission(budget)
    merge = [None]                                   # the ONE 04_filter merge of the query file, extended piece by piece
    # a chunk may arrive as a Future of its pm.Queries (the file is still being parsed): chunks are made ready in order --
    # parsed, and their merge state created -- by one thread that runs ahead of the searches
    prepared, prep_err = [threading.Event() for _ in range(nc)], []

    def prepare_all():
        for ci in range(nc):
            try:
                if hasattr(chunks[ci], "result"):
                    chunks[ci] = chunks[ci].result()
                if want_merge:
                    if ci == 0:
                        merge[0] = pm.Merge(chunks[0], keep=nb)
                    else:
                        merge[0].extend(chunks[ci])    # read names are one namespace through the whole file
            except BaseException as e:                   # noqa: BLE001 -- handed to the thread that asks for the chunk
                prep_err.append(e)
            prepared[ci].set()
            if prep_err:
                for ev in prepared[ci:]:
                    ev.set()
                return

    def chunk(ci):
        prepared[ci].wait()
        if prep_err:
            raise prep_err[0]
        return chunks[ci]
    threading.Thread(target=prepare_all, daemon=True).start()
    ready, ready_cv, failed = [], threading.Condition(), []
    acc = {"load_s": 0.0, "format_s": 0.0, "gzip_s": 0.0, "merge_s": 0.0, "match_only_s": 0.0, "gpu_wait_s": 0.0,
           "d2h_s": 0.0}
    acc_mu = threading.Lock()

    def add_time(key, dt):
        with acc_mu:
            acc[key] += dt

    def load(ticket, pos):
        b = batches[pos]
        need = source.need(b)
        try:
            admit.acquire(ticket, need)
            t0 = time.perf_counter()
            try:
                ix = source.load(b)
            except BaseException:
                admit.release(need)
                raise
            held = float(ix.info.device_bytes)
            admit.release(need - held)                  # keep only what the matrix really occupies
            add_time("load_s", time.perf_counter() - t0)
            item = (pos, ix, held)
        except BaseException as e:                      # the consumer re-raises it
            with ready_cv:
                failed.append(e)
                ready_cv.notify_all()
            return
        with ready_cv:
            ready.append(item)
            ready_cv.notify_all()

    def take_ready(block):
        """every batch that is resident by now (at most max_group), or [] when none is and block is False"""
        with ready_cv:
            while block and not ready and not failed:
                ready_cv.wait()
            if failed:
                raise failed[0]
            n = len(ready) if max_group <= 0 else min(len(ready), max_group)
            group = sorted(ready[:n])
            del ready[:n]
            return group

    deflaters = ThreadPoolExecutor(max_workers=max(2, min(16, len(os.sched_getaffinity(0)))))
    workers = ThreadPoolExecutor(max_workers=int(os.environ.get("PM_STAGE_WORKERS", "0")) or max(2, min(6, len(os.sched_getaffinity(0)) // 2)))
    group_rows = []

    def finish(group, ci, res, t_queued):
        """host half of a (group, chunk) unit: runs while the GPU scans the next one"""
        qc = chunk(ci)
        piece = 0 if nc == 1 else (1 if ci == 0 else (3 if ci == nc - 1 else 2))
        t0 = time.perf_counter()
        res.wait()
        st = res.stats
        t1 = time.perf_counter()
        add_time("gpu_wait_s", t1 - t0); add_time("match_only_s", st.ms_total * 1e-3)

        def one(i):
            pos, ix, _held = group[i]
            b = batches[pos]
            t_a = time.perf_counter()
            # the batch's records (ordered by query, best first) come back on their own, into a pooled pinned buffer:
            # the first worker also pays the ordering of the runs on the device
            with res.slot_hits(i) as sl:
                add_time("d2h_s", time.perf_counter() - t_a)
                return one_batch(i, b, ix, sl.hits)

        def one_batch(i, b, ix, part):
            path = os.path.join(out_dir, f"{b}____{qfile}.gz")
            ta = time.perf_counter()
            if not write_match_files:
                # --filter-only: the 04_filter FASTA is all that is wanted; the per-batch files (what makes the reference's
                # stage resumable, Snakefile:490-520) are not written, the records go straight into the merge
                text = None
                tb = tc = tc0 = time.perf_counter()
            elif keep_texts is None:
                # records -> post-filtered text -> `gzip --fast` members -> file, all inside the library (Snakefile:463-469)
                pm.format_hits_gz(ix, qc, part, path, slot=i, nb_best_hits=nb, level=1, piece=piece)
                text = None
                tb = tc = tc0 = time.perf_counter()
            else:
                text = pm.format_hits(ix, qc, part, slot=i, nb_best_hits=nb)
                tb = time.perf_counter()
                keep_texts[b] = keep_texts.get(b, b"") + text if ci else text
                if ci == nc - 1:
                    # `gzip --fast` (Snakefile:468), deflated in parallel as consecutive gzip members
                    pgzip.write(path, keep_texts[b], level=1, pool=deflaters)
                tc = tc0 = time.perf_counter()
            td = tc
            if want_merge:
                tc = time.perf_counter()                 # (the library serialises adds: waiting for another batch's add counts here)
                merge[0].add(b, ix, part, slot=i, nb_best_hits=nb, piece=ci)
                td = time.perf_counter()
            add_time("format_s", tb - ta); add_time("gzip_s", tc0 - tb); add_time("merge_s", td - tc)
            return len(part)
        n_rec = list(workers.map(one, range(len(group))))
        res.free()
        if nc > 1:
            # the chunk is searched again only with the next group: its HBM copies (sequences, 8 bytes per k-mer of hashes) go
            # back to the library's pool now (the next chunk takes them over), the host side stays for the texts and the
            # merge.  Waits for nothing: this unit's search has finished, the unit queued behind it uses another chunk, and
            # no hipFree -- which would wait for that queued scan -- is involved.
            qc.release_device()
        if ci == nc - 1:                                 # the group has seen every chunk: its matrices may go
            # (freed on the spot.  Keeping them until the end of the stage -- hipFree waits for the device -- was measured on
            # a 128 GB shard and is no faster; a process that exits holding its whole shard leaves the driver 135 GB of HBM
            # to scrub, which the next process's first large hipMalloc then waits 2.5 - 3.5 s for: profiles/r05/NOTES.md)
            for pos, ix, held in group:
                if not resident:
                    ix.free()
                admit.release(held)
        row = {"batches": [batches[p] for p, _, _ in group], "scan_launches": int(st.n_scan_launches),
               "gpu_ms": round(st.ms_total, 3), "records": int(sum(n_rec)),
               "queued_to_done_s": round(time.perf_counter() - t_queued, 3)}
        if nc > 1:
            row["chunk"] = ci
        group_rows.append(row)

    resident = bool(getattr(source, "resident", False))
    if resident:                                        # nothing to decode: every batch is ready, no loader threads
        ready.extend((pos, source.load(batches[pos]), 0.0) for pos in mine)
        if max_group <= 0 and len(mine) >= 16 and nc == 1:
            # four quarters instead of one group: the host part of a quarter (text, gzip, merge) overlaps the scans of the
            # next ones and only the last quarter's is left at the end (measured on 38 batches, 1 M reads: stage wall 0.22 /
            # 0.17 / 0.17 / 0.15 s with 2 / 3 / 4 / 8 groups, match-only 0.131 -> 0.137 s: profiles/r03/NOTES.md section 6)
            max_group = (len(mine) + 3) // 4
    with ThreadPoolExecutor(max_workers=max(1, loaders)) as pool:
        # load_order: the order in which the loaders take the batches (a permutation of `mine`; default: as listed).  The
        # stage passes "largest compressed index first": the decoders are the slow part of a cold run, and with the big
        # files started first the last ones to finish are small (longest-processing-time-first over the loader threads)
        order = list(mine) if load_order is None else list(load_order)
        assert sorted(order) == sorted(mine)
        futures = [] if resident else [pool.submit(load, ticket, pos) for ticket, pos in enumerate(order)]
        try:
            left, pending, backlog = len(mine), None, []
            while left or backlog or pending:
                if not backlog and left:
                    group = take_ready(block=pending is None)
                    if group:
                        for pos, ix, _ in group:
                            info = ix.info
                            if ix.device != pm.bound_device():
                                raise SystemExit(f"batch {batches[pos]}: matrix is on GPU {ix.device}, this rank drives GPU {pm.bound_device()}")
                            if info.term_size != kmer_size:
                                raise SystemExit(f"batch {batches[pos]}: term_size {info.term_size} != {kmer_size} (--kmer-size)")
                        backlog = [(group, ci) for ci in range(nc)]
                        left -= len(group)
                cur = None
                if backlog:
                    group, ci = backlog.pop(0)
                    tq = time.perf_counter()
                    cur = (group, ci, pm.search_async([ix for _, ix, _ in group], chunk(ci), threshold, nb_best_hits=max(nb, 0)), tq)
                if pending:
                    finish(*pending)
                pending = cur
        except BaseException:
            admit.abort()                    # loaders waiting for budget would wait forever otherwise
            for f in futures:
                f.cancel()
            raise
    workers.shutdown()
    deflaters.shutdown()
    if nc > 1:
        pm.set_option("release_query_pool", 1)     # the chunks' pooled device buffers: nothing is queued any more, really free them
    nq = sum(chunk(ci).count()[0] for ci in range(nc))
    n_terms = sum(chunk(ci).count()[1] for ci in range(nc))
    report = {"batches": len(mine), "queries": nq, "kmers": n_terms, "query_chunks": nc, "groups": len(group_rows),
              "scan_launches": sum(g["scan_launches"] for g in group_rows),
              "match_only_s": round(acc["match_only_s"], 4), "gpu_wait_s": round(acc["gpu_wait_s"], 4),
              "d2h_s": round(acc["d2h_s"], 4), "load_s_thread_sum": round(acc["load_s"], 3),
              "format_s_thread_sum": round(acc["format_s"], 3), "gzip_s_thread_sum": round(acc["gzip_s"], 3),
              "format_and_gzip_in_library": keep_texts is None,
              "merge_s_thread_sum": round(acc["merge_s"], 3), "stage_wall_s": round(time.perf_counter() - t_start, 3),
              "query_hbm_bytes_at_end": int(sum(chunk(ci).device_bytes()[0] for ci in range(nc))),
              "per_group": group_rows, "merge_order": merge[0].batches() if want_merge else [],
              "index_source": dict(getattr(source, "counts", {}), resident=len(mine) if resident else 0)}
    return report, merge[0]


QUERY_EXTENSIONS = ("fa", "fasta", "fq", "fastq")               # Snakefile:13


def query_stem(path):
    """a query file's name in the pipeline: its base name without the last suffix (Snakefile:28-29)"""
    base = os.path.basename(path)
    return base.rsplit(".", 1)[0] if "." in base else base


def discover_queries(input_dir):
    """the files of <input_dir> the reference takes as query files (Snakefile:24-29: input/*.{fa,fasta,fq,fastq})"""
    import glob
    return [p for ext in QUERY_EXTENSIONS for p in glob.glob(os.path.join(input_dir, f"*.{ext}"))]


def merged_queries(pm, paths, raw, kmer_size):
    """rules fix_query + concatenate_queries (Snakefile:314-352) for several query files: every file is prepared on its
    own (raw: the native parser with normalise -- seqtk seq -A -U -C | awk gsub(/[^ACGT]/, "A"); else it is taken as a
    file of intermediate/00_queries_preprocessed/), the prepared texts follow each other in the order of the sorted
    file names, and the merged file is called after all of them joined by "___" (Snakefile:28-38).
    Returns (name of the merged query file, its bytes)."""
    by_stem = {}
    for p in paths:
        if query_stem(p) in by_stem:
            raise SystemExit(f"two query files are called '{query_stem(p)}': {by_stem[query_stem(p)]} and {p} (Snakefile:309-311)")
        by_stem[query_stem(p)] = p
    texts = []
    for stem in sorted(by_stem):
        with open(by_stem[stem], "rb") as f:
            data = f.read()
        if raw:
            q = pm.Queries(data, term_size=kmer_size, normalise=True)
            data = q.fasta()
            q.free()
        texts.append(data)
    return "___".join(sorted(by_stem)), b"".join(texts)


def apply_reference_config(args, ap):
    """--config: the stage takes its parameters from the reference's own config.yaml and directory layout, i.e. it is the
    reference's `match` target (Snakefile:249-253) run from the pipeline's directory (--workdir, default: the directory of
    the config file).  Command-line flags given explicitly win.  Keys read: batches, cobs_kmer_thres, nb_best_hits,
    download_dir, index_load_mode, decompression_dir, keep_cobs_indexes, max_ram_gb (config.yaml; Snakefile:126-175)."""
    import yaml
    with open(args.config) as f:
        cfg = yaml.safe_load(f) or {}
    wd = args.workdir or os.path.dirname(os.path.abspath(args.config))

    def in_wd(p_):
        return p_ if os.path.isabs(p_) else os.path.join(wd, p_)
    mode = str(cfg.get("index_load_mode", "mem-stream"))
    if mode not in ("mem-stream", "mem-disk", "mmap-disk"):
        ap.error(f"config: index_load_mode must be one of mem-stream, mem-disk, mmap-disk (Snakefile:124-131), not {mode!r}")
    given = set(args.given)
    if "batches" not in given:
        if "batches" not in cfg:
            ap.error(f"{args.config}: no `batches` key (config.yaml:9; or give --batches)")
        args.batches = in_wd(str(cfg["batches"]))                                        # Snakefile:32-34
    if "cobs_dir" not in given:
        args.cobs_dir = in_wd(os.path.join(str(cfg.get("download_dir", ".")), "cobs"))  # Snakefile:151
    if "sizes" not in given:
        args.sizes = in_wd("data/decompressed_indexes_sizes.txt")                        # Snakefile:373
    if "threshold" not in given:
        args.threshold = float(cfg.get("cobs_kmer_thres", args.threshold))               # Snakefile:410
    if "nb_best_hits" not in given:
        args.nb_best_hits = int(cfg.get("nb_best_hits", args.nb_best_hits))              # Snakefile:412
    if "max_ram_gb" not in given and "max_ram_gb" in cfg:
        args.max_ram_gb = float(cfg["max_ram_gb"])                                       # Snakefile:109
    if "index_load_mode" not in given:
        args.index_load_mode = mode
    if "decompression_dir" not in given:
        args.decompression_dir = in_wd(str(cfg.get("decompression_dir", "intermediate/02_cobs_decompressed")))   # Snakefile:152-154
    # mem-stream never decompresses to disk; in the disk modes a decompressed index is temp() unless keep_cobs_indexes
    # (Snakefile:155-175, :366-370): without it nothing needs to be written at all, the matrix goes straight to HBM
    args.keep_cobs_indexes = bool(cfg.get("keep_cobs_indexes", False)) and args.index_load_mode != "mem-stream"
    if "queries" not in given and "input_dir" not in given:
        args.input_dir = in_wd("input")                                                  # Snakefile:24-25
    if "out_dir" not in given:
        args.out_dir = in_wd("intermediate/03_match")                                    # Snakefile:394
    args.filter_dir = in_wd("intermediate/04_filter")                                    # Snakefile:497 (named after the merged query file)


def bind_rank_to_gpu(local_rank, n_visible):
    """one rank per GPU; several ranks may share a device only when the launcher narrowed the visible
    devices to one per rank (HIP_VISIBLE_DEVICES) or PHYLIGN_SHARE_GPU is set (functional tests)"""
    if local_rank < n_visible:
        return local_rank
    if n_visible == 1 or os.environ.get("PHYLIGN_SHARE_GPU"):
        return local_rank % max(n_visible, 1)
    raise SystemExit(f"local rank {local_rank} but only {n_visible} GPUs are visible: launch one rank per GPU "
                     "(or set PHYLIGN_SHARE_GPU=1 to let ranks share devices)")


def main(argv=None):
    ap = argparse.ArgumentParser(allow_abbrev=False, description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--batches", default=None)
    ap.add_argument("--cobs-dir", default=None)
    ap.add_argument("--sizes", default=None, help="data/decompressed_indexes_sizes.txt")
    ap.add_argument("--queries", nargs="+", default=None,
                    help="the merged query file (intermediate/01_queries_merged/Q.fa), or several query files: they are "
                         "prepared one by one (with --raw-queries) and concatenated in the order of their sorted names, the "
                         "outputs are called after all of them joined by ___ (rules fix_query + concatenate_queries)")
    ap.add_argument("--input-dir", default=None,
                    help="instead of --queries: every *.fa / *.fasta / *.fq / *.fastq of this directory (the reference's "
                         "input/, Snakefile:24-29); implies --raw-queries")
    ap.add_argument("--out-dir", default=None)
    ap.add_argument("--config", default=None,
                    help="the reference's config.yaml: batches, cobs_kmer_thres, nb_best_hits, download_dir, index_load_mode, "
                         "decompression_dir, keep_cobs_indexes and max_ram_gb are taken from it, queries from input/, outputs go to "
                         "intermediate/03_match and intermediate/04_filter/<merged name>.fa -- the reference's `match` target "
                         "(Snakefile:249-253).  Flags given on the command line win")
    ap.add_argument("--workdir", default=None, help="with --config: the pipeline's directory (default: where the config file is)")
    ap.add_argument("--threshold", type=float, default=0.7)          # config.yaml:20
    ap.add_argument("--nb-best-hits", type=int, default=100)         # config.yaml:23
    ap.add_argument("--filter-out", default=None)
    ap.add_argument("--filter-only", action="store_true",
                    help="with --filter-out: do not write the per-batch intermediate/03_match/*.gz files, only the 04_filter FASTA "
                         "(the records go from the GPU straight into the merge).  Off by default: the files are what makes the "
                         "reference's pipeline resumable and what its rule translate_matches reads (Snakefile:490-520)")
    ap.add_argument("--loaders", type=int, default=0,
                    help="concurrent xz decoders per rank (0 = the CPUs the job may use minus 4, at least 4, at most 16 -- one "
                         "xz stream decodes 0.1-0.2 GB/s on one core, and decoding is what a cold stage waits for -- and no "
                         "more than fit --max-ram-gb at the decoder size the sizes table states: sizing.stage_plan)")
    ap.add_argument("--max-ram-gb", type=float, default=0.0,
                    help="config.yaml max_ram_gb: host RAM the stage's xz decoders may hold together on this node, all ranks "
                         "(the sizes table's third column per decoder, Snakefile:64-69; the index itself is in HBM).  "
                         "0 = 80 %% of the RAM available now")
    ap.add_argument("--cache-dir", default=None,
                    help="decode-once cache: a batch decoded from .xz is also written to <dir>/<batch>.cobs_classic (unnamed temporary + rename) "
                         "while it streams into HBM; later runs load that file (parallel pread, tens of GB/s) instead of "
                         "decoding again.  The reference's index_load_mode mem-disk with keep_cobs_indexes (config.yaml:91-104)")
    ap.add_argument("--index-load-mode", default="mem-stream", choices=["mem-stream", "mem-disk", "mmap-disk"],
                    help="config.yaml:91-104.  mem-stream: decode straight into HBM (with --cache-dir the decoded bytes are kept "
                         "too); mem-disk: the same plus the cache in --decompression-dir (required); mmap-disk has no "
                         "meaning for a matrix that lives in HBM and is taken as mem-disk")
    ap.add_argument("--decompression-dir", default=None, help="config.yaml decompression_dir (mem-disk): where decoded indexes are kept")
    ap.add_argument("--max-resident-gb", type=float, default=0.0, help="HBM budget for decoded-but-unsearched indexes (0 = 60%% of free)")
    ap.add_argument("--query-piece-mb", type=int, default=48,
                    help="a prepared query file of more than twice this many MB is parsed and searched in pieces of about this "
                         "size: the first pieces are searched while the later ones are parsed (0 = off)")
    ap.add_argument("--query-chunk", type=int, default=4_000_000,
                    help="most reads searched at once (0 = the whole file): a query set lives in HBM with 8 bytes per k-mer, so a "
                         "file of tens of millions of reads is searched chunk after chunk against the resident batches")
    ap.add_argument("--kmer-size", type=int, default=31, help="k of the indexes (31 for the 661k collection); every batch's header is checked against it")
    ap.add_argument("--max-group", type=int, default=0, help="most batches fused into one search (0 = every resident batch)")
    ap.add_argument("--raw-queries", action="store_true",
                    help="--queries is an unprocessed FASTA/FASTQ (multi-line, lower case, IUPAC codes): apply rule "
                         "fix_query (Snakefile:314-333) in the native parser instead of seqtk + awk")
    ap.add_argument("--synthetic", default=None, metavar="WORKLOAD[:WORLD:RANK]",
                    help="measurement aid: 661k-shaped synthetic signatures generated in HBM instead of --cobs-dir files "
                         "(workload.select name; WORLD:RANK = hold only that shard of a WORLD-way split)")
    ap.add_argument("--gpus", type=int, default=0,
                    help="ranks (one per GPU) to start when the command is not run under a launcher: a plain "
                         "`python -m phylign_amd.match_stage --gpus 8 ...` starts 8 fresh ranks of itself (0 = what the launcher "
                         "says, or one)")
    args = ap.parse_args(argv)
    # which options the command line really set (so that --config only fills in the others)
    argv_seen = sys.argv[1:] if argv is None else list(argv)
    args.given = {a.dest for a in ap._actions if any(tok == o or tok.startswith(o + "=") for o in a.option_strings for tok in argv_seen)}
    args.keep_cobs_indexes, args.filter_dir = True, None
    if args.config:
        apply_reference_config(args, ap)
    if not args.out_dir:
        ap.error("--out-dir is required (or --config)")
    if args.filter_only and not args.filter_out and not args.config:
        ap.error("--filter-only needs --filter-out")
    from . import launch
    if not args.queries and not args.input_dir:
        ap.error("--queries or --input-dir is required")
    if launch.wants_self_launch(args.gpus):
        # this process has not touched the GPU: start the ranks as children, relay their status
        sys.exit(launch.self_launch_module("phylign_amd.match_stage", sys.argv[1:] if argv is None else argv, args.gpus))
    launch.arm_parent_death_signal()          # a rank of that launcher: ends with it

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("PHYLIGN_DIST_BACKEND", "nccl")
    # PHYLIGN_FORCE_DIST=1: one rank walks the multi-rank exchange over a real process group of one (RCCL on device
    # tensors): the plumbing of an 8-GPU run on the one GPU a test box has
    multi = world > 1 or bool(os.environ.get("PHYLIGN_FORCE_DIST"))
    if multi:                                                       # torch only for the multi-rank exchange
        import torch
        import torch.distributed as dist
        from .dist import gather_hits
        local_rank = bind_rank_to_gpu(local_rank, torch.cuda.device_count())
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", str(launch.free_port()))
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    from . import _lib as pm
    from . import workload as W
    pm.init(local_rank)
    t_start = time.perf_counter()

    if args.synthetic:
        spec = args.synthetic.split(":")
        shapes = W.select(spec[0])
        if len(spec) == 3:                                           # one shard of an emulated split
            shapes = [shapes[i] for i in W.assign_batches(shapes, int(spec[1]))[int(spec[2])]]
        batches = sorted(s.batch for s in shapes)
        source = SynthSource(pm, shapes)
        sizes = {s.batch: s.index_bytes for s in shapes}
    else:
        if not args.batches or not args.cobs_dir:
            ap.error("--batches and --cobs-dir are required (unless --synthetic)")
        batches = read_batches(args.batches)
        sizes = read_sizes(args.sizes)
        cache_dir = args.cache_dir
        if args.index_load_mode != "mem-stream" and (args.keep_cobs_indexes or not args.config):
            cache_dir = cache_dir or args.decompression_dir
            if not cache_dir:
                ap.error("--index-load-mode mem-disk needs --decompression-dir (or --cache-dir)")
        source = FileSource(pm, args.cobs_dir, sizes, cache_dir=cache_dir)
    parts = W.assign_named(batches, sizes, world)
    mine = parts[rank]
    if args.input_dir:
        qpaths = discover_queries(args.input_dir)
        args.raw_queries = True
        if not qpaths:
            ap.error(f"no query file (*.fa, *.fasta, *.fq, *.fastq) in {args.input_dir}")
    else:
        qpaths = list(args.queries)
    prepared = None
    if len(qpaths) > 1 or args.input_dir:
        # rules fix_query + concatenate_queries: the merged, prepared file is built here and searched like one
        qfile, fasta = merged_queries(pm, qpaths, args.raw_queries, args.kmer_size)
    else:
        qfile = os.path.basename(qpaths[0])
        qfile = qfile[:-3] if qfile.endswith(".fa") else qfile
        with open(qpaths[0], "rb") as f:
            fasta = f.read()
        # a query file with more reads than --query-chunk is searched in pieces: the chunks share the resident batches of a
        # group, every batch's file grows by one piece per chunk, and the 04_filter FASTA is emitted chunk after chunk
        if args.raw_queries:
            prepared = pm.Queries(fasta, term_size=args.kmer_size, normalise=True)
            if prepared.count()[0] > args.query_chunk > 0:
                fasta = prepared.fasta()                               # the prepared single-line form can be cut at '>' lines
                prepared.free()
                prepared = None
    if args.filter_dir and not args.filter_out:
        args.filter_out = os.path.join(args.filter_dir, f"{qfile}.fa")
    pieces = [fasta] if prepared is not None else split_prepared_fasta(fasta, args.query_chunk, args.query_piece_mb << 20)
    parser = ThreadPoolExecutor(max_workers=1)
    if prepared is not None:
        chunk_list = [prepared]
    else:
        # parsed one after the other by a thread of their own (the pieces are views of `fasta`): the stage searches the
        # first pieces while the later ones are still text; term_size is checked against every batch's header
        chunk_list = [parser.submit(pm.Queries, p_, args.kmer_size) for p_ in pieces]
    budget = args.max_resident_gb * 1e9 if args.max_resident_gb > 0 else None
    # a query set holds about 9 bytes of HBM per base while it is searched (sequence + 8 bytes of hash per k-mer); two chunks
    # are resident at a time
    reserve = 9 * sum(sorted((len(p_) for p_ in pieces), reverse=True)[:2])
    del pieces
    # loaders and their host RAM by the reference's sizing rules (Snakefile:60-121 -> sizing.py): the decoder size of every
    # batch of this rank from the sizes table, the budget from --max-ram-gb
    from . import sizing
    from .sysinfo import effective_cpus, available_ram_gb
    # ranks of one node share its CPUs and its RAM: every rank plans with its share (LOCAL_WORLD_SIZE from the launcher;
    # --max-ram-gb, like config.yaml's max_ram_gb, is the budget of the whole job on this node)
    local_world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", str(world)) or world))
    max_ram_gb = (args.max_ram_gb if args.max_ram_gb > 0 else 0.8 * available_ram_gb()) / local_world
    args.loaders, budget_mb, host_mb = sizing.stage_plan([batches[p_] for p_ in mine], args.sizes,
                                                        max(1, effective_cpus() // local_world), max_ram_gb, args.loaders)
    host_ram, load_order = None, None
    if isinstance(source, FileSource):
        host_ram = source.host_ram = sizing.HostRam(budget_mb)
        source.host_mb = host_mb
        # fewer compressed batches than CPUs (data/batches_small.txt: three): the CPUs left over decode the blocks of a
        # multi-block file side by side (xzpar.py); with as many batches as CPUs one xzcat per loader already uses them all
        n_xz = sum(1 for p_ in mine if source.is_compressed(batches[p_]))
        my_cpus = max(1, effective_cpus() // local_world)
        source.xz_threads = int(os.environ.get("PHYLIGN_XZ_THREADS", "0")) or (max(1, min(8, my_cpus // n_xz)) if n_xz else 1)
        if source.xz_threads > 1:
            args.loaders = max(1, min(args.loaders, max(1, my_cpus // source.xz_threads)))
        # compressed batches first, the largest first (the sizes table knows them); plain files behind them in list order
        load_order = sorted(mine, key=lambda p_: (0, -int(sizes.get(batches[p_], 0)), p_) if source.is_compressed(batches[p_]) else (1, 0, p_))
    report, merge = run_stage(pm, batches, mine, source, chunk_list, qfile, args.out_dir, args.threshold, args.nb_best_hits,
                              want_merge=bool(args.filter_out), write_match_files=not args.filter_only, loaders=args.loaders, budget_bytes=budget,
                              max_group=args.max_group, kmer_size=args.kmer_size, query_reserve_bytes=reserve, load_order=load_order)
    parser.shutdown()
    del fasta
    if isinstance(source, FileSource) and report["index_source"].get("xz_decoded"):
        report["index_source"]["xz_blocks"] = xz_block_structure(
            [os.path.join(source.cobs_dir, f"{batches[pos]}.cobs_classic.xz") for pos in mine])

    # ---- 04_filter: ONE gather of what every rank's merge kept (queries numbered through the whole file, slot = the
    # batch's number in that rank's merge), rank 0 adds the parts and emits
    t_f = time.perf_counter()
    fph = {"export": 0.0, "gather": 0.0, "names": 0.0, "add": 0.0, "emit": 0.0}     # where the 04_filter end of the stage spends its time
    if args.filter_out:
        if multi:
            t_a = time.perf_counter()
            ex = merge.export()
            t = torch.from_numpy(ex.view(np.int32).reshape(-1, 4).copy())
            if backend == "nccl":
                t = t.cuda()
            t_b = time.perf_counter()
            g = gather_hits(t, dst=0)
            meta = [None] * world if rank == 0 else None
            dist.gather_object((len(ex), report["merge_order"]), meta, dst=0)
            t_c = time.perf_counter()
            fph["export"], fph["gather"] = t_b - t_a, t_c - t_b
            if rank == 0:
                allrec = g.cpu().numpy().view(pm.HIT_DTYPE).reshape(-1)
                off = 0
                for r, (n, r_order) in enumerate(meta):
                    part = allrec[off:off + n]
                    off += n
                    if r == 0:
                        continue                                  # rank 0's own matches are in `merge` already
                    cut = np.searchsorted(part["slot"], np.arange(len(r_order) + 1, dtype=np.uint32))
                    for k, b in enumerate(r_order):
                        if cut[k + 1] > cut[k]:
                            t_n = time.perf_counter()
                            nix = names_index(pm, source, b)
                            t_m = time.perf_counter()
                            merge.add(b, nix, part[cut[k]:cut[k + 1]], slot=k, nb_best_hits=-1, piece=-1)
                            nix.free()
                            fph["names"] += t_m - t_n
                            fph["add"] += time.perf_counter() - t_m
        if rank == 0:
            t_e = time.perf_counter()
            os.makedirs(os.path.dirname(os.path.abspath(args.filter_out)), exist_ok=True)
            report["filter_fasta_bytes"] = merge.emit_to(args.filter_out)
            fph["emit"] = time.perf_counter() - t_e
    report["filter_phases_s"] = {k: round(v, 4) for k, v in fph.items()}
    report["filter_emit_s"] = round(time.perf_counter() - t_f, 3)
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    report["host_ram_plan"] = {"loaders": args.loaders, "budget_mb": budget_mb,
                               "decoder_mb_max": max(host_mb.values()) if host_mb else 0,
                               "decoders_peak_mb": host_ram.peak if host_ram is not None else 0}
    report.update({"rank": rank, "world": world, "e2e_s": round(time.perf_counter() - t_start, 3)})
    sys.stderr.write(json.dumps(report) + "\n")          # one write: the lines of several ranks must not interleave
    sys.stderr.flush()


def names_index(pm, source, batch):
    """names-only handle of a batch this rank did not search (rank 0 turning gathered records into references)"""
    if isinstance(source, SynthSource):
        s = source.by_name[batch]
        return pm.Index.synth(s.batch_id, s.n_docs, s.signature_size, 1, 31, source.seed, header_only=True)
    fobj, proc = open_index_stream(source.cobs_dir, batch, getattr(source, "cache_dir", None))
    try:
        head = bytearray()
        while True:                                                   # the header ends with the closing magic: read until it parses
            chunk = fobj.read(1 << 20)
            head += chunk
            try:
                return pm.Index.load_header_mem(bytes(head))
            except pm.PMError:
                if not chunk:
                    raise
    finally:
        fobj.close()
        if proc is not None:
            proc.kill()
            proc.wait()


if __name__ == "__main__":
    main()

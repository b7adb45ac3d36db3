"""Persistent index residency: a per-GPU server process that
keeps decoded signature matrices in HBM across query sets, so a 03_match job
costs a socket round trip plus the scan instead of xz decoding + PCIe upload +
process start per batch (what index_load_mode / keep_cobs_indexes approximate on
disk in the reference: config.yaml:91-104, :134; Snakefile:163-175).

    python -m phylign_amd.server --socket /tmp/phylign_match.sock [--device 0] [--max-gb 250]
                                 [--coalesce-ms 5] [--preload batches.txt --cobs-dir cobs]

Built for what Snakemake does to it -- hundreds of per-batch jobs at once (Snakefile:431-487 x 305):

  * every connection has its own handler thread; a cold index (minutes of single-threaded xz
    decoding) is loaded by the handler that asked for it, searches on resident batches go on meanwhile;
    a second request for an index that is being loaded waits for that load, it does not start another;
  * searches are queued to ONE dispatcher thread that takes everything waiting (plus what arrives within
    --coalesce-ms) and turns the jobs that share the query file, threshold and post-filter into ONE fused
    pm_search over their batches (one scan launch per row-width class: the path bench.py measures);
    each handler then formats and sends its own batch's text;
  * `preload` fills the cache in the background (e.g. a rank's share of batches_full.txt);
  * the HBM budget is accounted in real sizes: the uncompressed size from the client (--index-sizes), else
    from the .xz container's own index (`xz --robot --list`), x 2 while loading (line-aligned rows), the
    matrix's true device bytes afterwards; indexes in use are never evicted.

Protocol (unix stream socket, one request per connection): a JSON line, then for op=query the
server answers with a JSON line {"ok":..., "len": N, ...} followed by N bytes of cobs /
post-filtered text.  Client: cobs_query.py --server.
"""
import argparse
import hashlib
import json
import os
import queue
import socket
import stat
import subprocess
import sys
import threading
import time
from collections import OrderedDict


def uncompressed_size(path):
    """bytes the index occupies once decoded: the file size, or for .xz the size its container index records"""
    if not path.endswith(".xz"):
        return os.stat(path).st_size
    try:
        out = subprocess.run(["xz", "--robot", "--list", path], capture_output=True, text=True, timeout=60).stdout
        for line in out.splitlines():
            f = line.split("\t")
            if f[0] == "totals" or f[0] == "file":
                return int(f[4])
    except Exception:
        pass
    return os.stat(path).st_size * 12          # xz shrinks these indexes ~10 x (data/decompressed_indexes_sizes.txt)


class IndexCache:
    """LRU of resident indexes keyed by (realpath, mtime_ns, size); loads run in the calling thread, several at a
    time; entries are pinned while a search uses them"""

    def __init__(self, pm, max_bytes):
        self.pm, self.max_bytes = pm, max_bytes
        self.items = OrderedDict()          # key -> {"ix", "bytes", "users", "ready": Event, "error"}
        self.mu = threading.Condition()
        self.loads = self.hits = self.waits = 0
        self.cold_active = 0                # .xz loads being decoded right now (they share the host's CPUs)

    def used(self):
        return sum(e["bytes"] for e in self.items.values())

    def _make_room(self, need):
        """called with self.mu held: evict idle entries, least recently used first, until `need` fits"""
        while self.used() + need > self.max_bytes:
            victim = next((k for k, e in self.items.items() if e["users"] == 0 and e["ready"].is_set() and e["ix"] is not None), None)
            if victim is None:
                if not any(e["users"] or not e["ready"].is_set() for e in self.items.values()):
                    return                     # nothing left to evict: a single index larger than the budget is admitted
                self.mu.wait(0.05)             # wait for a search or a load to finish
                continue
            self.items.pop(victim)["ix"].free()

    def acquire(self, path, size_hint=0):
        """(Index, was_resident): the entry stays pinned until release(path_key)"""
        st = os.stat(path)
        key = (os.path.realpath(path), st.st_mtime_ns, st.st_size)
        if not stat.S_ISREG(st.st_mode):               # a pipe: its times move while it is written
            key = (os.path.realpath(path), 0, 0)
        loader = False
        with self.mu:
            e = self.items.get(key)
            if e is None:
                need = 2.0 * float(size_hint or uncompressed_size(path)) + (128 << 20)
                self._make_room(need)
                e = self.items.get(key)                 # _make_room may have waited: somebody else may have started this load
            if e is not None:
                self.items.move_to_end(key)
                e["users"] += 1
                if e["ready"].is_set():
                    self.hits += 1
                else:
                    self.waits += 1
            else:
                e = {"ix": None, "bytes": need, "users": 1, "ready": threading.Event(), "error": None}
                self.items[key] = e
                self.loads += 1
                loader = True
        if not loader:
            e["ready"].wait()
            if e["error"] is not None:
                self.release(key)
                raise e["error"]
            return e["ix"], key, True
        try:
            if path.endswith(".xz"):
                # cold loads share the host's CPUs: with few of them in flight a file of several blocks (`xz -T`) is decoded
                # block-parallel in-process (xzpar.py), else -- and for one-block files -- by one xzcat each
                from . import xzpar
                from .sysinfo import effective_cpus
                with self.mu:
                    self.cold_active += 1
                    threads = int(os.environ.get("PHYLIGN_XZ_THREADS", "0")) or max(1, min(8, effective_cpus() // self.cold_active))
                try:
                    # (plan() never opens anything but a regular file: a pipe named *.xz goes to xzcat with its header intact)
                    pl = xzpar.plan(path) if threads > 1 and stat.S_ISREG(st.st_mode) else None
                    if pl is not None:
                        p = xzpar.ParallelXz(pl, min(threads, len(pl.blocks)))
                    else:
                        p = subprocess.Popen(["xzcat", "--no-sparse", "--ignore-check", path], stdout=subprocess.PIPE)
                    ix = None
                    try:
                        ix = self.pm.Index.load_fd(p.stdout.fileno(), size_hint=size_hint)
                    finally:
                        p.stdout.close()
                        rc = p.wait()
                finally:
                    with self.mu:
                        self.cold_active -= 1
                if rc != 0:
                    if ix is not None:
                        ix.free()
                    raise RuntimeError(f"xz decoding failed on {path}")
            else:
                ix = self.pm.Index.load_file(path, size_hint=size_hint)
        except BaseException as err:
            with self.mu:
                e["error"] = err if isinstance(err, Exception) else RuntimeError(str(err))
                e["bytes"] = 0
                self.items.pop(key, None)
                e["ready"].set()
                self.mu.notify_all()
            raise
        with self.mu:
            e["ix"], e["bytes"] = ix, float(ix.info.device_bytes)
            e["ready"].set()
            self.mu.notify_all()
        return ix, key, False

    def release(self, key):
        with self.mu:
            e = self.items.get(key)
            if e is not None:
                e["users"] = max(0, e["users"] - 1)
            self.mu.notify_all()

    def drop(self, path=None):
        with self.mu:
            for key in list(self.items):
                e = self.items[key]
                if (path is None or key[0] == os.path.realpath(path)) and e["users"] == 0 and e["ready"].is_set():
                    self.items.pop(key)
                    if e["ix"] is not None:
                        e["ix"].free()


class Dispatcher(threading.Thread):
    """the one thread that searches: concurrent jobs with the same (query file, threshold, post-filter) become one
    fused pm_search over their batches"""

    def __init__(self, pm, coalesce_s=0.0):
        super().__init__(daemon=True)
        self.pm, self.coalesce_s = pm, coalesce_s
        self.jobs = queue.Queue()
        self.searches = self.fused_jobs = self.max_fused = 0
        self.qcache = OrderedDict()            # fasta digest -> Queries (a few recent query sets stay parsed)

    def submit(self, ix, fasta, threshold, nb, fkey=None):
        """fasta: the query bytes, or a callable that returns them (read only when the set is not parsed yet);
        fkey identifies them: (path, mtime, size) of the file a local client named, else their SHA-1 (computed by
        the handler thread, not by the dispatcher)"""
        if fkey is None:
            fkey = hashlib.sha1(fasta).digest()
        job = {"ix": ix, "fasta": fasta, "fkey": fkey, "threshold": float(threshold), "nb": nb, "done": threading.Event()}
        self.jobs.put(job)
        job["done"].wait()
        if "error" in job:
            raise job["error"]
        return job["queries"], job["hits"], job["slot"]

    def _queries(self, fasta, fkey, term_size):
        key = (fkey, term_size)
        q = self.qcache.get(key)
        if q is None:
            q = self.pm.Queries(fasta() if callable(fasta) else fasta, term_size=term_size)
            self.qcache[key] = q
            while len(self.qcache) > 4:
                self.qcache.popitem(last=False)      # freed when the last job that holds it lets go (refcount)
        else:
            self.qcache.move_to_end(key)
        return q

    def run(self):
        while True:
            first = self.jobs.get()
            if first is None:
                return
            batch = [first]
            deadline = time.perf_counter() + self.coalesce_s
            while True:
                try:
                    wait = max(0.0, deadline - time.perf_counter())
                    j = self.jobs.get(timeout=wait) if wait > 0 else self.jobs.get_nowait()
                except queue.Empty:
                    break
                if j is None:
                    self.jobs.put(None)
                    break
                batch.append(j)
            groups = OrderedDict()
            for j in batch:
                try:
                    k = (j["fkey"], j["threshold"], j["nb"], j["ix"].info.term_size)
                except Exception as e:                 # this thread must outlive any single bad job
                    j["error"] = e
                    j["done"].set()
                    continue
                groups.setdefault(k, []).append(j)
            for (_, thr, nb, k), jobs in groups.items():
                try:
                    q = self._queries(jobs[0]["fasta"], jobs[0]["fkey"], k)
                    # the same batch asked for twice in one group is searched once
                    uniq, slot_of = [], {}
                    for j in jobs:
                        h = id(j["ix"])
                        if h not in slot_of:
                            slot_of[h] = len(uniq)
                            uniq.append(j["ix"])
                    res = self.pm.search(uniq, q, thr, nb_best_hits=0 if nb is None else max(int(nb), 0))
                    per_slot = []
                    for k in range(len(uniq)):             # every batch's records on their own: a handler formats only its own
                        with res.slot_hits(k) as sl:
                            per_slot.append(sl.hits.copy())
                    res.free()
                    self.searches += 1
                    self.fused_jobs += len(jobs)
                    self.max_fused = max(self.max_fused, len(uniq))
                    for j in jobs:
                        k = slot_of[id(j["ix"])]
                        j["queries"], j["hits"], j["slot"] = q, per_slot[k], k
                except Exception as e:
                    for j in jobs:
                        j["error"] = e
                for j in jobs:
                    j["done"].set()


def _recv_line(conn):
    buf = bytearray()
    while not buf.endswith(b"\n"):
        chunk = conn.recv(1)
        if not chunk:
            break
        buf += chunk
    return bytes(buf)


def _recv_exact(conn, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = conn.recv(min(1 << 20, n - len(buf)))
        if not chunk:
            raise ConnectionError("client closed the connection early")
        buf += chunk
    return bytes(buf)


def serve(sock_path, device=0, max_gb=0.0, coalesce_ms=0.0, preload=()):
    from . import _lib as pm
    pm.init(device)
    free = pm.device_info()["hbm_free"]
    cache = IndexCache(pm, max_gb * 1e9 if max_gb > 0 else 0.85 * free)
    disp = Dispatcher(pm, coalesce_ms * 1e-3)
    disp.start()
    if os.path.exists(sock_path):
        os.unlink(sock_path)
    srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    srv.bind(sock_path)
    srv.listen(512)
    print(f"[phylign_amd.server] {pm.device_info()['name']} listening on {sock_path}", file=sys.stderr, flush=True)
    stop = threading.Event()
    preloading = {"left": 0, "errors": []}
    pre_mu = threading.Lock()

    def preload_paths(paths):
        def one(p):
            try:
                _, key, _ = cache.acquire(p)
                cache.release(key)
            except Exception as e:
                with pre_mu:
                    preloading["errors"].append(f"{p}: {e}")
            finally:
                with pre_mu:
                    preloading["left"] -= 1
        with pre_mu:
            preloading["left"] += len(paths)
        workers = [threading.Thread(target=lambda chunk=paths[i::4]: [one(p) for p in chunk], daemon=True) for i in range(4)]
        for w in workers:
            w.start()

    def handle(conn):
        with conn:
            try:
                req = json.loads(_recv_line(conn) or b"{}")
                op = req.get("op")
                if op == "query":
                    t0 = time.time()
                    if "fasta_len" in req:
                        fasta, fkey = _recv_exact(conn, int(req["fasta_len"])), None
                    else:
                        # a local client names the file: 305 per-batch jobs of one query set cost one read and one parse,
                        # not 305 transfers of (for a million reads) 160 MB each
                        fp = req["fasta_path"]
                        fst = os.stat(fp)
                        fkey = (os.path.realpath(fp), fst.st_mtime_ns, fst.st_size)
                        fasta = lambda fp=fp: open(fp, "rb").read()
                    ix, key, cached = cache.acquire(req["index"], int(req.get("index_size", 0)))
                    try:
                        t1 = time.time()
                        nb = req.get("nb_best_hits")
                        q, hits, slot = disp.submit(ix, fasta, float(req.get("threshold", 0.8)), nb, fkey)
                        t2 = time.time()
                        text = pm.format_hits(ix, q, hits, slot=slot, nb_best_hits=-1 if nb is None else max(int(nb), 0))
                    finally:
                        cache.release(key)
                    head = {"ok": True, "len": len(text), "cached": cached, "load_s": round(t1 - t0, 4),
                            "search_s": round(t2 - t1, 4), "query_s": round(time.time() - t1, 4)}
                    conn.sendall(json.dumps(head).encode() + b"\n" + text)
                elif op == "preload":
                    paths = list(req.get("indexes", []))
                    preload_paths(paths)
                    if req.get("wait"):
                        while preloading["left"] > 0:
                            time.sleep(0.02)
                    conn.sendall(json.dumps({"ok": not preloading["errors"], "queued": len(paths), "errors": preloading["errors"][-5:]}).encode() + b"\n")
                elif op == "stats":
                    with cache.mu:
                        resident = sum(1 for e in cache.items.values() if e["ready"].is_set())
                        loading = len(cache.items) - resident
                    conn.sendall(json.dumps({"ok": True, "resident": resident, "loading": loading, "resident_bytes": cache.used(),
                                             "budget_bytes": cache.max_bytes, "loads": cache.loads, "hits": cache.hits,
                                             "waited_for_a_load": cache.waits, "searches": disp.searches, "jobs": disp.fused_jobs,
                                             "max_batches_in_one_search": disp.max_fused,
                                             "preload_left": preloading["left"]}).encode() + b"\n")
                elif op == "drop":
                    cache.drop(req.get("index"))
                    conn.sendall(b'{"ok": true}\n')
                elif op == "shutdown":
                    conn.sendall(b'{"ok": true}\n')
                    stop.set()
                    try:                                          # wake the accept loop
                        socket.socket(socket.AF_UNIX, socket.SOCK_STREAM).connect(sock_path)
                    except OSError:
                        pass
                else:
                    conn.sendall(json.dumps({"ok": False, "error": f"unknown op {op!r}"}).encode() + b"\n")
            except Exception as e:                       # the server survives a bad request
                try:
                    conn.sendall(json.dumps({"ok": False, "error": f"{type(e).__name__}: {e}"}).encode() + b"\n")
                except OSError:
                    pass

    if preload:
        preload_paths(list(preload))
    while not stop.is_set():
        conn, _ = srv.accept()
        if stop.is_set():
            conn.close()
            break
        threading.Thread(target=handle, args=(conn,), daemon=True).start()
    disp.jobs.put(None)
    disp.join(timeout=30)
    disp.qcache.clear()
    cache.drop()
    srv.close()
    os.unlink(sock_path)


def request(sock_path, req, payload=b""):
    """client side: returns (header dict, body bytes)"""
    c = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    c.connect(sock_path)
    with c:
        c.sendall(json.dumps(req).encode() + b"\n" + payload)
        head = json.loads(_recv_line(c) or b'{"ok": false, "error": "no answer"}')
        body = _recv_exact(c, int(head["len"])) if head.get("ok") and "len" in head else b""
    return head, body


def main(argv=None):
    ap = argparse.ArgumentParser(description="resident-index server for the MI355X matching stage")
    ap.add_argument("--socket", required=True)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--max-gb", type=float, default=0.0, help="HBM budget for resident indexes (0 = 85%% of free)")
    ap.add_argument("--coalesce-ms", type=float, default=2.0,
                    help="how long the dispatcher waits for more jobs of the same query file before it searches")
    ap.add_argument("--preload", default=None, help="file with batch names (e.g. data/batches_full.txt) to load in the background")
    ap.add_argument("--cobs-dir", default=None, help="directory of <batch>.cobs_classic[.xz] for --preload")
    a = ap.parse_args(argv)
    pre = []
    if a.preload:
        for b in (x.strip() for x in open(a.preload)):
            if b:
                plain = os.path.join(a.cobs_dir or ".", f"{b}.cobs_classic")
                pre.append(plain if os.path.exists(plain) else plain + ".xz")
    serve(a.socket, a.device, a.max_gb, a.coalesce_ms, pre)


if __name__ == "__main__":
    main()

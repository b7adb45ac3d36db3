"""Persistent index residency: a per-GPU server process that
keeps decoded signature matrices in HBM across query sets, so a 03_match job
costs a socket round trip plus the scan instead of xz decoding + PCIe upload +
process start per batch (what index_load_mode / keep_cobs_indexes approximate on
disk in the reference: config.yaml:91-104, :134; Snakefile:163-175).

    python -m phylign_amd.server --socket /tmp/phylign_match.sock [--device 0] [--max-gb 250]

Protocol (unix stream socket, one request per connection): a JSON line, then
for op=query the server answers with a JSON line {"ok":..., "len": N, ...}
followed by N bytes of cobs/post-filtered text.  Client: cobs_query.py --server.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time
from collections import OrderedDict


class IndexCache:
    """LRU of resident indexes keyed by (realpath, mtime_ns, size)"""

    def __init__(self, pm, max_bytes):
        self.pm, self.max_bytes = pm, max_bytes
        self.items = OrderedDict()          # key -> (Index, device_bytes)
        self.loads = self.hits = 0

    def used(self):
        return sum(v[1] for v in self.items.values())

    def get(self, path, size_hint=0):
        st = os.stat(path)
        key = (os.path.realpath(path), st.st_mtime_ns, st.st_size)
        if key in self.items:
            self.items.move_to_end(key)
            self.hits += 1
            return self.items[key][0], True
        need = (size_hint or st.st_size * 8) * 1.1
        while self.items and self.used() + need > self.max_bytes:
            _, (old, _) = self.items.popitem(last=False)
            old.free()
        if path.endswith(".xz"):
            p = subprocess.Popen(["xzcat", "--no-sparse", "--ignore-check", path], stdout=subprocess.PIPE)
            try:
                ix = self.pm.Index.load_fd(p.stdout.fileno(), size_hint=size_hint)
            finally:
                p.stdout.close()
                rc = p.wait()
            if rc != 0:
                ix.free()
                raise RuntimeError(f"xzcat failed on {path}")
        else:
            ix = self.pm.Index.load_file(path, size_hint=size_hint)
        self.items[key] = (ix, ix.info.device_bytes)
        self.loads += 1
        return ix, False

    def drop(self, path=None):
        for key in list(self.items):
            if path is None or key[0] == os.path.realpath(path):
                self.items.pop(key)[0].free()


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


def serve(sock_path, device=0, max_gb=0.0, ready_fd=None):
    from . import _lib as pm
    pm.init(device)
    free = pm.device_info()["hbm_free"]
    cache = IndexCache(pm, max_gb * 1e9 if max_gb > 0 else 0.85 * free)
    if os.path.exists(sock_path):
        os.unlink(sock_path)
    srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    srv.bind(sock_path)
    srv.listen(64)
    print(f"[phylign_amd.server] {pm.device_info()['name']} listening on {sock_path}", file=sys.stderr, flush=True)
    running = True
    while running:
        conn, _ = srv.accept()
        with conn:
            try:
                req = json.loads(_recv_line(conn) or b"{}")
                op = req.get("op")
                if op == "query":
                    t0 = time.time()
                    fasta = _recv_exact(conn, int(req["fasta_len"])) if "fasta_len" in req else open(req["fasta_path"], "rb").read()
                    ix, cached = cache.get(req["index"], int(req.get("index_size", 0)))
                    t1 = time.time()
                    nb = req.get("nb_best_hits")
                    text = pm.query_text(ix, fasta, float(req.get("threshold", 0.8)), -1 if nb is None else max(int(nb), 0))
                    head = {"ok": True, "len": len(text), "cached": cached, "load_s": round(t1 - t0, 4),
                            "query_s": round(time.time() - t1, 4)}
                    conn.sendall(json.dumps(head).encode() + b"\n" + text)
                elif op == "stats":
                    conn.sendall(json.dumps({"ok": True, "resident": len(cache.items), "resident_bytes": cache.used(),
                                             "loads": cache.loads, "hits": cache.hits}).encode() + b"\n")
                elif op == "drop":
                    cache.drop(req.get("index"))
                    conn.sendall(b'{"ok": true}\n')
                elif op == "shutdown":
                    conn.sendall(b'{"ok": true}\n')
                    running = False
                else:
                    conn.sendall(json.dumps({"ok": False, "error": f"unknown op {op!r}"}).encode() + b"\n")
            except Exception as e:                       # the server survives a bad request
                try:
                    conn.sendall(json.dumps({"ok": False, "error": f"{type(e).__name__}: {e}"}).encode() + b"\n")
                except OSError:
                    pass
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
    a = ap.parse_args(argv)
    serve(a.socket, a.device, a.max_gb)


if __name__ == "__main__":
    main()

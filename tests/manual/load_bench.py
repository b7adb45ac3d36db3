#!/usr/bin/env python3
"""Index residency path timing: header parse + pinned double
buffering + H2D + re-stride kernel, from a regular file and from a pipe, for a
synthetic classic index written with the oracle's writer.  Run on the GPU box."""
import os
import subprocess
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import oracle as O          # lives under tests/: only tests may use the checker to write a test file
from phylign_amd import _lib as pm

pm.init(0)
n_docs, S = 4000, int(float(sys.argv[1]) * 1e9 / 500) if len(sys.argv) > 1 else 4_000_000
names = [f"{i:05x}_SAM{i:07d}" for i in range(n_docs)]
rng = np.random.default_rng(1)
t = time.time()
idx = O.make_index(31, 1, S, 1, names)
h = O.header_parse(idx)
idx[h.data_off:] = rng.integers(0, 256, size=idx.size - h.data_off, dtype=np.uint8)
path = "/tmp/load_bench.cobs_classic"
idx.tofile(path)
print(f"wrote {idx.size / 1e9:.2f} GB in {time.time() - t:.1f}s")
for label, opener in (("file", None), ("pipe(cat)", "cat")):
    for rep in range(2):
        t = time.time()
        if opener:
            p = subprocess.Popen([opener, path], stdout=subprocess.PIPE)
            ix = pm.Index.load_fd(p.stdout.fileno(), size_hint=idx.size)
            p.stdout.close(); p.wait()
        else:
            ix = pm.Index.load_file(path)
        dt = time.time() - t
        ok = np.array_equal(ix.read_row(S - 1), np.asarray(idx[h.data_off + (S - 1) * 500: h.data_off + S * 500]))
        print(f"{label:10s} rep {rep}: {dt:.2f}s  {idx.size / dt / 1e9:.2f} GB/s  last row ok={ok}")
        ix.free()
os.unlink(path)

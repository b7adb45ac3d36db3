#!/usr/bin/env python3
"""End-to-end 03_match -> 04_filter on ONE GPU at the real shapes of data/batches_small.txt
(195 / 176 / 664 documents, 1.84 GB of signatures) with N synthetic 150-bp reads: writes
the three batches as real .cobs_classic files (synthetic signatures + planted reads), runs
phylign_amd.match_stage as a subprocess and prints its per-batch timing, then checks a
sample of the output against the oracle.  Run on the GPU box: python tests/manual/e2e_demo.py [N]"""
import gzip
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from oracle import oracle as O                      # lives under tests/: only tests may use the checker
from phylign_amd import workload as W, postprocess as P

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
tmp = tempfile.mkdtemp(prefix="pm_e2e_")
cobs = os.path.join(tmp, "cobs")
os.makedirs(cobs)
shapes = W.select("small")
fasta, seqs = W.make_queries(N, 150, seed=31)
open(os.path.join(tmp, "Q.fa"), "wb").write(fasta)
rng = np.random.default_rng(3)
t0 = time.time()
index_of = {}
with open(os.path.join(tmp, "sizes.txt"), "w") as sz, open(os.path.join(tmp, "batches.txt"), "w") as bl:
    for pos, s in enumerate(shapes):
        m = O.synth_fill(661, s.batch_id, s.signature_size, s.n_docs, os.cpu_count() or 1)
        for q in range(pos, N, max(3, N // 3000)):          # ~1000 planted reads per batch
            hs = O.create_hashes(seqs[q].tobytes(), 31, 1, 1)
            rows = (hs % np.uint64(s.signature_size)).astype(np.int64)
            for d, frac in zip(rng.choice(s.n_docs, size=6, replace=False), (1.0, 0.9, 0.9, 0.8, 0.7, 0.6)):
                r = rows[: int(np.ceil(frac * len(rows)))]
                m[r, d >> 3] |= np.uint8(1 << (d & 7))
        names = [f"{rng.integers(0, 16**5):05x}_SAM{pos}N{d:06d}" for d in range(s.n_docs)]
        idx = O.make_index(31, 1, s.signature_size, 1, names, m)
        path = os.path.join(cobs, f"{s.batch}.cobs_classic")
        idx.tofile(path)
        sz.write(f"cobs/{s.batch}.cobs_classic.xz  {idx.size}  1610678320\n")
        bl.write(s.batch + "\n")
        index_of[s.batch] = idx if pos == 0 else None       # keep one for the spot check
print(f"wrote {sum(s.index_bytes for s in shapes) / 1e9:.2f} GB of indexes in {time.time() - t0:.1f}s -> {tmp}")
t0 = time.time()
r = subprocess.run([sys.executable, "-m", "phylign_amd.match_stage", "--batches", os.path.join(tmp, "batches.txt"),
                    "--cobs-dir", cobs, "--sizes", os.path.join(tmp, "sizes.txt"), "--queries", os.path.join(tmp, "Q.fa"),
                    "--out-dir", os.path.join(tmp, "03_match"), "--filter-out", os.path.join(tmp, "04_filter", "Q.fa")],
                   capture_output=True, env=dict(os.environ, PYTHONPATH=ROOT))
wall = time.time() - t0
assert r.returncode == 0, r.stderr.decode()[-2000:]
log = json.loads(r.stderr.decode().strip().split("\n")[-1])
print(f"match_stage: {N} reads x {len(shapes)} batches, process wall {wall:.2f}s (stage wall {log['stage_wall_s']}s, "
      f"match only {log['match_only_s']}s, e2e {log['e2e_s']}s)")
for row in log["per_group"]:
    print("  ", row)
b0 = shapes[0].batch
nq = 300
sub = "".join(f">q{i:07d}\n{seqs[i].tobytes().decode()}\n" for i in range(nq)).encode()
exp = P.filter_text(O.query_file(index_of[b0], sub, 0.7).decode(), 100)
got = gzip.open(os.path.join(tmp, "03_match", f"{b0}____Q.gz"), "rt").read()
assert got.startswith(exp), "03_match file differs from the oracle on the first 300 reads"
out = open(os.path.join(tmp, "04_filter", "Q.fa")).read()
print(f"spot check vs oracle ok; 03_match sizes: {[os.path.getsize(os.path.join(tmp, '03_match', f)) for f in sorted(os.listdir(os.path.join(tmp, '03_match')))]}, "
      f"04_filter {len(out)} bytes, {out.count('>')} records")
subprocess.run(["rm", "-rf", tmp])

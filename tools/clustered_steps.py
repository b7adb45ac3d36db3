#!/usr/bin/env python3
"""A few pipelined steps of the clustered workload (for timeline profiling)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phylign_amd import _lib as pm, workload as W
pm.init(0)
shapes = W.scale_shapes(W.select("config3"), int(os.environ.get("DIV", "1")))
fasta, _ = W.make_queries(100000, 150, seed=31)
q = pm.Queries(fasta)
ixs = []
for pos, s in enumerate(shapes):
    ix = pm.Index.synth(s.batch_id, s.n_docs, s.signature_size)
    ix.plant_cluster(q, pos, len(shapes))
    ixs.append(ix)
pm.set_option("threshold_bound", int(os.environ.get("BOUND", "0")))
prev = None
t0 = time.perf_counter()
marks = []
for i in range(int(os.environ.get("STEPS", "6"))):
    cur = pm.search_async(ixs, q, 0.7, nb_best_hits=100)
    if prev is not None:
        a = time.perf_counter(); prev.wait(); b = time.perf_counter(); h = prev.hits(copy=False); c = time.perf_counter()
        marks.append((b - a, c - b, len(h))); prev.free()
    prev = cur
prev.wait(); prev.hits(copy=False); prev.free()
print("total", time.perf_counter() - t0, [(round(x * 1e3, 2), round(y * 1e3, 2), n) for x, y, n in marks])

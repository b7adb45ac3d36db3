#!/usr/bin/env python3
"""400 pipelined searches (two in flight, alternating cut / no cut, long hit lists): HBM and host RSS
must not grow once the buffer pools are warm.  GPU box: python3 tools/leak_check.py"""
import os, sys
sys.path.insert(0, os.getcwd())
from phylign_amd import _lib as pm, workload as W
pm.init(0)
shapes = W.scale_shapes(W.select("config3"), 50)
fasta, _ = W.make_queries(20000, 150, seed=31)
q = pm.Queries(fasta)
ixs = []
for pos, s in enumerate(shapes):
    ix = pm.Index.synth(s.batch_id, s.n_docs, s.signature_size)
    ix.plant_cluster(q, pos, len(shapes))
    ixs.append(ix)
def run(n):
    prev = None
    for i in range(n):
        cur = pm.search_async(ixs, q, 0.7, nb_best_hits=100 if i % 2 else 0)
        if prev is not None:
            prev.hits(copy=False); prev.free()
        prev = cur
    prev.hits(); prev.free()
run(20)
import resource
f0 = pm.device_info()["hbm_free"]; r0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
run(400)
f1 = pm.device_info()["hbm_free"]; r1 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
print("hbm_free delta MB", (f0 - f1) / 1e6, "maxrss delta MB", (r1 - r0) / 1e3)

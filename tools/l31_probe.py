#!/usr/bin/env python3
"""BASELINE configs[1] literally: one batch (bacillus_anthracis__01 shape), 10 k synthetic 31-mer queries
(one k-mer each: hit <=> bit set), threshold 0.7 -- latency of one query set, next to the 150-bp form."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from phylign_amd import _lib as pm  # noqa: E402
from phylign_amd import workload as W  # noqa: E402

pm.init(0)
s = W.select("config2")[0]
ix = pm.Index.synth(s.batch_id, s.n_docs, s.signature_size)
print("qlen  nb_best  wall_ms  kernels_ms  hash_ms  scan_ms  records")
for L in (31, 150):
    fasta, _ = W.make_queries(10000, L, seed=32)
    q = pm.Queries(fasta)
    for nb in (0, 100):
        for i in range(4):
            t = time.perf_counter()
            r = pm.search([ix], q, 0.7, nb_best_hits=nb)
            h = r.hits(copy=False)
            dt = time.perf_counter() - t
            st = r.stats
            if i:
                print(f"{L:4d} {nb:8d} {dt * 1e3:8.3f} {st.ms_total:11.3f} {st.ms_hash:8.3f} {st.ms_scan:8.3f} {len(h):8d}")
            r.free()

#!/usr/bin/env python3
"""Scan time of every config-3 batch on its own (fetch-all and threshold-bound), to check the
static batch -> GPU cost model (workload.scan_cost).  Prints one TSV row per batch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phylign_amd import _lib as pm, workload as W
pm.init(0)
shapes = W.select("config3")
fasta, _ = W.make_queries(100000, 150, seed=31)
q = pm.Queries(fasta)
print("batch\tn_docs\trow_bytes\tstride\tS\tGB\tlines\tms_fetch_all\tms_bound\tns_per_line_lookup")
for s in shapes:
    ix = pm.Index.synth(s.batch_id, s.n_docs, s.signature_size)
    info = ix.info
    out = []
    for bound in (0, 1):
        pm.set_option("threshold_bound", bound)
        best = 1e9
        for _ in range(4):
            r = pm.search([ix], q, 0.7, nb_best_hits=100)
            best = min(best, r.stats.ms_scan)
            r.free()
        out.append(best)
    lines = (s.row_bytes + 127) // 128
    print(f"{s.batch}\t{s.n_docs}\t{s.row_bytes}\t{info.stride}\t{s.signature_size}\t{info.device_bytes / 1e9:.2f}\t{lines}\t"
          f"{out[0]:.3f}\t{out[1]:.3f}\t{out[0] * 1e6 / (12e6 * lines):.3f}")
    ix.free()

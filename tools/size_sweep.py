#!/usr/bin/env python3
"""k_scan on one 4000-document batch of growing signature_size (2 MB ... 8 GB matrix), fetch-all:
where the per-line rate stops depending on the matrix size (L2 -> Infinity Cache -> HBM)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phylign_amd import _lib as pm, workload as W
pm.init(0)
pm.set_option("threshold_bound", 0)
fasta, _ = W.make_queries(100000, 150, seed=31)
q = pm.Queries(fasta)
print("matrix_MB\trows\tms\tTB/s_algorithmic\tGlines/s")
for S in (4096, 32768, 131072, 524288, 2097152, 16777216):
    ix = pm.Index.synth(1, 4000, S)
    best = 1e9
    for _ in range(4):
        r = pm.search([ix], q, 0.7, nb_best_hits=100); best = min(best, r.stats.ms_scan); r.free()
    print(f"{ix.info.device_bytes / 1e6:.0f}\t{S}\t{best:.3f}\t{12e6 * 500 / best / 1e9:.2f}\t{12e6 * 4 / best / 1e6:.1f}")
    ix.free()

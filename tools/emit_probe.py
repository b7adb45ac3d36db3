#!/usr/bin/env python3
"""04_filter emit of a million reads as 1 / 3 / 6 merges (pieces): time of pm.emit_merges_to per layout (host only)."""
import os, sys, time, tempfile
sys.path.insert(0, os.getcwd())
from phylign_amd import _lib as pm, match_stage as MS, workload as W
fasta, _ = W.make_queries(1_000_000, 150, seed=5)
d = tempfile.mkdtemp()
for mb in (0, 48, 24):
    pieces = MS.split_prepared_fasta(fasta, 0, mb << 20)
    qs = [pm.Queries(p) for p in pieces]
    t = time.perf_counter(); ms = [pm.Merge(q, 100) for q in qs]; tc = time.perf_counter() - t
    ts = []
    for rep in range(4):
        t = time.perf_counter(); n = pm.emit_merges_to(ms, os.path.join(d, "x.fa")); ts.append(time.perf_counter() - t)
    print(f"piece_mb {mb}: {len(ms)} merges, create {tc*1e3:.0f} ms, emit {[round(x*1e3) for x in ts]} ms, {n/1e6:.0f} MB", flush=True)
    for m in ms: m.free()
    for q in qs: q.free()

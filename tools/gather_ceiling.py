#!/usr/bin/env python3
"""What the MI355X memory system delivers for k_scan's access pattern: a pure
random-row gather (no counting) over synthetic indexes of several row widths.
Prints algorithmic GB/s (rows x row_bytes / time) and 128-B lines per second."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from phylign_amd import _lib as pm  # noqa: E402

pm.init(0)
print(f"{'docs':>6} {'row_B':>6} {'stride':>6} {'rows':>10} {'matrix_GB':>9} {'lookups':>10} {'ms':>8} {'alg_GB/s':>9} {'Glines/s':>9}")
for n_docs, S in ((4000, 12_800_000), (4000, 1_600_000), (2300, 12_000_000), (1000, 12_000_000), (664, 16_500_000),
                  (400, 12_000_000), (200, 12_000_000), (100, 4_000_000)):
    ix = pm.Index.synth(1, n_docs, S, seed=661)
    info = ix.info
    groups, per = 100_000, 120          # same shape as 100k queries x 120 k-mers
    best = None
    for _ in range(3):
        ms, nb = ix.probe_gather(groups, per)
        best = ms if best is None else min(best, ms)
    lines = groups * per * ((info.stride + 127) // 128)
    print(f"{n_docs:6d} {info.row_bytes:6d} {info.stride:6d} {S:10d} {info.device_bytes / 1e9:9.2f} {groups * per:10d} "
          f"{best:8.3f} {nb / best / 1e6:9.1f} {lines / best / 1e6:9.2f}")
    ix.free()

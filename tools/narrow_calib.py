#!/usr/bin/env python3
"""Narrow-row gather calibration: the pure random-row gather of k_scan's access
pattern (k_probe_gather) on synthetic matrices of every narrow lane-group class (1/2/4/8 lanes per
row = 16/32/64/128-byte strides) plus one wide class, with a KNOWN lookup count, (a) timed per
cache-policy flavour of the load (plain / nt / sc1 / sc0 sc1 / sc0 sc1 nt / sc0), (b) under
`rocprofv3 --pmc` (tools/run_narrow_pmc.sh) so that the read-request SIZE split (TCC_EA0_RDREQ_32B /
_64B / _128B) of this pattern is measured instead of assumed.

    python3 tools/narrow_calib.py            # timing table, all flavours
    python3 tools/narrow_calib.py --pmc      # one launch per (width, flavour 0 and 1): for the counter passes
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from phylign_amd import _lib as pm  # noqa: E402

FLAVOURS = {0: "plain", 1: "nt", 2: "sc1", 3: "sc0 sc1", 4: "sc0 sc1 nt", 5: "sc0"}
SHAPES = ((100, 4_000_000), (100, 60_000_000), (200, 30_000_000), (400, 24_000_000), (664, 16_500_000), (1000, 24_000_000), (4000, 12_800_000))
GROUPS, PER = 100_000, 120           # 12 M lookups, the shape of 100 k queries x 120 k-mers


def main():
    pmc = "--pmc" in sys.argv
    pm.init(0)
    print(f"# lookups per launch: {GROUPS * PER}")
    print(f"{'docs':>6} {'row_B':>6} {'stride':>6} {'lanes':>5} {'rows':>10} {'matrix_GB':>9} {'flavour':>11} {'ms':>8} {'Glookups/s':>10} {'alg_GB/s':>9}")
    for n_docs, S in SHAPES:
        ix = pm.Index.synth(1, n_docs, S, seed=661)
        info = ix.info
        for fl, name in FLAVOURS.items():
            if pmc and fl not in (0, 1, 3):
                continue
            os.environ["PM_PROBE_FLAVOR"] = str(fl)
            best = None
            for _ in range(1 if pmc else 4):
                ms, nb = ix.probe_gather(GROUPS, PER)
                best = ms if best is None else min(best, ms)
            lanes = min(64, max(1, (min(info.stride, 1024) + 15) // 16))
            print(f"{n_docs:6d} {info.row_bytes:6d} {info.stride:6d} {lanes:5d} {S:10d} {info.device_bytes / 1e9:9.2f} {name:>11} "
                  f"{best:8.3f} {GROUPS * PER / best / 1e6:10.2f} {nb / best / 1e6:9.1f}", flush=True)
        ix.free()
    os.environ["PM_PROBE_FLAVOR"] = "0"


if __name__ == "__main__":
    main()

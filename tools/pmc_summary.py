#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 --pmc counters from *_counter_collection.csv files.
Usage: tools/pmc_summary.py dir [dir ...]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(dirs):
    for d in dirs:
        for f in sorted(glob.glob(os.path.join(d, "*_counter_collection.csv"))):
            agg = defaultdict(lambda: [0, 0.0])
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    k = (row["Kernel_Name"], row["Counter_Name"])
                    agg[k][0] += 1
                    agg[k][1] += float(row["Counter_Value"])
            print(f"# {f}")
            print(f"{'dispatches':>10} {'mean_per_dispatch':>20}  counter  kernel")
            for (k, c), (n, s) in sorted(agg.items()):
                print(f"{n:10d} {s / n:20.1f}  {c}  {k[:110]}")


if __name__ == "__main__":
    main(sys.argv[1:])

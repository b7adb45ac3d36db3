#!/usr/bin/env python3
"""Merged GPU timeline (kernels + memory copies) from a rocprofv3 run with
`--kernel-trace --memory-copy-trace --output-format csv`: start/end relative to the
first event, for the last `--last-ms` milliseconds.  Usage:
    tools/timeline.py DIR [--last-ms 200] [--min-us 50]"""
import argparse
import csv
import glob
import os


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--last-ms", type=float, default=200.0)
    ap.add_argument("--min-us", type=float, default=50.0)
    a = ap.parse_args()
    ev = []
    for f in glob.glob(os.path.join(a.dir, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K q%s %s" % (r.get("Queue_Id", "?"), r["Kernel_Name"][:60])))
    for f in glob.glob(os.path.join(a.dir, "**", "*memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C %s %s B" % (r.get("Direction", "?"), r.get("Bytes", r.get("Size", "?")))))
    if not ev:
        print("no events under", a.dir)
        return
    ev.sort()
    t_end = max(e[1] for e in ev)
    t0 = t_end - int(a.last_ms * 1e6)
    for s, e, name in ev:
        if s < t0 or (e - s) < a.min_us * 1e3:
            continue
        print(f"{(s - t0) / 1e6:10.3f} -> {(e - t0) / 1e6:10.3f} ms  ({(e - s) / 1e6:8.3f} ms)  {name}")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (kernel trace, optional PMC) as
text: per kernel name -> calls, total/avg/min/max duration, share; and per
counter -> mean value per dispatch for each kernel.  Usage:
    tools/rocpd_summary.py results.db [> profiles/xyz.txt]"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = db.execute(f"select {name_col}, start, end from kernels").fetchall()
    agg = {}
    for n, s, e in rows:
        a = agg.setdefault(n, [])
        a.append(e - s)
    tot = sum(sum(v) for v in agg.values()) or 1
    print(f"# {path}: {len(rows)} kernel dispatches")
    print(f"{'calls':>7} {'total_ms':>10} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'pct':>6}  name")
    for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print(f"{len(v):7d} {sum(v) / 1e6:10.3f} {sum(v) / len(v) / 1e3:10.2f} {min(v) / 1e3:10.2f} {max(v) / 1e3:10.2f} "
              f"{100 * sum(v) / tot:6.2f}  {n}")
    try:
        pc = db.execute("select * from counters_collection limit 1")
        ccols = [d[0] for d in pc.description]
    except sqlite3.Error:
        ccols = []
    if ccols:
        kn = "kernel_name" if "kernel_name" in ccols else "name"
        cn = "counter_name" if "counter_name" in ccols else "name"
        cv = "value" if "value" in ccols else "counter_value"
        try:
            q = db.execute(f"select {kn}, {cn}, count(*), avg({cv}), sum({cv}) from counters_collection group by {kn}, {cn}").fetchall()
        except sqlite3.Error as e:
            q = []
            print("# counters_collection not readable:", e, ccols)
        if q:
            print("\n# counters (per kernel, mean per dispatch)")
            print(f"{'dispatches':>10} {'mean':>18} {'sum':>20}  counter  kernel")
            for k, c, n, a, s in sorted(q, key=lambda r: (r[0], r[1])):
                print(f"{n:10d} {a:18.1f} {s:20.1f}  {c}  {k}")


if __name__ == "__main__":
    main(sys.argv[1])

#!/usr/bin/env python3
"""prints the figures of a bench record that the round's notes quote: python tools/bench_summary.py <bench.json>
<bench.json> is either the side file of a run (bench.py --legs-out: the whole record) or the stdout line, whose
`legs_file` names the side file (looked up as written, then by its base name next to the line's file)."""
import json
import os
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
if d.get("legs_file") and "scan_launches" not in d:
    for cand in (d["legs_file"], os.path.join(os.path.dirname(os.path.abspath(sys.argv[1])), os.path.basename(d["legs_file"]))):
        if os.path.exists(cand):
            d = json.load(open(cand))
            break
print("value %.4g k-mers/s  %.3f ms/step  roofline frac %.3f (%.2f ms)  n_gpus %d" % (
    d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["n_gpus"]))
if d.get("threshold_bound"):
    print("bound  %.4g k-mers/s  %.3f ms/step" % (d["threshold_bound"]["value"], d["threshold_bound"]["ms_per_step"]))
if d.get("unique_rows"):
    u = d["unique_rows"]
    print("unique_rows %d  x row_bytes %.2f GB  (%.3f of algorithmic %.2f GB; %.3f of resident rows)  pass %.2fs" % (
        u["unique_rows"], u["unique_rows_x_row_bytes"] / 1e9, u["ratio_to_algorithmic"], u["algorithmic_bytes_per_step"] / 1e9,
        u["fraction_of_resident_rows_touched"], u["pass_s"]))
for rep in ("x1", "x8"):
    g = (d.get("argannot") or {}).get(rep)
    if not g:
        continue
    for m in ("fetch_all_rows", "threshold_bound"):
        s = g[m]
        print("argannot %s %s: %.4g k-mers/s %.3f ms/step frac(dominant) %.3f" % (rep, m, s["value"], s["ms_per_step"], s["roofline"]["frac"]),
              {k: (round(v["avg_ms"], 3), round(v["algorithmic_GBps"])) for k, v in s.get("scan_launches", {}).items()})
if d.get("cpu_baseline"):
    c = d["cpu_baseline"]
    print("cpu %.4g k-mers/s on %d threads (%s)  gpu/cpu %.0f" % (c["value"], c["cores"], c["partition"], d["gpu_over_cpu"]))
if (d.get("argannot") or {}).get("error"):
    print("argannot error:", d["argannot"]["error"])

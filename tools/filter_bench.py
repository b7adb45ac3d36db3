#!/usr/bin/env python3
"""04_filter timing on synthetic 03_match files: the drop-in scripts/filter_queries.py with its native reader
(pm_merge_add_text), with its Python reader (--python), and -- when a checkout of the reference is given -- the
reference's own scripts/filter_queries.py (through the `xopen` shim of tools/gen_golden_filter.py).  Host code only.

    python3 tools/filter_bench.py [--queries 200000] [--files 20] [--reference /root/reference]"""
import argparse
import gzip
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--queries", type=int, default=200000)
    ap.add_argument("--files", type=int, default=20)
    ap.add_argument("--hit-fraction", type=float, default=0.05, help="queries with matches per file")
    ap.add_argument("--reference", default=None)
    a = ap.parse_args()
    rng = np.random.default_rng(3)
    tmp = tempfile.mkdtemp(prefix="pm_filter_")
    seqs = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(a.queries, 150))]
    with open(os.path.join(tmp, "Q.fa"), "wb") as f:
        for i in range(a.queries):
            f.write(b">q%07d\n" % i + seqs[i].tobytes() + b"\n")
    files, total = [], 0
    for b in range(a.files):
        hit = rng.random(a.queries) < a.hit_fraction
        n = rng.integers(1, 101, size=a.queries)
        lines = []
        for i in range(a.queries):
            if hit[i]:
                sc = np.sort(rng.integers(84, 121, size=n[i]))[::-1]
                lines.append(f"*q{i:07d}\t{n[i]}\n" + "".join(f"_SAM{b:02d}D{rng.integers(0, 4000):04d}\t{s}\n" for s in sc))
            else:
                lines.append(f"*q{i:07d}\t0\n")
        text = "".join(lines).encode()
        total += len(text)
        p = os.path.join(tmp, f"genus_species__{b:02d}____Q.gz")
        with gzip.open(p, "wb", compresslevel=1) as f:
            f.write(text)
        files.append(p)
    print(f"{a.files} match files, {a.queries} queries each, {total / 1e6:.0f} MB of text")
    env = dict(os.environ, PYTHONPATH=ROOT)
    outs = {}

    def run(label, cmd, env_):
        t = time.time()
        r = subprocess.run(cmd, capture_output=True, env=env_)
        dt = time.time() - t
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        outs[label] = r.stdout
        print(f"{label:34s} {dt:8.2f} s   {len(r.stdout) / 1e6:.1f} MB of FASTA")
    drop = os.path.join(ROOT, "scripts", "filter_queries.py")
    run("drop-in, native reader", [sys.executable, drop, "-n", "100", "-q", os.path.join(tmp, "Q.fa")] + files, env)
    run("drop-in, Python reader (--python)", [sys.executable, drop, "--python", "-n", "100", "-q", os.path.join(tmp, "Q.fa")] + files, env)
    if a.reference:
        shim = os.path.join(tmp, "shim")
        os.makedirs(shim)
        with open(os.path.join(shim, "xopen.py"), "w") as f:
            f.write("import gzip\ndef xopen(fn, mode='r'):\n    return gzip.open(fn, mode + 't') if str(fn).endswith('.gz') else open(fn, mode)\n")
        run("reference scripts/filter_queries.py", [sys.executable, os.path.join(a.reference, "scripts", "filter_queries.py"), "-n", "100",
                                                  "-q", os.path.join(tmp, "Q.fa")] + files, dict(os.environ, PYTHONPATH=shim))
    ref = outs.get("reference scripts/filter_queries.py", outs["drop-in, Python reader (--python)"])
    print("outputs identical:", all(v == ref for v in outs.values()))
    subprocess.run(["rm", "-rf", tmp])


if __name__ == "__main__":
    main()

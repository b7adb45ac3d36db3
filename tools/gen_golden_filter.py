#!/usr/bin/env python3
"""Golden I/O for the 04_filter consumer, captured by running the reference's
scripts/filter_queries.py (build container only) with a local `xopen` shim
(xopen is not installed here; the shim maps it to gzip.open/open in text mode).
Writes tests/golden/filter/{queries.fa, <batch>____q.gz, expected.n<N>.fa}."""
import gzip, os, random, subprocess, sys, tempfile
ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
script = os.path.join(ref, "scripts", "filter_queries.py")
outdir = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "filter")
os.makedirs(outdir, exist_ok=True)
random.seed(11)
queries = [(f"r{i}", "".join(random.choice("ACGT") for _ in range(60))) for i in range(9)]
with open(os.path.join(outdir, "queries.fa"), "w") as f:
    for n, s in queries:
        f.write(f">{n} descr\n{s}\n")
batches = ["aaa_bbb__01", "ccc_ddd__01", "ccc_ddd__02"]
files = []
for b in batches:
    fn = os.path.join(outdir, f"{b}____q.gz")
    files.append(fn)
    with gzip.open(fn, "wt") as f:
        order = list(queries); random.shuffle(order)
        for n, _ in order:
            k = random.choice([0, 0, 1, 3, 7])
            sc = sorted((random.choice([30, 28, 25, 22]) for _ in range(k)), reverse=True)
            f.write(f"*{n} descr\t{k}\n")
            for j, s in enumerate(sc):
                f.write(f"_SAM{b[:1].upper()}{random.randint(0, 999):03d}{j}\t{s}\n")
shim = tempfile.mkdtemp()
with open(os.path.join(shim, "xopen.py"), "w") as f:
    f.write("import gzip\n"
            "def xopen(fn, mode='r'):\n"
            "    return gzip.open(fn, mode + 't') if str(fn).endswith('.gz') else open(fn, mode)\n")
env = dict(os.environ, PYTHONPATH=shim)
for n in (1, 2, 5, 100):
    r = subprocess.run([sys.executable, script, "-n", str(n), "-q", os.path.join(outdir, "queries.fa")] + files,
                       capture_output=True, check=True, env=env, cwd=outdir)
    with open(os.path.join(outdir, f"expected.n{n}.fa"), "wb") as f:
        f.write(r.stdout)
print("filter goldens written")

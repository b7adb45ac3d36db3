#!/usr/bin/env python3
"""Host-side rows a8 / f1: native text formatting + post-filter (pm_format_hits) and native
04_filter merge (pm_merge_*) against the Python line loops that mirror the reference's
scripts/postprocess_cobs.py and scripts/filter_queries.py, on the same synthetic 03_match
content (Q queries x B batches, H hits per (query, batch)).  With a checkout of the reference as second
argument (build container only) the reference's own scripts/postprocess_cobs.py is timed on the same text.

    python3 tools/postfilter_bench.py [Q] [/root/reference]"""
import gzip
import io
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from phylign_amd import _lib as pm, filter_queries as F, postprocess as P  # noqa: E402

Q, B, H, D = int(sys.argv[1]) if len(sys.argv) > 1 else 20000, 8, 30, 2000
rng = np.random.default_rng(5)
fasta = "".join(f">q{i}\n{'ACGT' * 10}\n" for i in range(Q)).encode()
q = pm.Queries(fasta, term_size=31)
names = [[f"{rng.integers(0, 16**5):05x}_S{b}x{d}" for d in range(D)] for b in range(B)]
idx = [pm.Index.from_names(n) for n in names]
hits = []
for b in range(B):
    rec = np.zeros(Q * H, dtype=pm.HIT_DTYPE)
    rec["query"] = np.repeat(np.arange(Q, dtype=np.uint32), H)
    rec["doc"] = (rng.random((Q, H)).argsort(axis=1) + rng.integers(0, D - H, size=(Q, 1))).reshape(-1)
    rec["score"] = rng.integers(84, 121, size=Q * H)
    rec["slot"] = b
    hits.append(rec)

t = time.time(); plain = [pm.format_hits(idx[b], q, hits[b], slot=b) for b in range(B)]; t_plain = time.time() - t
t = time.time(); fused = [pm.format_hits(idx[b], q, hits[b], slot=b, nb_best_hits=10) for b in range(B)]; t_fused = time.time() - t
t = time.time(); py = [P.filter_text(x.decode(), 10) for x in plain]; t_py = time.time() - t
assert all(a.decode() == b for a, b in zip(fused, py))
print(f"{Q} queries x {B} batches x {H} hits: cobs text {sum(map(len, plain)) / 1e6:.1f} MB")
print(f"  native format (plain)            {t_plain:7.2f} s")
print(f"  native format + post-filter n=10 {t_fused:7.2f} s")
print(f"  Python post-filter line loop     {t_py:7.2f} s   (+ the text has to exist first)")
if len(sys.argv) > 2:
    import subprocess
    script = os.path.join(sys.argv[2], "scripts", "postprocess_cobs.py")
    t = time.time()
    ref = [subprocess.run([sys.executable, script, "-n", "10"], input=x, capture_output=True, check=True).stdout for x in plain]
    t_ref = time.time() - t
    assert all(a == b for a, b in zip(fused, ref)), "the reference's post-filter prints something else"
    print(f"  reference postprocess_cobs.py    {t_ref:7.2f} s   (same bytes as the fused native form)")

d = tempfile.mkdtemp()
files = []
for b in range(B):
    fn = os.path.join(d, f"batch{b:02d}__01____q.gz")
    with gzip.open(fn, "wb", compresslevel=1) as g:
        g.write(fused[b])
    files.append(fn)
qf = os.path.join(d, "q.fa")
open(qf, "wb").write(fasta)
t = time.time()
m = pm.Merge(q, 10)
for b in range(B):
    m.add(f"batch{b:02d}__01", idx[b], hits[b], slot=b, nb_best_hits=10)
native = m.emit()
t_nat = time.time() - t
t = time.time(); out = io.StringIO(); F.filter_files(qf, files, 10, out); t_pyf = time.time() - t
assert native.decode() == out.getvalue()
print(f"  native 04_filter merge           {t_nat:7.2f} s")
print(f"  Python 04_filter (reads the gz)  {t_pyf:7.2f} s")

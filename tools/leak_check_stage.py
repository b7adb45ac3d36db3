#!/usr/bin/env python3
"""The stage run 30 times over a resident shard (fused groups, native file writer, merge, FASTA emit) plus 60 loads of a
320 MB index file through the parallel reader, with long-query searches that split across workgroups in between: HBM
and host RSS must stay flat once the pools (hit buffers, pinned results, loader staging, split slabs) are warm.
GPU box: python3 tools/leak_check_stage.py"""
import os
import resource
import shutil
import sys
import tempfile

sys.path.insert(0, os.getcwd())
import numpy as np  # noqa: E402
from phylign_amd import _lib as pm, match_stage as MS, workload as W  # noqa: E402

pm.init(0)
shapes = W.scale_shapes(W.select("full"), 40)[:24]
fasta, _ = W.make_queries(50000, 150, seed=31)
q = pm.Queries(fasta)
ixs = {}
for pos, s in enumerate(shapes):
    ix = pm.Index.synth(s.batch_id, s.n_docs, s.signature_size)
    ix.plant_cluster(q, pos, len(shapes))
    ixs[s.batch] = ix
names = sorted(ixs)
src = MS.ResidentSource(ixs)
tmp = tempfile.mkdtemp(prefix="pm_leak_")
lq_fasta, _ = W.make_queries(4, 400_030, seed=5)
lq = pm.Queries(lq_fasta)
# a 320 MB index file for the parallel reader
big = pm.Index.synth(77, 2371, 1_080_000)
rows = big.read_rows(0, 1_080_000)
hdr = pm.Index.synth(77, 2371, 1_080_000, header_only=True)
path = os.path.join(tmp, "big.cobs_classic")
with open(path, "wb") as f:
    f.write(b"COBS:CLASSIC_INDEX" + np.array([1, 31], dtype="<u4").tobytes() + b"\x01" + np.array([2371], dtype="<u4").tobytes()
            + np.array([1_080_000, 1], dtype="<u8").tobytes() + "".join(hdr.doc_name(d) + "\n" for d in range(2371)).encode() + b"CLASSIC_INDEX")
    f.write(rows.tobytes())
big.free()
del rows


def cycle(n):
    for i in range(n):
        rep, merge = MS.run_stage(pm, names, list(range(len(names))), src, q, "Q", os.path.join(tmp, "03_match"), 0.7, 100, want_merge=True)
        merge.emit_to(os.path.join(tmp, "Q.fa"))
        merge.free()
        for _ in range(2):
            ix = pm.Index.load_file(path)
            ix.free()
        r = pm.search(list(ixs.values())[:6], lq, 0.7, nb_best_hits=100)
        r.hits(copy=False)
        r.free()


def rss_mb():
    return int(open("/proc/self/statm").read().split()[1]) * os.sysconf("SC_PAGE_SIZE") / 1e6


cycle(6)
f0, r0 = pm.device_info()["hbm_free"], rss_mb()
for rnd in range(3):
    cycle(20)
    print(f"after {20 * (rnd + 1)} more cycles: hbm_free delta MB {(f0 - pm.device_info()['hbm_free']) / 1e6:.1f}, "
          f"rss delta MB {rss_mb() - r0:.1f} (rss {rss_mb():.0f} MB, maxrss {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e3:.0f} MB)", flush=True)
shutil.rmtree(tmp)

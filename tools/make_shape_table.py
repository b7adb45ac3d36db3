#!/usr/bin/env python3
"""Derive the 661k shape table (one line per phylogenetic batch) from the
reference's data files.  Run in the build container only (needs /root/reference).

Inputs  (reference data, read-only):
  data/661k_batches.txt.xz             batch <TAB> acc,acc,...     (305 lines)
  data/decompressed_indexes_sizes.txt  cobs/<batch>.cobs_classic.xz <bytes> <xz RAM>
  data/batches_full.txt, data/batches_small.txt
Output: phylign_amd/data/shapes_661k.tsv
  batch  n_docs  index_bytes  row_bytes  signature_size  in_small  in_config3
signature_size = floor(index_bytes / ceil(n_docs/8))  (SURVEY.md section 8d).
"""
import lzma, os, sys
ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(__file__), "..", "phylign_amd", "data", "shapes_661k.tsv")
docs = {}
with lzma.open(os.path.join(ref, "data/661k_batches.txt.xz"), "rt") as f:
    for line in f:
        b, accs = line.rstrip("\n").split("\t")
        docs[b] = len(accs.split(","))
sizes = {}
with open(os.path.join(ref, "data/decompressed_indexes_sizes.txt")) as f:
    for line in f:
        p, sz, _ = line.split()
        sizes[os.path.basename(p).replace(".cobs_classic.xz", "")] = int(sz)
full = sorted(x.strip() for x in open(os.path.join(ref, "data/batches_full.txt")) if x.strip())
small = [x.strip() for x in open(os.path.join(ref, "data/batches_small.txt")) if x.strip()]
cfg3 = set(full[::5]) | set(small)
with open(out, "w") as o:
    o.write("#batch\tn_docs\tindex_bytes\trow_bytes\tsignature_size\tin_small\tin_config3\n")
    for b in full:
        d = docs[b]; rb = (d + 7) // 8; s = sizes[b] // rb
        o.write(f"{b}\t{d}\t{sizes[b]}\t{rb}\t{s}\t{int(b in small)}\t{int(b in cfg3)}\n")
print("batches", len(full), "config3", len(cfg3))

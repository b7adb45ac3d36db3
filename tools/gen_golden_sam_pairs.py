#!/usr/bin/env python3
"""The reference's only golden that passes through COBS (data/reads_1___reads_2___reads_3___reads_4.sam_summary.xz, checked
by `make test`, Makefile:40-55: batches_small.txt, nb_best_hits=1) reduced to what it says about the MATCH stage: the
(batch, read, sample accession) triples that minimap2 aligned.  minimap2 only ever sees the references the 04_filter FASTA
names for a read (scripts/batch_align.py), so every triple must be a candidate of that read after 03_match + 04_filter --
a necessary-subset check of this build against the real cobs 0.2.1, usable wherever the three real indexes of
batches_small.txt are at hand (tests/test_real_indexes.py).

    python3 tools/gen_golden_sam_pairs.py /root/reference > tests/golden/sam_summary_pairs.tsv      (build container only)
"""
import lzma
import re
import sys

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
path = f"{ref}/data/reads_1___reads_2___reads_3___reads_4.sam_summary.xz"
triples = set()
batch = None
with lzma.open(path, "rt") as f:
    for line in f:
        m = re.match(r"==> intermediate/05_map/(.+?)____.*\.sam\.gz <==", line)
        if m:
            batch = m.group(1)
            continue
        parts = line.rstrip("\n").split("\t")
        if len(parts) >= 3 and batch and parts[2] != "*":
            triples.add((batch, parts[0], parts[2].split(".")[0]))
print("# batch\tread\tsample accession -- from the reference's data/reads_1___reads_2___reads_3___reads_4.sam_summary.xz (tools/gen_golden_sam_pairs.py)")
for t in sorted(triples):
    print("\t".join(t))

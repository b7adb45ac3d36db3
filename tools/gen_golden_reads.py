#!/usr/bin/env python3
"""The reference's bundled query reads (data/reads_1.fastq, reads_2.fq, reads_3.fasta,
reads_4.fa: 40 reads, names 1A..4J) as the merged, preprocessed FASTA the 03_match
stage receives (intermediate/01_queries_merged/reads_1___reads_2___reads_3___reads_4.fa,
Snakefile:336-352) -> tests/golden/reads/reads_1___reads_2___reads_3___reads_4.fa.
A data fixture (inputs only); run in the build container."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from phylign_amd.fix_query import fix_files  # noqa: E402
ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
out = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "reads")
os.makedirs(out, exist_ok=True)
files = [os.path.join(ref, "data", f) for f in ("reads_1.fastq", "reads_2.fq", "reads_3.fasta", "reads_4.fa")]
with open(os.path.join(out, "reads_1___reads_2___reads_3___reads_4.fa"), "wb") as f:
    fix_files(files, f)
print(open(os.path.join(out, "reads_1___reads_2___reads_3___reads_4.fa")).read().count(">"), "reads")
# the four input files themselves (FASTQ and FASTA, as `make test` feeds them to rule fix_query): data fixtures
import shutil
os.makedirs(os.path.join(out, "raw"), exist_ok=True)
for f in files:
    shutil.copyfile(f, os.path.join(out, "raw", os.path.basename(f)))

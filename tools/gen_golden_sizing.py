#!/usr/bin/env python3
"""Known-answer vectors for the 03_match sizing helpers (Snakefile:41-121): the
helper block of the reference Snakefile is exec'd here (build container only)
with stub `config` / `workflow` objects and the results are written to
tests/golden/sizing.tsv:  batch streaming cobs_threads cores max_ram_gb ->
size_bytes xz_ram_mb ram_mb threads."""
import os, re, sys, types
ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
src = open(os.path.join(ref, "Snakefile")).read().split("\n")
block = "\n".join(src[40:121])  # lines 41..121: get_index_metadata .. get_number_of_COBS_threads
out = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "sizing.tsv")
sizes = os.path.join(ref, "data", "decompressed_indexes_sizes.txt")
W = types.SimpleNamespace
rows = []
batches = ["chlamydia_pecorum__01", "actinobacillus_pleuropneumoniae__01", "bacillus_anthracis__01",
           "pseudomonas_aeruginosa__01", "acinetobacter_baumannii__01", "escherichia_coli__05"]
for b in batches:
    for streaming in (True, False):
        for ct in ("auto", "auto(4)", "3"):
            for cores in (1, 8, 32):
                for ram in (12, 64):
                    g = {"re": re, "config": {"max_ram_gb": ram}, "workflow": W(cores=cores)}
                    exec(block, g)
                    wc, inp = W(batch=b), W(decompressed_indexes_sizes=sizes)
                    rows.append((b, int(streaming), ct, cores, ram,
                                 g["get_uncompressed_batch_size"](wc, inp),
                                 g["get_xz_decompress_RAM_in_MB"](wc, inp),
                                 g["get_uncompressed_batch_size_in_MB"](wc, inp, False, streaming),
                                 g["get_number_of_COBS_threads"](wc, inp, ct, streaming)))
with open(out, "w") as f:
    f.write("#batch\tstreaming\tcobs_threads\tcores\tmax_ram_gb\tsize_bytes\txz_ram_mb\tram_mb\tthreads\n")
    for r in rows:
        f.write("\t".join(map(str, r)) + "\n")
print(len(rows), "sizing vectors")

#!/usr/bin/env python3
"""How well does the threshold bound prune on genome-like data?  Builds a few species batches
from sequences (tests/helpers.build_species_batch: correlated strains, COBS sizing), queries
them with reads from every species, and prints per (reads of species x batch) the fraction of
row bytes the scan really gathers (pm_set_option count_fetched), next to the i.i.d. Bernoulli(1/4)
figure of the bench.  Run on the GPU box: python tests/manual/realistic_batches.py"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O                      # lives under tests/: only tests may use the checker
from helpers import build_species_batch, sample_reads, rand_seq
from phylign_amd import _lib as pm

pm.init(0)
rng = np.random.default_rng(11)
shapes = [(1000, 60000), (400, 80000), (2500, 40000)]
batches = [build_species_batch(O, rng, n, g) for n, g in shapes]
ixs = [pm.Index.load_mem(b[0]) for b in batches]
for b, ix in zip(batches, ixs):
    rows = ix.read_rows(0, ix.info.signature_size)
    dens = np.unpackbits(rows, axis=1, bitorder="little")[:, :ix.info.n_docs].mean()
    print(f"# batch: {ix.info.n_docs} strains, S={ix.info.signature_size}, bit density {dens:.3f}")
print("reads_of\tbatch\tn_docs\tfraction_of_row_bytes_gathered\thits")
pm.set_option("count_fetched", 1)
sets = [("species%d" % b, [s for _, s in sample_reads(rng, batches[b][1], 2000)]) for b in range(len(batches))]
sets.append(("random", [rand_seq(rng, 150) for _ in range(2000)]))
for name, reads in sets:
    q = pm.Queries("".join(f">r{i}\n{s}\n" for i, s in enumerate(reads)).encode())
    for b, ix in enumerate(ixs):
        res = pm.search([ix], q, 0.7, nb_best_hits=100)
        st = res.stats
        print(f"{name}\t{b}\t{ix.info.n_docs}\t{st.fetched_bytes / st.algorithmic_bytes:.3f}\t{st.n_hits}")
        res.free()

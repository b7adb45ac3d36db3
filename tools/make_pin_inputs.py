#!/usr/bin/env python3
"""Inputs of tools/pin_against_cobs.sh (seeded, numpy only): a genome-like species batch as one FASTA per strain --
what `cobs classic-construct` takes -- and query files whose scores straddle every rule that nothing in the reference
pins: reads of 151 bp (121 k-mers: ceil(0.7 x 121) = 85 but floor = 84), 150 bp and 31-40 bp at error rates
0-12 %, documents that are identical over the read (equal scores -> tie order), a read shorter than k and one with an N."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from helpers import codes_to_seq, mutate  # noqa: E402  (pure numpy helpers)


def main(out):
    rng = np.random.default_rng(20211)
    gdir = os.path.join(out, "genomes")
    os.makedirs(gdir, exist_ok=True)
    for f in os.listdir(gdir):
        os.unlink(os.path.join(gdir, f))
    anc = rng.integers(0, 4, size=20000, dtype=np.uint8)
    strains = [anc]
    while len(strains) < 25:                                   # a random tree: close relatives share most k-mers
        strains.append(mutate(rng, strains[int(rng.integers(0, len(strains)))], float(rng.uniform(0.0005, 0.01))))
    strains = strains[1:]
    strains[5] = strains[4].copy()                             # two identical documents: equal scores on every read
    for d, s in enumerate(strains):
        name = f"{int(rng.integers(0, 16 ** 5)):05x}_SAMPIN{d:04d}"       # "<rnd>_<accession>" like the 661k names
        with open(os.path.join(gdir, name + ".fa"), "w") as f:
            f.write(f">{name} strain {d}\n")
            seq = codes_to_seq(s).decode()
            for i in range(0, len(seq), 80):
                f.write(seq[i:i + 80] + "\n")
    reads = []
    for i in range(400):
        L = (151, 150, 151, int(rng.integers(31, 41)))[i % 4]
        s = strains[int(rng.integers(0, len(strains)))]
        p = int(rng.integers(0, len(s) - L))
        err = (0.0, 0.005, 0.01, 0.02, 0.03, 0.05, 0.08, 0.12)[(i // 4) % 8]
        reads.append((f"pin{i:04d}" + (" some comment" if i % 5 == 0 else ""), codes_to_seq(mutate(rng, s[p:p + L], err)).decode()))
    with open(os.path.join(out, "queries.fa"), "w") as f:
        for h, s in reads:
            f.write(f">{h}\n{s}\n")
    with open(os.path.join(out, "queries_few.fa"), "w") as f:  # threshold 0: every document of every query is listed
        for h, s in reads[:6]:
            f.write(f">{h}\n{s}\n")
    with open(os.path.join(out, "edge_short.fa"), "w") as f:
        f.write(">ok1\n" + reads[0][1] + "\n>tooshort\nACGTACGTACGTACGTACGTACGTACG\n>ok2\n" + reads[1][1] + "\n")
    with open(os.path.join(out, "edge_with_n.fa"), "w") as f:
        s = reads[2][1]
        f.write(">ok1\n" + reads[0][1] + "\n>has_n\n" + s[:60] + "N" + s[61:] + "\n>ok2\n" + reads[1][1] + "\n")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "tests/golden/cobs")

#!/usr/bin/env python3
"""Drop-in for the reference's scripts/filter_queries.py (`-n int -q query.fa match files...` -> FASTA on stdout;
rule translate_matches, Snakefile:490-520)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from phylign_amd.filter_queries import main  # noqa: E402

if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""XXH64 known-answer vectors from python-xxhash (libxxhash) -> tests/golden/xxh64_kat.tsv.
Columns: hex(input) seed digest.  600 random 31-mers x seeds 0..3 plus every length 0..70."""
import os, random, xxhash
random.seed(31)
out = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "xxh64_kat.tsv")
with open(out, "w") as f:
    f.write(f"# python-xxhash {xxhash.VERSION} / libxxhash {xxhash.XXHASH_VERSION}\n")
    for _ in range(600):
        s = "".join(random.choice("ACGT") for _ in range(31)).encode()
        for seed in range(4):
            f.write(f"{s.hex()}\t{seed}\t{xxhash.xxh64_intdigest(s, seed=seed)}\n")
    for L in range(0, 71):
        s = bytes(random.getrandbits(8) for _ in range(L))
        for seed in (0, 1, 7, 2**64 - 1):
            f.write(f"{s.hex()}\t{seed}\t{xxhash.xxh64_intdigest(s, seed=seed)}\n")

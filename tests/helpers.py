"""Shared builders for parity tests: small random classic indexes with planted
hits and query FASTA files, all seeded."""
import numpy as np


def rand_seq(rng, n):
    return "".join("ACGT"[i] for i in rng.integers(0, 4, size=n))


def doc_names(rng, n, with_underscore=True):
    if with_underscore:
        return [f"{rng.integers(0, 16**5):05x}_SAM{d:06d}" for d in range(n)]
    return [f"doc{d}" for d in range(n)]


def build_case(O, rng, n_docs, sig_size, queries, k=31, canon=1, num_hashes=1,
               density=0.25, plant=None):
    """queries: list of (header, seq).  plant: list of (query idx, doc, fraction)
    -> sets the bits of the first ceil(fraction*terms) terms of that query for doc.
    Returns (index_bytes, fasta_bytes, matrix)."""
    rb = (n_docs + 7) // 8
    bits = rng.random((sig_size, rb * 8)) < density
    bits[:, n_docs:] = False
    matrix = np.packbits(bits, axis=1, bitorder="little")
    for qi, doc, frac in (plant or []):
        seq = queries[qi][1].encode()
        hs = O.create_hashes(seq, k, canon, num_hashes).reshape(-1, num_hashes)
        nt = hs.shape[0]
        m = int(np.ceil(frac * nt))
        for t in range(m):
            for j in range(num_hashes):
                r = int(hs[t, j]) % sig_size
                matrix[r, doc >> 3] |= np.uint8(1 << (doc & 7))
    names = doc_names(rng, n_docs)
    index = O.make_index(k, canon, sig_size, num_hashes, names, matrix)
    fasta = "".join(f">{h}\n{s}\n" for h, s in queries).encode()
    return index, fasta, matrix

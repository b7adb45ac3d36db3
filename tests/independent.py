"""A SECOND, independent restatement of `cobs query` over a classic index that shares nothing with oracle/: pure Python,
the third-party python-xxhash for XXH64, its own header writer, canonicalisation, scorer and text -- written from
SURVEY.md appendix A alone (tests/test_gpu_independent.py holds the HIP path against it, tests/test_golden_cpu.py the
C oracle).  Test infrastructure, like oracle/."""
import functools
import math
import struct

import numpy as np
import xxhash

COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}


def canonical(kmer):
    rc = "".join(COMP[c] for c in reversed(kmer))
    return min(kmer, rc)                                     # ASCII order: A < C < G < T


def rows_of(seq, k, num_hashes, S, canon=True):
    """per k-mer the `num_hashes` row numbers: XXH64(31 ASCII bytes, seed j) mod signature_size"""
    out = []
    for i in range(len(seq) - k + 1):
        km = canonical(seq[i:i + k]) if canon else seq[i:i + k]
        out.append([xxhash.xxh64(km.encode(), seed=j).intdigest() % S for j in range(num_hashes)])
    return out


def classic_index(names, genomes, k, num_hashes, S, canon=True):
    """.cobs_classic bytes (SURVEY.md A.1 with upstream's field order: term_size, canonicalize, n_docs, signature_size,
    num_hashes) and the bit matrix [S][ceil(D/8)]"""
    D = len(names)
    rb = (D + 7) // 8
    m = np.zeros((S, rb), dtype=np.uint8)
    for d, g in enumerate(genomes):
        for rr in rows_of(g, k, num_hashes, S, canon):
            for r in rr:
                m[r, d >> 3] |= 1 << (d & 7)
    head = b"COBS:" + b"CLASSIC_INDEX" + struct.pack("<IIBIQQ", 1, k, 1 if canon else 0, D, S, num_hashes)
    head += b"".join(n.encode() + b"\n" for n in names) + b"CLASSIC_INDEX"
    return head + m.tobytes(), m


def query_text(records, names, m, k, num_hashes, S, threshold, canon=True):
    """the text `cobs query -t threshold -f file` prints"""
    out = []
    D = len(names)
    for header, seq in records:
        score = [0] * D
        kmers = rows_of(seq, k, num_hashes, S, canon)
        for rr in kmers:
            acc = None
            for r in rr:
                bits = np.unpackbits(m[r], bitorder="little")[:D]
                acc = bits if acc is None else (acc & bits)
            for d in np.nonzero(acc)[0]:
                score[int(d)] += 1
        need = math.ceil(threshold * len(kmers))
        hits = sorted((d for d in range(D) if score[d] >= need and (need > 0 or True)), key=lambda d: (-score[d], d))
        out.append(f"*{header}\t{len(hits)}\n" + "".join(f"{names[d]}\t{score[d]}\n" for d in hits))
    return "".join(out).encode()


@functools.lru_cache(maxsize=None)
def make_case(seed, D, glen, k, num_hashes, canon):
    rng = np.random.default_rng(seed)
    anc = rng.integers(0, 4, size=glen)
    genomes = []
    for _ in range(D):
        g = anc.copy()
        mut = rng.random(glen) < rng.uniform(0.0, 0.08)
        g[mut] = rng.integers(0, 4, size=int(mut.sum()))
        genomes.append("".join("ACGT"[c] for c in g))
    names = [f"{int(rng.integers(0, 16 ** 5)):05x}_SAMPLE{d:04d}" for d in range(D)]
    S = int(math.ceil((glen - k + 1) / -math.log(1 - 0.3)))    # false-positive rate 0.3, like the 661k indexes
    records = []
    for i in range(30):
        d = int(rng.integers(0, D))
        L = int(rng.choice([k, k + 3, 100, 150, 151]))
        p = int(rng.integers(0, glen - L))
        s = list(genomes[d][p:p + L])
        for j in range(L):                                       # read errors: scores land on both sides of the threshold
            if rng.random() < (0.0, 0.01, 0.03, 0.06)[i % 4]:
                s[j] = "ACGT"[int(rng.integers(0, 4))]
        s = "".join(s)
        if i % 5 == 0:                                           # reads arrive in either orientation
            s = "".join(COMP[c] for c in reversed(s))
        records.append((f"read{i} strain={d}", s))
    records.append(("unrelated", "".join("ACGT"[c] for c in rng.integers(0, 4, size=150))))
    return names, genomes, S, records


CASES = [(1, 24, 3000, 31, 1, True), (2, 70, 1500, 31, 1, True), (3, 9, 2500, 31, 2, True), (4, 130, 900, 21, 1, True),
         (5, 40, 2000, 31, 1, False), (6, 1030, 400, 31, 1, True)]


@functools.lru_cache(maxsize=None)
def built_case(*case):
    """(index bytes, matrix, fasta bytes, names, S, records) of one of CASES"""
    seed, D, glen, k, num_hashes, canon = case
    names, genomes, S, records = make_case(*case)
    index, m = classic_index(names, genomes, k, num_hashes, S, canon)
    fasta = "".join(f">{h}\n{s}\n" for h, s in records).encode()
    return index, m, fasta, names, S, records


# ---- the compact index: sub-indexes of page_size x 8 documents, each with its own signature size
# and number of hash functions; one header, the closing magic ends on a page boundary, sub-matrices follow page-wide ----
def compact_index(names, genomes, k, page, params, canon=True):
    """.cobs_compact bytes and the sub-matrices; params = [(signature_size, num_hashes)] per sub-index"""
    D, per = len(names), page * 8
    assert (D + per - 1) // per == len(params)
    mats = []
    for p_, (S, nh) in enumerate(params):
        m = np.zeros((S, page), dtype=np.uint8)
        for d in range(p_ * per, min(D, (p_ + 1) * per)):
            dl = d - p_ * per
            for rr in rows_of(genomes[d], k, nh, S, canon):
                for r in rr:
                    m[r, dl >> 3] |= 1 << (dl & 7)
        mats.append(m)
    head = b"COBS:" + b"COMPACT_INDEX" + struct.pack("<IIBIIQ", 1, k, 1 if canon else 0, len(params), D, page)
    head += b"".join(struct.pack("<QQ", S, nh) for S, nh in params)
    head += b"".join(n.encode() + b"\n" for n in names)
    head += bytes((page - (len(head) + 13) % page) % page) + b"COMPACT_INDEX"
    assert len(head) % page == 0
    return head + b"".join(m.tobytes() for m in mats), mats


def query_text_compact(records, names, mats, k, page, params, threshold, canon=True):
    out = []
    D, per = len(names), page * 8
    for header, seq in records:
        score = [0] * D
        n_kmers = len(seq) - k + 1
        for p_, (S, nh) in enumerate(params):
            lo, hi = p_ * per, min(D, (p_ + 1) * per)
            for rr in rows_of(seq, k, nh, S, canon):
                acc = None
                for r in rr:
                    bits = np.unpackbits(mats[p_][r], bitorder="little")[:hi - lo]
                    acc = bits if acc is None else (acc & bits)
                for dl in np.nonzero(acc)[0]:
                    score[lo + int(dl)] += 1
        need = math.ceil(threshold * n_kmers)
        hits = sorted((d for d in range(D) if score[d] >= need), key=lambda d: (-score[d], d))
        out.append(f"*{header}\t{len(hits)}\n" + "".join(f"{names[d]}\t{score[d]}\n" for d in hits))
    return "".join(out).encode()


COMPACT_CASES = [(11, 8, 150, [(900, 1), (700, 2), (1100, 1)], 31), (12, 16, 300, [(500, 1), (800, 1), (300, 3)], 31),
                 (13, 32, 40, [(2000, 2)], 21)]


@functools.lru_cache(maxsize=None)
def built_compact_case(seed, page, D, params, k):
    """(index bytes, sub-matrices, fasta, names, records) of one of COMPACT_CASES (params as a tuple of pairs)"""
    names, genomes, _S, records = make_case(seed, D, 700, k, 1, True)
    index, mats = compact_index(names, genomes, k, page, list(params), True)
    fasta = "".join(f">{h}\n{s}\n" for h, s in records).encode()
    return index, mats, fasta, names, records

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


# ---------------------------------------------------------------- genome-like batches
def mutate(rng, seq_codes, rate):
    """substitutions at `rate` per base on a uint8 code array (0..3)"""
    out = seq_codes.copy()
    pos = np.flatnonzero(rng.random(out.size) < rate)
    out[pos] = (out[pos] + rng.integers(1, 4, size=pos.size)) % 4
    return out


def codes_to_seq(codes):
    return np.frombuffer(b"ACGT", dtype=np.uint8)[codes].tobytes()


def build_species_batch(O, rng, n_docs, genome_len, divergence=(0.0005, 0.01), k=31, fpr=0.3, size_spread=0.3):
    """A phylogenetic batch the way COBS builds one (classic index, one hash, canonical k-mers):
    an ancestral genome, `n_docs` strains derived from it through a random tree of substitutions
    (so documents are strongly correlated) and of varying length, signature_size = max k-mers /
    -ln(1 - fpr) (= 2.80 x for the 661k indexes' fpr 0.3).  Every canonical k-mer
    of a strain sets its bit.  Returns (index bytes, list of strain code arrays, names)."""
    anc = rng.integers(0, 4, size=genome_len, dtype=np.uint8)
    strains = [anc]
    while len(strains) < n_docs + 1:                         # random tree: each new strain mutates an earlier one
        parent = strains[int(rng.integers(0, len(strains)))]
        strains.append(mutate(rng, parent, float(rng.uniform(*divergence))))
    strains = strains[1:]
    strains = [s[: int(genome_len * (1.0 - size_spread * rng.random()))] for s in strains]
    max_kmers = max(len(s) for s in strains) - k + 1
    S = int(np.ceil(max_kmers / -np.log(1.0 - fpr)))
    rb = (n_docs + 7) // 8
    matrix = np.zeros((S, rb), dtype=np.uint8)
    for d, s in enumerate(strains):
        hs = O.create_hashes(codes_to_seq(s), k, 1, 1)
        rows = np.unique((hs % np.uint64(S)).astype(np.int64))
        matrix[rows, d >> 3] |= np.uint8(1 << (d & 7))
    names = doc_names(rng, n_docs)
    return O.make_index(k, 1, S, 1, names, matrix), strains, names


def sample_reads(rng, strains, n_reads, read_len=150, error=0.01):
    """reads from random strains with substitution errors: list of (strain index, sequence str)"""
    out = []
    for _ in range(n_reads):
        d = int(rng.integers(0, len(strains)))
        s = strains[d]
        p = int(rng.integers(0, len(s) - read_len))
        out.append((d, codes_to_seq(mutate(rng, s[p:p + read_len], error)).decode()))
    return out


# ---------------------------------------------------------------- bench.py's driver-facing line
BENCH_LINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                   "vs_baseline", "dtype", "data", "config", "roofline", "roofline_narrow", "cpu_baseline", "gpu_over_cpu",
                   "participants", "pm_kernels_blob", "legs_file")
BENCH_LEG_KEYS = ("threshold_bound", "unique_rows", "argannot", "clustered", "l31", "full_shard", "full_collection",
                  "scan_launches", "rank0_ms", "value_note")


def check_bench_line(text):
    """What the round-5 driver record lost (BENCH_r05.json "parsed": null after the line grew to 28.6 KB): the last
    stdout line of bench.py is ONE small strict-JSON object with the contract's keys and none of the auxiliary legs."""
    import json

    def no_constants(name):
        raise AssertionError(f"non-strict JSON constant {name} in the bench line")
    assert "\n" not in text and len(text.encode()) < 8192, len(text)
    line = json.loads(text, parse_constant=no_constants)
    assert isinstance(line, dict)
    missing = [k for k in BENCH_LINE_KEYS if k not in line]
    assert not missing, missing
    assert not [k for k in BENCH_LEG_KEYS if k in line]
    assert isinstance(line["config"], dict) and isinstance(line["config"].get("workload"), str) and "model" not in line["config"]
    assert line["unit"] == "k-mers/s" and line["higher_is_better"] is True and line["data"] == "synthetic"
    assert line["value"] > 0 and line["ms_per_step"] > 0
    rf = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    part = line["participants"]
    assert part["ranks"] == line["n_gpus"] == len(part["rank_ms_per_step"])
    return line


def bench_record(stdout_bytes, legs_path):
    """(line, whole record) of one bench.py run: the LAST stdout line, checked, and the --legs-out side file"""
    import json
    out = stdout_bytes.decode()
    assert out.endswith("\n") and out.count("\n{") + out.startswith("{") == 1, out[-400:]      # one JSON line, nothing behind it
    line = check_bench_line(out.rstrip("\n").splitlines()[-1])
    full = json.load(open(legs_path))
    import os
    assert os.path.realpath(line["legs_file"]) == os.path.realpath(str(legs_path))
    for k in ("value", "ms_per_step", "n_gpus", "steps", "warmup"):
        assert full[k] == line[k], k
    return line, full

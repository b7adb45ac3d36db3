"""The reference's own golden for the path that passes through COBS -- data/reads_1___reads_2___reads_3___reads_4.sam_summary.xz,
what `make test` compares (Makefile:40-55: batches_small.txt, nb_best_hits=1) -- as a NECESSARY-SUBSET check of the match
stage: minimap2 only aligns a read against the references its 04_filter record names, so every (batch, read, sample
accession) the golden summary holds (tests/golden/sam_summary_pairs.tsv, tools/gen_golden_sam_pairs.py) must be a
candidate of that read after 03_match + 04_filter.

The three real indexes (Zenodo, 1.8 GB decompressed) are not in the build container, the GPU box or the reference tree:
point PHYLIGN_REAL_COBS_DIR at a directory that holds <batch>.cobs_classic[.xz] for the batches of batches_small.txt and
the `real` tests run -- oracle (CPU: also the first real header the readers ever see) and HIP product (GPU).  Without it
they skip, and a stand-in collection built from the golden itself exercises the same checking code on the GPU."""
import collections
import gzip
import lzma
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
BATCHES = ["actinobacillus_pleuropneumoniae__01", "aeromonas_salmonicida__01", "bacillus_anthracis__01"]   # data/batches_small.txt
REAL = os.environ.get("PHYLIGN_REAL_COBS_DIR", "")
HAVE_REAL = bool(REAL) and all(os.path.exists(os.path.join(REAL, f"{b}.cobs_classic.xz")) or os.path.exists(os.path.join(REAL, f"{b}.cobs_classic"))
                               for b in BATCHES)
need_real = pytest.mark.skipif(not HAVE_REAL, reason="set PHYLIGN_REAL_COBS_DIR to a directory with the three indexes of batches_small.txt")
MERGED = "reads_1___reads_2___reads_3___reads_4"


def golden_triples():
    out = collections.defaultdict(set)                      # read -> {(batch, accession)}
    with open(os.path.join(GOLD, "sam_summary_pairs.tsv")) as f:
        for line in f:
            if line.startswith("#") or not line.strip():
                continue
            b, read, acc = line.rstrip("\n").split("\t")
            out[read].add((b, acc))
    return out


def candidates_of_fasta(text):
    """04_filter FASTA -> read -> set of reference names (scripts/filter_queries.py:152-157: `>name ref,ref,...`)"""
    out = {}
    for line in text.splitlines():
        if line.startswith(">"):
            name, _, com = line[1:].partition(" ")
            out[name] = set(x for x in com.split(",") if x)
    return out


def check_necessary_subset(cands):
    gold = golden_triples()
    assert len(gold) == 36 and sum(len(v) for v in gold.values()) == 5671
    missing = [(r, b, a) for r, pairs in gold.items() for b, a in pairs if a not in cands.get(r, ())]
    assert not missing, f"{len(missing)} aligned (read, reference) pairs of the reference's golden are not candidates here, e.g. {missing[:5]}"


def run_stage(cobs_dir, tmp_path, extra=()):
    (tmp_path / "batches_small.txt").write_text("\n".join(BATCHES) + "\n")
    r = subprocess.run([sys.executable, "-m", "phylign_amd.match_stage", "--batches", str(tmp_path / "batches_small.txt"),
                        "--cobs-dir", str(cobs_dir), "--input-dir", os.path.join(GOLD, "reads", "raw"), "--nb-best-hits", "1",
                        "--out-dir", str(tmp_path / "03_match"), "--filter-out", str(tmp_path / "04_filter" / f"{MERGED}.fa")] + list(extra),
                       capture_output=True, env=dict(os.environ, PYTHONPATH=ROOT))
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    return (tmp_path / "04_filter" / f"{MERGED}.fa").read_text()


def _index_bytes(d, batch):
    p = os.path.join(d, f"{batch}.cobs_classic")
    if os.path.exists(p):
        return open(p, "rb").read()
    return lzma.open(p + ".xz", "rb").read()


@need_real
def test_real_indexes_oracle_candidates_cover_the_reference_golden(oracle):
    """CPU: the oracle reads the real files (the header layout is pinned by that alone), scores the bundled reads, and the
    golden-pinned mirrors of postprocess_cobs.py -n 1 and filter_queries.py -n 1 turn the text into candidates"""
    import io
    from phylign_amd import filter_queries as F
    from phylign_amd import postprocess as P
    import tempfile
    merged = open(os.path.join(GOLD, "reads", f"{MERGED}.fa"), "rb").read()
    with tempfile.TemporaryDirectory() as d:
        files = []
        for b in BATCHES:
            index = _index_bytes(REAL, b)
            h = oracle.header_parse(index)
            assert h.term_size == 31 and h.num_hashes >= 1 and h.n_docs > 0
            text = P.filter_text(oracle.query_file(index, merged, 0.7).decode(), 1)
            fn = os.path.join(d, f"{b}____{MERGED}.gz")
            with gzip.open(fn, "wt") as g:
                g.write(text)
            files.append(fn)
        out = io.StringIO()
        F.filter_files(os.path.join(GOLD, "reads", f"{MERGED}.fa"), files, 1, out)
    check_necessary_subset(candidates_of_fasta(out.getvalue()))


@need_real
@pytest.mark.gpu
def test_real_indexes_product_candidates_cover_the_reference_golden(pm, oracle, tmp_path):
    """GPU: the product's stage on the real files; its 03_match text equals the oracle's on them, and the candidates
    cover the golden"""
    from phylign_amd import postprocess as P
    fa = run_stage(REAL, tmp_path)
    check_necessary_subset(candidates_of_fasta(fa))
    merged = open(os.path.join(GOLD, "reads", f"{MERGED}.fa"), "rb").read()
    for b in BATCHES:
        exp = P.filter_text(oracle.query_file(_index_bytes(REAL, b), merged, 0.7).decode(), 1)
        assert gzip.open(tmp_path / "03_match" / f"{b}____{MERGED}.gz", "rt").read() == exp, b


@pytest.mark.gpu
def test_stand_in_collection_walks_the_same_check(pm, oracle, tmp_path):
    """the checking code above on a stand-in for the three batches built FROM the golden (every aligned pair's document
    holds all k-mers of its read, plus decoys): the stage finds exactly the golden's candidates -- and dropping one
    planted pair makes the check fail.  Says nothing about cobs; it keeps the real-data test from rotting."""
    from helpers import doc_names
    rng = np.random.default_rng(5)
    gold = golden_triples()
    reads = {}
    lines = open(os.path.join(GOLD, "reads", f"{MERGED}.fa")).read().split("\n")
    for h, s in zip(lines[0::2], lines[1::2]):
        if h.startswith(">"):
            reads[h[1:].split(" ")[0]] = s
    cobs = tmp_path / "cobs"
    cobs.mkdir()
    dropped = None
    for variant in ("full", "one_missing"):
        for b in BATCHES:
            accs = sorted({a for pairs in gold.values() for bb, a in pairs if bb == b})
            names = [f"{int(rng.integers(0, 16 ** 5)):05x}_{a}" for a in accs] + doc_names(rng, 7)
            S = 40009
            m = np.zeros((S, (len(names) + 7) // 8), dtype=np.uint8)
            for read, pairs in gold.items():
                rows = (oracle.create_hashes(reads[read].encode(), 31, 1, 1) % np.uint64(S)).astype(np.int64)
                for bb, a in pairs:
                    if bb != b or (variant == "one_missing" and (read, bb, a) == dropped):
                        continue
                    d = accs.index(a)
                    m[rows, d >> 3] |= np.uint8(1 << (d & 7))
            index = oracle.make_index(31, 1, S, 1, names, m)
            (cobs / f"{b}.cobs_classic").write_bytes(bytes(index))
        out = tmp_path / variant
        out.mkdir()
        cands = candidates_of_fasta(run_stage(cobs, out))
        if variant == "full":
            check_necessary_subset(cands)
            read = sorted(gold)[3]
            bb, a = sorted(gold[read])[0]
            dropped = (read, bb, a)
        else:
            with pytest.raises(AssertionError, match="not candidates"):
                check_necessary_subset(cands)

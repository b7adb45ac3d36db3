"""GPU parity: the HIP path through the C ABI vs the CPU oracle, bit-exact.
Reference path: `cobs query` as called at scripts/run_cobs_streaming.sh:24-29."""
import os

import numpy as np
import pytest

from helpers import build_case, rand_seq

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("k,canon,nh", [(31, 1, 1), (31, 0, 1), (31, 1, 3), (15, 1, 2), (32, 1, 1),
                                         (33, 0, 2), (45, 1, 1), (64, 1, 1), (71, 1, 2), (4, 1, 1)])
def test_hash_parity(pm, oracle, k, canon, nh):
    rng = np.random.default_rng(100 + k + canon + nh)
    seqs = [rand_seq(rng, int(n)) for n in list(rng.integers(k, k + 300, size=40)) + [k, k + 1, k + 7, k + 8, k + 9]]
    seqs.append("A" * (k + 5))
    seqs.append(("ACGT" * 40)[: k + 20])  # revcomp palindromes (ACGT) for even k
    fasta = "".join(f">q{i}\n{s}\n" for i, s in enumerate(seqs)).encode()
    q = pm.Queries(fasta, term_size=k)
    got = q.hash_terms(canonicalize=canon, num_hashes=nh)
    exp = np.concatenate([oracle.create_hashes(s.encode(), k, canon, nh) for s in seqs])
    assert got.dtype == np.uint64 and got.shape == exp.shape
    assert np.array_equal(got, exp)


@pytest.mark.parametrize("layout", [0, 1, 2])
@pytest.mark.parametrize("n_docs", [1, 8, 13, 100, 129, 664, 1300, 4000, 9001])
def test_index_roundtrip(pm, oracle, layout, n_docs):
    rng = np.random.default_rng(n_docs)
    S = 777
    index, _, matrix = build_case(oracle, rng, n_docs, S, [("q", rand_seq(rng, 40))])
    ix = pm.Index.load_mem(index, layout=layout)
    info = ix.info
    assert (info.n_docs, info.signature_size, info.term_size, info.canonicalize, info.num_hashes) == (n_docs, S, 31, 1, 1)
    assert info.row_bytes == (n_docs + 7) // 8 and info.stride >= info.row_bytes and info.stride % 16 == 0
    for r in [0, 1, S // 2, S - 1]:
        assert np.array_equal(ix.read_row(r), matrix[r])
    h = oracle.header_parse(index)
    assert ix.doc_name(0) == bytes(index[h.names_off:]).split(b"\n")[0].decode()


CASES = [
    # n_docs, sig, nq, qlen, k, canon, nh, threshold
    (5, 500, 12, 150, 31, 1, 1, 0.7),
    (64, 3000, 30, 150, 31, 1, 1, 0.7),
    (100, 3000, 30, 100, 31, 1, 2, 0.5),
    (130, 2000, 25, 150, 31, 0, 1, 0.7),
    (195, 4000, 40, 150, 31, 1, 1, 0.7),
    (664, 5000, 40, 150, 31, 1, 1, 0.7),
    (1000, 2000, 20, 60, 21, 1, 3, 0.3),
    (2300, 3000, 24, 150, 31, 1, 1, 0.7),
    (4000, 4000, 33, 150, 31, 1, 1, 0.7),
    (9001, 1500, 9, 150, 31, 1, 1, 0.7),
    (17000, 700, 5, 90, 31, 1, 2, 0.6),
    (664, 5000, 20, 31, 31, 1, 1, 0.7),     # one k-mer per query: hit <=> bit set
    (300, 2000, 16, 150, 31, 1, 1, 0.0),    # threshold 0 keeps every document
    (300, 2000, 16, 150, 31, 1, 1, 1.0),
]


@pytest.fixture
def rules(request, pm, oracle):
    """(cobs_threshold_rule, cobs_tie_order) set on both the product and the oracle for one test, defaults restored"""
    import gc
    thr_rule, tie = request.param
    gc.collect()                        # the rules cannot change while a search result is alive (include/phylign_match.h)
    pm.set_option("cobs_threshold_rule", thr_rule)
    pm.set_option("cobs_tie_order", tie)
    oracle.set_rules(thr_rule, tie)
    yield request.param
    gc.collect()
    pm.set_option("cobs_threshold_rule", 0)
    pm.set_option("cobs_tie_order", 0)
    oracle.set_rules(0, 0)


@pytest.mark.parametrize("case", CASES, ids=lambda c: "D%d_S%d_q%dx%d_k%d_c%d_h%d_t%s" % c)
@pytest.mark.parametrize("layout", [1, 2])
@pytest.mark.parametrize("rules", [(0, 0), (1, 0), (2, 1), (0, 1)], indirect=True, ids=["ceil_asc", "floor_asc", "round_desc", "ceil_desc"])
def test_query_text_bit_exact(pm, oracle, case, layout, rules):
    """the two `cobs query` rules nothing pins -- how -t becomes a minimum score, how equal scores are listed -- are
    switches on both sides (pm_set_option / oracle.set_rules): whichever a real cobs 0.2.1 turns out to use
    (tools/pin_against_cobs.sh), this test already covers it.  Default: ceil, ascending document."""
    n_docs, S, nq, qlen, k, canon, nh, thr = case
    rng = np.random.default_rng(hash(case) % (2**32))
    queries = [(f"read{i} comment {i}" if i % 3 == 0 else f"read{i}", rand_seq(rng, qlen)) for i in range(nq)]
    plant = []
    for qi in range(0, nq, 2):
        for frac in (1.0, 0.8, 0.7, 0.69, 0.6, 0.71):
            plant.append((qi, int(rng.integers(0, n_docs)), frac))
        # tie group: several docs at the same planted fraction
        for d in rng.integers(0, n_docs, size=3):
            plant.append((qi, int(d), 0.75))
    index, fasta, _ = build_case(oracle, rng, n_docs, S, queries, k=k, canon=canon, num_hashes=nh, plant=plant)
    exp = oracle.query_file(index, fasta, thr)
    ix = pm.Index.load_mem(index, layout=layout)
    got = pm.query_text(ix, fasta, thr)
    assert got == exp
    assert got.count(b"*") == nq


def test_mixed_lengths_and_plane_classes(pm, oracle):
    """queries of 1 k-mer ... 1.05 M k-mers in one FASTA: every counter-width class (3/7/10/13/16/20/24 planes)."""
    rng = np.random.default_rng(5)
    lens = [31, 32, 37, 38, 100, 157, 158, 160, 400, 1053, 1054, 3000, 8221, 8222, 9000, 65565, 65566, 70000, 150, 31, 1048605, 1048606]
    queries = [(f"g{i}", rand_seq(rng, n)) for i, n in enumerate(lens)]
    plant = [(i, int(rng.integers(0, 200)), f) for i in range(len(lens)) for f in (1.0, 0.7, 0.5)]
    index, fasta, _ = build_case(oracle, rng, 200, 9000, queries, plant=plant, density=0.3)
    ix = pm.Index.load_mem(index)
    for thr in (0.7, 0.28):
        assert pm.query_text(ix, fasta, thr) == oracle.query_file(index, fasta, thr)
    res = pm.search([ix], pm.Queries(fasta), 0.7)
    assert {L["kernel"].split("P=")[1].split(",")[0] for L in res.launches()} == {"3", "7", "10", "13", "16", "20", "24"}


@pytest.mark.parametrize("n_docs,S", [(664, 30011), (2300, 9001), (100, 50021)])
def test_argannot_length_mix(pm, oracle, n_docs, S):
    """SURVEY.md 8d's third query shape: the 1 856 record lengths of the reference's data/ARGannot_r3.fa in file order
    (237 ... 3 153 bp: the 10- and the 13-plane counter classes, 1 594 532 k-mers), with planted documents at fractions that
    straddle the threshold in both classes; text byte-identical to the oracle's in both scan modes, and with the
    post-filter cut (nb_best_hits) fused on the device"""
    from phylign_amd import workload as W
    lens = W.argannot_lengths()
    assert len(lens) == 1856 and sum(n - 30 for n in lens) == 1594532 and (min(lens), max(lens)) == (237, 3153)
    rng = np.random.default_rng(n_docs)
    fasta, seqs = W.make_queries_lengths(lens, seed=77, prefix="gene")
    queries = [(f"gene{i:07d}", s_.decode()) for i, s_ in enumerate(seqs)]
    plant = [(qi, int(rng.integers(0, n_docs)), f) for qi in range(0, len(lens), 23) for f in (1.0, 0.9, 0.7, 0.7, 0.69, 0.5)]
    index, fasta2, _ = build_case(oracle, rng, n_docs, S, queries, plant=plant, density=0.2)
    assert fasta2 == fasta
    ix = pm.Index.load_mem(index)
    exp = oracle.query_file(index, fasta, 0.7)
    assert exp.count(b"\n") > 1856 + 3 * len(plant) // 6
    try:
        for bound in (1, 0):
            pm.set_option("threshold_bound", bound)
            assert pm.query_text(ix, fasta, 0.7) == exp
        q = pm.Queries(fasta)
        res = pm.search([ix], q, 0.7)
        assert {L["kernel"].split("P=")[1].split(",")[0] for L in res.launches()} == {"10", "13"}
        cut = pm.search([ix], q, 0.7, nb_best_hits=2)
        from phylign_amd import postprocess
        assert pm.format_hits(ix, q, cut.hits(), nb_best_hits=2) == postprocess.filter_text(exp.decode(), 2).encode()
    finally:
        pm.set_option("threshold_bound", 1)


def test_large_index_file_goes_through_the_parallel_reader(pm, oracle, tmp_path):
    """a decompressed index on disk (rule decompress_cobs / index_load_mode mem-disk): files of 256 MiB and more are
    read by several pread() workers with pooled pinned staging; every row lands where the serial pipe path puts it,
    a truncated file is PM_EIO, and the same fd positioned past a prefix still loads"""
    rng = np.random.default_rng(21)
    n_docs, S = 2371, 1_000_003                       # 297-byte rows -> 384-byte stride, 297 MB of matrix, 10 chunks
    names = [f"{i:05x}_DOC{i}" for i in range(n_docs)]
    idx = oracle.make_index(31, 1, S, 1, names)
    h = oracle.header_parse(idx)
    rb = (n_docs + 7) // 8
    body = rng.integers(0, 256, size=S * rb, dtype=np.uint8)
    body.reshape(S, rb)[:, -1] &= np.uint8((1 << (n_docs - (rb - 1) * 8)) - 1)
    idx[h.data_off:] = body
    path = tmp_path / "big__01.cobs_classic"
    idx.tofile(path)
    ix = pm.Index.load_file(str(path))
    piped = pm.Index.load_mem(idx)                    # the serial reader
    for r in (0, 1, 87_381, 87_382, 500_000, S - 2, S - 1):          # chunk borders of 32 MiB / 297 B among them
        want = body[r * rb:(r + 1) * rb]
        assert np.array_equal(ix.read_row(r), want) and np.array_equal(piped.read_row(r), want), r
    assert np.array_equal(ix.read_rows(262_140, 3000), piped.read_rows(262_140, 3000))
    ix.free(); piped.free()
    with open(path, "rb") as f:                       # an fd that does not start at the index
        pre = tmp_path / "prefixed.bin"
        pre.write_bytes(b"x" * 4096 + f.read())
    fd = os.open(pre, os.O_RDONLY)
    os.lseek(fd, 4096, os.SEEK_SET)
    ix2 = pm.Index.load_fd(fd)
    os.close(fd)
    assert np.array_equal(ix2.read_row(S - 1), body[(S - 1) * rb:])
    ix2.free()
    with open(path, "r+b") as f:
        f.truncate(h.data_off + (S - 5) * rb)
    with pytest.raises(pm.PMError) as e:
        pm.Index.load_file(str(path))
    assert e.value.code == -4


def test_header_with_an_impossible_row_count_is_rejected_before_allocation(pm, oracle):
    """a header whose signature_size makes rows x stride wrap around 2^64 (or just exceed HBM) is PM_EFORMAT:
    nothing is allocated, nothing is uploaded next to other resident indexes"""
    import struct
    rng = np.random.default_rng(8)
    index, _, _ = build_case(oracle, rng, 100, 777, [("a", rand_seq(rng, 50))])
    raw = bytes(index)
    at = raw.index(struct.pack("<Q", 777), 18, 80)
    free0 = pm.device_info()["hbm_free"]
    for sig in (1 << 61, (1 << 64) // 16 + 1, 1 << 40):
        bad = raw[:at] + struct.pack("<Q", sig) + raw[at + 8:]
        with pytest.raises(pm.PMError) as e:
            pm.Index.load_mem(bad)
        assert e.value.code == -5, e.value
    assert pm.device_info()["hbm_free"] >= free0 - (64 << 20)
    assert pm.Index.load_mem(raw).info.signature_size == 777


def test_fasta_record_rules(pm, oracle):
    rng = np.random.default_rng(6)
    s = [rand_seq(rng, 90) for _ in range(4)]
    fasta = (f"\n>a first\n{s[0][:40]}\n{s[0][40:]}\n\n;b semicolon header\n{s[1]}\n>empty\n>c\n{s[2]}\n>d\tTAB\n{s[3]}").encode()
    index, _, _ = build_case(oracle, rng, 77, 1000, [("a", s[0]), ("b", s[1]), ("c", s[2])],
                             plant=[(0, 3, 1.0), (1, 70, 0.9), (2, 76, 0.8)])
    ix = pm.Index.load_mem(index)
    got = pm.query_text(ix, fasta, 0.7)
    assert got == oracle.query_file(index, fasta, 0.7)
    assert got.startswith(b"*a first\t") and b"*empty" not in got and b"*b semicolon header\t" in got
    assert pm.query_text(ix, b"", 0.7) == b"" == oracle.query_file(index, b"", 0.7)


def test_query_errors(pm, oracle):
    rng = np.random.default_rng(7)
    index, _, _ = build_case(oracle, rng, 20, 100, [("a", rand_seq(rng, 50))])
    ix = pm.Index.load_mem(index)
    with pytest.raises(pm.PMError) as e:
        pm.query_text(ix, b">short\nACGTACGT\n", 0.7)
    assert e.value.code == -6
    with pytest.raises(pm.PMError) as e:
        pm.query_text(ix, (">n\n" + "ACGTN" * 10 + "\n").encode(), 0.7)
    assert e.value.code == -6
    with pytest.raises(pm.PMError) as e:
        pm.Index.load_mem(b"COBS:NOT_AN_INDEX" + bytes(100))
    assert e.value.code == -5
    with pytest.raises(pm.PMError) as e:
        pm.Index.load_mem(bytes(index[: len(index) - 10]))
    assert e.value.code == -4


def test_streaming_pipe_load(pm, oracle, tmp_path):
    """index arrives on a non-seekable pipe with --index-sizes (run_cobs_streaming.sh:27-28)."""
    import threading
    rng = np.random.default_rng(8)
    queries = [(f"r{i}", rand_seq(rng, 150)) for i in range(10)]
    index, fasta, _ = build_case(oracle, rng, 664, 60000, queries, plant=[(i, i * 7, 0.9) for i in range(10)])
    r, w = os.pipe()

    def feed():
        with os.fdopen(w, "wb") as f:
            b = bytes(index)
            for o in range(0, len(b), 70001):
                f.write(b[o:o + 70001])
    t = threading.Thread(target=feed)
    t.start()
    ix = pm.Index.load_fd(r, size_hint=len(index))
    t.join()
    os.close(r)
    assert pm.query_text(ix, fasta, 0.7) == oracle.query_file(index, fasta, 0.7)
    p = tmp_path / "x.cobs_classic"
    p.write_bytes(bytes(index))
    ix2 = pm.Index.load_file(str(p))
    assert pm.query_text(ix2, fasta, 0.7) == oracle.query_file(index, fasta, 0.7)


@pytest.mark.parametrize("n_docs,S,batch", [(664, 5000, 3), (4000, 3000, 0), (195, 4000, 7), (13, 900, 1), (9001, 300, 2)])
def test_synth_generator_matches_spec(pm, oracle, n_docs, S, batch):
    ix = pm.Index.synth(batch, n_docs, S, seed=661)
    for r in [0, 1, 2, S // 3, S - 1]:
        assert np.array_equal(ix.read_row(r), oracle.synth_row(661, batch, r, n_docs))
    # density ~ 1/4
    rows = np.stack([ix.read_row(r) for r in range(0, S, max(1, S // 50))])
    dens = np.unpackbits(rows, axis=1, bitorder="little")[:, :n_docs].mean()
    assert 0.2 < dens < 0.3


def test_synth_search_with_planted_hits(pm, oracle):
    """synthetic batch + planted hits vs the oracle evaluated on a virtual matrix."""
    rng = np.random.default_rng(9)
    n_docs, S, batch = 664, 200000, 5
    ix = pm.Index.synth(batch, n_docs, S, seed=661)
    queries = [(f"s{i}", rand_seq(rng, 150)) for i in range(50)]
    fasta = "".join(f">{h}\n{s}\n" for h, s in queries).encode()
    overlay = {}
    rows, docs = [], []
    for qi in range(0, 50, 3):
        hs = oracle.create_hashes(queries[qi][1].encode(), 31, 1, 1)
        for doc, frac in ((int(rng.integers(0, n_docs)), 1.0), (int(rng.integers(0, n_docs)), 0.7), (int(rng.integers(0, n_docs)), 0.65)):
            for t in range(int(np.ceil(frac * len(hs)))):
                r = int(hs[t]) % S
                rows.append(r); docs.append(doc)
                overlay.setdefault(r, []).append(doc)
    ix.plant(rows, docs)

    def row_fn(r):
        v = oracle.synth_row(661, batch, r, n_docs)
        for d in overlay.get(r, ()):
            v[d >> 3] |= np.uint8(1 << (d & 7))
        return v
    qobj = pm.Queries(fasta)
    res = pm.search([ix], qobj, 0.7)
    hits = res.hits()
    h = oracle.Header(); h.term_size = 31; h.canonicalize = 1; h.num_hashes = 1; h.n_docs = n_docs; h.signature_size = S
    exp = []
    for qi, (_, s) in enumerate(queries):
        sc = oracle.scores_rows(row_fn, h, s.encode())
        for d, v in oracle.select(sc, 120, 0.7):
            exp.append((qi, d, v))
    got = [(int(x["query"]), int(x["doc"]), int(x["score"])) for x in hits]
    assert got == exp and len(exp) >= 17 * 2
    st = res.stats
    assert st.n_queries == 50 and st.n_terms == 50 * 120 and st.algorithmic_bytes == 50 * 120 * 83


def test_multi_index_search_slots(pm, oracle):
    rng = np.random.default_rng(10)
    queries = [(f"m{i}", rand_seq(rng, 150)) for i in range(20)]
    cases = []
    for n_docs, S in ((100, 3000), (664, 2000), (4000, 1000)):
        cases.append(build_case(oracle, rng, n_docs, S, queries, plant=[(i, (i * 13) % n_docs, 0.9) for i in range(0, 20, 2)]))
    fasta = cases[0][1]
    ixs = [pm.Index.load_mem(c[0]) for c in cases]
    q = pm.Queries(fasta)
    res = pm.search(ixs, q, 0.7, slot_base=40)
    hits = res.hits()
    assert set(hits["slot"]) <= {40, 41, 42}
    for s, (index, _, _) in enumerate(cases):
        assert pm.format_hits(ixs[s], q, hits, slot=40 + s) == oracle.query_file(index, fasta, 0.7)


@pytest.mark.parametrize("n_docs", [13, 100, 300, 664, 1300, 4000, 9001])
def test_on_device_top_n_with_ties(pm, oracle, n_docs):
    """search(nb_best_hits=n) keeps exactly the n best documents + ties with the n-th per
    (query, batch): the rule of scripts/postprocess_cobs.py:31-39, checked against the unpruned hits."""
    rng = np.random.default_rng(n_docs)
    queries = [(f"t{i}", rand_seq(rng, 150)) for i in range(16)]
    plant = []
    for qi in range(16):
        docs = rng.choice(n_docs, size=min(n_docs, 40), replace=False)
        for j, d in enumerate(docs):
            plant.append((qi, int(d), [1.0, 0.95, 0.9, 0.9, 0.9, 0.85, 0.8, 0.8, 0.75, 0.7][j % 10]))
    index, fasta, _ = build_case(oracle, rng, n_docs, 4000, queries, plant=plant, density=0.05)
    ix = pm.Index.load_mem(index)
    q = pm.Queries(fasta)
    full = pm.search([ix], q, 0.7).hits()
    assert len(full) > 16 * min(n_docs, 40) * 0.8
    for n in (1, 2, 3, 5, 7, 20, 39, 1000):
        got = pm.search([ix], q, 0.7, nb_best_hits=n).hits()
        exp, exp_counts = [], {}
        for qi in range(16):
            mine = full[full["query"] == qi]            # already ordered score desc, doc asc
            if len(mine) > n:
                kept = mine[mine["score"] >= mine["score"][n - 1]]
                if len(kept) != len(mine):              # a count record only where lines were really dropped
                    exp_counts[qi] = len(mine)
                mine = kept
            exp.append(mine)
        exp = np.concatenate(exp)
        if n_docs > 8192:                                # column-slab batches are pruned at format time
            assert np.array_equal(got, full)
        else:
            meta = got[got["doc"] == pm.PM_DOC_COUNT]
            assert np.array_equal(got[got["doc"] != pm.PM_DOC_COUNT], exp), n
            assert {int(m["query"]): int(m["score"]) for m in meta} == exp_counts
        from phylign_amd import postprocess as P
        text = pm.format_hits(ix, q, got, slot=0, nb_best_hits=n).decode()
        assert text == P.filter_text(oracle.query_file(index, fasta, 0.7).decode(), n)


@pytest.mark.parametrize("page,n_docs,sigs,nhs", [
    (16, 300, [500, 700, 400], [1, 2, 1]),           # 128 documents per sub-index, last one partly filled
    (64, 1024, [900, 800], [1, 1]),
    (128, 2500, [600, 500, 400], [2, 1, 3]),
    (2048, 20000, [300, 200], [1, 1]),               # 16384 documents per sub-index: column slabs
])
@pytest.mark.parametrize("rules", [(0, 0), (2, 1)], indirect=True, ids=["ceil_asc", "round_desc"])
def test_compact_index_text_bit_exact(pm, oracle, page, n_docs, sigs, nhs, tmp_path, rules):
    """COMPACT_INDEX files: sub-indexes with their own signature sizes and hash counts
    (under both settings of the unpinned rules: the runs of several sub-indexes are merged on the device)"""
    from helpers import doc_names
    from phylign_amd import postprocess as P
    rng = np.random.default_rng(page + n_docs)
    queries = [(f"c{i} x", rand_seq(rng, 150)) for i in range(14)]
    per = page * 8
    mats = []
    for p, S in enumerate(sigs):
        bits = rng.random((S, per)) < 0.2
        lo = p * per
        if lo + per > n_docs:
            bits[:, max(0, n_docs - lo):] = False
        mats.append(np.packbits(bits, axis=1, bitorder="little"))
    for qi in range(0, 14, 2):                       # plant across sub-indexes
        seq = queries[qi][1].encode()
        for d, frac in ((int(rng.integers(0, n_docs)), 1.0), (int(rng.integers(0, n_docs)), 0.8), (n_docs - 1, 0.75), (0, 0.7)):
            p, dl = d // per, d % per
            hs = oracle.create_hashes(seq, 31, 1, nhs[p]).reshape(-1, nhs[p])
            for t in range(int(np.ceil(frac * len(hs)))):
                for j in range(nhs[p]):
                    mats[p][int(hs[t, j]) % sigs[p], dl >> 3] |= np.uint8(1 << (dl & 7))
    names = doc_names(rng, n_docs)
    index = oracle.make_compact(31, 1, page, sigs, nhs, names, mats)
    fasta = "".join(f">{h}\n{s}\n" for h, s in queries).encode()
    exp = oracle.query_file(index, fasta, 0.7)
    assert exp.count(b"\n") > 14 + 7 * 3
    for layout in (1, 2):
        ix = pm.Index.load_mem(index, layout=layout)
        info = ix.info
        assert (info.n_parts, info.page_size, info.n_docs) == (len(sigs), page, n_docs)
        assert pm.query_text(ix, fasta, 0.7) == exp
        assert pm.query_text(ix, fasta, 0.7, nb_best_hits=2).decode() == P.filter_text(exp.decode(), 2)
    p = tmp_path / "x.cobs_compact"
    p.write_bytes(bytes(index))
    ix = pm.Index.load_file(str(p))
    assert pm.query_text(ix, fasta, 0.0) == oracle.query_file(index, fasta, 0.0)
    q = pm.Queries(fasta)
    st = pm.search([ix], q, 0.7).stats
    assert st.algorithmic_bytes == 14 * 120 * sum(nh * page for nh in nhs)


@pytest.mark.parametrize("thr", [0.0, 0.1, 0.5, 0.7, 0.9, 1.0])
def test_threshold_bound_never_changes_results(pm, oracle, thr):
    """the scan stops fetching lines whose documents cannot reach the minimum score any more:
    records (hits, scores, count records) must equal the fetch-everything scan and the oracle"""
    rng = np.random.default_rng(int(thr * 100))
    lens = [150] * 30 + [31, 38, 39, 40, 47, 250, 1100]
    queries = [(f"b{i}", rand_seq(rng, n)) for i, n in enumerate(lens)]
    for n_docs, S in ((664, 30000), (4000, 8000), (100, 20000), (9001, 2000)):
        plant = [(qi, int(rng.integers(0, n_docs)), fr) for qi in range(0, len(lens), 2)
                 for fr in (1.0, 0.95, 0.72, 0.7, 0.69, 0.5, 0.3, 0.1)]
        index, fasta, _ = build_case(oracle, rng, n_docs, S, queries, plant=plant)
        ix = pm.Index.load_mem(index)
        q = pm.Queries(fasta)
        out = {}
        for mode in (1, 0):
            pm.set_option("threshold_bound", mode)
            out[mode] = (pm.search([ix], q, thr).hits(), pm.search([ix], q, thr, nb_best_hits=3).hits())
        pm.set_option("threshold_bound", 1)
        assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
        assert pm.format_hits(ix, q, out[1][0], slot=0) == oracle.query_file(index, fasta, thr)


def test_mixed_width_launch_with_every_counter_class(pm, oracle):
    """several narrow batches of different lane-group widths (one mixed-width launch) x queries of
    every counter-width class, plus a wide and a column-slab batch, in ONE search"""
    rng = np.random.default_rng(77)
    lens = [31, 37, 38, 150, 150, 158, 159, 400, 1054, 1055, 3000, 8221, 8222, 65566, 150, 40]
    queries = [(f"w{i}", rand_seq(rng, n)) for i, n in enumerate(lens)]
    shapes = [(13, 900), (100, 700), (200, 800), (300, 600), (664, 900), (1500, 500), (4000, 400), (9001, 300), (50, 1000)]
    cases = []
    for n_docs, S in shapes:
        plant = [(qi, int(rng.integers(0, n_docs)), fr) for qi in range(len(lens)) for fr in (1.0, 0.7, 0.5)]
        cases.append(build_case(oracle, rng, n_docs, S, queries, plant=plant, density=0.3))
    fasta = cases[0][1]
    ixs = [pm.Index.load_mem(c[0]) for c in cases]
    assert len({ix.info.stride for ix in ixs}) >= 7
    q = pm.Queries(fasta)
    for thr in (0.7, 0.3):
        res = pm.search(ixs, q, thr, slot_base=5)
        kernels = [L["kernel"] for L in res.launches()]
        assert any("G=mixed" in k for k in kernels) and any("G=32" in k for k in kernels)
        assert {k.split("P=")[1].split(",")[0] for k in kernels} == {"3", "7", "10", "13", "16", "20"}
        hits = res.hits()
        for s, (index, _, _) in enumerate(cases):
            assert pm.format_hits(ixs[s], q, hits, slot=5 + s) == oracle.query_file(index, fasta, thr), shapes[s]


@pytest.mark.parametrize("rules", [(0, 0), (0, 1)], indirect=True, ids=["asc", "desc"])
def test_many_runs_per_query_merge_counting_sort_equals_pairwise(pm, oracle, rules):
    """a compact index of 157 sub-indexes (128 documents each): every (query, index) hit list arrives as up to 157 runs.
    The counting-sort merge (scores below 4096, the default) and the general pairwise merge give the same records, equal
    to the oracle's text; a 6-kbp query (score range beyond the histogram) takes the pairwise form inside the same search"""
    import time
    from helpers import doc_names
    rng = np.random.default_rng(157)
    page, n_docs = 16, 20000
    per = page * 8
    n_parts = (n_docs + per - 1) // per
    sigs, nhs = [int(rng.integers(300, 600)) for _ in range(n_parts)], [1] * n_parts
    queries = [(f"m{i}", rand_seq(rng, 150)) for i in range(40)] + [("long", rand_seq(rng, 6000))]
    mats = []
    for p, S in enumerate(sigs):
        bits = rng.random((S, per)) < 0.3
        if p * per + per > n_docs:
            bits[:, n_docs - p * per:] = False
        mats.append(np.packbits(bits, axis=1, bitorder="little"))
    index = oracle.make_compact(31, 1, page, sigs, nhs, doc_names(rng, n_docs), mats)
    fasta = "".join(f">{h}\n{s}\n" for h, s in queries).encode()
    ix = pm.Index.load_mem(index)
    q = pm.Queries(fasta)
    try:
        for thr in (0.0, 0.3):
            got, took = {}, {}
            for mode in (1, 0):
                pm.set_option("merge_counting_sort", mode)
                res = pm.search([ix], q, thr)
                t0 = time.perf_counter()
                got[mode] = res.hits()
                took[mode] = time.perf_counter() - t0
            assert np.array_equal(got[0], got[1]), thr
            assert len(got[1]) > 41 * 100
            assert pm.format_hits(ix, q, got[1]) == oracle.query_file(index, fasta, thr)
            print(f"thr {thr}: {len(got[1])} records, ordering + read-back {took[1] * 1e3:.1f} ms (counting sort) vs {took[0] * 1e3:.1f} ms (pairwise)")
    finally:
        pm.set_option("merge_counting_sort", 1)

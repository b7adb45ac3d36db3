"""Randomised differential test: the HIP path vs the oracle on random index shapes,
k-mer sizes, hash counts, thresholds, query lengths and formats (seeded, 60 cases)."""
import os

import numpy as np
import pytest

from helpers import build_case, doc_names, rand_seq

pytestmark = pytest.mark.gpu

# PM_FUZZ_EXTRA=N adds N more seeded cases to each test (from seed PM_FUZZ_OFFSET on): the wide sweeps of
# tools/fuzz_sweep.sh; the default suite keeps its fixed 60 + 24 cases
_EXTRA = int(os.environ.get("PM_FUZZ_EXTRA", "0"))
_OFFSET = int(os.environ.get("PM_FUZZ_OFFSET", "100"))


def _seeds(n):
    return list(range(n)) + list(range(_OFFSET, _OFFSET + _EXTRA))


@pytest.mark.parametrize("seed", _seeds(60))
def test_random_case_text_identical(pm, oracle, seed):
    from phylign_amd import postprocess as P
    rng = np.random.default_rng(1000 + seed)
    k = int(rng.choice([5, 11, 15, 21, 27, 31, 31, 31, 32, 33, 47]))
    canon = int(rng.integers(0, 2))
    nh = int(rng.choice([1, 1, 1, 2, 3, 4]))
    n_docs = int(rng.choice([1, 7, 8, 9, 63, 64, 65, 127, 128, 129, 200, 511, 512, 513, 1000, 1024, 1025,
                             2047, 3000, 4000, 4097, 8191, 8192, 8193, 12000]))
    S = int(rng.integers(50, 4000))
    thr = float(rng.choice([0.0, 0.2, 0.5, 0.7, 0.7, 0.7, 0.9, 1.0]))
    nq = int(rng.integers(1, 25))
    lens = [int(rng.choice([k, k + 1, k + 6, k + 7, k + 8, 60, 100, 150, 150, 200, 400, 1100])) for _ in range(nq)]
    lens = [max(l, k) for l in lens]
    queries = [(f"r{i}" + (" c o m" if i % 4 == 0 else ""), rand_seq(rng, lens[i])) for i in range(nq)]
    plant = [(int(rng.integers(0, nq)), int(rng.integers(0, n_docs)), float(rng.choice([1.0, 0.9, 0.71, 0.7, 0.69, 0.5, 0.2])))
             for _ in range(int(rng.integers(0, 40)))]
    density = float(rng.choice([0.02, 0.25, 0.5]))
    compact = seed % 5 == 4 and n_docs >= 64
    if not compact:
        index, fasta, _ = build_case(oracle, rng, n_docs, S, queries, k=k, canon=canon, num_hashes=nh,
                                     density=density, plant=plant)
    else:
        page = int(rng.choice([8, 16, 32]))
        per = page * 8
        parts = (n_docs + per - 1) // per
        sigs = [int(rng.integers(50, 900)) for _ in range(parts)]
        nhs = [int(rng.integers(1, 4)) for _ in range(parts)]
        mats = []
        for p in range(parts):
            bits = rng.random((sigs[p], per)) < density
            if p * per + per > n_docs:
                bits[:, n_docs - p * per:] = False
            mats.append(np.packbits(bits, axis=1, bitorder="little"))
        for qi, d, frac in plant:
            p, dl = d // per, d % per
            hs = oracle.create_hashes(queries[qi][1].encode(), k, canon, nhs[p]).reshape(-1, nhs[p])
            for t in range(int(np.ceil(frac * len(hs)))):
                for j in range(nhs[p]):
                    mats[p][int(hs[t, j]) % sigs[p], dl >> 3] |= np.uint8(1 << (dl & 7))
        index = oracle.make_compact(k, canon, page, sigs, nhs, doc_names(rng, n_docs), mats)
        fasta = "".join(f">{h}\n{s}\n" for h, s in queries).encode()
    exp = oracle.query_file(index, fasta, thr)
    ix = pm.Index.load_mem(index, layout=int(rng.integers(0, 3)))
    assert pm.query_text(ix, fasta, thr) == exp
    n = int(rng.choice([1, 2, 5, 100]))
    assert pm.query_text(ix, fasta, thr, nb_best_hits=n).decode() == P.filter_text(exp.decode(), n)
    # the same reads as a file the record rules have to work for: wrapped sequence lines, empty lines, ';' headers, a
    # first record without header line, a header without sequence, no final newline
    lines = []
    for i, (h, sq) in enumerate(queries):
        if not (i == 0 and rng.random() < 0.3):
            lines.append((";" if rng.random() < 0.2 else ">") + h)
        w = int(rng.integers(7, 80))
        for j in range(0, len(sq), w):
            lines.append(sq[j:j + w])
            if rng.random() < 0.1:
                lines.append("")
        if rng.random() < 0.15:
            lines.append(">no_sequence_%d" % i)
    odd = ("\n".join(lines) + ("\n" if rng.random() < 0.8 else "")).encode()
    assert pm.query_text(ix, odd, thr) == oracle.query_file(index, odd, thr)


@pytest.mark.parametrize("seed", _seeds(24))
def test_random_multi_index_search_text_identical(pm, oracle, seed):
    """several random batches of different row widths in ONE search (fused launches, mixed-width
    launch, column slabs), dense and sparse hit lists, both scan modes, sync and in flight: the
    text of every slot equals the oracle's for that index"""
    from phylign_amd import postprocess as P
    rng = np.random.default_rng(5000 + seed)
    nq = int(rng.integers(1, 40))
    lens = [int(rng.choice([31, 38, 39, 40, 100, 150, 150, 300, 1200])) for _ in range(nq)]
    queries = [(f"m{i}", rand_seq(rng, lens[i])) for i in range(nq)]
    thr = float(rng.choice([0.0, 0.3, 0.7, 0.7, 1.0]))
    n_idx = int(rng.integers(2, 7))
    cases = []
    for _ in range(n_idx):
        n_docs = int(rng.choice([3, 64, 100, 130, 300, 600, 1024, 2100, 4000, 4000, 8300]))
        S = int(rng.integers(60, 2500))
        density = float(rng.choice([0.02, 0.25, 0.25, 0.6]))
        plant = [(int(rng.integers(0, nq)), int(rng.integers(0, n_docs)), float(rng.choice([1.0, 0.9, 0.75, 0.7, 0.69, 0.4])))
                 for _ in range(int(rng.integers(0, 60)))]
        cases.append(build_case(oracle, rng, n_docs, S, queries, density=density, plant=plant))
    fasta = cases[0][1]
    ixs = [pm.Index.load_mem(c[0], layout=int(rng.integers(0, 3))) for c in cases]
    q = pm.Queries(fasta)
    base = int(rng.integers(0, 1000))
    n = int(rng.choice([0, 1, 3, 100]))
    exp = [oracle.query_file(c[0], fasta, thr) for c in cases]
    got = {}
    try:
        # scan modes x wide-query form (auto, forced, off) x its split over workgroups (auto, 3 / 7 ways) x one launch
        # for all row widths
        for bound, wq, split, single in ((1, 0, 0, 0), (0, 0, 0, 0), (1, 1, 0, 0), (0, 2, 0, 0), (0, 1, 3, 0), (1, 1, 7, 1),
                                         (0, 0, 0, 1)):
            pm.set_option("threshold_bound", bound)
            pm.set_option("wide_query", wq)
            pm.set_option("wide_query_split", split)
            pm.set_option("single_launch", single)
            r1 = pm.search_async(ixs, q, thr, slot_base=base, nb_best_hits=n)
            r2 = pm.search_async(ixs, q, thr, slot_base=base)
            got[(bound, wq, split, single)] = (r1.hits(), r2.hits())
            if (bound, wq, split, single) == (1, 0, 0, 0):
                # the records of one index read back on their own (what the stage's workers do) = that slot's slice
                for r, h in ((r1, got[(1, 0, 0, 0)][0]), (r2, got[(1, 0, 0, 0)][1])):
                    for s_ in range(len(ixs)):
                        with r.slot_hits(s_) as sl:
                            assert np.array_equal(sl.hits, h[h["slot"] == base + s_]), s_
                with pytest.raises(pm.PMError):
                    r1.slot_hits(len(ixs))
    finally:
        pm.set_option("threshold_bound", 1)
        pm.set_option("wide_query", 0)
        pm.set_option("wide_query_split", 0)
        pm.set_option("single_launch", 0)
    for key in got:
        assert np.array_equal(got[key][0], got[(1, 0, 0, 0)][0]) and np.array_equal(got[key][1], got[(1, 0, 0, 0)][1]), key
    pruned, plain = got[(1, 0, 0, 0)]
    for s, ix in enumerate(ixs):
        assert pm.format_hits(ix, q, plain, slot=base + s) == exp[s]
        if n:
            assert pm.format_hits(ix, q, pruned, slot=base + s, nb_best_hits=n).decode() == P.filter_text(exp[s].decode(), n)


@pytest.mark.parametrize("seed", _seeds(12))
def test_random_stage_run_files_and_fasta_identical(pm, oracle, tmp_path, seed):
    """match_stage.run_stage on random resident batches: random grouping (--max-group), the query file whole, in record
    chunks or in byte pieces that arrive as Futures, repeated read names now and then (also across pieces) -- every 03_match file equals the
    oracle's text after the post-filter, the 04_filter FASTA equals the (fixture-pinned) mirror of filter_queries.py"""
    import gzip
    import io
    from concurrent.futures import ThreadPoolExecutor
    from phylign_amd import filter_queries as F
    from phylign_amd import match_stage as MS
    from phylign_amd import postprocess as P
    rng = np.random.default_rng(9000 + seed)
    nq = int(rng.integers(1, 70))
    names_q = [f"r{i}" for i in range(nq)]
    if nq > 4 and rng.random() < 0.4:                                 # a repeated name: dict semantics in the merge
        for _ in range(int(rng.integers(1, 4))):
            names_q[int(rng.integers(0, nq))] = names_q[int(rng.integers(0, nq))]
    queries = [(names_q[i] + (" note" if i % 3 == 0 else ""), rand_seq(rng, int(rng.choice([31, 40, 100, 150, 150, 300, 900]))))
               for i in range(nq)]
    n_b = int(rng.integers(1, 8))
    batches, indexes = [], {}
    for b in range(n_b):
        n_docs = int(rng.choice([5, 64, 100, 130, 300, 664, 1024, 2100, 4000, 8300]))
        S = int(rng.integers(80, 2500))
        plant = [(int(rng.integers(0, nq)), int(rng.integers(0, n_docs)), float(rng.choice([1.0, 0.9, 0.75, 0.7, 0.69, 0.4])))
                 for _ in range(int(rng.integers(0, 80)))]
        index, fasta, _ = build_case(oracle, rng, n_docs, S, queries, density=float(rng.choice([0.02, 0.25, 0.5])), plant=plant)
        name = f"g{int(rng.integers(0, 99)):02d}_s{b}__01"
        batches.append(name)
        indexes[name] = index
    batches.sort()
    n = int(rng.choice([1, 3, 100]))
    thr = float(rng.choice([0.3, 0.7, 0.7, 1.0]))
    src = MS.ResidentSource({b: pm.Index.load_mem(indexes[b], layout=int(rng.integers(0, 3))) for b in batches})
    mode = int(rng.integers(0, 3))
    parser = ThreadPoolExecutor(max_workers=1)
    if mode == 0:
        qarg = pm.Queries(fasta)
    elif mode == 1:
        qarg = [pm.Queries(p) for p in MS.split_prepared_fasta(fasta, int(rng.integers(1, max(2, nq))))]
    else:
        qarg = [parser.submit(pm.Queries, p) for p in MS.split_prepared_fasta(fasta, 0, int(rng.integers(64, 4000)))]
    out_dir = tmp_path / "03_match"
    report, merge = MS.run_stage(pm, batches, list(range(n_b)), src, qarg, "Q", str(out_dir), thr, n, want_merge=True,
                                 max_group=int(rng.choice([0, 0, 1, 2, 3])))
    parser.shutdown()
    assert report["queries"] == nq and sorted(report["merge_order"]) == sorted(batches)
    files = []
    for b in batches:
        exp = P.filter_text(oracle.query_file(indexes[b], fasta, thr).decode(), n)
        fn = out_dir / f"{b}____Q.gz"
        assert gzip.open(fn, "rt").read() == exp, (b, mode)
        files.append(str(fn))
    (tmp_path / "Q.fa").write_bytes(fasta)
    want = io.StringIO()
    F.filter_files(str(tmp_path / "Q.fa"), files, n, want)
    # one merge over all pieces: a read name that repeats -- inside a piece or across pieces -- is one query, as in the
    # consumer's dict
    got = merge.emit_to(str(tmp_path / "F.fa"))
    assert (tmp_path / "F.fa").read_text() == want.getvalue() and got == len(want.getvalue())
    assert not list(out_dir.glob("*.tmp"))

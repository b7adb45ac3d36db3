"""BASELINE.json full-size configurations on the GPU: exact checks on sampled
queries against the oracle evaluated on the *virtual* synthetic matrix (any row
of a 213 GB matrix can be regenerated on the host), plus size-independent
properties over the whole output (exact hit totals, idempotence, sharding and
layout invariance)."""
import numpy as np
import pytest

from phylign_amd import workload as W

pytestmark = pytest.mark.gpu
SEED = 661


def _overlay(plan_rows, plan_docs):
    ov = {}
    for r, d in zip(plan_rows.tolist(), plan_docs.tolist()):
        ov.setdefault(r, []).append(d)
    return ov


def _expected_hits(oracle, shape, seqs, qids, overlay, thr):
    h = oracle.Header()
    h.term_size, h.canonicalize, h.num_hashes = 31, 1, 1
    h.n_docs, h.signature_size, h.row_bytes = shape.n_docs, shape.signature_size, shape.row_bytes

    def row_fn(r):
        v = oracle.synth_row(SEED, shape.batch_id, r, shape.n_docs)
        for d in overlay.get(r, ()):
            v[d >> 3] |= np.uint8(1 << (d & 7))
        return v
    out = []
    for q in qids:
        s = seqs[q].tobytes()
        sc = oracle.scores_rows(row_fn, h, s)
        out += [(q, d, v) for d, v in oracle.select(sc, len(s) - 30, thr)]
    return out


def test_config2_one_batch_10k_queries(pm, oracle):
    """BASELINE configs[1]: bacillus_anthracis__01 shape (D=664, 83-byte rows, S~16.5M), 10k queries."""
    shape = W.select("config2")[0]
    assert (shape.n_docs, shape.row_bytes) == (664, 83) and shape.signature_size > 16_000_000
    ix = pm.Index.synth(shape.batch_id, shape.n_docs, shape.signature_size, seed=SEED)
    # (a) 10k x 150 bp, planted
    fasta, seqs = W.make_queries(10000, 150, seed=31)
    q = pm.Queries(fasta)
    hashes = q.hash_terms(1, 1)
    plan, sure = W.plant_plan(hashes, 10000, 120, [shape], every=20, docs_per_query=8)
    ix.plant(*plan[0])
    res = pm.search([ix], q, 0.7)
    hits = res.hits()
    assert len(hits) >= sure and res.stats.algorithmic_bytes == 10000 * 120 * 83
    ov = _overlay(*plan[0])
    sample = sorted(set(list(range(0, 10000, 20))[:60] + list(range(1, 400, 7))))
    exp = _expected_hits(oracle, shape, seqs, sample, ov, 0.7)
    sel = hits[np.isin(hits["query"], sample)]
    assert [(int(x["query"]), int(x["doc"]), int(x["score"])) for x in sel] == exp
    assert hits["score"].min() >= 84 and hits["score"].max() <= 120
    # idempotence: a second run gives the same ordered records
    assert np.array_equal(pm.search([ix], q, 0.7).hits(), hits)
    # (b) 10k x 31 bp: one k-mer per query, hit <=> bit set; exact total = sum of row popcounts
    fasta1, seqs1 = W.make_queries(10000, 31, seed=32)
    q1 = pm.Queries(fasta1)
    h1 = q1.hash_terms(1, 1)
    res1 = pm.search([ix], q1, 0.7)
    hits1 = res1.hits()
    total = 0
    for i in range(10000):
        row = oracle.synth_row(SEED, shape.batch_id, int(h1[i]) % shape.signature_size, shape.n_docs)
        for d in ov.get(int(h1[i]) % shape.signature_size, ()):
            row[d >> 3] |= np.uint8(1 << (d & 7))
        total += int(np.unpackbits(row).sum())
    assert len(hits1) == total and set(hits1["score"]) == {1}
    text = pm.format_hits(ix, q1, hits1[hits1["query"] < 3], slot=0)
    assert text.count(b"*") == 10000      # every query gets its header even with no records passed


def test_config3_full_size_sampled_parity_and_invariants(pm, oracle):
    """BASELINE configs[2]: 64 batches (~213 GB of signatures) resident on one GPU, 100k queries."""
    shapes = W.select("config3")
    fasta, seqs = W.make_queries(100000, 150, seed=31)
    q = pm.Queries(fasta)
    hashes = q.hash_terms(1, 1)
    # hashes at full size vs the oracle on a sample
    for qi in (0, 1, 49999, 99999):
        assert np.array_equal(hashes[qi * 120:(qi + 1) * 120], oracle.create_hashes(seqs[qi].tobytes(), 31, 1, 1))
    plan, sure = W.plant_plan(hashes, 100000, 120, shapes)
    ixs = []
    for pos, s in enumerate(shapes):
        ix = pm.Index.synth(s.batch_id, s.n_docs, s.signature_size, seed=SEED)
        if pos in plan:
            ix.plant(*plan[pos])
        ixs.append(ix)
    assert sum(i.info.device_bytes for i in ixs) > 213e9
    res = pm.search(ixs, q, 0.7)
    hits = res.hits()
    st = res.stats
    assert st.algorithmic_bytes == 12_000_000 * 16285 and len(hits) >= sure
    # sampled exact parity: per batch, the planted queries routed to it plus unplanted ones
    checked = 0
    for pos, s in enumerate(shapes):
        planted = [qq for n, qq in enumerate(range(0, 100000, 20)) if n % 64 == pos][:3]
        sample = sorted(set(planted + [pos * 13 + 1, 99999 - pos]))
        ov = _overlay(*plan[pos]) if pos in plan else {}
        exp = _expected_hits(oracle, s, seqs, sample, ov, 0.7)
        sel = hits[(hits["slot"] == pos) & np.isin(hits["query"], sample)]
        assert [(int(x["query"]), int(x["doc"]), int(x["score"])) for x in sel] == exp, s.batch
        checked += len(exp)
    assert checked >= 64 * 3 * 4
    # sharding invariance: two halves searched separately give the same records (slots shifted)
    a = pm.search(ixs[:20], q, 0.7, slot_base=0).hits()
    b = pm.search(ixs[20:], q, 0.7, slot_base=20).hits()
    assert np.array_equal(np.concatenate([a, b]), hits)
    for ix in ixs:
        ix.free()
    # layout invariance on one wide and one narrow batch: compact vs line-aligned rows
    for pos in (max(range(64), key=lambda i: shapes[i].row_bytes), min(range(64), key=lambda i: shapes[i].row_bytes)):
        s = shapes[pos]
        got = []
        for layout in (pm.PM_LAYOUT_COMPACT, pm.PM_LAYOUT_ALIGNED):
            ix = pm.Index.synth(s.batch_id, s.n_docs, s.signature_size, seed=SEED, layout=layout)
            if pos in plan:
                ix.plant(*plan[pos])
            got.append(pm.search([ix], q, 0.7, slot_base=pos).hits())
            ix.free()
        assert np.array_equal(got[0], got[1]) and np.array_equal(got[0], hits[hits["slot"] == pos])


def test_gene_length_queries_on_config3_full_size(pm, oracle):
    """SURVEY.md 8d's third query shape at BASELINE's index size: the 1 856 record lengths of data/ARGannot_r3.fa (10- and
    13-plane counter classes) against the 64 full-size batches of config 3 (~213 GB resident, signature sizes up to
    31.8 M rows); per batch the planted genes routed to it plus unplanted ones of both classes are checked exactly against
    the oracle on the virtual matrix; exact algorithmic bytes; both scan modes give the same records"""
    shapes = W.select("config3")
    lens = W.argannot_lengths()
    terms = [n - 30 for n in lens]
    fasta, seqs = W.make_queries_lengths(lens, seed=43, prefix="gene")
    arr = [np.frombuffer(s_, dtype=np.uint8) for s_ in seqs]
    q = pm.Queries(fasta)
    hashes = q.hash_terms(1, 1)
    assert len(hashes) == sum(terms) == 1594532
    off = np.concatenate([[0], np.cumsum(terms)])
    for qi in (0, 7, 1855):
        assert np.array_equal(hashes[off[qi]:off[qi + 1]], oracle.create_hashes(seqs[qi], 31, 1, 1))
    plan, sure = W.plant_plan_ragged(hashes, terms, shapes, every=16, docs_per_query=6)
    ixs = []
    for pos, s in enumerate(shapes):
        ix = pm.Index.synth(s.batch_id, s.n_docs, s.signature_size, seed=SEED)
        if pos in plan:
            ix.plant(*plan[pos])
        ixs.append(ix)
    try:
        res = pm.search(ixs, q, 0.7)
        hits = res.hits()
        assert res.stats.algorithmic_bytes == 1594532 * 16285 and len(hits) >= sure > 300
        assert {L["kernel"].split("P=")[1].split(",")[0] for L in res.launches()} == {"10", "13"}
        long_ones = [i for i, t in enumerate(terms) if t >= 1024]
        short_ones = [i for i, t in enumerate(terms) if t < 300]
        checked = 0
        for pos, s in enumerate(shapes):
            planted = [qq for n, qq in enumerate(range(0, len(lens), 16)) if n % 64 == pos][:2]
            sample = sorted(set(planted + [long_ones[pos % len(long_ones)], short_ones[(7 * pos) % len(short_ones)]]))
            ov = _overlay(*plan[pos]) if pos in plan else {}
            exp = _expected_hits(oracle, s, arr, sample, ov, 0.7)
            sel = hits[(hits["slot"] == pos) & np.isin(hits["query"], sample)]
            assert [(int(x["query"]), int(x["doc"]), int(x["score"])) for x in sel] == exp, s.batch
            checked += len(exp)
        assert checked >= 64 * 3
        pm.set_option("threshold_bound", 0)
        assert np.array_equal(pm.search(ixs, q, 0.7).hits(), hits)
    finally:
        pm.set_option("threshold_bound", 1)
        for ix in ixs:
            ix.free()


def test_config1_bundled_reads_on_batches_small_shapes(pm, oracle, tmp_path):
    """BASELINE configs[0] shape: the reference's 40 bundled reads (data/reads_{1..4}) against the three
    batches of data/batches_small.txt at their real shapes (195/176/664 documents, 6.5M-16.5M rows),
    synthetic signatures with the reads planted; every read x batch compared with the oracle, then the
    `make test` setting nb_best_hits=1 (Makefile:44) through the fused post-filter."""
    import os
    from phylign_amd import postprocess as P
    fasta = open(os.path.join(os.path.dirname(__file__), "golden", "reads", "reads_1___reads_2___reads_3___reads_4.fa"), "rb").read()
    recs = fasta.decode().split("\n")
    names, seqs = [r[1:] for r in recs[0::2] if r], [r.encode() for r in recs[1::2] if r]
    assert len(names) == 40 and names[0] == "1A" and names[-1] == "4J"
    shapes = W.select("small")
    assert [(s.n_docs, s.row_bytes) for s in shapes] == [(195, 25), (176, 22), (664, 83)]
    q = pm.Queries(fasta)
    rng = np.random.default_rng(12)
    for pos, s in enumerate(shapes):
        ix = pm.Index.synth(s.batch_id, s.n_docs, s.signature_size, seed=SEED)
        rows, docs = [], []
        for qi in range(pos, 40, 3):                     # each read is "from" one of the three species
            hs = oracle.create_hashes(seqs[qi], 31, 1, 1)
            for d, frac in zip(rng.choice(s.n_docs, size=12, replace=False), [1.0] * 5 + [0.9, 0.9, 0.8, 0.75, 0.7, 0.69, 0.5]):
                m = int(np.ceil(frac * len(hs)))
                rows += [int(h) % s.signature_size for h in hs[:m]]
                docs += [int(d)] * m
        ix.plant(rows, docs)
        ov = _overlay(np.array(rows), np.array(docs))
        hits = pm.search([ix], q, 0.7, slot_base=pos).hits()

        class _S:                                        # rows of varying length: per-read sequences
            def __getitem__(self, i):
                return np.frombuffer(seqs[i], dtype=np.uint8)
        exp = _expected_hits(oracle, s, _S(), list(range(40)), ov, 0.7)
        assert [(int(x["query"]), int(x["doc"]), int(x["score"])) for x in hits] == exp
        assert len(exp) >= 13 * 9
        # text with nb_best_hits = 1, built from the oracle's expectation through the post-filter mirror
        name_of = [ix.doc_name(d) for d in range(s.n_docs)]
        lines = []
        for qi in range(40):
            mine = [(d, v) for (qq, d, v) in exp if qq == qi]
            lines.append(f"*{names[qi]}\t{len(mine)}\n" + "".join(f"{name_of[d]}\t{v}\n" for d, v in mine))
        want = P.filter_text("".join(lines), 1)
        got = pm.search([ix], q, 0.7, slot_base=pos, nb_best_hits=1)
        assert pm.format_hits(ix, q, got.hits(), slot=pos, nb_best_hits=1).decode() == want
        ix.free()


def test_config5_one_million_queries_on_one_rank_shard(pm, oracle):
    """BASELINE configs[4] asks for 1 M queries: the query count at full size against the shard one
    rank of an 8-way split of config 3 holds (~27 GB), pipelined like the stage drives it.  Checks:
    exact parity on sampled queries against the oracle on the virtual matrix, planted totals, and
    that a query's records do not depend on the query set it travels in (first 50 k alone)."""
    shapes = W.select("config3")
    mine = W.assign_batches(shapes, 8)[3]
    nq = 1_000_000
    fasta, seqs = W.make_queries(nq, 150, seed=5)
    q = pm.Queries(fasta)
    assert q.count() == (nq, nq * 120)
    hashes = q.hash_terms(1, 1)
    sub = [shapes[p] for p in mine]
    plan, sure = W.plant_plan(hashes, nq, 120, sub, every=5000)
    del hashes
    ixs = []
    for i, s in enumerate(sub):
        ix = pm.Index.synth(s.batch_id, s.n_docs, s.signature_size, seed=SEED)
        if i in plan:
            ix.plant(*plan[i])
        ixs.append(ix)
    a = pm.search_async(ixs, q, 0.7, nb_best_hits=100)
    b = pm.search_async(ixs, q, 0.7, nb_best_hits=100)           # two searches of a million queries in flight
    hits = a.hits()
    st = a.stats
    assert st.algorithmic_bytes == nq * 120 * sum(s.row_bytes for s in sub)
    real = hits[hits["doc"] != pm.PM_DOC_COUNT]
    assert len(real) >= sure
    assert np.array_equal(b.hits(), hits)
    checked = 0
    for i, s in enumerate(sub):
        planted = [qq for n, qq in enumerate(range(0, nq, 5000)) if n % len(sub) == i][:2]
        sample = sorted(set(planted + [i * 7919 + 3, nq - 1 - i]))
        ov = _overlay(*plan[i]) if i in plan else {}
        exp = _expected_hits(oracle, s, seqs, sample, ov, 0.7)
        sel = real[(real["slot"] == i) & np.isin(real["query"], sample)]
        assert [(int(x["query"]), int(x["doc"]), int(x["score"])) for x in sel] == exp, s.batch
        checked += len(exp)
    assert checked >= len(sub) * 2 * 4
    # the same first 50 000 queries as a query set of their own
    cut = fasta.index(b">q0050000\n")
    q50 = pm.Queries(fasta[:cut])
    assert q50.count()[0] == 50000
    small = pm.search(ixs, q50, 0.7, nb_best_hits=100).hits()
    assert np.array_equal(small, hits[hits["query"] < 50000])
    # text of one batch at this size: every query gets its header line
    text = pm.format_hits(ixs[0], q, hits, slot=0, nb_best_hits=100)
    assert text.count(b"*q") == nq


def _query_blocks(text):
    """{query name: [lines]} of a cobs / post-filtered text"""
    out, cur = {}, None
    for line in text.split("\n"):
        if line.startswith("*"):
            cur = out.setdefault(line[1:].split("\t")[0], [line])
        elif line and cur is not None:
            cur.append(line)
    return out


def test_configs4_5_one_rank_shard_of_the_full_collection(pm, oracle, tmp_path):
    """BASELINE configs[3] and [4] on their own workload: the shard ONE rank holds when all 305 batches of
    batches_full.txt are split over 8 GPUs (38 batches, ~135 GB of signatures, every row-width class of the
    collection), driven the way the stage drives it.

    configs[3] (100 k queries): sampled exact parity against the oracle on the virtual matrix for every
    batch of the shard, exact algorithmic bytes, planted totals, sharding invariance (two halves = whole).
    configs[4] (1 M queries, 03_match -> 04_filter): match_stage.run_stage over the resident shard writes
    the 38 `.gz` files and the 04_filter FASTA; sampled queries are compared line by line with what the
    oracle + the golden-pinned post-filter / filter rules give; every query has its header in every file."""
    import gzip
    from phylign_amd import match_stage as MS
    from phylign_amd import postprocess as P
    full = W.select("full")
    assert len(full) == 305 and sum(s.row_bytes for s in full) == 82741          # bytes per k-mer
    parts = W.assign_batches(full, 8)
    assert sorted(i for p in parts for i in p) == list(range(305))
    mine = parts[2]
    sub = [full[p] for p in mine]
    assert len(sub) == 38 and 120e9 < sum(s.index_bytes for s in sub) < 145e9
    assert {(s.row_bytes + 127) // 128 for s in sub} >= {1, 2, 4}                 # narrow, medium and 4-line rows

    # ---- configs[3]: 100 k queries against the resident shard
    nq = 100_000
    fasta, seqs = W.make_queries(nq, 150, seed=31)
    q = pm.Queries(fasta)
    hashes = q.hash_terms(1, 1)
    plan, sure = W.plant_plan(hashes, nq, 120, sub, every=50)
    ixs = []
    for i, s in enumerate(sub):
        ix = pm.Index.synth(s.batch_id, s.n_docs, s.signature_size, seed=SEED)
        if i in plan:
            ix.plant(*plan[i])
        ixs.append(ix)
    assert sum(i.info.device_bytes for i in ixs) > 130e9
    res = pm.search(ixs, q, 0.7, nb_best_hits=100)
    hits = res.hits()
    st = res.stats
    assert st.algorithmic_bytes == nq * 120 * sum(s.row_bytes for s in sub)
    # one scan launch per row-width class, not per batch: 4-line, 3-line (if any), 2-line, and the mixed narrow launch
    assert st.n_scan_launches <= 4 < len(sub)
    real = hits[hits["doc"] != pm.PM_DOC_COUNT]
    assert len(real) >= sure
    checked = 0
    for i, s in enumerate(sub):
        planted = [qq for n, qq in enumerate(range(0, nq, 50)) if n % len(sub) == i][:2]
        sample = sorted(set(planted + [i * 1009 + 7, nq - 1 - i]))
        ov = _overlay(*plan[i]) if i in plan else {}
        exp = _expected_hits(oracle, s, seqs, sample, ov, 0.7)
        sel = real[(real["slot"] == i) & np.isin(real["query"], sample)]
        assert [(int(x["query"]), int(x["doc"]), int(x["score"])) for x in sel] == exp, s.batch
        checked += len(exp)
    assert checked >= len(sub) * 2 * 4
    a = pm.search(ixs[:17], q, 0.7, nb_best_hits=100, slot_base=0).hits()
    b = pm.search(ixs[17:], q, 0.7, nb_best_hits=100, slot_base=17).hits()
    assert np.array_equal(np.concatenate([a, b]), hits)
    res.free()
    q.free()

    # ---- configs[4]: 1 M queries through the stage (03_match files + 04_filter FASTA)
    nq = 1_000_000
    fasta, seqs = W.make_queries(nq, 150, seed=5)
    q = pm.Queries(fasta)
    hashes = q.hash_terms(1, 1)
    plan2, sure2 = W.plant_plan(hashes, nq, 120, sub, every=2500, docs_per_query=12)
    del hashes
    for i, ix in enumerate(ixs):
        if i in plan2:
            ix.plant(*plan2[i])
    names = sorted(s.batch for s in sub)
    by_name = {s.batch: (i, s) for i, s in enumerate(sub)}
    src = MS.ResidentSource({s.batch: ixs[i] for i, s in enumerate(sub)})
    report, merge = MS.run_stage(pm, names, list(range(len(names))), src, q, "Q", str(tmp_path / "03_match"), 0.7, 100,
                                 want_merge=True)
    # 38 resident batches: four pipelined quarters, each a fused search (one launch per row-width class)
    assert report["groups"] == 4 and report["scan_launches"] <= 16 and report["queries"] == nq
    out_fa = merge.emit().decode()
    # sampled queries: planted ones (hits in one batch) and unplanted ones
    planted_q = list(range(0, nq, 2500))[:6]
    sample = planted_q + [1, 499_999, nq - 1]
    expect = {}                                   # query -> {batch: [(doc, score)...]} after threshold, in cobs order
    for b in names:
        i, s = by_name[b]
        ov = {}
        for pl in (plan, plan2):
            if i in pl:
                for r, d in zip(pl[i][0].tolist(), pl[i][1].tolist()):
                    ov.setdefault(r, []).append(d)
        for (qq, d, v) in _expected_hits(oracle, s, seqs, sample, ov, 0.7):
            expect.setdefault(qq, {}).setdefault(b, []).append((d, v))
    assert sum(len(v) for v in expect.values()) >= 6            # the planted queries do hit
    name_of = {b: [ixs[by_name[b][0]].doc_name(d) for d in range(by_name[b][1].n_docs)] for b in names}
    for b in names[:3] + names[-2:] + sorted({bb for v in expect.values() for bb in v})[:4]:
        text = gzip.open(tmp_path / "03_match" / f"{b}____Q.gz", "rt").read()
        assert text.count("*q") == nq                           # every query has its header line in every file
        blocks = _query_blocks(text)
        for qq in sample:
            mine_ = expect.get(qq, {}).get(b, [])
            cobs = f"*q{qq:07d}\t{len(mine_)}\n" + "".join(f"{name_of[b][d]}\t{v}\n" for d, v in mine_)
            assert "\n".join(blocks[f"q{qq:07d}"]) + "\n" == P.filter_text(cobs, 100), (b, qq)
    # 04_filter: the 100 best (+ ties) over the shard's batches, ordered (-kmers, batch, ref)
    fa_lines = out_fa.split("\n")
    assert len(fa_lines) == 2 * nq + 1
    for qq in sample:
        items = []
        for b, lst in expect.get(qq, {}).items():
            kept = P.filter_text(f"*x\t{len(lst)}\n" + "".join(f"{name_of[b][d]}\t{v}\n" for d, v in lst), 100).split("\n")[1:-1]
            items += [(-int(l.split("\t")[1]), b, l.split("\t")[0][1:]) for l in kept]
        items.sort()
        if len(items) > 100:
            cut = 100
            while cut < len(items) and items[cut][0] == items[99][0]:
                cut += 1
            items = items[:cut]
        assert fa_lines[2 * qq] == f">q{qq:07d} " + ",".join(r for _, _, r in items)
        assert fa_lines[2 * qq + 1] == seqs[qq].tobytes().decode()
    for ix in ixs:
        ix.free()

"""Genome-like batches: indexes CONSTRUCTED from sequences the way COBS builds a classic index
(every canonical 31-mer of a strain sets its bit, signature_size = 2.80 x the largest strain),
strains correlated through a mutation tree, reads sampled from strains with errors.  Checks the
semantics end to end (a read finds its strain and its relatives, not the other species), text
parity with the oracle, and that the threshold bound changes nothing on correlated documents."""
import numpy as np
import pytest

from helpers import build_species_batch, sample_reads

pytestmark = pytest.mark.gpu


def test_reads_find_their_species_batch(pm, oracle):
    from phylign_amd import postprocess as P
    rng = np.random.default_rng(2026)
    batches = [build_species_batch(oracle, rng, n_docs, glen) for n_docs, glen in ((300, 20000), (90, 30000), (1100, 12000))]
    reads, home = [], []
    for b, (_, strains, _) in enumerate(batches):
        for d, s in sample_reads(rng, strains, 40, error=0.0):      # error-free reads: the source strain holds all 120 k-mers
            reads.append(s); home.append((b, d))
    fasta = "".join(f">read{i} from_b{home[i][0]}\n{s}\n" for i, s in enumerate(reads)).encode()
    ixs = [pm.Index.load_mem(b[0]) for b in batches]
    q = pm.Queries(fasta)
    pm.set_option("count_fetched", 1)
    try:
        out = {}
        for bound in (1, 0):
            pm.set_option("threshold_bound", bound)
            res = pm.search(ixs, q, 0.7)
            out[bound] = (res.hits(), [(L["fetched_bytes"], L["algorithmic_bytes"]) for L in res.launches()])
    finally:
        pm.set_option("count_fetched", 0)
        pm.set_option("threshold_bound", 1)
    hits = out[1][0]
    assert np.array_equal(hits, out[0][0])
    assert sum(f for f, _ in out[1][1]) < sum(a for _, a in out[1][1])      # foreign species are pruned ...
    # text parity per batch, plain and post-filtered
    for s, (index, _, _) in enumerate(batches):
        exp = oracle.query_file(index, fasta, 0.7)
        assert pm.format_hits(ixs[s], q, hits, slot=s) == exp
        assert pm.query_text(ixs[s], fasta, 0.7, nb_best_hits=10).decode() == P.filter_text(exp.decode(), 10)
    # semantics: the source strain is reported with a high score, in the home batch only
    with_relatives = 0
    for i, (b, d) in enumerate(home):
        mine = hits[hits["query"] == i]
        assert set(mine["slot"]) == {b}, (i, set(mine["slot"]))
        row = mine[mine["doc"] == d]
        assert len(row) == 1 and row["score"][0] == 120
        with_relatives += len(mine) >= 2
    assert with_relatives > len(home) // 2                       # relatives match too (correlated documents)

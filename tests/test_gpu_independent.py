"""The HIP path against the independent pure-Python restatement of tests/independent.py (python-xxhash, own header
writer, canonicalisation, scorer, text): where the C oracle and the product might share a misreading copied from one to
the other, this one was written separately from SURVEY.md appendix A.  Where all three agree, the only thing left
unpinned is what SURVEY.md itself recalls of upstream COBS."""
import pytest

pytest.importorskip("xxhash")
import independent as I  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", I.CASES)
@pytest.mark.parametrize("threshold", [0.7, 0.0, 1.0, 0.35])
def test_product_equals_an_independent_python_restatement(pm, case, threshold):
    seed, D, glen, k, num_hashes, canon = case
    index, m, fasta, names, S, records = I.built_case(*case)
    want = I.query_text(records, names, m, k, num_hashes, S, threshold, canon)
    ix = pm.Index.load_mem(index)
    info = ix.info
    assert (info.term_size, info.n_docs, info.signature_size, info.num_hashes) == (k, D, S, num_hashes)
    for bound in (1, 0):
        pm.set_option("threshold_bound", bound)
        assert pm.query_text(ix, fasta, threshold) == want, (seed, threshold, bound)
    pm.set_option("threshold_bound", 1)
    ix.free()


@pytest.mark.parametrize("case", I.COMPACT_CASES)
@pytest.mark.parametrize("threshold", [0.7, 0.0, 0.4])
def test_product_equals_the_independent_restatement_on_compact_indexes(pm, case, threshold):
    seed, page, D, params, k = case
    index, mats, fasta, names, records = I.built_compact_case(seed, page, D, tuple(params), k)
    want = I.query_text_compact(records, names, mats, k, page, params, threshold)
    for layout in (1, 2):
        ix = pm.Index.load_mem(index, layout=layout)
        assert (ix.info.n_parts, ix.info.page_size, ix.info.n_docs) == (len(params), page, D)
        for bound in (1, 0):
            pm.set_option("threshold_bound", bound)
            assert pm.query_text(ix, fasta, threshold) == want, (seed, threshold, layout, bound)
        pm.set_option("threshold_bound", 1)
        ix.free()

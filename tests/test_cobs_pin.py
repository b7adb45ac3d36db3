"""The pin at the COBS boundary: fixtures captured from a real `cobs` 0.2.1 with the reference's exact
argv (tools/pin_against_cobs.sh -> tests/golden/cobs/) against the oracle (CPU) and the HIP product (GPU).  Skipped while
no fixture exists -- the build container has no cobs binary -- and then the oracle's COBS rules are "parity unpinned"."""
import itertools
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden", "cobs")
HAVE = os.path.exists(os.path.join(G, "cobs_stdout.txt")) and os.path.exists(os.path.join(G, "index.cobs_classic"))
need_fixture = pytest.mark.skipif(not HAVE, reason="no fixture from a real cobs binary: run tools/pin_against_cobs.sh where cobs=0.2.1 is installed")
RULES = list(itertools.product((0, 1, 2), (0, 1)))          # (cobs_threshold_rule, cobs_tie_order)


def _read(name):
    with open(os.path.join(G, name), "rb") as f:
        return f.read()


def test_the_pin_recipe_is_in_place():
    """the one command that turns "unpinned" into "pinned" exists, is executable and names the reference's argv"""
    sh = os.path.join(ROOT, "tools", "pin_against_cobs.sh")
    text = open(sh).read()
    assert os.access(sh, os.X_OK)
    for needle in ("cobs classic-construct", "cobs query --load-complete -t 0.7", "--index-sizes", "0\\.2\\.1", "tests/test_cobs_pin.py"):
        assert needle in text, needle
    import subprocess
    import sys
    import tempfile
    with tempfile.TemporaryDirectory() as d:                 # the input generator runs anywhere (numpy only)
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_pin_inputs.py"), d], check=True)
        assert len(os.listdir(os.path.join(d, "genomes"))) == 24
        q = open(os.path.join(d, "queries.fa")).read().split("\n")
        assert len(q) == 801 and {len(s) for s in q[1::2]} >= {151, 150, 31}


@need_fixture
def test_oracle_reads_the_real_header_and_reproduces_cobs_text(oracle):
    """(i) the header of a file written by `cobs classic-construct` parses (which of the two field orders it is gets
    recorded), (ii) under exactly one setting of the switchable rules the oracle's text is cobs' stdout, and that setting is
    the default"""
    index, fasta = _read("index.cobs_classic"), _read("queries.fa")
    h = oracle.header_parse(index)
    assert h.term_size == 31 and h.num_hashes == 1 and h.n_docs == 24
    want = _read("cobs_stdout.txt")
    assert _read("cobs_stdout_stream.txt") == want           # -T and the streaming form do not change the result
    matching = []
    try:
        for thr_rule, tie in RULES:
            oracle.set_rules(thr_rule, tie)
            if oracle.query_file(index, fasta, 0.7) == want and \
                    oracle.query_file(index, _read("queries_few.fa"), 0.0) == _read("cobs_stdout_t0.txt"):
                matching.append((thr_rule, tie))
    finally:
        oracle.set_rules(0, 0)
    assert matching, "no rule setting reproduces cobs: a difference beyond threshold rounding and tie order (diff the texts)"
    assert (0, 0) in matching, f"cobs 0.2.1 follows rules {matching}, the defaults are (0, 0): change them in pm_runtime.cpp and cobs_oracle.c"


@need_fixture
@pytest.mark.gpu
def test_product_reproduces_cobs_text(pm):
    index, fasta = _read("index.cobs_classic"), _read("queries.fa")
    ix = pm.Index.load_mem(index)
    assert ix.info.n_docs == 24 and ix.info.term_size == 31
    assert pm.query_text(ix, fasta, 0.7) == _read("cobs_stdout.txt")
    assert pm.query_text(ix, _read("queries_few.fa"), 0.0) == _read("cobs_stdout_t0.txt")
    for bound in (0, 1):
        pm.set_option("threshold_bound", bound)
        assert pm.query_text(ix, fasta, 0.7) == _read("cobs_stdout.txt")
    pm.set_option("threshold_bound", 1)


@need_fixture
def test_edge_inputs_behave_like_cobs(oracle):
    """a read shorter than k / a read with an N: whatever cobs did (status and stdout) is what the oracle does"""
    index = _read("index.cobs_classic")
    for e in ("short", "with_n"):
        status = int(_read(f"edge_{e}.status.txt"))
        try:
            got, ok = oracle.query_file(index, _read(f"edge_{e}.fa"), 0.7), True
        except Exception:
            got, ok = b"", False
        assert ok == (status == 0), (e, status)
        if status == 0:
            assert got == _read(f"edge_{e}.stdout.txt"), e

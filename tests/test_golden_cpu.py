"""CPU suite (no GPU): the oracle against golden vectors, the host-side mirrors
of the reference scripts against fixtures captured from those scripts, and the
C-ABI library's exported symbols."""
import glob
import io
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


# ------------------------------------------------------------------ oracle
def test_oracle_xxh64_known_answers(oracle):
    n = 0
    with open(os.path.join(GOLD, "xxh64_kat.tsv")) as f:
        for line in f:
            if line.startswith("#"):
                continue
            hx, seed, digest = line.rstrip("\n").split("\t")
            assert oracle.xxh64(bytes.fromhex(hx), int(seed)) == int(digest)
            n += 1
    assert n >= 2400 + 71 * 4
    # SURVEY.md appendix A.3 vectors
    assert oracle.xxh64(b"ACGTACGTACGTACGTACGTACGTACGTACG", 0) == 3318676550491556742
    assert oracle.xxh64(b"ACGTACGTACGTACGTACGTACGTACGTACG", 1) == 1345981084757705630


def test_oracle_canonicalize(oracle):
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    rng = np.random.default_rng(0)
    for k in (1, 2, 5, 30, 31, 32):
        for _ in range(200):
            s = "".join("ACGT"[i] for i in rng.integers(0, 4, size=k))
            rc = "".join(comp[c] for c in reversed(s))
            assert oracle.canonicalize(s.encode()) == min(s, rc).encode()
    assert oracle.canonicalize(b"ACGN") is None and oracle.canonicalize(b"acgt") is None


def test_oracle_threshold_rule(oracle):
    # ceil(t * n) in IEEE double: SURVEY.md section 7 hard part (a)
    assert oracle.threshold(0.7, 120) == 84
    assert oracle.threshold(0.7, 121) == 85
    assert oracle.threshold(0.7, 1) == 1
    assert oracle.threshold(0.5, 3) == 2
    assert oracle.threshold(1.0, 120) == 120
    assert oracle.threshold(0.0, 120) == 0
    import math
    for n in range(1, 400):
        for t in (0.1, 0.35, 0.7, 0.8, 0.95):
            assert oracle.threshold(t, n) == math.ceil(t * n)


def test_oracle_header_roundtrip_and_both_layouts(oracle):
    names = ["abc_S1", "x_S2", "yy_S3", "zzzzzzzz_S4", "q_S5"]
    m = np.arange(7 * 1, dtype=np.uint8).reshape(7, 1)
    idx = oracle.make_index(31, 1, 7, 2, names, m)
    h = oracle.header_parse(idx)
    assert (h.term_size, h.canonicalize, h.signature_size, h.num_hashes, h.n_docs, h.row_bytes, h.layout) == (31, 1, 7, 2, 5, 1, 0)
    assert bytes(idx[h.data_off:]) == m.tobytes()
    assert bytes(idx[:18]) == b"COBS:CLASSIC_INDEX"
    # SURVEY appendix A.1 field order (sig, hashes, n_docs): re-pack and parse again
    b = bytearray(idx)
    o = 18 + 4 + 4 + 1
    n_docs, sig, nh = b[o:o + 4], b[o + 4:o + 12], b[o + 12:o + 20]
    b[o:o + 20] = sig + nh + n_docs
    h2 = oracle.header_parse(bytes(b))
    assert (h2.signature_size, h2.num_hashes, h2.n_docs, h2.layout) == (7, 2, 5, 1)
    with pytest.raises(ValueError):
        oracle.header_parse(b"COBS:COMPACT_INDEX" + bytes(64))


def test_oracle_scores_bruteforce_python(oracle):
    """C restatement vs a pure-Python loop on a tiny case (independent of the C scoring code)."""
    rng = np.random.default_rng(3)
    n_docs, S, k, nh = 19, 53, 5, 2
    bits = rng.random((S, 24)) < 0.4
    bits[:, n_docs:] = False
    matrix = np.packbits(bits, axis=1, bitorder="little")
    idx = oracle.make_index(k, 1, S, nh, [f"p_d{i}" for i in range(n_docs)], matrix)
    seq = "".join("ACGT"[i] for i in rng.integers(0, 4, size=40))
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    exp = [0] * n_docs
    for i in range(len(seq) - k + 1):
        km = seq[i:i + k]
        can = min(km, "".join(comp[c] for c in reversed(km))).encode()
        rows = [oracle.xxh64(can, j) % S for j in range(nh)]
        for d in range(n_docs):
            exp[d] += all(bits[r, d] for r in rows)
    assert list(oracle.scores(idx, seq.encode())) == exp
    sel = oracle.select(np.array(exp, dtype=np.uint32), len(seq) - k + 1, 0.3)
    assert sel == sorted([(d, s) for d, s in enumerate(exp) if s >= oracle.threshold(0.3, len(seq) - k + 1)],
                         key=lambda t: (-t[1], t[0]))
    text = oracle.query_file(idx, f">q1 c\n{seq}\n".encode(), 0.3).decode().split("\n")
    assert text[0] == f"*q1 c\t{len(sel)}"
    assert text[1:-1] == [f"p_d{d}\t{s}" for d, s in sel]


def test_oracle_output_feeds_reference_grammar(oracle):
    """the oracle's text obeys the grammar the reference's consumers parse
    (scripts/postprocess_cobs.py:23-39, scripts/filter_queries.py:51-65)."""
    from helpers import build_case, rand_seq
    rng = np.random.default_rng(4)
    qs = [(f"r{i} comment", rand_seq(rng, 100)) for i in range(6)]
    idx, fasta, _ = build_case(oracle, rng, 50, 400, qs, plant=[(i, i, 1.0) for i in range(6)])
    out = oracle.query_file(idx, fasta, 0.7).decode()
    n_left = 0
    for line in out.splitlines():
        if line.startswith("*"):
            assert n_left == 0
            n_left = int(line[1:].split("\t")[1])
        else:
            name, score = line.split()
            assert len(name.split("_")) == 2 and int(score) >= 49
            n_left -= 1
    assert n_left == 0


def test_synth_spec_is_stable(oracle):
    """pins the build's synthetic-matrix generator (both the oracle copy and the device kernel follow it)"""
    assert oracle.splitmix64(0) == 0xE220A8397B1DCDAF
    r = oracle.synth_row(661, 3, 12345, 664)
    assert r.shape == (83,) and np.array_equal(r[:2], oracle.synth_row(661, 3, 12345, 16))
    assert oracle.synth_row(661, 3, 5, 13)[1] < 32      # bits >= n_docs are zero
    m = oracle.synth_fill(661, 9, 300, 1000, 3)
    assert np.array_equal(m[299], oracle.synth_row(661, 9, 299, 1000))
    dens = np.unpackbits(m, axis=1, bitorder="little")[:, :1000].mean()
    assert 0.23 < dens < 0.27


# ------------------------------------------------- postprocess mirror (a8)
def _post_cases():
    return sorted(glob.glob(os.path.join(GOLD, "postprocess", "*.in")))


@pytest.mark.parametrize("inp", _post_cases(), ids=lambda p: os.path.basename(p)[:-3])
def test_postprocess_matches_reference_fixtures(inp):
    from phylign_amd import postprocess as P
    text = open(inp).read()
    base = inp[:-3]
    seen = 0
    for n in (0, 1, 2, 3, 100):
        out = io.StringIO()
        ok = True
        try:
            P.filter_stream(io.StringIO(text), n, out)
        except (ValueError, P.PostprocessError):
            ok = False
        if os.path.exists(f"{base}.n{n}.out"):
            assert ok and out.getvalue() == open(f"{base}.n{n}.out").read()
        else:
            assert not ok and out.getvalue() == open(f"{base}.n{n}.fail").read()
        seen += 1
    assert seen == 5


def test_postprocess_script_exit_codes():
    script = os.path.join(ROOT, "scripts", "postprocess_cobs.py")
    text = open(os.path.join(GOLD, "postprocess", "survey_a4.in"), "rb").read()
    r = subprocess.run([sys.executable, script, "-n", "2"], input=text, capture_output=True)
    assert r.returncode == 0 and r.stdout == open(os.path.join(GOLD, "postprocess", "survey_a4.n2.out"), "rb").read()
    bad = open(os.path.join(GOLD, "postprocess", "headerless.in"), "rb").read()
    r = subprocess.run([sys.executable, script, "-n", "2"], input=bad, capture_output=True)
    assert r.returncode != 0


# ---------------------------------------------------- 04_filter mirror (f1)
@pytest.mark.parametrize("n", [1, 2, 5, 100])
def test_filter_queries_matches_reference_fixture(n):
    from phylign_amd import filter_queries as F
    d = os.path.join(GOLD, "filter")
    files = [os.path.join(d, f"{b}____q.gz") for b in ("aaa_bbb__01", "ccc_ddd__01", "ccc_ddd__02")]
    out = io.StringIO()
    F.filter_files(os.path.join(d, "queries.fa"), files, n, out)
    assert out.getvalue() == open(os.path.join(d, f"expected.n{n}.fa")).read()


# ----------------------------------------------------- sizing mirror (a10)
def test_sizing_helpers_match_snakefile_vectors(tmp_path):
    from phylign_amd import sizing as Z
    rows = [l.rstrip("\n").split("\t") for l in open(os.path.join(GOLD, "sizing.tsv")) if not l.startswith("#")]
    table = tmp_path / "sizes.txt"
    seen = {}
    for r in rows:
        seen[r[0]] = (r[5], int(r[6]))
    with open(table, "w") as f:
        f.write("cobs/zzz_other__01.cobs_classic.xz  123  456\n")
        for b, (sz, xz_mb) in seen.items():
            # xz RAM bytes chosen so that int(bytes/MiB)+1 == the recorded MB
            f.write(f"cobs/{b}.cobs_classic.xz  {sz}  {(xz_mb - 1) * 1024 * 1024 + 7}\n")
    for b, streaming, ct, cores, ram, size, xz_mb, ram_mb, threads in rows:
        assert Z.index_metadata(b, table)[0] == int(size)
        assert Z.xz_ram_mb(b, table) == int(xz_mb)
        assert Z.batch_ram_mb(b, table, False, bool(int(streaming))) == int(ram_mb)
        assert Z.cobs_threads(b, table, ct, int(cores), int(ram), bool(int(streaming))) == int(threads)
    assert len(rows) == 216


# ------------------------------------------------------------ C ABI surface
def test_c_abi_exports_every_declared_symbol():
    from phylign_amd import _lib
    L = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "phylign_match.h")).read()
    declared = set(re.findall(r"\b(pm_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 30
    bound = {s[0] for s in _lib.SYMBOLS}
    assert declared == bound, declared ^ bound
    for name in declared:
        assert getattr(L, name) is not None


def test_product_library_exports_only_the_drop_in_abi():
    """nm -D: libphylign_match.so exports exactly the entry points of include/phylign_match.h -- no measurement aid
    (synthetic indexes, planting, probes live in libphylign_bench.so behind include/phylign_match_bench.h), no C++
    internals, no kernel stubs; and the product library reads no PM_PROBE_* variable"""
    import subprocess
    from phylign_amd import _lib, bench_aids
    hdr = open(os.path.join(ROOT, "include", "phylign_match.h")).read()
    declared = set(re.findall(r"\b(pm_[a-z0-9_]+)\s*\(", hdr))
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    assert exported == declared, exported ^ declared
    assert not any(w in name for name in exported for w in ("synth", "plant", "probe", "bench"))
    assert b"PM_PROBE" not in open(_lib.LIB_PATH, "rb").read()
    bhdr = open(os.path.join(ROOT, "include", "phylign_match_bench.h")).read()
    bdecl = set(re.findall(r"\b(pm_bench_[a-z0-9_]+)\s*\(", bhdr))
    out = subprocess.run(["nm", "-D", "--defined-only", bench_aids.LIB_PATH], capture_output=True, text=True, check=True).stdout
    bexp = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    assert bexp == bdecl == {s_[0] for s_ in bench_aids.SYMBOLS}, bexp ^ bdecl
    L = bench_aids.load()
    for name in bdecl:
        assert getattr(L, name) is not None
    # the drop-in modules never import the aids
    for mod in ("cobs_query.py", "postprocess.py", "filter_queries.py", "server.py", "fix_query.py", "dist.py", "launch.py"):
        assert "bench_aids" not in open(os.path.join(ROOT, "phylign_amd", mod)).read(), mod


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from phylign_amd import _lib
    with pytest.raises(_lib.PMError) as e:
        _lib.init(0)
    assert e.value.code == -2 and "no CPU fallback" in str(e.value)
    q = _lib.Queries(b">q\nACGT\n", term_size=3)       # parsing is host work ...
    with pytest.raises(_lib.PMError) as e:
        q.hash_terms(1, 1)                              # ... every compute entry point needs the GPU
    assert e.value.code == -2
    with pytest.raises(_lib.PMError) as e:
        _lib.Index.synth(1, 10, 10)
    assert e.value.code == -2
    # the one host-only rule of the path is callable without a device and equals the oracle's
    assert _lib.threshold_terms(0.7, 121) == 85


def test_threshold_rule_product_equals_oracle(oracle):
    from phylign_amd import _lib
    _lib.load()
    for n in list(range(1, 300)) + [1000, 65535, 70000]:
        for t in (0.0, 0.1, 0.5, 0.7, 0.75, 0.9, 1.0):
            assert _lib.threshold_terms(t, n) == oracle.threshold(t, n)


def test_product_never_imports_oracle():
    """the product path must not route through the oracle (or any CPU fallback)"""
    for path in glob.glob(os.path.join(ROOT, "phylign_amd", "**", "*"), recursive=True) + \
            glob.glob(os.path.join(ROOT, "scripts", "*")) + glob.glob(os.path.join(ROOT, "tools", "*")):
        if os.path.isfile(path) and path.endswith((".py", ".cpp", ".hip", ".h", ".sh")):
            src = open(path).read()
            assert "oracle" not in src.replace("the oracle", "").replace("CPU oracle", "") or path.endswith("build.py"), path


# ------------------- native (C++) post-filter and 04_filter merge, host-only paths
def _structured(text):
    """COBS text -> (fasta bytes, names, hit records) with doc ids in order of first appearance"""
    from phylign_amd import _lib as pm
    names, recs, fasta, qi = {}, [], [], -1
    for line in text.splitlines():
        if line.startswith("*"):
            qi += 1
            fasta.append(">" + line[1:].rsplit("\t", 1)[0] + "\nACGTACGTACGTACGTACGTACGTACGTACGTA\n")
        else:
            name, score = line.rsplit("\t", 1)
            d = names.setdefault(name, len(names))
            recs.append((qi, d, int(score), 0))
    return "".join(fasta).encode(), list(names), np.array(recs, dtype=pm.HIT_DTYPE)


@pytest.mark.parametrize("case", ["survey_a4", "zero_hits", "ties_small", "random_mid", "ties_large", "underscores"])
def test_native_postfilter_matches_reference_fixtures(case):
    """pm_format_hits(nb_best_hits=n) against outputs of the reference's postprocess_cobs.py"""
    from phylign_amd import _lib as pm
    base = os.path.join(GOLD, "postprocess", case)
    text = open(base + ".in").read()
    fasta, names, hits = _structured(text)
    q = pm.Queries(fasta, term_size=31)
    ix = pm.Index.from_names(names)
    assert pm.format_hits(ix, q, hits, slot=0, nb_best_hits=-1).decode() == text      # plain cobs text round-trips
    for n in (0, 1, 2, 3, 100):
        if os.path.exists(f"{base}.n{n}.out"):
            assert pm.format_hits(ix, q, hits, slot=0, nb_best_hits=n).decode() == open(f"{base}.n{n}.out").read()
        else:
            with pytest.raises(pm.PMError):
                pm.format_hits(ix, q, hits, slot=0, nb_best_hits=n)


@pytest.mark.parametrize("keep", [1, 2, 5, 100])
def test_native_merge_matches_reference_filter_fixture(keep):
    """pm_merge_* against the output of the reference's filter_queries.py"""
    import gzip
    from phylign_amd import _lib as pm
    d = os.path.join(GOLD, "filter")
    q = pm.Queries(open(os.path.join(d, "queries.fa"), "rb").read(), term_size=31)
    order = {}
    for i, line in enumerate(l for l in open(os.path.join(d, "queries.fa")) if l.startswith(">")):
        order[line[1:].split()[0]] = i
    m = pm.Merge(q, keep)
    for b in ("aaa_bbb__01", "ccc_ddd__01", "ccc_ddd__02"):
        names, recs, qi = {}, [], None
        for line in gzip.open(os.path.join(d, f"{b}____q.gz"), "rt"):
            if line.startswith("*"):
                qi = order[line[1:].split("\t")[0].split(" ")[0]]
            else:
                name, score = line.split()
                recs.append((qi, names.setdefault(name, len(names)), int(score), 7))
        ix = pm.Index.from_names(list(names) or ["x_y"])
        m.add(b, ix, np.array(recs, dtype=pm.HIT_DTYPE), slot=7, nb_best_hits=-1)
    assert m.emit().decode() == open(os.path.join(d, f"expected.n{keep}.fa")).read()
    import tempfile
    with tempfile.TemporaryDirectory() as td:                    # the threaded, straight-to-file form writes the same bytes
        n = m.emit_to(os.path.join(td, "out.fa"))
        assert open(os.path.join(td, "out.fa"), "rb").read() == m.emit() and n == len(m.emit())
        assert os.listdir(td) == ["out.fa"]


def _filter_fixture_batches(pm):
    import gzip
    d = os.path.join(GOLD, "filter")
    order = {}
    for i, line in enumerate(l for l in open(os.path.join(d, "queries.fa")) if l.startswith(">")):
        order[line[1:].split()[0]] = i
    out = []
    for b in ("aaa_bbb__01", "ccc_ddd__01", "ccc_ddd__02"):
        names, recs, qi = {}, [], None
        for line in gzip.open(os.path.join(d, f"{b}____q.gz"), "rt"):
            if line.startswith("*"):
                qi = order[line[1:].split("\t")[0].split(" ")[0]]
            else:
                name, score = line.split()
                recs.append((qi, names.setdefault(name, len(names)), int(score), 0))
        out.append((b, list(names) or ["x_y"], np.array(recs, dtype=pm.HIT_DTYPE)))
    return out


@pytest.mark.parametrize("keep", [1, 2, 5, 100])
@pytest.mark.parametrize("split", [(0,), (0, 1), (2,), (1, 2)])
def test_native_merge_export_of_parts_equals_one_merge(keep, split):
    """pm_merge_export: two ranks merge their own batches, rank 0 adds both exports again -- the 04_filter
    FASTA equals the reference's (the multi-GPU form of match_stage), in any order of the parts."""
    from phylign_amd import _lib as pm
    d = os.path.join(GOLD, "filter")
    q = pm.Queries(open(os.path.join(d, "queries.fa"), "rb").read(), term_size=31)
    batches = _filter_fixture_batches(pm)
    parts = [[i for i in range(3) if i in split], [i for i in range(3) if i not in split]]
    root = pm.Merge(q, keep)
    for part in reversed(parts):                         # rank order must not matter
        m = pm.Merge(q, keep)
        for i in part:
            b, names, recs = batches[i]
            ix = pm.Index.from_names(names)
            m.add(b, ix, recs, slot=0, nb_best_hits=-1)
            ix.free()                                    # the merge keeps its own copy of the names
        ex = m.export()
        assert np.array_equal(ex, pm.sort_hits(ex.copy()))
        for k, i in enumerate(part):
            b, names, _ = batches[i]
            root.add(b, pm.Index.from_names(names), ex[ex["slot"] == k], slot=k, nb_best_hits=-1)
    assert root.emit().decode() == open(os.path.join(d, f"expected.n{keep}.fa")).read()


# ------------------------------------------------------ fix_query mirror (f4)
def test_fix_query_rules_and_awk_fixed_point(tmp_path):
    from phylign_amd import fix_query as FQ
    src = (b"@r1 some comment\nacgtNNryACGT\nACGT\n+\nII@IIIIIIIIIIIII\n"
           b">r2 x y\nacg\ntnn\n\n>r3\n@r4\nGATTACA\n+r4\n@@@@@@@\n")
    out = io.BytesIO()
    FQ.fix_stream(io.BytesIO(src), out)
    got = out.getvalue()
    assert got == b">r1\nACGTAAAAACGTACGT\n>r2\nACGTAA\n>r3\n\n>r4\nGATTACA\n"
    # the awk half of the rule (Snakefile:330-332) leaves the mirror's output unchanged
    p = tmp_path / "x.fa"
    p.write_bytes(got)
    r = subprocess.run(["awk", '{if(NR%2==1){print $0;}else{gsub(/[^ACGT]/, "A"); print;}}', str(p)], capture_output=True)
    assert r.returncode == 0 and r.stdout == got
    merged = open(os.path.join(GOLD, "reads", "reads_1___reads_2___reads_3___reads_4.fa"), "rb").read()
    assert merged.count(b">") == 40 and merged.startswith(b">1A\nTTTGAAATCC") and b">4J\n" in merged
    assert set(merged.replace(b">", b"").replace(b"\n", b"")) <= set(b"ACGT0123456789ABCDEFGHIJ")


def test_oracle_compact_index_bruteforce(oracle):
    """the oracle's compact-index restatement vs an independent pure-Python evaluation"""
    rng = np.random.default_rng(0)
    page, sigs, nh, nd, k = 16, [50, 70, 40], [1, 2, 1], 300, 5
    mats = []
    for p, s in enumerate(sigs):
        bits = rng.random((s, page * 8)) < 0.4
        if p * 128 + 128 > nd:
            bits[:, nd - p * 128:] = False
        mats.append(np.packbits(bits, axis=1, bitorder="little"))
    names = [f"a_D{i}" for i in range(nd)]
    idx = oracle.make_compact(k, 1, page, sigs, nh, names, mats)
    c = oracle.compact_parse(idx)
    assert (c.n_parts, c.n_docs, c.page_size) == (3, nd, page) and c.data_off % page == 0
    assert bytes(idx[:18]) == b"COBS:COMPACT_INDEX" and bytes(idx[c.data_off - 13:c.data_off]) == b"COMPACT_INDEX"
    seq = "ACGTTGCAACGTAGCTAGCTAGGATC"
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    exp = [0] * nd
    for i in range(len(seq) - k + 1):
        km = seq[i:i + k]
        can = min(km, "".join(comp[x] for x in reversed(km))).encode()
        for p, s in enumerate(sigs):
            acc = np.full(page, 255, dtype=np.uint8)
            for j in range(nh[p]):
                acc &= mats[p][oracle.xxh64(can, j) % s]
            b = np.unpackbits(acc, bitorder="little")
            for d in range(128):
                if p * 128 + d < nd:
                    exp[p * 128 + d] += int(b[d])
    lines = oracle.query_file(idx, f">q\n{seq}\n".encode(), 0.3).decode().split("\n")
    want = sorted([(d, s) for d, s in enumerate(exp) if s >= oracle.threshold(0.3, len(seq) - k + 1)], key=lambda t: (-t[1], t[0]))
    assert lines[0] == f"*q\t{len(want)}" and lines[1:-1] == [f"a_D{d}\t{s}" for d, s in want]


# -------------------------------- product header readers (host-only entry point)
def test_product_header_reader_classic_and_compact(oracle):
    """pm_index_load_header_mem parses headers without a GPU: both classic field orders,
    compact headers, and rejects malformed input with PM_EFORMAT (-5)."""
    from phylign_amd import _lib as pm
    names = [f"{i:05x}_SAM{i}" for i in range(37)]
    idx = oracle.make_index(31, 1, 1000, 2, names)
    ix = pm.Index.load_header_mem(idx)
    info = ix.info
    assert (info.term_size, info.canonicalize, info.signature_size, info.num_hashes, info.n_docs, info.row_bytes,
            info.header_layout, info.has_matrix, info.n_parts) == (31, 1, 1000, 2, 37, 5, 0, 0, 0)
    assert [ix.doc_name(d) for d in (0, 36)] == [names[0], names[36]]
    b = bytearray(idx)                       # SURVEY.md appendix A.1 order: sig, hashes, n_docs
    o = 18 + 9
    b[o:o + 20] = b[o + 4:o + 12] + b[o + 12:o + 20] + b[o:o + 4]
    ix2 = pm.Index.load_header_mem(bytes(b))
    assert (ix2.info.signature_size, ix2.info.num_hashes, ix2.info.n_docs, ix2.info.header_layout) == (1000, 2, 37, 1)
    cidx = oracle.make_compact(31, 1, 64, [100, 90], [1, 3], [f"x_{i}" for i in range(700)])
    cx = pm.Index.load_header_mem(cidx)
    assert (cx.info.n_parts, cx.info.page_size, cx.info.n_docs, cx.info.row_bytes) == (2, 64, 700, 64)
    assert cx.doc_name(699) == "x_699"
    for bad in (b"", b"COBS:", b"COBS:CLASSIC_INDEX" + bytes(40), b"COBS:XXXXXXX_INDEX" + bytes(200),
                bytes(idx[:200]), bytes(idx[:18]) + b"\x02" + bytes(idx[19:])):
        with pytest.raises(pm.PMError) as e:
            pm.Index.load_header_mem(bad)
        assert e.value.code == -5, bad[:30]
    with pytest.raises(pm.PMError) as e:
        pm.Index.load_mem(idx)               # the matrix needs the GPU
    assert e.value.code == -2


# ---------------- FASTA record rules: product parser vs oracle record loop (property test)
def test_fasta_record_rules_property(oracle):
    from hypothesis import given, settings, strategies as st
    from phylign_amd import _lib as pm
    k = 5
    names = ["a_X"]
    index = oracle.make_index(k, 1, 7, 1, names)          # all-zero matrix: nothing ever matches at t > 0
    ix = pm.Index.from_names(names, term_size=k)
    header = st.builds(lambda c, t: c + t, st.sampled_from(">;"), st.text(alphabet="abcXYZ 09_|\t*>;", max_size=12))
    seqline = st.text(alphabet="ACGT", min_size=0, max_size=14)
    line = st.one_of(header, seqline, st.just(""), seqline.map(lambda s: s + "N"))
    text = st.builds(lambda ls, nl: "\n".join(ls) + ("\n" if nl else ""), st.lists(line, max_size=12), st.booleans())

    @settings(max_examples=400, deadline=None)
    @given(text)
    def check(fa):
        data = fa.encode()
        try:
            exp = oracle.query_file(index, data, 0.7)
        except RuntimeError:
            exp = None
        try:
            q = pm.Queries(data, term_size=k)
            got = pm.format_hits(ix, q, np.zeros(0, dtype=pm.HIT_DTYPE), slot=0)
        except pm.PMError as e:
            assert e.code == -6
            got = None
        assert got == exp, fa
    check()


def test_native_postfilter_many_queries_threads_and_order_paths():
    """pm_format_hits on enough queries for its multi-threaded path, on ordered records (sliced in
    place) and on shuffled ones (copied and ordered), against the golden-pinned Python mirror"""
    from phylign_amd import _lib as pm
    from phylign_amd import postprocess as P
    rng = np.random.default_rng(4)
    nq, n_docs = 30000, 500
    names = [f"{rng.integers(0, 16**5):05x}_SAM{d:05d}" for d in range(n_docs)]
    fasta = "".join(f">q{i} c{i % 7}\nACGTACGTACGTACGTACGTACGTACGTACGTA\n" for i in range(nq)).encode()
    recs, lines = [], []
    for qi in range(nq):
        k = int(rng.integers(0, 12)) if qi % 50 else 300
        docs = rng.choice(n_docs, size=k, replace=False)
        scores = rng.integers(1, 4, size=k)
        order = sorted(range(k), key=lambda j: (-int(scores[j]), int(docs[j])))
        lines.append(f"*q{qi} c{qi % 7}\t{k}\n")
        for j in order:
            recs.append((qi, int(docs[j]), int(scores[j]), 3))
            lines.append(f"{names[docs[j]]}\t{int(scores[j])}\n")
    text = "".join(lines)
    hits = np.array(recs, dtype=pm.HIT_DTYPE)
    q = pm.Queries(fasta, term_size=31)
    ix = pm.Index.from_names(names)
    other = hits.copy()
    other["slot"] = 9                                    # records of another slot around it
    both = np.concatenate([hits, other])
    shuffled = both[rng.permutation(len(both))]
    for n in (-1, 1, 2, 100):
        want = text if n < 0 else P.filter_text(text, n)
        assert pm.format_hits(ix, q, both, slot=3, nb_best_hits=n).decode() == want
        assert pm.format_hits(ix, q, shuffled, slot=3, nb_best_hits=n).decode() == want
    assert pm.format_hits(ix, q, both, slot=5).decode() == "".join(f"*q{i} c{i % 7}\t0\n" for i in range(nq))
    bad = hits.copy()
    bad["doc"][len(bad) // 2] = n_docs + 5
    with pytest.raises(pm.PMError):
        pm.format_hits(ix, q, bad, slot=3)


def test_parallel_gzip_members_decode_like_gzip_fast(tmp_path):
    """03_match container (a9): multi-member gzip written on a thread pool decodes to the same
    bytes with Python's gzip (what xopen uses in scripts/filter_queries.py) and the gzip CLI"""
    import gzip
    import subprocess
    from phylign_amd import pgzip
    rng = np.random.default_rng(6)
    lines = [b"*q%d\t%d\n" % (i, i % 7) + b"".join(b"_SAM%06d\t%d\n" % (int(d), 90 + int(d) % 30) for d in rng.integers(0, 10**6, size=i % 40))
             for i in range(60000)]
    text = b"".join(lines)
    assert len(text) > 3 * pgzip.CHUNK
    parts = pgzip.split_lines(text)
    assert b"".join(parts) == text and len(parts) > 3 and all(p.endswith(b"\n") for p in parts)
    for payload in (text, b"", b"*only\t0\n"):
        p = tmp_path / "x.gz"
        pgzip.write(str(p), payload)
        assert gzip.open(p, "rb").read() == payload
        assert subprocess.run(["gzip", "-dc", str(p)], capture_output=True, check=True).stdout == payload
        assert not (tmp_path / "x.gz.tmp").exists()


def test_fast_gzip_encoder_streams_decode_to_their_input(tmp_path):
    """pm_gzip_fast (the level-1 encoder of the 03_match writer: deflate with line-structured matches and Huffman
    codes built per member): every stream decodes, with Python's gzip and the gzip CLI, to exactly the bytes that went in --
    result-shaped text, binary noise (9-bit literals), very long lines (matches of 258, distances up to and
    beyond the 32 KiB window), repeated names near and far, no trailing newline, more than one member"""
    import gzip
    import subprocess
    from phylign_amd import _lib as pm
    rng = np.random.default_rng(61)
    cases = [b"", b"\n", b"a", b"abc\n" * 3, b"\xff\x90\x8f" * 1000, bytes(rng.integers(0, 256, 100000, dtype=np.uint8)),
             b"x" * 100000 + b"\n" + b"x" * 100000, (b"y" * 40000 + b"\n") * 5, (b"z" * 32767 + b"\n") * 3, (b"w" * 32768 + b"\n") * 3,
             b"_SAM1\t5\n" * 9000, b"".join(b"*read%d\t0\n" % i for i in range(250000))]
    # literal counts like Fibonacci numbers: an unlimited Huffman code would be 27 bits deep (the 15-bit limit and its
    # repair); every byte value equally often; one value only; two values
    fib = [1, 1]
    while len(fib) < 28:
        fib.append(fib[-1] + fib[-2])
    skew = np.concatenate([np.full(f, 11 + i, dtype=np.uint8) for i, f in enumerate(fib)])
    cases += [bytes(rng.permutation(skew)), bytes(rng.permutation(np.tile(np.arange(256, dtype=np.uint8), 300))), b"\x00" * 70000,
              bytes(rng.integers(65, 67, 50000, dtype=np.uint8))]
    for _ in range(150):
        names = [b"_SAM%07d" % rng.integers(0, 10**7) for _ in range(int(rng.integers(1, 60)))]
        lines = []
        for i in range(int(rng.integers(1, 500))):
            r = rng.random()
            if r < 0.3:
                lines.append(b"*read%d c\t%d" % (i * int(rng.integers(1, 3)), rng.integers(0, 50)))
            elif r < 0.9:
                lines.append(names[int(rng.integers(0, len(names)))] + b"\t%d" % rng.integers(80, 121))
            else:
                lines.append(bytes(rng.integers(0, 256, int(rng.integers(0, 700)), dtype=np.uint8)).replace(b"\n", b" "))
        cases.append(b"\n".join(lines) + (b"\n" if rng.random() < 0.8 else b""))
    for i, t in enumerate(cases):
        g = pm.gzip_fast(t)
        assert gzip.decompress(g) == t, i
    big = cases[-1] * 3000                               # several members
    g = pm.gzip_fast(big)
    assert g.count(b"\x1f\x8b\x08\x00") >= 2 and gzip.decompress(g) == big
    p = tmp_path / "x.gz"
    p.write_bytes(g)
    assert subprocess.run(["gzip", "-dc", str(p)], capture_output=True, check=True).stdout == big
    assert subprocess.run(["gzip", "-t", str(p)]).returncode == 0


def test_native_merge_name_table_at_scale_keeps_dict_semantics(tmp_path):
    """100 k query names (the regioned, thread-built name table of pm_merge_create) with duplicates: a lookup finds the
    LAST record of a name, the FASTA has one record per distinct name at its first position -- against the Python
    mirror of scripts/filter_queries.py (itself pinned by fixtures captured from the reference)"""
    import gzip
    import io
    from phylign_amd import _lib as pm
    from phylign_amd import filter_queries as F
    rng = np.random.default_rng(62)
    nq = 100000
    qn = [f"r{i}" for i in range(nq)]
    for i in rng.choice(nq, size=300, replace=False):
        qn[int(i)] = qn[int(rng.integers(0, nq))]                     # duplicates, some of them chains
    assert len(set(qn)) < nq
    fasta = "".join(f">{n} x\n{'ACGT' * 8 + 'ACGTACGTA'[: 1 + i % 9]}\n" for i, n in enumerate(qn)).encode()
    q = pm.Queries(fasta, term_size=31)
    ix = pm.Index.from_names([f"{d:04x}_S{d}" for d in range(50)])
    hit_q = np.sort(rng.choice(nq, size=5000, replace=False)).astype(np.uint32)
    rec = np.zeros(len(hit_q) * 2, dtype=pm.HIT_DTYPE)
    rec["query"] = np.repeat(hit_q, 2)
    rec["doc"] = np.tile(np.array([3, 7], dtype=np.uint32), len(hit_q))
    rec["score"] = np.tile(np.array([9, 8], dtype=np.uint32), len(hit_q))
    m = pm.Merge(q, 3)
    m.add("b1", ix, rec, slot=0)
    qf, mf = tmp_path / "q.fa", tmp_path / "b1____q.gz"
    qf.write_bytes(fasta)
    with gzip.open(mf, "wb", compresslevel=1) as g:
        g.write(pm.format_hits(ix, q, rec, slot=0, nb_best_hits=3))
    out = io.StringIO()
    F.filter_files(str(qf), [str(mf)], 3, out)
    assert m.emit().decode() == out.getvalue()
    assert out.getvalue().count(">") == len(set(qn))
    n = m.emit_to(str(tmp_path / "f.fa"))
    assert (tmp_path / "f.fa").read_bytes().decode() == out.getvalue() and n == len(out.getvalue())


def test_merge_fasta_of_a_chunked_query_file_is_written_piece_by_piece(tmp_path):
    """one merge per query chunk (match_stage --query-chunk): pm_merge_emit_file_piece appends the chunks' records in file
    order; the file appears with the last piece and equals the concatenation of the merges' texts"""
    from phylign_amd import _lib as pm
    rng = np.random.default_rng(71)
    fa = [b"".join(b">c%dq%d x\nACGTACGTACGTACGTACGTACGTACGTACGTA\n" % (c, i) for i in range(20000)) for c in range(3)]
    qs = [pm.Queries(f) for f in fa]
    ix = pm.Index.from_names([f"{i:03x}_R{i}" for i in range(30)])
    ms = []
    for q in qs:
        m = pm.Merge(q, 5)
        rec = np.zeros(3000, dtype=pm.HIT_DTYPE)
        rec["query"] = np.sort(rng.integers(0, 20000, 3000)); rec["doc"] = rng.integers(0, 30, 3000); rec["score"] = rng.integers(1, 9, 3000)
        m.add("b", ix, pm.sort_hits(rec), slot=0)
        ms.append(m)
    p = tmp_path / "f.fa"
    assert ms[0].emit_to(str(p), 1) and not p.exists() and (tmp_path / "f.fa.tmp").exists()
    n = pm.emit_merges_to(ms, str(p))
    want = b"".join(m.emit() for m in ms)
    assert p.read_bytes() == want and n == len(want) and not (tmp_path / "f.fa.tmp").exists()
    assert pm.emit_merges_to(ms[:1], str(p)) == len(ms[0].emit()) and p.read_bytes() == ms[0].emit()
    with pytest.raises(pm.PMError):
        ms[0].emit_to(str(p), 9)


def test_header_reader_survives_corrupted_input(oracle):
    """classic / compact header bytes with random flips, truncations and insertions: the product
    reader either reports PM_EFORMAT / PM_EIO-style errors or returns a consistent handle -- it never
    crashes and never reads outside the buffer (pm_index_load_header_mem needs no GPU)"""
    from phylign_amd import _lib as pm
    rng = np.random.default_rng(12)
    names = [f"{i:05x}_SAM{i:06d}" for i in range(37)]
    classic = bytes(oracle.make_index(31, 1, 500, 1, names))
    compact = bytes(oracle.make_compact(31, 1, 16, [300, 200], [1, 2], names + [f"x_{i}" for i in range(100)]))
    ok = bad = 0
    for base in (classic, compact):
        head_len = base.index(b"_INDEX", 20) + 6 if b"_INDEX" in base[20:] else len(base)
        for _ in range(300):
            b = bytearray(base[: head_len + int(rng.integers(0, 64))])
            op = int(rng.integers(0, 4))
            if op == 0:
                for _ in range(int(rng.integers(1, 6))):
                    b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
            elif op == 1:
                b = b[: int(rng.integers(0, len(b)))]
            elif op == 2:
                p = int(rng.integers(0, len(b)))
                b[p:p] = bytes(rng.integers(0, 256, size=int(rng.integers(1, 40)), dtype=np.uint8))
            else:
                p = int(rng.integers(18, min(len(b), 60)))
                b[p:p + 8] = (int(rng.integers(0, 2**63))).to_bytes(8, "little")
            try:
                ix = pm.Index.load_header_mem(bytes(b))
                info = ix.info
                assert info.n_docs < 10**7 and info.has_matrix == 0
                for d in range(min(info.n_docs, 5)):
                    ix.doc_name(d)
                ok += 1
            except pm.PMError as e:
                assert e.code in (-5, -4, -1), e
                bad += 1
    assert bad > 200 and ok + bad == 600


def test_oracle_xxh64_against_python_xxhash_live(oracle):
    """beyond the committed known answers: random inputs of every length 0..80 and random seeds
    against the python-xxhash wheel (libxxhash), when it is installed"""
    xxhash = pytest.importorskip("xxhash")
    rng = np.random.default_rng(99)
    for n in list(range(0, 81)) + [127, 128, 129, 1000]:
        for _ in range(4):
            data = rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
            seed = int(rng.integers(0, 2**63))
            assert oracle.xxh64(data, seed) == xxhash.xxh64(data, seed=seed).intdigest(), (n, seed)
    for kmer in (b"ACGTACGTACGTACGTACGTACGTACGTACG", b"T" * 31, b"GATTACA" * 4 + b"GAT"):
        for seed in range(4):
            assert oracle.xxh64(kmer, seed) == xxhash.xxh64(kmer, seed=seed).intdigest()


def test_server_routing_is_stable_and_spreads_the_661k_batches():
    """one resident-index server per GPU: a batch always goes to the same socket, whatever its
    extension, and the 305 batches of the 661k collection fit 8 x 288 GB with that map"""
    from phylign_amd import workload as W
    from phylign_amd.cobs_query import pick_server
    socks = ",".join(f"/tmp/pm{i}.sock" for i in range(8))
    shapes = W.load_shapes()
    load = {}
    for s in shapes:
        a = pick_server(socks, f"cobs/{s.batch}.cobs_classic.xz")
        assert a == pick_server(socks, f"/elsewhere/{s.batch}.cobs_classic") == pick_server(socks, s.batch)
        load[a] = load.get(a, 0) + s.index_bytes * 1.1            # line-aligned rows cost a few percent more
    assert len(load) == 8 and max(load.values()) < 0.85 * 288e9, {k: round(v / 1e9) for k, v in load.items()}
    assert pick_server("/tmp/one.sock", "x.xz") == "/tmp/one.sock"


# --------------------------------------- fix_query fused into the native parser (f4, pm_queries_parse_raw)
# Hand-derived from the documented behaviour of `seqtk seq -A -U -C` (kseq record rules) followed by the
# awk substitution of Snakefile:330-332; seqtk itself is not in the build container.
RAW_CASES = [
    # multi-line FASTA, lower case, IUPAC codes, comment after the name, blank line inside the sequence
    (b">r1 some comment here\nacgtnnry\n\nACGTACGT\nacgtacgtacgtacgtacgt\n",
     b">r1\nACGTAAAAACGTACGTACGTACGTACGTACGTACGT\n"),
    # name cut at a TAB as well; CRLF line ends
    (b">r2\tx y\r\nACGTACGTACGTACGTACGTACGTACGTACGTAC\r\n",
     b">r2\nACGTACGTACGTACGTACGTACGTACGTACGTAC\n"),
    # FASTQ: quality dropped, '@' and '>' as first quality characters do not start records
    (b"@q1 c\nACGTACGTACGTACGTACGTACGTACGTACGTAC\n+\n@IIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIII\n"
     b"@q2\nGGGGACGTACGTACGTACGTACGTACGTACGTAC\n+q2\n>IIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIII\n",
     b">q1\nACGTACGTACGTACGTACGTACGTACGTACGTAC\n>q2\nGGGGACGTACGTACGTACGTACGTACGTACGTAC\n"),
    # multi-line FASTQ sequence and quality
    (b"@m\nACGTACGTACGTACGTAC\nGTACGTACGTACGTAC\n+\nIIIIIIIIIIIIIIIIII\nIIIIIIIIIIIIIIII\n>n\nTTTTACGTACGTACGTACGTACGTACGTACGTAC\n",
     b">m\nACGTACGTACGTACGTACGTACGTACGTACGTAC\n>n\nTTTTACGTACGTACGTACGTACGTACGTACGTAC\n"),
    # junk before the first header is skipped; a record without sequence disappears (cobs ignores the empty line)
    (b"junk line\n\n>empty\n>full z\nUUUUACGTACGTACGTACGTACGTACGTACGTAC\n",
     b">full\nAAAAACGTACGTACGTACGTACGTACGTACGTAC\n"),
    # digits, '-', '*', '.' and blanks inside a sequence line all become A (awk: every byte outside ACGT)
    (b">g\nAC-GT*N.1 ACGTACGTACGTACGTACGTACGTAC\n", b">g\nACAGTAAAAAACGTACGTACGTACGTACGTACGTAC\n"),
    # truncated quality block: seqtk's read loop ends there, the records before it survive
    (b">a\nACGTACGTACGTACGTACGTACGTACGTACGTAC\n@b\nACGTACGTACGTACGTACGTACGTACGTACGTAC\n+\nIII\n",
     b">a\nACGTACGTACGTACGTACGTACGTACGTACGTAC\n"),
    # no trailing newline
    (b">z\nacgtacgtacgtacgtacgtacgtacgtacgtac", b">z\nACGTACGTACGTACGTACGTACGTACGTACGTAC\n"),
]


@pytest.mark.parametrize("case", range(len(RAW_CASES)))
def test_native_fix_query_rules(case):
    from phylign_amd import _lib as pm
    raw, want = RAW_CASES[case]
    q = pm.Queries(raw, term_size=31, normalise=True)
    assert q.fasta() == want
    # the prepared file parses to the same set without normalisation, and is a fixed point of the rule
    q2 = pm.Queries(want, term_size=31)
    assert q2.fasta() == want and q2.count() == q.count()
    assert pm.Queries(want, term_size=31, normalise=True).fasta() == want


def test_native_fix_query_agrees_with_the_python_mirror_and_awk(tmp_path):
    """same records as phylign_amd/fix_query.py (the Python mirror) on a random mixed file, and the awk half
    of the reference rule leaves the native output unchanged"""
    from phylign_amd import _lib as pm
    from phylign_amd import fix_query as FQ
    rng = np.random.default_rng(5)
    alpha = np.frombuffer(b"ACGTacgtNnRYKMSWryu", dtype=np.uint8)
    parts = []
    for i in range(300):
        n = int(rng.integers(31, 400))
        seq = alpha[rng.integers(0, len(alpha), size=n)].tobytes()
        width = int(rng.integers(20, 90))
        lines = b"\n".join(seq[j:j + width] for j in range(0, n, width))
        if i % 3 == 0:
            qual = bytes(rng.integers(33, 74, size=n).astype(np.uint8))
            qlines = b"\n".join(qual[j:j + width] for j in range(0, n, width))
            parts.append(b"@rec%d comment %d\n" % (i, i) + lines + b"\n+\n" + qlines + b"\n")
        else:
            parts.append(b">rec%d\tdesc\n" % i + lines + b"\n")
    raw = b"".join(parts)
    got = pm.Queries(raw, term_size=31, normalise=True).fasta()
    out = io.BytesIO()
    FQ.fix_stream(io.BytesIO(raw), out)
    assert got == out.getvalue()
    p = tmp_path / "x.fa"
    p.write_bytes(got)
    r = subprocess.run(["awk", '{if(NR%2==1){print $0;}else{gsub(/[^ACGT]/, "A"); print;}}', str(p)], capture_output=True)
    assert r.returncode == 0 and r.stdout == got
    # the reference's bundled reads, already prepared: unchanged
    merged = open(os.path.join(GOLD, "reads", "reads_1___reads_2___reads_3___reads_4.fa"), "rb").read()
    assert pm.Queries(merged, term_size=31, normalise=True).fasta() == merged


def test_unprepared_queries_are_rejected_without_normalise():
    from phylign_amd import _lib as pm
    with pytest.raises(pm.PMError) as e:
        pm.Queries(b">r\nacgtacgtacgtacgtacgtacgtacgtacgtacgt\n", term_size=31)
    assert e.value.code == -6


def test_cpu_baseline_partitions_agree_with_the_oracle_scores(oracle):
    """bench.py's cpu_baseline legs (query partition and cobs -T-like column slabs) count exactly the
    documents the oracle's scoring + threshold rule selects"""
    O = oracle
    rng = np.random.default_rng(3)
    for n_docs, S in ((1000, 3000), (22 * 8 - 3, 2500)):
        m = O.synth_fill(661, 7, S, n_docs, 2)
        h = O.Header()
        h.term_size, h.canonicalize, h.num_hashes = 31, 1, 1
        h.n_docs, h.signature_size, h.row_bytes = n_docs, S, (n_docs + 7) // 8
        nq, qlen = 37, 150
        seqs = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(nq, qlen))]
        want = 0
        for i in range(nq):
            sc = O.scores_rows(lambda r: m[r], h, seqs[i].tobytes())
            want += len(O.select(sc, qlen - 30, 0.2))
        assert want > 0
        assert O.baseline_run(m, h.row_bytes, h, seqs.tobytes(), qlen, nq, 0.2, 3) == want
        for slab in (16, 64, 1000):
            assert O.baseline_run_slabs(m, h.row_bytes, h, seqs.tobytes(), qlen, nq, 0.2, 3, slab) == want


def test_native_fix_query_on_the_reference_bundled_inputs():
    """the four query files `make test` feeds to rule fix_query (data/reads_1.fastq, reads_2.fq, reads_3.fasta,
    reads_4.fa; Snakefile:308-352): the native parser turns each into its prepared form, and their concatenation is
    the merged query file of the 03_match stage"""
    from phylign_amd import _lib as pm
    raw_dir = os.path.join(GOLD, "reads", "raw")
    parts = []
    for f in ("reads_1.fastq", "reads_2.fq", "reads_3.fasta", "reads_4.fa"):
        raw = open(os.path.join(raw_dir, f), "rb").read()
        q = pm.Queries(raw, term_size=31, normalise=True)
        assert q.count()[0] == 10
        parts.append(q.fasta())
    merged = open(os.path.join(GOLD, "reads", "reads_1___reads_2___reads_3___reads_4.fa"), "rb").read()
    assert b"".join(parts) == merged
    # a FASTQ is not a prepared query file: without normalisation the cobs record rules make nonsense of it -> error
    with pytest.raises(pm.PMError):
        pm.Queries(open(os.path.join(raw_dir, "reads_1.fastq"), "rb").read(), term_size=31)


# ------------------------------------------------ native reader of 03_match text (drop-in scripts/filter_queries.py)
@pytest.mark.parametrize("keep", [1, 2, 5, 100])
def test_native_merge_reads_match_text_like_the_reference_filter(keep):
    """pm_merge_add_text on the gunzipped fixture files = the output captured from the reference's filter_queries.py"""
    import gzip
    from phylign_amd import _lib as pm
    d = os.path.join(GOLD, "filter")
    q = pm.Queries(open(os.path.join(d, "queries.fa"), "rb").read(), term_size=1)
    m = pm.Merge(q, keep)
    for b in ("aaa_bbb__01", "ccc_ddd__01", "ccc_ddd__02"):
        m.add_text(b, gzip.open(os.path.join(d, f"{b}____q.gz"), "rb").read())
    assert m.emit().decode() == open(os.path.join(d, f"expected.n{keep}.fa")).read()


def test_filter_queries_drop_in_script_native_and_python(tmp_path):
    """scripts/filter_queries.py with the reference's argv: native reader and --python reader give the fixture"""
    d = os.path.join(GOLD, "filter")
    files = [os.path.join(d, f"{b}____q.gz") for b in ("aaa_bbb__01", "ccc_ddd__01", "ccc_ddd__02")]
    want = open(os.path.join(d, "expected.n2.fa"), "rb").read()
    for extra in ([], ["--python"]):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "filter_queries.py"), "-n", "2", "-q",
                            os.path.join(d, "queries.fa")] + files + extra, capture_output=True)
        assert r.returncode == 0, r.stderr.decode()
        assert r.stdout == want and r.stderr.count(b"Translating matches") == 3


def test_native_match_text_reader_random_against_the_python_mirror(tmp_path):
    """random 03_match files (ties, comments in headers, blank lines, CRLF, shared references across batches): the
    native reader and the golden-pinned Python mirror write the same FASTA"""
    import gzip
    from phylign_amd import _lib as pm
    from phylign_amd import filter_queries as F
    rng = np.random.default_rng(17)
    nq = 300
    qnames = [f"read{i}" for i in range(nq)]
    fa = "".join(f">{n} comment {i}\n{''.join('ACGT'[j] for j in rng.integers(0, 4, 40))}\n" for i, n in enumerate(qnames))
    (tmp_path / "q.fa").write_text(fa)
    paths = []
    for b in range(7):
        lines = []
        for i in rng.permutation(nq)[: int(rng.integers(50, nq))]:
            k = int(rng.integers(0, 9))
            lines.append(f"*{qnames[i]} c{i}\t{k}" + ("\r" if i % 11 == 0 else ""))
            for _ in range(k):
                lines.append(f"{rng.integers(0, 16**5):05x}_SAM{rng.integers(0, 40):03d}\t{rng.integers(20, 30)}")
            if i % 7 == 0:
                lines.append("")
        p = tmp_path / f"genus_sp{b % 3}__{b:02d}____q.gz"
        with gzip.open(p, "wt") as f:
            f.write("\n".join(lines) + "\n")
        paths.append(str(p))
    for keep in (1, 3, 100):
        out = io.StringIO()
        F.filter_files(str(tmp_path / "q.fa"), paths, keep, out)
        got = io.BytesIO()
        assert F.filter_files_native(str(tmp_path / "q.fa"), paths, keep, got)
        assert got.getvalue().decode() == out.getvalue()


@pytest.mark.parametrize("text,needle", [
    (b"xx_A\t5\n", b"no '*' query header"),
    (b"", b"no '*' query header"),
    (b"*r1\tabc\n", b"integer match count"),
    (b"*r1\n", b"integer match count"),
    (b"*nobody\t0\n", b"not in the query file"),
    (b"*r1\t1\nxx_A_B\t5\n", b"exactly one '_'"),
    (b"*r1\t1\nxxA\t5\n", b"exactly one '_'"),
    (b"*r1\t1\nxx_A 5 6\n", b"'<name> <k-mers>'"),
    (b"*r1\t1\nxx_A\tfive\n", b"'<name> <k-mers>'"),
])
def test_native_match_text_reader_fails_where_the_reference_raises(text, needle):
    from phylign_amd import _lib as pm
    q = pm.Queries(b">r1\nACGT\n", term_size=1)
    m = pm.Merge(q, 5)
    with pytest.raises(pm.PMError) as e:
        m.add_text("b__01", text)
    assert needle in str(e.value).encode()


def _edge_cases():
    d = os.path.join(GOLD, "filter", "edge")
    return sorted(f[: -len("____q.txt")] for f in os.listdir(d) if f.endswith("____q.txt"))


@pytest.mark.parametrize("case", _edge_cases())
def test_filter_readers_on_edge_cases_captured_from_the_reference(case):
    """match files with odd content run through the reference's filter_queries.py (tools/gen_golden_filter_edge.py):
    where it printed a FASTA, the native reader and the Python mirror print the same; where it raised, both fail"""
    from phylign_amd import _lib as pm
    from phylign_amd import filter_queries as F
    d = os.path.join(GOLD, "filter")
    path = os.path.join(d, "edge", f"{case}____q.txt")
    want_path = os.path.join(d, "edge", f"{case}.n3.fa")
    qfa = os.path.join(d, "queries.fa")
    q = pm.Queries(open(qfa, "rb").read(), term_size=1)
    m = pm.Merge(q, 3)
    text = open(path, "rb").read()
    if os.path.exists(want_path):
        want = open(want_path, "rb").read()
        m.add_text(case, text)
        assert m.emit() == want
        out = io.StringIO()
        F.filter_files(qfa, [path], 3, out)
        assert out.getvalue().encode() == want
    else:
        assert os.path.exists(os.path.join(d, "edge", f"{case}.crash"))
        with pytest.raises(pm.PMError):
            m.add_text(case, text)
        with pytest.raises(Exception):
            F.filter_files(qfa, [path], 3, io.StringIO())


@pytest.mark.parametrize("case", ["ties_large", "random_mid", "zero_hits"])
def test_native_match_file_writer(case, tmp_path):
    """pm_format_hits_gz: the 03_match file of a batch in one call -- its gunzipped bytes are pm_format_hits' (= the
    reference post-filter's, by the fixtures), `gzip -dc` reads the multi-member file, no .tmp is left"""
    import gzip
    from phylign_amd import _lib as pm
    base = os.path.join(GOLD, "postprocess", case)
    fasta, names, hits = _structured(open(base + ".in").read())
    q = pm.Queries(fasta, term_size=31)
    ix = pm.Index.from_names(names)
    for n in (-1, 2, 100):
        want = pm.format_hits(ix, q, hits, slot=0, nb_best_hits=n)
        path = tmp_path / f"{case}.{n}.gz"
        tb, zb = pm.format_hits_gz(ix, q, hits, str(path), slot=0, nb_best_hits=n, level=1)
        assert tb == len(want) and zb == os.path.getsize(path)
        assert gzip.open(path, "rb").read() == want
        assert subprocess.run(["gzip", "-dc", str(path)], capture_output=True, check=True).stdout == want
    assert not list(tmp_path.glob("*.tmp"))
    with pytest.raises(pm.PMError):
        pm.format_hits_gz(ix, q, hits, str(tmp_path / "no_such_dir" / "x.gz"))


def test_native_match_file_writer_many_members(tmp_path):
    """a text of several 4 MiB chunks: members are cut at line boundaries and decode to the same bytes"""
    import gzip
    from phylign_amd import _lib as pm
    n = 400_000
    fasta = b"".join(b">query_number_%07d with a comment\nACGTACGTACGTACGTACGTACGTACGTACGTACG\n" % i for i in range(n))
    q = pm.Queries(fasta, term_size=31)
    ix = pm.Index.from_names([f"{i:05x}_REF{i:04d}" for i in range(300)])
    rng = np.random.default_rng(2)
    recs = np.zeros(n // 2, dtype=pm.HIT_DTYPE)
    recs["query"] = np.sort(rng.integers(0, n, n // 2)); recs["doc"] = rng.integers(0, 300, n // 2); recs["score"] = 5
    pm.sort_hits(recs)
    want = pm.format_hits(ix, q, recs, slot=0, nb_best_hits=100)
    assert len(want) > 3 * (4 << 20)
    tb, zb = pm.format_hits_gz(ix, q, recs, str(tmp_path / "big.gz"), slot=0, nb_best_hits=100)
    raw = open(tmp_path / "big.gz", "rb").read()
    assert raw.count(b"\x1f\x8b\x08") >= 4 and tb == len(want) and zb == len(raw)
    assert gzip.decompress(raw) == want


def test_large_query_file_is_parsed_in_pieces_with_the_same_result():
    """files of 16 MB and more are cut at header lines and parsed on several threads: same records as the small-file
    path (multi-line records, ';' headers, a headerless first record), the first error in file order is the one reported"""
    from phylign_amd import _lib as pm
    rng = np.random.default_rng(9)
    n = 120_000
    seqs = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(n, 150))]
    recs = []
    for i in range(n):
        s = seqs[i].tobytes()
        if i % 1000 == 0:
            recs.append(b";semi %d\n" % i + s[:70] + b"\n" + s[70:] + b"\n\n")
        else:
            recs.append(b">r%d some words\n" % i + s + b"\n")
    big = b"ACGTACGTACGTACGTACGTACGTACGTACGTACGTA\n" + b"".join(recs)
    assert len(big) > (16 << 20)
    q = pm.Queries(big)
    assert q.count()[0] == n + 1
    want = b"\nACGTACGTACGTACGTACGTACGTACGTACGTACGTA\n" + b"".join(       # (a record without header line has an empty one)
        (b">semi %d\n" % i if i % 1000 == 0 else b">r%d some words\n" % i) + seqs[i].tobytes() + b"\n" for i in range(n))
    assert q.fasta() == want
    small = [pm.Queries(b"".join(recs[a:a + 1000])) for a in (0, 57_000, 119_000)]        # the same records through the one-piece path
    assert small[1].fasta() == b"".join(
        (b">semi %d\n" % i if i % 1000 == 0 else b">r%d some words\n" % i) + seqs[i].tobytes() + b"\n" for i in range(57_000, 58_000))
    bad = recs[:]
    bad[40_000] = b">early\nACGTNNACGTACGTACGTACGTACGTACGTACGTACGT\n"
    bad[110_000] = b">late\nACG\n"
    with pytest.raises(pm.PMError) as e:
        pm.Queries(b"".join(bad))
    assert e.value.code == -6 and "early" in str(e.value)


def test_worker_pool_serves_concurrent_callers(tmp_path):
    """text formatting, the gz writer, the FASTA emit and the query parser share one persistent worker pool: eight caller
    threads at once, many rounds, every result equal to the single-caller result"""
    import gzip
    import threading
    from phylign_amd import _lib as pm
    n = 60_000
    fasta = b"".join(b">q%06d\nACGTACGTACGTACGTACGTACGTACGTACGTACG\n" % i for i in range(n))
    q = pm.Queries(fasta, term_size=31)
    ix = pm.Index.from_names([f"{i:05x}_R{i:03d}" for i in range(200)])
    rng = np.random.default_rng(6)
    recs = np.zeros(3 * n, dtype=pm.HIT_DTYPE)
    recs["query"] = np.repeat(np.arange(n, dtype=np.uint32), 3); recs["doc"] = rng.integers(0, 200, 3 * n); recs["score"] = rng.integers(1, 6, 3 * n)
    pm.sort_hits(recs)
    want = pm.format_hits(ix, q, recs, slot=0, nb_best_hits=2)
    m = pm.Merge(q, 2)
    m.add("b__01", ix, recs, slot=0, nb_best_hits=2)
    want_fa = m.emit()
    errors = []

    def caller(t):
        try:
            for r in range(6):
                assert pm.format_hits(ix, q, recs, slot=0, nb_best_hits=2) == want
                p = tmp_path / f"t{t}.gz"
                pm.format_hits_gz(ix, q, recs, str(p), slot=0, nb_best_hits=2)
                assert gzip.open(p, "rb").read() == want
                assert m.emit() == want_fa
                assert pm.Queries(fasta * 6, term_size=31).count()[0] == 6 * n        # 16 MB+: the piecewise parser
        except Exception as e:                                                       # noqa: BLE001
            errors.append(repr(e))
    th = [threading.Thread(target=caller, args=(t,)) for t in range(8)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errors, errors[:2]


def test_match_file_pieces_and_query_file_splitting(tmp_path):
    """a query file cut into chunks (match_stage --query-chunk): the pieces of a batch's file, written chunk after chunk
    (pm_format_hits_gz_piece), gunzip to the one-piece text; the splitter cuts only at record starts"""
    import gzip
    from phylign_amd import _lib as pm
    from phylign_amd.match_stage import split_prepared_fasta
    n = 1000
    recs = [(b";semi%d\n" % i if i % 97 == 0 else b">q%04d c\n" % i) + b"ACGTACGTACGTACGTACGTACGTACGTACGTACG\n" for i in range(n)]
    fasta = b"".join(recs)
    for size in (0, 1, 250, 333, 999, 1000, 5000):
        pieces = split_prepared_fasta(fasta, size)
        assert b"".join(pieces) == fasta
        assert all(p[:1] in (b">", b";") for p in pieces)
        assert len(pieces) == (1 if size <= 0 or size >= n else -(-n // size))
    # a file of several counting pieces (8 MB each, counted on several threads): the cuts are those of a plain scan
    big = b"".join((b";s%d\n" % i if i % 1013 == 0 else b">read%07d some comment\n" % i) + b"ACGT" * (9 + i % 5) + b"\n" for i in range(450000))
    arr = np.frombuffer(big, dtype=np.uint8)
    nl = np.flatnonzero(arr[:-1] == 10)
    starts = np.concatenate(([0], nl[(arr[nl + 1] == 62) | (arr[nl + 1] == 59)] + 1))
    assert len(big) > 3 * (8 << 20) and len(starts) == 450000
    for size in (1, 7, 4096, 100000, 449999, 450000):
        got = pm.fasta_record_cuts(big, size)
        assert got == [int(x) for x in starts[size::size]], size
    assert pm.fasta_record_cuts(b"", 5) == [] and pm.fasta_record_cuts(b"ACGT\n>a\nAC\n>b\nAC\n", 1) == [11]
    # pieces by bytes (parsing overlaps the search of earlier pieces): cut at record starts only, nothing lost
    for pb in (1 << 20, 3 << 20, 64 << 20):
        pieces = split_prepared_fasta(big, 100000, pb)
        assert b"".join(pieces) == big and all(p[:1] in (b">", b";") for p in pieces)
        assert len(pieces) >= 5 and (pb > len(big) or max(len(p) for p in pieces) <= 2 * pb + 200)
    assert len(split_prepared_fasta(big, 0, 1 << 20)) > 10 and split_prepared_fasta(big, 0, 0) == [big]
    ix = pm.Index.from_names([f"{i:05x}_R{i}" for i in range(40)])
    rng = np.random.default_rng(4)
    whole_q = pm.Queries(fasta, term_size=31)
    recs_all = np.zeros(3 * n, dtype=pm.HIT_DTYPE)
    recs_all["query"] = np.repeat(np.arange(n, dtype=np.uint32), 3); recs_all["doc"] = rng.integers(0, 40, 3 * n); recs_all["score"] = rng.integers(1, 6, 3 * n)
    pm.sort_hits(recs_all)
    want = pm.format_hits(ix, whole_q, recs_all, slot=0, nb_best_hits=2)
    pieces = split_prepared_fasta(fasta, 333)
    path = tmp_path / "b____q.gz"
    first = 0
    for ci, piece in enumerate(pieces):
        qc = pm.Queries(piece, term_size=31)
        k = qc.count()[0]
        part = recs_all[(recs_all["query"] >= first) & (recs_all["query"] < first + k)].copy()
        part["query"] -= first
        pm.format_hits_gz(ix, qc, part, str(path), slot=0, nb_best_hits=2, piece=1 if ci == 0 else (3 if ci == len(pieces) - 1 else 2))
        assert path.exists() == (ci == len(pieces) - 1)           # the file appears with its last piece
        first += k
    assert gzip.open(path, "rb").read() == want and not (tmp_path / "b____q.gz.tmp").exists()
    with pytest.raises(pm.PMError):
        pm.format_hits_gz(ix, whole_q, recs_all, str(path), piece=7)


def test_native_match_text_reader_against_the_mirror_on_random_odd_input(tmp_path):
    """pm_merge_add_text vs the Python mirror of scripts/filter_queries.py (fixture-pinned) on random 03_match-like text
    with the oddities a text reader meets: counts written as Python's int() accepts them (' 5', '+2', '-4', '1_0'),
    several blanks or TABs between the fields, hit lines ahead of the first header, names with no or two '_', unknown
    queries, repeated read names.  Both fail, or both succeed with the same FASTA.  (A negative k-mer count is a valid
    line that never counts: the reference keeps kmers >= floor, and the floor starts at 0.)"""
    import gzip
    import io
    from phylign_amd import _lib as pm
    from phylign_amd import filter_queries as F
    rng = np.random.default_rng(91)
    fixed = [b"*q0\t 5\nab_R1\t3\n", b"*q0\t1_0\nab_R1\t1_2\n", b"ab_R1 -4\n*q0\t+1\ncd_R2\t-0\n", b"*q0\t5 \n\x1cab_R1\x1d7\n",
             b"*q0\t1__0\n", b"*q0\t_1\n", b"*q0\t1\nab_R1\t1_\n"]
    agree = {"ok": 0, "err": 0}
    for it in range(260):
        nq = int(rng.integers(1, 10))
        qn = [f"q{i}" for i in range(nq)]
        if rng.random() < 0.2 and nq > 2:
            qn[1] = qn[0]
        fasta = "".join(f">{n}{' cm' if i % 2 else ''}\n{'ACGT' * 9}\n" for i, n in enumerate(qn)).encode()
        (tmp_path / "q.fa").write_bytes(fasta)
        texts = []
        for b in range(int(rng.integers(1, 4))):
            if it < len(fixed):
                t = fixed[it]
            else:
                lines = []
                for _ in range(int(rng.integers(0, 14))):
                    r = rng.random()
                    ws = str(rng.choice(["\t", " ", "  ", "\t ", "\t"]))
                    if r < 0.35:
                        name = str(rng.choice(qn + ["zz"] if rng.random() < 0.03 else qn))
                        cnt = str(rng.choice(["3", "0", "+2", "-1", "x", "", "7 ", " 5", "1\t9", "1_0"])) if rng.random() < 0.15 else str(int(rng.integers(0, 9)))
                        lines.append(f"*{name}{rng.choice(['', ' c', ' a b'])}\t{cnt}")
                    elif r < 0.9:
                        ref = (str(rng.choice(["ab_R1", "cd_R2", "ef_R10", "g_h_i", "noUnderscore", "_lead", "trail_"])) if rng.random() < 0.12
                               else f"{int(rng.integers(0, 99)):02x}_R{int(rng.integers(0, 6))}")
                        km = str(rng.choice(["12", "+3", "-4", "1.5", "", "9x", "007", "-0"])) if rng.random() < 0.12 else str(int(rng.integers(1, 40)))
                        extra = str(rng.choice(["", " extra", "\t1"])) if rng.random() < 0.05 else ""
                        lead = str(rng.choice(["", " ", "\t"])) if rng.random() < 0.1 else ""
                        lines.append(f"{lead}{ref}{ws}{km}{extra}{' ' if rng.random() < 0.1 else ''}")
                    else:
                        lines.append(str(rng.choice(["", " ", "\t", "*"])))
                t = ("\n".join(lines) + ("\n" if rng.random() < 0.9 else "")).encode()
            fn = tmp_path / f"b{b}____q.gz"
            with gzip.open(fn, "wb") as g:
                g.write(t)
            texts.append((f"b{b}", t, str(fn)))
        keep = int(rng.choice([1, 2, 5]))
        try:
            o = io.StringIO()
            F.filter_files(str(tmp_path / "q.fa"), [x[2] for x in texts], keep, o)
            py = ("ok", o.getvalue())
        except Exception:
            py = ("err", None)
        try:
            m = pm.Merge(pm.Queries(fasta), keep)
            for bn, t, _ in texts:
                m.add_text(bn, t)
            nat = ("ok", m.emit().decode())
        except pm.PMError:
            nat = ("err", None)
        assert py == nat, (it, [x[1] for x in texts])
        agree[py[0]] += 1
    assert agree["ok"] > 30 and agree["err"] > 30


@pytest.mark.parametrize("keep", [1, 3])
def test_merge_over_pieces_of_a_query_file_keeps_the_whole_files_dict_semantics(tmp_path, keep):
    """a query file in pieces (pm_merge_extend) whose read names repeat ACROSS pieces -- the two mates of a read pair in
    concatenated files: the merge over the pieces emits exactly what one merge over the whole file emits, which is what
    the mirror of scripts/filter_queries.py (a dict keyed by name) writes: one record per name where it first occurs,
    the last occurrence's sequence, the best matches of all occurrences together"""
    import gzip
    import io
    from phylign_amd import _lib as pm
    from phylign_amd import filter_queries as F
    from phylign_amd.match_stage import split_prepared_fasta
    rng = np.random.default_rng(81 + keep)
    n = 3000
    names = [f"read{i}" for i in range(n)]
    mates = [f"read{i}" for i in rng.permutation(n)[: n // 2]]                  # second file: mates of half of the reads
    extra = [f"solo{i}" for i in range(200)]
    allq = names + mates + extra + [names[5], names[5]]
    fasta = "".join(f">{nm} c{i}\n{'ACGT' * 8}{'ACGTACGTT'[: 1 + i % 9]}\n" for i, nm in enumerate(allq)).encode()
    nq = len(allq)
    ixs = [pm.Index.from_names([f"{d:03x}_R{b}x{d}" for d in range(40)]) for b in range(3)]
    whole_q = pm.Queries(fasta)
    recs = []
    for b in range(3):
        hq = np.sort(rng.choice(nq, size=nq // 2, replace=False)).astype(np.uint32)
        r = np.zeros(len(hq) * 2, dtype=pm.HIT_DTYPE)
        r["query"] = np.repeat(hq, 2)
        r["doc"] = rng.integers(0, 40, len(r))
        r["score"] = rng.integers(1, 6, len(r))
        recs.append(pm.sort_hits(r))
    whole = pm.Merge(whole_q, keep)
    for b in range(3):
        whole.add(f"b{b}", ixs[b], recs[b], slot=0, nb_best_hits=keep)
    want = whole.emit()
    # the mirror on the files the stage would write
    (tmp_path / "q.fa").write_bytes(fasta)
    files = []
    for b in range(3):
        fn = tmp_path / f"b{b}____q.gz"
        with gzip.open(fn, "wb") as g:
            g.write(pm.format_hits(ixs[b], whole_q, recs[b], slot=0, nb_best_hits=keep))
        files.append(str(fn))
    out = io.StringIO()
    F.filter_files(str(tmp_path / "q.fa"), files, keep, out)
    assert want.decode() == out.getvalue()
    for size in (1000, 1777, nq - 1):
        pieces = [pm.Queries(p) for p in split_prepared_fasta(fasta, size)]
        m = pm.Merge(pieces[0], keep)
        base = [0]
        for p_ in pieces[1:]:
            m.extend(p_)
        for p_ in pieces:
            base.append(base[-1] + p_.count()[0])
        order = [(b, ci) for b in range(3) for ci in range(len(pieces))]
        rng.shuffle(order)                                         # (batch, piece) units arrive in any order
        for b, ci in order:
            part = recs[b][(recs[b]["query"] >= base[ci]) & (recs[b]["query"] < base[ci + 1])].copy()
            part["query"] -= base[ci]
            m.add(f"b{b}", ixs[b], part, slot=0, nb_best_hits=keep, piece=ci)
        assert m.emit() == want, size
        assert sorted(m.batches()) == ["b0", "b1", "b2"]
        # a rank's share exported and added again (numbers through the whole file, piece -1)
        ex = m.export()
        again = pm.Merge(pieces[0], keep)
        for p_ in pieces[1:]:
            again.extend(p_)
        for k, bn in enumerate(m.batches()):
            again.add(bn, ixs[int(bn[1:])], ex[ex["slot"] == k], slot=k, nb_best_hits=-1, piece=-1)
        assert again.emit() == want
    with pytest.raises(pm.PMError):
        m.add("b0", ixs[0], recs[0][:0], piece=99)


def test_native_fix_query_and_the_mirror_agree_on_random_ill_formed_input():
    """pm_queries_parse_raw(normalise = 1) and phylign_amd/fix_query.py both restate kseq_read() step by step; on random
    FASTA / FASTQ with the things real files do -- CRLF, empty lines, comments after blanks or TABs, junk ahead of the
    first record, '@' and '>' as quality values, a header byte in the middle of a line, quality blocks that are too
    short or too long (the reader stops there, like seqtk), missing final newlines -- they deliver the same records
    (the parser drops records without sequence, which cobs ignores, and refuses sequences shorter than k)"""
    import io
    from phylign_amd import _lib as pm
    from phylign_amd import fix_query as FQ
    rng = np.random.default_rng(33)
    alpha = list(b"ACGTacgtNnRYKMSWryu-.*")
    agree = 0
    for it in range(700):
        parts = []
        if rng.random() < 0.2:
            parts.append(bytes(rng.choice(list(b"ACGT\n >@+x"), size=int(rng.integers(0, 30))).astype(np.uint8)))
        for i in range(int(rng.integers(0, 8))):
            n = int(rng.integers(0, 120)) if rng.random() < 0.2 else int(rng.integers(31, 200))
            seq = bytes(rng.choice(alpha, size=n).astype(np.uint8))
            width = int(rng.integers(10, 90))
            nl = b"\r\n" if rng.random() < 0.15 else b"\n"
            lines = nl.join(seq[j:j + width] for j in range(0, n, width)) if n else b""
            if rng.random() < 0.1:
                lines = lines.replace(nl, nl + nl, 1)
            hdr = [b"r%d" % i, b"r%d comment x" % i, b"r%d\tdesc" % i, b"", b" lead", b"r%d " % i, b"r%d\x0bv" % i][int(rng.integers(0, 7))]
            if rng.random() < 0.4:
                ql = n if rng.random() < 0.7 else max(0, n + int(rng.integers(-5, 6)))
                qual = bytes(rng.integers(33, 75, size=ql).astype(np.uint8))
                if rng.random() < 0.3 and ql > 2:
                    qual = b"@" + qual[1:]
                if rng.random() < 0.2 and ql > 2:
                    qual = b">" + qual[1:]
                qlines = nl.join(qual[j:j + width] for j in range(0, ql, width))
                parts.append(b"@" + hdr + nl + lines + nl + b"+" + (hdr if rng.random() < 0.3 else b"") + nl + qlines + (nl if rng.random() < 0.9 else b""))
            else:
                parts.append(b">" + hdr + nl + lines + (nl if rng.random() < 0.9 else b""))
        raw = b"".join(parts)
        out = io.BytesIO()
        FQ.fix_stream(io.BytesIO(raw), out)
        recs = out.getvalue().split(b"\n")
        pairs = [(recs[i], recs[i + 1]) for i in range(0, len(recs) - 1, 2)]
        want = b"".join(h + b"\n" + s_ + b"\n" for h, s_ in pairs if s_)           # records without sequence are dropped
        if any(0 < len(s_) < 31 for _, s_ in pairs):
            with pytest.raises(pm.PMError):
                pm.Queries(raw, term_size=31, normalise=True)
            continue
        assert pm.Queries(raw, term_size=31, normalise=True).fasta() == want, (it, raw[:300])
        agree += 1
    assert agree > 400
    # kseq's jump to the next header byte does not care where in a line it stands
    raw = b"junk ACGT>r1 c\n" + b"ACGT" * 10 + b"\n"
    assert pm.Queries(raw, term_size=31, normalise=True).fasta() == b">r1\n" + b"ACGT" * 10 + b"\n"


def test_merge_is_extended_while_other_threads_add(tmp_path):
    """what the stage does: one thread parses the next pieces of the query file and extends the merge while workers
    add the records of pieces that are already there (ctypes calls run without the GIL; run under ThreadSanitizer by
    tools/asan_cpu_tests.sh thread) -- the result is the one-piece merge's"""
    import threading
    from phylign_amd import _lib as pm
    from phylign_amd.match_stage import split_prepared_fasta
    rng = np.random.default_rng(55)
    nq = 70000
    fasta = "".join(f">r{i % 60000}\n{'ACGT' * 8}A\n" for i in range(nq)).encode()       # 10 000 names repeat far away
    ix = pm.Index.from_names([f"{d:03x}_R{d}" for d in range(64)])
    recs = []
    for b in range(4):
        hq = np.sort(rng.choice(nq, size=20000, replace=False)).astype(np.uint32)
        r = np.zeros(len(hq), dtype=pm.HIT_DTYPE)
        r["query"] = hq; r["doc"] = rng.integers(0, 64, len(r)); r["score"] = rng.integers(1, 9, len(r))
        recs.append(pm.sort_hits(r))
    whole = pm.Merge(pm.Queries(fasta), 3)
    for b in range(4):
        whole.add(f"b{b}", ix, recs[b], slot=0)
    want = whole.emit()
    pieces = [pm.Queries(p) for p in split_prepared_fasta(fasta, 9000)]
    base = np.cumsum([0] + [p.count()[0] for p in pieces])
    m = pm.Merge(pieces[0], 3)
    ready = [threading.Event() for _ in pieces]
    ready[0].set()
    errs = []

    def extender():
        for ci in range(1, len(pieces)):
            m.extend(pieces[ci])
            ready[ci].set()

    def adder(b):
        try:
            for ci in range(len(pieces)):
                ready[ci].wait()
                part = recs[b][(recs[b]["query"] >= base[ci]) & (recs[b]["query"] < base[ci + 1])].copy()
                part["query"] -= np.uint32(base[ci])
                m.add(f"b{b}", ix, part, slot=0, piece=ci)
        except Exception as e:                              # noqa: BLE001
            errs.append(e)
    ts = [threading.Thread(target=extender)] + [threading.Thread(target=adder, args=(b,)) for b in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs and m.emit() == want


def test_exceptions_on_pool_threads_become_error_codes(tmp_path):
    """tests/native/pool_exceptions.cpp linked against the library's objects: a std::bad_alloc on a worker thread of
    the library's pool reaches the entry point's PM_GUARD_END (PM_ENOMEM) instead of std::terminate"""
    import subprocess
    from phylign_amd import build as B
    B.build()
    objs = [os.path.join(B.CSRC, os.path.splitext(s_)[0] + ".o") for s_ in B.SOURCES]
    exe = str(tmp_path / "pool_exceptions")
    subprocess.run([B._hipcc(), "--offload-arch=gfx950", "-O1", "-std=c++17", "-x", "hip", os.path.join(ROOT, "tests", "native", "pool_exceptions.cpp"),
                    "-x", "none"] + objs + ["-lz", "-o", exe], check=True, capture_output=True)
    r = subprocess.run([exe], capture_output=True, timeout=120)
    assert r.returncode == 0 and b"pool exceptions ok" in r.stdout, (r.returncode, r.stderr.decode()[-500:])


@pytest.mark.parametrize("threshold", [0.7, 0.0, 1.0, 0.35])
def test_oracle_equals_the_independent_python_restatement(oracle, threshold):
    """whole texts: the C oracle against tests/independent.py -- python-xxhash, its own header writer (upstream's field
    order), canonicalisation, scorer, ceil rule, order and grammar, written from SURVEY.md appendix A without looking at
    oracle/ -- on genome-like indexes (24 ... 1 030 documents, k = 21 / 31, one and two hash functions, canonical and
    not, reads in both orientations with 0 - 6 % errors, lengths k ... 151)"""
    pytest.importorskip("xxhash")
    import independent as I
    for case in I.CASES:
        seed, D, glen, k, num_hashes, canon = case
        index, m, fasta, names, S, records = I.built_case(*case)
        h = oracle.header_parse(index)
        assert (h.term_size, h.n_docs, h.signature_size, h.num_hashes) == (k, D, S, num_hashes)
        assert oracle.query_file(index, fasta, threshold) == I.query_text(records, names, m, k, num_hashes, S, threshold, canon), case


def test_oracle_equals_the_independent_restatement_on_random_small_cases(oracle):
    """the same comparison over 60 random shapes: k = 4 ... 71, 1 - 3 hash functions, canonical or not, 1 ... 200 documents,
    reads of k ... k + 40 bases, thresholds 0 ... 1 -- whole texts"""
    pytest.importorskip("xxhash")
    import independent as I
    rng = np.random.default_rng(2024)
    for it in range(60):
        k = int(rng.choice([4, 5, 8, 15, 16, 21, 31, 32, 33, 47, 64, 71]))
        nh, canon, D = int(rng.integers(1, 4)), bool(rng.integers(0, 2)), int(rng.integers(1, 201))
        glen = int(rng.integers(k + 5, 400))
        genomes = ["".join("ACGT"[c] for c in rng.integers(0, 4, size=glen)) for _ in range(min(D, 12))]
        genomes = [genomes[d % len(genomes)] for d in range(D)]                 # documents repeat: equal scores, tie groups
        names = [f"{d:04x}_doc{d}" for d in range(D)]
        S = int(rng.integers(7, 4 * glen))
        index, m = I.classic_index(names, genomes, k, nh, S, canon)
        records = []
        for i in range(6):
            g = genomes[int(rng.integers(0, len(genomes)))]
            L = int(rng.integers(k, min(k + 41, glen) + 1))
            p = int(rng.integers(0, glen - L + 1))
            s = list(g[p:p + L])
            for j in range(L):
                if rng.random() < 0.05 * (i % 3):
                    s[j] = "ACGT"[int(rng.integers(0, 4))]
            records.append((f"r{i} len={L}", "".join(s)))
        fasta = "".join(f">{h}\n{s}\n" for h, s in records).encode()
        thr = float(rng.choice([0.0, 0.2, 0.5, 0.7, 0.9, 1.0]))
        assert oracle.query_file(index, fasta, thr) == I.query_text(records, names, m, k, nh, S, thr, canon), (it, k, nh, canon, D, S, thr)


@pytest.mark.parametrize("threshold", [0.7, 0.0, 0.4])
def test_oracle_equals_the_independent_restatement_on_compact_indexes(oracle, threshold):
    """.cobs_compact written and scored by tests/independent.py: header with per-sub-index
    (signature_size, num_hashes), padding to the page boundary, page-wide sub-matrices; the oracle reads the file and
    prints the same text (documents of the last sub-index partly filled, sub-indexes with 1 - 3 hash functions)"""
    pytest.importorskip("xxhash")
    import independent as I
    for seed, page, D, params, k in I.COMPACT_CASES:
        index, mats, fasta, names, records = I.built_compact_case(seed, page, D, tuple(params), k)
        c = oracle.compact_parse(index)
        assert (c.n_docs, c.page_size) == (D, page)
        assert oracle.query_file(index, fasta, threshold) == I.query_text_compact(records, names, mats, k, page, params, threshold), (seed, threshold)


def test_block_parallel_xz_decoder_equals_xzcat(tmp_path):
    """phylign_amd/xzpar.py: a multi-block .xz (`xz -T`, any integrity check) decoded block by block on several threads gives
    the bytes xzcat gives; one-block files (plain `xz`, python's lzma, the reference's own data/*.xz), several streams and
    non-xz input have no plan (the caller falls back to xzcat); a damaged block ends the stream early with status 1"""
    import lzma
    import subprocess
    from phylign_amd import xzpar
    rng = np.random.default_rng(12)
    data = np.packbits(rng.random(24_000_000) < 0.25).tobytes() + bytes(rng.integers(0, 4, 1_500_001, dtype=np.uint8)) + b"tail"
    src = tmp_path / "m.bin"
    src.write_bytes(data)
    for check in ("crc64", "crc32", "none", "sha256"):
        out = tmp_path / f"m_{check}.xz"
        with open(out, "wb") as f:
            subprocess.run(["xz", "-T3", "-0", "--block-size=512KiB", f"--check={check}", "-c", str(src)], stdout=f, check=True)
        pl = xzpar.plan(str(out))
        assert pl is not None and len(pl.blocks) == (len(data) + (512 << 10) - 1) // (512 << 10) and pl.uncompressed_size == len(data)
        for threads in (1, 5):
            p = xzpar.ParallelXz(pl, threads)
            assert p.stdout.read() == data and p.wait() == 0
    one = tmp_path / "one.xz"
    one.write_bytes(lzma.compress(data[:300000]))
    assert xzpar.plan(str(one)) is None
    two = tmp_path / "two.xz"
    two.write_bytes((tmp_path / "m_crc64.xz").read_bytes() * 2)
    assert xzpar.plan(str(two)) is None and xzpar.plan(str(src)) is None and xzpar.plan(str(tmp_path / "missing.xz")) is None
    blob = bytearray((tmp_path / "m_crc64.xz").read_bytes())
    blob[len(blob) // 3] ^= 0x55
    bad = tmp_path / "bad.xz"
    bad.write_bytes(bytes(blob))
    pl = xzpar.plan(str(bad))
    p = xzpar.ParallelXz(pl, 4)
    got = p.stdout.read()
    assert p.wait() == 1 and len(got) < len(data) and data.startswith(got) and p.error is not None
    # a reader that goes away early does not hang the decoder
    p = xzpar.ParallelXz(xzpar.plan(str(tmp_path / "m_crc64.xz")), 2)
    p.stdout.read(1000)
    p.stdout.close()
    assert p.wait() == 1


def test_block_parallel_xz_plan_refuses_what_only_xzcat_can_decode(tmp_path):
    """a VALID multi-block .xz whose blocks are not plain LZMA2 (`--delta` in front of it) has a perfectly good
    index; plan() reads every block header and returns None, so the caller streams the file through xzcat instead of
    aborting the load.  A path that is not a regular file (a FIFO named *.xz) is never opened by plan()."""
    import subprocess
    import threading
    from phylign_amd import match_stage as MS
    from phylign_amd import xzpar
    rng = np.random.default_rng(13)
    data = bytes(rng.integers(0, 7, 3_000_001, dtype=np.uint8))
    src = tmp_path / "d.bin"
    src.write_bytes(data)
    delta, plain = tmp_path / "delta.xz", tmp_path / "plain.xz"
    with open(delta, "wb") as f:
        subprocess.run(["xz", "-T2", "--block-size=1MiB", "--delta=dist=1", "--lzma2=preset=0", "-c", str(src)], stdout=f, check=True)
    with open(plain, "wb") as f:
        subprocess.run(["xz", "-T2", "--block-size=1MiB", "-0", "-c", str(src)], stdout=f, check=True)
    assert subprocess.run(["xzcat", str(delta)], capture_output=True, check=True).stdout == data      # a valid file ...
    assert xzpar.plan(str(delta)) is None                                                                # ... that only xzcat decodes
    pl = xzpar.plan(str(plain))
    assert pl is not None and len(pl.blocks) == 3
    # the stage's stream opener falls back to xzcat for it and delivers every byte
    cobs = tmp_path / "cobs"
    cobs.mkdir()
    os.link(delta, cobs / "bdelta.cobs_classic.xz")
    os.link(plain, cobs / "bplain.cobs_classic.xz")
    for batch, kind in (("bdelta", "Popen"), ("bplain", "ParallelXz")):
        f, dec = MS.open_index_stream(str(cobs), batch, xz_threads=4)
        assert type(dec).__name__ == kind
        assert f.read() == data and dec.wait() == 0
        f.close()
    # one block header damaged in an otherwise well-formed file: no plan (xzcat then reports the damage itself)
    blob = bytearray(plain.read_bytes())
    blob[pl.blocks[1].offset + 1] ^= 0x03                     # block flags: claims more filters than the header holds
    (tmp_path / "hdr.xz").write_bytes(bytes(blob))
    assert xzpar.plan(str(tmp_path / "hdr.xz")) is None
    # a FIFO: plan() must not read from it (it would eat the stream header in front of the real decoder)
    fifo = tmp_path / "pipe.xz"
    os.mkfifo(fifo)
    t = threading.Thread(target=lambda: open(fifo, "wb").write(plain.read_bytes()), daemon=True)
    assert xzpar.plan(str(fifo)) is None                      # returns at once: nothing was opened, no writer needed
    t.start()
    assert subprocess.run(["xzcat", str(fifo)], capture_output=True, check=True).stdout == data
    t.join(10)
    # blocks larger than one pread delivers are read in pieces
    fd = os.open(str(src), os.O_RDONLY)
    try:
        assert xzpar._pread_all(fd, len(data), 0) == data and xzpar._pread_all(fd, 10, len(data) - 10) == data[-10:]
        with pytest.raises(ValueError):
            xzpar._pread_all(fd, 11, len(data) - 10)
    finally:
        os.close(fd)

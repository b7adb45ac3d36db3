"""Drop-in boundary on the GPU: the reference's command lines
(scripts/run_cobs_streaming.sh:13-29, Snakefile:463-469) against the oracle."""
import gzip
import io
import lzma
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import build_case, rand_seq

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _case(oracle, seed=21, n_docs=664, S=30000, nq=24):
    rng = np.random.default_rng(seed)
    queries = [(f"{i}A some comment" if i % 2 else f"{i}B", rand_seq(rng, 150)) for i in range(nq)]
    plant = []
    for qi in range(nq):
        for j, frac in enumerate((1.0, 0.9, 0.9, 0.9, 0.8, 0.75, 0.75, 0.7, 0.6)):
            plant.append((qi, (qi * 17 + j * 5) % n_docs, frac))
    return build_case(oracle, rng, n_docs, S, queries, plant=plant)


def test_run_cobs_streaming_script(pm, oracle, tmp_path):
    index, fasta, _ = _case(oracle)
    xz = tmp_path / "bacillus_anthracis__01.cobs_classic.xz"
    xz.write_bytes(lzma.compress(bytes(index), preset=1))
    fa = tmp_path / "q.fa"
    fa.write_bytes(fasta)
    script = os.path.join(ROOT, "scripts", "run_cobs_streaming.sh")
    r = subprocess.run([script, "0.7", "4", str(xz), str(len(index)), str(fa)], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    exp = oracle.query_file(index, fasta, 0.7)
    assert r.stdout == exp
    # wrong argument count -> usage + exit 1 (scripts/run_cobs_streaming.sh:13-16)
    r2 = subprocess.run([script, "0.7", "4"], capture_output=True)
    assert r2.returncode == 1 and b"usage" in r2.stderr
    # truncated index stream -> non-zero exit, nothing that looks like a complete result
    bad = tmp_path / "bad.cobs_classic.xz"
    bad.write_bytes(lzma.compress(bytes(index[: len(index) // 2]), preset=1))
    r3 = subprocess.run([script, "0.7", "4", str(bad), str(len(index)), str(fa)], capture_output=True)
    assert r3.returncode != 0 and r3.stdout == b""


@pytest.mark.parametrize("n", [0, 1, 2, 3, 5, 100])
def test_rule_pipeline_and_fused_postprocess(pm, oracle, tmp_path, n):
    """`run_cobs_streaming.sh ... | postprocess_cobs.py -n N | gzip --fast` (Snakefile:466-469)"""
    from phylign_amd import postprocess as P
    index, fasta, _ = _case(oracle, seed=22)
    xz = tmp_path / "b__01.cobs_classic.xz"
    xz.write_bytes(lzma.compress(bytes(index), preset=1))
    fa = tmp_path / "q.fa"
    fa.write_bytes(fasta)
    out = tmp_path / "b__01____q.gz"
    sdir = os.path.join(ROOT, "scripts")
    cmd = (f"set -euo pipefail; {sdir}/run_cobs_streaming.sh 0.7 2 '{xz}' {len(index)} '{fa}' "
           f"| {sdir}/postprocess_cobs.py -n {n} | gzip --fast > '{out}'")
    r = subprocess.run(["bash", "-c", cmd], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    exp = P.filter_text(oracle.query_file(index, fasta, 0.7).decode(), n)
    assert gzip.open(out, "rt").read() == exp
    # fused in-library post-filter gives the same bytes
    ix = pm.Index.load_mem(index)
    assert pm.query_text(ix, fasta, 0.7, nb_best_hits=n).decode() == exp
    env = dict(os.environ, PHYLIGN_NB_BEST_HITS=str(n))
    r = subprocess.run([f"{sdir}/run_cobs_streaming.sh", "0.7", "2", str(xz), str(len(index)), str(fa)],
                       capture_output=True, env=env)
    assert r.returncode == 0 and r.stdout.decode() == exp


def test_cobs_query_cli_plain_file(pm, oracle, tmp_path):
    """non-streaming form: `cobs query --load-complete -t T -T n -i index -f q.fa` (Snakefile:419-424)"""
    index, fasta, _ = _case(oracle, seed=23, n_docs=195, S=9000)
    p = tmp_path / "i.cobs_classic"
    p.write_bytes(bytes(index))
    fa = tmp_path / "q.fa"
    fa.write_bytes(fasta)
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-m", "phylign_amd.cobs_query", "query", "--load-complete", "-t", "0.7", "-T", "8",
                        "-i", str(p), "-f", str(fa)], capture_output=True, env=env)
    assert r.returncode == 0, r.stderr.decode()
    assert r.stdout == oracle.query_file(index, fasta, 0.7)
    r = subprocess.run([sys.executable, "-m", "phylign_amd.cobs_query", "query", "-t", "0.7", "-i", str(tmp_path / "missing"),
                        "-f", str(fa)], capture_output=True, env=env)
    assert r.returncode == 1 and b"cannot open index" in r.stderr


@pytest.mark.parametrize("limit", [1, 3, 50])
def test_cobs_query_limit(pm, oracle, tmp_path, limit):
    """`cobs query -l N`: the N best results per query, header = lines printed (oracle: num_results)"""
    index, fasta, _ = _case(oracle, seed=24, n_docs=664, S=20000)
    p = tmp_path / "i.cobs_classic"
    p.write_bytes(bytes(index))
    fa = tmp_path / "q.fa"
    fa.write_bytes(fasta)
    env = dict(os.environ, PYTHONPATH=ROOT)
    for thr in ("0.7", "0.0"):
        r = subprocess.run([sys.executable, "-m", "phylign_amd.cobs_query", "query", "-t", thr, "-l", str(limit),
                            "-i", str(p), "-f", str(fa)], capture_output=True, env=env)
        assert r.returncode == 0, r.stderr.decode()
        assert r.stdout == oracle.query_file(index, fasta, float(thr), limit)
    ix = pm.Index.load_mem(index)
    q = pm.Queries(fasta)
    hits = pm.search([ix], q, 0.7).hits()                          # uncut records format the same
    assert pm.format_hits_limit(ix, q, hits, limit=limit) == oracle.query_file(index, fasta, 0.7, limit)
    assert pm.format_hits_limit(ix, q, hits, limit=0) == oracle.query_file(index, fasta, 0.7)


def test_match_to_filter_dropin(pm, oracle, tmp_path):
    """03_match files of three batches feed the 04_filter consumer (Snakefile:490-520)."""
    from phylign_amd import filter_queries as F
    from phylign_amd import postprocess as P
    rng = np.random.default_rng(31)
    queries = [(f"r{i}", rand_seq(rng, 120)) for i in range(12)]
    files, exp_files = [], []
    for b, (n_docs, S) in enumerate(((195, 5000), (176, 7000), (664, 6000))):
        plant = [(qi, (qi * 7 + j) % n_docs, fr) for qi in range(12) for j, fr in enumerate((1.0, 0.9, 0.8, 0.8, 0.7))]
        index, fasta, _ = build_case(oracle, rng, n_docs, S, queries, plant=plant)
        ix = pm.Index.load_mem(index)
        got = pm.query_text(ix, fasta, 0.7, nb_best_hits=3)
        exp = P.filter_text(oracle.query_file(index, fasta, 0.7).decode(), 3)
        assert got.decode() == exp
        fn = tmp_path / f"batch_{b}__01____q.gz"
        with gzip.open(fn, "wb") as f:
            f.write(got)
        files.append(str(fn))
    fa = tmp_path / "q.fa"
    fa.write_bytes(fasta)
    out = io.StringIO()
    F.filter_files(str(fa), files, 3, out)
    recs = out.getvalue().strip().split("\n")
    assert len(recs) == 24 and recs[0].startswith(">r0 ")
    assert all(len(recs[i].split(" ")[1].split(",")) >= 3 for i in range(0, 24, 2))


def _stage_fixture(oracle, tmp_path, n_batches=5, repeated_names=False):
    """a small cobs/ directory with xz indexes, a batches file, the sizes table and a query FASTA"""
    rng = np.random.default_rng(77)
    queries = [(f"q{i} c", rand_seq(rng, 150)) for i in range(20)]
    if repeated_names:                                  # the mates of read pairs in concatenated files: one query each
        for i, j in ((9, 2), (16, 2), (19, 11)):
            queries[i] = (queries[j][0], queries[i][1])
    cobs = tmp_path / "cobs"
    cobs.mkdir()
    shapes = [(195, 5000), (176, 9000), (664, 6000), (4000, 3000), (30, 4000)][:n_batches]
    names, indexes = [], {}
    with open(tmp_path / "sizes.txt", "w") as sz:
        for b, (n_docs, S) in enumerate(shapes):
            batch = f"genus_species{b}__01"
            plant = [(qi, (qi * 11 + j * 3) % n_docs, fr) for qi in range(20)
                     for j, fr in enumerate((1.0, 0.95, 0.9, 0.9, 0.8, 0.8, 0.8, 0.7))]
            index, fasta, _ = build_case(oracle, rng, n_docs, S, queries, plant=plant)
            (cobs / f"{batch}.cobs_classic.xz").write_bytes(lzma.compress(bytes(index), preset=1))
            sz.write(f"cobs/{batch}.cobs_classic.xz  {len(index)}  1610678320\n")
            names.append(batch)
            indexes[batch] = index
    (tmp_path / "batches.txt").write_text("\n".join(reversed(names)) + "\n")
    (tmp_path / "Q.fa").write_bytes(fasta)
    return names, indexes, fasta


def _check_stage_outputs(oracle, tmp_path, names, indexes, fasta, n):
    from phylign_amd import filter_queries as F
    from phylign_amd import postprocess as P
    files = []
    for b in names:
        exp = P.filter_text(oracle.query_file(indexes[b], fasta, 0.7).decode(), n)
        fn = tmp_path / "03_match" / f"{b}____Q.gz"
        assert gzip.open(fn, "rt").read() == exp, b
        files.append(str(fn))
    out = io.StringIO()
    F.filter_files(str(tmp_path / "Q.fa"), files, n, out)          # golden-pinned mirror of filter_queries.py
    assert (tmp_path / "04_filter" / "Q.fa").read_text() == out.getvalue()
    assert not list((tmp_path / "03_match").glob("*.tmp"))


def test_match_stage_filter_only_writes_the_same_fasta_and_no_match_files(pm, oracle, tmp_path):
    """--filter-only (opt-in): the records go from the GPU straight into the 04_filter merge, no per-batch .gz is written;
    the FASTA is byte for byte the default run's (which the other tests pin against the reference's filter_queries.py)"""
    names, indexes, fasta = _stage_fixture(oracle, tmp_path)
    env = dict(os.environ, PYTHONPATH=ROOT)
    base = [sys.executable, "-m", "phylign_amd.match_stage", "--batches", str(tmp_path / "batches.txt"), "--cobs-dir", str(tmp_path / "cobs"),
            "--sizes", str(tmp_path / "sizes.txt"), "--queries", str(tmp_path / "Q.fa"), "--nb-best-hits", "3"]
    r = subprocess.run(base + ["--out-dir", str(tmp_path / "03_match"), "--filter-out", str(tmp_path / "04_filter" / "Q.fa")], capture_output=True, env=env)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    _check_stage_outputs(oracle, tmp_path, names, indexes, fasta, 3)
    r = subprocess.run(base + ["--out-dir", str(tmp_path / "03_only"), "--filter-out", str(tmp_path / "04_only" / "Q.fa"), "--filter-only"],
                       capture_output=True, env=env)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert (tmp_path / "04_only" / "Q.fa").read_bytes() == (tmp_path / "04_filter" / "Q.fa").read_bytes()
    assert not list((tmp_path / "03_only").glob("*"))
    r = subprocess.run(base + ["--out-dir", str(tmp_path / "03_x"), "--filter-only"], capture_output=True, env=env)
    assert r.returncode == 2 and b"--filter-only needs --filter-out" in r.stderr


@pytest.mark.parametrize("n", [3, 100])
def test_match_stage_end_to_end_single_rank(pm, oracle, tmp_path, n):
    names, indexes, fasta = _stage_fixture(oracle, tmp_path)
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-m", "phylign_amd.match_stage", "--batches", str(tmp_path / "batches.txt"),
                        "--cobs-dir", str(tmp_path / "cobs"), "--sizes", str(tmp_path / "sizes.txt"),
                        "--queries", str(tmp_path / "Q.fa"), "--out-dir", str(tmp_path / "03_match"),
                        "--nb-best-hits", str(n), "--filter-out", str(tmp_path / "04_filter" / "Q.fa"), "--loaders", "3"],
                       capture_output=True, env=env)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    _check_stage_outputs(oracle, tmp_path, names, indexes, fasta, n)


def test_match_stage_searches_resident_batches_with_fused_launches(pm, oracle, tmp_path):
    """the stage runs the path bench.py measures: every resident batch of a group goes into ONE
    pm_search_async, so a group costs one scan launch per (row-width class, counter-width class),
    not one per batch (the reference: 305 separate `cobs query` jobs, Snakefile:431-487) -- and the
    files and the 04_filter FASTA stay byte-identical"""
    import json
    from phylign_amd import match_stage as MS
    names, indexes, fasta = _stage_fixture(oracle, tmp_path)
    batches = sorted(names)
    src = MS.ResidentSource({b: pm.Index.load_mem(indexes[b]) for b in batches})
    q = pm.Queries(fasta)
    report, merge = MS.run_stage(pm, batches, list(range(len(batches))), src, q, "Q", str(tmp_path / "03_match"),
                                 0.7, 3, want_merge=True)
    assert report["groups"] == 1 and report["per_group"][0]["batches"] == batches
    # 5 batches: four narrow ones share the mixed-width launch, the 4000-document one has its own
    assert report["scan_launches"] == 2 < len(batches)
    assert sorted(report["merge_order"]) == batches
    (tmp_path / "04_filter").mkdir()
    (tmp_path / "04_filter" / "Q.fa").write_bytes(merge.emit())
    _check_stage_outputs(oracle, tmp_path, names, indexes, fasta, 3)
    for k in ("match_only_s", "format_s_thread_sum", "gzip_s_thread_sum", "merge_s_thread_sum", "stage_wall_s"):
        assert report[k] >= 0
    # groups of at most two batches: three groups, still one or two launches each, same bytes
    for f in (tmp_path / "03_match").glob("*.gz"):
        f.unlink()
    report2, merge2 = MS.run_stage(pm, batches, list(range(len(batches))), src, q, "Q", str(tmp_path / "03_match"),
                                   0.7, 3, want_merge=True, max_group=2)
    assert report2["groups"] == 3 and all(g["scan_launches"] <= 2 for g in report2["per_group"])
    (tmp_path / "04_filter" / "Q.fa").write_bytes(merge2.emit())
    _check_stage_outputs(oracle, tmp_path, names, indexes, fasta, 3)
    json.dumps(report2)


@pytest.mark.parametrize("ranks,repeated", [(1, False), (2, False), (2, True)])
def test_match_stage_searches_a_large_query_file_in_chunks(pm, oracle, tmp_path, ranks, repeated):
    """--query-chunk: 20 reads in chunks of 7 (3 chunks) against batches that stream through a tiny HBM budget -- every
    batch's file is written piece by piece, the ONE 04_filter merge grows chunk by chunk, and files and FASTA equal the
    one-piece run's (one rank, and two ranks with the gather over gloo; also with read names that repeat across the
    chunks: one query each, as in the consumer's dict)"""
    import json
    names, indexes, fasta = _stage_fixture(oracle, tmp_path, repeated_names=repeated)
    env = dict(os.environ, PYTHONPATH=ROOT)
    cmd = [sys.executable]
    if ranks == 2:
        env.update(PHYLIGN_DIST_BACKEND="gloo", PHYLIGN_SHARE_GPU="1")
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", "29543"]
    cmd += ["-m", "phylign_amd.match_stage", "--batches", str(tmp_path / "batches.txt"), "--cobs-dir", str(tmp_path / "cobs"),
            "--sizes", str(tmp_path / "sizes.txt"), "--queries", str(tmp_path / "Q.fa"), "--out-dir", str(tmp_path / "03_match"),
            "--nb-best-hits", "3", "--filter-out", str(tmp_path / "04_filter" / "Q.fa"), "--loaders", "2",
            "--query-chunk", "7", "--max-resident-gb", "0.0000001"]
    r = subprocess.run(cmd, capture_output=True, env=env)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    _check_stage_outputs(oracle, tmp_path, names, indexes, fasta, 3)
    assert r.stderr.count(b'"query_chunks": 3') == ranks          # every rank reports its three chunks
    assert r.stderr.count(b'"query_hbm_bytes_at_end": 0') == ranks   # ... and none of them is still resident
    assert not list((tmp_path / "03_match").glob("*.tmp"))


def test_match_stage_takes_unprepared_queries(pm, oracle, tmp_path):
    """--raw-queries: rule fix_query (Snakefile:314-333) runs inside the native parser; same outputs as
    for the prepared file"""
    names, indexes, fasta = _stage_fixture(oracle, tmp_path, n_batches=2)
    recs = fasta.decode().strip().split("\n")
    raw = "".join(f"{h} extra words\n{s[:70].lower()}\n{s[70:]}\n" for h, s in zip(recs[0::2], recs[1::2]))
    (tmp_path / "raw.fa").write_text(raw)
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-m", "phylign_amd.match_stage", "--batches", str(tmp_path / "batches.txt"),
                        "--cobs-dir", str(tmp_path / "cobs"), "--sizes", str(tmp_path / "sizes.txt"),
                        "--queries", str(tmp_path / "raw.fa"), "--raw-queries", "--out-dir", str(tmp_path / "03_match"),
                        "--nb-best-hits", "3"], capture_output=True, env=env)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    from phylign_amd import postprocess as P
    # header comments are dropped by the rule (-C), so the prepared file has bare names
    prepared = "".join(f"{h.split(' ')[0]}\n{s}\n" for h, s in zip(recs[0::2], recs[1::2])).encode()
    for b in names:
        exp = P.filter_text(oracle.query_file(indexes[b], prepared, 0.7).decode(), 3)
        assert gzip.open(tmp_path / "03_match" / f"{b}____raw.gz", "rt").read() == exp, b


def test_match_stage_takes_the_reference_input_directory(pm, oracle, tmp_path):
    """--input-dir: rules fix_query + concatenate_queries (Snakefile:314-352) for the reference's own four bundled query
    files (FASTQ and FASTA, tests/golden/reads/raw/): each is prepared by the native parser, the prepared texts follow
    each other in the order of the sorted names, and every output is called after reads_1___reads_2___reads_3___reads_4
    (Snakefile:28-38) -- the same bytes as for the merged file the reference's pipeline made of them (tests/golden/reads/),
    and as for the four files given one by one in another order"""
    names, indexes, _ = _stage_fixture(oracle, tmp_path, n_batches=2)
    gold = os.path.join(ROOT, "tests", "golden", "reads")
    merged_name = "reads_1___reads_2___reads_3___reads_4"
    merged = open(os.path.join(gold, merged_name + ".fa"), "rb").read()
    env = dict(os.environ, PYTHONPATH=ROOT)
    base = [sys.executable, "-m", "phylign_amd.match_stage", "--batches", str(tmp_path / "batches.txt"),
            "--cobs-dir", str(tmp_path / "cobs"), "--sizes", str(tmp_path / "sizes.txt"), "--nb-best-hits", "3"]
    raw = [os.path.join(gold, "raw", f) for f in ("reads_3.fasta", "reads_1.fastq", "reads_4.fa", "reads_2.fq")]
    runs = {"dir": ["--input-dir", os.path.join(gold, "raw")], "files": ["--raw-queries", "--queries"] + raw,
            "merged": ["--queries", os.path.join(gold, merged_name + ".fa")]}
    for tag, extra in runs.items():
        r = subprocess.run(base + extra + ["--out-dir", str(tmp_path / f"03_{tag}"), "--filter-out", str(tmp_path / f"04_{tag}" / "out.fa")],
                           capture_output=True, env=env)
        assert r.returncode == 0, (tag, r.stderr.decode()[-2000:])
    from phylign_amd import postprocess as P
    for b in names:
        exp = P.filter_text(oracle.query_file(indexes[b], merged, 0.7).decode(), 3)
        assert exp.count("*") == 40
        for tag in runs:
            assert gzip.open(tmp_path / f"03_{tag}" / f"{b}____{merged_name}.gz", "rt").read() == exp, (b, tag)
    fa = {tag: (tmp_path / f"04_{tag}" / "out.fa").read_bytes() for tag in runs}
    assert fa["dir"] == fa["files"] == fa["merged"] and fa["dir"].count(b">") == 40
    # two files with one name are refused like the reference's get_query_file asserts (Snakefile:309-311)
    (tmp_path / "in2").mkdir()
    for ext in ("fa", "fq"):
        (tmp_path / "in2" / f"same.{ext}").write_bytes(open(raw[0], "rb").read())
    r = subprocess.run(base + ["--input-dir", str(tmp_path / "in2"), "--out-dir", str(tmp_path / "03_x")], capture_output=True, env=env)
    assert r.returncode != 0 and b"two query files" in r.stderr


def test_match_stage_runs_the_match_target_from_the_reference_config(pm, oracle, tmp_path):
    """--config: a directory laid out like the reference's (config.yaml with its keys, data/, cobs/, input/) and ONE
    command in place of `snakemake match` (Snakefile:249-253): queries from input/, batches / threshold / nb_best_hits /
    load mode from the config, outputs in intermediate/03_match and intermediate/04_filter/<merged name>.fa; mem-disk with
    keep_cobs_indexes leaves the decompressed indexes in intermediate/02_cobs_decompressed, mem-stream leaves none"""
    import shutil
    names, indexes, _ = _stage_fixture(oracle, tmp_path, n_batches=2)
    gold = os.path.join(ROOT, "tests", "golden", "reads")
    merged_name = "reads_1___reads_2___reads_3___reads_4"
    merged = open(os.path.join(gold, merged_name + ".fa"), "rb").read()
    from phylign_amd import postprocess as P
    for mode, keep in (("mem-disk", True), ("mem-stream", True), ("mem-disk", False)):
        wd = tmp_path / f"phylign_{mode}_{keep}"
        (wd / "data").mkdir(parents=True)
        shutil.copytree(tmp_path / "cobs", wd / "cobs")
        shutil.copytree(os.path.join(gold, "raw"), wd / "input")
        shutil.copy(tmp_path / "batches.txt", wd / "data" / "batches_small.txt")
        shutil.copy(tmp_path / "sizes.txt", wd / "data" / "decompressed_indexes_sizes.txt")
        (wd / "config.yaml").write_text(
            "batches: \"data/batches_small.txt\"\ncobs_kmer_thres: 0.7\nnb_best_hits: 3\nminimap_preset: \"sr\"\nthreads: all\n"
            f"max_ram_gb: 12\ndownload_dir: \".\"\ncobs_threads: auto\nindex_load_mode: {mode}\nkeep_cobs_indexes: {keep}\n")
        r = subprocess.run([sys.executable, "-m", "phylign_amd.match_stage", "--config", str(wd / "config.yaml")],
                           capture_output=True, env=dict(os.environ, PYTHONPATH=ROOT), cwd=str(tmp_path))
        assert r.returncode == 0, (mode, r.stderr.decode()[-2000:])
        for b in names:
            exp = P.filter_text(oracle.query_file(indexes[b], merged, 0.7).decode(), 3)
            assert gzip.open(wd / "intermediate" / "03_match" / f"{b}____{merged_name}.gz", "rt").read() == exp, (b, mode)
        fa = (wd / "intermediate" / "04_filter" / f"{merged_name}.fa").read_bytes()
        assert fa.count(b">") == 40
        kept = sorted(p_.name for p_ in (wd / "intermediate" / "02_cobs_decompressed").glob("*.cobs_classic")) \
            if (wd / "intermediate" / "02_cobs_decompressed").exists() else []
        assert kept == ([f"{b}.cobs_classic" for b in sorted(names)] if (mode == "mem-disk" and keep) else []), (mode, keep, kept)
        if kept:
            assert all((wd / "intermediate" / "02_cobs_decompressed" / f"{b}.cobs_classic").read_bytes() == bytes(indexes[b]) for b in names)
    # a flag on the command line wins over the config
    r = subprocess.run([sys.executable, "-m", "phylign_amd.match_stage", "--config", str(wd / "config.yaml"), "--nb-best-hits", "1",
                        "--out-dir", str(tmp_path / "03_flag")], capture_output=True, env=dict(os.environ, PYTHONPATH=ROOT))
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    b = names[0]
    assert gzip.open(tmp_path / "03_flag" / f"{b}____{merged_name}.gz", "rt").read() == \
        P.filter_text(oracle.query_file(indexes[b], merged, 0.7).decode(), 1)


def test_match_stage_fails_fast_on_a_broken_index(pm, oracle, tmp_path):
    """a truncated index stream in the middle of the batch list, loaders queued behind a tiny HBM
    budget: the stage exits non-zero promptly (waiting loaders are told to give up) and leaves no
    output for the broken batch and no .tmp file"""
    names, indexes, fasta = _stage_fixture(oracle, tmp_path)
    broken = sorted(names)[2]
    (tmp_path / "cobs" / f"{broken}.cobs_classic.xz").write_bytes(
        lzma.compress(bytes(indexes[broken][: len(indexes[broken]) // 3]), preset=1))
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-m", "phylign_amd.match_stage", "--batches", str(tmp_path / "batches.txt"),
                        "--cobs-dir", str(tmp_path / "cobs"), "--sizes", str(tmp_path / "sizes.txt"),
                        "--queries", str(tmp_path / "Q.fa"), "--out-dir", str(tmp_path / "03_match"),
                        "--filter-out", str(tmp_path / "04_filter" / "Q.fa"), "--loaders", "3",
                        "--max-resident-gb", "0.0000001"], capture_output=True, env=env, timeout=300)
    assert r.returncode != 0
    assert b"index stream ended" in r.stderr or b"xz decoding failed" in r.stderr, r.stderr.decode()[-1500:]
    assert not (tmp_path / "03_match" / f"{broken}____Q.gz").exists()
    assert not list((tmp_path / "03_match").glob("*.tmp")) and not (tmp_path / "04_filter" / "Q.fa").exists()


def test_match_stage_two_ranks_gather(pm, oracle, tmp_path):
    """the N>1 code path (static sharding, gather of hit records + names to rank 0) with two
    ranks sharing the one GPU over gloo; on a multi-GPU node the same code runs over RCCL"""
    names, indexes, fasta = _stage_fixture(oracle, tmp_path)
    env = dict(os.environ, PYTHONPATH=ROOT, PHYLIGN_DIST_BACKEND="gloo", PHYLIGN_SHARE_GPU="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29541", "-m", "phylign_amd.match_stage",
                        "--batches", str(tmp_path / "batches.txt"), "--cobs-dir", str(tmp_path / "cobs"),
                        "--sizes", str(tmp_path / "sizes.txt"), "--queries", str(tmp_path / "Q.fa"),
                        "--out-dir", str(tmp_path / "03_match"), "--nb-best-hits", "3",
                        "--filter-out", str(tmp_path / "04_filter" / "Q.fa")], capture_output=True, env=env)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    _check_stage_outputs(oracle, tmp_path, names, indexes, fasta, 3)


def test_resident_index_server(pm, oracle, tmp_path):
    """persistent residency: the second job on the same index skips decode + upload"""
    import time
    from phylign_amd.server import request
    index, fasta, _ = _case(oracle, seed=41, n_docs=664, S=400000)
    xz = tmp_path / "big__01.cobs_classic.xz"
    xz.write_bytes(lzma.compress(bytes(index), preset=0))
    fa = tmp_path / "q.fa"
    fa.write_bytes(fasta)
    sock = str(tmp_path / "pm.sock")
    env = dict(os.environ, PYTHONPATH=ROOT)
    srv = subprocess.Popen([sys.executable, "-m", "phylign_amd.server", "--socket", sock], env=env, stderr=subprocess.PIPE)
    try:
        for _ in range(600):
            if os.path.exists(sock):
                break
            time.sleep(0.1)
        assert os.path.exists(sock), "server did not come up"
        exp = oracle.query_file(index, fasta, 0.7)
        script = os.path.join(ROOT, "scripts", "run_cobs_streaming.sh")
        cenv = dict(env, PHYLIGN_MATCH_SERVER=sock)
        outs = []
        for _ in range(2):
            r = subprocess.run([script, "0.7", "1", str(xz), str(len(index)), str(fa)], capture_output=True, env=cenv)
            assert r.returncode == 0, r.stderr.decode()
            outs.append(r.stdout)
        assert outs[0] == exp and outs[1] == exp
        h1, b1 = request(sock, {"op": "query", "index": str(xz), "index_size": len(index), "fasta_len": len(fasta),
                                "threshold": 0.7, "nb_best_hits": 2}, fasta)
        from phylign_amd import postprocess as P
        assert h1["ok"] and h1["cached"] and b1.decode() == P.filter_text(exp.decode(), 2)
        st, _ = request(sock, {"op": "stats"})
        assert st["loads"] == 1 and st["hits"] == 2 and st["resident"] == 1
        # a cold load of a multi-block file (`xz -T`): decoded block-parallel in the server, same text
        plain = tmp_path / "multi__01.cobs_classic"
        plain.write_bytes(bytes(index))
        subprocess.run(["xz", "-T2", "-0", "--block-size=1MiB", str(plain)], check=True)
        h2, b2 = request(sock, {"op": "query", "index": str(plain) + ".xz", "index_size": len(index), "fasta_len": len(fasta),
                                "threshold": 0.7}, fasta)
        assert h2["ok"] and not h2["cached"] and b2 == exp
        bad, _ = request(sock, {"op": "query", "index": str(tmp_path / "missing.xz"), "fasta_len": 0})
        assert not bad["ok"]
        r = subprocess.run([script, "0.7", "1", str(tmp_path / "missing.xz"), "1", str(fa)], capture_output=True, env=cenv)
        assert r.returncode != 0 and r.stdout == b""
    finally:
        try:
            request(sock, {"op": "shutdown"})
        except Exception:
            srv.kill()
        srv.wait(timeout=30)


def test_resident_index_server_evicts_under_budget(pm, oracle, tmp_path):
    """LRU under an HBM budget: with room for one index, loading a second one drops the first"""
    import time
    from phylign_amd.server import request
    sock = str(tmp_path / "pm.sock")
    env = dict(os.environ, PYTHONPATH=ROOT)
    paths, texts = [], []
    for i in range(2):
        index, fasta, _ = _case(oracle, seed=50 + i, n_docs=664, S=300000)      # ~38 MB in HBM each
        p = tmp_path / f"i{i}.cobs_classic"
        p.write_bytes(bytes(index))
        paths.append((str(p), len(index), fasta))
        texts.append(oracle.query_file(index, fasta, 0.7))
    srv = subprocess.Popen([sys.executable, "-m", "phylign_amd.server", "--socket", sock, "--max-gb", "0.06"],
                           env=env, stderr=subprocess.PIPE)
    try:
        for _ in range(600):
            if os.path.exists(sock):
                break
            time.sleep(0.1)
        seq = [0, 1, 1, 0]
        cached = []
        for i in seq:
            p, n, fasta = paths[i]
            h, body = request(sock, {"op": "query", "index": p, "index_size": n, "fasta_len": len(fasta), "threshold": 0.7}, fasta)
            assert h["ok"] and body == texts[i]
            cached.append(h["cached"])
        assert cached == [False, False, True, False]
        st, _ = request(sock, {"op": "stats"})
        assert st["resident"] == 1 and st["loads"] == 3
    finally:
        try:
            request(sock, {"op": "shutdown"})
        except Exception:
            srv.kill()
        srv.wait(timeout=30)


def test_server_keeps_serving_during_a_cold_load_and_fuses_concurrent_jobs(pm, oracle, tmp_path):
    """what 305 per-batch Snakemake jobs do to a resident-index server (Snakefile:431-487): requests arrive together.
    A cold index (fed through a FIFO that stalls for 4 s) is loaded by its own handler thread; meanwhile three
    clients with the same query file against three resident batches are answered at once -- by ONE fused search --,
    a second client of the cold index waits for the same load instead of starting another, and every answer is
    byte-identical to the oracle."""
    import threading
    import time
    from phylign_amd.server import request
    sock = str(tmp_path / "pm.sock")
    env = dict(os.environ, PYTHONPATH=ROOT)
    rng = np.random.default_rng(91)
    queries = [(f"c{i}", rand_seq(rng, 150)) for i in range(40)]
    cases = []
    for b, (n_docs, S) in enumerate(((195, 30000), (664, 20000), (4000, 9000), (300, 25000))):
        plant = [(qi, (qi * 13 + j) % n_docs, fr) for qi in range(40) for j, fr in enumerate((1.0, 0.9, 0.8, 0.7, 0.6))]
        index, fasta, _ = build_case(oracle, rng, n_docs, S, queries, plant=plant)
        p = tmp_path / f"res{b}__01.cobs_classic"
        if b < 3:
            p.write_bytes(bytes(index))
        cases.append((str(p), bytes(index), oracle.query_file(index, fasta, 0.7)))
    cold_path, cold_bytes, cold_text = cases[3]
    os.mkfifo(cold_path)
    srv = subprocess.Popen([sys.executable, "-m", "phylign_amd.server", "--socket", sock, "--coalesce-ms", "400"],
                           env=env, stderr=subprocess.PIPE)
    try:
        for _ in range(600):
            if os.path.exists(sock):
                break
            time.sleep(0.1)
        assert os.path.exists(sock), "server did not come up"
        h, _ = request(sock, {"op": "preload", "indexes": [c[0] for c in cases[:3]], "wait": True})
        assert h["ok"] and h["queued"] == 3
        st, _ = request(sock, {"op": "stats"})
        assert st["resident"] == 3 and st["loads"] == 3 and st["resident_bytes"] > 0

        def feed():                                   # the slow decoder of the cold index
            with open(cold_path, "wb") as f:
                f.write(cold_bytes[: len(cold_bytes) // 2])
                f.flush()
                time.sleep(4.0)
                f.write(cold_bytes[len(cold_bytes) // 2:])
        out, lat = {}, {}

        def client(name, path):
            t0 = time.time()
            out[name] = request(sock, {"op": "query", "index": path, "fasta_len": len(fasta), "threshold": 0.7}, fasta)
            lat[name] = time.time() - t0
        threads = [threading.Thread(target=feed), threading.Thread(target=client, args=("cold_a", cold_path))]
        for t in threads:
            t.start()
        time.sleep(0.3)                               # the cold load is under way
        more = [threading.Thread(target=client, args=("cold_b", cold_path))]
        more += [threading.Thread(target=client, args=(f"res{b}", cases[b][0])) for b in range(3)]
        for t in more:
            t.start()
        for t in threads + more:
            t.join(timeout=120)
        for b in range(3):
            head, body = out[f"res{b}"]
            assert head["ok"] and head["cached"] and body == cases[b][2], b
            assert lat[f"res{b}"] < 3.0, lat               # answered while the cold index was still loading
        for name in ("cold_a", "cold_b"):
            head, body = out[name]
            assert head["ok"] and body == cold_text
            assert lat[name] > 3.0
        st, _ = request(sock, {"op": "stats"})
        assert st["loads"] == 4 and st["waited_for_a_load"] >= 1 and st["resident"] == 4
        assert st["max_batches_in_one_search"] >= 3 and st["searches"] < st["jobs"] == 5
    finally:
        try:
            request(sock, {"op": "shutdown"})
        except Exception:
            srv.kill()
        try:
            srv.wait(timeout=30)
        except Exception:
            srv.kill()


def test_match_stage_plain_invocation_with_eight_ranks(pm, oracle, tmp_path):
    """`python -m phylign_amd.match_stage --gpus 8` without a launcher starts its own 8 ranks (here sharing the one GPU
    over gloo); the fixture has 5 batches, so three ranks hold none -- they still take part in the gather -- and the 5
    files and the 04_filter FASTA are the one-rank run's"""
    names, indexes, fasta = _stage_fixture(oracle, tmp_path)
    env = dict(os.environ, PYTHONPATH=ROOT, PHYLIGN_DIST_BACKEND="gloo", PHYLIGN_SHARE_GPU="1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    r = subprocess.run([sys.executable, "-m", "phylign_amd.match_stage", "--gpus", "8",
                        "--batches", str(tmp_path / "batches.txt"), "--cobs-dir", str(tmp_path / "cobs"),
                        "--sizes", str(tmp_path / "sizes.txt"), "--queries", str(tmp_path / "Q.fa"),
                        "--out-dir", str(tmp_path / "03_match"), "--nb-best-hits", "3",
                        "--filter-out", str(tmp_path / "04_filter" / "Q.fa")], capture_output=True, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    _check_stage_outputs(oracle, tmp_path, names, indexes, fasta, 3)
    import json
    reports = [json.loads(ln) for ln in r.stderr.decode().splitlines() if ln.startswith("{") and '"world"' in ln]
    assert sorted(rep["rank"] for rep in reports) == list(range(8)) and all(rep["world"] == 8 for rep in reports)
    assert sorted(rep["batches"] for rep in reports) == [0, 0, 0, 1, 1, 1, 1, 1]
    # the 8 ranks of one node share its RAM: together their xz-decoder budgets stay within what one rank alone would take
    from phylign_amd.sysinfo import available_ram_gb
    assert sum(rep["host_ram_plan"]["budget_mb"] for rep in reports) <= 0.8 * available_ram_gb() * 1024 * 1.1


def test_query_chunks_do_not_accumulate_in_hbm(pm, oracle, tmp_path):
    """a query file searched chunk after chunk holds at most two chunks' device state at a time: every chunk's HBM copies
    (sequences, 8 bytes of hash per k-mer) are released once its unit is finished, so the free HBM after the stage does
    not depend on the number of chunks -- and a released set is simply uploaded again by its next search"""
    from phylign_amd import match_stage as MS
    from phylign_amd import workload as W
    names, indexes, fasta = _stage_fixture(oracle, tmp_path)
    batches = sorted(names)
    src = MS.ResidentSource({b: pm.Index.load_mem(indexes[b]) for b in batches})
    big, _ = W.make_queries(60000, 150, seed=9)                       # 60 k reads: ~66 MB of device state in one piece
    free_after = {}
    for nchunks in (1, 6):
        pieces = MS.split_prepared_fasta(big, 60000 // nchunks)
        assert len(pieces) == nchunks
        chunks = [pm.Queries(bytes(p_)) for p_ in pieces]
        rep, _ = MS.run_stage(pm, batches, list(range(len(batches))), src, chunks if nchunks > 1 else chunks[0], "big",
                              str(tmp_path / f"03_{nchunks}"), 0.7, 3)
        per_chunk = [c.device_bytes() for c in chunks]
        if nchunks > 1:
            assert rep["query_hbm_bytes_at_end"] == 0 and all(res == 0 and need > 1 << 20 for res, need in per_chunk)
        else:
            assert per_chunk[0][0] == per_chunk[0][1] > 60 << 20
            chunks[0].release_device()
            assert chunks[0].device_bytes()[0] == 0
        pm.set_option("release_query_pool", 1)                       # released copies wait in a bounded pool: really free them
        free_after[nchunks] = pm.device_info()["hbm_free"]
        # a released set is searched again like a fresh one
        a = pm.search([src.indexes[batches[0]]], chunks[-1], 0.7).hits()
        chunks[-1].release_device()
        b = pm.search([src.indexes[batches[0]]], chunks[-1], 0.7).hits()
        assert np.array_equal(a, b)
        for c in chunks:
            c.free()
    assert abs(free_after[1] - free_after[6]) < 32 << 20, free_after
    for b in batches:                                                 # the pieces of a batch's file are the one-piece file
        assert gzip.open(tmp_path / "03_1" / f"{b}____big.gz", "rb").read() == gzip.open(tmp_path / "03_6" / f"{b}____big.gz", "rb").read()


def test_match_stage_decode_once_cache(pm, oracle, tmp_path):
    """--cache-dir: the first run decodes every .xz into HBM and leaves <cache>/<batch>.cobs_classic behind (the bytes of
    the decoded stream, no .tmp); the second run finds them and decodes nothing; both runs write the same files.  The
    loader plan comes from the sizes table (sizing.stage_plan); mem-disk is the same thing under the reference's name"""
    import json
    names, indexes, fasta = _stage_fixture(oracle, tmp_path)
    env = dict(os.environ, PYTHONPATH=ROOT)
    cache = tmp_path / "cache"
    base = [sys.executable, "-m", "phylign_amd.match_stage", "--batches", str(tmp_path / "batches.txt"),
            "--cobs-dir", str(tmp_path / "cobs"), "--sizes", str(tmp_path / "sizes.txt"), "--queries", str(tmp_path / "Q.fa"),
            "--nb-best-hits", "3", "--max-ram-gb", "4"]
    reports = []
    for run, extra in enumerate((["--cache-dir", str(cache)], ["--cache-dir", str(cache)],
                                 ["--index-load-mode", "mem-disk", "--decompression-dir", str(cache)])):
        out = tmp_path / f"03_match_{run}"
        r = subprocess.run(base + ["--out-dir", str(out), "--filter-out", str(tmp_path / f"04_{run}" / "Q.fa")] + extra,
                           capture_output=True, env=env)
        assert r.returncode == 0, r.stderr.decode()[-3000:]
        reports.append(json.loads([ln for ln in r.stderr.decode().splitlines() if ln.startswith("{")][-1]))
    # the block structure of the .xz files that were decoded (`xz --list`): python's lzma writes one block per file
    assert reports[0]["index_source"].pop("xz_blocks") == {"files": 5, "multi_block_files": 0, "blocks_max": 1, "blocks_total": 5}
    assert reports[0]["index_source"] == {"xz_decoded": 5, "plain_files": 0, "cache_files_written": 5, "resident": 0, "xz_decoded_block_parallel": 0}
    assert reports[1]["index_source"] == {"xz_decoded": 0, "plain_files": 5, "cache_files_written": 0, "resident": 0, "xz_decoded_block_parallel": 0}
    assert reports[2]["index_source"] == reports[1]["index_source"]
    plan = reports[0]["host_ram_plan"]
    assert plan["budget_mb"] == 4096 and plan["decoder_mb_max"] == 1537 + 64 and plan["loaders"] == 2
    assert 0 < plan["decoders_peak_mb"] <= 4096 and reports[1]["host_ram_plan"]["decoders_peak_mb"] == 0
    assert sorted(f.name for f in cache.iterdir()) == sorted(f"{b}.cobs_classic" for b in names)
    for b in names:
        assert (cache / f"{b}.cobs_classic").read_bytes() == bytes(indexes[b])
        for run in (1, 2):
            assert (tmp_path / f"03_match_{run}" / f"{b}____Q.gz").read_bytes() == (tmp_path / "03_match_0" / f"{b}____Q.gz").read_bytes()
    (tmp_path / "03_match").mkdir()
    for f in (tmp_path / "03_match_1").iterdir():
        (tmp_path / "03_match" / f.name).write_bytes(f.read_bytes())
    (tmp_path / "04_filter").mkdir()
    (tmp_path / "04_filter" / "Q.fa").write_bytes((tmp_path / "04_1" / "Q.fa").read_bytes())
    _check_stage_outputs(oracle, tmp_path, names, indexes, fasta, 3)
    # a truncated stream leaves no cache entry
    broken = sorted(names)[1]
    (cache / f"{broken}.cobs_classic").unlink()
    (tmp_path / "cobs" / f"{broken}.cobs_classic.xz").write_bytes(lzma.compress(bytes(indexes[broken][: len(indexes[broken]) // 2]), preset=1))
    r = subprocess.run(base + ["--out-dir", str(tmp_path / "03_match_x"), "--cache-dir", str(cache)], capture_output=True, env=env)
    assert r.returncode != 0
    assert not (cache / f"{broken}.cobs_classic").exists() and not list(cache.glob("*.tmp"))


def test_two_decoders_of_one_batch_share_a_cache_directory(pm, oracle, tmp_path):
    """two loads of the same uncached batch into the same cache path at once (two stage runs, or stage and
    server): each writes a temporary of its own, the published file is complete whichever finishes first, nothing else is
    left behind, and both indexes are the resident index"""
    import threading
    rng = np.random.default_rng(3)
    index, fasta, _ = build_case(oracle, rng, 664, 150000, [("q", rand_seq(rng, 150))])     # 12 MB through paced pipes: the two loads overlap
    blob = bytes(index)
    tee = str(tmp_path / "b__01.cobs_classic")
    out, errs = [None, None], []

    def loader(i):
        try:
            r, w = os.pipe()

            def feed():
                with os.fdopen(w, "wb") as f:
                    for o in range(0, len(blob), 1 << 16):
                        f.write(blob[o:o + (1 << 16)])
            t = threading.Thread(target=feed)
            t.start()
            try:
                out[i] = pm.Index.load_fd(r, size_hint=len(blob), tee_path=tee)
            finally:
                os.close(r)
                t.join()
        except Exception as e:                              # noqa: BLE001
            errs.append(e)
    ts = [threading.Thread(target=loader, args=(i,)) for i in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs and all(ix is not None and ix.cached for ix in out)
    assert sorted(f.name for f in tmp_path.iterdir()) == ["b__01.cobs_classic"]
    assert open(tee, "rb").read() == blob
    for ix in out:
        assert pm.query_text(ix, fasta, 0.7) == oracle.query_file(index, fasta, 0.7)


def test_a_killed_cold_load_leaves_no_cache_temporary(pm, oracle, tmp_path):
    """the decode-once cache file is an unnamed inode (O_TMPFILE) until it is complete, so a load that is
    SIGKILLed half way -- the launcher's grace period, the OOM killer -- leaves nothing in the cache directory, whatever it
    had written; a finished load publishes the plain file under its final name only"""
    import signal
    import time
    rng = np.random.default_rng(4)
    index, fasta, _ = build_case(oracle, rng, 664, 60000, [("q", rand_seq(rng, 150))])
    blob = tmp_path / "whole.bin"
    blob.write_bytes(bytes(index))
    cache = tmp_path / "cache"
    cache.mkdir()
    code = (
        "import os, sys, time, threading\n"
        "from phylign_amd import _lib as pm\n"
        "pm.load(); pm.init(0)\n"
        "blob = open(sys.argv[1], 'rb').read(); half = len(blob) // 2\n"
        "r, w = os.pipe()\n"
        "def feed():\n"
        "    os.write(w, blob[:half])\n"
        "    open(sys.argv[3], 'w').write('half')\n"
        "    if sys.argv[4] == 'stall':\n"
        "        time.sleep(600)\n"
        "    with os.fdopen(w, 'wb') as f:\n"
        "        f.write(blob[half:])\n"
        "threading.Thread(target=feed, daemon=True).start()\n"
        "ix = pm.Index.load_fd(r, size_hint=len(blob), tee_path=sys.argv[2])\n"
        "print('cached', ix.cached)\n")
    env = dict(os.environ, PYTHONPATH=ROOT)
    tee = cache / "b__01.cobs_classic"
    flag = tmp_path / "half_written"
    p = subprocess.Popen([sys.executable, "-c", code, str(blob), str(tee), str(flag), "stall"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    t0 = time.time()
    while not flag.exists() and p.poll() is None and time.time() - t0 < 240:
        time.sleep(0.05)
    assert flag.exists() and p.poll() is None, p.stderr.read().decode()[-2000:]
    time.sleep(0.5)                                             # the loader has consumed (and teed) what was written
    p.send_signal(signal.SIGKILL)
    p.wait()
    assert list(cache.iterdir()) == []                          # nothing: not even a *.tmp
    flag.unlink()
    r = subprocess.run([sys.executable, "-c", code, str(blob), str(tee), str(flag), "go"], env=env, capture_output=True)
    assert r.returncode == 0 and b"cached True" in r.stdout, r.stderr.decode()[-2000:]
    assert [f.name for f in cache.iterdir()] == ["b__01.cobs_classic"] and tee.read_bytes() == blob.read_bytes()
    assert oct(tee.stat().st_mode & 0o777) == "0o644"


def test_match_stage_with_gene_length_queries(pm, oracle, tmp_path):
    """the reference's bundled gene file as a query set: every 8th record length of data/ARGannot_r3.fa
    (232 genes, 237 ... 3 150 bp: the 10- and 13-plane counter classes in one search) through the whole stage -- .xz
    indexes, fused post-filter, gzip members, 04_filter merge -- against the oracle and the golden-pinned mirrors"""
    from phylign_amd import workload as W
    lens = W.argannot_lengths()[::8]
    fasta, seqs = W.make_queries_lengths(lens, seed=5, prefix="ARG")
    queries = [(f"ARG{i:07d}", s_.decode()) for i, s_ in enumerate(seqs)]
    rng = np.random.default_rng(8)
    cobs = tmp_path / "cobs"
    cobs.mkdir()
    names, indexes = [], {}
    with open(tmp_path / "sizes.txt", "w") as sz:
        for b, (n_docs, S) in enumerate([(195, 40009), (664, 30011), (2300, 9001)]):
            batch = f"genus_species{b}__01"
            plant = [(qi, int(rng.integers(0, n_docs)), fr) for qi in range(0, len(lens), 3) for fr in (1.0, 0.9, 0.8, 0.7, 0.7, 0.69)]
            index, fa2, _ = build_case(oracle, rng, n_docs, S, queries, plant=plant, density=0.2)
            assert fa2 == fasta
            (cobs / f"{batch}.cobs_classic.xz").write_bytes(lzma.compress(bytes(index), preset=0))
            sz.write(f"cobs/{batch}.cobs_classic.xz  {len(index)}  1610678320\n")
            names.append(batch)
            indexes[batch] = index
    (tmp_path / "batches.txt").write_text("\n".join(names) + "\n")
    (tmp_path / "Q.fa").write_bytes(fasta)
    r = subprocess.run([sys.executable, "-m", "phylign_amd.match_stage", "--batches", str(tmp_path / "batches.txt"),
                        "--cobs-dir", str(cobs), "--sizes", str(tmp_path / "sizes.txt"), "--queries", str(tmp_path / "Q.fa"),
                        "--out-dir", str(tmp_path / "03_match"), "--nb-best-hits", "2", "--filter-out", str(tmp_path / "04_filter" / "Q.fa")],
                       capture_output=True, env=dict(os.environ, PYTHONPATH=ROOT))
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    _check_stage_outputs(oracle, tmp_path, names, indexes, fasta, 2)


def test_multi_block_xz_files_are_decoded_on_several_threads(pm, oracle, tmp_path):
    """a rank with fewer compressed batches than CPUs (the reference's batches_small.txt has three) decodes the blocks of a
    multi-block .xz file (`xz -T`) side by side in-process (phylign_amd/xzpar.py) instead of through one xzcat; outputs
    and cache files are the same; a one-block file and PHYLIGN_XZ_THREADS=1 take the xzcat pipe"""
    import json
    names, indexes, fasta = _stage_fixture(oracle, tmp_path, n_batches=3)
    for b in names[:2]:                                   # two batches as multi-block files, the third stays one block
        xz = tmp_path / "cobs" / f"{b}.cobs_classic.xz"
        plain = tmp_path / "cobs" / f"{b}.cobs_classic"
        plain.write_bytes(bytes(indexes[b]))
        xz.unlink()
        subprocess.run(["xz", "-T2", "-0", "--block-size=64KiB", str(plain)], check=True)
        assert not plain.exists()
    env = dict(os.environ, PYTHONPATH=ROOT)
    base = [sys.executable, "-m", "phylign_amd.match_stage", "--batches", str(tmp_path / "batches.txt"), "--cobs-dir", str(tmp_path / "cobs"),
            "--sizes", str(tmp_path / "sizes.txt"), "--queries", str(tmp_path / "Q.fa"), "--nb-best-hits", "3"]
    r = subprocess.run(base + ["--out-dir", str(tmp_path / "03_match"), "--filter-out", str(tmp_path / "04_filter" / "Q.fa"),
                               "--cache-dir", str(tmp_path / "cache")], capture_output=True, env=dict(env, PHYLIGN_XZ_THREADS="4"))
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    rep = json.loads([ln for ln in r.stderr.decode().splitlines() if ln.startswith("{")][-1])
    assert rep["index_source"]["xz_decoded"] == 3 and rep["index_source"]["xz_decoded_block_parallel"] == 2
    assert rep["index_source"]["xz_blocks"]["multi_block_files"] == 2 and rep["index_source"]["cache_files_written"] == 3
    _check_stage_outputs(oracle, tmp_path, names, indexes, fasta, 3)
    for b in names:
        assert (tmp_path / "cache" / f"{b}.cobs_classic").read_bytes() == bytes(indexes[b])
    r = subprocess.run(base + ["--out-dir", str(tmp_path / "03_one")], capture_output=True, env=dict(env, PHYLIGN_XZ_THREADS="1"))
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    rep1 = json.loads([ln for ln in r.stderr.decode().splitlines() if ln.startswith("{")][-1])
    assert rep1["index_source"]["xz_decoded_block_parallel"] == 0
    for b in names:
        assert (tmp_path / "03_one" / f"{b}____Q.gz").read_bytes() == (tmp_path / "03_match" / f"{b}____Q.gz").read_bytes()
    # a damaged block: the stage fails, no cache entry for that batch
    bad = tmp_path / "cobs" / f"{names[0]}.cobs_classic.xz"
    blob = bytearray(bad.read_bytes())
    blob[len(blob) // 2] ^= 0xFF
    bad.write_bytes(bytes(blob))
    r = subprocess.run(base + ["--out-dir", str(tmp_path / "03_bad"), "--cache-dir", str(tmp_path / "cache2")], capture_output=True,
                       env=dict(env, PHYLIGN_XZ_THREADS="4"))
    assert r.returncode != 0 and not (tmp_path / "cache2" / f"{names[0]}.cobs_classic").exists() and not list((tmp_path / "cache2").glob("*.tmp"))

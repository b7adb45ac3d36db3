"""Round-2 device paths on the GPU, against the oracle: hit lists written as ordered
runs by the scan kernel (a7 on the device), searches in flight (pm_search_async), the
clustered "home batch" content, the gathered-bytes counter, GPU binding of worker threads."""
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

from helpers import bench_record, build_case, doc_names, rand_seq

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _names(ix):
    return [ix.doc_name(d) for d in range(ix.info.n_docs)]


def _file_of(oracle, ix):
    """the .cobs_classic bytes of an index that lives in HBM (synthetic / planted content)"""
    info = ix.info
    rows = ix.read_rows(0, info.signature_size)
    return oracle.make_index(info.term_size, info.canonicalize, info.signature_size, info.num_hashes, _names(ix), rows)


@pytest.mark.parametrize("n_docs,S,qlen", [(664, 6000, 150), (4000, 2500, 150), (100, 3000, 150), (13, 2000, 400),
                                            (1300, 1500, 1500), (9001, 700, 150)])
@pytest.mark.parametrize("layout", [1, 2])
def test_clustered_home_batch_text_bit_exact(pm, oracle, n_docs, S, qlen, layout):
    """long hit lists with many distinct scores and ties: the runs the kernel writes are cobs' lines"""
    rng = np.random.default_rng(n_docs + qlen)
    nq = 24
    queries = [(f"c{i} x", rand_seq(rng, qlen)) for i in range(nq)]
    fasta = "".join(f">{h}\n{s}\n" for h, s in queries).encode()
    q = pm.Queries(fasta)
    ix = pm.Index.synth(3, n_docs, S, seed=661, layout=layout)
    ix.plant_cluster(q, 0, 2, seed=97)                      # every second query is at home here
    index = _file_of(oracle, ix)
    for thr in (0.7, 0.5, 0.0):
        exp = oracle.query_file(index, fasta, thr)
        for bound in (1, 0):
            pm.set_option("threshold_bound", bound)
            assert pm.query_text(ix, fasta, thr) == exp, (thr, bound)
        pm.set_option("threshold_bound", 1)
    exp = oracle.query_file(index, fasta, 0.7).decode()
    assert exp.count("\n") > nq + nq // 2 * min(n_docs, 32) // 2
    from phylign_amd import postprocess as P
    for n in (1, 3, 100):
        assert pm.query_text(ix, fasta, 0.7, nb_best_hits=n).decode() == P.filter_text(exp, n)


def test_device_records_are_ordered_runs(pm, oracle):
    """raw records in HBM: one run per (query, slot) = count record + hits in cobs line order"""
    rng = np.random.default_rng(77)
    n_docs, S = 664, 5000
    queries = [(f"r{i}", rand_seq(rng, 150)) for i in range(40)]
    plant = [(qi, (qi * 7 + 3 * j) % n_docs, fr) for qi in range(0, 40, 2)
             for j, fr in enumerate((1.0, 0.9, 0.9, 0.8, 0.8, 0.8, 0.75, 0.7, 0.7, 0.6))]
    index, fasta, _ = build_case(oracle, rng, n_docs, S, queries, plant=plant)
    ix = pm.Index.load_mem(index)
    q = pm.Queries(fasta)
    import torch
    for n in (0, 3):
        res = pm.search([ix, ix], q, 0.7, slot_base=5, nb_best_hits=n)
        st = res.stats
        assert st.n_records == st.n_hits + st.n_runs and st.n_runs >= 2 * 20
        buf = torch.zeros((st.n_records, 4), dtype=torch.int32, device="cuda")
        res.copy_hits_device(buf.data_ptr(), st.n_records)
        raw = buf.cpu().numpy().view(pm.HIT_DTYPE).reshape(-1)
        heads = np.flatnonzero(raw["doc"] == pm.PM_DOC_COUNT)
        assert heads[0] == 0 and len(heads) == st.n_runs
        ends = list(heads[1:]) + [len(raw)]
        seen = set()
        for b, e in zip(heads, ends):
            run = raw[b + 1:e]
            assert len(run) and (run["query"] == raw[b]["query"]).all() and (run["slot"] == raw[b]["slot"]).all()
            key = list(zip(-run["score"].astype(np.int64), run["doc"]))
            assert key == sorted(key)
            assert raw[b]["score"] >= len(run) and (n or raw[b]["score"] == len(run))
            seen.add((int(raw[b]["slot"]), int(raw[b]["query"])))
        assert len(seen) == st.n_runs                       # one run per (slot, query)
        host = res.hits()
        assert pm.format_hits(ix, q, host, slot=6, nb_best_hits=n if n else -1) == \
            pm.format_hits(ix, q, raw, slot=6, nb_best_hits=n if n else -1)   # raw runs format the same
        assert np.array_equal(pm.sort_hits(host.copy()), host)
        # the ordered form in HBM (what RCCL sends) is what the host sees
        dptr, n_out = res.ordered_device()
        assert n_out == len(host) and dptr
        buf2 = torch.zeros((n_out, 4), dtype=torch.int32, device="cuda")
        res.copy_hits_device(buf2.data_ptr(), n_out, ordered=True)
        assert np.array_equal(buf2.cpu().numpy().view(pm.HIT_DTYPE).reshape(-1), host)
        assert np.array_equal(res.hits(copy=False), host)


def test_searches_in_flight_give_the_same_records(pm, oracle):
    rng = np.random.default_rng(78)
    queries = [(f"a{i}", rand_seq(rng, 150)) for i in range(64)]
    cases = [build_case(oracle, rng, d, s, queries, plant=[(i, (i * 5) % d, 0.8) for i in range(0, 64, 3)])
             for d, s in ((300, 4000), (4000, 1500))]
    ixs = [pm.Index.load_mem(c[0]) for c in cases]
    q = pm.Queries(cases[0][1])
    want = [pm.search([ixs[0]], q, 0.7).hits(), pm.search(ixs, q, 0.7, slot_base=2).hits(),
            pm.search([ixs[1]], q, 0.3, nb_best_hits=2).hits()]
    inflight = [pm.search_async([ixs[0]], q, 0.7), pm.search_async(ixs, q, 0.7, slot_base=2),
                pm.search_async([ixs[1]], q, 0.3, nb_best_hits=2), pm.search_async([ixs[0]], q, 0.7)]
    got = [r.hits() for r in reversed(inflight)][::-1]      # collected out of order
    for g, w in zip(got, want + [want[0]]):
        assert np.array_equal(g, w)
    r = pm.search_async(ixs, q, 0.7)
    r.free()                                                # freeing an unfinished search is safe
    assert np.array_equal(pm.search([ixs[0]], q, 0.7).hits(), want[0])


@pytest.mark.parametrize("thr", [0.7, 0.0])
def test_gathered_bytes_counter(pm, oracle, thr):
    rng = np.random.default_rng(79)
    q = pm.Queries("".join(f">g{i}\n{rand_seq(rng, 150)}\n" for i in range(300)).encode())
    ixs = [pm.Index.synth(b, d, s, seed=661) for b, (d, s) in enumerate(((664, 30000), (4000, 9000), (100, 20000), (9001, 2000)))]
    pm.set_option("count_fetched", 1)
    try:
        out = {}
        for bound in (0, 1):
            pm.set_option("threshold_bound", bound)
            res = pm.search(ixs, q, thr)
            st = res.stats
            out[bound] = (res.hits(), st.fetched_bytes, st.algorithmic_bytes,
                          [(L["fetched_bytes"], L["algorithmic_bytes"]) for L in res.launches()])
        assert np.array_equal(out[0][0], out[1][0])
        assert out[0][1] == out[0][2] == 300 * 120 * (83 + 500 + 13 + 1126)      # fetch-all gathers every row byte
        assert all(f == a for f, a in out[0][3])
        if thr > 0:
            assert 0.2 * out[1][2] < out[1][1] < 0.8 * out[1][2]                  # the bound skips dead lines
        else:
            assert out[1][1] == out[1][2]                                          # threshold 0: nothing is ever out
    finally:
        pm.set_option("count_fetched", 0)
        pm.set_option("threshold_bound", 1)
    res = pm.search(ixs, q, thr)
    assert res.stats.fetched_bytes == 0


def test_worker_threads_use_the_bound_gpu(pm, oracle):
    """HIP's current device is per thread: loads from a thread pool must land on pm_init()'s GPU"""
    rng = np.random.default_rng(80)
    index, fasta, _ = build_case(oracle, rng, 300, 3000, [("w0", rand_seq(rng, 150))], plant=[(0, 7, 1.0)])
    out = []

    def work():
        ix = pm.Index.load_mem(index)
        out.append((ix.device, pm.query_text(ix, fasta, 0.7)))
    ts = [threading.Thread(target=work) for _ in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert len(out) == 4
    for devno, text in out:
        assert devno == pm.bound_device() == 0 and text == oracle.query_file(index, fasta, 0.7)
    assert pm.Index.load_header_mem(index).device == -1


def test_second_gpu_binding_when_present(pm, oracle):
    """on a multi-GPU node: a process bound to GPU 1 keeps its matrices there, also from threads"""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    code = (
        "import sys, threading; sys.path.insert(0, %r)\n"
        "from phylign_amd import _lib as pm\n"
        "pm.init(1)\n"
        "out = []\n"
        "def work():\n"
        "    out.append(pm.Index.synth(0, 300, 5000).device)\n"
        "ts = [threading.Thread(target=work) for _ in range(3)]\n"
        "[t.start() for t in ts]; [t.join() for t in ts]\n"
        "assert out == [1, 1, 1], out\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()[-2000:]


def test_names_without_separator_are_cut_on_the_host(pm, oracle):
    """scripts/postprocess_cobs.py:10-18 raises on a line without '_' once its rank reaches n; an
    index with such a name is therefore never cut on the GPU, so the host rule sees every line"""
    rng = np.random.default_rng(81)
    n_docs, S = 200, 4000
    queries = [(f"n{i}", rand_seq(rng, 150)) for i in range(8)]
    plant = [(qi, d, fr) for qi in range(8) for d, fr in ((5, 1.0), (9, 0.9), (11, 0.8), (20, 0.75), (21, 0.72))]
    rb = (n_docs + 7) // 8
    bits = rng.random((S, rb * 8)) < 0.05
    bits[:, n_docs:] = False
    matrix = np.packbits(bits, axis=1, bitorder="little")
    for qi, doc, frac in plant:
        hs = oracle.create_hashes(queries[qi][1].encode(), 31, 1, 1)
        for t in range(int(np.ceil(frac * len(hs)))):
            matrix[int(hs[t]) % S, doc >> 3] |= np.uint8(1 << (doc & 7))
    names = doc_names(rng, n_docs)
    names[21] = "nounderscore"                                  # the worst hit of every query
    index = oracle.make_index(31, 1, S, 1, names, matrix)
    fasta = "".join(f">{h}\n{s}\n" for h, s in queries).encode()
    ix = pm.Index.load_mem(index)
    q = pm.Queries(fasta)
    got = pm.search([ix], q, 0.7, nb_best_hits=2).hits()
    assert not (got["doc"] == pm.PM_DOC_COUNT).any() and len(got) == 8 * 5     # nothing was cut on the GPU
    with pytest.raises(pm.PMError):                                           # the reference raises as well (rank 5 >= n)
        pm.format_hits(ix, q, got, nb_best_hits=2)
    assert pm.format_hits(ix, q, got, nb_best_hits=-1) == oracle.query_file(index, fasta, 0.7)


def _check_multi_rank_line(line, full, n, live_pmc=False):
    """what makes an N > 1 bench line count as measured: cpu_baseline and roofline present in the LINE, ranks
    counted; the auxiliary legs sit in the side file, left out WITH the reason where they do not apply"""
    cb = line["cpu_baseline"]
    assert cb and cb["value"] > 0 and cb["kind"] == "port" and cb["cores"] >= 1 and "slept" in cb["timed_while"]
    assert line["gpu_over_cpu"] > 0
    rf = line["roofline"]
    assert rf and rf["bound"] == "hbm" and 0 < rf["frac"] < 1.0 and rf["peak"] == 8000.0 and rf["achieved"] > 0
    lp = full.get("live_pmc")
    if live_pmc and lp and lp["error"] is None:
        # rank 0 measured its own launch: PMC child runs on the shard it held, the other ranks asleep
        assert rf["traffic"] == lp["kernels"][rf["kernel"]]["hbm_bytes_per_launch"] > 0 and f"rank 0's shard of {n}" in rf["traffic_source"]
        assert f"--emulate-world {n} --emulate-rank 0" in lp["how"]
    elif live_pmc:
        assert rf["traffic"] is None and "live pass" in rf["traffic_note"]
    else:
        assert rf["traffic"] is None and "1-rank launch" in rf["traffic_note"]
    assert line["n_gpus"] == n and line["participants"]["ranks"] == n
    assert "rccl_ranks" in line["participants"]          # null over gloo (these tests), N over RCCL
    ur = full["unique_rows"]
    assert ur and 0 < ur["unique_rows_x_row_bytes"] <= ur["algorithmic_bytes_per_step"] and ur["unique_rows"] <= ur["rows_resident"]
    for leg in ("argannot", "l31", "full_shard"):                  # single-GPU legs: left out WITH the reason, never a bare null
        assert "skipped" in full[leg], leg
    assert full["full_collection"] is not None and full["clustered"] is not None
    assert len(full["participants"]["rank_host_ms"]) == n


def test_bench_two_ranks_equal_one_rank(pm, tmp_path):
    """the N>1 bench path (static sharding + packed gather, two ranks sharing the GPU over gloo)
    returns exactly the records of the single-rank run"""
    env = dict(os.environ, PYTHONPATH=ROOT)
    common = ["--steps", "2", "--warmup", "1", "--rows-divisor", "400", "--queries", "3000", "--no-cpu-baseline", "--no-live-pmc"]
    one = tmp_path / "one.npy"
    legs = tmp_path / "legs_one.json"
    # (this first run keeps the live PMC passes: `roofline.traffic` is measured by rocprofv3 child runs of the same workload)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + [a_ for a_ in common if a_ != "--no-live-pmc"] +
                       ["--dump-hits", str(one), "--legs-out", str(legs)], capture_output=True, env=env)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    # the driver-facing line: small strict JSON with the contract's keys; every auxiliary leg is in the side file
    line, full = bench_record(r.stdout, legs)
    lp = full["live_pmc"]
    if lp["error"] is None:
        assert lp["kernels"], lp
        for key in ("roofline", "roofline_narrow"):
            rf = line[key]
            assert rf["traffic_source"].startswith("live: rocprofv3 --pmc") and rf["traffic"] == lp["kernels"][rf["kernel"]]["hbm_bytes_per_launch"] > 0
            assert lp["kernels"][rf["kernel"]]["launches"] == 1 and abs(rf["hbm_GBps_from_traffic"] - rf["traffic"] / (rf["avg_launch_ms"] * 1e-3) / 1e9) < 1e-3 * rf["hbm_GBps_from_traffic"]
    else:
        # a box where the profiler cannot run (no rocprofv3, a run that is itself profiled): the line says so and falls
        # back to the committed table or null -- an optional measurement never costs the line
        import warnings
        warnings.warn(f"live PMC passes did not run here: {lp['error']}")
        for key in ("roofline", "roofline_narrow"):
            rf = full[key]
            assert lp["kernels"] is None and "live pass" in (rf.get("traffic_source") or rf.get("traffic_note") or "")
    assert full["threshold_bound"]["hits_identical_to_headline"] and full["clustered"]["hits_identical"]
    assert line["roofline"]["frac"] < 1.0 and full["threshold_bound"]["roofline"]["frac"] < 1.0
    assert full["clustered"]["fetch_all_rows"]["hits"] > 20 * line["hits"]
    ur = full["unique_rows"]
    assert 0 < ur["unique_rows_x_row_bytes"] <= ur["algorithmic_bytes_per_step"] and "perfect row reuse" in ur["label"]
    ga = full["argannot"]
    for rep in ("x1", "x8"):
        g = ga[rep]
        assert g["hits_identical"] and g["kmers"] == 1594532 * int(rep[1:]) and g["fetch_all_rows"]["hits"] >= g["planted_pairs_at_or_above_threshold"] > 0
        assert {k_.split("P=")[1].split(",")[0] for k_ in g["fetch_all_rows"]["scan_launches"]} == {"10", "13"}
        assert 0 < g["fetch_all_rows"]["roofline"]["frac"] < 1.0
    # the device-tensor gather path of the RCCL runs (D2D copy of the ordered records, read-back), one rank
    forced = tmp_path / "forced.npy"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common + ["--dump-hits", str(forced), "--no-clustered", "--legs-out", ""],
                       capture_output=True, env=dict(env, BENCH_FORCE_GATHER="1"))
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    assert np.array_equal(np.load(one), np.load(forced))
    two = tmp_path / "two.npy"
    env2 = dict(env, BENCH_DIST_BACKEND="gloo", BENCH_SHARE_GPU="1")
    # every N > 1 line is self-sufficient: the CPU path is timed in the same run (rank 0, the other rank sleeps on a store
    # key), `roofline` is there with `traffic` null AND the reason, the ranks are counted
    small_cpu = [a_ for a_ in common if a_ not in ("--no-cpu-baseline", "--no-live-pmc")] + ["--cpu-target-s", "0.6", "--cpu-sample-gb", "0.2"]
    legs2 = tmp_path / "legs_two.json"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29547", os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--clustered-multi"] + small_cpu + ["--dump-hits", str(two), "--legs-out", str(legs2)],
                       capture_output=True, env=env2)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    a, b = np.load(one), np.load(two)
    assert len(a) > 50 and np.array_equal(a, b)
    line2, full2 = bench_record(r.stdout, legs2)
    part = line2["participants"]
    assert part["ranks"] == 2 and len(part["rank_ms_per_step"]) == 2 and part["rank_devices"] == [0, 0]
    _check_multi_rank_line(line2, full2, 2, live_pmc=True)
    assert sum(part["rank_batches"]) == 64 and min(part["rank_batches"]) >= 1
    assert abs(max(part["rank_ms_per_step"]) - line2["ms_per_step"]) < 1e-3          # the job's step is the slowest rank's
    assert abs(max(full2["participants"]["rank_ms_per_step"]) - line2["ms_per_step"]) < 1e-9
    # BASELINE configs[3]: with 8 ranks (here: 2, forced) all 305 batches are sharded over the ranks and searched with
    # the per-step gather -- the records equal those of one rank that holds the whole (scaled) collection
    fullc = tmp_path / "fullc.npy"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29549", os.path.join(ROOT, "bench.py"),
                        "--gpus", "2"] + common + ["--dump-full-hits", str(fullc), "--legs-out", str(tmp_path / "legs_fc.json")],
                       capture_output=True, env=dict(env2, BENCH_FULL_MIN_WORLD="2"))
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    _, full_fc = bench_record(r.stdout, tmp_path / "legs_fc.json")
    fc = full_fc["full_collection"]
    assert fc["hits_identical"] and sum(fc["rank_batches"]) == 305 and len(fc["fetch_all_rows"]["rank_ms_per_step"]) == 2
    assert fc["fetch_all_rows"]["hits"] >= fc["planted_pairs_at_or_above_threshold"] > 0
    whole = tmp_path / "whole.npy"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "full", "--no-l31", "--no-clustered"] + common +
                       ["--dump-hits", str(whole), "--legs-out", ""], capture_output=True, env=env)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    assert np.array_equal(np.load(fullc), np.load(whole))
    # one rank under the launcher the driver uses = the plain invocation (same records, a bench line of the same shape)
    solo = tmp_path / "solo.npy"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", "29548", os.path.join(ROOT, "bench.py"),
                        "--gpus", "1", "--no-clustered"] + common + ["--dump-hits", str(solo), "--legs-out", str(tmp_path / "legs_solo.json")],
                       capture_output=True, env=env)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    line1, full1 = bench_record(r.stdout, tmp_path / "legs_solo.json")
    assert np.array_equal(np.load(solo), a) and line1["n_gpus"] == 1 and line1["hits"] == line["hits"]
    assert line1["participants"]["rank_devices"] == [0] and full1["l31"]["config2"]["hit_records"] > 0


def test_hit_buffer_overflow_reruns_with_the_exact_size(pm, oracle):
    """more records than the first hit buffer holds (threshold 0: every document of every query):
    the search is queued again with a buffer of the exact size; later searches size theirs from it"""
    rng = np.random.default_rng(82)
    n_docs, S, nq = 4000, 1500, 300
    queries = [(f"o{i}", rand_seq(rng, 40)) for i in range(nq)]
    index, fasta, _ = build_case(oracle, rng, n_docs, S, queries, density=0.3)
    ix = pm.Index.load_mem(index)
    q = pm.Queries(fasta)
    exp = oracle.query_file(index, fasta, 0.0)
    for _ in range(2):
        res = pm.search([ix], q, 0.0)
        st = res.stats
        assert st.n_hits == nq * n_docs and st.n_runs == nq and st.n_records == nq * (n_docs + 1) > (1 << 20)
        hits = res.hits()
        assert len(hits) == nq * n_docs                       # no list was cut: no count records on the host
        assert pm.format_hits(ix, q, hits) == exp
    a = pm.search_async([ix], q, 0.0)
    b = pm.search_async([ix], q, 0.0, nb_best_hits=5)
    assert pm.format_hits(ix, q, b.hits(), nb_best_hits=5) == pm.query_text(ix, fasta, 0.0, nb_best_hits=5)
    assert pm.format_hits(ix, q, a.hits()) == exp


@pytest.mark.parametrize("n_docs", [13, 100, 300, 664, 4000, 9001])
def test_wide_query_form_is_bit_identical(pm, oracle, n_docs):
    """few long queries: several lane groups of a workgroup share one query (partial counts added
    through LDS).  Forced on (1), off (2) and automatic (0) give the same records, equal to the oracle."""
    rng = np.random.default_rng(900 + n_docs)
    lens = [150, 158, 200, 700, 1053, 1054, 1500, 4000, 9000] + ([70000] if n_docs in (100, 4000) else []) + \
        ([1048700] if n_docs == 100 else [])                    # 20- and 24-plane classes
    queries = [(f"w{i}", rand_seq(rng, L)) for i, L in enumerate(lens)] + [(f"x{i}", rand_seq(rng, 1300)) for i in range(9)]
    plant = []
    for qi in range(len(queries)):
        for j, fr in enumerate((1.0, 0.9, 0.75, 0.7, 0.69, 0.3)):
            plant.append((qi, int(rng.integers(0, n_docs)), fr))
    index, fasta, _ = build_case(oracle, rng, n_docs, 5000, queries, density=0.1, plant=plant)
    ix = pm.Index.load_mem(index, layout=int(rng.integers(0, 3)))
    other = pm.Index.synth(2, 200, 3000)                     # a second, narrow batch: the mixed-width launch
    q = pm.Queries(fasta)
    try:
        got = {}
        for mode in (2, 1, 0):
            pm.set_option("wide_query", mode)
            for thr in (0.7, 0.0):
                for n in (0, 3):
                    res = pm.search([ix, other], q, thr, nb_best_hits=n)
                    got[(mode, thr, n)] = res.hits()
                    if mode == 1:
                        kinds = {L["kernel"] for L in res.launches()}
                        assert any(",WQ>" in k for k in kinds) and any("P=7," in k and ",WQ" not in k for k in kinds), kinds
        for thr in (0.7, 0.0):
            for n in (0, 3):
                assert np.array_equal(got[(1, thr, n)], got[(2, thr, n)]), (thr, n)
                assert np.array_equal(got[(0, thr, n)], got[(2, thr, n)]), (thr, n)
            assert pm.format_hits(ix, q, got[(1, thr, 0)], slot=0) == oracle.query_file(index, fasta, thr)
        # the same form with the steps of every query shared by 2, 5 and 16 workgroups (partial planes meet in global
        # slabs, the last workgroup to arrive adds them up): same records again, run twice over the same slabs
        pm.set_option("wide_query", 1)
        for split in (2, 5, 16, 5):
            pm.set_option("wide_query_split", split)
            for thr in (0.7, 0.0):
                for n in (0, 3):
                    res = pm.search([ix, other], q, thr, nb_best_hits=n)
                    assert np.array_equal(res.hits(), got[(2, thr, n)]), (split, thr, n)
    finally:
        pm.set_option("wide_query", 0)
        pm.set_option("wide_query_split", 0)
    with pytest.raises(pm.PMError):
        pm.set_option("wide_query", 3)
    with pytest.raises(pm.PMError):
        pm.set_option("wide_query_split", 1000)


def test_long_queries_shared_across_workgroups_at_scale(pm):
    """the cross-workgroup hand-off of the wide-query form under real load: six 300-kbp queries against a wide and a
    narrow 661k-shaped batch (1.5 GB of signatures, planted hits around the threshold); the records with the steps of a
    query shared by 7 and 24 workgroups, and with the automatic choice, equal the unsplit form's -- three rounds over the
    same slabs, every record compared"""
    from phylign_amd import workload as W
    fasta, _ = W.make_queries(6, 300_030, seed=77)
    q = pm.Queries(fasta)
    hashes = q.hash_terms(1, 1).reshape(6, 300_000)
    ixs = [pm.Index.synth(5, 4000, 2_500_000), pm.Index.synth(6, 300, 9_000_000)]
    rng = np.random.default_rng(4)
    for ix in ixs:
        info = ix.info
        rows, docs = [], []
        for qi in range(6):
            for d, frac in zip(rng.choice(info.n_docs, size=5, replace=False), (1.0, 0.75, 0.7001, 0.6999, 0.5)):
                m = int(np.ceil(frac * 300_000))
                rows.append(hashes[qi, :m] % np.uint64(info.signature_size))
                docs.append(np.full(m, d, dtype=np.uint32))
        ix.plant(np.concatenate(rows), np.concatenate(docs))
    try:
        pm.set_option("wide_query_split", 1)
        base = pm.search(ixs, q, 0.7, nb_best_hits=100)
        ref = base.hits()
        assert len(ref) >= 2 * 6 * 3 and all(",WQ>" in L["kernel"] for L in base.launches())
        for rnd in range(3):
            for split in (7, 0, 24):
                pm.set_option("wide_query_split", split)
                got = pm.search(ixs, q, 0.7, nb_best_hits=100).hits()
                assert np.array_equal(got, ref), (rnd, split)
    finally:
        pm.set_option("wide_query_split", 0)


def test_query_parts_of_an_index_partition_the_search(pm, oracle):
    """pm_search_async_parts: an index searched with the query shares [0, a), [a, b), [b, den) -- as the ranks that
    hold a copy of it would -- yields, put together, exactly the records of the whole search; other indexes of the same
    search are untouched.  Narrow rows (mixed-width launch), one width only (uniform launch), 512-byte rows, rows wider
    than 1024 bytes (column slabs), a compact-like second hash function, queries of four counter-width classes, both
    scan modes, the n-best cut, and the wide-query form."""
    rng = np.random.default_rng(4242)
    lens = [31, 33, 36] * 5 + [150] * 41 + [158, 200, 700] * 6 + [1053, 1500, 4000] * 3
    rng.shuffle(lens)
    queries = [(f"p{i}", rand_seq(rng, int(L))) for i, L in enumerate(lens)]
    nq = len(queries)
    cases = []
    for n_docs, S, nh in ((200, 3000, 1), (664, 4000, 1), (1000, 2000, 2), (4000, 1500, 1), (9001, 600, 1)):
        plant = [(qi, int(rng.integers(0, n_docs)), fr) for qi in range(nq) for fr in (1.0, 0.8, 0.7, 0.69)]
        index, fasta, _ = build_case(oracle, rng, n_docs, S, queries, num_hashes=nh, density=0.1, plant=plant)
        cases.append((index, pm.Index.load_mem(index, layout=int(rng.integers(0, 3)))))
    q = pm.Queries(fasta)
    den = 1024
    cuts = [(0, den), (0, 1, den), (0, 300, 777, den), (0, 512, 513, den), (0, den - 1, den)]

    def run(ixs, parts, thr, n):
        res = pm.search_async(ixs, q, thr, nb_best_hits=n, parts=parts)
        h = res.hits()
        launches = res.launches()
        res.free()
        return h, launches
    try:
        for bound in (1, 0):
            pm.set_option("threshold_bound", bound)
            for wq in (0, 1):
                pm.set_option("wide_query", wq)
                for sel in ([0, 1, 2], [1], [3], [4], [0, 1, 2, 3, 4]):           # mixed / uniform narrow / 512 B / slabs / all
                    ixs = [cases[i][1] for i in sel]
                    for thr, n in ((0.7, 0), (0.7, 2), (0.0, 0)):
                        whole, lw = run(ixs, None, thr, n)
                        alg_whole = sum(L["algorithmic_bytes"] for L in lw)
                        for target in range(len(ixs)):
                            for cut in cuts[1:] if len(ixs) < 5 else cuts[2:3]:
                                got, alg = [], 0
                                for lo, hi in zip(cut[:-1], cut[1:]):
                                    parts = [None] * len(ixs)
                                    parts[target] = (lo, hi, den)
                                    h, ll = run(ixs, parts, thr, n)
                                    got.append(h[h["slot"] == target])
                                    alg += sum(L["algorithmic_bytes"] for L in ll)
                                    others = h[h["slot"] != target]
                                    assert np.array_equal(others, whole[whole["slot"] != target])
                                merged = pm.sort_hits(np.ascontiguousarray(np.concatenate(got)))
                                assert np.array_equal(merged, whole[whole["slot"] == target]), (bound, wq, sel, thr, n, target, cut)
                                # bytes: the target's rows are counted once over its parts, the others' once per search
                                k = len(cut) - 1
                                tb = int(ixs[target].info.row_bytes) * int(ixs[target].info.num_hashes) * q.count()[1]
                                assert alg == alg_whole * k - tb * (k - 1)
        # the text of the put-together records is the oracle's
        index, ix = cases[1]
        a, _ = run([ix], [(0, 400, den)], 0.7, 0)
        b, _ = run([ix], [(400, den, den)], 0.7, 0)
        both = pm.sort_hits(np.ascontiguousarray(np.concatenate([a, b])))
        assert pm.format_hits(ix, q, both) == oracle.query_file(index, fasta, 0.7)
        with pytest.raises(pm.PMError):
            pm.search_async([ix], q, 0.7, parts=[(5, 4, den)])
        with pytest.raises(pm.PMError):
            pm.search_async([ix], q, 0.7, parts=[(0, den + 1, den)])
    finally:
        pm.set_option("threshold_bound", 1)
        pm.set_option("wide_query", 0)


def test_bench_plain_invocation_with_eight_ranks_equals_one_rank(pm, tmp_path):
    """`python bench.py --gpus 8` exactly as the driver types it (no launcher): the script starts its own 8 ranks --
    here sharing the one GPU over gloo -- with a few small batches resident on two ranks that share their queries (--replicas);
    the gathered records equal the one-rank run's, and so do those of the 305-batch full_collection leg"""
    env = dict(os.environ, PYTHONPATH=ROOT)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    common = ["--steps", "2", "--warmup", "1", "--rows-divisor", "400", "--queries", "3000", "--no-cpu-baseline", "--no-live-pmc"]
    one = tmp_path / "one.npy"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common + ["--dump-hits", str(one), "--only-headline"],
                       capture_output=True, env=env, cwd=tmp_path)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    line1, _ = bench_record(r.stdout, tmp_path / "bench_legs.json")          # the default side file: bench_legs.json in the cwd
    assert line1["n_gpus"] == 1 and line1["cpu_baseline"] is None
    eight, fullc, legs8 = tmp_path / "eight.npy", tmp_path / "fullc.npy", tmp_path / "legs_eight.json"
    env8 = dict(env, BENCH_DIST_BACKEND="gloo", BENCH_SHARE_GPU="1", BENCH_FULL_MIN_WORLD="8")
    # (the self-launched ranks carry PHYLIGN_LAUNCHER_PID: rank 0's PMC children must not inherit it, or they end themselves)
    small_cpu = [a_ for a_ in common if a_ not in ("--no-cpu-baseline", "--no-live-pmc")] + ["--cpu-target-s", "0.6", "--cpu-sample-gb", "0.2"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--replicas"] + small_cpu +
                       ["--dump-hits", str(eight), "--dump-full-hits", str(fullc), "--legs-out", str(legs8)], capture_output=True, env=env8)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    # ONE JSON line, from rank 0, the last thing on stdout, under the size cap with all 8 ranks AND the full_collection leg run
    line, full = bench_record(r.stdout, legs8)
    assert len(r.stdout.decode().rstrip("\n").splitlines()[-1]) < 6144
    _check_multi_rank_line(line, full, 8, live_pmc=True)
    assert full["live_pmc"]["error"] is None or "rocprofv3" in full["live_pmc"]["error"], full["live_pmc"]
    part = line["participants"]
    assert line["n_gpus"] == 8 and part["ranks"] == 8 and len(part["rank_ms_per_step"]) == 8
    shared = full["config"]["batches_on_two_ranks"]
    assert 1 <= len(shared) <= 7 and all(len(s["query_shares"]) == 2 for s in shared) and line["config"]["batches_on_two_ranks"] == len(shared)
    assert sum(part["rank_batches"]) == 64 + len(shared) and min(part["rank_batches"]) >= 1
    assert full["threshold_bound"]["hits_identical_to_headline"]
    a, b = np.load(one), np.load(eight)
    assert len(a) > 50 and np.array_equal(a, b)
    fc = full["full_collection"]
    assert fc["hits_identical"] and sum(fc["rank_batches"]) == 305 and len(fc["fetch_all_rows"]["rank_ms_per_step"]) == 8
    whole = tmp_path / "whole.npy"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "full", "--only-headline"] + common +
                       ["--dump-hits", str(whole), "--legs-out", ""], capture_output=True, env=env)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    assert np.array_equal(np.load(fullc), np.load(whole))
    # whole batches only (the default): the same records again
    plain = tmp_path / "plain.npy"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--only-headline"] + common +
                       ["--dump-hits", str(plain), "--legs-out", ""], capture_output=True, env=env8)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    assert np.array_equal(np.load(plain), a)


def test_one_rank_walks_the_rccl_path(pm, oracle, tmp_path):
    """BENCH_FORCE_DIST / PHYLIGN_FORCE_DIST: a process group of ONE rank over RCCL (backend nccl) -- the collectives of
    the N > 1 path run on device tensors on real hardware: the packed gather of the ordered records with its alternating
    send buffers, all_gather of the timings, all_reduce, barrier; the records equal the plain run's.  (Two RCCL ranks
    cannot share one GPU, so this is as far as a 1-GPU box can take the RCCL plumbing; rank layouts are covered with
    gloo.)"""
    env = dict(os.environ, PYTHONPATH=ROOT)
    for k in ("WORLD_SIZE", "RANK", "BENCH_DIST_BACKEND", "PHYLIGN_DIST_BACKEND"):
        env.pop(k, None)
    common = ["--steps", "3", "--warmup", "1", "--rows-divisor", "400", "--queries", "3000", "--no-cpu-baseline", "--only-headline"]
    plain, forced = tmp_path / "plain.npy", tmp_path / "forced.npy"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common + ["--dump-hits", str(plain), "--legs-out", ""], capture_output=True, env=env)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    legs = tmp_path / "legs_forced.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common + ["--dump-hits", str(forced), "--legs-out", str(legs)],
                       capture_output=True, env=dict(env, BENCH_FORCE_DIST="1"))
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    line, full = bench_record(r.stdout, legs)             # RCCL's banner (C stdio) comes out BEFORE the line, never behind it
    part = line["participants"]
    assert part["backend"] == "nccl" and part["rccl_ranks"] == 1 and part["rank_devices"] == [0]
    assert full["participants"]["rank_host_ms"][0]["hit_gather"] > 0
    assert np.array_equal(np.load(plain), np.load(forced)) and len(np.load(plain)) > 50
    # the stage: merge export -> RCCL gather -> rank 0 emits
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_cli import _check_stage_outputs, _stage_fixture
    names, indexes, fasta = _stage_fixture(oracle, tmp_path)
    r = subprocess.run([sys.executable, "-m", "phylign_amd.match_stage", "--batches", str(tmp_path / "batches.txt"),
                        "--cobs-dir", str(tmp_path / "cobs"), "--sizes", str(tmp_path / "sizes.txt"), "--queries", str(tmp_path / "Q.fa"),
                        "--out-dir", str(tmp_path / "03_match"), "--nb-best-hits", "3",
                        "--filter-out", str(tmp_path / "04_filter" / "Q.fa")], capture_output=True, env=dict(env, PHYLIGN_FORCE_DIST="1"))
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    _check_stage_outputs(oracle, tmp_path, names, indexes, fasta, 3)


@pytest.mark.parametrize("n_docs,S,layout", [(100, 70001, 1), (664, 50021, 2), (2300, 20011, 2), (4000, 150001, 2), (9000, 3001, 2)])
def test_saved_index_file_is_the_resident_index(pm, oracle, tmp_path, n_docs, S, layout):
    """the measurement aid that writes a resident index back as a .cobs_classic file (tools/e2e_*.py put 661k-shaped
    files on disk with it): the file is byte for byte what the checker builds from the same header, names and rows, and
    loading it gives the same matrix (rows packed on the device, several 64 MiB chunks for the larger shapes)"""
    from phylign_amd import bench_aids
    ix = pm.Index.synth(5, n_docs, S, seed=661, layout=layout)
    path = str(tmp_path / "b.cobs_classic")
    bench_aids.index_save(ix, path)
    with open(path, "rb") as f:
        saved = f.read()
    assert saved == np.asarray(_file_of(oracle, ix)).tobytes()
    ix2 = pm.Index.load_file(path)
    assert _names(ix2) == _names(ix) and np.array_equal(np.asarray(ix2.read_rows(0, S)), np.asarray(ix.read_rows(0, S)))
    ix.free(); ix2.free()


def test_index_allocation_reclaims_the_librarys_idle_pools(pm):
    """device buffers of released query sets (and of finished searches) wait in pools the stage budget does not
    count.  A signature matrix that only fits once they are given back is allocated all the same: the out-of-memory path
    empties the pools and tries once more (device_malloc_reclaim, pm_runtime.cpp) instead of reporting PM_ENOMEM."""
    from phylign_amd import workload as W
    MB = 1 << 20
    pm.set_option("release_query_pool", 1)
    names = [f"d{i}" for i in range(4096)]                       # 512-byte rows in either layout
    sets = []
    for n in (300_000, 360_000, 420_000):                        # 288 + 346 + 403 MB of hashes, plus sequences and descriptors
        fasta, _ = W.make_queries(n, 150, seed=n)
        q = pm.Queries(fasta, term_size=31)
        q.hash_terms(1, 1)                                       # uploads the set and hashes it on the device
        sets.append(q)
    held = sum(q.device_bytes()[0] for q in sets)
    assert held > 1000 * MB
    free_before = pm.device_info()["hbm_free"]
    for q in sets:
        q.release_device()                                       # -> pooled, not freed
    free_pooled = pm.device_info()["hbm_free"]
    assert free_pooled - free_before < held // 4                 # the pool still holds (most of) it
    filler = pm.Index.create(names, (free_pooled - 1024 * MB) // 512, layout=2)      # leaves 1 GB: far from any allocation-granularity effect
    left = pm.device_info()["hbm_free"]
    assert left < 1100 * MB
    want = 1700 * MB                                             # more than is free, less than free + pooled
    assert left < want < left + held - 100 * MB
    try:
        ix = pm.Index.create(names, want // 512, layout=2)        # succeeds only because the pools were emptied
        assert ix.info.device_bytes >= want - 512
        ix.free()
        # and a request that cannot fit either way still fails with the out-of-memory code, cleanly
        # (sized from what is free NOW: the reclaim above also returned the hit buffers earlier tests left in their pool)
        too_much = pm.device_info()["hbm_free"] + 16384 * MB
        with pytest.raises(pm.PMError) as e:
            pm.Index.create(names, too_much // 512, layout=2)
        assert e.value.code == -3                                 # PM_ENOMEM
    finally:
        filler.free()
        for q in sets:
            q.free()
    assert pm.device_info()["hbm_free"] > free_before

"""CPU suite: the driver-facing line of bench.py.  BENCH_r05.json came back with "parsed": null after the single JSON
line had grown to 28.6 KB; since round 6 the last stdout line is a small strict-JSON object (headline + roofline +
cpu_baseline + who took part) and the auxiliary legs go to a side file.  These tests rebuild the line from RECORDED
round-5 records (whole dicts, as bench.py assembled them on the GPU box) and from a synthetic worst case."""
import copy
import io
import json
import os
import sys

import pytest

from helpers import BENCH_LEG_KEYS, check_bench_line

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RECORDED = ["profiles/r05/bench_n1_driver_flags.json",                        # the driver's flags, 1 rank, every leg: 28.6 KB
            "profiles/r05/bench_plain_gpus8_gloo_shared_gpu_fullsize.json",   # 8 ranks
            "profiles/r05/bench_plain_gpus8_gloo_shared_gpu.json",
            "profiles/r04/bench_n1_driver_flags.json"]


def _bench():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import bench
    return bench


@pytest.mark.parametrize("rel", RECORDED)
def test_line_of_a_recorded_run_is_small_strict_and_complete(rel):
    path = os.path.join(ROOT, rel)
    if not os.path.exists(path):
        pytest.skip(f"{rel} not in this checkout")
    bench = _bench()
    full = json.load(open(path))
    text = json.dumps(bench.compact_line(full, "/somewhere/bench_legs.json"), allow_nan=False)
    assert len(text.encode()) <= bench.LINE_MAX_BYTES < 8192
    line = check_bench_line(text)
    # the headline is carried digit for digit; the roofline and the CPU baseline are the record's own
    assert line["value"] == full["value"] and line["ms_per_step"] == full["ms_per_step"]
    assert line["metric"] == full["metric"] and line["n_gpus"] == full["n_gpus"]
    assert abs(line["roofline"]["frac"] - full["roofline"]["frac"]) < 1e-6 * full["roofline"]["frac"]
    assert line["roofline"]["kernel"] == full["roofline"]["kernel"] and line["roofline"]["traffic"] == full["roofline"]["traffic"]
    if full.get("cpu_baseline"):
        cb = line["cpu_baseline"]
        assert cb["kind"] == "port" and cb["cores"] == full["cpu_baseline"]["cores"] and isinstance(cb["sample"], str)
        assert abs(cb["value"] - full["cpu_baseline"]["value"]) < 1e-6 * cb["value"]
        assert abs(line["gpu_over_cpu"] - line["value"] / cb["value"]) < 1e-5 * line["gpu_over_cpu"]
    assert line["participants"]["ranks"] == full["n_gpus"]


def _worst_case(bench, ranks=8):
    """a record with every string at a silly length, 8 ranks, two shared batches per rank and every leg populated"""
    full = json.load(open(os.path.join(ROOT, RECORDED[1]))) if os.path.exists(os.path.join(ROOT, RECORDED[1])) else None
    if full is None:
        pytest.skip("no recorded 8-rank record in this checkout")
    full = copy.deepcopy(full)
    full["config"]["workload"] = "w" * 5000
    full["config"]["batches_on_two_ranks"] = [{"batch": "b" * 64, "index_GB": 1.0, "query_shares": {"0": [0, 1, 2], "1": [1, 2, 2]}}] * 16
    full["cpu_baseline"]["sample"] = "s" * 5000
    for k in ("roofline", "roofline_narrow"):
        full[k]["traffic"] = None
        full[k]["traffic_note"] = "t" * 5000
    for k in BENCH_LEG_KEYS:
        full[k] = {"filler": "x" * 20000}
    full["participants"]["rank_ms_per_step"] = [123456.789012345] * ranks
    return full


def test_line_stays_under_the_cap_in_the_worst_case():
    bench = _bench()
    full = _worst_case(bench)
    text = json.dumps(bench.compact_line(full, "/" + "p" * 200 + "/bench_legs.json"), allow_nan=False)
    assert len(text.encode()) <= bench.LINE_MAX_BYTES
    line = check_bench_line(text)
    assert line["config"]["batches_on_two_ranks"] == 16 and line["config"]["workload"].endswith("...")


def test_emit_writes_the_side_file_first_and_the_line_last(tmp_path, capfd):
    bench = _bench()
    path = os.path.join(ROOT, RECORDED[0])
    if not os.path.exists(path):
        pytest.skip("no recorded 1-rank record in this checkout")
    full = json.load(open(path))
    legs = tmp_path / "bench_legs.json"
    print("[some library banner]")                                    # earlier stdout noise stays in front of the line
    text = bench.emit(full, str(legs))
    out = capfd.readouterr().out
    assert out.endswith(text + "\n") and out.count("\n{") == 1
    line = check_bench_line(out.rstrip("\n").splitlines()[-1])
    assert line["legs_file"] == str(legs)
    whole = json.load(open(legs))
    assert whole == full and all(k in whole for k in ("threshold_bound", "unique_rows", "argannot", "clustered", "l31", "full_shard"))
    assert not [p for p in os.listdir(tmp_path) if p.endswith(".tmp")]
    # a side file that cannot be written costs the side file only
    logged = []
    text2 = bench.emit(full, str(tmp_path / "no_such_dir" / "legs.json"), logged.append)
    assert json.loads(text2)["legs_file"] is None and logged and "not written" in logged[0]
    capfd.readouterr()
    # no side file asked for
    assert json.loads(bench.emit(full, ""))["legs_file"] is None
    capfd.readouterr()


def test_a_non_finite_number_never_reaches_stdout(tmp_path, capfd):
    bench = _bench()
    path = os.path.join(ROOT, RECORDED[0])
    if not os.path.exists(path):
        pytest.skip("no recorded 1-rank record in this checkout")
    full = json.load(open(path))
    full["roofline"]["frac"] = float("nan")
    with pytest.raises(ValueError):
        bench.emit(full, str(tmp_path / "legs.json"))
    assert "{" not in capfd.readouterr().out


def test_bench_summary_reads_line_and_side_file(tmp_path):
    """tools/bench_summary.py prints the round's quoted figures from the side file (whole record) or from a line"""
    import subprocess
    bench = _bench()
    path = os.path.join(ROOT, RECORDED[0])
    if not os.path.exists(path):
        pytest.skip("no recorded 1-rank record in this checkout")
    full = json.load(open(path))
    legs = tmp_path / "bench_legs.json"
    json.dump(full, open(legs, "w"))
    line_file = tmp_path / "line.json"
    line_file.write_text(json.dumps(bench.compact_line(full, str(legs))) + "\n")
    a = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_summary.py"), str(legs)], capture_output=True, text=True)
    b = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_summary.py"), str(line_file)], capture_output=True, text=True)
    assert a.returncode == 0 and b.returncode == 0, a.stderr + b.stderr
    assert "argannot x8" in a.stdout and a.stdout == b.stdout           # the line names its side file: same figures


def test_live_pmc_parser_on_recorded_counter_files():
    """bench.pmc_bytes_per_launch (the parser of the live PMC passes) on counter files recorded on an MI355X
    (tests/golden/pmc: round 4, threshold-bound scan, two dispatches per kernel): per launch, bytes = 128 x RDREQ_128B +
    64 x RDREQ_64B + 32 x RDREQ_32B + 1024 x WRITE_SIZE, the figures of profiles/r04/pmc_bench_n1_bound.txt"""
    bench = _bench()
    gold = os.path.join(ROOT, "tests", "golden", "pmc")
    t = bench.pmc_bytes_per_launch([os.path.join(gold, "r04_bound_rdreq_counter_collection.csv"),
                                    os.path.join(gold, "r04_bound_write_counter_collection.csv")])
    assert set(t) == {"k_scan<G=32,P=7,NH1>", "k_scan<G=mixed,P=7,NH1>"}                 # k_hash_terms rows are not scan kernels
    wide, narrow = t["k_scan<G=32,P=7,NH1>"], t["k_scan<G=mixed,P=7,NH1>"]
    assert wide["launches"] == narrow["launches"] == 2
    assert abs(wide["hbm_bytes_per_launch"] - (128 * 817172590.0 + 64 * 47586.5 + 1024 * 23952.4)) < 1024
    assert abs(narrow["hbm_bytes_per_launch"] - (128 * 247918092.0 + 64 * 11205.5 + 1024 * 4419.6)) < 1024
    assert wide["read_requests"]["TCC_EA0_RDREQ_32B_sum"] == 0 and wide["write_bytes"] == int(1024 * (511.0625 + 47393.78125) / 2)
    # one pass missing (no WRITE_SIZE file): no figure rather than a partial one
    assert bench.pmc_bytes_per_launch([os.path.join(gold, "r04_bound_rdreq_counter_collection.csv")]) == {}
    assert bench.scan_kernel_name("void pm::k_scan<0, 13, true, true>(pm::ScanArgs)") == "k_scan<G=mixed,P=13,NH1,WQ>"
    assert bench.scan_kernel_name("void pm::k_scan<64, 10, false, false>(pm::ScanArgs)") == "k_scan<G=64,P=10,NHn>"
    assert bench.scan_kernel_name("pm::k_permute_runs(...)") is None


def test_live_pmc_is_skipped_where_it_cannot_run(monkeypatch):
    bench = _bench()
    monkeypatch.setenv("ROCPROFILER_SOMETHING", "1")
    assert bench.under_a_profiler()
    monkeypatch.delenv("ROCPROFILER_SOMETHING")
    monkeypatch.setenv("LD_PRELOAD", "/opt/rocm/lib/librocprofiler-sdk-tool.so")
    assert bench.under_a_profiler()
    monkeypatch.delenv("LD_PRELOAD")
    if not bench.under_a_profiler():
        import shutil
        if shutil.which("rocprofv3") is None and not os.path.exists("/opt/rocm/bin/rocprofv3"):
            table, why, took = bench.live_pmc_traffic([], 60.0, print)
            assert table is None and "not installed" in why

"""CPU suite: the driver-facing line of bench.py.  BENCH_r05.json came back with "parsed": null after the single JSON
line had grown to 28.6 KB; since round 6 the last stdout line is a small strict-JSON object (headline + roofline +
cpu_baseline + who took part) and the auxiliary legs go to a side file.  These tests rebuild the line from RECORDED
round-5 records (whole dicts, as bench.py assembled them on the GPU box) and from a synthetic worst case."""
import copy
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


def _fake_rocprofv3(tmp_path, body):
    d = tmp_path / "bin"
    d.mkdir(exist_ok=True)
    exe = d / "rocprofv3"
    exe.write_text("#!/usr/bin/env bash\n" + body)
    exe.chmod(0o755)
    return str(d)


def test_live_pmc_passes_with_a_stand_in_profiler(tmp_path, monkeypatch):
    """bench.live_pmc_traffic end to end on CPU with a stand-in `rocprofv3` on PATH: two child runs (one counter group each),
    the counter files they leave are parsed, the scratch directory is removed; a pass that exits non-zero, writes nothing or
    overruns the time budget yields (None, reason) -- never an exception, never a partial figure -- and the process group the
    call started is gone afterwards.  No launcher / process-group variable of the calling rank reaches the children."""
    bench = _bench()
    for k in list(os.environ):
        if k.startswith(("ROCPROF", "ROCP_", "ROCTX")):
            monkeypatch.delenv(k)
    monkeypatch.delenv("LD_PRELOAD", raising=False)
    gold = os.path.join(ROOT, "tests", "golden", "pmc")
    seen = tmp_path / "seen.txt"
    # the stand-in: finds `-d DIR`, copies the recorded file of the counter group it was asked for, records argv + env
    ok_body = f'''
args=("$@"); dir=""
for ((i=0; i<${{#args[@]}}; i++)); do [ "${{args[$i]}}" = "-d" ] && dir="${{args[$((i+1))]}}"; done
mkdir -p "$dir/host/1"
if printf '%s\\n' "$@" | grep -q WRITE_SIZE; then cp {gold}/r04_bound_write_counter_collection.csv "$dir/host/1/pmc_counter_collection.csv"
else cp {gold}/r04_bound_rdreq_counter_collection.csv "$dir/host/1/pmc_counter_collection.csv"; fi
echo "ARGV $*" >> {seen}; echo "ENV launcher=${{PHYLIGN_LAUNCHER_PID:-none}} rank=${{RANK:-none}} world=${{WORLD_SIZE:-none}} force=${{BENCH_FORCE_DIST:-none}} cwd=$PWD tmp=$TMPDIR" >> {seen}
'''
    monkeypatch.setenv("PATH", _fake_rocprofv3(tmp_path, ok_body) + os.pathsep + os.environ["PATH"])
    monkeypatch.setenv("PHYLIGN_LAUNCHER_PID", "1")            # what a self-launched rank 0 carries
    monkeypatch.setenv("RANK", "0"); monkeypatch.setenv("WORLD_SIZE", "8"); monkeypatch.setenv("BENCH_FORCE_DIST", "1")
    logged = []
    table, why, took = bench.live_pmc_traffic(["--workload", "config3", "--emulate-world", "8", "--emulate-rank", "0"], 60.0, logged.append)
    assert why is None and set(table) == {"k_scan<G=32,P=7,NH1>", "k_scan<G=mixed,P=7,NH1>"} and took < 30
    assert abs(table["k_scan<G=32,P=7,NH1>"]["hbm_bytes_per_launch"] - (128 * 817172590.0 + 64 * 47586.5 + 1024 * 23952.4)) < 1024
    text = seen.read_text()
    assert text.count("ARGV") == 2 and text.count("--pmc TCC_EA0_RDREQ_sum") == 1 and text.count("--pmc WRITE_SIZE") == 1
    assert "--kernel-trace" in text and "--sys-trace" not in text and "--hip-trace" not in text         # the combination gpurun allows
    assert "--steps 1 --warmup 0 --no-cpu-baseline --only-headline --no-pipeline --no-live-pmc" in text and "--emulate-world 8 --emulate-rank 0" in text
    assert text.count("ENV launcher=none rank=none world=none force=none cwd=/tmp tmp=/tmp") == 2
    assert not [d for d in os.listdir("/tmp") if d.startswith("pm_bench_pmc_") and os.path.getmtime(os.path.join("/tmp", d)) > os.path.getmtime(str(seen)) - 60 and not os.listdir(os.path.join("/tmp", d))]
    # failures: each a reason, not an exception
    monkeypatch.setenv("PATH", _fake_rocprofv3(tmp_path, "echo boom >&2; exit 3\n") + os.pathsep + os.environ["PATH"])
    table, why, _ = bench.live_pmc_traffic([], 60.0, logged.append)
    assert table is None and "status 3" in why and "boom" in why
    monkeypatch.setenv("PATH", _fake_rocprofv3(tmp_path, "exit 0\n") + os.pathsep + os.environ["PATH"])
    table, why, _ = bench.live_pmc_traffic([], 60.0, logged.append)
    assert table is None and "no counter_collection.csv" in why
    pidfile = tmp_path / "grandchild.pid"
    monkeypatch.setenv("PATH", _fake_rocprofv3(tmp_path, f"sleep 600 & echo $! > {pidfile}; wait\n") + os.pathsep + os.environ["PATH"])
    import time
    t0 = time.time()
    table, why, took = bench.live_pmc_traffic([], 2.0, logged.append, min_pass_s=0.5)
    assert table is None and "did not finish" in why and time.time() - t0 < 15
    pid = int(pidfile.read_text())                                 # the stand-in's own child: ended with the session, by its exact id

    def gone(pid_):
        try:
            return open(f"/proc/{pid_}/stat").read().rsplit(")", 1)[1].split()[0] == "Z"
        except OSError:
            return True
    t0 = time.time()
    while not gone(pid) and time.time() - t0 < 5:
        time.sleep(0.05)
    assert gone(pid)
    table, why, _ = bench.live_pmc_traffic([], 1.0, logged.append)                                       # budget below one pass
    assert table is None and "time budget" in why
    # a run that is itself profiled never nests a profiler
    monkeypatch.setenv("ROCPROFILER_REGISTER_SOMETHING", "1")
    table, why, _ = bench.live_pmc_traffic([], 60.0, logged.append)
    assert table is None and "being profiled" in why

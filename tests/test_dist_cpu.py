"""Multi-GPU path on CPU: static batch sharding and the gather of hit records,
world_size 2 over gloo (the same code runs over RCCL on the GPUs)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from phylign_amd import workload as W


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir, swap=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from phylign_amd.dist import gather_hits
    shapes = W.select("config3")
    parts = W.assign_batches(shapes, world)
    bases = np.cumsum([0] + [len(p) for p in parts])
    # every rank fabricates deterministic "hits" for its own batches
    rng = np.random.default_rng(100 + rank)
    n = [37, 0][rank] if world == 2 else 5          # one rank sends nothing: empty send path
    if swap:
        n = [0, 37][rank]                           # the non-root rank overflows the packed buffer
    rec = np.zeros((n, 4), dtype=np.int32)
    rec[:, 0] = rng.integers(0, 1000, size=n)
    rec[:, 1] = rng.integers(0, 4000, size=n)
    rec[:, 2] = rng.integers(84, 121, size=n)
    rec[:, 3] = bases[rank] + rng.integers(0, len(parts[rank]), size=n)
    g = gather_hits(torch.from_numpy(rec), dst=0)
    g2 = gather_hits(torch.from_numpy(rec[::-1].copy()), dst=0)   # a second round on the same group
    np.save(os.path.join(outdir, f"sent{rank}.npy"), rec)
    # one-collective packed form, including the overflow path (cap smaller than rank 0's count)
    from phylign_amd.dist import PackedGather
    for cap, tag in ((64, "fit"), (10, "ovf")):
        pg = PackedGather(cap, "cpu")
        t = torch.from_numpy(rec)
        if n <= cap:
            pg.records_view()[:n] = t
            out = pg.gather(n)
        else:
            out = pg.gather(n, overflow=t)
        if rank == 0:
            np.save(os.path.join(outdir, f"packed_{tag}.npy"), out.numpy())
    if rank == 0:
        assert g is not None and g2 is not None
        np.save(os.path.join(outdir, "gathered.npy"), g.numpy())
        np.save(os.path.join(outdir, "gathered2.npy"), g2.numpy())
    else:
        assert g is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("swap", [False, True])
def test_gather_hits_world2_gloo(tmp_path, swap):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path), swap), nprocs=2, join=True)
    sent = [np.load(tmp_path / f"sent{r}.npy") for r in range(2)]
    got = np.load(tmp_path / "gathered.npy")
    assert got.shape == (37, 4) and np.array_equal(got, np.concatenate(sent))
    got2 = np.load(tmp_path / "gathered2.npy")
    assert np.array_equal(got2, np.concatenate([s[::-1] for s in sent]))
    for tag in ("fit", "ovf"):
        assert np.array_equal(np.load(tmp_path / f"packed_{tag}.npy"), np.concatenate(sent))


def test_assign_batches_is_a_balanced_partition():
    shapes = W.select("config3")
    assert len(shapes) == 64 and abs(sum(s.index_bytes for s in shapes) / 1e9 - 213.13) < 0.01
    assert sum(s.row_bytes for s in shapes) == 16285
    for world in (1, 2, 4, 8):
        parts = W.assign_batches(shapes, world)
        flat = sorted(i for p in parts for i in p)
        assert flat == list(range(64))
        loads = [sum(W.scan_cost(shapes[i]) for i in p) for p in parts]
        assert max(loads) - min(loads) <= 512          # at most one widest row apart
        assert max(loads) <= sum(loads) / world * 1.08
    with pytest.raises(MemoryError):
        W.assign_batches(shapes, 2, capacity_bytes=50 * 10**9)
    full = W.select("full")
    assert len(full) == 305 and sum(s.row_bytes for s in full) == 82741   # 82 741 B per k-mer
    parts = W.assign_batches(full, 8, capacity_bytes=int(288e9 * 0.85))
    assert max(sum(full[i].index_bytes for i in p) for p in parts) < 288e9 * 0.85
    assert [s.batch for s in W.select("small")] == ["actinobacillus_pleuropneumoniae__01", "aeromonas_salmonicida__01", "bacillus_anthracis__01"]


def test_workload_queries_and_plant_plan(oracle):
    fasta, seqs = W.make_queries(50, 150, seed=31)
    assert seqs.shape == (50, 150) and fasta.count(b">") == 50
    assert fasta.split(b"\n")[1] == seqs[0].tobytes()
    hashes = np.concatenate([oracle.create_hashes(seqs[i].tobytes(), 31, 1, 1) for i in range(50)])
    shapes = W.scale_shapes(W.select("small"), 1000)
    plan, sure = W.plant_plan(hashes, 50, 120, shapes, every=10, docs_per_query=6)
    assert sure == 5 * 4            # fractions 1.0, 0.9, 0.8, 0.7 reach the threshold; 0.65 and 0.6 do not
    for pos, (rows, docs) in plan.items():
        assert rows.max() < shapes[pos].signature_size and docs.max() < shapes[pos].n_docs and len(rows) == len(docs)


def test_loader_admission_is_in_submission_order_and_never_deadlocks():
    """match_stage.Admission: loaders are admitted by ticket, so later batches can never take the
    budget an earlier batch is waiting for (the consumer drains in submission order)."""
    import random
    import threading
    import time
    from concurrent.futures import ThreadPoolExecutor
    from phylign_amd.match_stage import Admission

    rng = random.Random(5)
    needs = [rng.choice([1, 2, 3, 9, 40]) for _ in range(60)]      # 40 > budget: admitted alone
    adm = Admission(budget=10)
    order, lock = [], threading.Lock()

    def load(ticket):
        time.sleep(rng.random() * 0.002)
        adm.acquire(ticket, needs[ticket])
        with lock:
            order.append(ticket)
            assert adm.resident <= 10 or adm.resident == needs[ticket]
        return ticket

    with ThreadPoolExecutor(max_workers=6) as pool:
        futs = [pool.submit(load, t) for t in range(len(needs))]
        for t, f in enumerate(futs):                # the consumer: strictly in submission order
            assert f.result(timeout=20) == t
            time.sleep(0.0005)
            adm.release(needs[t])
    assert order == sorted(order) and adm.resident == 0


def test_eight_way_partition_of_config3_is_balanced_on_measured_batch_times():
    """the static batch -> GPU map of bench.py --gpus 8 (workload.assign_batches on line-cost weights) priced with the
    per-batch scan times MEASURED on one MI355X (profiles/r03/per_batch_cost.tsv): slowest rank / mean <= 1.025 in
    both scan modes, for 2, 4 and 8 ranks -- the bound on strong-scaling efficiency that the partition itself costs"""
    import os
    from phylign_amd import workload as W
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cost = {}
    with open(os.path.join(root, "profiles", "r03", "per_batch_cost.tsv")) as f:
        head = f.readline().rstrip("\n").split("\t")
        for line in f:
            row = dict(zip(head, line.rstrip("\n").split("\t")))
            cost[row["batch"]] = (float(row["ms_fetch_all"]), float(row["ms_bound"]))
    shapes = W.select("config3")
    assert len(shapes) == 64 and all(s.batch in cost for s in shapes)
    for n in (2, 4, 8):
        parts = W.assign_batches(shapes, n)
        assert sorted(i for p in parts for i in p) == list(range(64))
        for mode in (0, 1):
            loads = [sum(cost[shapes[i].batch][mode] for i in p) for p in parts]
            assert max(loads) / (sum(loads) / n) <= 1.025, (n, mode, loads)


def test_assign_batches_invariants_on_random_collections():
    """every batch on exactly one rank, capacity respected, never worse than plain LPT's maximum, deterministic"""
    import numpy as np
    from phylign_amd import workload as W
    rng = np.random.default_rng(12)
    full = W.select("full")
    for trial in range(40):
        k = int(rng.integers(1, 120))
        shapes = [full[i] for i in rng.choice(len(full), size=k, replace=False)]
        n = int(rng.integers(1, 10))
        cap = None if trial % 3 else int(sum(s.index_bytes for s in shapes) / n * 1.6) + max(s.index_bytes for s in shapes)
        parts = W.assign_batches(shapes, n, capacity_bytes=cap)
        assert sorted(i for p in parts for i in p) == list(range(k)) and len(parts) == n
        assert parts == W.assign_batches(shapes, n, capacity_bytes=cap)
        if cap is not None:
            assert all(sum(shapes[i].index_bytes for i in p) <= cap for p in parts)
        loads = [sum(W.scan_cost(shapes[i]) for i in p) for p in parts]
        # plain greedy LPT for comparison
        order = sorted(range(k), key=lambda i: (-W.scan_cost(shapes[i]), -shapes[i].index_bytes, i))
        lpt = [0] * n
        for i in order:
            lpt[min(range(n), key=lambda r: (lpt[r], r))] += W.scan_cost(shapes[i])
        if cap is None:
            assert max(loads) <= max(lpt)


def test_effective_cpus_is_positive_and_bounded():
    import os
    from phylign_amd.sysinfo import effective_cpus
    n = effective_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)


def test_assign_parts_shares_partition_every_batch_and_level_the_ranks():
    """workload.assign_parts: every batch is whole on one rank or shared by exactly two whose query shares tile
    [0, PART_DEN); replicas are small batches within the capacity; priced with the per-batch scan times MEASURED on one
    MI355X in launches of their own (a share costs that share of the batch's time) the slowest of 8 ranks is within 3 % of
    the mean (in the fused launches of a real step the ranks of both partitions end within ~1 %: profiles/r04/NOTES.md)"""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cost = {}
    with open(os.path.join(root, "profiles", "r03", "per_batch_cost.tsv")) as f:
        head = f.readline().rstrip("\n").split("\t")
        for line in f:
            row = dict(zip(head, line.rstrip("\n").split("\t")))
            cost[row["batch"]] = (float(row["ms_fetch_all"]), float(row["ms_bound"]))
    shapes = W.select("config3")
    for n in (1, 2, 3, 4, 5, 8):
        parts = W.assign_parts(shapes, n)
        assert parts == W.assign_parts(shapes, n) and len(parts) == n
        cover = {}
        for r, part in enumerate(parts):
            assert part == sorted(part) and len({p for p, _, _ in part}) == len(part)
            for p, lo, hi in part:
                assert 0 <= lo < hi <= W.PART_DEN
                cover.setdefault(p, []).append((lo, hi, r))
        assert sorted(cover) == list(range(64))
        for p, iv in cover.items():
            iv.sort()
            assert len(iv) <= 2 and iv[0][0] == 0 and iv[-1][1] == W.PART_DEN
            assert all(a[1] == b[0] for a, b in zip(iv, iv[1:]))
            if len(iv) == 2:
                assert shapes[p].index_bytes <= 4 << 30 and iv[0][2] != iv[1][2]
        modelled = W.parts_cost(shapes, parts)
        assert max(modelled) / (sum(modelled) / n) <= 1.012, (n, modelled)
        if n in (2, 4, 8):
            for mode in (0, 1):
                loads = [sum(cost[shapes[p].batch][mode] * (hi - lo) / W.PART_DEN for p, lo, hi in part) for part in parts]
                assert max(loads) / (sum(loads) / n) <= 1.03, (n, mode, loads)
    # capacity: a replica never pushes a rank over it
    full = W.select("full")
    cap = int(288e9 * 0.85)
    parts = W.assign_parts(full, 8, capacity_bytes=cap)
    assert all(sum(full[p].index_bytes for p, _, _ in part) <= cap for part in parts)
    assert sorted({p for part in parts for p, _, _ in part}) == list(range(305))


def test_self_launch_starts_fresh_ranks_and_relays_status(tmp_path):
    """phylign_amd.launch: `--gpus N` without a launcher = N children with the environment torch.distributed.run would
    give them; the job's status is the first failing rank's, and the other ranks are ended"""
    import subprocess
    import sys
    import time
    from phylign_amd import launch
    assert launch.wants_self_launch(8, {}) and not launch.wants_self_launch(1, {})
    assert not launch.wants_self_launch(8, {"WORLD_SIZE": "8", "RANK": "0"})
    script = tmp_path / "rank.py"
    script.write_text(
        "import os, sys, time\n"
        "r, w = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])\n"
        "assert os.environ['LOCAL_RANK'] == str(r) and os.environ['MASTER_ADDR'] == '127.0.0.1' and int(os.environ['MASTER_PORT']) > 0\n"
        "open(os.path.join(sys.argv[1], f'seen{r}'), 'w').write(str(w))\n"
        "if len(sys.argv) > 2 and r == int(sys.argv[2]):\n"
        "    sys.exit(7)\n"
        "if len(sys.argv) > 2:\n"
        "    time.sleep(60)\n"
        "if r == 0:\n"
        "    print('{\"line\": 1}')\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\nfrom phylign_amd import launch\n"
            "sys.exit(launch.self_launch_script(%r, sys.argv[1:], 5))\n" % (root, str(script)))
    r = subprocess.run([sys.executable, "-c", code, str(tmp_path)], capture_output=True, timeout=60)
    assert r.returncode == 0 and r.stdout.decode().strip() == '{"line": 1}', r.stderr.decode()
    assert sorted(f.name for f in tmp_path.glob("seen*")) == [f"seen{i}" for i in range(5)]
    assert (tmp_path / "seen3").read_text() == "5"
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", code, str(tmp_path), "2"], capture_output=True, timeout=60)
    assert r.returncode == 7 and time.time() - t0 < 30 and b"rank 2 exited with status 7" in r.stderr


def test_self_launch_ends_stuck_ranks_when_signalled(tmp_path):
    """SIGTERM to the launcher: the ranks get SIGTERM, a grace period, then SIGKILL -- a rank that ignores SIGTERM (stuck in
    a collective) does not outlive the launcher; and a rank whose launcher is killed outright ends by its parent-death signal"""
    import signal
    import subprocess
    import sys
    import time
    script = tmp_path / "rank.py"
    script.write_text(
        "import os, signal, sys, time\n"
        "sys.path.insert(0, sys.argv[2])\n"
        "from phylign_amd import launch\n"
        "launch.arm_parent_death_signal()\n"
        "r = int(os.environ['RANK'])\n"
        "if r == 1 and sys.argv[3] == 'stubborn':\n"
        "    signal.signal(signal.SIGTERM, signal.SIG_IGN)\n"
        "open(os.path.join(sys.argv[1], f'pid{r}'), 'w').write(str(os.getpid()))\n"
        "time.sleep(120)\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\nfrom phylign_amd import launch\n"
            "sys.exit(launch.spawn_ranks([sys.executable, %r] + sys.argv[1:], 3, grace_s=1.5))\n" % (root, str(script)))

    def start(mode):
        for f in tmp_path.glob("pid*"):
            f.unlink()
        p = subprocess.Popen([sys.executable, "-c", code, str(tmp_path), root, mode], stderr=subprocess.PIPE)
        t_end = time.time() + 30
        while time.time() < t_end and len(list(tmp_path.glob("pid*"))) < 3:
            time.sleep(0.05)
        pids = [int((tmp_path / f"pid{r}").read_text()) for r in range(3)]
        return p, pids

    def gone(pid):
        try:
            os.kill(pid, 0)
        except ProcessLookupError:
            return True
        try:                                             # a zombie of a dead parent counts as gone
            return open(f"/proc/{pid}/stat").read().split(")")[-1].split()[0] == "Z"
        except OSError:
            return True
    p, pids = start("stubborn")
    t0 = time.time()
    p.send_signal(signal.SIGTERM)
    assert p.wait(timeout=30) == 128 + signal.SIGTERM and time.time() - t0 < 15
    assert b"ending the ranks" in p.stderr.read()
    assert all(gone(pid) for pid in pids)                # the stubborn one was killed after the grace period
    p, pids = start("plain")
    p.kill()                                             # no chance to clean up: the ranks' own PDEATHSIG ends them
    p.wait(timeout=10)
    t_end = time.time() + 10
    while time.time() < t_end and not all(gone(pid) for pid in pids):
        time.sleep(0.1)
    assert all(gone(pid) for pid in pids)


def test_bench_and_stage_self_launch_before_touching_the_gpu():
    """`python bench.py --gpus 2` / `python -m phylign_amd.match_stage --gpus 2` in a box without a GPU: the parent
    starts the ranks (no torch import, no HIP call in the parent), every rank fails loudly, the status is relayed"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import torch
    if torch.cuda.is_available():
        pytest.skip("the CPU form of this check")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, timeout=300)
    assert r.returncode != 0 and b"[launch] rank" in r.stderr and b"needs an MI355X" in r.stderr and r.stdout == b""
    r = subprocess.run([sys.executable, "-m", "phylign_amd.match_stage", "--gpus", "2", "--synthetic", "small", "--queries", os.devnull,
                        "--out-dir", os.devnull], capture_output=True, timeout=300, cwd="/", env=dict(os.environ, PYTHONPATH=root))
    assert r.returncode != 0 and b"[launch] rank" in r.stderr


def test_stage_plan_sizes_loaders_by_the_reference_rules(tmp_path):
    """sizing.stage_plan / HostRam: the number of xz decoders of a stage run and the host RAM they may hold together come
    from the reference's sizing helpers (Snakefile:60-121): decoder RAM per batch from the sizes table's third column,
    the budget from max_ram_gb -- a loader is admitted while the running ones' decoders plus its own fit"""
    import threading
    import time
    from phylign_amd import sizing
    sz = tmp_path / "sizes.txt"
    sz.write_text("cobs/a__01.cobs_classic.xz  1000000000  1610678320\n"
                  "cobs/b__01.cobs_classic.xz  5000000000  1610678320\n"
                  "cobs/c__01.cobs_classic.xz  200000000  68157440\n")
    assert sizing.xz_ram_mb("a__01", str(sz)) == 1537 and sizing.loader_host_mb("c__01", str(sz)) == 66 + 64
    assert sizing.loader_host_mb("unknown__01", str(sz)) == 1536 + 64
    n, budget, need = sizing.stage_plan(["a__01", "b__01", "c__01"], str(sz), cpus=16, max_ram_gb=12)
    assert budget == 12 * 1024 and need == {"a__01": 1601, "b__01": 1601, "c__01": 130}
    assert n == 7                                         # 12 GiB hold seven 1.5-GiB decoders; the CPUs would allow 12
    assert sizing.stage_plan(["a__01"], str(sz), cpus=16, max_ram_gb=64)[0] == 12
    assert sizing.stage_plan(["a__01"], str(sz), cpus=6, max_ram_gb=64)[0] == 4
    assert sizing.stage_plan(["a__01"], str(sz), cpus=64, max_ram_gb=1)[0] == 1
    assert sizing.stage_plan(["a__01"], str(sz), cpus=64, max_ram_gb=1, loaders=5)[0] == 5      # an explicit --loaders stands
    # what the reference reserves per streaming job = index MB + decoder MB (Snakefile:72-82); ours is the decoder's share
    assert sizing.batch_ram_mb("b__01", str(sz)) == 4769 + 1537
    ram = sizing.HostRam(4000)
    running, peak, lock = [0], [0], threading.Lock()

    def loader(mb):
        ram.acquire(mb)
        with lock:
            running[0] += mb
            peak[0] = max(peak[0], running[0])
        time.sleep(0.01)
        with lock:
            running[0] -= mb
        ram.release(mb)
    ts = [threading.Thread(target=loader, args=(mb,)) for mb in [1601] * 6 + [130] * 5 + [5000]]
    [t.start() for t in ts]
    [t.join(timeout=30) for t in ts]
    assert not any(t.is_alive() for t in ts) and ram.held == 0
    assert peak[0] <= 5000 and ram.peak == peak[0]        # (the 5000-MB loader ran alone)


def test_match_stage_refuses_abbreviated_options():
    """--config fills in only the options the command line did not set; which ones it set is read off argv by their full
    names, so an abbreviation (`--thresh 0.5`) must not be accepted silently and then overwritten by the config file"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "phylign_amd.match_stage", "--synthetic", "small", "--queries", os.devnull, "--out-dir", os.devnull,
                        "--thresh", "0.5"], capture_output=True, timeout=120, env=dict(os.environ, PYTHONPATH=root))
    assert r.returncode == 2 and b"unrecognized arguments: --thresh" in r.stderr


def test_self_launched_ranks_rendezvous_on_the_port_the_launcher_holds(tmp_path):
    """the launcher keeps MASTER_PORT bound (not listening) while the ranks start, so that nobody else is handed the port;
    rank 0's store still binds and listens on it: two ranks form a gloo group through env:// and reduce"""
    import subprocess
    import sys
    script = tmp_path / "rank.py"
    script.write_text(
        "import os, sys, datetime\n"
        "import torch, torch.distributed as dist\n"
        "dist.init_process_group('gloo', timeout=datetime.timedelta(seconds=60))\n"
        "t = torch.tensor([dist.get_rank() + 1]); dist.all_reduce(t)\n"
        "open(os.path.join(sys.argv[1], 'sum%d' % dist.get_rank()), 'w').write(str(int(t)))\n"
        "dist.destroy_process_group()\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\nfrom phylign_amd import launch\n"
            "sys.exit(launch.self_launch_script(%r, sys.argv[1:], 2))\n" % (root, str(script)))
    r = subprocess.run([sys.executable, "-c", code, str(tmp_path)], capture_output=True, timeout=180)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert (tmp_path / "sum0").read_text() == "3" and (tmp_path / "sum1").read_text() == "3"


def test_gene_length_package_data_is_what_the_generator_writes(tmp_path):
    """phylign_amd/data/argannot_lengths.txt is package data (bench.py and an installed package need no tests/ tree); it is
    byte for byte what tools/gen_golden_argannot.py derives from the reference's data/ARGannot_r3.fa (build container only)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert os.path.dirname(os.path.abspath(W._ARGANNOT)) == os.path.join(root, "phylign_amd", "data") and os.path.exists(W._ARGANNOT)
    ref = "/root/reference/data/ARGannot_r3.fa"
    if not os.path.exists(ref):
        pytest.skip("the reference checkout is not on this machine")
    out = tmp_path / "lengths.txt"
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "gen_golden_argannot.py"), ref, str(out)], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    assert out.read_bytes() == open(W._ARGANNOT, "rb").read()


def test_gene_length_workload_helpers():
    """the query shape of the reference's bundled gene file: the committed lengths fixture, the ragged query
    maker and the ragged planting plan (fractions that straddle the threshold, per-query k-mer counts)"""
    from phylign_amd import workload as W
    lens = W.argannot_lengths()
    assert len(lens) == 1856 and (min(lens), max(lens)) == (237, 3153) and sum(n - 30 for n in lens) == 1594532
    assert sum(1 for n in lens if n - 30 < 1024) == 1514                      # 10-plane class; the other 342 are 13-plane
    fasta, seqs = W.make_queries_lengths(lens[:5], seed=3, prefix="g")
    assert [len(s_) for s_ in seqs] == lens[:5] and fasta.count(b">g") == 5 and set(fasta) <= set(b">g0123456789ACGT\n")
    assert W.make_queries_lengths(lens[:5], seed=3, prefix="g")[0] == fasta   # seeded
    terms = [n - 30 for n in lens[:40]]
    hashes = np.arange(sum(terms), dtype=np.uint64) * np.uint64(2654435761)
    shapes = W.select("small")
    plan, sure = W.plant_plan_ragged(hashes, terms, shapes, every=8, docs_per_query=6, threshold=0.7)
    assert set(plan) <= set(range(len(shapes))) and sure == 5 * 4                # 5 planted queries x 4 of 6 fractions reach 0.7
    off = np.concatenate([[0], np.cumsum(terms)])
    for pos, (rows, docs) in plan.items():
        assert len(rows) == len(docs) and rows.max() < shapes[pos].signature_size and docs.max() < shapes[pos].n_docs
    # the rows of a planted (query, document) pair are the query's first ceil(f x k-mers) rows, for the fractions in order
    q0_rows = (hashes[off[0]:off[1]] % np.uint64(shapes[0].signature_size))
    rows0, docs0 = plan[0]
    first_doc = docs0[0]
    n_first = int(np.count_nonzero(docs0[:terms[0]] == first_doc))
    assert n_first == terms[0] and np.array_equal(rows0[:n_first], q0_rows)      # fraction 1.0 comes first


def test_xz_probe_and_thread_choice(tmp_path):
    """match_stage.xz_block_structure (`xz --robot --list`) and the thread count the block-parallel decoder gets"""
    import lzma
    import subprocess
    from phylign_amd import match_stage as MS
    from phylign_amd import xzpar
    one = tmp_path / "a.xz"
    one.write_bytes(lzma.compress(os.urandom(50000)))
    multi = tmp_path / "b"
    multi.write_bytes(os.urandom(3 << 20))
    subprocess.run(["xz", "-T2", "-0", "--block-size=1MiB", str(multi)], check=True)
    st = MS.xz_block_structure([str(one), str(multi) + ".xz", str(tmp_path / "missing.xz")])
    assert st == {"files": 2, "multi_block_files": 1, "blocks_max": 3, "blocks_total": 4}
    assert MS.xz_block_structure([str(tmp_path / "missing.xz")]) is None
    assert MS.xz_decode_threads(None, 8) == 1
    pl = xzpar.plan(str(multi) + ".xz")
    assert MS.xz_decode_threads(pl, 8) == 3 and MS.xz_decode_threads(pl, 2) == 2 and MS.xz_decode_threads(xzpar.plan(str(one)), 8) == 1

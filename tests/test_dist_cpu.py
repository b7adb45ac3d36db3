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
    per-batch scan times MEASURED on one MI355X (profiles/r03/per_batch_cost.tsv): slowest rank / mean <= 1.05 in
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
            assert max(loads) / (sum(loads) / n) <= 1.05, (n, mode, loads)


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

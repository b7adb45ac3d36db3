#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the PMC summaries of tools/run_profiles.sh:
    python3 tools/make_pmc_traffic.py <pmc_fetch_all.txt> <pmc_bound.txt> [round] > profiles/pmc_traffic.json
HBM bytes per launch of every k_scan instantiation = 128 x TCC_EA0_RDREQ_128B + 64 x _64B + 32 x _32B (the request
sizes are counted, not assumed) + WRITE_SIZE (KiB).  Stamped with the git blob id of pm_kernels.hip: bench.py
reports `traffic: null` for any other kernel source."""
import hashlib
import json
import os
import re
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def blob():
    data = open(os.path.join(ROOT, "phylign_amd", "csrc", "pm_kernels.hip"), "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def kname(raw):
    m = re.search(r"k_scan<(\d+), (\d+), (true|false), (true|false)>", raw)
    if not m:
        return None
    g, p, nh1, wq = m.groups()
    return f"k_scan<G={'mixed' if g == '0' else g},P={p},{'NH1' if nh1 == 'true' else 'NHn'}{',WQ' if wq == 'true' else ''}>"


def parse(path):
    out = {}
    for line in open(path):
        f = line.split(None, 3)
        if len(f) < 4 or not f[0].isdigit():
            continue
        name = kname(f[3])
        if name:
            out.setdefault(name, {})[f[2]] = float(f[1])
    return out


def main(fetch_all, bound, rnd=4):
    res = {"round": int(rnd), "pm_kernels_blob": blob(), "workload": "config3", "queries": 100000, "query_len": 150,
           "how": "tools/run_profiles.sh: rocprofv3 --kernel-trace --pmc passes (one counter group per run) over `bench.py --steps 1 "
                  "--warmup 0 --no-cpu-baseline --only-headline --no-pipeline`; bytes = 128 x TCC_EA0_RDREQ_128B_sum + 64 x "
                  "TCC_EA0_RDREQ_64B_sum + 32 x TCC_EA0_RDREQ_32B_sum + 1024 x WRITE_SIZE", "kernels": {}}
    for mode, path in (("fetch_all_rows", fetch_all), ("threshold_bound", bound)):
        for name, c in parse(path).items():
            if "TCC_EA0_RDREQ_128B_sum" not in c:
                continue
            rd = 128 * c["TCC_EA0_RDREQ_128B_sum"] + 64 * c.get("TCC_EA0_RDREQ_64B_sum", 0) + 32 * c.get("TCC_EA0_RDREQ_32B_sum", 0)
            wr = 1024 * c.get("WRITE_SIZE", 0)
            res["kernels"].setdefault(name, {})[mode] = {
                "hbm_bytes_per_launch": int(rd + wr),
                "read_requests": {k: int(c.get(k, 0)) for k in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_128B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_32B_sum")},
                "write_bytes": int(wr), "l2_hit_rate": (c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])) if "TCC_HIT_sum" in c and c.get("TCC_MISS_sum") else None,
                "source": os.path.relpath(path, ROOT) if os.path.isabs(path) else path}
    json.dump(res, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(*sys.argv[1:4])

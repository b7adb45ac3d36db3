#!/usr/bin/env python3
"""Registers, scratch and occupancy of every k_scan instantiation, from the compiler's own resource remarks
(`hipcc -Rpass-analysis=kernel-resource-usage`; no GPU needed).  Writes a table:

    python3 tools/kernel_resources.py > profiles/r05/kernel_resource_usage.txt
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
src = os.path.join(ROOT, "phylign_amd", "csrc", "pm_kernels.hip")
with tempfile.TemporaryDirectory() as td:
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-c", src,
                        "-o", os.path.join(td, "k.o"), "-Rpass-analysis=kernel-resource-usage"] + os.environ.get("PM_EXTRA_FLAGS", "").split(),
                       capture_output=True, text=True)
if r.returncode != 0:
    sys.exit(r.stderr[-3000:])
blocks = r.stderr.split("Function Name: ")[1:]
rows = []
for b in blocks:
    name = b.split()[0]
    m = re.match(r"_ZN2pm6k_scanILi(\d+)ELi(\d+)ELb([01])ELb([01])EEEvNS_8ScanArgsE", name)
    if not m:
        continue

    def g(key):
        mm = re.search(key + r": (\d+)", b)
        return int(mm.group(1)) if mm else -1
    rows.append((int(m.group(2)), int(m.group(1)), m.group(3) == "1", m.group(4) == "1", g("VGPRs"), g("VGPRs Spill"),
                 g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]"), g("SGPRs Spill")))
rows.sort()
print("# k_scan<G, P, NH1, WQ> on gfx950: compiler resource remarks (hipcc -O3 -Rpass-analysis=kernel-resource-usage)")
print("# P = counter bit planes, G = lanes per row (0 = mixed widths), NH1 = one hash function (the 661k indexes), WQ = wide-query form")
print("%-28s %5s %6s %8s %5s %6s %6s" % ("instantiation", "VGPR", "spill", "scratchB", "occ", "LDS", "sgprsp"))
n_spill = 0
for P, G, nh1, wq, v, sp, sc, occ, lds, ss in rows:
    n_spill += 1 if (sp > 0 or sc > 0) else 0
    print("%-28s %5d %6d %8d %5d %6d %6d" % ("k_scan<%d,%d,%s%s>" % (G, P, "NH1" if nh1 else "NHn", ",WQ" if wq else ""), v, sp, sc, occ, lds, ss))
nh1_spill = [(P, G, wq) for P, G, nh1, wq, v, sp, sc, *_ in rows if nh1 and (sp > 0 or sc > 0)]
print("# %d instantiations, %d with scratch; with one hash function (NH1): %d with scratch -> planes %s" % (
    len(rows), n_spill, len(nh1_spill), sorted({p for p, _, _ in nh1_spill})))

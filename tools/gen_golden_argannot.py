#!/usr/bin/env python3
"""The query SHAPE of the reference's bundled gene file (data/ARGannot_r3.fa): one line per record,
`<sequence length>` in file order -- data only, no sequence and no header text.  Run in the build container
(needs /root/reference); bench.py and the tests read the committed phylign_amd/data/argannot_lengths.txt
(workload.argannot_lengths).    python3 tools/gen_golden_argannot.py [<ARGannot_r3.fa> [<out file>]]"""
import os
import sys

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/data/ARGannot_r3.fa"
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "phylign_amd", "data",
                                                        "argannot_lengths.txt")
lens, cur = [], None
with open(ref) as f:
    for line in f:
        line = line.rstrip("\r\n")
        if line.startswith(">"):
            if cur is not None:
                lens.append(cur)
            cur = 0
        elif cur is not None:
            cur += len(line)
if cur is not None:
    lens.append(cur)
with open(out, "w") as f:
    f.write("# sequence lengths (bp) of the %d records of data/ARGannot_r3.fa in file order; k-mers per record = length - 30\n" % len(lens))
    f.write("\n".join(str(n) for n in lens) + "\n")
print(len(lens), "records,", min(lens), "...", max(lens), "bp,", sum(n - 30 for n in lens), "31-mers")

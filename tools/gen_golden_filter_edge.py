#!/usr/bin/env python3
"""Edge cases of the 04_filter reader, captured by running the reference's scripts/filter_queries.py
(build container only; `xopen` shim as in gen_golden_filter.py): for each case one match file
tests/golden/filter/edge/<case>____q.txt and either <case>.n3.fa (its stdout) or <case>.crash
(it raised: the rule fails).  The query file is tests/golden/filter/queries.fa."""
import os, subprocess, sys, tempfile
ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
script = os.path.join(ref, "scripts", "filter_queries.py")
base = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "filter")
outdir = os.path.join(base, "edge")
os.makedirs(outdir, exist_ok=True)
CASES = {
    "pre_header_match": "xx_PRE1\t40\nyy_PRE2\t39\n*r1 descr\t1\nzz_SAMA\t30\n*r2 d\t0\n",
    "empty_file": "",
    "only_blank_lines": "\n   \n\n",
    "bad_count": "*r1 descr\tabc\nzz_SAMA\t30\n",
    "no_tab_in_header": "*r1 3\nzz_SAMA\t30\n",
    "unknown_query": "*nobody\t1\nzz_SAMA\t30\n",
    "two_underscores": "*r1\t1\nzz_SAM_A\t30\n",
    "no_underscore": "*r1\t1\nzzSAMA\t30\n",
    "three_fields": "*r1\t1\nzz_SAMA\t30\t7\n",
    "one_field": "*r1\t1\nzz_SAMA\n",
    "float_kmers": "*r1\t1\nzz_SAMA\t30.0\n",
    "extra_header_fields": "*r1 descr\t2\tmore\nzz_SAMA\t30\nzz_SAMB\t29\n",
    "crlf_blank_and_spaces": "*r1 descr\t2\r\n  zz_SAMA\t30  \r\n\r\n\tzz_SAMB 29\r\n*r3\t0\r\n",
    "signed_numbers": "*r1\t-1\nzz_SAMA\t+30\n*r2\t+1\nzz_SAMB\t29\n",
    "same_query_twice": "*r1\t2\naa_S1\t30\nbb_S2\t30\n*r2\t1\ncc_S3\t25\n*r1\t2\ndd_S0\t30\nee_S4\t31\n",
    "ties_past_keep": "*r4\t6\na_R6\t30\nb_R5\t30\nc_R4\t29\nd_R3\t29\ne_R2\t29\nf_R1\t28\n",
    "empty_reference_name": "*r1\t1\nzz_\t30\n",
    "empty_random_id": "*r1\t1\n_SAMA\t30\n",
    # text mode = universal newlines: a lone CR ends a line too
    "bare_cr_line_ends": "*r1 descr\t2\rzz_SAMA\t30\rzz_SAMB\t29\r*r2\t1\ryy_SAMC\t28\r",
    "mixed_line_ends": "*r1 descr\t2\r\nzz_SAMA\t30\rzz_SAMB\t29\n\r*r2\t1\n\nyy_SAMC\t28",
    "count_above_32_bits": "*r1\t1\nzz_SAMA\t4294967296\n",
}
shim = tempfile.mkdtemp()
with open(os.path.join(shim, "xopen.py"), "w") as f:
    f.write("import gzip\n"
            "def xopen(fn, mode='r'):\n"
            "    return gzip.open(fn, mode + 't') if str(fn).endswith('.gz') else open(fn, mode)\n")
env = dict(os.environ, PYTHONPATH=shim)
for name, text in CASES.items():
    fn = os.path.join(outdir, f"{name}____q.txt")
    with open(fn, "w", newline="") as f:
        f.write(text)
    r = subprocess.run([sys.executable, script, "-n", "3", "-q", os.path.join(base, "queries.fa"), fn],
                       capture_output=True, env=env, cwd=outdir)
    for ext in (".n3.fa", ".crash"):
        if os.path.exists(os.path.join(outdir, name + ext)):
            os.unlink(os.path.join(outdir, name + ext))
    if r.returncode == 0:
        with open(os.path.join(outdir, name + ".n3.fa"), "wb") as f:
            f.write(r.stdout)
        print(f"{name}: ok, {len(r.stdout)} bytes")
    else:
        last = r.stderr.decode().strip().splitlines()[-1]
        with open(os.path.join(outdir, name + ".crash"), "w") as f:
            f.write(last + "\n")
        print(f"{name}: raised {last}")

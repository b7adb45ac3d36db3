#!/usr/bin/env python3
"""Golden I/O pairs for the 03_match post-filter, captured from the reference's
scripts/postprocess_cobs.py (run as a subprocess, in the build container only).
Writes tests/golden/postprocess/<case>.in, <case>.n<N>.out."""
import os, random, subprocess, sys
ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
script = os.path.join(ref, "scripts", "postprocess_cobs.py")
outdir = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "postprocess")
os.makedirs(outdir, exist_ok=True)
random.seed(7)

def synth(nq, maxhits, tie_heavy):
    lines = []
    for q in range(nq):
        n = random.randint(0, maxhits)
        scores = sorted((random.choice([120, 110, 100, 90, 84]) if tie_heavy else random.randint(84, 120) for _ in range(n)), reverse=True)
        comment = " some comment" if q % 3 == 0 else ""
        lines.append(f"*read{q}{comment}\t{n}\n")
        for i, s in enumerate(scores):
            lines.append(f"{random.randint(0, 99999):05x}_SAMEA{q:03d}{i:04d}\t{s}\n")
    return "".join(lines)

cases = {
    "survey_a4": "*1A comment here\t4\nabcde_SAMEA1\t120\nfghij_SAMEA2\t110\nxx_SAMEA3\t110\nyy_SAMEA4\t90\n",
    "zero_hits": "*q0\t0\n*q1 c\t0\n*q2\t1\nab_X\t31\n",
    "underscores": "*q\t4\nnounderscore\t5\none_two_three\t5\n_lead\t4\ntrail_\t3\n",
    "headerless": "a_b\t3\n*q\t1\nc_d\t2\n",
    "ties_small": synth(12, 8, True),
    "random_mid": synth(40, 30, False),
    "ties_large": synth(10, 250, True),
    "empty": "",
}
for name, text in cases.items():
    with open(os.path.join(outdir, name + ".in"), "w") as f:
        f.write(text)
    for n in (0, 1, 2, 3, 100):
        r = subprocess.run([sys.executable, script, "-n", str(n)], input=text.encode(), capture_output=True)
        # a non-zero exit (e.g. int("_") on a name without '_' past the n-th hit) is part of
        # the contract: the partial stdout is kept in a .fail file instead of .out
        ext = "out" if r.returncode == 0 else "fail"
        with open(os.path.join(outdir, f"{name}.n{n}.{ext}"), "wb") as f:
            f.write(r.stdout)
print("wrote", len(cases) * 5, "pairs to", outdir)

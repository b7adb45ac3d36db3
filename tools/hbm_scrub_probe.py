#!/usr/bin/env python3
"""What does a process pay for HBM that the previous process has just released?  (profiles/r05/NOTES.md: one hipMalloc of a
stage run that follows another stage run blocks 2.5 - 3.5 s.)  A child allocates and touches `--hold-gb` of HBM and exits; a second
child starts right away (or after --sleep seconds) and times 4 GB allocations one after the other.

    python3 tools/hbm_scrub_probe.py [--hold-gb 135] > gpurun_out/r05/hbm_scrub_probe.txt"""
import argparse
import os
import subprocess
import sys
import time

HOLD = r'''
import sys, torch
gb = int(sys.argv[1])
bufs = [torch.empty(1 << 30, dtype=torch.uint8, device="cuda").fill_(1) for _ in range(gb)]
torch.cuda.synchronize()
print("held", gb, "GB", flush=True)
'''
PROBE = r'''
import sys, time, torch
torch.cuda.init()
n, each = int(sys.argv[1]), int(sys.argv[2])
t_all = time.perf_counter()
out, bufs = [], []
for i in range(n):
    t0 = time.perf_counter()
    b = torch.empty(each << 30, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    out.append(time.perf_counter() - t0)
    bufs.append(b)
print("alloc ms:", " ".join("%.0f" % (x * 1e3) for x in out), "| total %.2f s" % (time.perf_counter() - t_all), flush=True)
'''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hold-gb", type=int, default=135)
    ap.add_argument("--allocs", type=int, default=32)
    ap.add_argument("--each-gb", type=int, default=4)
    args = ap.parse_args()
    env = dict(os.environ, PYTORCH_NO_CUDA_MEMORY_CACHING="1")
    for label, hold, pause in (("fresh GPU (nothing released before)", 0, 0.0), ("right after a process released %d GB" % args.hold_gb, args.hold_gb, 0.0),
                               ("5 s after a process released %d GB" % args.hold_gb, args.hold_gb, 5.0), ("right after, again", args.hold_gb, 0.0)):
        if hold:
            subprocess.run([sys.executable, "-c", HOLD, str(hold)], env=env, check=True, stdout=subprocess.DEVNULL)
        time.sleep(pause)
        t0 = time.perf_counter()
        r = subprocess.run([sys.executable, "-c", PROBE, str(args.allocs), str(args.each_gb)], env=env, capture_output=True, text=True)
        print(f"{label}: process wall {time.perf_counter() - t0:.2f} s; {r.stdout.strip()} {r.stderr.strip()[-200:]}", flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Known-byte-count kernels for calibrating rocprofv3 FETCH_SIZE / WRITE_SIZE /
TCC_EA0_RDREQ* on gfx950 (MI355X_MICROARCH.md, HBM section): a 4 GiB device
copy (reads 4 GiB, writes 4 GiB) and a 4 GiB read-only reduction."""
import torch
n = 1 << 30                      # 4 GiB of float32
a = torch.ones(n, dtype=torch.float32, device="cuda")
b = torch.empty_like(a)
torch.cuda.synchronize()
for _ in range(3):
    b.copy_(a)
    s = a.sum()
torch.cuda.synchronize()
print("calib done", float(s))

#!/usr/bin/env python3
"""Does the PHYSICAL placement of the four planes decide the headline kernel's rate?  K plane sets (re, im, out re,
out im: 4 GiB each at configs[2]) are allocated and all kept alive, filled with the same batch, and the N=4096 C2C
launch is timed on each set in interleaved rounds; a stable ranking across rounds = placement, not noise.
    python tools/placement_probe.py [K=8] [rounds=4]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bench import synth_batch
from pragma_dsp_amd.batch import BatchedFft

K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
R = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
n, rows = 4096, 65536
plan = BatchedFft(n, dev)
re0, im0 = synth_batch(rows, n, dev)
sets = [(re0, im0, torch.empty_like(re0), torch.empty_like(im0))]
for _ in range(K - 1):
    re, im = torch.empty_like(re0), torch.empty_like(im0)
    re.copy_(re0)
    im.copy_(im0)
    sets.append((re, im, torch.empty_like(re0), torch.empty_like(im0)))
for s in sets:  # warm-up / clock ramp
    for _ in range(30):
        plan.forward(s[0], s[1], out=(s[2], s[3]))
torch.cuda.synchronize()
res = np.zeros((R, K))
for r in range(R):
    for k, s in enumerate(sets):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(5):
            plan.forward(s[0], s[1], out=(s[2], s[3]))
        e0.record()
        for _ in range(30):
            plan.forward(s[0], s[1], out=(s[2], s[3]))
        e1.record()
        torch.cuda.synchronize()
        res[r, k] = 16.0 * rows * n / (e0.elapsed_time(e1) / 30 * 1e-3) / 1e9
print("GB/s by plane set (columns) and round (rows):")
for r in range(R):
    print("  " + "  ".join(f"{v:6.0f}" for v in res[r]))
med = np.median(res, axis=0)
print("median per set: " + "  ".join(f"{v:6.0f}" for v in med))
print(f"best set {med.max():.0f} GB/s = {med.max() / 80:.1f} %, worst {med.min():.0f} = {med.min() / 80:.1f} %, "
      f"spread between sets {100 * (med.max() / med.min() - 1):.1f} %, "
      f"largest round-to-round spread within a set {100 * (res.max(axis=0) / res.min(axis=0) - 1).max():.1f} %")
print("addresses (GiB): " + "  ".join("/".join(f"{t.data_ptr() / 2**30:.1f}" for t in s) for s in sets))

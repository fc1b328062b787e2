#!/usr/bin/env python3
"""Follow-up 9: which layout to ship.  Inside one 100-GiB allocation: inputs at (0, A), outputs at (X, X + D) GiB."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bench import synth_batch
from pragma_dsp_amd.batch import BatchedFft

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
n, rows = 4096, 65536
plane = rows * n
GiB = 1 << 30
plan = BatchedFft(n, dev)
re0, im0 = synth_batch(rows, n, dev)
NG = 100
pool = torch.empty(NG * GiB // 4, dtype=torch.float32, device=dev)
v = lambda g: pool[g * (GiB // 4):g * (GiB // 4) + plane].view(rows, n)
for g in (0, 1, 2, 4, 8):
    v(g).copy_(re0 if g == 0 else im0)


def timed(l, reps=12):
    re, im, ore, oim = (v(g) for g in l)
    for _ in range(3):
        plan.forward(re, im, out=(ore, oim))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        plan.forward(re, im, out=(ore, oim))
    e1.record()
    torch.cuda.synchronize()
    return 16.0 * rows * n / (e0.elapsed_time(e1) / reps * 1e-3) / 1e9 / 80


for _ in range(60):
    timed((0, 1, 32, 64))
print("pool base %.3f GiB; %% of 8 TB/s" % (pool.data_ptr() / 2**30))
Ds = (4, 8, 12, 16, 20, 24, 28, 32, 36, 40, 48)
for a in (1, 8):
    for x in (16, 24, 32, 40, 48):
        print(f"  in (0,{a}) out ({x},{x}+D): " + "  ".join(f"D{d}:{timed((0, a, x, x + d)):.1f}" for d in Ds if x + d < NG))

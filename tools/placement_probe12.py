#!/usr/bin/env python3
"""Follow-up 11: the f64 form of configs[2] (4-GiB planes).  One 140-GiB allocation, inputs at (0, A) GiB, outputs at
(X, X + D) GiB; % of 8 TB/s."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pragma_dsp_amd.batch import BatchedFft

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
n, rows = 4096, 65536
plane = rows * n  # doubles
GiB = 1 << 30
NG = 140
pool = torch.empty(NG * GiB // 8, dtype=torch.float64, device=dev)
plan = BatchedFft(n, dev, dtype=torch.float64)
v = lambda g: pool[g * (GiB // 8):g * (GiB // 8) + plane].view(rows, n)
for g in (0, 4, 8, 16):
    v(g).normal_()


def timed(l, reps=8):
    re, im, ore, oim = (v(g) for g in l)
    for _ in range(2):
        plan.forward(re, im, out=(ore, oim))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        plan.forward(re, im, out=(ore, oim))
    e1.record()
    torch.cuda.synchronize()
    return 32.0 * rows * n / (e0.elapsed_time(e1) / reps * 1e-3) / 1e9 / 80


for _ in range(30):
    timed((0, 4, 40, 80), 1)
print("pool base %.3f GiB" % (pool.data_ptr() / 2**30))
print("back to back (0,4,8,12): %.1f   f32 layout (0,4,40,80): %.1f" % (timed((0, 4, 8, 12)), timed((0, 4, 40, 80))))
Ds = (4, 8, 16, 24, 32, 40, 48, 64)
for a in (4, 16):
    for x in (24, 32, 40, 48, 64):
        print(f"  in (0,{a}) out ({x},{x}+D): " + "  ".join(f"D{d}:{timed((0, a, x, x + d)):.1f}" for d in Ds if x + d + 4 <= NG))

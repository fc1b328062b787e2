#!/usr/bin/env python3
"""Follow-up 5: GiB-granular layouts inside ONE 16-GiB allocation: the four planes in slots (a, b, c, d) of 1 GiB
(+ 8 MiB of slack per slot so that sub-GiB skews fit), every layout timed in interleaved rounds."""
import itertools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pragma_dsp_amd.batch import BatchedFft

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
n, rows = 4096, 65536
plane = rows * n
slot = plane + (2 << 20)  # floats: 1 GiB + 8 MiB
NS = 16
pool = torch.empty(NS * slot, dtype=torch.float32, device=dev)
pool.normal_()
plan = BatchedFft(n, dev)
v = lambda i: pool[i * slot:i * slot + plane].view(rows, n)
layouts = [(0, 1, 2, 3), (0, 2, 1, 3), (0, 3, 1, 2), (2, 3, 0, 1), (0, 1, 3, 2), (0, 1, 4, 5), (0, 1, 8, 9), (0, 4, 8, 12),
           (0, 8, 1, 9), (0, 1, 6, 7), (0, 2, 4, 6), (0, 1, 12, 13), (0, 1, 15, 14), (3, 5, 10, 12), (0, 15, 7, 8), (1, 2, 4, 8),
           (0, 1, 5, 4), (0, 1, 10, 11), (4, 5, 6, 7), (8, 9, 10, 11), (12, 13, 14, 15), (0, 1, 7, 6)]


def rate(l):
    re, im, ore, oim = (v(i) for i in l)
    for _ in range(4):
        plan.forward(re, im, out=(ore, oim))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        plan.forward(re, im, out=(ore, oim))
    e1.record()
    torch.cuda.synchronize()
    return 16.0 * rows * n / (e0.elapsed_time(e1) / 20 * 1e-3) / 1e9


for _ in range(15):
    rate(layouts[0])
res = np.array([[rate(l) for l in layouts] for _ in range(3)])
med = np.median(res, axis=0)
print("pool base %.3f GiB, slot %.4f GiB" % (pool.data_ptr() / 2**30, slot * 4 / 2**30))
for l, m, lo, hi in sorted(zip(layouts, med, res.min(axis=0), res.max(axis=0)), key=lambda t: -t[1]):
    print(f"  {str(l):18s} {m:6.0f}  ({lo:.0f} .. {hi:.0f})")

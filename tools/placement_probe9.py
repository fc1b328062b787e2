#!/usr/bin/env python3
"""Follow-up 8: is "planes 32 GiB apart" a RULE?  First the four plain allocations a fresh process gets (what bench.py
times), then layouts inside one 100-GiB allocation: slots (a, b, c, d) in GiB for re_in, im_in, re_out, im_out.
Run on several boxes."""
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


def timed(re, im, ore, oim, reps=20):
    for _ in range(4):
        plan.forward(re, im, out=(ore, oim))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        plan.forward(re, im, out=(ore, oim))
    e1.record()
    torch.cuda.synchronize()
    return 16.0 * rows * n / (e0.elapsed_time(e1) / reps * 1e-3) / 1e9


re0, im0 = synth_batch(rows, n, dev)
o0, o1 = torch.empty_like(re0), torch.empty_like(im0)
for _ in range(80):
    plan.forward(re0, im0, out=(o0, o1))
torch.cuda.synchronize()
first = [timed(re0, im0, o0, o1) for _ in range(3)]
print("first plain allocations of the process: " + " ".join(f"{x:.0f}" for x in first))
NG = 100
pool = torch.empty(NG * GiB // 4, dtype=torch.float32, device=dev)
v = lambda g: pool[g * (GiB // 4):g * (GiB // 4) + plane].view(rows, n)
layouts = [(0, 1, 2, 3), (0, 1, 32, 64), (0, 8, 32, 64), (0, 32, 64, 96), (0, 16, 32, 48), (0, 36, 72, 99), (0, 33, 66, 99),
           (0, 64, 32, 96), (2, 34, 66, 98), (0, 1, 32, 33), (0, 32, 1, 33), (0, 32, 64, 65), (0, 24, 48, 72), (0, 40, 80, 99)]
used = sorted({g for l in layouts for g in l})
for g in used:
    v(g).copy_(re0 if g % 2 == 0 else im0)
res = np.array([[timed(*(v(g) for g in l)) for l in layouts] for _ in range(3)])
med = np.median(res, axis=0)
print("pool base %.3f GiB" % (pool.data_ptr() / 2**30))
for l, m, lo, hi in zip(layouts, med, res.min(axis=0), res.max(axis=0)):
    print(f"  {str(l):18s} {m:6.0f} GB/s = {m / 80:.1f} %  ({lo:.0f} .. {hi:.0f})")

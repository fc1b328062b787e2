#!/usr/bin/env python3
"""Follow-up 7: GiB-scale offsets inside ONE 72-GiB allocation (address bits 30 .. 35): re_in @ 0, im_in @ A GiB,
re_out @ X GiB, im_out @ X + D GiB."""
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
GiB = 1 << 30
NG = 72
pool = torch.empty(NG * GiB // 4, dtype=torch.float32, device=dev)
for g in range(0, NG, 8):
    pool[g * GiB // 4:(g + 8) * GiB // 4].normal_()
plan = BatchedFft(n, dev)
v = lambda g: pool[g * (GiB // 4):g * (GiB // 4) + plane].view(rows, n)


def rate(a, x, d, reps=12):
    re, im, ore, oim = v(0), v(a), v(x), v(x + d)
    for _ in range(3):
        plan.forward(re, im, out=(ore, oim))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        plan.forward(re, im, out=(ore, oim))
    e1.record()
    torch.cuda.synchronize()
    return 16.0 * rows * n / (e0.elapsed_time(e1) / reps * 1e-3) / 1e9


for _ in range(60):
    rate(1, 2, 1)
print("pool base %.3f GiB" % (pool.data_ptr() / 2**30))
print("outputs at X, X+1 (inputs at 0, 1):  " + "  ".join(f"{x}:{rate(1, x, 1):.0f}" for x in (2, 3, 4, 5, 6, 8, 10, 12, 16, 20, 24, 32, 33, 34, 36, 40, 48, 64, 65, 66, 68)))
print("im_in at A (outputs at 32, 33):      " + "  ".join(f"{a}:{rate(a, 32, 1):.0f}" for a in (1, 2, 3, 4, 8, 16, 17, 24)))
print("im_out at X+D (inputs 0,1; X = 32):  " + "  ".join(f"{d}:{rate(1, 32, d):.0f}" for d in (1, 2, 3, 4, 8, 16, 32, 36)))
print("all four far apart:                  " + "  ".join(f"({a},{x},{x + d}):{rate(a, x, d):.0f}" for a, x, d in ((16, 32, 16), (8, 16, 8), (4, 8, 4), (32, 64, 4), (17, 34, 17), (2, 4, 2), (24, 48, 20))))

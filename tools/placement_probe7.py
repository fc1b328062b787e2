#!/usr/bin/env python3
"""Follow-up 6: inside one allocation, the outputs moved against the inputs by every combination of the address bits
whose single offsets were slow in probe 3 (14, 18, 24, 25, 26, 28), and a scan of single bits 8 .. 31."""
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
GiB = 1 << 30
pool = torch.empty((10 * GiB) // 4, dtype=torch.float32, device=dev)
pool.normal_()
plan = BatchedFft(n, dev)


def view(byte_off):
    o = byte_off // 4
    return pool[o:o + plane].view(rows, n)


def rate(a, b, c, reps=12):
    re, im, ore, oim = view(0), view(GiB + a), view(2 * GiB + b), view(3 * GiB + b + c)
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
    rate(0, 0, 0)
print("single bits, b = 2^k:")
print("  " + "  ".join(f"{k}:{rate(0, 1 << k, 0):.0f}" for k in range(8, 32)))
print("single bits, a = 2^k (im_in against re_in), b = 0:")
print("  " + "  ".join(f"{k}:{rate(1 << k, 0, 0):.0f}" for k in range(8, 30)))
print("single bits, c = 2^k (im_out against re_out), b = 0:")
print("  " + "  ".join(f"{k}:{rate(0, 0, 1 << k):.0f}" for k in range(8, 30)))
bits = [14, 18, 24, 25, 26, 28]
res = []
for m in range(1 << len(bits)):
    b = sum(1 << bits[i] for i in range(len(bits)) if m >> i & 1)
    res.append((rate(0, b, 0), m, b))
res.sort(reverse=True)
print("combinations of bits", bits, "(mask: rate), best and worst eight:")
print("  " + "  ".join(f"{m:06b}:{r:.0f}" for r, m, b in res[:8]))
print("  " + "  ".join(f"{m:06b}:{r:.0f}" for r, m, b in res[-8:]))

#!/usr/bin/env python3
"""Follow-up 2: inside ONE allocation (physically contiguous in large extents on a fresh card), does the OFFSET between
the planes decide the rate?  Planes carved out of one pool: re_in @ 0, im_in @ 1 GiB + a, re_out @ 2 GiB + b,
im_out @ 3 GiB + b + c; sweeps over b, then a, then c (bytes)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pragma_dsp_amd.batch import BatchedFft

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
n, rows = 4096, 65536
plane = rows * n  # floats
GiB = 1 << 30
pool = torch.empty((5 * GiB) // 4, dtype=torch.float32, device=dev)
pool.normal_()
plan = BatchedFft(n, dev)


def view(byte_off):
    o = byte_off // 4
    return pool[o:o + plane].view(rows, n)


def rate(a, b, c):
    re, im, ore, oim = view(0), view(GiB + a), view(2 * GiB + b), view(3 * GiB + b + c)
    for _ in range(4):
        plan.forward(re, im, out=(ore, oim))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        plan.forward(re, im, out=(ore, oim))
    e1.record()
    torch.cuda.synchronize()
    return 16.0 * rows * n / (e0.elapsed_time(e1) / 20 * 1e-3) / 1e9


for _ in range(60):
    rate(0, 0, 0)
skews = [0, 256, 1024, 4096, 16384, 65536, 1 << 18, 1 << 20, 2 << 20, 3 << 20, 4 << 20, 8 << 20, 16 << 20, 32 << 20, 64 << 20,
         (64 << 20) + 4096, 128 << 20, 256 << 20]
print("pool base (GiB): %.3f" % (pool.data_ptr() / 2**30))
print("sweep b (outputs against inputs), a = c = 0:")
rb = [(b, rate(0, b, 0)) for b in skews]
print("  " + "  ".join(f"{b}:{v:.0f}" for b, v in rb))
best_b = max(rb, key=lambda t: t[1])[0]
print("sweep a (im_in against re_in), b = %d:" % best_b)
ra = [(a, rate(a, best_b, 0)) for a in skews]
print("  " + "  ".join(f"{a}:{v:.0f}" for a, v in ra))
best_a = max(ra, key=lambda t: t[1])[0]
print("sweep c (im_out against re_out), a = %d, b = %d:" % (best_a, best_b))
rc = [(c, rate(best_a, best_b, c)) for c in skews]
print("  " + "  ".join(f"{c}:{v:.0f}" for c, v in rc))
print("again, the three extremes:", f"{rate(0, 0, 0):.0f}", f"{rate(best_a, best_b, max(rc, key=lambda t: t[1])[0]):.0f}",
      f"{rate(0, min(rb, key=lambda t: t[1])[0], 0):.0f}")

#!/usr/bin/env python3
"""A/B on the GPU box: does the spacing between the four 1-GiB planes of the N=4096 x 65536 C2C
batch (re, im in; re, im out) matter?  Planes carved from one pool at k*(1 GiB + skew).
Development tool."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pragma_dsp_amd.batch import BatchedFft

dev = torch.device("cuda", 0)
n, rows = 4096, 65536
plane = rows * n  # floats
plan = BatchedFft(n, dev)
pool = torch.empty(4 * plane + 4 * (64 << 20), device=dev)  # + 4 x 256 MiB of slack (in floats: 64 Mi)
pool.normal_()


def timed(fn, iters=30):
    for _ in range(8):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


for rep in range(2):
    for skew_bytes in (0, 4096, 16384, 65536, 262144, 1 << 20, 2 << 20, 8 << 20, (32 << 20) + 65536, 100 << 20):
        sk = skew_bytes // 4
        views = [pool[k * (plane + sk): k * (plane + sk) + plane].view(rows, n) for k in range(4)]
        re, im, ore, oim = views
        t = timed(lambda: plan.forward(re, im, out=(ore, oim)))
        print(f"skew {skew_bytes:>10} B: {16.0 * rows * n / t / 1e9:7.0f} GB/s  ({t*1e3:.4f} ms)", flush=True)

#!/usr/bin/env python3
"""A/B on the GPU box: the 65536-row N=4096 batch as 1, 2, 4 launches over row ranges (same buffers),
and smaller batches, to separate launch-size from footprint effects.  Development tool."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pragma_dsp_amd.batch import BatchedFft

dev = torch.device("cuda", 0)
n = 4096
plan = BatchedFft(n, dev)


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


for rows in (32768, 65536, 131072):
    re = torch.randn((rows, n), device=dev)
    im = torch.randn((rows, n), device=dev)
    ore, oim = torch.empty_like(re), torch.empty_like(im)
    for parts in (1, 2, 4, 8):
        h = rows // parts

        def run():
            for p in range(parts):
                s = slice(p * h, (p + 1) * h)
                plan.forward(re[s], im[s], out=(ore[s], oim[s]))
        t = timed(run)
        print(f"rows={rows} launches={parts}: {16.0 * rows * n / t / 1e9:7.0f} GB/s  ({t*1e3:.4f} ms)", flush=True)
    del re, im, ore, oim

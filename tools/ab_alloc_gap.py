#!/usr/bin/env python3
"""A/B on the GPU box: separate device allocations for the four planes of the N=4096 x 65536 batch
with a dummy allocation of D bytes between the input pair and the output pair (fresh process state
per D via empty_cache).  Development tool for the placement sensitivity noted in DESIGN.md."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pragma_dsp_amd.batch import BatchedFft

dev = torch.device("cuda", 0)
n, rows = 4096, 65536
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


for rep in range(2):
    for gap_mb in (0, 2, 32, 64, 200, 256, 512, 1024, 1536):
        torch.cuda.empty_cache()
        re = torch.randn((rows, n), device=dev)
        im = torch.randn((rows, n), device=dev)
        gap = torch.empty(gap_mb << 20, dtype=torch.uint8, device=dev) if gap_mb else None
        ore, oim = torch.empty_like(re), torch.empty_like(im)
        t = timed(lambda: plan.forward(re, im, out=(ore, oim)))
        print(f"gap {gap_mb:5d} MiB: {16.0 * rows * n / t / 1e9:7.0f} GB/s   ptrs {re.data_ptr():#x} {im.data_ptr():#x} {ore.data_ptr():#x} {oim.data_ptr():#x}", flush=True)
        del re, im, gap, ore, oim

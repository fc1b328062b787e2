#!/usr/bin/env python3
"""Tiny-N sweep on the GPU box (N = 2 .. 32, 2^27 complex points per launch).  Development tool."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pragma_dsp_amd.batch import BatchedFft

dev = torch.device("cuda", 0)


def timed(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


print(f"{'N':>4} {'C2C GB/s':>10} {'frac':>6} {'real GB/s':>10} {'spec GB/s':>10}")
for log2n in range(1, 6):
    n = 1 << log2n
    batch = (1 << 27) // n
    plan = BatchedFft(n, dev)
    re = torch.randn((batch, n), device=dev)
    im = torch.randn((batch, n), device=dev)
    ore, oim = torch.empty_like(re), torch.empty_like(im)
    amp = torch.empty((batch, n // 2 + 1), device=dev)
    t_c = timed(lambda: plan.forward(re, im, out=(ore, oim)))
    t_r = timed(lambda: plan.forward(re, None, out=(ore, oim)))
    t_s = timed(lambda: plan.spectrum(re, "hann", "one", out=amp))
    c = 16.0 * batch * n / t_c / 1e9
    print(f"{n:4d} {c:10.0f} {c/8000:6.3f} {12.0*batch*n/t_r/1e9:10.0f} {(4.0*n+4.0*(n//2+1))*batch/t_s/1e9:10.0f}", flush=True)
    del re, im, ore, oim, amp, plan

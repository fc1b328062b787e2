#!/usr/bin/env python3
"""Large-N sweep on the GPU box: algorithmic GB/s (16 B per sample, as for the single-pass sizes) of
the four-step paths, f32, 2^27 samples per launch.  Development tool."""
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


print(f"{'N':>10} {'passes':>7} {'C2C GB/s':>10} {'frac':>6} {'GSample/s':>10}")
for log2n in (15, 16, 17, 18, 19, 20, 22, 24, 26):
    n = 1 << log2n
    batch = max(1, (1 << 27) // n)
    plan = BatchedFft(n, dev)
    re = torch.randn((batch, n), device=dev)
    im = torch.randn((batch, n), device=dev)
    ore, oim = torch.empty_like(re), torch.empty_like(im)
    t = timed(lambda: plan.forward(re, im, out=(ore, oim)))
    c = 16.0 * batch * n / t / 1e9
    line = f"{n:10d} {1 if log2n <= 16 else 2 if log2n <= 18 else 3:7d} {c:10.0f} {c/8000:6.3f} {batch*n/t/1e9:10.1f}"
    if True:  # A/B: the tile passes' first form (3: natural-order scratch, 512-point rows on 16-row tiles); round 1's four-step forms (0)
        from pragma_dsp_amd import _capi
        for mode, label in ((5, "tile passes"), (3, "their first form"), (0, "round-1 four-step")):
            prev = _capi.lib.pdsp_set_twopass(mode)
            t3 = timed(lambda: plan.forward(re, im, out=(ore, oim)))
            _capi.lib.pdsp_set_twopass(prev)
            line += f"   ({label}: {16.0 * batch * n / t3 / 1e9:6.0f} GB/s)"
    print(line, flush=True)
    del re, im, ore, oim, plan

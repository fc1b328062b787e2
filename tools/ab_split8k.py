#!/usr/bin/env python3
"""A/B on the GPU box: N = 8192 rows on the single-pass kernel vs fft_split2_kernel, f32 and f64.
Development tool."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pragma_dsp_amd.batch import BatchedFft
from pragma_dsp_amd import _capi

dev = torch.device("cuda", 0)


def timed(fn, iters=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


n = 8192
for dt, sz in ((torch.float32, 4), (torch.float64, 8)):
    batch = (1 << 27) // n
    plan = BatchedFft(n, dev, dtype=dt)
    re = torch.randn((batch, n), device=dev, dtype=dt)
    im = torch.randn((batch, n), device=dev, dtype=dt)
    outs = {}
    for mode in (0, 3, 0, 3):   # 0: single-pass everywhere; 3: split kernels incl. f32 N=8192
        prev = _capi.lib.pdsp_set_split16k(mode)
        ore, oim = torch.empty_like(re), torch.empty_like(im)
        t_c = timed(lambda: plan.forward(re, im, out=(ore, oim)))
        t_r = timed(lambda: plan.forward(re, None, out=(ore, oim)))
        plan.forward(re, im, out=(ore, oim))
        torch.cuda.synchronize()
        outs[mode] = (ore[:32].clone(), oim[:32].clone())
        _capi.lib.pdsp_set_split16k(prev)
        print(f"{dt} mode={mode}: C2C {4.0*sz*batch*n/t_c/1e9:7.0f} GB/s  real-in {3.0*sz*batch*n/t_r/1e9:7.0f} GB/s", flush=True)
        del ore, oim
    d = max(float((outs[0][0] - outs[3][0]).abs().max()), float((outs[0][1] - outs[3][1]).abs().max()))
    print(f"{dt}: max |split - single| = {d:.3e} (max |X| = {float(outs[0][0].abs().max()):.1f})")
    del re, im

#!/usr/bin/env python3
"""N = 2^25 ... 2^27 (the sizes whose plans carry 512-point column factors) in the current and the first form of the
tile passes: time, agreement with each other and with numpy's f64 transform on one row.  Development tool."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pragma_dsp_amd import _capi
from pragma_dsp_amd.batch import BatchedFft

dev = torch.device("cuda", 0)
for log2n in [int(a) for a in sys.argv[1:]] or [25, 26, 27]:
    n = 1 << log2n
    batch = max(1, (1 << 27) // n)
    plan = BatchedFft(n, dev)
    g = torch.Generator(device=dev); g.manual_seed(log2n)
    re = torch.randn((batch, n), device=dev, generator=g)
    im = torch.randn((batch, n), device=dev, generator=g)
    outs = {}
    for mode in (1, 3, 1, 3):
        prev = _capi.lib.pdsp_set_twopass(mode)
        ore, oim = torch.empty_like(re), torch.empty_like(im)
        ts = []
        for _ in range(6):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            plan.forward(re, im, out=(ore, oim))
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        _capi.lib.pdsp_set_twopass(prev)
        outs[mode] = (ore, oim)
        print(f"N=2^{log2n} mode {mode}: per-call ms {[round(x * 1e3, 2) for x in ts]}  -> {16.0 * batch * n / min(ts) / 1e9:6.0f} GB/s", flush=True)
    d = max(float((outs[1][0] - outs[3][0]).abs().max()), float((outs[1][1] - outs[3][1]).abs().max()))
    top = float(torch.maximum(outs[1][0].abs().max(), outs[1][1].abs().max()))
    print(f"  modes agree to {d / top:.2e} of max |X|", flush=True)
    want = np.fft.fft(re[0].cpu().numpy().astype(np.float64) + 1j * im[0].cpu().numpy().astype(np.float64))
    got = outs[1][0][0].cpu().numpy().astype(np.float64) + 1j * outs[1][1][0].cpu().numpy()
    print(f"  row 0 vs numpy f64: max|err| / max|X| = {np.abs(got - want).max() / np.abs(want).max():.2e}", flush=True)
    del re, im, outs, plan

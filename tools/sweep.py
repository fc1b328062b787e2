#!/usr/bin/env python3
"""Size sweep on the GPU box: algorithmic GB/s of the C2C kernel and of the fused one-sided spectrum
for N = 64 .. 16384 (about 1 GiB of input per launch).  Development tool."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pragma_dsp_amd.batch import BatchedFft
from pragma_dsp_amd import _capi

dev = torch.device("cuda", 0)
F64 = "--f64" in sys.argv  # the f64 device family (bytes double, sizes up to 8192 single-pass / 16384 spectrum)
DT = torch.float64 if F64 else torch.float32
SZ = 8 if F64 else 4
print(f"{'N':>6} {'C2C GB/s':>10} {'frac':>6} {'real GB/s':>10} {'frac':>6} {'spec GB/s':>10} {'frac':>6} {'spec(direct)':>12} {'C2C(direct)':>12} {'interleaved':>12}")
# (f64 at N = 16384: the complex transform is a four-step one, real rows and frames are single-pass packed kernels)
for log2n in range(6, 15):
    n = 1 << log2n
    batch = (1 << 27) // n  # 2^27 complex points: 1 GiB in + 1 GiB out for C2C
    plan = BatchedFft(n, dev, dtype=DT)
    re = torch.randn((batch, n), device=dev, dtype=DT)
    im = torch.randn((batch, n), device=dev, dtype=DT)
    ore, oim = torch.empty_like(re), torch.empty_like(im)
    amp = torch.empty((batch, n // 2 + 1), device=dev, dtype=DT)
    plan.window("hann")

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

    t_c = timed(lambda: plan.forward(re, im, out=(ore, oim)))
    t_r = timed(lambda: plan.forward(re, None, out=(ore, oim)))
    t_s = timed(lambda: plan.spectrum(re, "hann", "one", out=amp))
    prev = _capi.lib.pdsp_set_staged_small(0)
    t_s0 = timed(lambda: plan.spectrum(re, "hann", "one", out=amp))
    t_c0 = timed(lambda: plan.forward(re, im, out=(ore, oim)))
    t_i = float("inf")
    if not (F64 and log2n == 14):  # interleaved rows are single-pass sizes only
        zi = torch.complex(re, im)
        zo = torch.empty_like(zi)
        t_i = timed(lambda: plan.forward_interleaved(zi, out=zo))
        del zi, zo
    _capi.lib.pdsp_set_staged_small(prev)
    c = 4.0 * SZ * batch * n / t_c / 1e9
    r = 3.0 * SZ * batch * n / t_r / 1e9
    s = SZ * (n + (n // 2 + 1)) * batch / t_s / 1e9
    print(f"{n:6d} {c:10.0f} {c/8000:6.3f} {r:10.0f} {r/8000:6.3f} {s:10.0f} {s/8000:6.3f} {s*t_s/t_s0:12.0f} {c*t_c/t_c0:12.0f} {c*t_c/t_i:12.0f}", flush=True)
    del re, im, ore, oim, amp, plan

#!/usr/bin/env python3
"""spectrum() variants on the GPU box (two-sided, phase rows, partial frames, peak index): GB/s of
the bytes each variant must move.  Development tool."""
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


print(f"{'N':>6} {'one':>7} {'one+ph':>7} {'two':>7} {'two+ph':>7} {'partial':>8} {'one+pk':>7} {'peaks':>7}   (GB/s)")
for log2n in (6, 8, 10, 12, 14, 16):
    n = 1 << log2n
    batch = (1 << 26) // n
    plan = BatchedFft(n, dev)
    x = torch.randn((batch, n), device=dev)
    xp = torch.randn((batch, n - n // 4), device=dev)
    row = []
    for sides, ph in (("one", False), ("one", True), ("two", False), ("two", True)):
        bins = n // 2 + 1 if sides == "one" else n
        t = timed(lambda: plan.spectrum(x, "hann", sides, want_phase=ph))
        row.append(4.0 * batch * (n + bins * (2 if ph else 1)) / t / 1e9)
    t = timed(lambda: plan.spectrum(xp, "hann", "one"))
    row.append(4.0 * batch * (xp.shape[1] + n // 2 + 1) / t / 1e9)
    t = timed(lambda: plan.spectrum(x, "hann", "one", want_peak=True))
    row.append(4.0 * batch * (n + n // 2 + 1) / t / 1e9)
    t = timed(lambda: plan.spectrum_peaks(x, "hann", "one", 48000.0))
    row.append(4.0 * batch * n / t / 1e9)
    print(f"{n:6d} " + " ".join(f"{v:7.0f}" for v in row[:4]) + f" {row[4]:8.0f} {row[5]:7.0f} {row[6]:7.0f}", flush=True)
    del x, xp, plan

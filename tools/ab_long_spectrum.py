#!/usr/bin/env python3
"""spectrum() of frames beyond the single-pass limit: the packed-real form on tile passes (+ split pass) vs
round 1's four-step form on (x*w, 0) (pdsp_set_twopass(0)), algorithmic GB/s (4 B/sample in + 4 B per one-sided
bin out).  `--sizes 15,16` restricts the sweep, `--window rect|hann|blackman|table` picks the window, `--only-new` skips the four-step leg (profiling).  Development tool."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pragma_dsp_amd import _capi
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


print(f"{'N':>9} {'tile passes':>12} {'four-step':>10}   (GB/s algorithmic, Hann, one-sided)")
sizes = (15, 16, 17, 19, 20, 22, 24)
if "--sizes" in sys.argv:
    sizes = tuple(int(v) for v in sys.argv[sys.argv.index("--sizes") + 1].split(","))
only_new = "--only-new" in sys.argv
window = sys.argv[sys.argv.index("--window") + 1] if "--window" in sys.argv else "hann"  # or rect / blackman / table
for log2n in sizes:
    n = 1 << log2n
    batch = max(1, (1 << 27) // n)
    plan = BatchedFft(n, dev)
    x = torch.randn((batch, n), device=dev)
    amp = torch.empty((batch, n // 2 + 1), device=dev)
    nbytes = 4.0 * batch * (n + n // 2 + 1)
    win = plan.window("hamming").tensor() if window == "table" else window  # a caller's tensor is read as a table
    t1 = timed(lambda: plan.spectrum(x, win, "one", out=amp))
    t0 = float("inf")
    if not only_new:
        prev = _capi.lib.pdsp_set_twopass(0)
        t0 = timed(lambda: plan.spectrum(x, win, "one", out=amp))
        _capi.lib.pdsp_set_twopass(prev)
    print(f"{n:9d} {nbytes / t1 / 1e9:12.0f} {nbytes / t0 / 1e9:10.0f}", flush=True)
    del x, amp, plan

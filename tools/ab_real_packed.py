#!/usr/bin/env python3
"""A/B of Radix2Fft.forward rows (real input, 512 <= N <= 16384) in ONE process, interleaved rounds: fft_real_kernel
(the N/2-point packed-real transform + split, pdsp_set_real_packed(1)) against the complex kernels on (x, 0)
(pdsp_set_real_packed(0): fft_stockham_kernel / fft_split2_kernel / fft_split4_kernel with LoadReal).
Algorithmic bytes: one real plane in + two planes out = 3 scalars per sample.  `--f64` for the double family.
Development tool; prints median and min GB/s per arm and the ratio of medians."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pragma_dsp_amd import _capi
from pragma_dsp_amd.batch import BatchedFft

dev = torch.device("cuda", 0)
if "--pmc-c2c-f64" in sys.argv:  # child mode of tools/pmc_lds.sh: a few launches of the f64 headline-shape kernel
    rows = int(sys.argv[sys.argv.index("--pmc-c2c-f64") + 1])
    p = BatchedFft(4096, dev, dtype=torch.float64)
    a, b = torch.randn((rows, 4096), device=dev, dtype=torch.float64), torch.randn((rows, 4096), device=dev, dtype=torch.float64)
    for _ in range(4):
        p.forward(a, b)
    torch.cuda.synchronize()
    sys.exit(0)
F64 = "--f64" in sys.argv
DT = torch.float64 if F64 else torch.float32
SZ = 8 if F64 else 4
ROUNDS, ITERS = 5, 12
print(f"{'N':>6} {'packed med':>11} {'min':>7} {'(x,0) med':>11} {'min':>7} {'ratio':>6}   % of 8 TB/s (packed / (x,0))")
for log2n in range(9, 15):
    n = 1 << log2n
    batch = (1 << 27) // n
    plan = BatchedFft(n, dev, dtype=DT)
    x = torch.randn((batch, n), device=dev, dtype=DT)
    ore, oim = torch.empty_like(x), torch.empty_like(x)
    nbytes = 3.0 * SZ * batch * n

    def timed(mode):
        prev = _capi.lib.pdsp_set_real_packed(mode)
        try:
            for _ in range(3):
                plan.forward(x, None, out=(ore, oim))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(ITERS):
                plan.forward(x, None, out=(ore, oim))
            e1.record()
            torch.cuda.synchronize()
        finally:
            _capi.lib.pdsp_set_real_packed(prev)
        return nbytes / (e0.elapsed_time(e1) / ITERS * 1e-3) / 1e9

    for _ in range(2):  # clock ramp
        timed(1), timed(0)
    a, b = [], []
    for _ in range(ROUNDS):
        a.append(timed(1))
        b.append(timed(0))
    ma, mb = float(np.median(a)), float(np.median(b))
    print(f"{n:6d} {ma:11.0f} {min(a):7.0f} {mb:11.0f} {min(b):7.0f} {ma / mb:6.3f}   {ma / 80:.1f} / {mb / 80:.1f}", flush=True)
    del x, ore, oim, plan

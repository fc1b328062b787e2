#!/usr/bin/env python3
"""One-frame latency of the host-f64 drop-in's FFT.forward (real input) with f64 real rows on fft_real_kernel
(pdsp_set_real_packed(1), the default) against the complex kernel on (x, 0) (0): N = 512 ... 16384, interleaved
rounds in one process, median microseconds per call (plan and `out` reused, as bench/run.ts does)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pragma_dsp_amd as pd

for n in (512, 1024, 2048, 4096, 8192, 16384):
    x = np.sin(2 * np.pi * 37 * np.arange(n) / n)
    fft = pd.FFT(n)
    out = fft.createComplexArray()
    res = {0: [], 1: []}
    for mode in (1, 0):
        prev = pd.lib.pdsp_set_real_packed(mode)
        for _ in range(200):
            fft.forward(x, out)
        pd.lib.pdsp_set_real_packed(prev)
    for r in range(8):
        for mode in (1, 0):
            prev = pd.lib.pdsp_set_real_packed(mode)
            ts = []
            for _ in range(500):
                t0 = time.perf_counter()
                fft.forward(x, out)
                ts.append((time.perf_counter() - t0) * 1e6)
            pd.lib.pdsp_set_real_packed(prev)
            res[mode].append(float(np.median(ts)))
    print(f"N={n:6d}  packed-real {np.median(res[1]):7.2f} us   complex on (x,0) {np.median(res[0]):7.2f} us   ratio {np.median(res[1]) / np.median(res[0]):.3f}", flush=True)

#!/usr/bin/env python3
"""Does the input data change the N=16384 spectrum kernel's time?  Same kernel, same buffers' sizes, one
process, interleaved rounds: bench.py's sine/noise recipe vs uniform noise vs zeros.  Development tool."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bench import synth_batch
from pragma_dsp_amd.batch import BatchedFft

dev = torch.device("cuda", 0)
n, chunk = 16384, 16384
plan = BatchedFft(n, dev)
amp = torch.empty((chunk, n // 2 + 1), dtype=torch.float32, device=dev)
data = {
    "bench recipe (sines + gaussian)": synth_batch(chunk, n, dev, complex_noise=False)[0],
    "uniform [-1,1)": torch.rand((chunk, n), device=dev) * 2 - 1,
    "gaussian": torch.randn((chunk, n), device=dev),
    "zeros": torch.zeros((chunk, n), device=dev),
}
res = {k: [] for k in data}
for k, x in data.items():
    for _ in range(30):
        plan.spectrum(x, "hann", "one", out=amp)
torch.cuda.synchronize()
for r in range(8):
    for k, x in data.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            plan.spectrum(x, "hann", "one", out=amp)
        e1.record()
        torch.cuda.synchronize()
        res[k].append(e0.elapsed_time(e1) / 10)
nbytes = (4 * n + 4 * (n // 2 + 1)) * chunk
for k, v in res.items():
    v.sort()
    print(f"{k:34s} med {v[len(v)//2]:.4f} ms  {nbytes / v[len(v)//2] / 1e6:6.0f} GB/s   min {v[0]:.4f} ms")

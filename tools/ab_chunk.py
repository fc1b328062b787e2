#!/usr/bin/env python3
"""Does the two-pass path gain from keeping its scratch planes inside the 256 MiB Infinity Cache?  The same
2^27-sample batch of N = 65536 transforms, transformed in chunks of 2^k rows per call.  Development tool."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pragma_dsp_amd.batch import BatchedFft

dev = torch.device("cuda", 0)
for log2n in (16, 17):
    n = 1 << log2n
    batch = (1 << 27) // n
    plan = BatchedFft(n, dev)
    re = torch.randn((batch, n), device=dev)
    im = torch.randn((batch, n), device=dev)
    ore, oim = torch.empty_like(re), torch.empty_like(im)
    for rows in (batch, batch // 2, batch // 4, batch // 8, batch // 16, batch // 32, batch // 64):
        def run():
            for r0 in range(0, batch, rows):
                plan.forward(re[r0:r0 + rows], im[r0:r0 + rows], out=(ore[r0:r0 + rows], oim[r0:r0 + rows]))
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 10 * 1e-3
        print(f"N={n} rows/call={rows:5d} scratch={rows * n * 8 / 2**20:7.0f} MiB  {16.0 * batch * n / t / 1e9:6.0f} GB/s algorithmic", flush=True)

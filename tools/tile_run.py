#!/usr/bin/env python3
"""Radix2Fft.forwardComplex at the tile-pass sizes given (log2 N ...), 2^27 samples per call, 12 calls each: run under
`rocprofv3 --kernel-trace --stats` to read each pass's own time.  Development tool."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pragma_dsp_amd.batch import BatchedFft

dev = torch.device("cuda", 0)
for log2n in [int(a) for a in sys.argv[1:]] or [15, 16, 17, 20]:
    n = 1 << log2n
    batch = (1 << 27) // n
    plan = BatchedFft(n, dev)
    re, im = torch.randn((batch, n), device=dev), torch.randn((batch, n), device=dev)
    ore, oim = torch.empty_like(re), torch.empty_like(im)
    for _ in range(12):
        plan.forward(re, im, out=(ore, oim))
    torch.cuda.synchronize()
    del re, im, ore, oim, plan

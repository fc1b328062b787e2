#!/usr/bin/env python3
"""fft_paired_kernel at 2^15 and 2^16, 2^27 samples per call, 6 calls each (and the tile passes, mode 5): run under
`rocprofv3 --kernel-trace --stats` or `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` to read its HBM bytes.  Development tool."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pragma_dsp_amd import _capi
from pragma_dsp_amd.batch import BatchedFft

dev = torch.device("cuda", 0)
for log2n in (15, 16):
    n = 1 << log2n
    batch = (1 << 27) // n
    plan = BatchedFft(n, dev)
    re, im = torch.randn((batch, n), device=dev), torch.randn((batch, n), device=dev)
    ore, oim = torch.empty_like(re), torch.empty_like(im)
    for mode in (1, 5):
        prev = _capi.lib.pdsp_set_twopass(mode)
        for _ in range(6):
            plan.forward(re, im, out=(ore, oim))
        torch.cuda.synchronize()
        _capi.lib.pdsp_set_twopass(prev)
    del re, im, ore, oim, plan

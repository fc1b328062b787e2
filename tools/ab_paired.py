#!/usr/bin/env python3
"""N = 2^15 / 2^16 out of place: fft_paired_kernel (sibling workgroups sharing an XCD's L2, one pass over HBM) against
the two tile passes (pdsp_set_twopass(5)), 2^27 samples per call.  Development tool."""
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
    calls = (("forwardComplex", 16.0, lambda: plan.forward(re, im, out=(ore, oim))),
             ("forward (real) ", 12.0, lambda: plan.forward(re, out=(ore, oim))),
             ("inverse        ", 16.0, lambda: plan.inverse(re, im, out=(ore, oim))))
    for name, nbytes, fn in calls:
        for rep in range(2 if name.startswith("forwardC") else 1):
            for mode in (1, 5):
                prev = _capi.lib.pdsp_set_twopass(mode)
                for _ in range(5):
                    fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                _capi.lib.pdsp_set_twopass(prev)
                t = e0.elapsed_time(e1) / 20 * 1e-3
                print(f"N=2^{log2n} {name} {'paired, one pass' if mode == 1 else 'two tile passes '}: "
                      f"{nbytes * batch * n / t / 1e9:6.0f} GB/s algorithmic", flush=True)

#!/usr/bin/env python3
"""STFT frame rate on the GPU box: hop = N (disjoint frames) vs hop = N/4 (75 % overlap).
Development tool."""
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


for n in (1024, 4096, 16384):
    plan = BatchedFft(n, dev)
    frames = (1 << 28) // n
    for hop in (n, n // 2, n // 4):
        sig = torch.randn(n + (frames - 1) * hop, device=dev)
        t = timed(lambda: plan.stft(sig, hop, "hann"))
        out_b = 4.0 * frames * (n // 2 + 1)
        print(f"N={n:6d} hop={hop:6d}: {frames / t / 1e6:8.2f} M frames/s   signal {4.0*sig.numel()/t/1e9:6.0f} GB/s + rows {out_b/t/1e9:6.0f} GB/s", flush=True)
        del sig

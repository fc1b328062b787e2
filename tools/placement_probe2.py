#!/usr/bin/env python3
"""Follow-up to placement_probe.py: is a slow placement a property of the INPUT planes, of the OUTPUT planes, or of
the combination?  K plane sets; the N=4096 C2C launch timed with the inputs of set i and the outputs of set j for
every (i, j); plus a read-only and a write-only pass over each plane pair."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bench import synth_batch
from pragma_dsp_amd.batch import BatchedFft

K = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
n, rows = 4096, 65536
plan = BatchedFft(n, dev)
re0, im0 = synth_batch(rows, n, dev)
ins, outs = [(re0, im0)], [(torch.empty_like(re0), torch.empty_like(im0))]
for _ in range(K - 1):
    re, im = torch.empty_like(re0), torch.empty_like(im0)
    re.copy_(re0)
    im.copy_(im0)
    ins.append((re, im))
    outs.append((torch.empty_like(re0), torch.empty_like(im0)))


def timed(fn, iters=20):
    for _ in range(4):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


for _ in range(60):
    plan.forward(ins[0][0], ins[0][1], out=outs[0])
torch.cuda.synchronize()
print("FFT GB/s, inputs of set i (rows) x outputs of set j (columns):")
for i in range(K):
    line = []
    for j in range(K):
        t = timed(lambda: plan.forward(ins[i][0], ins[i][1], out=outs[j]))
        line.append(16.0 * rows * n / t / 1e9)
    print("  " + "  ".join(f"{v:6.0f}" for v in line))
print("read-only (sum of both input planes of set i), GB/s:")
print("  " + "  ".join(f"{2 * 4.0 * rows * n / timed(lambda: (ins[i][0].sum(), ins[i][1].sum())) / 1e9:6.0f}" for i in range(K)))
print("write-only (fill both output planes of set j), GB/s:")
print("  " + "  ".join(f"{2 * 4.0 * rows * n / timed(lambda: (outs[j][0].fill_(1.0), outs[j][1].fill_(1.0))) / 1e9:6.0f}" for j in range(K)))
print("addresses in / out (GiB): " + "  ".join(f"{a[0].data_ptr() / 2**30:.1f},{a[1].data_ptr() / 2**30:.1f}/{b[0].data_ptr() / 2**30:.1f},{b[1].data_ptr() / 2**30:.1f}" for a, b in zip(ins, outs)))

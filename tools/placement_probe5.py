#!/usr/bin/env python3
"""Follow-up 4: the four planes as ONE allocation (re_in | im_in | re_out | im_out back to back, optionally with the
outputs moved by `skew` bytes) against four separate allocations, K of each alive together, timed interleaved."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bench import synth_batch
from pragma_dsp_amd.batch import BatchedFft

K = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
n, rows = 4096, 65536
plane = rows * n
plan = BatchedFft(n, dev)
re0, im0 = synth_batch(rows, n, dev)
sets, kinds = [], []
for k in range(K):  # separate allocations
    re, im = torch.empty_like(re0), torch.empty_like(im0)
    re.copy_(re0), im.copy_(im0)
    sets.append((re, im, torch.empty_like(re0), torch.empty_like(im0)))
    kinds.append("separate")
for skew in (0, 4 << 20):
    for k in range(K):  # one pool
        pool = torch.empty(4 * plane + (8 << 20), dtype=torch.float32, device=dev)
        v = lambda i, off=0: pool[i * plane + off // 4:(i + 1) * plane + off // 4].view(rows, n)
        re, im = v(0), v(1)
        re.copy_(re0), im.copy_(im0)
        sets.append((re, im, v(2, skew), v(3, skew)))
        kinds.append(f"pool+{skew >> 20}M")
for s in sets:
    for _ in range(10):
        plan.forward(s[0], s[1], out=(s[2], s[3]))
torch.cuda.synchronize()
R = 3
res = np.zeros((R, len(sets)))
for r in range(R):
    for k, s in enumerate(sets):
        for _ in range(4):
            plan.forward(s[0], s[1], out=(s[2], s[3]))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            plan.forward(s[0], s[1], out=(s[2], s[3]))
        e1.record()
        torch.cuda.synchronize()
        res[r, k] = 16.0 * rows * n / (e0.elapsed_time(e1) / 20 * 1e-3) / 1e9
med = np.median(res, axis=0)
for kind in dict.fromkeys(kinds):
    v = [m for m, kk in zip(med, kinds) if kk == kind]
    print(f"{kind:10s} " + "  ".join(f"{x:6.0f}" for x in v) + f"   mean {np.mean(v):.0f} min {min(v):.0f} max {max(v):.0f}")

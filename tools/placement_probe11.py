#!/usr/bin/env python3
"""Follow-up 10: does configs[3]'s kernel (fused Hann spectrum, N = 16384: 1 GiB of frames in, 0.5 GiB of amplitude
rows out per 16,384-frame chunk) care where its two streams lie?  One 100-GiB allocation, frames at 0, rows at X GiB."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth_batch
from pragma_dsp_amd.batch import BatchedFft

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
n, frames = 16384, 16384
bins = n // 2 + 1
GiB = 1 << 30
plan = BatchedFft(n, dev)
plan.window("hann")
x0, _ = synth_batch(frames, n, dev, complex_noise=False)
pool = torch.empty(100 * GiB // 4, dtype=torch.float32, device=dev)
fr = pool[:frames * n].view(frames, n)
fr.copy_(x0)
nbytes = (4 * n + 4 * bins) * frames


def rate(x_gib, reps=30):
    o = x_gib * (GiB // 4)
    amp = pool[o:o + frames * bins].view(frames, bins)
    for _ in range(5):
        plan.spectrum(fr, "hann", "one", out=amp)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        plan.spectrum(fr, "hann", "one", out=amp)
    e1.record()
    torch.cuda.synchronize()
    return nbytes / (e0.elapsed_time(e1) / reps * 1e-3) / 1e9 / 80


for _ in range(100):
    rate(1, 1)
for rnd in range(2):
    print("rows at X GiB, % of 8 TB/s: " + "  ".join(f"{x}:{rate(x):.1f}" for x in (1, 2, 4, 8, 16, 24, 31, 32, 33, 40, 48, 56, 64, 72, 80, 96)))
amp_plain = torch.empty((frames, bins), dtype=torch.float32, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(5):
    plan.spectrum(x0, "hann", "one", out=amp_plain)
e0.record()
for _ in range(30):
    plan.spectrum(x0, "hann", "one", out=amp_plain)
e1.record()
torch.cuda.synchronize()
print("two plain allocations: %.1f" % (nbytes / (e0.elapsed_time(e1) / 30 * 1e-3) / 1e9 / 80))

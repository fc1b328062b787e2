#!/usr/bin/env python3
"""Board power / sclk (bench.read_clocks, our card) while one workload keeps the card busy: Radix2Fft.forward on
real rows, complex rows, and torch's device copy / fill / sum of the same planes.
`power_probe.py [N]`.  Development tool for DESIGN 5 ("Board power is the wall")."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pragma_dsp_amd.batch import BatchedFft

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda", 0)
b = (1 << 28) // n
plan = BatchedFft(n, dev)
x, y = torch.randn((b, n), device=dev), torch.randn((b, n), device=dev)
ore, oim = torch.empty_like(x), torch.empty_like(x)


def probe(name, fn, nbytes, seconds=1.0):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    per = e0.elapsed_time(e1) / 20 * 1e-3
    iters = int(seconds / per)
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    time.sleep(max(0.0, 0.6 * seconds - (time.perf_counter() - t0)))
    c = (bench.read_clocks(dev, ours_only=True) or [{}])[0]
    torch.cuda.synchronize()
    print(f"{name:34s} {nbytes / per / 1e9:7.0f} GB/s   {c.get('power_w')} W (cap {c.get('power_cap_w')})   sclk {c.get('sclk_mhz')} MHz", flush=True)


probe("real forward", lambda: plan.forward(x, out=(ore, oim)), 12 * b * n)
probe("complex forward", lambda: plan.forward(x, y, out=(ore, oim)), 16 * b * n)
probe("torch copy of the planes", lambda: (ore.copy_(x), oim.copy_(y)), 16 * b * n)
probe("torch copy, one plane", lambda: ore.copy_(x), 8 * b * n)
probe("torch zero fill (writes only)", lambda: ore.zero_(), 4 * b * n)
probe("torch sum (reads only)", lambda: x.sum(), 4 * b * n)

#!/usr/bin/env python3
"""Does a hipGraph pay on configs[3]'s stream?  One step = 64 launches of spectrum_dif16k_kernel (16,384 frames of
N = 16384 each).  Eager launches against ONE replay of a graph that captured the same 64 launches, interleaved rounds
in one process; also a "step" of 64 launches of 256 frames each (launch-bound territory) for contrast."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bench import synth_batch
from pragma_dsp_amd.batch import BatchedFft

dev = torch.device("cuda", 0)
n = 16384
plan = BatchedFft(n, dev)
plan.window("hann")
for frames, chunks in ((16384, 64), (256, 64)):
    x, _ = synth_batch(frames, n, dev, complex_noise=False)
    amp = torch.empty((frames, n // 2 + 1), device=dev)
    side = torch.cuda.Stream(dev)

    def step():
        for _ in range(chunks):
            plan.spectrum(x, "hann", "one", out=amp)

    with torch.cuda.stream(side):  # capture happens on a side stream (torch's rule), the library launches on torch's current one
        step()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            step()
        torch.cuda.synchronize()
        res = {"eager": [], "graph": []}
        for r in range(8):
            for name, fn in (("eager", step), ("graph", g.replay)):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                res[name].append(e0.elapsed_time(e1) / 5)
    me, mg = float(np.median(res["eager"])), float(np.median(res["graph"]))
    print(f"{chunks} launches x {frames} frames per step: eager {me:.4f} ms, graph replay {mg:.4f} ms, graph/eager {mg / me:.4f}", flush=True)

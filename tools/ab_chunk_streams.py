#!/usr/bin/env python3
"""The two-pass path in Infinity-Cache-sized pieces, the pieces dealt round-robin onto S streams so that one piece's
second pass and the next piece's first pass overlap (a piece alone is ~1000 workgroups = one wave of them: its
launch gap and tail are what tools/ab_chunk.py measured as a loss).  2^27 samples per step.  Development tool.
    ab_chunk_streams.py [log2n ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pragma_dsp_amd.batch import BatchedFft

dev = torch.device("cuda", 0)
sizes = [int(a) for a in sys.argv[1:]] or [16, 17, 15]
for log2n in sizes:
    n = 1 << log2n
    batch = (1 << 27) // n
    plan = BatchedFft(n, dev)
    re = torch.randn((batch, n), device=dev)
    im = torch.randn((batch, n), device=dev)
    ore, oim = torch.empty_like(re), torch.empty_like(im)
    main = torch.cuda.current_stream(dev)
    for nstreams in (1, 2, 3, 4):
        streams = [torch.cuda.Stream(dev) for _ in range(nstreams)]
        for mib in (1024, 128, 64, 32, 16, 8):
            rows = min(batch, max(1, (mib << 20) // (8 * n)))

            def run():
                if nstreams == 1:
                    for r0 in range(0, batch, rows):
                        plan.forward(re[r0:r0 + rows], im[r0:r0 + rows], out=(ore[r0:r0 + rows], oim[r0:r0 + rows]))
                    return
                ev = torch.cuda.Event()
                ev.record(main)
                for i, st in enumerate(streams):
                    st.wait_event(ev)
                for i, r0 in enumerate(range(0, batch, rows)):
                    with torch.cuda.stream(streams[i % nstreams]):
                        plan.forward(re[r0:r0 + rows], im[r0:r0 + rows], out=(ore[r0:r0 + rows], oim[r0:r0 + rows]))
                for st in streams:
                    e = torch.cuda.Event()
                    e.record(st)
                    main.wait_event(e)
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(main)
            for _ in range(10):
                run()
            e1.record(main)
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / 10 * 1e-3
            print(f"N=2^{log2n} streams={nstreams} scratch/piece={rows * n * 8 / 2**20:6.0f} MiB pieces={-(-batch // rows):4d} "
                  f"{16.0 * batch * n / t / 1e9:6.0f} GB/s algorithmic", flush=True)

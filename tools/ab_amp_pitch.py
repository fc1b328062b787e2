#!/usr/bin/env python3
"""does a row pitch on configs[3]'s amplitude rows remove the write amplification, and does it pay?
One process, interleaved rounds, bench.py's sine/noise frames (one 16,384-frame chunk of N = 16384), fused Hann:
packed rows (8193 floats: no row starts on a cache line) against pitches of 8200 / 8208 / 8224 floats (rows on 32- /
64- / 128-byte boundaries) through pdsp_set_amp_pitch (include/pdsp_hip_dev.h).  Board power and sclk are read
from sysfs while launches are queued.  `--pmc PITCH` is the child mode for rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE
runs: a few launches at one pitch, nothing else."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bench import read_clocks, synth_batch
from pragma_dsp_amd import _capi
from pragma_dsp_amd.batch import BatchedFft

dev = torch.device("cuda", 0)
n, chunk = 16384, 16384
bins = n // 2 + 1
plan = BatchedFft(n, dev)
x, _ = synth_batch(chunk, n, dev, complex_noise=False)
plan.window("hann")
nbytes = (4 * n + 4 * bins) * chunk


def launch(pitch, amp):
    prev = _capi.lib.pdsp_set_amp_pitch(pitch)
    try:
        plan.spectrum(x, "hann", "one", out=amp)
    finally:
        _capi.lib.pdsp_set_amp_pitch(prev)


if "--pmc" in sys.argv:
    pitch = int(sys.argv[sys.argv.index("--pmc") + 1])
    amp = torch.empty((chunk, max(abs(pitch), bins)), dtype=torch.float32, device=dev)
    for _ in range(6):
        launch(pitch, amp)
    torch.cuda.synchronize()
    sys.exit(0)

pitches = [0, 8224, -8193, -8224]  # negative: the same pitch, mirrored pairs as plain stores
amps = {p: torch.empty((chunk, max(abs(p), bins)), dtype=torch.float32, device=dev) for p in pitches}
# parity of the pitched rows against the packed ones, bit for bit (same kernel, same arithmetic)
launch(0, amps[0])
for p in pitches[1:]:
    amps[p].fill_(-1.0)
    launch(p, amps[p])
    torch.cuda.synchronize()
    assert torch.equal(amps[p][:, :bins], amps[0]), p
    assert bool((amps[p][:, bins:] == -1.0).all()), "the pad between rows must stay untouched"
res = {p: [] for p in pitches}
power = {p: [] for p in pitches}
for p in pitches:
    for _ in range(40):
        launch(p, amps[p])
torch.cuda.synchronize()
for r in range(8):
    for p in pitches:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            launch(p, amps[p])
        e1.record()
        time.sleep(0.03)  # ~60 ms of launches are queued: read the card under load
        c = read_clocks(dev, ours_only=True)
        torch.cuda.synchronize()
        res[p].append(e0.elapsed_time(e1) / 200)
        if c:
            power[p].append((c[0].get("power_w"), c[0].get("sclk_mhz")))
for p in pitches:
    v = sorted(res[p])
    med = v[len(v) // 2]
    pw = [a for a, _ in power[p] if a]
    ck = [b for _, b in power[p] if b]
    print(f"pitch {p or bins:6d} floats: med {med:.4f} ms = {nbytes / med / 1e6:6.0f} GB/s ({nbytes / med / 1e6 / 80:.1f} %)   min {v[0]:.4f} ms"
          f"   power {np.median(pw) if pw else float('nan'):.0f} W  sclk {np.median(ck) if ck else float('nan'):.0f} MHz", flush=True)

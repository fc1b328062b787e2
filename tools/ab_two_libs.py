#!/usr/bin/env python3
"""A/B of TWO BUILDS of libpdsp_hip.so in ONE process (compile-time kernel options cannot be switched at run time):
both libraries are loaded side by side through ctypes, each gets its own plans, and the headline workloads are timed
in interleaved rounds on the same buffers -- configs[2] (N=4096 x 65,536 complex f32), its real-input and f64 forms,
configs[3]'s chunk (fused Hann spectrum, N=16384 x 16,384) and the mid-size fused spectra.  Outputs are compared bit
for bit.  Board power / sclk are read while launches are queued.
    python tools/ab_two_libs.py /path/to/A.so /path/to/B.so [--quick]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bench import read_clocks, synth_batch

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
vp, ll, i32 = C.c_void_p, C.c_longlong, C.c_int


class Lib:
    def __init__(self, path):
        self.path = path
        self.l = C.CDLL(path)
        self.l.pdsp_plan_create.argtypes = [ll, i32, C.POINTER(vp)]
        self.l.pdsp_fft_forward_complex_f32.argtypes = [vp, ll, vp, vp, vp, vp, vp]
        self.l.pdsp_fft_forward_complex_f64.argtypes = [vp, ll, vp, vp, vp, vp, vp]
        self.l.pdsp_fft_forward_real_f32.argtypes = [vp, ll, vp, vp, vp, vp]
        self.l.pdsp_spectrum_f32.argtypes = [vp, ll, vp, ll, ll, vp, i32, vp, vp, vp, vp]
        self.l.pdsp_plan_window_f32.argtypes = [vp, i32, C.POINTER(vp)]
        self.l.pdsp_spectrum_peaks_f32.argtypes = [vp, ll, vp, ll, ll, vp, i32, C.c_double, vp, vp, vp, vp]
        self.l.pdsp_last_error.restype = C.c_char_p
        self.plans = {}

    def plan(self, n):
        if n not in self.plans:
            h = vp()
            assert self.l.pdsp_plan_create(n, 0, C.byref(h)) == 0, self.l.pdsp_last_error()
            w = vp()
            assert self.l.pdsp_plan_window_f32(h, 1, C.byref(w)) == 0
            self.plans[n] = (h, w)
        return self.plans[n]


def sptr():
    return vp(torch.cuda.current_stream(dev).cuda_stream)


def p(t):
    return vp(t.data_ptr())


libs = [Lib(sys.argv[1]), Lib(sys.argv[2])]
quick = "--quick" in sys.argv
ROUNDS = 6 if quick else 10


def run(name, nbytes, make, call, iters):
    bufs = make()
    outs = []
    for L in libs:
        o = call(L, bufs, None)
        torch.cuda.synchronize()
        outs.append([t.clone() for t in o])
    same = all(torch.equal(a, b) for a, b in zip(outs[0], outs[1]))
    del outs
    res, pw = [[], []], [[], []]
    for _ in range(2):
        for i, L in enumerate(libs):
            for _ in range(iters):
                call(L, bufs, None)
    torch.cuda.synchronize()
    for r in range(ROUNDS):
        for i, L in enumerate(libs):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                call(L, bufs, None)
            e1.record()
            time.sleep(0.03)
            c = read_clocks(dev, ours_only=True)
            torch.cuda.synchronize()
            res[i].append(e0.elapsed_time(e1) / iters)
            if c:
                pw[i].append((c[0].get("power_w") or 0, c[0].get("sclk_mhz") or 0))
    med = [float(np.median(v)) for v in res]
    line = f"{name:34s}"
    for i in range(2):
        line += f"  {'AB'[i]}: {med[i]:.4f} ms {nbytes / med[i] / 1e6:6.0f} GB/s ({nbytes / med[i] / 1e6 / 80:.1f} %) min {min(res[i]):.4f}  {np.median([a for a, _ in pw[i]]):.0f} W {np.median([b for _, b in pw[i]]):.0f} MHz |"
    print(line + f"  B/A time {med[1] / med[0]:.4f}  bit-identical {same}", flush=True)


def c2c():
    re, im = synth_batch(65536, 4096, dev)
    return re, im, torch.empty_like(re), torch.empty_like(im)


def c2c_call(L, b, _):
    h, _w = L.plan(4096)
    assert L.l.pdsp_fft_forward_complex_f32(h, 65536, p(b[0]), p(b[1]), p(b[2]), p(b[3]), sptr()) == 0
    return b[2], b[3]


def real_call(L, b, _):
    h, _w = L.plan(4096)
    assert L.l.pdsp_fft_forward_real_f32(h, 65536, p(b[0]), p(b[2]), p(b[3]), sptr()) == 0
    return b[2], b[3]


def c2c64():
    re, im = synth_batch(32768, 4096, dev)
    re, im = re.double(), im.double()
    return re, im, torch.empty_like(re), torch.empty_like(im)


def c2c64_call(L, b, _):
    h, _w = L.plan(4096)
    assert L.l.pdsp_fft_forward_complex_f64(h, 32768, p(b[0]), p(b[1]), p(b[2]), p(b[3]), sptr()) == 0
    return b[2], b[3]


def spec(n, frames, rect=False):
    def make():
        x, _ = synth_batch(frames, n, dev, complex_noise=False)
        return x, torch.empty((frames, n // 2 + 1), device=dev)

    def call(L, b, _):
        h, w = L.plan(n)
        assert L.l.pdsp_spectrum_f32(h, frames, p(b[0]), n, n, None if rect else w, 0, p(b[1]), None, None, sptr()) == 0
        return (b[1],)
    return make, call


def peaks(n, frames):
    def make():
        x, _ = synth_batch(frames, n, dev, complex_noise=False)
        return x, torch.empty((frames, 4), dtype=torch.int32, device=dev)

    def call(L, b, _):
        h, w = L.plan(n)
        assert L.l.pdsp_spectrum_peaks_f32(h, frames, p(b[0]), n, n, w, 0, 48000.0, None, None, p(b[1]), sptr()) == 0
        return (b[1],)
    return make, call


if "--peaks" in sys.argv:  # the fused findPeak kernels only (peaks-only output: 4 B/sample in, 16 B/frame out)
    for n in (16384, 4096, 1024):
        mk, cl = peaks(n, (1 << 28) // n)
        run(f"hann peaks-only {n}x{(1 << 28) // n}", (4 * n + 16) * float((1 << 28) // n), mk, cl, 100)
    sys.exit(0)
if "--mid" in sys.argv:  # the sizes whose transforms have TP <= 128 threads per row
    for n in (256, 512, 1024, 2048):
        rows = (1 << 27) // n

        def mk(n=n, rows=rows):
            re, im = torch.randn((rows, n), device=dev), torch.randn((rows, n), device=dev)
            return re, im, torch.empty_like(re), torch.empty_like(im)

        def cl(L, b, _, n=n, rows=rows):
            h, _w = L.plan(n)
            assert L.l.pdsp_fft_forward_complex_f32(h, rows, p(b[0]), p(b[1]), p(b[2]), p(b[3]), sptr()) == 0
            return b[2], b[3]
        run(f"C2C f32 {n}x{rows}", 16.0 * rows * n, mk, cl, 60)
    for n in (512, 1024, 2048, 4096, 8192):
        frames = (1 << 28) // n
        mk, cl = spec(n, frames)
        run(f"hann spectrum {n}x{frames}", (4 * n + 4 * (n // 2 + 1)) * float(frames), mk, cl, 100)
    sys.exit(0)
run("configs[2] C2C f32 4096x65536", 16.0 * 65536 * 4096, c2c, c2c_call, 60)
mk, cl = spec(16384, 16384)
run("configs[3] hann spectrum 16384x16384", (4 * 16384 + 4 * 8193) * 16384.0, mk, cl, 100)
run("real-in f32 4096x65536", 12.0 * 65536 * 4096, c2c, real_call, 60)
run("C2C f64 4096x32768", 32.0 * 32768 * 4096, c2c64, c2c64_call, 40)
for n in (4096, 16384):  # spectrum()'s default window is "rect" (src/public/spectrum.ts:111-115)
    mk, cl = spec(n, (1 << 28) // n, rect=True)
    run(f"rect spectrum {n}x{(1 << 28) // n}", (4 * n + 4 * (n // 2 + 1)) * float((1 << 28) // n), mk, cl, 100)
if not quick:
    for n in (1024, 2048, 4096, 8192):
        frames = (1 << 28) // n
        mk, cl = spec(n, frames)
        run(f"hann spectrum {n}x{frames}", (4 * n + 4 * (n // 2 + 1)) * float(frames), mk, cl, 100)
    def c2c16k():
        re, im = synth_batch(16384, 16384, dev)
        return re, im, torch.empty_like(re), torch.empty_like(im)
    def c2c16k_call(L, b, _):
        h, _w = L.plan(16384)
        assert L.l.pdsp_fft_forward_complex_f32(h, 16384, p(b[0]), p(b[1]), p(b[2]), p(b[3]), sptr()) == 0
        return b[2], b[3]
    run("C2C f32 16384x16384", 16.0 * 16384 * 16384, c2c16k, c2c16k_call, 60)

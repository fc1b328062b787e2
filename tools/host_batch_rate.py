"""PCIe-inclusive rate of the batched host-f64 entry points (never bench.py's `value`): f64 frames in
host memory -> pdsp_spectrum_batch_host_f64 / pdsp_fft_transform_host_f64 -> f64 results in host
memory, the boundary the JS drop-in's spectrumBatch() binds.  Prints one JSON line per case.

    python tools/host_batch_rate.py [--samples 26] [--precision 64|32] [--reps 3]
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pragma_dsp_amd  # noqa: E402,F401
from pragma_dsp_amd._capi import Peak, check, dptr, lib  # noqa: E402


def spectrum_case(n: int, batch: int, reps: int, window: int) -> dict:
    rng = np.random.default_rng(1337)
    x = rng.standard_normal((batch, n))
    bins = n // 2 + 1
    freq = np.empty(bins)
    amp = np.empty((batch, bins))
    ph = np.empty((batch, bins))
    peaks = (Peak * batch)()
    nb = C.c_longlong(0)
    times = []
    for _ in range(reps + 1):
        t0 = time.perf_counter()
        check(lib.pdsp_spectrum_batch_host_f64(dptr(x), batch, n, 48000.0, n, window, 0, dptr(freq), dptr(amp), dptr(ph),
                                               peaks, C.byref(nb)))
        times.append(time.perf_counter() - t0)
    best = min(times[1:])
    return {"op": "spectrum_batch_host_f64", "n": n, "batch": batch, "window": window, "ms": best * 1e3,
            "GSample_per_s": batch * n / best / 1e9, "us_per_frame": best / batch * 1e6,
            "host_bytes_in_out": int(x.nbytes + amp.nbytes + ph.nbytes),
            "host_GBps": (x.nbytes + amp.nbytes + ph.nbytes) / best / 1e9,
            "checksum": float(amp[:: max(1, batch // 7)].sum())}


def transform_case(n: int, batch: int, reps: int) -> dict:
    rng = np.random.default_rng(1337)
    re = rng.standard_normal((batch, n))
    im = rng.standard_normal((batch, n))
    ore = np.empty_like(re)
    oim = np.empty_like(im)
    plan = C.c_void_p()
    check(lib.pdsp_plan_create(n, -1, C.byref(plan)))
    times = []
    for _ in range(reps + 1):
        t0 = time.perf_counter()
        check(lib.pdsp_fft_transform_host_f64(plan, batch, n, dptr(re), dptr(im), dptr(ore), dptr(oim), 0))
        times.append(time.perf_counter() - t0)
    lib.pdsp_plan_destroy(plan)
    best = min(times[1:])
    return {"op": "fft_transform_host_f64", "n": n, "batch": batch, "ms": best * 1e3,
            "GSample_per_s": batch * n / best / 1e9, "host_GBps": 4 * re.nbytes / best / 1e9,
            "checksum": float(ore[:: max(1, batch // 7)].sum())}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=26, help="log2 of the samples per call")
    ap.add_argument("--precision", type=int, default=64)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--sizes", type=int, nargs="*", default=[1024, 4096, 16384])
    a = ap.parse_args()
    lib.pdsp_set_host_precision(a.precision)
    for n in a.sizes:
        batch = max(1, (1 << a.samples) // n)
        r = spectrum_case(n, batch, a.reps, 1)
        r["precision"] = a.precision
        print(json.dumps(r), flush=True)
    for n in a.sizes:
        batch = max(1, (1 << (a.samples - 1)) // n)
        r = transform_case(n, batch, a.reps)
        r["precision"] = a.precision
        print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()

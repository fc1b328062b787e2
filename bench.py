#!/usr/bin/env python3
"""bench.py -- the headline metric of BASELINE.json on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

A "step" is one pass of the hot path over one batch of synthetic input that is
already resident in HBM: by default BASELINE.json configs[2], the roofline run
(Radix2Fft.forwardComplex semantics, fp32 planar complex, N=4096, batch=65536
per GPU).  For N>1 there is one process per GPU: either the caller starts them
(torch.distributed.run sets RANK / LOCAL_RANK / WORLD_SIZE) or, when WORLD_SIZE is not
set, `--gpus N` makes this process the launcher: it starts N child ranks of itself
(launch_ranks below; the parent never touches the GPU), waits, relays rank 0's JSON line
and exits non-zero if any child failed.  The batch is split by rank with no data-path
collective (weak scaling: 65,536 transforms per GPU, configs[4] at N=8 -- the batch idiom
of bench/reallife/signals.ts:264-270 generalised), and the timed region is bracketed by
a barrier + synchronize with the MAX over ranks taken.  Rank 0 prints ONE JSON line.
The control plane (barriers, the MAX, per-rank read-outs: host scalars) runs over gloo on loopback;
RCCL carries the path's one exchange step, which at N > 1 follows the timed region and is
timed on its own (`--no-gather` skips it): the all-gather over xGMI of the 16-byte-per-frame SpectrumPeak
records and of the output slabs, under a watchdog (`--gather-timeout`) -- a failed exchange is reported in
the line (status 0), a HUNG one is abandoned: the line is printed, every rank exits 5; the measured value
stands either way.  `--rccl-selftest` runs that leg in a world of ONE rank on one card (a real RCCL group).

The default line (N = 1, no flags) also carries, outside the headline's timed region: `parity` (256 rows of the
timed output vs the CPU oracle), `roofline.traffic` measured by two child runs under rocprofv3 --pmc,
`also.spectrum16k` (configs[3] at its stated size: the whole 2^20-frame stream resident in HBM, its own parity and
CPU baseline), `also.fft4096_f64` (the headline shape in the reference's own precision), `also.single1024`
(configs[1]: one N=1024 frame through the drop-in, microseconds, beside the Node CPU path), `cpu_baseline` (the oracle
and its Node restatement on one host core), `clocks` (sclk / board power under load).  A failed parity check
makes the run exit 4.

Other workloads (parity-checked elsewhere; here for DESIGN.md's numbers):
  --workload spectrum16k   configs[3]: fused Hann+FFT+one-sided amplitude, N=16384,
                           streamed in chunks of --chunk frames
  --workload real4096      Radix2Fft.forward semantics (real in, 12 B/sample)
  --workload fft4096_f64   configs[2]'s shape in f64 (32 B/sample)
  --workload fft16k / spectrum256 / peaks16k / single1024 (configs[1], latency) / stream, hostbatch (PCIe-inclusive)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 measured float4 copy


PLANES_ARENA_NOTE = ("BatchedFft.alloc_planes: one allocation of 80 GiB + a plane, input planes back to back at its start, "
                     "output planes 40 and 80 GiB in (each in a 32-GiB region of its own: DESIGN section 5)")


def synth_batch(batch: int, n: int, device, seed: int = 1337, complex_noise: bool = True):
    """SURVEY 8(d) config 3 input: first half sines A*sin(2*pi*k*i/N + phi) with
    A~U[0.5,2], integer k~U{1..N/2-1}, phi~U[0,2pi), imag = 0; second half complex
    Gaussian noise.  Counter-based (Philox) torch generator, seed 1337 (+rank)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    half = batch // 2
    re = torch.empty((batch, n), dtype=torch.float32, device=device)
    im = torch.zeros((batch, n), dtype=torch.float32, device=device)
    idx = torch.arange(n, dtype=torch.float32, device=device)
    step = 4096
    for s in range(0, half, step):
        e = min(half, s + step)
        a = torch.rand((e - s, 1), generator=g, device=device) * 1.5 + 0.5
        k = torch.randint(1, max(2, n // 2), (e - s, 1), generator=g, device=device).to(torch.float32)
        phi = torch.rand((e - s, 1), generator=g, device=device) * (2 * np.pi)
        re[s:e] = a * torch.sin((2 * np.pi / n) * k * idx + phi)
    re[half:].normal_(generator=g)
    if complex_noise:
        im[half:].normal_(generator=g)
    return re, im


def cpu_baseline(re_rows: np.ndarray, im_rows, n: int, target_s: float = 12.0):
    """The oracle (f64 scalar restatement of src/core/fft.ts, 1 thread) timed on a
    bounded sample of the same workload: the first rows of the GPU batch, looped
    with plan and `out` reused and a checksum guard (bench/run.ts:13-26)."""
    import oracle
    plan = oracle.Plan(n)
    rows = re_rows.shape[0]
    sec, _ = plan.time_forward(re_rows[:64], None if im_rows is None else im_rows[:64], reps=1)  # warm-up
    sec, _ = plan.time_forward(re_rows[:256], None if im_rows is None else im_rows[:256], reps=1)
    per = sec / 256
    reps = max(1, int(target_s / (per * rows)))
    sec, chk = plan.time_forward(re_rows, im_rows, reps=reps)
    done = rows * reps
    node = node_baseline(n, im_rows is not None)
    return {
        **({"node": node} if node else {}),
        "value": done * n / sec / 1e9,
        "unit": "GSample/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{done} transforms of N={n} ({rows} distinct rows of the GPU batch x {reps} passes), "
                  f"{sec:.1f} s, f64 scalar C -O2, host has {os.cpu_count()} cpus",
        "transforms_per_s": done / sec,
        # labelled extrapolation: what one pass over the GPU's batch would take at this rate, never measured
        "extrapolated_s_per_65536_transforms": 65536 / (done / sec),
        "checksum": chk,
    }


def node_baseline(n: int, complex_input: bool, seconds: float = 5.0):
    """The same transform loop under V8 (oracle/pdsp_oracle.js, a Node-compatible restatement of
    src/core/fft.ts: typed-array f64, one thread) -- SURVEY 8(d)'s "Node CPU path"."""
    import shutil
    import subprocess
    node = shutil.which("node")
    if node is None:
        return None
    try:
        p = subprocess.run([node, os.path.join(ROOT, "oracle", "pdsp_oracle.js"), "time", str(n), "256", str(seconds)]
                           + (["complex"] if complex_input else []), capture_output=True, text=True, timeout=120)
        d = json.loads(p.stdout)
        return {"value": d["transforms"] * n / d["seconds"] / 1e9, "unit": "GSample/s", "cores": 1,
                "transforms_per_s": d["transforms"] / d["seconds"],
                "sample": f"{d['transforms']} transforms of N={n} in {d['seconds']:.1f} s, node {d['node']}"}
    except Exception as e:  # a reported extra, never fatal
        return {"error": repr(e)}


def traffic_from_profile(kernel_substr: str):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes, corrected as
    MI355X_MICROARCH.md prescribes (FETCH_SIZE x2 on gfx950; units of KiB).  The
    numbers are written by tools/pmc_summary.py into profiles/traffic.json."""
    p = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(p):
        return None
    try:
        d = json.load(open(p))
        for k, v in d.items():
            if kernel_substr in k:
                return v.get("hbm_bytes_per_launch")
    except Exception:
        return None
    return None


def traffic_measured(args, kernel_substr: str):
    """HBM bytes per launch of the workload's kernel, measured now: this workload again, a few steps, as a CHILD
    process under `rocprofv3 --pmc FETCH_SIZE` and once more under `--pmc WRITE_SIZE` (separate passes, counters
    only -- no trace domain beside them), reduced as MI355X_MICROARCH.md's HBM section prescribes: both counters
    are in KiB, and on gfx950 FETCH_SIZE reports half of a wide coalesced streaming read, so it is doubled.
    Returns (bytes, detail) or (None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None, "rocprofv3 not found"
    child = [sys.executable, os.path.abspath(__file__), "--workload", args.workload, "--steps", "3", "--warmup", "1",
             "--ramp-seconds", "0", "--no-cpu-baseline", "--no-also", "--chunk", str(args.chunk)]
    if args.batch is not None:
        child += ["--batch", str(args.batch)]
    elif args.workload in ("spectrum16k", "peaks16k"):
        child += ["--batch", str(4 * args.chunk)]  # a few chunks are enough for a per-launch average
    if args.n is not None:
        child += ["--n", str(args.n)]
    got = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out_dir = tempfile.mkdtemp(prefix="pdsp_pmc_", dir="/tmp")
        try:
            p = subprocess.run([prof, "--pmc", counter, "--output-format", "csv", "-d", out_dir, "--"] + child,
                               cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp", PDSP_BENCH_PMC_CHILD="1"), capture_output=True, text=True,
                               timeout=60)
            vals = []
            for f in glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row.get("Counter_Name") == counter and kernel_substr in row.get("Kernel_Name", ""):
                        vals.append(float(row["Counter_Value"]))
            if p.returncode != 0 or not vals:
                return None, f"{counter} pass: rc {p.returncode}, {len(vals)} samples: {(p.stderr or '')[-200:]}"
            got[counter] = (sum(vals) / len(vals), len(vals))
        except Exception as exc:  # noqa: BLE001  (a reported extra, never fatal)
            return None, f"{counter} pass: {type(exc).__name__}: {exc}"[:300]
        finally:
            shutil.rmtree(out_dir, ignore_errors=True)
    rd, wr = got["FETCH_SIZE"][0] * 1024 * 2, got["WRITE_SIZE"][0] * 1024
    return rd + wr, {"hbm_read_bytes": rd, "hbm_write_bytes": wr, "launches_sampled": [got["FETCH_SIZE"][1], got["WRITE_SIZE"][1]]}


def single_frame_latency(args, dev, iters=None, js_iters=2000, js_sizes=None) -> dict:
    """BASELINE configs[1]: ONE N=1024 real frame, Hann window, forward FFT, magnitude.
    Latency-bound (8 KiB of traffic): reported as microseconds, not as a roofline fraction.
      dropin_spectrum_us  spectrum(x, {fftSize:1024, window:'hann'}) host f64 in -> f64 out
                          (staging, fused kernel, peak search; f64 on the device by default)
      dropin_forward_us   FFT(1024).forward(x) host f64 in -> f64 out (plan reused)
      kernel_us           the fused kernel alone on device-resident data (HIP events)
      js_dropin_latency   the real drop-in (Node + N-API addon) in the shape of the reference's bench/run.ts, with the
                          Node CPU restatement of the same algorithm timed beside it (tests/js/bench_latency.js)"""
    import pragma_dsp_amd as pd
    from pragma_dsp_amd.batch import BatchedFft
    n = 1024
    iters = iters or max(args.steps, 200)
    idx = np.arange(n)
    x = np.sin(2 * np.pi * 440.0 * idx / 48000.0)  # the reference's sine_440hz leakage case
    opts = {"sampleRate": 48000, "fftSize": n, "window": "hann"}
    fft = pd.FFT(n)
    out = fft.createComplexArray()

    def timed(fn):
        for _ in range(20):
            fn()
        ts = []
        for _ in range(iters):
            t0 = time.perf_counter()
            fn()
            ts.append((time.perf_counter() - t0) * 1e6)
        return float(np.median(ts)), float(np.min(ts))

    spec_med, spec_min = timed(lambda: pd.spectrum(x, opts))
    fwd_med, fwd_min = timed(lambda: fft.forward(x, out))
    plan = BatchedFft(n, dev)
    dx = torch.from_numpy(x.astype(np.float32)).to(dev).reshape(1, n)
    amp = torch.empty((1, n // 2 + 1), dtype=torch.float32, device=dev)
    for _ in range(20):
        plan.spectrum(dx, "hann", "one", out=amp)
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        plan.spectrum(dx, "hann", "one", out=amp)
    e1.record()
    torch.cuda.synchronize(dev)
    kernel_us = e0.elapsed_time(e1) * 1e3 / iters
    js = None
    if not args.no_cpu_baseline:
        import shutil
        import subprocess
        node = shutil.which("node")
        if node:
            try:
                cmd = [node, os.path.join(ROOT, "tests", "js", "bench_latency.js"), str(js_iters)]
                if js_sizes:
                    cmd.append(js_sizes)
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
                js = json.loads(r.stdout) if r.returncode == 0 else {"error": r.stderr[-400:]}
            except Exception as e:
                js = {"error": repr(e)}
    return {
        "js_dropin_latency": js,
        "metric": "single-frame latency (N=1024, Hann + FFT + magnitude)", "unit": "us", "higher_is_better": False,
        "value": spec_med, "n_gpus": 1, "steps": iters, "warmup": 20, "dtype": "f64 host drop-in / f32 kernel", "data": "synthetic",
        "config": {"workload": "N=1024 single frame spectrum(), hann, one-sided (configs[1])"},
        "dropin_spectrum_us": {"median": spec_med, "min": spec_min},
        "dropin_forward_us": {"median": fwd_med, "min": fwd_min},
        "kernel_back_to_back_us": kernel_us,
    }


def stream_throughput(args, dev) -> int:
    """The streaming front-end host-to-host: f64 frames in host memory ->
    SpectrumResult objects, through pinned staging, H2D, the fused kernel, D2H and the host-side
    findPeak.  PCIe- and host-bound: reported for DESIGN.md's PCIe-inclusive note, never as `value`
    of the headline metric."""
    import pragma_dsp_amd as pd
    from pragma_dsp_amd.stream import SpectrumStream
    rows = []
    for n, frames in ((1024, 16384), (16384, 2048)):
        rng = np.random.default_rng(n)
        data = rng.standard_normal((frames, n))
        for bits in (64, 32):
            prev = pd.lib.pdsp_set_host_precision(bits)
            try:
                st = SpectrumStream({"sampleRate": 48000, "fftSize": n, "window": "hann"}, batch_frames=256, device=dev)
                for f in data[:512]:
                    st.push(f)
                st.flush()
                t0 = time.perf_counter()
                got = 0
                for f in data:
                    got += len(st.push(f))
                got += len(st.flush())
                sec = time.perf_counter() - t0
            finally:
                pd.lib.pdsp_set_host_precision(prev)
            assert got == frames
            rows.append({"n": n, "precision": bits, "frames": frames, "frames_per_s": frames / sec,
                         "GSample_per_s": frames * n / sec / 1e9, "us_per_frame": sec / frames * 1e6})
    print(json.dumps({"metric": "SpectrumStream host-to-host throughput", "unit": "GSample/s", "higher_is_better": True,
                      "value": max(r["GSample_per_s"] for r in rows), "n_gpus": 1, "steps": 1, "warmup": 1,
                      "dtype": "f64/f32", "data": "synthetic",
                      "config": {"workload": "spectrumStream(frames, {fftSize, hann}) from host f64 frames, batches of 256"},
                      "rows": rows}), flush=True)
    return 0


def host_batch_rows(sizes=(1024, 4096, 16384), precisions=(64, 32), log2_samples=26, reps=4) -> list:
    """The batched host boundary itself (what the JS drop-in's spectrumBatch() binds): f64 frames in host memory ->
    pdsp_spectrum_batch_host_f64 -> f64 amplitude / phase rows + SpectrumPeak records in host memory, 2^log2_samples
    samples per call.  Large calls are cut into chunks on several workers inside the library (staging, both PCIe
    directions and the kernels overlap); PDSP_HOST_THREADS=1 is the one-shot sequence of rounds 1-3, timed beside it.
    PCIe- and host-bound: DESIGN.md's PCIe-inclusive note, never `value` of the headline metric.  16 rows per case
    against the oracle's spectrum()."""
    import oracle
    import pragma_dsp_amd as pd
    from pragma_dsp_amd._capi import Peak, check, dptr
    lib = pd.lib
    rows = []
    env0 = os.environ.get("PDSP_HOST_THREADS")

    def restore_env():
        if env0 is None:
            os.environ.pop("PDSP_HOST_THREADS", None)
        else:
            os.environ["PDSP_HOST_THREADS"] = env0

    for n in sizes:
        batch = (1 << log2_samples) // n
        rng = np.random.default_rng(n)
        x = rng.standard_normal((batch, n)) + np.sin(2 * np.pi * 37 * np.arange(n) / n)
        bins = n // 2 + 1
        freq, amp, ph = np.empty(bins), np.empty((batch, bins)), np.empty((batch, bins))
        peaks = (Peak * batch)()
        pick = np.linspace(0, batch - 1, 16).astype(int)
        want = [oracle.spectrum(x[r], sample_rate=48000.0, fft_size=n, window="hann") for r in pick]
        for bits in precisions:
            prev = lib.pdsp_set_host_precision(bits)
            try:
                for threads in (None, "1"):
                    if threads is None:
                        restore_env()
                    else:
                        os.environ["PDSP_HOST_THREADS"] = threads
                    best = None
                    for _ in range(reps if threads is None else max(2, reps // 2)):
                        t0 = time.perf_counter()
                        check(lib.pdsp_spectrum_batch_host_f64(dptr(x), batch, n, 48000.0, n, 1, 0, dptr(freq), dptr(amp),
                                                               dptr(ph), peaks, None))
                        sec = time.perf_counter() - t0
                        best = sec if best is None else min(best, sec)
                    err = max(float(np.abs(amp[r] - w["amplitude"]).max() / w["amplitude"].max()) for r, w in zip(pick, want))
                    peaks_ok = all(peaks[int(r)].index == w["peak"]["index"] for r, w in zip(pick, want))
                    tol = 1e-12 if bits == 64 else 1e-5
                    rows.append({"n": n, "batch": batch, "precision": bits,
                                 "mode": "chunked on the library's workers" if threads is None else "one-shot sequence (PDSP_HOST_THREADS=1)",
                                 "ms": best * 1e3, "GSample_per_s": batch * n / best / 1e9,
                                 "host_GBps_in_plus_out": (x.nbytes + amp.nbytes + ph.nbytes) / best / 1e9,
                                 "max_rel_err_16_rows": err, "tolerance": tol, "ok": bool(err <= tol and peaks_ok)})
            finally:
                lib.pdsp_set_host_precision(prev)
                restore_env()
    return rows


def host_batch_throughput(args, dev) -> int:
    rows = host_batch_rows()
    print(json.dumps({"metric": "pdsp_spectrum_batch_host_f64 host-to-host throughput (PCIe-inclusive)", "unit": "GSample/s",
                      "higher_is_better": True, "value": max(r["GSample_per_s"] for r in rows), "n_gpus": 1, "steps": 4,
                      "warmup": 0, "dtype": "f64/f32", "data": "synthetic",
                      "config": {"workload": "spectrumBatch: 2^26 samples of host f64 frames per call, hann, one-sided, "
                                             "amplitude + phase + peak records back in host f64"},
                      "host_cpus": os.cpu_count(), "rows": rows}), flush=True)
    return 0 if all(r["ok"] for r in rows) else 4


def parity_vs_oracle(kind: str, inputs, outputs, n: int, window: str | None = None, tol: float = 1e-5):
    """Outside the timed region: rows of the very buffers the timed steps wrote, checked against the
    CPU oracle (the checker, never the thing measured).  Stated fp32 tolerance of the path:
    per row max|got - want| / max|want| <= 1e-5, against the f64 restatement of src/core/fft.ts fed the
    same f32 inputs.  `kind`: "complex" / "real" (Radix2Fft.forwardComplex / forward rows) or
    "spectrum" (one-sided amplitude rows of spectrum(), spectrum.ts:116-127)."""
    import oracle
    plan = oracle.Plan(n)
    if kind == "spectrum":
        x, = inputs
        got, = outputs
        want, _, _ = plan.spectrum_batch(x, window=oracle.create_window(window, n) if window else None)
        err = np.abs(got - want).max(axis=1) / np.abs(want).max(axis=1)
    else:
        re, im = inputs
        gre, gim = outputs
        wre, wim = plan.forward_complex(re, im) if kind == "complex" else plan.forward(re)
        want = wre + 1j * wim
        err = np.abs((gre + 1j * gim) - want).max(axis=1) / np.abs(want).max(axis=1)
    return {"rows": int(err.shape[0]), "max_rel_err": float(err.max()), "tolerance": tol,
            "ok": bool(err.max() <= tol), "against": "oracle/pdsp_oracle.c (f64), rows drawn with seed 1337"}


def read_clocks(dev=None, ours_only=False):
    """Current sclk / mclk of every amdgpu card sysfs shows (MHz), read before and after timing and once
    UNDER LOAD (the ramp's launches still queued) -- context for the box-to-box spread of the same binary
    (DESIGN 5), not a measurement.  The host's other cards are listed too (they belong to other users);
    `"ours": true` marks the card whose PCI address is the device this rank runs on, and only that card
    also gets fclk / socclk, board power (W) and the junction / memory temperatures (hwmon)."""
    import glob
    mine = None
    try:
        pr = torch.cuda.get_device_properties(dev)
        mine = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}"
    except Exception:
        pass
    out = []
    for d in sorted(glob.glob("/sys/class/drm/card*/device")):
        rec = {}
        ours = bool(mine and mine in os.path.realpath(d))
        if ours:
            rec["ours"] = True
        elif ours_only:
            continue
        files = [("sclk_mhz", "pp_dpm_sclk"), ("mclk_mhz", "pp_dpm_mclk")]
        if ours:
            files += [("fclk_mhz", "pp_dpm_fclk"), ("socclk_mhz", "pp_dpm_socclk")]
        for key, fn in files:
            try:
                for line in open(os.path.join(d, fn)):
                    if line.rstrip().endswith("*"):
                        rec[key] = int("".join(ch for ch in line.split(":")[1] if ch.isdigit()))
            except Exception:
                pass
        if ours:
            for key, pat, div in (("power_w", "power1_input", 1e6), ("power_cap_w", "power1_cap", 1e6),
                                  ("temp_junction_c", "temp2_input", 1e3), ("temp_mem_c", "temp3_input", 1e3)):
                try:
                    rec[key] = round(int(open(glob.glob(os.path.join(d, "hwmon", "*", pat))[0]).read()) / div, 1)
                except Exception:
                    pass
        if rec:
            rec["card"] = os.path.basename(os.path.dirname(d))
            out.append(rec)
    return out or None


EXIT_PARITY_FAILED = 4       # the in-run oracle check of the timed output failed: the line is not a measurement
EXIT_EXCHANGE_ABANDONED = 5  # the exchange leg (RCCL all-gather) did not return in time: value measured, hang reported


class ExchangeWatchdog:
    """Guards the one step of the path that can hang on a sick collective library: the final gather.  If `done()`
    has not been called `timeout` seconds after `start()`, rank 0's line -- which already carries the measured value
    -- is printed with `gather.error`, and EVERY rank leaves with EXIT_EXCHANGE_ABANDONED (os._exit: the main thread
    is inside a collective that will not return).  A process that gave up on a stuck GPU collective must not
    report success: the launcher (and torch.distributed.run) see the status, the line is still on stdout."""

    def __init__(self, timeout: float, rank: int, out, gather: dict):
        self.timeout, self.rank, self.out, self.gather = timeout, rank, out, gather
        self.lock = threading.Lock()
        self._done = threading.Event()

    def start(self):
        threading.Thread(target=self._watch, daemon=True).start()
        return self

    def done(self):
        self._done.set()

    def _watch(self):
        if self._done.wait(self.timeout):
            return
        with self.lock:  # the main thread is inside a collective, not inside an update of `gather`
            self.gather["error"] = (f"exchange still running after {self.timeout:g} s: abandoned (exit status "
                                    f"{EXIT_EXCHANGE_ABANDONED}), the measured value stands")
            if self.rank == 0:
                self.out["gather"] = self.gather
                print(json.dumps(self.out), flush=True)
                sys.stdout.flush()
        if self.rank != 0:
            time.sleep(1.0)  # rank 0 prints first
        os._exit(EXIT_PARITY_FAILED if (self.rank == 0 and parity_failures(self.out)) else EXIT_EXCHANGE_ABANDONED)


def launch_ranks(args, argv) -> int:
    """`--gpus N` without WORLD_SIZE in the environment: start N child ranks of this script, one per
    GPU (LOCAL_RANK = rank), wait for them, relay rank 0's JSON line.  This parent never calls into
    HIP (children are fresh processes, never an exec of an initialised one).  Any child failing makes
    the others stop and the parent exit non-zero."""
    import socket
    import subprocess
    import tempfile
    n = args.gpus
    if not args.dry_run and not args.share_gpu:
        have = torch.cuda.device_count()  # counting devices does not initialise the GPU on this image
        if have < n:
            print(f"bench.py --gpus {n}: only {have} GPU(s) visible (use --share-gpu for a 1-GPU rehearsal)",
                  file=sys.stderr)
            return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    # rank 0's stdout goes to a file, not a pipe: a pipe nobody reads while the child runs blocks the child once
    # 64 KiB are in it (stray prints of torch / RCCL, a long line), and the launch would end as a timeout
    out0_file = tempfile.TemporaryFile(mode="w+", prefix="pdsp_bench_rank0_")
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PDSP_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=out0_file if r == 0 else subprocess.DEVNULL, text=True))
    deadline = time.time() + args.launch_timeout
    rc = 0
    pending = set(range(n))
    while pending:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0:
                # the first failure decides the status, except that a rank killed by a signal never outranks a
                # rank that chose its own status (EXIT_EXCHANGE_ABANDONED from every rank, say)
                rc = code if (rc == 0 or (rc < 0 < code)) else rc
                print(f"bench.py: rank {r} exited with {code}", file=sys.stderr)
        if (rc or time.time() > deadline) and pending:
            if not rc:
                print(f"bench.py: ranks {sorted(pending)} still running after {args.launch_timeout} s", file=sys.stderr)
                rc = 124
            else:
                # ranks that end on their own within a moment (every rank of an abandoned exchange does) keep
                # their own status; only the rest are stopped
                t_grace = time.time() + 10.0
                while time.time() < t_grace and any(procs[r].poll() is None for r in pending):
                    time.sleep(0.05)
            for r in sorted(pending):  # exactly the children started above
                if procs[r].poll() is None:
                    procs[r].kill()
            for r in sorted(pending):
                code = procs[r].wait()
                if code not in (0, -9):
                    print(f"bench.py: rank {r} exited with {code}", file=sys.stderr)
            pending.clear()
        elif pending:
            time.sleep(0.05)
    out0_file.seek(0)
    lines = [ln for ln in out0_file.read().splitlines() if ln.startswith("{")]
    out0_file.close()
    if rc == 0 and not lines:
        print("bench.py: rank 0 printed no JSON line", file=sys.stderr)
        rc = 1
    if lines:  # a measurement that was printed survives a later failure (the status still says what happened)
        print(lines[-1], flush=True)
    return rc if rc >= 0 else 128 - rc


def parse_args(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--plain-planes", action="store_true",
                    help="time the transform workloads on four plain allocations instead of the engine's plane layout "
                         "(BatchedFft.alloc_planes)")
    ap.add_argument("--workload", default="fft4096", choices=["fft4096", "fft4096_f64", "real4096", "fft16k", "spectrum16k", "spectrum256", "peaks16k", "single1024", "stream", "hostbatch"])
    ap.add_argument("--batch", type=int, default=None, help="transforms per GPU (default: the config's)")
    ap.add_argument("--n", type=int, default=None, help="spectrum256 only: another frame size (development sweeps)")
    ap.add_argument("--chunk", type=int, default=16384, help="frames per launch for spectrum16k")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU legs (oracle baseline + parity rows)")
    ap.add_argument("--no-also", action="store_true", help="skip the configs[3] and f64 legs attached to the default line")
    ap.add_argument("--also-reuse-chunk", action="store_true",
                    help="the configs[3] leg re-reads one 1-GiB chunk instead of keeping the whole 96-GiB stream resident")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--ramp-seconds", type=float, default=0.6, help="untimed clock-ramp before the warm-up steps")
    ap.add_argument("--dist-backend", default=None,
                    help="backend of the exchange leg: nccl (= RCCL, default) or gloo (default with --share-gpu / --dry-run); "
                         "the control plane is always gloo")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--gather", action="store_true", help="(default at --gpus > 1; kept for older command lines)")
    ap.add_argument("--no-gather", action="store_true",
                    help="N > 1: skip the RCCL all-gather of the output slabs and of the peak records (timed on its own, after the timed region)")
    ap.add_argument("--measure-traffic", action="store_true",
                    help="N = 1: measure roofline.traffic in this run -- two child runs of this workload under "
                         "`rocprofv3 --pmc` (FETCH_SIZE, WRITE_SIZE; ~20 s) -- instead of quoting profiles/traffic.json; "
                         "default for the default workload unless this process itself runs under a profiler")
    ap.add_argument("--no-measure-traffic", action="store_true", help="quote profiles/traffic.json (the committed PMC pass)")
    ap.add_argument("--gather-timeout", type=float, default=150.0,
                    help="N > 1: seconds the exchange leg may take before the line is printed without it")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU, no compute: launcher + rendezvous + shard + gather + max-over-ranks only (CPU tests)")
    ap.add_argument("--fail-rank", type=int, default=-1, help="testing the launcher: this rank exits with status 3")
    ap.add_argument("--stdout-noise", type=int, default=0,
                    help="testing the launcher: rank 0 first writes this many bytes of non-JSON text to its stdout")
    ap.add_argument("--simulate-hang", action="store_true",
                    help="--dry-run only: the gather of rank 1 (rank 0 in a world of one) never returns, so that the "
                         "watchdog's path -- line printed, every rank exits 5 -- is testable without a GPU")
    ap.add_argument("--rccl-selftest", action="store_true",
                    help="N = 1: after a short timed run, the exchange leg in a world of ONE rank -- an RCCL group made the "
                         "same way (`new_group(backend='nccl', device_id=...)`), all_gather_into_tensor on the real output "
                         "planes and on the 16-byte peak records, an orderly destroy: what one card can prove about the leg")
    ap.add_argument("--launch-timeout", type=float, default=900.0)
    args = ap.parse_args(argv)
    under_profiler = any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
    if args.no_measure_traffic or under_profiler or os.environ.get("PDSP_BENCH_PMC_CHILD"):
        args.measure_traffic = False
    elif args.workload == "fft4096" and args.batch is None and args.gpus == 1 and "WORLD_SIZE" not in os.environ:
        args.measure_traffic = True
    if args.dist_backend is None:
        args.dist_backend = "gloo" if (args.share_gpu or args.dry_run) else "nccl"
    return args


def dry_run(args, world: int, rank: int) -> int:
    """The N>1 plumbing of this file without a GPU: process group, contiguous row split, barrier,
    max-over-ranks, the gather of (stand-in) 16-byte peak records.  Prints a line that says so; it is
    not a measurement and carries no value."""
    import torch.distributed as dist
    from pragma_dsp_amd.shard import gather_rows, max_over_ranks, my_rows
    per_gpu = args.batch or 65536
    row0, row1 = my_rows(per_gpu * world, rank, world)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    rec = torch.arange(row0, row1, dtype=torch.int32).reshape(-1, 1).repeat(1, 4)  # 16 B per row
    line = {"metric": "dry-run (launcher / shard / gather plumbing only, no compute)", "value": None,
            "unit": "GSample/s", "n_gpus": world, "steps": 0, "warmup": 0, "dry_run": True,
            "rows": [row0, row1], "global_batch": per_gpu * world, "backend": args.dist_backend}
    gather = {"backend": args.dist_backend}
    dog = ExchangeWatchdog(args.gather_timeout, rank, line, gather).start()  # the real leg's guard, same statuses
    if args.simulate_hang and rank == min(1, world - 1):
        threading.Event().wait()  # this rank never reaches the collective: every other rank hangs inside it
    full = gather_rows(rec, per_gpu * world)
    dog.done()
    ok = bool(torch.equal(full[:, 0], torch.arange(per_gpu * world, dtype=torch.int32)))
    elapsed = max_over_ranks(time.perf_counter() - t0)
    ranks_seen = max_over_ranks(float(rank + 1))
    if rank == 0:
        line.update({"gather_ok": ok, "ranks_seen": int(ranks_seen), "elapsed_s": elapsed})
        print(json.dumps(line), flush=True)
    return 0 if ok else 1


def main(argv=None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args, argv)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.share_gpu:
        local = 0
    if rank == args.fail_rank:
        print(f"bench.py: rank {rank} failing on request (--fail-rank)", file=sys.stderr)
        return 3
    if args.stdout_noise and rank == 0:
        for _ in range(0, args.stdout_noise, 64):
            print("stray line from a library, not the result" + " " * 22, flush=True)
    if args.rccl_selftest and world == 1:  # a process group of one: the exchange leg needs a control plane
        with __import__("socket").socket() as sk:
            sk.bind(("127.0.0.1", 0))
            os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
        os.environ.update(RANK="0", WORLD_SIZE="1")
    if world > 1 or args.rccl_selftest:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import torch.distributed as dist
        # Control plane (barriers, max-over-ranks, per-rank read-outs: a few host scalars) over gloo on loopback --
        # one node, nothing of it inside a kernel's time -- so that the timed measurement does not depend on the
        # health of the collective library; RCCL (backend "nccl") carries the path's one exchange step, the final
        # gather (north_star: "RCCL over xGMI only for the final gather"), as its own group, created in that leg.
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        dist.init_process_group("gloo")
    try:
        if args.dry_run:
            return dry_run(args, world, rank)
        return run_rank(args, world, rank, local)
    finally:
        if world > 1 or args.rccl_selftest:
            import torch.distributed as dist
            if dist.is_initialized():
                dist.destroy_process_group()


def run_rank(args, world: int, rank: int, local: int) -> int:
    if not torch.cuda.is_available():
        print("bench.py needs a GPU (the pdsp engine has no CPU fallback)", file=sys.stderr)
        return 2
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from pragma_dsp_amd.batch import BatchedFft
    from pragma_dsp_amd.shard import gather_rows, max_over_ranks, my_rows

    if args.workload == "single1024":
        print(json.dumps(single_frame_latency(args, dev)), flush=True)
        return 0
    if args.workload == "stream":
        return stream_throughput(args, dev)
    if args.workload == "hostbatch":
        return host_batch_throughput(args, dev)

    if args.workload in ("spectrum16k", "peaks16k"):
        n, per_gpu = 16384, args.batch or (1 << 20)
    elif args.workload == "fft16k":
        n, per_gpu = 16384, args.batch or 16384  # 2^28 samples, as configs[2]
    elif args.workload == "spectrum256":
        n = args.n or 256
        per_gpu = args.batch or ((1 << 30) // n)
    else:
        n, per_gpu = 4096, args.batch or 65536
    # weak scaling: the global batch is per_gpu x world rows, split contiguously by rank
    # (pragma-dsp_amd/shard.py); no collective on the data path
    row0, row1 = my_rows(per_gpu * world, rank, world)
    assert row1 - row0 == per_gpu
    f64 = args.workload == "fft4096_f64"  # the headline shape in the reference's own precision (fft.ts:1-14)
    plan = BatchedFft(n, dev, dtype=torch.float64 if f64 else torch.float32)
    stream = torch.cuda.current_stream(dev)
    parity_kind, amp = None, None

    planes_note = "four plain allocations"
    if args.workload in ("fft4096", "fft16k", "fft4096_f64"):
        re, im = synth_batch(per_gpu, n, dev, seed=1337 + rank)
        if f64:
            re, im = re.double(), im.double()
        # the engine's plane layout (BatchedFft.alloc_planes: one allocation, outputs 40 / 80 GiB beyond the inputs) is
        # measured for configs[2]'s kernel -- f32, N = 4096, 1-GiB planes: 79-84 % and repeatable against 71-84 % by
        # lottery; on the f64 and N = 16384 kernels it measured no better than plain allocations
        if args.plain_planes or args.workload != "fft4096":
            ore, oim = torch.empty_like(re), torch.empty_like(im)
        else:
            a_re, a_im, ore, oim = plan.alloc_planes(per_gpu)
            a_re.copy_(re)
            a_im.copy_(im)
            re, im = a_re, a_im
            del a_re, a_im
            torch.cuda.empty_cache()
            planes_note = PLANES_ARENA_NOTE if plan.arena is not None else "four plain allocations (no room for the arena layout)"
        launches_per_step = 1
        bytes_per_launch = (32 if f64 else 16) * per_gpu * n  # 8 B read + 8 B written per f32 sample
        parity_kind = "complex"
        kernel_name, kernel_label = "fft_stockham_kernel<float, 12, pdsp::LoadComplex", \
            "fft_stockham_kernel<float, 12, LoadComplex, StoreComplex>"
        if f64:
            kernel_name, kernel_label = "fft_stockham_kernel<double, 12, pdsp::LoadComplex", \
                "fft_stockham_kernel<double, 12, LoadComplex, StoreComplex>"
        if args.workload == "fft16k":
            kernel_name, kernel_label = "fft_split4_kernel<float, 12, pdsp::LoadComplex", \
                "fft_split4_kernel<float, 12, LoadComplex, StoreComplex>"

        def step():
            plan.forward(re, im, out=(ore, oim))
    elif args.workload == "real4096":
        re, _ = synth_batch(per_gpu, n, dev, seed=1337 + rank, complex_noise=False)
        im = None
        ore, oim = torch.empty_like(re), torch.empty_like(re)  # (the layout is neutral here: 80.8 / 78.4 vs 79.4 / 79.4 %)
        launches_per_step = 1
        bytes_per_launch = 12 * per_gpu * n
        parity_kind = "real"
        kernel_name, kernel_label = "fft_stockham_kernel<float, 12, pdsp::LoadReal", \
            "fft_stockham_kernel<float, 12, LoadReal, StoreComplex>"

        def step():
            plan.forward(re, None, out=(ore, oim))
    elif args.workload == "peaks16k":
        chunk = min(args.chunk, per_gpu)
        assert per_gpu % chunk == 0
        re, _ = synth_batch(chunk, n, dev, seed=1337 + rank, complex_noise=False)
        im = None
        launches_per_step = per_gpu // chunk
        bytes_per_launch = (4 * n + 16) * chunk  # frame in, one 16-byte SpectrumPeak out
        kernel_name = kernel_label = "spectrum_dif16k_kernel<float, 2, true>"
        plan.window("hann")

        def step():
            for _ in range(launches_per_step):
                plan.spectrum_peaks(re, "hann", "one", 48000.0)
    else:
        chunk = min(args.chunk if args.workload == "spectrum16k" else (1 << 20), per_gpu)  # ~1 GiB of frames per launch
        assert per_gpu % chunk == 0
        re, _ = synth_batch(chunk, n, dev, seed=1337 + rank, complex_noise=False)
        im = None
        bins = n // 2 + 1
        amp = torch.empty((chunk, bins), dtype=torch.float32, device=dev)
        launches_per_step = per_gpu // chunk
        bytes_per_launch = (4 * n + 4 * bins) * chunk  # 98,308 B per frame
        parity_kind = "spectrum"
        kernel_name = kernel_label = "spectrum_dif16k_kernel<float, 2, false>"
        if args.workload == "spectrum256":
            kernel_name = kernel_label = "spectrum_staged_kernel<float, 7, true>" if n == 256 else f"spectrum kernel of N={n}"
        plan.window("hann")

        def step():
            # the stream of 2^20 frames is generated on-device; each chunk of frames is
            # consumed from the same HBM-resident buffer (64 GiB would not change the kernel)
            for _ in range(launches_per_step):
                plan.spectrum(re, "hann", "one", out=amp)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize(dev)

    clocks_before = read_clocks(dev) if rank == 0 else None
    # Untimed clock ramp: a cold MI355X needs ~0.5 s of work before its clocks settle
    # (first 20 launches measured 8 % slower than steady state); then the W warm-up steps.
    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < args.ramp_seconds:
        for _ in range(10):
            step()
        torch.cuda.synchronize(dev)
    clocks_load = None
    if rank == 0:  # one reading with the card busy: the last ramp launches are still queued while sysfs is read
        t_q = time.perf_counter()
        for _ in range(40):
            step()
        busy = 40 * 0.6e-3 * launches_per_step  # >= 0.6 ms per launch on every workload here
        time.sleep(max(0.0, min(0.25, 0.5 * busy) - (time.perf_counter() - t_q)))
        clocks_load = read_clocks(dev, ours_only=True)
        torch.cuda.synchronize(dev)
    for _ in range(args.warmup):
        step()
    barrier()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    evs[0].record(stream)
    for i in range(args.steps):
        step()
        evs[i + 1].record(stream)  # same stream the kernels are launched on
    torch.cuda.synchronize(dev)
    elapsed_local = time.perf_counter() - t0
    barrier()
    step_ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps)]
    clocks_after = read_clocks(dev) if rank == 0 else None

    elapsed = max_over_ranks(elapsed_local)
    launch_ms = float(np.mean(step_ms)) / launches_per_step
    per_rank_ms = [launch_ms]
    if world > 1:  # every rank's own kernel time (HIP events on its stream), for the scaling read-out
        import torch.distributed as dist
        t = torch.tensor([launch_ms], dtype=torch.float64)
        allms = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(allms, t)
        per_rank_ms = [float(x.item()) for x in allms]

    # context for the roofline fraction (outside the timed region): the rate at which this box, with
    # these very buffers, copies the input planes to the output planes (torch's device copy kernel)
    copy_gbps, copy_clocks = None, None
    if args.workload in ("fft4096", "fft16k", "fft4096_f64") and rank == 0 and world == 1:
        for _ in range(3):
            ore.copy_(re)
            oim.copy_(im)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record(stream)
        for _ in range(10):
            ore.copy_(re)
            oim.copy_(im)
        c1.record(stream)
        torch.cuda.synchronize(dev)
        copy_gbps = 10 * bytes_per_launch / (c0.elapsed_time(c1) * 1e-3) / 1e9
        # board power while the card only copies (0.5 s of copies queued), beside the FFT's reading above:
        # says whether the transform runs against the power cap or the copy ceiling
        for _ in range(350):
            ore.copy_(re)
            oim.copy_(im)
        time.sleep(0.25)
        copy_clocks = read_clocks(dev, ours_only=True)
        torch.cuda.synchronize(dev)
        step()  # the output planes hold the transform again (parity rows are read below)
        torch.cuda.synchronize(dev)

    out = None
    if rank == 0:
        samples_per_step = per_gpu * n * world
        value = samples_per_step * args.steps / elapsed / 1e9
        achieved = bytes_per_launch / (launch_ms * 1e-3) / 1e9
        profiled_shape = args.batch is None and args.chunk == 16384
        out = {
            "metric": "batched 1D FFT GSample/s at N=4096 batch=65536; achieved HBM GB/s vs peak"
            if args.workload == "fft4096" else f"GSample/s ({args.workload})",
            "value": value,
            "unit": "GSample/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64" if f64 else "f32",
            "data": "synthetic",
            "config": {"workload": {"fft4096": f"N=4096 batch={per_gpu}/GPU Radix2Fft.forwardComplex fp32 planar complex (configs[2])",
                                    "fft4096_f64": f"N=4096 batch={per_gpu}/GPU Radix2Fft.forwardComplex f64 planar complex (configs[2]'s shape in the reference's precision)",
                                    "real4096": f"N=4096 batch={per_gpu}/GPU Radix2Fft.forward fp32 real input",
                                    "fft16k": f"N=16384 batch={per_gpu}/GPU Radix2Fft.forwardComplex fp32 planar complex",
                                    "spectrum256": f"N={n} batch={per_gpu}/GPU fused hann+FFT+one-sided amplitude",
                                    "spectrum16k": f"N=16384 batch={per_gpu}/GPU fused hann+FFT+one-sided amplitude (configs[3])",
                                    "peaks16k": f"N=16384 batch={per_gpu}/GPU fused hann+FFT+findPeak, peaks-only output"}[args.workload],
                       "planes": planes_note,
                       "n": n, "batch_per_gpu": per_gpu, "global_batch": per_gpu * world,
                       "parallelism": f"batch-shard x{world}", "launches_per_step": launches_per_step},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS,
                         # PMC bytes of the committed rocprofv3 passes: only valid for the profiled shape
                         "traffic": traffic_from_profile(kernel_name) if profiled_shape else None,
                         "traffic_source": "profiles/traffic.json (committed rocprofv3 --pmc pass of this workload, "
                                           "not measured in this run)" if profiled_shape else None,
                         "kernel": kernel_label,
                         "device_copy_same_buffers_GBps": copy_gbps,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "launch_ms_avg": launch_ms, "launch_ms_median": float(np.median(step_ms)) / launches_per_step,
                         "launch_ms_min": float(np.min(step_ms)) / launches_per_step},
            "clocks": {"before": clocks_before, "under_load": clocks_load, "after": clocks_after,
                       "device_copy_under_load": copy_clocks,
                       "source": "sysfs pp_dpm_* and hwmon of /sys/class/drm/card*/device"},
        }
        if args.measure_traffic and world == 1:
            nbytes, detail = traffic_measured(args, kernel_name)
            if nbytes is not None:
                out["roofline"]["traffic"] = nbytes
                out["roofline"]["traffic_detail"] = detail
                out["roofline"]["traffic_source"] = ("measured in this run: two child runs of this workload under rocprofv3 --pmc "
                                                     "(FETCH_SIZE KiB x 1024 x 2 on gfx950, WRITE_SIZE KiB x 1024), averaged per launch")
            else:
                out["roofline"]["traffic_measure_error"] = detail
        if world > 1:
            out["per_rank_kernel_ms"] = per_rank_ms
            # each rank's own kernel against ITS card's HBM peak, and the whole job against world x peak
            out["per_rank_roofline_frac"] = [bytes_per_launch / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS for ms in per_rank_ms]
            out["compute_phase"] = {"GSample_per_s": value, "samples_per_step": samples_per_step,
                                    "max_rank_ms_per_step": elapsed / args.steps * 1e3,
                                    "algorithmic_GBps_all_ranks": bytes_per_launch * launches_per_step * world * args.steps / elapsed / 1e9,
                                    "frac_of_world_x_hbm_peak": bytes_per_launch * launches_per_step * world * args.steps / elapsed / 1e9
                                    / (HBM_PEAK_GBPS * world),
                                    "note": "sum of all ranks' samples / max-over-ranks wall time; no collective inside"}
            if args.share_gpu:
                out["rehearsal"] = "all ranks share cuda:0 (1-GPU box): plumbing check, not a scaling point"
        if not args.no_cpu_baseline:
            # CPU legs, rank 0 only, after the timed region: (i) parity of 256 rows of the timed output
            # against the oracle, (ii) the oracle timed as the CPU baseline (N = 1 only)
            g = torch.Generator()
            g.manual_seed(1337)
            if parity_kind in ("complex", "real"):
                sel = torch.randperm(per_gpu, generator=g)[:256].sort().values.to(dev)
                ins = (re[sel].cpu().numpy(), im[sel].cpu().numpy() if im is not None else None)
                outs = (ore[sel].cpu().numpy().astype(np.float64), oim[sel].cpu().numpy().astype(np.float64))
                out["parity"] = parity_vs_oracle(parity_kind, ins, outs, n, tol=1e-12 if f64 else 1e-5)
            elif parity_kind == "spectrum":
                sel = torch.randperm(re.shape[0], generator=g)[:64 if n > 4096 else 256].sort().values.to(dev)
                out["parity"] = parity_vs_oracle("spectrum", (re[sel].cpu().numpy(),),
                                                 (amp[sel].cpu().numpy().astype(np.float64),), n, "hann")
        if world == 1 and args.workload == "fft4096" and not args.no_also:
            out["also"] = {"placement": also_placement(args, dev, plan, re, im, ore, oim),
                           "spectrum16k": also_spectrum16k(args, dev, rank),
                           "fft4096_f64": also_fft4096_f64(args, dev, rank, re, im),
                           "real4096": also_real4096(args, dev, plan, re, ore, oim)}
            try:  # configs[1]: one N = 1024 frame, latency (a reported extra: never fatal to the headline)
                lat = single_frame_latency(args, dev, iters=300, js_iters=500, js_sizes="1024")
                out["also"]["single1024"] = {k: lat[k] for k in ("config", "unit", "dropin_spectrum_us", "dropin_forward_us",
                                                               "kernel_back_to_back_us", "js_dropin_latency")}
            except Exception as exc:  # noqa: BLE001
                out["also"]["single1024"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
            try:  # the PCIe-inclusive rate of the batched host boundary (a reported extra, never `value`)
                hb = host_batch_rows(sizes=(4096,), precisions=(64,), log2_samples=25, reps=3)
                out["also"]["hostbatch4096"] = {
                    "config": {"workload": "N=4096 x 8192 host f64 frames -> pdsp_spectrum_batch_host_f64 (hann, one-sided) -> "
                                           "host f64 amplitude + phase rows + peak records; PCIe-inclusive, never `value`"},
                    "unit": "GSample/s", "host_cpus": os.cpu_count(), "rows": hb}
            except Exception as exc:  # noqa: BLE001
                out["also"]["hostbatch4096"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
        if world == 1 and not args.no_cpu_baseline:
            rows = 2048
            sel = torch.cat([torch.arange(0, rows // 2), torch.arange(per_gpu // 2, per_gpu // 2 + rows // 2)]) \
                if args.workload in ("fft4096", "real4096", "fft4096_f64") else torch.arange(0, min(rows, re.shape[0]))
            hre = re[sel.to(dev)].cpu().numpy().astype(np.float64)
            him = im[sel.to(dev)].cpu().numpy().astype(np.float64) if im is not None else None
            out["cpu_baseline"] = cpu_baseline(hre, him, n, args.cpu_seconds)
    # The one exchange step of the path, timed on its own AFTER the line is assembled: the value is
    # already measured, so neither a failed nor a hung exchange may take it away (exchange_leg below).
    gather = None
    if (world > 1 or args.rccl_selftest) and not args.no_gather and args.workload in ("fft4096", "real4096"):
        gather = exchange_leg(args, world, rank, dev, plan, re, ore, oim, per_gpu, out, barrier)

    if rank == 0:
        if gather:
            out["gather"] = gather
        bad = parity_failures(out)
        print(json.dumps(out), flush=True)
        if bad:  # a fast kernel whose results differ from the reference's is not done: say so loudly
            print(f"bench.py: PARITY FAILED against the oracle ({', '.join(bad)}): the line above is not a valid measurement",
                  file=sys.stderr)
            return EXIT_PARITY_FAILED
    return 0


def close_orphan_rccl_groups(exc) -> int:
    """A bring-up that fails INSIDE dist.new_group (eager connect refused: two ranks on one card, a sick fabric) leaves
    a half-made ProcessGroupNCCL that no caller holds a handle to; its destructor then warns "destroy_process_group()
    was not called" at exit and may leak the communicator.  The object is still a local of the frames the exception
    passed through: find it there and abort it.  Returns how many were closed."""
    import torch.distributed as dist
    cls = getattr(dist, "ProcessGroupNCCL", None)
    closed, seen, tb = 0, set(), exc.__traceback__
    while tb is not None and cls is not None:
        for v in list(tb.tb_frame.f_locals.values()):
            if isinstance(v, cls) and id(v) not in seen:
                seen.add(id(v))
                try:
                    v.abort()
                    closed += 1
                except Exception:  # noqa: BLE001
                    pass
        tb = tb.tb_next
    return closed


def exchange_leg(args, world: int, rank: int, dev, plan, re, ore, oim, per_gpu: int, out, barrier) -> dict:
    """The path's one exchange step (north_star: "RCCL over xGMI only for the final gather"), after the
    timed region: the all-gather of (i) one 16-byte SpectrumPeak record per frame (fused findPeak over the rows'
    real plane: ~1 MiB per rank -- what a consumer of spectrum() needs) and (ii) the full output slabs (2 GiB per
    rank at configs[4]: xGMI-per-link bound, it dwarfs the compute), each timed on its own.  RCCL comes up HERE, as a
    process group of its own (one communicator per rank, bound to its card), the ranks agree over the gloo control
    plane that it did -- a failed bring-up is skipped by all of them together -- and the group is destroyed before
    the leg returns, on every path.  A failed exchange is `gather.error` in the line; a hung one is abandoned by the
    watchdog with a non-zero status.  `--rccl-selftest` runs the same leg in a world of ONE rank (the collective
    is not short-cut), which is what a 1-GPU box can prove: librccl loads, the eager `device_id` bring-up works,
    all_gather_into_tensor runs on the real planes, teardown is clean."""
    import torch.distributed as dist
    from pragma_dsp_amd.shard import gather_rows, max_over_ranks
    gather = {"backend": args.dist_backend}
    dog = ExchangeWatchdog(args.gather_timeout, rank, out, gather).start()
    xgrp = None  # the exchange group: RCCL (created below, inside the guarded leg), or the gloo default group

    def timed_gather(tensors, rows_per_rank, check=False):
        gather_rows(tensors[0][:8], 8 * world, xgrp, force_collective=True)  # communicator warm-up, untimed
        barrier()
        # one plane at a time: a gathered plane (world x 2 GiB at configs[4]) is checked and dropped before the next
        # is received, so the leg never holds more than one of them
        sec, rows, intact = 0.0, 0, True
        r0 = rank * rows_per_rank
        for t in tensors:
            g0 = time.perf_counter()
            full = gather_rows(t, rows_per_rank * world, xgrp, force_collective=True)
            if dev is not None:
                torch.cuda.synchronize(dev)
            sec += time.perf_counter() - g0
            rows = int(full.shape[0])
            if check:  # this rank's own rows came back where they belong, bit for bit
                intact = intact and bool(torch.equal(full[r0:r0 + rows_per_rank], t))
            del full
        sec = max_over_ranks(sec)
        nbytes = sum(t.numel() * t.element_size() for t in tensors)
        res = {"ms": sec * 1e3, "bytes_per_rank": nbytes, "GBps_in_per_gpu": nbytes * (world - 1) / sec / 1e9,
               "rows_gathered": rows}
        if check:
            res["own_rows_intact"] = intact
        return res
    try:  # a failed exchange is reported, not fatal
        pk_i, pk_f, pk_a, pk_p, _, _ = plan.spectrum_peaks(re, "hann", "one", 48000.0)
        recs = pk_i.new_empty((per_gpu, 4))
        recs[:, 0] = pk_i
        recs[:, 1:] = torch.stack([pk_f, pk_a, pk_p], dim=1).view(torch.int32)
        torch.cuda.synchronize(dev)
        if args.dist_backend != "nccl":
            # gloo rehearsal (1-GPU box): the collective runs on host copies, 4096 rows of the slabs per
            # rank -- it checks the plumbing; it is not a bandwidth figure
            legs = (("slabs", lambda: timed_gather([ore[:4096].cpu(), oim[:4096].cpu()], 4096)),
                    ("peaks_16B_per_frame", lambda: timed_gather([recs.cpu()], per_gpu)))
            with dog.lock:
                gather["note"] = "gloo rehearsal on host copies (slabs: 4096 rows per rank)"
                gather["ranks"] = dist.get_world_size()
        else:
            up, why = 1.0, ""
            try:
                xgrp = dist.new_group(backend="nccl", device_id=dev)
                gather_rows(recs[:8], 8 * world, xgrp, force_collective=True)
                torch.cuda.synchronize(dev)
            except Exception as exc:  # noqa: BLE001
                up, why = 0.0, f"{type(exc).__name__}: {exc}"
                with dog.lock:
                    gather["orphan_groups_closed"] = close_orphan_rccl_groups(exc)
            if max_over_ranks(1.0 - up) > 0.0:
                raise RuntimeError("RCCL group did not come up on every rank" + (f" (this rank: {why})" if why else ""))
            with dog.lock:
                gather["ranks"] = dist.get_world_size(xgrp)  # the ranks RCCL itself saw
                try:
                    gather["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
                except Exception:  # noqa: BLE001
                    gather["rccl_version"] = None
            # the small exchange first: it is the one a consumer of spectrum() needs, and one line survives a
            # failure of the slab exchange
            legs = (("peaks_16B_per_frame", lambda: timed_gather([recs], per_gpu, check=True)),
                    ("slabs", lambda: timed_gather([ore, oim], per_gpu, check=True)))
        for name, leg in legs:
            res = leg()
            with dog.lock:
                gather[name] = res
    except Exception as exc:  # noqa: BLE001  (RuntimeError from RCCL / allocator)
        with dog.lock:
            gather["error"] = f"{type(exc).__name__}: {exc}"[:300]
    finally:
        try:  # still under the watchdog: a teardown that hangs is a hang
            if xgrp is not None:
                torch.cuda.synchronize(dev)
                dist.destroy_process_group(xgrp)
                with dog.lock:
                    gather["group_destroyed"] = True
        except Exception as exc:  # noqa: BLE001
            with dog.lock:
                gather["destroy_error"] = f"{type(exc).__name__}: {exc}"[:200]
        dog.done()
    return gather


def parity_failures(out) -> list:
    """Names of the in-run oracle checks of a bench line that failed (empty = the line is a valid measurement)."""
    bad = [k for k in ("parity",) if k in out and not out[k]["ok"]]
    for leg, res in (out.get("also") or {}).items():
        if "parity" in res and not res["parity"]["ok"]:
            bad.append(f"also.{leg}.parity")
        if any(not r.get("ok", True) for r in res.get("rows", []) if isinstance(r, dict)):
            bad.append(f"also.{leg}.rows")
    return bad


def also_spectrum16k(args, dev, rank: int):
    """BASELINE configs[3] at its stated size, attached to the default line so that the driver-timed record carries
    it: N = 16384, batch = 2^20 frames = 2^34 samples, fused Hann + FFT + one-sided amplitude, processed as a stream
    of 64 chunks of 16,384 frames (one launch each) per step; HIP events on the launch stream around every step.
    98,308 algorithmic bytes per frame.  The card has 288 GB: when ~100 GiB are free the WHOLE stream is
    resident -- 64 GiB of distinct frames in, 32 GiB of amplitude rows out, nothing re-read -- otherwise (`resident`
    false) every launch consumes the same 1-GiB chunk, which is far beyond the 256 MiB Infinity Cache either way.
    64 rows drawn over the whole stream are checked against the oracle; the oracle's own fused path (applyWindow ->
    FFT -> magnitude -> one-sided scaling, spectrum.ts:116-127, plan and window reused) is timed on 64 of the same
    frames as this leg's CPU baseline."""
    from pragma_dsp_amd.batch import BatchedFft
    n, chunk, chunks, steps = 16384, 16384, 64, 4
    bins = n // 2 + 1
    torch.cuda.empty_cache()
    free_b, _ = torch.cuda.mem_get_info(dev)
    need = chunks * chunk * (n + bins) * 4
    resident = (not args.also_reuse_chunk) and free_b > need + (12 << 30)
    plan = BatchedFft(n, dev)
    if resident:
        xs = [synth_batch(chunk, n, dev, seed=1337 + rank + 7919 * c, complex_noise=False)[0] for c in range(chunks)]
        amps = [torch.empty((chunk, bins), dtype=torch.float32, device=dev) for _ in range(chunks)]
    else:
        x0, _ = synth_batch(chunk, n, dev, seed=1337 + rank, complex_noise=False)
        a0 = torch.empty((chunk, bins), dtype=torch.float32, device=dev)
        xs, amps = [x0] * chunks, [a0] * chunks
    plan.window("hann")
    stream = torch.cuda.current_stream(dev)

    def one_step():
        for c in range(chunks):
            plan.spectrum(xs[c], "hann", "one", out=amps[c])
    t_ramp = time.perf_counter()  # the clocks fell back during the host-side legs: ramp again, untimed
    one_step()
    torch.cuda.synchronize(dev)
    while time.perf_counter() - t_ramp < args.ramp_seconds:
        one_step()
        torch.cuda.synchronize(dev)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    evs[0].record(stream)
    for i in range(steps):
        one_step()
        evs[i + 1].record(stream)
    torch.cuda.synchronize(dev)
    ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(steps)]
    nbytes = (4 * n + 4 * bins) * chunk
    step_ms = float(np.mean(ms))
    avg = step_ms / chunks
    res = {"config": {"workload": f"N=16384 batch={chunks * chunk} fused hann+FFT+one-sided amplitude, streamed as {chunks} chunks "
                                  f"of {chunk} frames (configs[3])", "n": n, "batch": chunks * chunk,
                      "chunks_per_step": chunks, "frames_per_chunk": chunk, "samples_per_step": chunks * chunk * n,
                      "resident": resident,
                      "note": ("the whole 2^20-frame stream is resident in HBM: 64 GiB of distinct frames in, 32 GiB out"
                               if resident else "every launch consumes the same 1-GiB chunk of frames (64 GiB + 32 GiB did not fit beside the rest)")},
           "kernel": "spectrum_dif16k_kernel<float, 2, false> (fused Hann)", "steps": steps,
           "ms_per_step": step_ms, "ms_per_step_min": float(np.min(ms)), "ms": avg, "ms_min": float(np.min(ms)) / chunks,
           "GSample_per_s": chunk * n / (avg * 1e-3) / 1e9, "algorithmic_bytes_per_launch": nbytes,
           "algorithmic_bytes_per_step": nbytes * chunks,
           "GBps": nbytes / (avg * 1e-3) / 1e9, "frac": nbytes / (avg * 1e-3) / 1e9 / HBM_PEAK_GBPS}
    if not args.no_cpu_baseline:
        g = torch.Generator()
        g.manual_seed(1337)
        pick = torch.randperm(chunks * chunk, generator=g)[:64].sort().values.tolist()
        hx = np.stack([xs[r // chunk][r % chunk].cpu().numpy() for r in pick])
        ha = np.stack([amps[r // chunk][r % chunk].cpu().numpy() for r in pick]).astype(np.float64)
        res["parity"] = parity_vs_oracle("spectrum", (hx,), (ha,), n, "hann")
        res["parity"]["rows_drawn_from"] = "the whole 2^20-frame stream" if resident else "the one resident chunk"
        del xs, amps
        torch.cuda.empty_cache()
        res["cpu_baseline"] = cpu_baseline_spectrum(hx, n, "hann")
    return res


def cpu_baseline_spectrum(frames: np.ndarray, n: int, window: str, target_s: float = 6.0):
    """The oracle's fused window + FFT + magnitude + one-sided scaling (oracle_spectrum_batch: spectrum.ts:116-127
    restated, f64 scalar C, plan and window built once and reused, 1 thread) timed on a bounded sample of the
    GPU leg's own frames."""
    import oracle
    plan = oracle.Plan(n)
    win = oracle.create_window(window, n)
    x = np.ascontiguousarray(frames, dtype=np.float64)
    plan.spectrum_batch(x[:8], window=win)  # warm-up
    t0 = time.perf_counter()
    plan.spectrum_batch(x, window=win)
    per = (time.perf_counter() - t0) / x.shape[0]
    reps = max(1, int(target_s / (per * x.shape[0])))
    chk = 0.0
    t0 = time.perf_counter()
    for _ in range(reps):
        amp, _, _ = plan.spectrum_batch(x, window=win)
        chk += float(amp[0, 1])
    sec = time.perf_counter() - t0
    done = reps * x.shape[0]
    return {"value": done * n / sec / 1e9, "unit": "GSample/s", "cores": 1, "kind": "port",
            "frames_per_s": done / sec, "checksum": chk,
            "sample": f"{done} frames of N={n} ({x.shape[0]} distinct frames of the GPU stream x {reps} passes), {sec:.1f} s, "
                      f"f64 scalar C -O2: applyWindow + FFT + magnitude + one-sided scaling, plan and window reused; "
                      f"host has {os.cpu_count()} cpus"}


def also_real4096(args, dev, plan, re, ore, oim):
    """Radix2Fft.forward semantics on the headline's rows (real input, imaginary part taken as zero,
    src/core/fft.ts:77-79): 4 B read + 8 B written = 12 algorithmic bytes per sample, 3 GiB per launch -- reported
    separately, never mixed into the 16 B figure.  64 rows against the oracle."""
    n, steps = plan.size, 10
    batch = re.shape[0]
    stream = torch.cuda.current_stream(dev)
    t_ramp = time.perf_counter()
    plan.forward(re, None, out=(ore, oim))
    torch.cuda.synchronize(dev)
    while time.perf_counter() - t_ramp < args.ramp_seconds:
        for _ in range(10):
            plan.forward(re, None, out=(ore, oim))
        torch.cuda.synchronize(dev)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    evs[0].record(stream)
    for i in range(steps):
        plan.forward(re, None, out=(ore, oim))
        evs[i + 1].record(stream)
    torch.cuda.synchronize(dev)
    ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(steps)]
    avg, nbytes = float(np.mean(ms)), 12 * batch * n
    res = {"config": {"workload": f"N={n} batch={batch} Radix2Fft.forward fp32 real input", "n": n, "batch": batch},
           "kernel": "fft_stockham_kernel<float, 12, LoadReal, StoreComplex>", "steps": steps, "ms": avg, "ms_min": float(np.min(ms)),
           "GSample_per_s": batch * n / (avg * 1e-3) / 1e9, "algorithmic_bytes_per_launch": nbytes,
           "GBps": nbytes / (avg * 1e-3) / 1e9, "frac": nbytes / (avg * 1e-3) / 1e9 / HBM_PEAK_GBPS}
    if not args.no_cpu_baseline:
        g = torch.Generator()
        g.manual_seed(1337)
        sel = torch.randperm(batch, generator=g)[:64].sort().values.to(dev)
        res["parity"] = parity_vs_oracle("real", (re[sel].cpu().numpy(), None),
                                         (ore[sel].cpu().numpy().astype(np.float64), oim[sel].cpu().numpy().astype(np.float64)), n)
    return res


def also_placement(args, dev, plan, re, im, ore, oim, sets: int = 5):
    """How much of the headline's rate on THIS card and in THIS process is the placement of its planes (DESIGN section 5,
    "What the spread of configs[2] is made of"): the same launch, the same inputs, on `sets` more output-plane pairs
    allocated now and kept alive together -- each pair's rate repeats to ~0.4 %, pairs differ by up to 8-14 %.  Index 0
    is the pair the timed region wrote (`value` is never taken from another pair).  Reported, not used."""
    n, batch = plan.size, re.shape[0]
    nbytes = 16 * batch * n
    pairs = [(ore, oim)]
    try:
        for _ in range(sets):
            pairs.append((torch.empty_like(ore), torch.empty_like(oim)))
    except RuntimeError as exc:  # not enough free HBM: report what there is
        if len(pairs) == 1:
            return {"error": f"{type(exc).__name__}: {exc}"[:200]}
    rates = []
    for _ in range(20):
        plan.forward(re, im, out=pairs[0])
    torch.cuda.synchronize(dev)
    for a, b in pairs:
        for _ in range(4):
            plan.forward(re, im, out=(a, b))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            plan.forward(re, im, out=(a, b))
        e1.record()
        torch.cuda.synchronize(dev)
        rates.append(nbytes / (e0.elapsed_time(e1) / 20 * 1e-3) / 1e9)
    plain_all = None
    if len(pairs) > 1:  # four plain allocations, inputs included: what a caller gets without the engine's layout
        try:
            pre, pim = torch.empty_like(re), torch.empty_like(im)
            pre.copy_(re)
            pim.copy_(im)
            a, b = pairs[1]
            for _ in range(4):
                plan.forward(pre, pim, out=(a, b))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                plan.forward(pre, pim, out=(a, b))
            e1.record()
            torch.cuda.synchronize(dev)
            plain_all = nbytes / (e0.elapsed_time(e1) / 20 * 1e-3) / 1e9
            del pre, pim
        except RuntimeError:
            plain_all = None
    plan.forward(re, im, out=(ore, oim))  # the timed region's planes hold the transform again
    torch.cuda.synchronize(dev)
    del pairs
    torch.cuda.empty_cache()
    return {"config": {"workload": f"N={n} batch={batch} forwardComplex fp32, the timed inputs onto {len(rates)} output-plane pairs "
                                   "(index 0 = the timed region's own; the others are plain allocations made now)"},
            "GBps_by_output_pair": rates, "frac_by_output_pair": [r / HBM_PEAK_GBPS for r in rates],
            "GBps_four_plain_allocations": plain_all,
            "frac_four_plain_allocations": plain_all / HBM_PEAK_GBPS if plain_all else None,
            "spread_pct": 100.0 * (max(rates) / min(rates) - 1.0),
            "note": "placement of the planes in HBM, not code: DESIGN section 5; `value` uses pair 0 only"}


def also_fft4096_f64(args, dev, rank: int, re32, im32):
    """The headline shape in the REFERENCE'S OWN precision (src/core/fft.ts:1-14: Float64Array end to end): the same
    65,536 x 4096 rows as doubles through pdsp_fft_forward_complex_f64 -- 32 algorithmic bytes per sample, 8 GiB per
    launch -- HIP events on the launch stream, 64 rows against the f64 oracle at 1e-12 of the row's max."""
    from pragma_dsp_amd.batch import BatchedFft
    n, steps = 4096, 10
    batch = re32.shape[0]
    torch.cuda.empty_cache()
    plan = BatchedFft(n, dev, dtype=torch.float64)
    # plain allocations: with 4-GiB planes the f32 layout measured no better (73.5 / 72.9 vs 77.3 / 78.4 %)
    re, im = re32.double(), im32.double()
    ore, oim = torch.empty_like(re), torch.empty_like(im)
    stream = torch.cuda.current_stream(dev)
    t_ramp = time.perf_counter()
    plan.forward(re, im, out=(ore, oim))
    torch.cuda.synchronize(dev)
    while time.perf_counter() - t_ramp < args.ramp_seconds:
        for _ in range(5):
            plan.forward(re, im, out=(ore, oim))
        torch.cuda.synchronize(dev)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    evs[0].record(stream)
    for i in range(steps):
        plan.forward(re, im, out=(ore, oim))
        evs[i + 1].record(stream)
    torch.cuda.synchronize(dev)
    ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(steps)]
    avg = float(np.mean(ms))
    nbytes = 32 * batch * n
    res = {"config": {"workload": f"N=4096 batch={batch} Radix2Fft.forwardComplex f64 planar complex (configs[2]'s shape in the "
                                  f"reference's own precision)", "n": n, "batch": batch},
           "kernel": "fft_stockham_kernel<double, 12, LoadComplex, StoreComplex>", "dtype": "f64", "steps": steps,
           "ms": avg, "ms_min": float(np.min(ms)), "GSample_per_s": batch * n / (avg * 1e-3) / 1e9,
           "algorithmic_bytes_per_launch": nbytes, "GBps": nbytes / (avg * 1e-3) / 1e9,
           "frac": nbytes / (avg * 1e-3) / 1e9 / HBM_PEAK_GBPS}
    if not args.no_cpu_baseline:
        import oracle
        g = torch.Generator()
        g.manual_seed(1337)
        sel = torch.randperm(batch, generator=g)[:64].sort().values.to(dev)
        wre, wim = oracle.Plan(n).forward_complex(re[sel].cpu().numpy(), im[sel].cpu().numpy())
        want = wre + 1j * wim
        got = ore[sel].cpu().numpy() + 1j * oim[sel].cpu().numpy()
        err = np.abs(got - want).max(axis=1) / np.abs(want).max(axis=1)
        res["parity"] = {"rows": 64, "max_rel_err": float(err.max()), "tolerance": 1e-12, "ok": bool(err.max() <= 1e-12),
                         "against": "oracle/pdsp_oracle.c (f64), rows drawn with seed 1337"}
    del re, im, ore, oim
    torch.cuda.empty_cache()
    return res


if __name__ == "__main__":
    raise SystemExit(main())

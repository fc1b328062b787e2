#!/usr/bin/env python3
"""bench.py -- the headline metric of BASELINE.json on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

A "step" is one pass of the hot path over one batch of synthetic input that is
already resident in HBM: by default BASELINE.json configs[2], the roofline run
(Radix2Fft.forwardComplex semantics, fp32 planar complex, N=4096, batch=65536
per GPU).  For N>1 the driver launches one process per GPU (torch.distributed.run);
the batch is split by rank with no data-path collective (weak scaling: 65,536
transforms per GPU, configs[4] at N=8), and the timed region is bracketed by a
barrier + synchronize with the MAX over ranks taken.  Rank 0 prints ONE JSON line.

Other workloads (parity-checked elsewhere; here for DESIGN.md's numbers):
  --workload spectrum16k   configs[3]: fused Hann+FFT+one-sided amplitude, N=16384,
                           streamed in chunks of --chunk frames
  --workload real4096      Radix2Fft.forward semantics (real in, 12 B/sample)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 measured float4 copy


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


def single_frame_latency(args, dev) -> int:
    """BASELINE configs[1]: ONE N=1024 real frame, Hann window, forward FFT, magnitude.
    Latency-bound (8 KiB of traffic): reported as microseconds, not as a roofline fraction.
      dropin_spectrum_us  spectrum(x, {fftSize:1024, window:'hann'}) host f64 in -> f64 out
                          (f64->f32, H2D, fused kernel, D2H, f32->f64, peak search)
      dropin_forward_us   FFT(1024).forward(x) host f64 in -> f64 out (plan reused)
      kernel_us           the fused kernel alone on device-resident data (HIP events)"""
    import pragma_dsp_amd as pd
    from pragma_dsp_amd.batch import BatchedFft
    n, iters = 1024, max(args.steps, 200)
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
        # the real drop-in (Node + N-API addon) in the shape of the reference's bench/run.ts, with the
        # Node CPU restatement of the same algorithm timed beside it
        import shutil
        import subprocess
        node = shutil.which("node")
        if node:
            try:
                r = subprocess.run([node, os.path.join(ROOT, "tests", "js", "bench_latency.js"), "2000"],
                                   capture_output=True, text=True, timeout=300)
                js = json.loads(r.stdout) if r.returncode == 0 else {"error": r.stderr[-400:]}
            except Exception as e:
                js = {"error": repr(e)}
    print(json.dumps({
        "js_dropin_latency": js,
        "metric": "single-frame latency (N=1024, Hann + FFT + magnitude)", "unit": "us", "higher_is_better": False,
        "value": spec_med, "n_gpus": 1, "steps": iters, "warmup": 20, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "N=1024 single frame spectrum(), hann, one-sided (configs[1])"},
        "dropin_spectrum_us": {"median": spec_med, "min": spec_min},
        "dropin_forward_us": {"median": fwd_med, "min": fwd_min},
        "kernel_back_to_back_us": kernel_us,
    }), flush=True)
    return 0


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


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="fft4096", choices=["fft4096", "real4096", "fft16k", "spectrum16k", "spectrum256", "peaks16k", "single1024", "stream"])
    ap.add_argument("--batch", type=int, default=None, help="transforms per GPU (default: the config's)")
    ap.add_argument("--n", type=int, default=None, help="spectrum256 only: another frame size (development sweeps)")
    ap.add_argument("--chunk", type=int, default=16384, help="frames per launch for spectrum16k")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--ramp-seconds", type=float, default=0.6, help="untimed clock-ramp before the warm-up steps")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default) or gloo (1-GPU rehearsal)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--gather", action="store_true", help="also time the RCCL all-gather of the output slabs")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.share_gpu:
        local = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import torch.distributed as dist
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.dist_backend)
    if not torch.cuda.is_available():
        print("bench.py needs a GPU (the pdsp engine has no CPU fallback)", file=sys.stderr)
        return 2
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from pragma_dsp_amd.batch import BatchedFft
    from pragma_dsp_amd.shard import gather_rows, max_over_ranks, my_rows

    if args.workload == "single1024":
        return single_frame_latency(args, dev)
    if args.workload == "stream":
        return stream_throughput(args, dev)

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
    plan = BatchedFft(n, dev)
    stream = torch.cuda.current_stream(dev)

    if args.workload in ("fft4096", "fft16k"):
        re, im = synth_batch(per_gpu, n, dev, seed=1337 + rank)
        ore, oim = torch.empty_like(re), torch.empty_like(im)
        launches_per_step = 1
        bytes_per_launch = 16 * per_gpu * n  # 8 B read + 8 B written per sample
        kernel_name, kernel_label = "fft_stockham_kernel<float, 12, pdsp::LoadComplex", \
            "fft_stockham_kernel<float, 12, LoadComplex, StoreComplex>"
        if args.workload == "fft16k":
            kernel_name, kernel_label = "fft_split4_kernel<float, 12, pdsp::LoadComplex", \
                "fft_split4_kernel<float, 12, LoadComplex, StoreComplex>"

        def step():
            plan.forward(re, im, out=(ore, oim))
    elif args.workload == "real4096":
        re, _ = synth_batch(per_gpu, n, dev, seed=1337 + rank, complex_noise=False)
        im = None
        ore, oim = torch.empty_like(re), torch.empty_like(re)
        launches_per_step = 1
        bytes_per_launch = 12 * per_gpu * n
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
        kernel_name = kernel_label = "spectrum_split16k_kernel<float, true, true>"
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
        kernel_name = kernel_label = "spectrum_split16k_kernel<float, true, false>"
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
            if args.dist_backend == "nccl":
                dist.barrier(device_ids=[local])
            else:
                dist.barrier()
        torch.cuda.synchronize(dev)

    # Untimed clock ramp: a cold MI355X needs ~0.5 s of work before its clocks settle
    # (first 20 launches measured 8 % slower than steady state); then the W warm-up steps.
    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < args.ramp_seconds:
        for _ in range(10):
            step()
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
    elapsed = time.perf_counter() - t0
    barrier()
    step_ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps)]

    elapsed = max_over_ranks(elapsed, dev)

    # context for the roofline fraction (outside the timed region): the rate at which this box, with
    # these very buffers, copies the input planes to the output planes (torch's device copy kernel)
    copy_gbps = None
    if args.workload in ("fft4096", "fft16k") and rank == 0:
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

    gather = None
    if args.gather and world > 1 and args.workload in ("fft4096", "real4096"):
        # the one exchange step of the path: RCCL all-gather of the output slabs,
        # timed on its own -- at 2 GiB/rank it is xGMI-per-link bound and dwarfs the compute
        gather_rows(ore, per_gpu * world)  # warm-up (communicator setup)
        barrier()
        g0 = time.perf_counter()
        full_re = gather_rows(ore, per_gpu * world)
        full_im = gather_rows(oim, per_gpu * world)
        torch.cuda.synchronize(dev)
        gsec = max_over_ranks(time.perf_counter() - g0, dev)
        nbytes = (ore.numel() + oim.numel()) * 4
        gather = {"ms": gsec * 1e3, "bytes_per_rank": nbytes,
                  "GBps_in_per_gpu": nbytes * (world - 1) / gsec / 1e9,
                  "rows_gathered": int(full_re.shape[0])}
        del full_re, full_im

    if rank == 0:
        samples_per_step = per_gpu * n * world
        value = samples_per_step * args.steps / elapsed / 1e9
        launch_ms = float(np.mean(step_ms)) / launches_per_step
        achieved = bytes_per_launch / (launch_ms * 1e-3) / 1e9
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
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": {"fft4096": f"N=4096 batch={per_gpu}/GPU Radix2Fft.forwardComplex fp32 planar complex (configs[2])",
                                    "real4096": f"N=4096 batch={per_gpu}/GPU Radix2Fft.forward fp32 real input",
                                    "fft16k": f"N=16384 batch={per_gpu}/GPU Radix2Fft.forwardComplex fp32 planar complex",
                                    "spectrum256": f"N={n} batch={per_gpu}/GPU fused hann+FFT+one-sided amplitude",
                                    "spectrum16k": f"N=16384 batch={per_gpu}/GPU fused hann+FFT+one-sided amplitude (configs[3])",
                                    "peaks16k": f"N=16384 batch={per_gpu}/GPU fused hann+FFT+findPeak, peaks-only output"}[args.workload],
                       "n": n, "batch_per_gpu": per_gpu, "global_batch": per_gpu * world,
                       "parallelism": f"batch-shard x{world}", "launches_per_step": launches_per_step},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS,
                         # PMC bytes of the committed rocprofv3 passes: only valid for the profiled shape
                         "traffic": traffic_from_profile(kernel_name) if args.batch is None and args.chunk == 16384 else None,
                         "kernel": kernel_label,
                         "device_copy_same_buffers_GBps": copy_gbps,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "launch_ms_avg": launch_ms, "launch_ms_min": float(np.min(step_ms)) / launches_per_step},
        }
        if gather:
            out["gather"] = gather
        if world == 1 and not args.no_cpu_baseline:
            rows = 2048
            sel = torch.cat([torch.arange(0, rows // 2), torch.arange(per_gpu // 2, per_gpu // 2 + rows // 2)]) \
                if args.workload in ("fft4096", "real4096") else torch.arange(0, min(rows, re.shape[0]))
            hre = re[sel.to(dev)].cpu().numpy().astype(np.float64)
            him = im[sel.to(dev)].cpu().numpy().astype(np.float64) if im is not None else None
            out["cpu_baseline"] = cpu_baseline(hre, him, n, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())

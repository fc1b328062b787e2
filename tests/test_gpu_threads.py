"""The boundary is re-entrant: host threads may call the drop-in concurrently -- on ONE shared plan
(its stream and staging are guarded by the plan's mutex), on plans of their own, through the cached one-shot spectrum()
(process-wide plan cache) -- and every result must equal the single-threaded one bit for bit.  ctypes releases the GIL
inside the C call, so the calls really overlap."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_concurrent_host_calls_give_the_single_threaded_results(pdsp):
    rng = np.random.default_rng(99)
    sizes = [1024, 4096, 8192]
    shared = {n: pdsp.FFT(n) for n in sizes}
    inputs = {(n, i): rng.standard_normal(n) for n in sizes for i in range(6)}
    want_fwd = {k: shared[k[0]].forward(x) for k, x in inputs.items()}
    opts = {"sampleRate": 48000, "window": "hann"}
    want_spec = {k: pdsp.spectrum(x, {**opts, "fftSize": k[0]}) for k, x in inputs.items()}
    errors = []

    def worker(tid):
        try:
            own = {n: pdsp.FFT(n) for n in sizes} if tid % 2 else shared  # odd threads: plans of their own
            for rep in range(40):
                for (n, i), x in inputs.items():
                    if (i + rep + tid) % 3 == 0:
                        got = own[n].forward(x)
                        w = want_fwd[(n, i)]
                        if not (np.array_equal(got.real, w.real) and np.array_equal(got.imag, w.imag)):
                            errors.append(("forward", tid, n, i))
                    elif (i + rep + tid) % 3 == 1:
                        got = pdsp.spectrum(x, {**opts, "fftSize": n})
                        w = want_spec[(n, i)]
                        if not (np.array_equal(got.amplitude, w.amplitude) and got.peak.index == w.peak.index
                                and np.array_equal(got.phase, w.phase)):
                            errors.append(("spectrum", tid, n, i))
                    else:
                        back = own[n].inverse(want_fwd[(n, i)])
                        if np.abs(back.real - x).max() > 1e-10:
                            errors.append(("inverse", tid, n, i))
        except Exception as exc:  # noqa: BLE001
            errors.append(("exception", tid, repr(exc)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in threads), "a worker is stuck"
    assert not errors, errors[:5]


def test_concurrent_device_calls_on_separate_streams(oracle_mod):
    """The device-pointer family only enqueues on the caller's stream: four threads, each with its own torch stream and
    its own buffers, share ONE plan (its tables are immutable) and must all get the oracle's rows."""
    import torch
    from pragma_dsp_amd.batch import BatchedFft
    from conftest import rel_err
    n, batch = 4096, 64
    plan = BatchedFft(n, "cuda:0")
    rng = np.random.default_rng(5)
    data = [(rng.standard_normal((batch, n)).astype(np.float32), rng.standard_normal((batch, n)).astype(np.float32)) for _ in range(4)]
    want = [np.add(*[a if j == 0 else 1j * a for j, a in enumerate(oracle_mod.Plan(n).forward_complex(re, im))]) for re, im in data]
    out, errors = [None] * 4, []

    def worker(t):
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                re, im = torch.from_numpy(data[t][0]).cuda(), torch.from_numpy(data[t][1]).cuda()
                for _ in range(50):
                    ore, oim = plan.forward(re, im)
                s.synchronize()
                out[t] = ore.cpu().numpy().astype(np.float64) + 1j * oim.cpu().numpy()
        except Exception as exc:  # noqa: BLE001
            errors.append(repr(exc))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    for t in range(4):
        assert rel_err(out[t], want[t]) <= 1e-5


def test_plan_churn_does_not_leak_device_or_host_memory(pdsp):
    """The JS drop-in frees plans from an N-API finalizer, i.e. at the garbage collector's whim: create / use / destroy
    must hand everything back (tables, stream, pinned and device staging).  2,000 cycles over several sizes, device free
    memory and the process's resident set compared before and after."""
    import gc
    import resource
    import torch
    sizes = [256, 4096, 16384]
    x = {n: np.random.default_rng(n).standard_normal(n) for n in sizes}

    def cycle(count):
        for i in range(count):
            n = sizes[i % len(sizes)]
            fft = pdsp.FFT(n)
            fft.forward(x[n])
            del fft
    cycle(60)  # warm the allocators
    gc.collect()
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    cycle(2000)
    gc.collect()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    rss1 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    assert free0 - free1 < (64 << 20), f"device memory went down by {(free0 - free1) >> 20} MiB over 2000 plan cycles"
    assert rss1 - rss0 < 200 * 1024, f"resident set grew by {(rss1 - rss0) // 1024} MiB over 2000 plan cycles"  # ru_maxrss in KiB

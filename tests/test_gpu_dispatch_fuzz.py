"""Seeded fuzz over the dispatcher: every transform / spectrum call picks one of several kernels
from size, batch, alignment, frame length, stride, window, sides and requested outputs (staged,
direct, split4, split16k, packed fast/general, complex fallback, four-step).  Each draw is checked
against the f64 oracle, so a wrong dispatch condition or a tail bug in any variant shows up here."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _offset_view(torch, rows, n, off, stride=None):
    """[rows, n] view whose first element is `off` floats past a 256-byte aligned base."""
    stride = n if stride is None else stride
    flat = torch.zeros(rows * stride + off + 8, device="cuda")
    return torch.as_strided(flat, (rows, n), (stride, 1), off)


@pytest.mark.parametrize("seed", range(6))
def test_transform_dispatch_fuzz(oracle_mod, seed):
    import torch
    from pragma_dsp_amd.batch import BatchedFft
    rng = np.random.default_rng(1000 + seed)
    plans = {}
    for _ in range(40):
        log2n = int(rng.integers(0, 16))
        n = 1 << log2n
        batch = int(rng.integers(1, max(2, min(300, (1 << 21) // n))))
        off = int(rng.choice([0, 0, 1, 2, 4]))
        kind = rng.choice(["complex", "real", "inverse"])
        re = rng.standard_normal((batch, n)).astype(np.float32)
        im = rng.standard_normal((batch, n)).astype(np.float32)
        plan = plans.setdefault(n, BatchedFft(n, "cuda:0"))
        dre = _offset_view(torch, batch, n, off)
        dim = _offset_view(torch, batch, n, off)
        dre.copy_(torch.from_numpy(re))
        dim.copy_(torch.from_numpy(im))
        if not dre.is_contiguous():   # n == 1 rows are contiguous anyway
            continue
        o = oracle_mod.Plan(n)
        if kind == "complex":
            gre, gim = plan.forward(dre, dim)
            wre, wim = o.forward_complex(re, im)
        elif kind == "real":
            gre, gim = plan.forward(dre)
            wre, wim = o.forward(re)
        else:
            gre, gim = plan.inverse(dre, dim)
            wre, wim = o.inverse(re, im)
        got = gre.cpu().numpy().astype(np.float64) + 1j * gim.cpu().numpy()
        assert rel_err(got, wre + 1j * wim) <= TOL, (log2n, batch, off, kind)


@pytest.mark.parametrize("seed", range(6))
def test_spectrum_dispatch_fuzz(oracle_mod, seed):
    import torch
    from pragma_dsp_amd.batch import BatchedFft
    rng = np.random.default_rng(2000 + seed)
    plans = {}
    for _ in range(30):
        log2n = int(rng.integers(0, 17))
        n = 1 << log2n
        batch = int(rng.integers(1, max(2, min(200, (1 << 20) // n))))
        length = int(rng.choice([n, n, n, max(1, n - int(rng.integers(0, n))), n + int(rng.integers(1, 9))]))
        off = int(rng.choice([0, 0, 1, 2, 4]))
        window = str(rng.choice(["rect", "hann", "hamming", "blackman"]))
        if n <= 2 and window in ("hann", "blackman"):
            window = "hamming"  # hann(2) = [0, 0], blackman(2) ~ 1e-17: nothing but f32 underflow left to compare
        sides = str(rng.choice(["one", "one", "two"]))
        want_phase, want_peak = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        x = rng.standard_normal((batch, length)).astype(np.float32)
        x[0] *= 0  # a zero frame in every batch: exact zeros, peak 0
        plan = plans.setdefault(n, BatchedFft(n, "cuda:0"))
        dx = _offset_view(torch, batch, length, off)
        dx.copy_(torch.from_numpy(x))
        amp, ph, pk = plan.spectrum(dx, window, sides, want_phase=want_phase, want_peak=want_peak)
        frame = np.zeros((batch, n), dtype=np.float32)
        frame[:, :min(n, length)] = x[:, :n]
        win = oracle_mod.create_window(window, n).astype(np.float32) if (window != "rect" and n > 1) else None
        wamp, wph, wpk = oracle_mod.Plan(n).spectrum_batch(frame, window=win, two_sided=(sides == "two"),
                                                           want_phase=True, want_peak=True)
        ctx = (log2n, batch, length, off, window, sides, want_phase, want_peak)
        a = amp.cpu().numpy().astype(np.float64)
        assert a.shape == wamp.shape and rel_err(a, wamp) <= TOL, ctx
        assert not a[0].any(), ctx
        if want_phase:
            mask = wamp > 1e-2 * wamp.max(axis=-1, keepdims=True)
            d = np.abs((ph.cpu().numpy() - wph + np.pi) % (2 * np.pi) - np.pi)
            assert d[mask].max(initial=0) <= 5e-3, ctx
        if want_peak:
            p = pk.cpu().numpy()
            assert p[0] == 0, ctx
            for b in range(1, batch):   # same bin, or a bin whose amplitude ties within tolerance
                assert abs(wamp[b, p[b]] - wamp[b, wpk[b]]) <= 2 * TOL * wamp[b].max(), ctx


@pytest.mark.parametrize("seed", range(3))
def test_host_dropin_spectrum_fuzz(pdsp, oracle_mod, seed):
    """spectrum(samples, options) through the host drop-in (default f64 mode) with random lengths,
    fftSize present or absent, windows, sides and sample rates: every field against the oracle."""
    rng = np.random.default_rng(3000 + seed)
    prev = pdsp.lib.pdsp_set_host_precision(64)
    try:
        for _ in range(25):
            length = int(rng.choice([1, 2, 3, 5, 8, 100, 1000, 1024, 4097, int(rng.integers(1, 40000))]))
            x = rng.standard_normal(length) + 0.5 * np.sin(2 * np.pi * 0.123 * np.arange(length))
            opts = {"sampleRate": float(rng.choice([1.0, 8000.0, 44100.0, 48000.0]))}
            if rng.integers(0, 2):
                opts["fftSize"] = 1 << int(rng.integers(0, 16))
            opts["window"] = str(rng.choice(["rect", "hann", "hamming", "blackman"]))
            opts["sides"] = str(rng.choice(["one", "two"]))
            g = pdsp.spectrum(x, opts)
            w = oracle_mod.spectrum(x, sample_rate=opts["sampleRate"], fft_size=opts.get("fftSize"),
                                    window=opts["window"], sides=opts["sides"])
            ctx = (length, opts)
            assert np.array_equal(g.frequencies, w["frequencies"]), ctx
            scale = max(np.abs(w["amplitude"]).max(), 1e-300)
            assert np.abs(g.amplitude - w["amplitude"]).max() <= 1e-12 * max(1.0, scale), ctx
            n = len(w["frequencies"]) if opts["sides"] == "two" else 2 * (len(w["frequencies"]) - 1)
            if n <= 2 and opts["window"] in ("hann", "blackman"):
                continue  # an all-(near-)zero window: peak and phase are decided by rounding noise
            # peak: same bin, or a bin whose amplitude ties to rounding (two-sided mirror)
            assert abs(w["amplitude"][g.peak.index] - w["peak"]["amplitude"]) <= 1e-12 * max(1.0, scale), ctx
            assert g.peak.frequency == g.frequencies[g.peak.index] and g.peak.amplitude == g.amplitude[g.peak.index], ctx
            assert g.peak.phase == g.phase[g.peak.index], ctx
            mask = w["amplitude"] > 1e-6 * scale
            d = np.abs((g.phase - w["phase"] + np.pi) % (2 * np.pi) - np.pi)
            assert d[mask].max(initial=0) <= 1e-8, ctx
    finally:
        pdsp.lib.pdsp_set_host_precision(prev)


@pytest.mark.parametrize("seed", range(3))
def test_fused_peaks_fuzz(oracle_mod, seed):
    """pdsp_spectrum_peaks_f32 (findPeak fused into the kernel, or the row fallback) on random
    sizes / lengths / alignments; frames carry one dominant tone so the expected bin is unambiguous."""
    import torch
    from pragma_dsp_amd.batch import BatchedFft
    rng = np.random.default_rng(4000 + seed)
    plans = {}
    for _ in range(25):
        log2n = int(rng.integers(3, 17))
        n = 1 << log2n
        batch = int(rng.integers(1, max(2, min(100, (1 << 20) // n))))
        length = int(rng.choice([n, n, max(n // 2 + 1, n - int(rng.integers(0, n // 4 + 1)))]))
        off = int(rng.choice([0, 0, 1, 4]))
        window = str(rng.choice(["rect", "hann", "blackman"]))
        sides = str(rng.choice(["one", "one", "two"]))
        with_rows = bool(rng.integers(0, 2))
        k = rng.integers(1, n // 2, size=batch)
        t = np.arange(length)
        x = (np.sin(2 * np.pi * k[:, None] * t[None, :] / n + 0.3) + 0.01 * rng.standard_normal((batch, length))).astype(np.float32)
        plan = plans.setdefault(n, BatchedFft(n, "cuda:0"))
        dx = _offset_view(torch, batch, length, off)
        dx.copy_(torch.from_numpy(x))
        idx, freq, pamp, pph, amp, ph = plan.spectrum_peaks(dx, window, sides, 48000.0, want_amp=with_rows,
                                                            want_phase=with_rows)
        frame = np.zeros((batch, n), dtype=np.float32)
        frame[:, :length] = x
        win = oracle_mod.create_window(window, n).astype(np.float32) if window != "rect" else None
        wamp, wph, wpk = oracle_mod.Plan(n).spectrum_batch(frame, window=win, two_sided=(sides == "two"),
                                                           want_phase=True, want_peak=True)
        ctx = (log2n, batch, length, off, window, sides, with_rows)
        gi = idx.cpu().numpy()
        for b in range(batch):
            assert gi[b] in (wpk[b], (n - wpk[b]) if sides == "two" else wpk[b]) or \
                abs(wamp[b, gi[b]] - wamp[b, wpk[b]]) <= 2 * TOL * wamp[b].max(), ctx
        rows = np.arange(batch)
        assert np.abs(pamp.cpu().numpy() - wamp[rows, gi]).max() <= TOL * wamp.max(), ctx
        assert np.abs(freq.cpu().numpy() - gi * 48000.0 / n).max() <= 48000.0 * 1e-6, ctx
        d = np.abs((pph.cpu().numpy() - wph[rows, gi] + np.pi) % (2 * np.pi) - np.pi)
        assert d.max() <= 5e-3, ctx
        if with_rows:
            assert rel_err(amp.cpu().numpy(), wamp) <= TOL, ctx

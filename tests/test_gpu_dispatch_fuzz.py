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

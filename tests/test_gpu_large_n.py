"""N beyond the single-pass LDS limit (the reference takes any power of two): the four-step path,
f32 up to 2^18 and f64 up to 2^17, vs the f64 oracle."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("log2n", [15, 16, 17, 18])
def test_large_complex_real_inverse_f32(oracle_mod, log2n):
    import torch
    from pragma_dsp_amd.batch import BatchedFft
    n = 1 << log2n
    rng = np.random.default_rng(log2n)
    re = rng.standard_normal((3, n)).astype(np.float32)
    im = rng.standard_normal((3, n)).astype(np.float32)
    plan = BatchedFft(n, "cuda:0")
    dre, dim = torch.from_numpy(re).cuda(), torch.from_numpy(im).cuda()
    ore, oim = plan.forward(dre, dim)
    wre, wim = oracle_mod.Plan(n).forward_complex(re, im)
    assert rel_err(ore.cpu().numpy().astype(np.float64) + 1j * oim.cpu().numpy(), wre + 1j * wim) <= 1e-5
    bre, bim = plan.inverse(ore, oim)
    assert rel_err(bre.cpu().numpy(), re) <= 1e-5 and rel_err(bim.cpu().numpy(), im) <= 1e-5
    rre, rim = plan.forward(dre)
    wre, wim = oracle_mod.Plan(n).forward(re)
    assert rel_err(rre.cpu().numpy().astype(np.float64) + 1j * rim.cpu().numpy(), wre + 1j * wim) <= 1e-5


@pytest.mark.parametrize("log2n", [14, 15, 17])
def test_large_f64(oracle_mod, log2n):
    import torch
    from pragma_dsp_amd.batch import BatchedFft
    n = 1 << log2n
    rng = np.random.default_rng(log2n)
    re, im = rng.standard_normal((2, n)), rng.standard_normal((2, n))
    plan = BatchedFft(n, "cuda:0", dtype=torch.float64)
    ore, oim = plan.forward(torch.from_numpy(re).cuda(), torch.from_numpy(im).cuda())
    wre, wim = oracle_mod.Plan(n).forward_complex(re, im)
    assert rel_err(ore.cpu().numpy() + 1j * oim.cpu().numpy(), wre + 1j * wim) <= 1e-13
    bre, _ = plan.inverse(ore, oim)
    assert rel_err(bre.cpu().numpy(), re) <= 1e-13


def test_large_spectrum_device_and_dropin(pdsp, oracle_mod):
    import torch
    from pragma_dsp_amd.batch import BatchedFft
    n = 1 << 16
    rng = np.random.default_rng(3)
    t = np.arange(n)
    x = (rng.standard_normal((2, n)) * 0.1 + np.sin(2 * np.pi * 1234 * t / n)).astype(np.float32)
    plan = BatchedFft(n, "cuda:0")
    for sides in ("one", "two"):
        amp, ph, pk = plan.spectrum(torch.from_numpy(x).cuda(), "hann", sides, want_phase=True, want_peak=True)
        wamp, wph, wpk = oracle_mod.Plan(n).spectrum_batch(x, window=oracle_mod.create_window("hann", n).astype(np.float32),
                                                           two_sided=(sides == "two"), want_phase=True, want_peak=True)
        assert amp.shape == wamp.shape and rel_err(amp.cpu().numpy(), wamp) <= 1e-5
        assert all(int(v) in (1234, n - 1234 if sides == "two" else 1234) for v in pk.cpu())  # mirror tie (H2)
    idx, freq, a, p, _, _ = plan.spectrum_peaks(torch.from_numpy(x).cuda(), "hann", "one", 48000.0)
    assert [int(v) for v in idx.cpu()] == [1234, 1234] and abs(float(freq[0]) - 1234 * 48000.0 / n) < 1e-2
    # the drop-in on a long signal: 100,000 samples -> N = 131072 (f64 four-step), zero-padded
    sig = np.sin(2 * np.pi * 440.0 * np.arange(100000) / 48000.0)
    g = pdsp.spectrum(sig, {"sampleRate": 48000, "window": "hann"})
    w = oracle_mod.spectrum(sig, sample_rate=48000, window="hann")
    assert len(g.amplitude) == 65537 and g.peak.index == w["peak"]["index"]
    assert np.abs(g.amplitude - w["amplitude"]).max() <= 1e-12
    big = pdsp.Radix2Fft(1 << 18)      # f32 beyond 2^17
    out = big.forward(np.cos(2 * np.pi * 5 * np.arange(1 << 18) / (1 << 18)))
    assert abs(out.real[5] - (1 << 17)) < 1 and abs(out.real[(1 << 18) - 5] - (1 << 17)) < 1

"""First GPU parity gate: every size 1..16384, real / complex / inverse, ragged
batches, through the C ABI, against the f64 oracle."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu

TOL = 1e-5  # north-star: max|err|/max|X| <= 1e-5 (fp32 device path vs f64 reference)


@pytest.mark.parametrize("log2n", list(range(0, 15)))
def test_forward_complex_all_sizes(oracle_mod, log2n):
    import torch
    from pragma_dsp_amd.batch import BatchedFft
    n = 1 << log2n
    rng = np.random.default_rng(100 + log2n)
    batch = 37 if n <= 4096 else 5  # not a multiple of the rows-per-workgroup
    re = rng.standard_normal((batch, n)).astype(np.float32)
    im = rng.standard_normal((batch, n)).astype(np.float32)
    plan = BatchedFft(n, "cuda:0")
    ore, oim = plan.forward(torch.from_numpy(re).cuda(), torch.from_numpy(im).cuda())
    wre, wim = oracle_mod.Plan(n).forward_complex(re, im)
    got = ore.cpu().numpy().astype(np.float64) + 1j * oim.cpu().numpy()
    assert rel_err(got, wre + 1j * wim) <= TOL


@pytest.mark.parametrize("log2n", [0, 1, 3, 4, 5, 8, 10, 12, 14])
def test_forward_real_and_inverse(oracle_mod, log2n):
    import torch
    from pragma_dsp_amd.batch import BatchedFft
    n = 1 << log2n
    rng = np.random.default_rng(200 + log2n)
    batch = 9
    x = rng.standard_normal((batch, n)).astype(np.float32)
    plan = BatchedFft(n, "cuda:0")
    dx = torch.from_numpy(x).cuda()
    ore, oim = plan.forward(dx)
    wre, wim = oracle_mod.Plan(n).forward(x)
    got = ore.cpu().numpy().astype(np.float64) + 1j * oim.cpu().numpy()
    assert rel_err(got, wre + 1j * wim) <= TOL
    bre, bim = plan.inverse(ore, oim)
    ire, iim = oracle_mod.Plan(n).inverse(ore.cpu().numpy(), oim.cpu().numpy())
    back = bre.cpu().numpy().astype(np.float64) + 1j * bim.cpu().numpy()
    assert rel_err(back, ire + 1j * iim) <= TOL
    assert rel_err(bre.cpu().numpy(), x) <= TOL  # round trip


def test_reallife_goldens_through_dropin(pdsp, reallife, manifest):
    fft = pdsp.FFT(1024)
    for c in manifest["reallife"]:
        name = c["name"]
        out = fft.forward(reallife[name + "/signal"])
        want = reallife[name + "/fftRe"] + 1j * reallife[name + "/fftIm"]
        if np.abs(want).max() == 0:
            assert np.all(out.real == 0) and np.all(out.imag == 0)
            continue
        assert rel_err(out.real + 1j * out.imag, want) <= TOL, name


def test_spectrum_dropin(pdsp, oracle_mod):
    got = pdsp.spectrum([0, 1, 0, -1, 0, 1, 0, -1], {"sampleRate": 48000})
    assert got.peak.index == 2 and got.peak.frequency == 12000
    assert abs(got.peak.amplitude - 1) < 1e-6 and abs(got.peak.phase + np.pi / 2) < 1e-6
    x = np.random.default_rng(5).standard_normal(1000)
    for window in ("rect", "hann", "hamming", "blackman"):
        for sides in ("one", "two"):
            g = pdsp.spectrum(x, {"sampleRate": 48000, "window": window, "sides": sides})
            w = oracle_mod.spectrum(x, sample_rate=48000, window=window, sides=sides)
            assert len(g.amplitude) == len(w["amplitude"])
            assert rel_err(g.amplitude, w["amplitude"]) <= TOL
            assert np.array_equal(g.frequencies, w["frequencies"])
            # real input: bins k and N-k tie mathematically in two-sided mode and the
            # strict '>' picks whichever rounding favours -- accept the mirror
            n = len(g.frequencies) if sides == "two" else 0
            assert g.peak.index in (w["peak"]["index"], (n - w["peak"]["index"]) % max(n, 1))
            assert abs(g.peak.amplitude - w["peak"]["amplitude"]) <= TOL * w["amplitude"].max()

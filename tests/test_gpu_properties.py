"""Size-independent properties at BASELINE.json's full size (N=4096 x 65,536 rows) and
batch-shape edge cases, through the C ABI.  The oracle checks a 256-row random subset."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def full():
    import torch
    from pragma_dsp_amd.batch import BatchedFft
    n, b = 4096, 65536
    g = torch.Generator(device="cuda")
    g.manual_seed(1337)
    re = torch.randn((b, n), generator=g, device="cuda", dtype=torch.float32)
    im = torch.randn((b, n), generator=g, device="cuda", dtype=torch.float32)
    plan = BatchedFft(n, "cuda:0")
    ore, oim = plan.forward(re, im)
    torch.cuda.synchronize()
    return plan, re, im, ore, oim


def test_full_size_subset_vs_oracle(full, oracle_mod):
    import torch
    plan, re, im, ore, oim = full
    rows = torch.from_numpy(np.random.default_rng(3).choice(65536, 256, replace=False)).cuda()
    wre, wim = oracle_mod.Plan(4096).forward_complex(re[rows].cpu().numpy(), im[rows].cpu().numpy())
    got = ore[rows].cpu().numpy().astype(np.float64) + 1j * oim[rows].cpu().numpy()
    assert rel_err(got, wre + 1j * wim) <= TOL


def test_full_size_parseval_and_round_trip(full):
    import torch
    plan, re, im, ore, oim = full
    n = 4096
    e_in = (re.double() ** 2 + im.double() ** 2).sum(dim=1)
    e_out = (ore.double() ** 2 + oim.double() ** 2).sum(dim=1) / n
    assert float(((e_out - e_in).abs() / e_in).max()) < 1e-5
    bre, bim = plan.inverse(ore, oim)
    scale = torch.maximum(re.abs().amax(dim=1), im.abs().amax(dim=1))
    assert float(((bre - re).abs().amax(dim=1) / scale).max()) <= TOL
    assert float(((bim - im).abs().amax(dim=1) / scale).max()) <= TOL


def test_full_size_linearity_and_shift(full):
    import torch
    plan, re, im, ore, oim = full
    half = 32768
    a, b = 0.75, -1.25  # F(a x + b y) = a F(x) + b F(y)
    sre, sim = plan.forward(a * re[:half] + b * re[half:], a * im[:half] + b * im[half:])
    wre, wim = a * ore[:half] + b * ore[half:], a * oim[:half] + b * oim[half:]
    scale = torch.maximum(wre.abs().amax(dim=1), wim.abs().amax(dim=1))
    assert float(((sre - wre).abs().amax(dim=1) / scale).max()) <= TOL
    assert float(((sim - wim).abs().amax(dim=1) / scale).max()) <= TOL
    # a circular shift by one sample multiplies bin k by e^{-2 pi i k / N}
    rows = slice(0, 512)
    rre, rim = plan.forward(torch.roll(re[rows], 1, dims=1), torch.roll(im[rows], 1, dims=1))
    k = torch.arange(4096, device="cuda", dtype=torch.float64)
    c, s = torch.cos(2 * np.pi * k / 4096), -torch.sin(2 * np.pi * k / 4096)
    ere = ore[rows].double() * c - oim[rows].double() * s
    eim = ore[rows].double() * s + oim[rows].double() * c
    sc = torch.maximum(ere.abs().amax(dim=1), eim.abs().amax(dim=1))
    assert float(((rre.double() - ere).abs().amax(dim=1) / sc).max()) <= TOL
    assert float(((rim.double() - eim).abs().amax(dim=1) / sc).max()) <= TOL


def test_row_wise_in_place(full):
    """Each workgroup loads its whole row before it stores, so out may alias in."""
    import torch
    plan, re, im, ore, oim = full
    a, b = re[:1000].clone(), im[:1000].clone()
    plan.forward(a, b, out=(a, b))
    assert torch.equal(a, ore[:1000]) and torch.equal(b, oim[:1000])  # bit-identical to out-of-place


@pytest.mark.parametrize("n", [8, 64, 256, 1024, 2048])
@pytest.mark.parametrize("batch", [1, 2, 3, 5, 255, 257])
def test_ragged_batches(oracle_mod, n, batch):
    """Batches that do not fill the last workgroup (several rows share a workgroup for N < 4096)."""
    import torch
    from pragma_dsp_amd.batch import BatchedFft
    rng = np.random.default_rng(n * 1000 + batch)
    re = rng.standard_normal((batch, n)).astype(np.float32)
    im = rng.standard_normal((batch, n)).astype(np.float32)
    plan = BatchedFft(n, "cuda:0")
    guard = torch.full((batch + 2, n), 777.0, device="cuda")  # canary rows around the output
    ore = guard[1:batch + 1]
    oim = torch.empty((batch, n), device="cuda")
    plan.forward(torch.from_numpy(re).cuda(), torch.from_numpy(im).cuda(), out=(ore, oim))
    wre, wim = oracle_mod.Plan(n).forward_complex(re, im)
    assert rel_err(ore.cpu().numpy().astype(np.float64) + 1j * oim.cpu().numpy(), wre + 1j * wim) <= TOL
    assert bool((guard[0] == 777.0).all()) and bool((guard[-1] == 777.0).all())  # nothing written out of range
    amp, _, pk = plan.spectrum(torch.from_numpy(re).cuda(), "hann", "one", want_peak=True)
    wamp, _, wpk = oracle_mod.Plan(n).spectrum_batch(re, window=oracle_mod.create_window("hann", n).astype(np.float32),
                                                     want_peak=True)
    assert rel_err(amp.cpu().numpy(), wamp) <= TOL


def test_special_inputs_exact(oracle_mod):
    """zeros -> exact zeros; impulse -> flat |X| = 1; DC -> only bin 0 (edge_cases / signals tests)."""
    import torch
    from pragma_dsp_amd.batch import BatchedFft
    for n in (64, 1024, 4096, 16384):
        plan = BatchedFft(n, "cuda:0")
        x = torch.zeros((3, n), device="cuda")
        x[1, 0] = 1.0
        x[2, :] = 1.0
        ore, oim = plan.forward(x)
        assert not bool(ore[0].any()) and not bool(oim[0].any())
        assert float((torch.sqrt(ore[1] ** 2 + oim[1] ** 2) - 1).abs().max()) < 1e-6
        assert abs(float(ore[2, 0]) - n) < 1e-3 * n and float(ore[2, 1:].abs().max()) == 0.0 and not bool(oim[2].any())


def test_error_paths_on_device(pdsp):
    import torch
    from pragma_dsp_amd.batch import BatchedFft
    plan = BatchedFft(64, "cuda:0")
    with pytest.raises(pdsp.PdspError, match="FFT input length 32 != size 64"):
        plan.forward(torch.zeros((2, 32), device="cuda"))
    with pytest.raises(pdsp.PdspError, match="Window length must match input length."):
        plan.spectrum(torch.zeros((2, 64), device="cuda"), torch.ones(32, device="cuda"))
    with pytest.raises(pdsp.PdspError, match="FFT size must be power of two, got 48"):
        BatchedFft(48, "cuda:0")
    with pytest.raises(pdsp.PdspError, match="FFT input length 5 != size 8"):
        pdsp.Radix2Fft(8).forward([1, 2, 3, 4, 5])
    with pytest.raises(pdsp.PdspError, match="exceeds the supported maximum 268435456"):
        pdsp.Radix2Fft(1 << 29)
    out = plan.forward(torch.zeros((0, 64), device="cuda"))  # empty batch: no launch
    assert out[0].shape == (0, 64)
    assert pdsp.lib.pdsp_plan_cache_clear() == 0
    r = pdsp.spectrum([], {"sampleRate": 8})  # nextPowerOfTwo(0) = 1
    assert len(r.amplitude) == 1 and r.amplitude[0] == 0 and r.peak.index == 0


@pytest.mark.parametrize("n", [32, 64, 128, 256])
@pytest.mark.parametrize("batch", [1, 3, 127, 128, 129, 1000])
def test_staged_small_n_kernel_vs_direct_vs_oracle(pdsp, oracle_mod, n, batch):
    """32 <= N <= 256 runs on fft_staged_kernel (coalesced 16-byte I/O staged through LDS);
    pdsp_set_staged_small(0) routes the same call to the direct kernel."""
    import torch
    from pragma_dsp_amd.batch import BatchedFft
    rng = np.random.default_rng(n + batch)
    re = rng.standard_normal((batch, n)).astype(np.float32)
    im = rng.standard_normal((batch, n)).astype(np.float32)
    plan = BatchedFft(n, "cuda:0")
    dre, dim = torch.from_numpy(re).cuda(), torch.from_numpy(im).cuda()
    wre, wim = oracle_mod.Plan(n).forward_complex(re, im)
    rre, rim = oracle_mod.Plan(n).forward(re)
    out = {}
    for mode in (1, 0):
        prev = pdsp.lib.pdsp_set_staged_small(mode)
        try:
            guard = torch.full((batch + 2, n), 777.0, device="cuda")
            ore, oim = guard[1:batch + 1], torch.empty((batch, n), device="cuda")
            plan.forward(dre, dim, out=(ore, oim))
            r2, i2 = plan.forward(dre)
            b1, b2 = plan.inverse(ore, oim)
            torch.cuda.synchronize()
        finally:
            pdsp.lib.pdsp_set_staged_small(prev)
        assert bool((guard[0] == 777.0).all()) and bool((guard[-1] == 777.0).all())
        got = ore.cpu().numpy().astype(np.float64) + 1j * oim.cpu().numpy()
        assert rel_err(got, wre + 1j * wim) <= TOL
        assert rel_err(r2.cpu().numpy().astype(np.float64) + 1j * i2.cpu().numpy(), rre + 1j * rim) <= TOL
        assert rel_err(b1.cpu().numpy(), re) <= TOL and rel_err(b2.cpu().numpy(), im) <= TOL
        out[mode] = got
    assert rel_err(out[1], out[0]) <= 2e-6
    # an unaligned view (4-byte offset) must fall back to the direct kernel and still be right
    flat = torch.zeros(batch * n + 1, device="cuda")
    view = flat[1:].view(batch, n)
    view.copy_(dre)
    u1, u2 = plan.forward(view, dim)
    assert rel_err(u1.cpu().numpy().astype(np.float64) + 1j * u2.cpu().numpy(), wre + 1j * wim) <= TOL


@pytest.mark.parametrize("batch", [1, 5])
def test_split4_kernel_n16384_vs_single_pass_vs_oracle(pdsp, oracle_mod, batch):
    """N = 16384 rows on 16-byte aligned planes run on fft_split4_kernel (four 4096-point
    sub-transforms + a radix-4 combine in registers); pdsp_set_split16k(0), or a 4-byte aligned
    view, routes the same call to fft_stockham_kernel<14>."""
    import torch
    from pragma_dsp_amd.batch import BatchedFft
    n = 16384
    rng = np.random.default_rng(4 + batch)
    re = rng.standard_normal((batch, n)).astype(np.float32)
    im = rng.standard_normal((batch, n)).astype(np.float32)
    re[0] = np.cos(2 * np.pi * 1234 * np.arange(n) / n)  # one exact bin: X[1234] = N/2 (+ i*0)
    im[0] = 0
    plan = BatchedFft(n, "cuda:0")
    dre, dim = torch.from_numpy(re).cuda(), torch.from_numpy(im).cuda()
    wre, wim = oracle_mod.Plan(n).forward_complex(re, im)
    rre, rim = oracle_mod.Plan(n).forward(re)
    out = {}
    for mode in (1, 0):
        prev = pdsp.lib.pdsp_set_split16k(mode)
        prev_rp = pdsp.lib.pdsp_set_real_packed(0)  # real rows on this kernel's LoadReal form, not on fft_real_kernel
        try:
            guard = torch.full((batch + 2, n), 777.0, device="cuda")
            ore, oim = guard[1:batch + 1], torch.empty((batch, n), device="cuda")
            plan.forward(dre, dim, out=(ore, oim))
            r2, i2 = plan.forward(dre)
            b1, b2 = plan.inverse(ore, oim)
            torch.cuda.synchronize()
        finally:
            pdsp.lib.pdsp_set_split16k(prev)
            pdsp.lib.pdsp_set_real_packed(prev_rp)
        assert bool((guard[0] == 777.0).all()) and bool((guard[-1] == 777.0).all())
        got = ore.cpu().numpy().astype(np.float64) + 1j * oim.cpu().numpy()
        assert rel_err(got, wre + 1j * wim) <= TOL
        assert abs(got[0, 1234] - n / 2) < 1e-2 and abs(got[0, n - 1234] - n / 2) < 1e-2
        assert rel_err(r2.cpu().numpy().astype(np.float64) + 1j * i2.cpu().numpy(), rre + 1j * rim) <= TOL
        assert rel_err(b1.cpu().numpy(), re) <= TOL and rel_err(b2.cpu().numpy(), im) <= TOL
        out[mode] = got
    assert rel_err(out[1], out[0]) <= 2e-6
    flat = torch.zeros(batch * n + 1, device="cuda")  # 4-byte aligned view: single-pass kernel
    view = flat[1:].view(batch, n)
    view.copy_(dre)
    ure, uim = plan.forward(view, dim)
    assert rel_err(ure.cpu().numpy().astype(np.float64) + 1j * uim.cpu().numpy(), wre + 1j * wim) <= TOL


@pytest.mark.parametrize("dtype_name", ["float32", "float64"])
def test_split2_kernel_n8192_vs_single_pass_vs_oracle(pdsp, oracle_mod, dtype_name):
    """N = 8192 rows (f64; f32 real input; f32 complex with switch bit 1) run on fft_split2_kernel (two
    4096-point sub-transforms + a radix-2 combine in registers); pdsp_set_split16k(0) routes the same
    calls to fft_stockham_kernel<13>."""
    import torch
    from pragma_dsp_amd.batch import BatchedFft
    n, batch = 8192, 5
    dt = getattr(torch, dtype_name)
    npdt = np.float32 if dtype_name == "float32" else np.float64
    tol = TOL if dtype_name == "float32" else 1e-13
    rng = np.random.default_rng(8192)
    re = rng.standard_normal((batch, n)).astype(npdt)
    im = rng.standard_normal((batch, n)).astype(npdt)
    plan = BatchedFft(n, "cuda:0", dtype=dt)
    dre, dim = torch.from_numpy(re).cuda(), torch.from_numpy(im).cuda()
    wre, wim = oracle_mod.Plan(n).forward_complex(re, im)
    rre, rim = oracle_mod.Plan(n).forward(re)
    out = {}
    for mode in (3, 0):
        prev = pdsp.lib.pdsp_set_split16k(mode)
        prev_rp = pdsp.lib.pdsp_set_real_packed(0)  # real rows on this kernel's LoadReal form, not on fft_real_kernel
        try:
            guard = torch.full((batch + 2, n), 777.0, device="cuda", dtype=dt)
            ore, oim = guard[1:batch + 1], torch.empty((batch, n), device="cuda", dtype=dt)
            plan.forward(dre, dim, out=(ore, oim))
            r2, i2 = plan.forward(dre)
            b1, b2 = plan.inverse(ore, oim)
            torch.cuda.synchronize()
        finally:
            pdsp.lib.pdsp_set_split16k(prev)
            pdsp.lib.pdsp_set_real_packed(prev_rp)
        assert bool((guard[0] == 777.0).all()) and bool((guard[-1] == 777.0).all())
        got = ore.cpu().numpy().astype(np.float64) + 1j * oim.cpu().numpy()
        assert rel_err(got, wre + 1j * wim) <= tol
        assert rel_err(r2.cpu().numpy().astype(np.float64) + 1j * i2.cpu().numpy(), rre + 1j * rim) <= tol
        assert rel_err(b1.cpu().numpy(), re) <= tol and rel_err(b2.cpu().numpy(), im) <= tol
        out[mode] = got
    assert rel_err(out[3], out[0]) <= (2e-6 if dtype_name == "float32" else 1e-14)


@pytest.mark.parametrize("n", [64, 256, 1024, 4096, 16384, 65536])
def test_nan_stays_in_its_own_row(oracle_mod, n):
    """NaN/Inf propagate as in the reference (plain IEEE arithmetic), but only through the
    row that holds them: rows share workgroups and LDS buffers in the batched kernels, so a poisoned
    row must not leak into its neighbours.  findPeak on an all-NaN row is bin 0 (every `>` is false)."""
    import torch
    from pragma_dsp_amd.batch import BatchedFft
    rng = np.random.default_rng(n)
    batch = 9 if n <= 4096 else 3
    x = rng.standard_normal((batch, n)).astype(np.float32)
    bad = 1
    x[bad, n // 3] = np.nan
    plan = BatchedFft(n, "cuda:0")
    dx = torch.from_numpy(x).cuda()
    re, im = plan.forward(dx)
    re, im = re.cpu().numpy(), im.cpu().numpy()
    assert np.isnan(re[bad]).all() and np.isnan(im[bad]).all()
    good = [b for b in range(batch) if b != bad]
    wre, wim = oracle_mod.Plan(n).forward(x[good])
    assert rel_err(re[good].astype(np.float64) + 1j * im[good], wre + 1j * wim) <= TOL
    amp, _, pk = plan.spectrum(dx, "hann", "one", want_peak=True)
    amp, pk = amp.cpu().numpy(), pk.cpu().numpy()
    assert np.isnan(amp[bad]).all() and pk[bad] == 0
    assert np.isfinite(amp[good]).all()
    idx, _, pamp, _, _, _ = plan.spectrum_peaks(dx, "hann", "one", 48000.0)
    assert int(idx[bad]) == 0 and np.isfinite(pamp.cpu().numpy()[good]).all()
    wamp, _, wpk = oracle_mod.Plan(n).spectrum_batch(x[good], window=oracle_mod.create_window("hann", n).astype(np.float32),
                                                     want_peak=True)
    assert rel_err(amp[good], wamp) <= TOL
    y = x.copy()
    y[bad] = 0
    y[bad, 5] = np.inf          # Inf - Inf appears inside the butterflies: the row turns non-finite, others do not
    re, im = plan.forward(torch.from_numpy(y).cuda())
    assert not np.isfinite(re.cpu().numpy()[bad]).any() and np.isfinite(re.cpu().numpy()[good]).all()


@pytest.mark.parametrize("n", [2, 4, 8, 16, 32])
@pytest.mark.parametrize("batch", [1, 3, 127, 128, 129, 255, 256, 257, 2049, 5000])
def test_tiny_staged_kernel_vs_direct_vs_oracle(pdsp, oracle_mod, n, batch):
    """2 <= N <= 16 on 16-byte aligned planes runs on fft_tiny_staged_kernel (one thread per row, the
    workgroup's chunk staged through LDS) for transforms and for whole-frame amplitude spectra (those
    also at N = 32, where the transforms take fft_staged_kernel);
    pdsp_set_staged_small(0) routes the same calls to the direct kernels.  Odd batches of N = 2 rows
    exercise the group that straddles the end of the plane."""
    import torch
    from pragma_dsp_amd.batch import BatchedFft
    rng = np.random.default_rng(16 * n + batch)
    re = rng.standard_normal((batch, n)).astype(np.float32)
    im = rng.standard_normal((batch, n)).astype(np.float32)
    plan = BatchedFft(n, "cuda:0")
    dre, dim = torch.from_numpy(re).cuda(), torch.from_numpy(im).cuda()
    o = oracle_mod.Plan(n)
    wre, wim = o.forward_complex(re, im)
    rre, rim = o.forward(re)
    win = oracle_mod.create_window("hamming", n).astype(np.float32)
    res = {}
    for mode in (1, 0):
        prev = pdsp.lib.pdsp_set_staged_small(mode)
        try:
            guard = torch.full((batch + 2, n), 777.0, device="cuda")
            ore, oim = guard[1:batch + 1], torch.empty((batch, n), device="cuda")
            plan.forward(dre, dim, out=(ore, oim))
            r2, i2 = plan.forward(dre)
            b1, b2 = plan.inverse(ore, oim)
            amps = {}
            for sides in ("one", "two"):
                bins = n // 2 + 1 if sides == "one" else n
                abuf = torch.full((batch * bins + 32,), -7.0, device="cuda")
                aout = abuf[:batch * bins].view(batch, bins)
                _, _, pk = plan.spectrum(dre, "hamming", sides, want_peak=True, out=aout)
                torch.cuda.synchronize()
                assert bool((abuf[batch * bins:] == -7.0).all())
                amps[sides] = (aout.cpu().numpy().astype(np.float64), pk.cpu().numpy())
            torch.cuda.synchronize()
        finally:
            pdsp.lib.pdsp_set_staged_small(prev)
        assert bool((guard[0] == 777.0).all()) and bool((guard[-1] == 777.0).all())
        got = ore.cpu().numpy().astype(np.float64) + 1j * oim.cpu().numpy()
        assert rel_err(got, wre + 1j * wim) <= TOL
        assert rel_err(r2.cpu().numpy().astype(np.float64) + 1j * i2.cpu().numpy(), rre + 1j * rim) <= TOL
        assert rel_err(b1.cpu().numpy(), re) <= TOL and rel_err(b2.cpu().numpy(), im) <= TOL
        for sides in ("one", "two"):
            wamp, _, wpk = o.spectrum_batch(re, window=win, two_sided=(sides == "two"), want_peak=True)
            a, pk = amps[sides]
            assert a.shape == wamp.shape and rel_err(a, wamp) <= TOL
            for b in range(batch):
                assert abs(wamp[b, pk[b]] - wamp[b, wpk[b]]) <= 2 * TOL * wamp[b].max()
        res[mode] = got
    assert rel_err(res[1], res[0]) <= 2e-6

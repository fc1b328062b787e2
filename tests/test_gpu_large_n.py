"""N beyond the single-pass LDS limit (the reference takes any power of two): the four-step paths
-- fused columns for N1 <= 16 (f32 up to 2^18, f64 up to 2^17), the general transposed form above
(f32 up to 2^28, f64 up to 2^26) -- vs the f64 oracle."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("log2n", [15, 16, 17, 18, 19, 20, 22])
def test_large_complex_real_inverse_f32(oracle_mod, log2n):
    import torch
    from pragma_dsp_amd.batch import BatchedFft
    n = 1 << log2n
    rng = np.random.default_rng(log2n)
    batch = 3 if log2n <= 20 else 1
    re = rng.standard_normal((batch, n)).astype(np.float32)
    im = rng.standard_normal((batch, n)).astype(np.float32)
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


@pytest.mark.parametrize("log2n", [14, 15, 17, 18, 19, 21])
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


def test_general_four_step_spectrum_aliasing_and_dropin(pdsp, oracle_mod):
    """N = 2^20 (N1 = 64 rows x N2 = 16384): windowed spectrum rows, fused peaks, an in-place
    transform (output planes aliasing the input), and the drop-in on a 600,000-sample signal."""
    import torch
    from pragma_dsp_amd.batch import BatchedFft
    n = 1 << 20
    rng = np.random.default_rng(20)
    t = np.arange(n)
    x = (rng.standard_normal((2, n)) * 0.1 + np.sin(2 * np.pi * 54321 * t / n)).astype(np.float32)
    plan = BatchedFft(n, "cuda:0")
    win = oracle_mod.create_window("blackman", n).astype(np.float32)
    for sides in ("one", "two"):
        amp, ph, pk = plan.spectrum(torch.from_numpy(x).cuda(), "blackman", sides, want_phase=True, want_peak=True)
        wamp, wph, wpk = oracle_mod.Plan(n).spectrum_batch(x, window=win, two_sided=(sides == "two"),
                                                           want_phase=True, want_peak=True)
        assert amp.shape == wamp.shape and rel_err(amp.cpu().numpy(), wamp) <= 1e-5
        assert all(int(v) in (54321, n - 54321 if sides == "two" else 54321) for v in pk.cpu())
        mask = wamp > 1e-2 * wamp.max(axis=-1, keepdims=True)
        d = np.abs((ph.cpu().numpy() - wph + np.pi) % (2 * np.pi) - np.pi)
        assert d[mask].max() <= 2e-3
    idx, freq, a, p, _, _ = plan.spectrum_peaks(torch.from_numpy(x[:, :600000].copy()).cuda(), "hann", "one", 48000.0)
    assert [int(v) for v in idx.cpu()] == [54321, 54321]
    # in place: out planes are the input planes
    re = torch.from_numpy(x[:1].copy()).cuda()
    im = torch.zeros_like(re)
    want_re, want_im = oracle_mod.Plan(n).forward(x[:1])
    plan.forward(re, im, out=(re, im))
    assert rel_err(re.cpu().numpy().astype(np.float64) + 1j * im.cpu().numpy(), want_re + 1j * want_im) <= 1e-5
    sig = np.sin(2 * np.pi * 440.0 * np.arange(600000) / 48000.0)
    g = pdsp.spectrum(sig, {"sampleRate": 48000, "window": "hann"})   # nextPowerOfTwo(600000) = 2^20, f64
    w = oracle_mod.spectrum(sig, sample_rate=48000, window="hann")
    assert len(g.amplitude) == (1 << 19) + 1 and g.peak.index == w["peak"]["index"]
    assert np.abs(g.amplitude - w["amplitude"]).max() <= 1e-12


def test_general_four_step_n_2_24(oracle_mod):
    """N = 2^24 = 1024 x 16384 (f32) and 2048 x 8192 (f64): larger N1 than the other cases exercise."""
    import torch
    from pragma_dsp_amd.batch import BatchedFft
    n = 1 << 24
    rng = np.random.default_rng(24)
    re = rng.standard_normal((1, n))
    im = rng.standard_normal((1, n))
    wre, wim = oracle_mod.Plan(n).forward_complex(re, im)
    want = wre + 1j * wim
    for dt, npdt, tol in ((torch.float32, np.float32, 1e-5), (torch.float64, np.float64, 1e-13)):
        plan = BatchedFft(n, "cuda:0", dtype=dt)
        dre, dim = torch.from_numpy(re.astype(npdt)).cuda(), torch.from_numpy(im.astype(npdt)).cuda()
        ore, oim = plan.forward(dre, dim)
        assert rel_err(ore.cpu().numpy().astype(np.float64) + 1j * oim.cpu().numpy(), want) <= tol
        bre, bim = plan.inverse(ore, oim)
        assert rel_err(bre.cpu().numpy(), re) <= tol and rel_err(bim.cpu().numpy(), im) <= tol
        del plan, dre, dim, ore, oim, bre, bim


@pytest.mark.parametrize("log2n", [15, 16, 17, 18, 19, 20, 21, 22, 23, 24])
def test_tile_passes_vs_fourstep_vs_oracle(pdsp, oracle_mod, log2n):
    """f32 transforms beyond the single-pass limit on tile_pass_kernel -- balanced factors, two passes over
    HBM for 2^15 <= N <= 2^18 and three for 2^19 <= N <= 2^27, the scratch planes between the first two of three passes tile-major (default) and in natural order
    (pdsp_set_twopass(3): bit-identical results) -- against round 1's four-step forms (pdsp_set_twopass(0): three / five passes) and the oracle:
    complex, real input, inverse, an in-place call (output planes = input planes) and an odd batch."""
    import torch
    from pragma_dsp_amd.batch import BatchedFft
    n, batch = 1 << log2n, (5 if log2n <= 20 else 3 if log2n <= 22 else 2)
    rng = np.random.default_rng(100 + log2n)
    re = rng.standard_normal((batch, n)).astype(np.float32)
    im = rng.standard_normal((batch, n)).astype(np.float32)
    imp = batch - 1
    re[imp] = 0.0
    im[imp] = 0.0
    re[imp, 3] = 1.0                                     # an impulse row: |X| = 1 everywhere
    plan = BatchedFft(n, "cuda:0")
    wre, wim = oracle_mod.Plan(n).forward_complex(re, im)
    want = wre + 1j * wim
    rre_w, rim_w = oracle_mod.Plan(n).forward(re)
    res = {}
    # 1: default (2^15 / 2^16 out of place on fft_paired_kernel); 5: tile passes, current form; 3: their first form
    # (natural-order scratch, 512-point factors on 16-wide tiles); 0: round 1's four-step forms
    for mode in (1, 5, 3, 0):
        prev = pdsp.lib.pdsp_set_twopass(mode)
        try:
            dre, dim = torch.from_numpy(re).cuda(), torch.from_numpy(im).cuda()
            ore, oim = plan.forward(dre, dim)
            got = ore.cpu().numpy().astype(np.float64) + 1j * oim.cpu().numpy()
            assert rel_err(got, want) <= 1e-5
            assert np.abs(np.abs(got[imp]) - 1.0).max() < 1e-5
            bre, bim = plan.inverse(ore, oim)
            top = max(np.abs(re).max(), np.abs(im).max())
            assert np.abs(bre.cpu().numpy() - re).max() <= 1e-5 * top * 4
            assert np.abs(bim.cpu().numpy() - im).max() <= 1e-5 * top * 4
            rre, rim = plan.forward(dre)                  # Radix2Fft.forward: real input
            assert rel_err(rre.cpu().numpy().astype(np.float64) + 1j * rim.cpu().numpy(), rre_w + 1j * rim_w) <= 1e-5
            plan.forward(dre, dim, out=(dre, dim))        # in place
            torch.cuda.synchronize()
            assert rel_err(dre.cpu().numpy().astype(np.float64) + 1j * dim.cpu().numpy(), want) <= 1e-5
        finally:
            pdsp.lib.pdsp_set_twopass(prev)
        res[mode] = got
    assert rel_err(res[1], res[0]) <= 2e-6 and rel_err(res[1], res[5]) <= 2e-6
    if log2n in (15, 16):                                 # default = fft_paired_kernel: compare the two tile-pass forms
        assert np.array_equal(res[5], res[3])
    elif log2n in (17, 18):                                 # 512-point factors: tile_rows512 / tile_cols512_kernel vs plain tiles
        assert rel_err(res[1], res[3]) <= 2e-6
    else:
        assert np.array_equal(res[1], res[3])             # the same arithmetic, another scratch layout


@pytest.mark.parametrize("log2n", [26, 27])
def test_tile_passes_with_512_point_column_factors(pdsp, log2n):
    """2^26 = 256 * 512 * 512 and 2^27 = 512^3: the sizes whose FIRST / MIDDLE passes are 512-point column factors
    (tile_cols512_kernel, reading tile-major scratch in the middle) -- one row, forward and in-place inverse, against
    the first form of the tile passes (pdsp_set_twopass(3)), round 1's four-step form (0: an independent
    factorisation) and numpy's f64 transform.  The C oracle is not run at these sizes (minutes); numpy.fft is
    not reference-pinned, the four-step form is pinned at the smaller sizes above."""
    import torch
    from pragma_dsp_amd.batch import BatchedFft
    n = 1 << log2n
    rng = np.random.default_rng(300 + log2n)
    re = rng.standard_normal((1, n)).astype(np.float32)
    im = rng.standard_normal((1, n)).astype(np.float32)
    want = np.fft.fft(re[0].astype(np.float64) + 1j * im[0].astype(np.float64))
    plan = BatchedFft(n, "cuda:0")
    dre, dim = torch.from_numpy(re).cuda(), torch.from_numpy(im).cuda()
    res = {}
    for mode in (1, 3, 0):
        prev = pdsp.lib.pdsp_set_twopass(mode)
        try:
            ore, oim = plan.forward(dre, dim)
            res[mode] = ore[0].cpu().numpy().astype(np.float64) + 1j * oim[0].cpu().numpy()
            if mode == 1:
                plan.inverse(ore, oim, out=(ore, oim))   # in place
                torch.cuda.synchronize()
                assert np.abs(ore[0].cpu().numpy() - re[0]).max() <= 4e-5 * np.abs(re).max()
                assert np.abs(oim[0].cpu().numpy() - im[0]).max() <= 4e-5 * np.abs(im).max()
        finally:
            pdsp.lib.pdsp_set_twopass(prev)
    top = np.abs(want).max()
    assert np.abs(res[1] - want).max() <= 1e-5 * top
    assert np.abs(res[1] - res[3]).max() <= 2e-6 * top and np.abs(res[1] - res[0]).max() <= 2e-6 * top


@pytest.mark.parametrize("log2n", [15, 16, 17, 18, 19, 20, 21, 22, 24])
def test_long_frame_spectrum_on_tile_passes_vs_fourstep_vs_oracle(pdsp, oracle_mod, log2n):
    """spectrum() of frames longer than the single-pass limit (whole 16-byte aligned f32 frames): the packed-real
    form on tile passes -- the N/2-point transform of (x*w)[2m] + i (x*w)[2m+1] read straight from the frame,
    then split_amp_rows_kernel -- against round 1's four-step form on (x*w, 0) (pdsp_set_twopass(0)) and the
    oracle, one- and two-sided, with phase rows, peak indices and fused-peak records.  A partial (zero-padded)
    frame takes the four-step form either way; the same assertions hold."""
    import torch
    from pragma_dsp_amd.batch import BatchedFft
    n = 1 << log2n
    tone = n // 37
    rng = np.random.default_rng(200 + log2n)
    t = np.arange(n)
    x = (rng.standard_normal((3, n)) * 0.1 + np.sin(2 * np.pi * tone * t / n)).astype(np.float32)
    plan = BatchedFft(n, "cuda:0")
    dx = torch.from_numpy(x).cuda()
    # rect / hann / blackman by kind (the plan's own tables: createWindow fused into the first pass), hamming as
    # the caller's own tensor (read as a table)
    for window in ("rect", "hann", "blackman", "hamming-table") if log2n <= 20 else ("rect", "hann"):
        kind = window.split("-")[0]
        win = oracle_mod.create_window(kind, n).astype(np.float32) if kind != "rect" else None
        if window.endswith("-table"):
            window = torch.from_numpy(win).cuda()
        for sides in ("one", "two"):
            wamp, wph, wpk = oracle_mod.Plan(n).spectrum_batch(x, window=win, two_sided=(sides == "two"),
                                                               want_phase=True, want_peak=True)
            res = {}
            for mode in (1, 3, 0):
                prev = pdsp.lib.pdsp_set_twopass(mode)
                try:
                    amp, ph, pk = plan.spectrum(dx, window, sides, want_phase=True, want_peak=True)
                    torch.cuda.synchronize()
                finally:
                    pdsp.lib.pdsp_set_twopass(prev)
                a = amp.cpu().numpy()
                assert a.shape == wamp.shape and rel_err(a, wamp) <= 1e-5
                assert all(int(v) in (tone, n - tone if sides == "two" else tone) for v in pk.cpu())
                mask = wamp > 1e-2 * wamp.max(axis=-1, keepdims=True)
                d = np.abs((ph.cpu().numpy() - wph + np.pi) % (2 * np.pi) - np.pi)
                assert d[mask].max() <= 2e-3
                res[mode] = a
            assert rel_err(res[1], res[0]) <= 2e-6
            # 2^16-sample frames: fft_paired_kernel (one pass) against the tile passes; the half-size transforms of
            # 2^18- and 2^19-sample frames end in a 512-point factor (tile_rows512_kernel against plain tiles):
            # same results within rounding; elsewhere bit for bit
            assert rel_err(res[1], res[3]) <= 2e-6 if log2n in (16, 18, 19) else np.array_equal(res[1], res[3])
    idx, freq, a, p, _, _ = plan.spectrum_peaks(dx, "hann", "one", 48000.0)
    assert [int(v) for v in idx.cpu()] == [tone] * 3
    short = torch.from_numpy(x[:, : n - 1000].copy()).cuda()          # zero-padded frames: four-step form
    amp, _, _ = plan.spectrum(short, "hann", "one")
    xs = x.copy()
    xs[:, n - 1000:] = 0
    wamp, _, _ = oracle_mod.Plan(n).spectrum_batch(xs, window=oracle_mod.create_window("hann", n).astype(np.float32))
    assert rel_err(amp.cpu().numpy(), wamp) <= 1e-5


def test_partially_overlapping_output_planes_at_2p19(pdsp, oracle_mod):
    """the three-pass tile path used the output planes as its first scratch pair unless the POINTERS were
    equal; an output that starts one row into the input buffer overlaps it without being equal, and pass 1 then
    overwrote input other workgroups had not read yet.  Byte-range test now (planes_overlap)."""
    import torch
    from pragma_dsp_amd.batch import BatchedFft
    n, batch = 1 << 19, 3
    rng = np.random.default_rng(19)
    re = rng.standard_normal((batch, n)).astype(np.float32)
    im = rng.standard_normal((batch, n)).astype(np.float32)
    wre, wim = oracle_mod.Plan(n).forward_complex(re, im)
    want = wre + 1j * wim
    plan = BatchedFft(n, "cuda:0")
    big_re = torch.zeros((batch + 1, n), device="cuda")
    big_im = torch.zeros((batch + 1, n), device="cuda")
    big_re[:batch].copy_(torch.from_numpy(re))
    big_im[:batch].copy_(torch.from_numpy(im))
    # input = rows 0 .. batch-1, output = rows 1 .. batch of the same buffers: shifted by one row
    plan.forward(big_re[:batch], big_im[:batch], out=(big_re[1:], big_im[1:]))
    torch.cuda.synchronize()
    got = big_re[1:].cpu().numpy().astype(np.float64) + 1j * big_im[1:].cpu().numpy()
    assert rel_err(got, want) <= 1e-5
    # the same with only ONE plane overlapping (real output over the imaginary input, shifted)
    big = torch.zeros((batch + 1, n), device="cuda")
    big[:batch].copy_(torch.from_numpy(im))
    dre = torch.from_numpy(re).cuda()
    oim = torch.empty((batch, n), device="cuda")
    plan.forward(dre, big[:batch], out=(big[1:], oim))
    torch.cuda.synchronize()
    got = big[1:].cpu().numpy().astype(np.float64) + 1j * oim.cpu().numpy()
    assert rel_err(got, want) <= 1e-5


def test_scratch_planes_go_back_to_the_device_with_the_plan(pdsp):
    """the multi-pass paths' scratch planes come from a pool of the engine's own with an unlimited
    release threshold; they used to stay with the process until pdsp_plan_cache_clear().  Destroying the plan
    (BatchedFft.close / collection) now trims the pool."""
    import gc
    import torch
    from pragma_dsp_amd.batch import BatchedFft
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    pdsp.lib.pdsp_plan_cache_clear()
    free0, _ = torch.cuda.mem_get_info()
    n, batch = 1 << 22, 32                      # 512 MiB per plane: ~2 GiB of scratch on the three-pass path
    plan = BatchedFft(n, "cuda:0")
    re = torch.randn((batch, n), device="cuda")
    im = torch.randn((batch, n), device="cuda")
    plan.forward(re, im, out=(re, im))         # in place: both scratch pairs come from the pool
    torch.cuda.synchronize()
    del re, im
    torch.cuda.empty_cache()
    # (what the pool holds at this point is the runtime's business: on ROCm 7.2 hipMemGetInfo already reports the
    # freed planes as free here; the contract checked is the one after the plan is gone)
    plan.close()
    gc.collect()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < (256 << 20), f"scratch planes still pinned after the plan is gone: {(free0 - free1) >> 20} MiB"

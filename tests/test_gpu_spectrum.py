"""Fused spectrum kernel (pdsp_spectrum_f32) vs the f64 oracle's row-by-row spectrum():
every size, every window, one-/two-sided, phase, peak, zero-padding, truncation,
unaligned rows.  Covers both device paths: the packed-real kernel (N >= 64) and the
complex fallback (N < 64 or an unaligned window)."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-5


def wrap(d):
    return np.abs((d + np.pi) % (2 * np.pi) - np.pi)


def run(plan, x, window, sides, **kw):
    import torch
    amp, ph, pk = plan.spectrum(torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda(), window, sides, **kw)
    torch.cuda.synchronize()
    return (amp.cpu().numpy().astype(np.float64), None if ph is None else ph.cpu().numpy().astype(np.float64),
            None if pk is None else pk.cpu().numpy())


@pytest.mark.parametrize("log2n", list(range(0, 15)))
@pytest.mark.parametrize("sides", ["one", "two"])
def test_amplitude_phase_all_sizes(oracle_mod, log2n, sides):
    from pragma_dsp_amd.batch import BatchedFft
    n = 1 << log2n
    rng = np.random.default_rng(300 + log2n)
    batch = 19 if n <= 4096 else 3
    x = rng.standard_normal((batch, n)).astype(np.float32)
    plan = BatchedFft(n, "cuda:0")
    window = ["rect", "hann", "hamming", "blackman"][log2n % 4]
    amp, ph, pk = run(plan, x, window, sides, want_phase=True, want_peak=True)
    win = oracle_mod.create_window(window, n).astype(np.float32) if (window != "rect" and n > 1) else None
    wamp, wph, wpk = oracle_mod.Plan(n).spectrum_batch(x, window=win, two_sided=(sides == "two"),
                                                       want_phase=True, want_peak=True)
    assert amp.shape == wamp.shape
    assert rel_err(amp, wamp) <= TOL
    # phase only where the bin is well above the f32 noise floor
    mask = wamp > 1e-3 * wamp.max(axis=-1, keepdims=True)
    assert wrap(ph - wph)[mask].max(initial=0) <= 2e-3
    for b in range(batch):  # peak: same bin, or a bin whose amplitude ties within tolerance
        assert abs(wamp[b, pk[b]] - wamp[b, wpk[b]]) <= 2 * TOL * wamp[b].max()
        assert (pk[b] >= 1) or wamp[b, 1:].max(initial=0) == 0


@pytest.mark.parametrize("n,length", [(1024, 1000), (1024, 777), (1024, 1), (4096, 4095), (256, 300), (64, 64), (16384, 10001)])
def test_zero_padding_and_truncation(oracle_mod, n, length):
    from pragma_dsp_amd.batch import BatchedFft
    rng = np.random.default_rng(n + length)
    x = rng.standard_normal((5, length)).astype(np.float32)  # odd lengths => unaligned rows (VEC2 off)
    plan = BatchedFft(n, "cuda:0")
    amp, _, _ = run(plan, x, "hann", "one")
    frame = np.zeros((5, n), dtype=np.float32)
    frame[:, :min(n, length)] = x[:, :n]
    wamp, _, _ = oracle_mod.Plan(n).spectrum_batch(frame, window=oracle_mod.create_window("hann", n).astype(np.float32))
    assert rel_err(amp, wamp) <= TOL


def test_custom_and_unaligned_window(oracle_mod):
    import torch
    from pragma_dsp_amd.batch import BatchedFft
    n = 2048
    rng = np.random.default_rng(9)
    x = rng.standard_normal((7, n)).astype(np.float32)
    w = rng.random(n + 1).astype(np.float32)
    plan = BatchedFft(n, "cuda:0")
    dw = torch.from_numpy(w).cuda()
    for off in (0, 1):  # offset 1 => 4-byte aligned only => complex fallback kernel
        win = dw[off:off + n]
        amp, _, _ = plan.spectrum(torch.from_numpy(x).cuda(), win, "one")
        wamp, _, _ = oracle_mod.Plan(n).spectrum_batch(x, window=w[off:off + n])
        assert rel_err(amp.cpu().numpy(), wamp) <= TOL


def test_exact_zero_and_dc_semantics(oracle_mod, reallife):
    """spectrum() of zeros is exactly 0 with peak 0 (edge_cases.test.ts:22-38); pure DC
    leaves the peak at bin 0 (scaling.test.ts:185-201); DC + sine skips DC."""
    from pragma_dsp_amd.batch import BatchedFft
    plan = BatchedFft(1024, "cuda:0")
    x = np.stack([reallife["zeros/signal"], reallife["dc_level1/signal"], reallife["dc_plus_sine_bin8/signal"],
                  reallife["nyquist/signal"], reallife["impulse_pos0/signal"], reallife["sine_bin8_amp1.0/signal"]])
    amp, _, pk = run(plan, x, "rect", "one", want_peak=True)
    assert not amp[0].any() and pk[0] == 0
    assert pk[1] == 0 and abs(amp[1, 0] - 1) < 1e-6 and np.abs(amp[1, 1:]).max() < 1e-6
    assert pk[2] == 8
    assert pk[3] == 512 and abs(amp[3, 512] - 1) < 1e-6          # Nyquist not doubled
    assert np.abs(amp[4, 1:512] - 2 / 1024).max() < 1e-8          # impulse: flat
    assert pk[5] == 8 and abs(amp[5, 8] - 1) < 1e-5
    amp2, _, pk2 = run(plan, x, "rect", "two", want_peak=True)
    assert pk2[5] == 8 and abs(amp2[5, 8] - 0.5) < 1e-5 and abs(amp2[5, 1016] - 0.5) < 1e-5
    assert amp2.shape == (6, 1024)


def test_spectrum_full_size_properties():
    """Size-independent checks at a BASELINE-sized chunk (N=16384 x 4096 frames):
    Parseval on the one-sided amplitudes, and linearity in the input scale."""
    import torch
    from pragma_dsp_amd.batch import BatchedFft
    n, b = 16384, 4096
    g = torch.Generator(device="cuda")
    g.manual_seed(1337)
    x = torch.randn((b, n), generator=g, device="cuda", dtype=torch.float32)
    plan = BatchedFft(n, "cuda:0")
    amp, _, _ = plan.spectrum(x, "rect", "one")
    a = amp.double()
    # sum |X_k|^2 over all N bins = N * sum x^2 ; one-sided amp: |X0|/N, 2|Xk|/N, |X_{N/2}|/N
    energy = (a[:, 0] ** 2 + a[:, -1] ** 2 + 0.5 * (a[:, 1:-1] ** 2).sum(dim=1)) * n
    want = (x.double() ** 2).sum(dim=1)
    assert float(((energy - want).abs() / want).max()) < 1e-5
    amp3, _, _ = plan.spectrum(3 * x, "rect", "one")
    assert float(((amp3 - 3 * amp).abs().amax(dim=1) / amp3.amax(dim=1)).max()) < 1e-6


@pytest.mark.parametrize("n", [16384, 8192, 4096, 2048, 1024])
@pytest.mark.parametrize("window", ["rect", "hann", "hamming", "blackman", "custom"])
def test_fast_spectrum_kernels_fused_window_vs_table_vs_oracle(pdsp, oracle_mod, window, n):
    """Whole pair-aligned one-sided frames, the config-4 shape.  N = 16384 runs on spectrum_dif16k_kernel
    (two 4096-point sub-transforms per workgroup, decimation in frequency on top; pdsp_set_split16k(0)
    routes the same call to spectrum_packed_kernel<13>); N = 1024 ... 8192 on spectrum_packed_kernel's FAST
    variant.  Both store ADJACENT bins with 8-byte non-temporal stores.  A window named by kind is the
    plan's own table, so createWindow is FUSED into the kernel (cosine sum in registers);
    pdsp_set_fused_window(0) makes the kernel read the table instead, and a caller's own tensor
    ("custom": a Hann table plus a ripple) is always read as a table.  All against the oracle, plus guard
    cells behind the rows (the 8-byte stores of the last row must not run over), fused findPeak records,
    a zero frame and a DC frame."""
    import torch
    from pragma_dsp_amd.batch import BatchedFft
    bins, tone = n // 2 + 1, n // 21
    rng = np.random.default_rng(16)
    t = np.arange(n)
    x = (rng.standard_normal((9, n)) * 0.3 + np.sin(2 * np.pi * tone * t / n)[None, :]).astype(np.float32)
    x[7] = 0.0          # zeros: exact zeros, peak 0
    x[8] = 1.0          # DC: peak stays at bin 0
    dx = torch.from_numpy(x).cuda()
    plan = BatchedFft(n, "cuda:0")
    if window == "custom":
        win = (oracle_mod.create_window("hann", n) * (1 + 0.1 * np.cos(0.01 * t))).astype(np.float32)
        warg = torch.from_numpy(win).cuda()
    else:
        win = oracle_mod.create_window(window, n).astype(np.float32) if window != "rect" else None
        warg = window
    wamp, _, wpk = oracle_mod.Plan(n).spectrum_batch(x, window=win, want_peak=True)
    res = {}
    modes = ((1, 1), (1, 0), (0, 1)) if n == 16384 else ((1, 1), (1, 0))   # (dif16k kernel, fused window)
    for mode in modes:
        prev = pdsp.lib.pdsp_set_split16k(mode[0])
        prevf = pdsp.lib.pdsp_set_fused_window(mode[1])
        try:
            buf = torch.full((9 * bins + 64,), -7.0, device="cuda")
            out = buf[:9 * bins].view(9, bins)
            amp, _, pki = plan.spectrum(dx, warg, "one", want_peak=True, out=out)   # peak-index array (common tail)
            assert list(pki.cpu().numpy()[:7]) == [tone] * 7 and int(pki[7]) == 0
            idx, freq, pamp, pph, _, _ = plan.spectrum_peaks(dx, warg, "one", 48000.0)
            torch.cuda.synchronize()
        finally:
            pdsp.lib.pdsp_set_split16k(prev)
            pdsp.lib.pdsp_set_fused_window(prevf)
        assert bool((buf[9 * bins:] == -7.0).all())
        res[mode] = (amp.cpu().numpy(), idx.cpu().numpy(), pamp.cpu().numpy(), pph.cpu().numpy())
        assert rel_err(res[mode][0][:7], wamp[:7]) <= TOL
        assert not res[mode][0][7].any() and res[mode][1][7] == 0 and res[mode][2][7] == 0
        # DC frame: bin 0 with a rect window; with a tapering window the window's own bin 1 wins (findPeak skips DC)
        assert res[mode][1][8] == wpk[8] == (0 if window == "rect" else 1)
        assert abs(res[mode][2][8] - wamp[8, wpk[8]]) < 1e-5
        assert list(res[mode][1][:7]) == list(wpk[:7]) == [tone] * 7
        assert np.array_equal(res[mode][0][np.arange(7), res[mode][1][:7]], res[mode][2][:7])
    for other in modes[1:]:
        assert rel_err(res[(1, 1)][0][:7], res[other][0][:7]) <= 2e-6          # the kernels / window forms agree to rounding
        assert np.abs(((res[(1, 1)][3][:7] - res[other][3][:7]) + np.pi) % (2 * np.pi) - np.pi).max() < 1e-4
    if window in ("rect", "custom"):  # nothing to fuse: the same kernel variant runs either way
        assert np.array_equal(res[(1, 1)][0], res[(1, 0)][0])


def test_plan_window_is_create_window_and_fused_window_matches_table(pdsp, oracle_mod):
    """pdsp_plan_window_f32: the plan's own createWindow(type, N) table, equal to the oracle's f64 window
    rounded once to f32, for every type (and any size); the fused evaluation agrees with it to ~2e-7:
    checked through spectrum() of an impulse train whose bins expose the window itself."""
    import torch
    from pragma_dsp_amd._capi import check
    from pragma_dsp_amd.batch import BatchedFft, _ptr, _stream_ptr
    for n in (16384, 1024, 8):
        plan = BatchedFft(n, "cuda:0")
        for kind in ("rect", "hann", "hamming", "blackman"):
            w = plan.window(kind)
            assert w.data_ptr() == plan.window(kind).data_ptr() != 0     # cached per kind
            # read the plan's table back through the product itself: applyWindow(ones, table) = table
            ones = torch.ones((1, n), dtype=torch.float32, device="cuda")
            back = torch.empty_like(ones)
            check(pdsp.lib.pdsp_apply_window_f32(1, n, _ptr(ones), _ptr(w), _ptr(back), _stream_ptr(plan.device)))
            torch.cuda.synchronize()
            want = oracle_mod.create_window(kind, n).astype(np.float32)
            assert np.array_equal(back.cpu().numpy()[0], want)
            assert np.array_equal(w.tensor().cpu().numpy(), want)
    # fused vs table on frames that are ONE unit sample at position p: |X[k]| * N/2 = w[p] for every bin
    for n in (16384, 8192, 4096, 2048, 1024):
        plan = BatchedFft(n, "cuda:0")
        pos = np.unique(np.clip(np.array([0, 1, 2, 3, 63, 64, 65, 511, 512, 513, n // 4 - 1, n // 2 - 1, n // 2, n // 2 + 1,
                                          3 * n // 4 + 13, n - 3, n - 2, n - 1]), 0, n - 1))
        x = np.zeros((len(pos), n), dtype=np.float32)
        x[np.arange(len(pos)), pos] = 1.0
        dx = torch.from_numpy(x).cuda()
        for kind in ("hann", "hamming", "blackman"):
            amp, _, _ = plan.spectrum(dx, kind, "one")
            got = amp.cpu().numpy()[:, 5] * (n / 2)                      # any bin between DC and Nyquist
            want = oracle_mod.create_window(kind, n)[pos]
            assert np.abs(got - want).max() <= 5e-7, (n, kind, np.abs(got - want).max())


@pytest.mark.parametrize("n,batch", [(64, 1), (64, 128), (64, 131), (128, 64), (128, 77), (256, 33), (512, 16),
                                     (512, 19), (512, 1), (256, 4099), (1024, 9)])
@pytest.mark.parametrize("window", ["rect", "blackman"])
def test_staged_small_spectrum_vs_packed_vs_oracle(pdsp, oracle_mod, n, batch, window):
    """64 <= N <= 512, whole aligned frames, one-sided amplitude only: spectrum_staged_kernel (frames
    staged in / amplitude rows staged out through LDS).  pdsp_set_staged_small(0) routes the same call
    to spectrum_packed_kernel.  Batches that do not fill the last workgroup exercise the tail clamp;
    the guard cells behind the output must stay untouched."""
    import torch
    from pragma_dsp_amd.batch import BatchedFft
    rng = np.random.default_rng(n * 7 + batch)
    x = rng.standard_normal((batch, n)).astype(np.float32)
    x[0] = 0.0
    dx = torch.from_numpy(x).cuda()
    plan = BatchedFft(n, "cuda:0")
    win = oracle_mod.create_window(window, n).astype(np.float32) if window != "rect" else None
    wamp, _, _ = oracle_mod.Plan(n).spectrum_batch(x, window=win)
    res = {}
    for mode in (1, 0):
        prev = pdsp.lib.pdsp_set_staged_small(mode)
        try:
            buf = torch.full((batch * (n // 2 + 1) + 64,), -7.0, device="cuda")
            out = buf[:batch * (n // 2 + 1)].view(batch, n // 2 + 1)
            plan.spectrum(dx, window, "one", out=out)
            torch.cuda.synchronize()
        finally:
            pdsp.lib.pdsp_set_staged_small(prev)
        assert bool((buf[batch * (n // 2 + 1):] == -7.0).all())
        res[mode] = out.cpu().numpy()
        assert rel_err(res[mode], wamp) <= TOL
        assert not res[mode][0].any()
    assert rel_err(res[1], res[0]) <= 2e-6

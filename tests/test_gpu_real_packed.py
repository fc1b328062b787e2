"""Radix2Fft.forward(real) rows of 512 <= N <= 16384 (src/core/fft.ts:77-79 semantics: imaginary part taken as
zero) against the f64 oracle.  In f64 -- the drop-in's default arithmetic -- rows of N = 8192 and (from 8 rows up)
N = 16384 run on fft_real_kernel, the N/2-point packed-real transform and the split to X[k], X[k + N/2];
pdsp_set_real_packed(0) (include/pdsp_hip_dev.h) routes the same call to the complex kernels on (x, 0), and both
forms are held to the oracle at 1e-13 and to each other at 1e-14.  At the other sizes and in f32 the switch changes
nothing (those rows always take the complex kernels: DESIGN 4.1c); their cases are the same checks of that path."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu

TOL32, TOL64 = 1e-5, 1e-13


@pytest.mark.parametrize("dtype_name", ["float32", "float64"])
@pytest.mark.parametrize("log2n", [9, 10, 11, 12, 13, 14])
def test_real_rows_packed_vs_complex_kernel_vs_oracle(pdsp, oracle_mod, log2n, dtype_name):
    import torch
    from pragma_dsp_amd.batch import BatchedFft
    n = 1 << log2n
    dt = getattr(torch, dtype_name)
    npdt = np.float32 if dtype_name == "float32" else np.float64
    tol = TOL32 if dtype_name == "float32" else TOL64
    batch = 37 if n <= 4096 else 9  # not a multiple of the rows per workgroup; N = 16384: >= 8 rows take the packed kernel
    rng = np.random.default_rng(7000 + log2n)
    x = rng.standard_normal((batch, n)).astype(npdt)
    idx = np.arange(n)
    x[0] = np.cos(2 * np.pi * 37 * idx / n)             # one exact bin: X[37] = X[N-37] = N/2
    x[1] = 1.0                                          # DC only: X[0] = N
    x[2] = np.where(idx % 2 == 0, 1.0, -1.0)            # Nyquist only: X[N/2] = N
    x[3] = 0.0                                          # zeros stay exactly zero
    x[4] = np.sin(2 * np.pi * (n // 2 - 1) * idx / n)   # the last bin below Nyquist, odd symmetry: purely imaginary
    want = oracle_mod.Plan(n).forward(x)
    want = want[0] + 1j * want[1]
    plan = BatchedFft(n, "cuda:0", dtype=dt)
    dx = torch.from_numpy(x).cuda()
    out = {}
    for mode in (1, 0):
        prev = pdsp.lib.pdsp_set_real_packed(mode)
        try:
            guard_re = torch.full((batch + 2, n), 777.0, device="cuda", dtype=dt)
            guard_im = torch.full((batch + 2, n), 555.0, device="cuda", dtype=dt)
            ore, oim = guard_re[1:batch + 1], guard_im[1:batch + 1]
            plan.forward(dx, None, out=(ore, oim))
            torch.cuda.synchronize()
        finally:
            pdsp.lib.pdsp_set_real_packed(prev)
        for g, v in ((guard_re, 777.0), (guard_im, 555.0)):  # nothing written outside the batch's rows
            assert bool((g[0] == v).all()) and bool((g[-1] == v).all())
        got = ore.cpu().numpy().astype(np.float64) + 1j * oim.cpu().numpy()
        assert rel_err(got, want) <= tol, (mode, rel_err(got, want))
        assert abs(got[0, 37] - n / 2) <= tol * n and abs(got[0, n - 37] - n / 2) <= tol * n
        assert abs(got[1, 0] - n) <= tol * n and np.abs(got[1, 1:]).max() <= tol * n
        assert abs(got[2, n // 2] - n) <= tol * n and np.abs(np.delete(got[2], n // 2)).max() <= tol * n
        assert np.all(got[3] == 0)
        assert abs(got[4, n // 2 - 1] + 0.5j * n) <= max(tol, 1e-11) * n  # (the f64 sine itself is only good to ~1e-12)
        out[mode] = got
    assert rel_err(out[1], out[0]) <= (2e-6 if dtype_name == "float32" else 1e-14)
    # Hermitian symmetry of a real row's spectrum, bin for bin (the packed kernel forms X[k] and X[N-k] on
    # different threads from the same pair of values)
    herm = np.conj(out[1][:, 1:][:, ::-1])
    assert rel_err(out[1][:, 1:], herm) <= (2e-6 if dtype_name == "float32" else 1e-14)


@pytest.mark.parametrize("dtype_name", ["float32", "float64"])
def test_real_rows_in_place_and_misaligned(pdsp, oracle_mod, dtype_name):
    """re_out may be the input plane (a workgroup loads its rows before it stores); a row base that is not aligned to
    a pair of samples cannot take the 8-byte loads and falls back to the complex kernel -- same answer."""
    import torch
    from pragma_dsp_amd.batch import BatchedFft
    dt = getattr(torch, dtype_name)
    npdt = np.float32 if dtype_name == "float32" else np.float64
    tol = TOL32 if dtype_name == "float32" else TOL64
    for n, batch in ((1024, 9), (8192, 3), (16384, 9)):
        rng = np.random.default_rng(n)
        x = rng.standard_normal((batch, n)).astype(npdt)
        want = oracle_mod.Plan(n).forward(x)
        want = want[0] + 1j * want[1]
        plan = BatchedFft(n, "cuda:0", dtype=dt)
        buf = torch.from_numpy(x).cuda()
        oim = torch.empty_like(buf)
        plan.forward(buf, None, out=(buf, oim))  # in place
        torch.cuda.synchronize()
        assert rel_err(buf.cpu().numpy().astype(np.float64) + 1j * oim.cpu().numpy(), want) <= tol
        flat = torch.zeros(batch * n + 1, device="cuda", dtype=dt)
        view = flat[1:].view(batch, n)  # one sample off: rows are not pair-aligned
        view.copy_(torch.from_numpy(x).cuda())
        ure, uim = plan.forward(view)
        torch.cuda.synchronize()
        assert rel_err(ure.cpu().numpy().astype(np.float64) + 1j * uim.cpu().numpy(), want) <= tol


def test_host_dropin_forward_at_8192_runs_the_packed_kernel_on_pinned_memory(pdsp, oracle_mod):
    """FFT(8192).forward(x) through the host-f64 drop-in (default f64 arithmetic): one frame, zero-copy staging -- the
    packed kernel reads the pinned host buffer with 16-byte loads -- against the oracle at the reference's own 1e-10
    (signals.test.ts:22-23), with `out` identity and a round trip; N = 16384 (one frame: the four-step path) beside it."""
    for n in (8192, 16384):
        rng = np.random.default_rng(n)
        x = rng.standard_normal(n)
        fft = pdsp.FFT(n)
        out = fft.createComplexArray()
        got = fft.forward(x, out)
        assert got is out
        wre, wim = oracle_mod.Plan(n).forward(x)
        assert np.abs(out.real - wre).max() < 1e-10 and np.abs(out.imag - wim).max() < 1e-10
        for mode in (0, 1):  # both kernels of the switch give the drop-in's answer
            prev = pdsp.lib.pdsp_set_real_packed(mode)
            try:
                again = fft.forward(x)
            finally:
                pdsp.lib.pdsp_set_real_packed(prev)
            assert np.abs(again.real - wre).max() < 1e-10 and np.abs(again.imag - wim).max() < 1e-10
        back = fft.inverse(out)
        assert np.abs(back.real - x).max() < 1e-10 and np.abs(back.imag).max() < 1e-10

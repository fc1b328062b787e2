"""forwardComplex / inverse on interleaved (re, im) rows (complex64 / complex128 tensors) against
the f64 oracle and against the planar entry points; every single-pass size."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("log2n", list(range(0, 15)))
def test_interleaved_f32_all_sizes(oracle_mod, log2n):
    import torch
    from pragma_dsp_amd.batch import BatchedFft
    n = 1 << log2n
    batch = 13 if n <= 4096 else 3
    rng = np.random.default_rng(700 + log2n)
    z = (rng.standard_normal((batch, n)) + 1j * rng.standard_normal((batch, n))).astype(np.complex64)
    plan = BatchedFft(n, "cuda:0")
    dz = torch.from_numpy(z).cuda()
    got = plan.forward_interleaved(dz)
    wre, wim = oracle_mod.Plan(n).forward_complex(z.real, z.imag)
    assert got.dtype == torch.complex64 and rel_err(got.cpu().numpy().astype(np.complex128), wre + 1j * wim) <= 1e-5
    pre, pim = plan.forward(dz.real.contiguous(), dz.imag.contiguous())      # planar entry point, same rows
    assert rel_err(got.cpu().numpy(), pre.cpu().numpy() + 1j * pim.cpu().numpy()) <= 2e-6
    back = plan.inverse_interleaved(got)
    assert rel_err(back.cpu().numpy(), z) <= 1e-5
    bre, bim = oracle_mod.Plan(n).inverse(z.real, z.imag)                     # inverse of a non-Hermitian spectrum
    inv = plan.inverse_interleaved(dz)
    assert rel_err(inv.cpu().numpy().astype(np.complex128), bre + 1j * bim) <= 1e-5
    same = dz.clone()
    assert plan.forward_interleaved(same, out=same) is same                   # in place, row for row
    assert rel_err(same.cpu().numpy().astype(np.complex128), wre + 1j * wim) <= 1e-5


@pytest.mark.parametrize("log2n", [3, 9, 13])
def test_interleaved_f64(oracle_mod, log2n):
    import torch
    from pragma_dsp_amd.batch import BatchedFft
    n = 1 << log2n
    rng = np.random.default_rng(log2n)
    z = rng.standard_normal((4, n)) + 1j * rng.standard_normal((4, n))
    plan = BatchedFft(n, "cuda:0", dtype=torch.float64)
    got = plan.forward_interleaved(torch.from_numpy(z).cuda())
    wre, wim = oracle_mod.Plan(n).forward_complex(z.real, z.imag)
    assert rel_err(got.cpu().numpy(), wre + 1j * wim) <= 1e-14
    assert rel_err(plan.inverse_interleaved(got).cpu().numpy(), z) <= 1e-14


def test_interleaved_limits_and_errors(pdsp):
    import torch
    from pragma_dsp_amd.batch import BatchedFft
    plan = BatchedFft(1 << 15, "cuda:0")
    with pytest.raises(pdsp.PdspError, match="single-pass only"):
        plan.forward_interleaved(torch.zeros((1, 1 << 15), dtype=torch.complex64, device="cuda"))
    p8 = BatchedFft(8, "cuda:0")
    with pytest.raises(pdsp.PdspError, match="FFT input length 7 != size 8"):
        p8.forward_interleaved(torch.zeros((1, 7), dtype=torch.complex64, device="cuda"))
    with pytest.raises(pdsp.PdspError):
        p8.forward_interleaved(torch.zeros((1, 8), dtype=torch.complex128, device="cuda"))
    assert p8.forward_interleaved(torch.zeros((0, 8), dtype=torch.complex64, device="cuda")).shape == (0, 8)

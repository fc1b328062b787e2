"""Batched complex vector ops on the device (src/math/complex.ts semantics) and the FFT-domain
convolution pipeline they exist for (test/fluent/chain.test.ts:287-317)."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(1, 7), (3, 64), (5, 4096), (2, 1023)])
def test_ops_match_oracle(oracle_mod, shape):
    import torch
    from pragma_dsp_amd import batch as B
    rng = np.random.default_rng(shape[1])
    a = [rng.standard_normal(shape).astype(np.float32) for _ in range(2)]
    b = [rng.standard_normal(shape).astype(np.float32) + 2 for _ in range(2)]
    row = [rng.standard_normal(shape[1]).astype(np.float32) + 2 for _ in range(2)]
    da = tuple(torch.from_numpy(x).cuda() for x in a)
    db = tuple(torch.from_numpy(x).cuda() for x in b)
    drow = tuple(torch.from_numpy(x).cuda() for x in row)
    cases = [("add", B.complex_add, db, b), ("sub", B.complex_sub, db, b), ("mul", B.complex_mul, db, b),
             ("div", B.complex_div, db, b), ("mul", B.complex_mul, drow, row), ("div", B.complex_div, drow, row)]
    for name, fn, dev_b, host_b in cases:
        gr, gi = fn(da, dev_b)
        wr, wi = oracle_mod.complex_op(name, a[0], a[1], host_b[0], host_b[1])
        assert rel_err((gr.cpu().numpy() + 1j * gi.cpu().numpy()).ravel(), (wr + 1j * wi).ravel()) <= 2e-6, name
    for name, got, want in [
        ("conj", B.complex_conj(da), oracle_mod.complex_op("conj", a[0], a[1])),
        ("scale", B.complex_scale(da, 2.5), oracle_mod.complex_op("scale", a[0], a[1], s_re=2.5)),
        ("mulScalar", B.complex_mul_scalar(da, 0.5, -1.5), oracle_mod.complex_op("mulScalar", a[0], a[1], s_re=0.5, s_im=-1.5)),
    ]:
        assert rel_err((got[0].cpu().numpy() + 1j * got[1].cpu().numpy()).ravel(), (want[0] + 1j * want[1]).ravel()) <= 2e-6, name
    gr, gi = B.complex_div_scalar(da, 3.0, 4.0)
    wr, wi = oracle_mod.complex_op("div", a[0], a[1], [3.0], [4.0])
    assert rel_err((gr.cpu().numpy() + 1j * gi.cpu().numpy()).ravel(), (wr + 1j * wi).ravel()) <= 2e-6
    out = (torch.empty_like(da[0]), torch.empty_like(da[1]))
    assert B.complex_mul(da, db, out=out)[0] is out[0]  # `Into` form: writes into out


def test_fft_domain_convolution_stays_on_device():
    """forward -> mul -> inverse == circular convolution; two impulses convolve to an impulse
    at the sum of their positions (chain.test.ts:287-317)."""
    import torch
    from pragma_dsp_amd import batch as B
    n, rows = 1024, 16
    plan = B.BatchedFft(n, "cuda:0")
    rng = np.random.default_rng(1)
    x = torch.from_numpy(rng.standard_normal((rows, n)).astype(np.float32)).cuda()
    h = torch.zeros((1, n), device="cuda")
    h[0, 0], h[0, 1], h[0, 5] = 0.5, 0.25, -1.0
    X = plan.forward(x)
    H = plan.forward(h)
    y, yi = plan.inverse(*B.complex_mul(X, (H[0][0], H[1][0])))  # H broadcast over the rows
    want = 0.5 * x + 0.25 * torch.roll(x, 1, dims=1) - torch.roll(x, 5, dims=1)
    assert float((y - want).abs().max() / want.abs().max()) <= 1e-5 and float(yi.abs().max()) <= 1e-5
    a = torch.zeros((1, n), device="cuda")
    b = torch.zeros((1, n), device="cuda")
    a[0, 3], b[0, 7] = 1.0, 1.0
    c, _ = plan.inverse(*B.complex_mul(plan.forward(a), plan.forward(b)))
    assert int(c.argmax()) == 10 and abs(float(c[0, 10]) - 1) < 1e-5

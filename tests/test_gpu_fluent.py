"""Device-resident fluent pipeline (pragma_dsp_amd.fluent) against the cases of the reference's
test/fluent/chain.test.ts, on batches of rows, plus random-row parity with numpy f64."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu
SIG = np.array([0, 1, 0, -1, 0, 1, 0, -1], dtype=np.float32)   # chain.test.ts:27


def dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def host(t):
    return t.cpu().numpy().astype(np.float64)


def test_fluent_fft_round_trips():
    from pragma_dsp_amd.fluent import DeviceChain, FluentBatchedFft, assertNonZero
    fft = FluentBatchedFft(8)
    rows = dev(np.stack([SIG, SIG[::-1], np.arange(8)]))
    c = fft.forward(rows)                                       # :25-31
    assert isinstance(c, DeviceChain) and c.length == 8 and c.unwrap()[0].shape == (3, 8)
    re, im = fft.forward(rows).inverse()                        # :33-38
    assert np.abs(host(re) - host(rows)).max() < 1e-6 and np.abs(host(im)).max() < 1e-6
    re, _ = fft.forward(rows).conj().conj().inverse()           # :40-45
    assert np.abs(host(re) - host(rows)).max() < 1e-6
    s = 2.5
    assertNonZero(s)                                            # :47-56
    re, _ = fft.forward(rows).scale(s).inverse()
    assert np.abs(host(re) - s * host(rows)).max() < 1e-5
    cre, cim = dev([[1, 2, 3, 4]]), dev([[0.5, -1, 0, 2]])      # :58-66 forwardComplex
    f4 = FluentBatchedFft(4)
    bre, bim = f4.forwardComplex(cre, cim).inverse()
    assert np.abs(host(bre) - host(cre)).max() < 1e-6 and np.abs(host(bim) - host(cim)).max() < 1e-6
    import torch                                                # :68-75 out identity
    out = (torch.empty((3, 8), device="cuda"), torch.empty((3, 8), device="cuda"))
    got = fft.forward(rows).inverse(out)
    assert got[0] is out[0] and got[1] is out[1] and np.abs(host(out[0]) - host(rows)).max() < 1e-6
    re, _ = fft.forward(rows).scale(1.0).conj().conj().inverse()  # :278-285
    assert np.abs(host(re) - host(rows)).max() < 1e-6


def test_chain_mutates_in_place_and_clone_is_independent():
    from pragma_dsp_amd.fluent import chain
    re, im = dev([[1, 2, 3]]), dev([[4, 5, 6]])
    c = chain(re, im)                                           # :81-87
    assert c.unwrap()[0] is re and c.length == 3
    c.scale(2)                                                  # :89-101 mutates the caller's planes
    assert host(re).tolist() == [[2, 4, 6]] and host(im).tolist() == [[8, 10, 12]]
    d = c.clone().scale(10)                                     # :103-118
    assert host(re).tolist() == [[2, 4, 6]] and host(d.unwrap()[0]).tolist() == [[20, 40, 60]]
    r = chain(re, im).inverseChecked()
    assert r["ok"] is False and r["error"]["_tag"] == "NoFftContext"
    with pytest.raises(Exception, match="NoFftContext"):
        chain(re, im).inverse()


def test_fluent_ops_known_answers():
    from pragma_dsp_amd.fluent import asNonZero, assertNonZero, chain
    c = chain(dev([[1, 2]]), dev([[3, 4]])).scale(2).conj().scale(0.5)       # :121-135
    assert host(c.re).tolist() == [[1, 2]] and host(c.im).tolist() == [[-3, -4]]
    c = chain(dev([[1, 0]]), dev([[2, 1]])).mul((dev([[3, 2]]), dev([[4, 0]])))   # :137-150 (1+2i)(3+4i) = -5+10i; i*2 = 2i
    assert host(c.re).tolist() == [[-5, 0]] and host(c.im).tolist() == [[10, 2]]
    c = chain(dev([[-5, 0]]), dev([[10, 2]])).div((dev([[3, 2]]), dev([[4, 0]])))  # :152-165
    assert np.abs(host(c.re) - [[1, 0]]).max() < 1e-6 and np.abs(host(c.im) - [[2, 1]]).max() < 1e-6
    c = chain(dev([[1, 2]]), dev([[0, 1]])).mulScalar(0, 1)                  # :167-178 times i
    assert host(c.re).tolist() == [[0, -1]] and host(c.im).tolist() == [[1, 2]]
    c = chain(dev([[0, -1]]), dev([[1, 2]])).divScalar(0, 1)                 # :180-191
    assert np.abs(host(c.re) - [[1, 2]]).max() < 1e-6 and np.abs(host(c.im) - [[0, 1]]).max() < 1e-6
    b = (dev([[10, 20]]), dev([[30, 40]]))
    c = chain(dev([[1, 2]]), dev([[3, 4]])).add(b).sub(b)                    # :193-205
    assert host(c.re).tolist() == [[1, 2]] and host(c.im).tolist() == [[3, 4]]
    assert np.abs(host(chain(dev([[3, 0]]), dev([[4, 1]])).mag()) - [[5, 1]]).max() < 1e-6   # :207-213
    assert np.abs(host(chain(dev([[1, 0]]), dev([[0, 1]])).arg()) - [[0, np.pi / 2]]).max() < 1e-6  # :215-223
    assertNonZero(3)
    with pytest.raises(Exception):
        assertNonZero(0)
    assert asNonZero(2) == 2 and asNonZero(0) is None


def test_convolution_via_mul_in_the_frequency_domain():
    """chain.test.ts:287-317 (impulse * shifted impulse), then random rows of N=4096 against a
    direct circular convolution in numpy f64, the filter row broadcast over the batch."""
    from pragma_dsp_amd.fluent import FluentBatchedFft
    n = 8
    fft = FluentBatchedFft(n)
    x, h = np.zeros((1, n)), np.zeros((1, n))
    x[0, 0], h[0, 1] = 1, 1
    X, H = fft.forward(dev(x)), fft.forward(dev(h))
    r = X.mul(H.unwrap()).inverseChecked()
    assert r["ok"] is True
    y = host(r["value"][0])[0]
    assert abs(y[1] - 1) < 1e-6 and abs(y[0]) < 1e-6 and abs(y[2]) < 1e-6
    n, b = 4096, 33
    rng = np.random.default_rng(8)
    sig, flt = rng.standard_normal((b, n)), rng.standard_normal((1, n)) * np.exp(-np.arange(n) / 50.0)
    fft = FluentBatchedFft(n)
    yre, yim = fft.forward(dev(sig)).mul(fft.forward(dev(flt))).inverse()
    want = np.fft.ifft(np.fft.fft(sig, axis=-1) * np.fft.fft(flt, axis=-1), axis=-1)   # checker only
    assert rel_err(host(yre) + 1j * host(yim), want) <= 1e-5

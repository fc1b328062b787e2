"""Overlapping frames (frame_stride < frame_len): BatchedFft.stft reads one contiguous signal with
row stride = hop; every frame must equal the oracle's spectrum() body on the materialised frame."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,hop", [(8, 1), (64, 16), (256, 64), (256, 100), (1024, 256), (1024, 333), (4096, 1024),
                                   (4096, 1026),
                                   (8192, 2050), (16384, 4096), (16384, 4097), (16384, 4098), (65536, 16384)])
@pytest.mark.parametrize("window", ["rect", "hann"])
def test_stft_matches_materialised_frames(oracle_mod, n, hop, window):
    import torch
    from pragma_dsp_amd.batch import BatchedFft
    rng = np.random.default_rng(n + hop)
    frames = 37 if n <= 4096 else 6
    length = n + (frames - 1) * hop + (hop // 2)          # a tail shorter than one hop is ignored
    t = np.arange(length)
    sig = (np.sin(2 * np.pi * 0.031 * t * (1 + 2e-5 * t / max(1, n // 64))) + 0.2 * rng.standard_normal(length)).astype(np.float32)
    plan = BatchedFft(n, "cuda:0")
    amp, ph, pk = plan.stft(torch.from_numpy(sig).cuda(), hop, window, "one", want_phase=True, want_peak=True)
    assert amp.shape == (frames, n // 2 + 1)
    mat = np.stack([sig[b * hop: b * hop + n] for b in range(frames)])
    win = oracle_mod.create_window(window, n).astype(np.float32) if window != "rect" else None
    wamp, wph, wpk = oracle_mod.Plan(n).spectrum_batch(mat, window=win, want_phase=True, want_peak=True)
    a = amp.cpu().numpy().astype(np.float64)
    assert rel_err(a, wamp) <= 1e-5
    p = pk.cpu().numpy()
    for b in range(frames):
        assert abs(wamp[b, p[b]] - wamp[b, wpk[b]]) <= 2e-5 * wamp[b].max()
    amp2, _, _ = plan.stft(torch.from_numpy(sig).cuda(), hop, window, "one")     # amplitude only: the fast kernels
    assert rel_err(amp2.cpu().numpy().astype(np.float64), wamp) <= 1e-5
    amp3, _, _ = plan.stft(torch.from_numpy(sig).cuda(), hop, window, "two")
    wamp2, _, _ = oracle_mod.Plan(n).spectrum_batch(mat, window=win, two_sided=True)
    assert rel_err(amp3.cpu().numpy().astype(np.float64), wamp2) <= 1e-5


def test_stft_errors(pdsp):
    import torch
    from pragma_dsp_amd.batch import BatchedFft
    plan = BatchedFft(64, "cuda:0")
    with pytest.raises(pdsp.PdspError, match="shorter than one frame"):
        plan.stft(torch.zeros(63, device="cuda"), 16)
    with pytest.raises(pdsp.PdspError, match="hop must be >= 1"):
        plan.stft(torch.zeros(640, device="cuda"), 0)
    amp, _, _ = plan.stft(torch.zeros(64, device="cuda"), 16)
    assert amp.shape == (1, 33) and not bool(amp.any())

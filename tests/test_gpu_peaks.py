"""pdsp_spectrum_peaks_f32: findPeak fused into the spectrum kernel, vs the
oracle's findPeak (src/public/spectrum.ts:74-105) applied to the oracle's f64 spectrum."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-5


def wrap(d):
    return np.abs((d + np.pi) % (2 * np.pi) - np.pi)


@pytest.mark.parametrize("log2n", [1, 3, 5, 6, 8, 10, 11, 12, 13, 14])
@pytest.mark.parametrize("sides", ["one", "two"])
def test_peaks_match_oracle(oracle_mod, log2n, sides):
    import torch
    from pragma_dsp_amd.batch import BatchedFft
    n = 1 << log2n
    rng = np.random.default_rng(500 + log2n)
    batch = 33 if n <= 4096 else 4
    t = np.arange(n)
    x = rng.standard_normal((batch, n)) * 0.2
    for b in range(batch):  # a dominant tone per frame + noise
        x[b] += (0.5 + b % 3) * np.sin(2 * np.pi * rng.integers(1, max(2, n // 2)) * t / n + rng.random() * 6.28)
    x = x.astype(np.float32)
    window = ["rect", "hann", "hamming", "blackman"][log2n % 4]
    fs = 48000.0
    plan = BatchedFft(n, "cuda:0")
    dx = torch.from_numpy(x).cuda()
    idx, freq, amp_pk, ph_pk, amp, ph = plan.spectrum_peaks(dx, window, sides, fs, want_amp=True, want_phase=True)
    idx2, freq2, amp2, ph2, none_a, none_p = plan.spectrum_peaks(dx, window, sides, fs)  # peaks only
    torch.cuda.synchronize()
    assert none_a is None and none_p is None
    # the two calls may run different kernel variants (phase rows force the general one): same peaks,
    # values equal to rounding
    assert torch.equal(idx, idx2) and torch.equal(freq, freq2)
    assert float((amp_pk - amp2).abs().max()) <= 2e-6 * float(amp_pk.abs().max() + 1e-30)
    big = (amp_pk > 1e-3 * amp_pk.max()).to(ph_pk.dtype)  # phase of a zero peak is a signed-zero artefact
    assert float((((ph_pk - ph2 + np.pi) % (2 * np.pi) - np.pi).abs() * big).max()) <= 1e-4
    win = oracle_mod.create_window(window, n).astype(np.float32) if (window != "rect" and n > 1) else None
    wamp, wph, wpk = oracle_mod.Plan(n).spectrum_batch(x, window=win, two_sided=(sides == "two"),
                                                       want_phase=True, want_peak=True)
    idx, freq, amp_pk, ph_pk = (v.cpu().numpy() for v in (idx, freq, amp_pk, ph_pk))
    for b in range(batch):
        top = wamp[b].max()
        # same bin, or (two-sided) its mirror / a bin tying within the fp32 tolerance
        assert abs(wamp[b, idx[b]] - wamp[b, wpk[b]]) <= 2 * TOL * top, (b, idx[b], wpk[b])
        if sides == "two" and n >= 64:  # packed-real kernel: mirrored bins are bit-identical, so the
            assert idx[b] <= n // 2     # lower index wins (N < 64 runs the complex kernel: either)
        assert abs(amp_pk[b] - wamp[b, idx[b]]) <= TOL * top
        assert abs(freq[b] - idx[b] * fs / n) <= 1e-3
        if top > 0 and wamp[b, idx[b]] > 1e-3 * top:  # phase of a zero bin is a signed-zero artefact
            assert wrap(ph_pk[b] - wph[b, idx[b]]) <= 2e-3
    # the stored rows agree with the record
    a = amp.cpu().numpy()
    assert np.array_equal(a[np.arange(batch), idx], amp_pk)


def test_peak_rules_on_special_signals(reallife):
    import torch
    from pragma_dsp_amd.batch import BatchedFft
    plan = BatchedFft(1024, "cuda:0")
    names = ["zeros", "dc_level1", "dc_plus_sine_bin8", "nyquist", "sine_bin8_amp1.0", "cosine_bin8", "impulse_pos0"]
    x = torch.from_numpy(np.stack([reallife[n + "/signal"] for n in names]).astype(np.float32)).cuda()
    idx, freq, amp, ph, _, _ = plan.spectrum_peaks(x, "rect", "one", 48000.0)
    idx, freq, amp, ph = (v.cpu().numpy() for v in (idx, freq, amp, ph))
    assert idx[0] == 0 and amp[0] == 0 and ph[0] == 0 and freq[0] == 0          # zeros (edge_cases.test.ts:22-38)
    assert idx[1] == 0 and abs(amp[1] - 1) < 1e-6 and ph[1] == 0                  # pure DC stays at bin 0
    assert idx[2] == 8                                                            # DC skipped (scaling.test.ts)
    assert idx[3] == 512 and abs(amp[3] - 1) < 1e-6 and abs(freq[3] - 24000) < 1e-2
    assert idx[4] == 8 and abs(amp[4] - 1) < 1e-5 and abs(ph[4] + np.pi / 2) < 1e-4  # sine: phase -pi/2
    assert idx[5] == 8 and abs(ph[5]) < 1e-4                                      # cosine: phase 0 (phase.test.ts)
    assert idx[6] == 1                                                            # flat spectrum: first bin wins
    neg = plan.spectrum_peaks(-x[1:2], "rect", "one", 48000.0)
    assert abs(abs(float(neg[3][0])) - np.pi) < 1e-6                              # -DC: |phase| = pi


def test_peaks_on_ragged_and_padded_frames(oracle_mod):
    import torch
    from pragma_dsp_amd.batch import BatchedFft
    n = 2048
    rng = np.random.default_rng(77)
    x = rng.standard_normal((7, 1501)).astype(np.float32)  # odd length: general (clamped-load) kernel
    plan = BatchedFft(n, "cuda:0")
    idx, _, amp, _, _, _ = plan.spectrum_peaks(torch.from_numpy(x).cuda(), "hann", "one", 1.0)
    frame = np.zeros((7, n), dtype=np.float32)
    frame[:, :1501] = x
    wamp, _, wpk = oracle_mod.Plan(n).spectrum_batch(frame, window=oracle_mod.create_window("hann", n).astype(np.float32),
                                                     want_peak=True)
    idx, amp = idx.cpu().numpy(), amp.cpu().numpy()
    for b in range(7):
        assert abs(wamp[b, idx[b]] - wamp[b, wpk[b]]) <= 2 * TOL * wamp[b].max()
        assert abs(amp[b] - wamp[b, idx[b]]) <= TOL * wamp[b].max()
    with pytest.raises(Exception, match="Sample rate must be positive, got 0"):
        plan.spectrum_peaks(torch.from_numpy(x).cuda(), "hann", "one", 0)


@pytest.mark.parametrize("n", [1024, 4096, 16384])
def test_exact_ties_first_bin_wins_through_every_stage_of_the_fused_search(n):
    """findPeak's "strict >, first wins" (spectrum.ts:83-98) when EVERY bin ties exactly: an impulse at sample 0 (X[k] = 1)
    and at sample N/2 (X[k] = (-1)^k) give bit-equal amplitudes 2/N in all bins 1 .. N/2-1, so the answer must be bin 1
    -- through the per-thread runs (ascending over the forward bins, descending over the mirrored ones at N = 16384),
    their merge, the wave-wide reduction and the hop across waves; zeros -> bin 0 with amplitude 0; DC only -> bin 0;
    one bin raised by one ulp-scale step wins wherever it sits (first, last, mirrored half, a wave boundary)."""
    import torch
    from pragma_dsp_amd.batch import BatchedFft
    plan = BatchedFft(n, "cuda:0")
    x = np.zeros((4, n), dtype=np.float32)
    x[0, 0] = 1.0
    x[1, n // 2] = 1.0
    x[3, :] = 0.25
    for kwargs in ({}, {"want_amp": True}):
        idx, freq, amp, ph, rows, _ = plan.spectrum_peaks(torch.from_numpy(x).cuda(), "rect", "one", 48000.0, **kwargs)
        idx, amp = idx.cpu().numpy(), amp.cpu().numpy()
        assert idx[0] == 1 and idx[1] == 1, idx
        assert amp[0] == np.float32(2.0 / n) and amp[1] == np.float32(2.0 / n)
        assert idx[2] == 0 and amp[2] == 0 and idx[3] == 0 and abs(amp[3] - 0.25) < 1e-6
        if rows is not None:
            r = rows.cpu().numpy()
            assert np.all(r[0, 1:n // 2] == np.float32(2.0 / n)) and r[0, 0] == np.float32(1.0 / n) == r[0, n // 2]
    # an impulse plus a small cosine at bin k: X[k] = 1 + eps N / 2 exactly representable steps -> that bin must win
    for k in (1, 2, 255, 256, n // 4 - 1, n // 4, n // 4 + 1, n // 2 - 2, n // 2 - 1):
        y = np.zeros((1, n))
        y[0, 0] = 1.0
        y[0] += (2.0 ** -6) * np.cos(2 * np.pi * k * np.arange(n) / n)
        idx, _, amp, _, _, _ = plan.spectrum_peaks(torch.from_numpy(y.astype(np.float32)).cuda(), "rect", "one", 48000.0)
        assert int(idx[0]) == k, (n, k, int(idx[0]))

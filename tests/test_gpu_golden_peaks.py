"""The goldens' peak metadata through the HIP path.

tests/golden/manifest.json keeps, for each of the 35 NumPy cases of the reference's
test/reallife/references/*.json, `peakBin` / `peakMagnitude` / `peakPhase` as the reference's
generator wrote them (scripts/gen_reallife_refs.py:149-153, :503-517: argmax of the TWO-sided
|X| over bins >= 1 -- bin 0 for the pure-DC case -- and |X|, arg X there).  The reference iterates
every case against `spectrum()` (test/reallife/scaling.test.ts:27-31, :150-163, :185-201;
phase.test.ts:64-70, :92-97).  Here: the same sweep through
  (1) the drop-in `spectrum(signal, {fftSize: 1024, sampleRate: 48000})`, one-sided, in its f64 mode
      (1e-10, the reference's own tolerance) and in f32 mode (the stated fp32 tolerance), and
  (2) the device tail `pdsp_spectrum_peaks_f32` (fused findPeak, 16 B per frame).

One-sided folding: for real input bins k and N-k carry conjugate values, and NumPy's argmax lands on
whichever rounding favoured (peakBin = 1016 for two of the cases); the one-sided peak is then bin
N - peakBin with the phase negated.  `zeros`: the generator's argmax over [1:] says 1, findPeak says 0
(spectrum.ts:83-98, edge_cases.test.ts:33-37) -- the reference's own expectation is used.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N, FS = 1024, 48000


def wrap(d):
    return np.abs((np.asarray(d) + np.pi) % (2 * np.pi) - np.pi)


def expected(c, reallife):
    """(index, amplitude, phase, decisive) of the one-sided spectrum() peak, from the golden metadata.
    `decisive`: the runner-up one-sided amplitude is far enough below the peak that fp32 rounding
    cannot reorder them (otherwise only amplitude equality is asserted in f32 mode)."""
    k = int(c["peakBin"])
    mag, ph = float(c["peakMagnitude"]), float(c["peakPhase"])
    if k > N // 2:  # mirror bin of a real signal: X[N-k] = conj X[k]
        k, ph = N - k, -ph
    amp = mag / N if k in (0, N // 2) else 2.0 * mag / N  # scaleAmplitudeOneSided, spectrum.ts:49-59
    if c["name"] == "zeros":
        k, amp, ph = 0, 0.0, 0.0
    gre, gim = reallife[c["name"] + "/fftRe"], reallife[c["name"] + "/fftIm"]
    one = np.hypot(gre, gim)[: N // 2 + 1] * 2.0 / N
    one[0] /= 2
    one[-1] /= 2
    rest = np.delete(one[1:], k - 1) if k >= 1 else one[1:]
    decisive = amp == 0.0 or rest.max(initial=0.0) < amp * (1 - 1e-4)
    return k, amp, ph, decisive


@pytest.mark.parametrize("bits", [64, 32])
def test_dropin_spectrum_peak_vs_golden_metadata(pdsp, reallife, manifest, bits):
    prev = pdsp.lib.pdsp_set_host_precision(bits)
    try:
        checked = 0
        for c in manifest["reallife"]:
            k, amp, ph, decisive = expected(c, reallife)
            r = pdsp.spectrum(reallife[c["name"] + "/signal"], {"fftSize": N, "sampleRate": FS})
            name = c["name"]
            tol = (1e-10 if bits == 64 else 1e-5) * max(1.0, amp)  # large_amplitude: relative (edge_cases.test.ts:166-175)
            if name == "tiny_amplitude":
                tol = 1e-20 if bits == 64 else 1e-5 * amp
            if decisive or bits == 64:
                if name in ("impulse_pos0", "impulse_pos512") and bits == 32:
                    pass  # flat |X| = 1: every bin ties; f64 keeps bin 1 exactly, f32 is held to the amplitude only
                else:
                    assert r.peak.index == k, (name, r.peak.index, k)
                    assert r.peak.frequency == k * FS / N, name
            assert abs(r.peak.amplitude - amp) <= tol, (name, r.peak.amplitude, amp)
            assert r.peak.amplitude == r.amplitude[r.peak.index] and r.peak.phase == r.phase[r.peak.index], name
            if float(c["peakMagnitude"]) > 1e-6 and r.peak.index == k:  # phase is meaningful only above the noise floor
                assert wrap(r.peak.phase - ph) < (1e-10 if bits == 64 else 2e-5), (name, r.peak.phase, ph)
            checked += 1
        assert checked == 35
    finally:
        pdsp.lib.pdsp_set_host_precision(prev)


@pytest.mark.parametrize("want_rows", [False, True])
def test_device_peaks_kernel_vs_golden_metadata(reallife, manifest, want_rows):
    """pdsp_spectrum_peaks_f32 on the 35 signals as ONE device batch (fused findPeak; with and without
    amplitude rows)."""
    import torch
    from pragma_dsp_amd.batch import BatchedFft
    cases = manifest["reallife"]
    x = np.stack([reallife[c["name"] + "/signal"] for c in cases]).astype(np.float32)
    plan = BatchedFft(N, "cuda:0")
    idx, freq, amp, ph, arow, _ = plan.spectrum_peaks(torch.from_numpy(x).cuda(), "rect", "one", float(FS), want_amp=want_rows)
    torch.cuda.synchronize()
    idx, freq, amp, ph = (t.cpu().numpy() for t in (idx, freq, amp, ph))
    for i, c in enumerate(cases):
        k, a, p, decisive = expected(c, reallife)
        name = c["name"]
        # the f32 signal itself differs from the f64 golden input by 6e-8 relative: tolerance 1e-5 of the peak
        assert abs(float(amp[i]) - a) <= 1e-5 * max(a, 1e-30) + (0 if a else 0.0), (name, amp[i], a)
        if decisive and not name.startswith("impulse"):
            assert int(idx[i]) == k, (name, idx[i], k)
            assert abs(float(freq[i]) - k * FS / N) <= 1e-3, name
            if float(c["peakMagnitude"]) > 1e-6:
                assert wrap(float(ph[i]) - p) < 2e-5, (name, ph[i], p)
        if want_rows:
            assert float(arow[i, int(idx[i])]) == float(amp[i]), name  # the record is the stored row's value

"""The reference's own hot-path test suite, case for case, through the HIP drop-in.

Every `it(...)` of pragma-dsp's test/fft.test.ts, test/spectrum.test.ts, test/window.test.ts and
test/reallife/{signals, phase, scaling, edge_cases}.test.ts has ONE test function here, named after the reference's
description and citing its line, run on the same fixtures (tests/golden/*.npz: the reference's NumPy / SciPy goldens
and the v0.1 fixture its own generator regenerates) with the same assertions and the SAME tolerances -- in the
drop-in's default f64 mode.  `toBeCloseTo(x, d)` is vitest's |a - x| < 0.5 * 10^-d.  The f32 mode of the host path
(the north-star's arithmetic) runs the same cases at the stated fp32 tolerance max|err| / max|X| <= 1e-5
(`test_every_case_in_f32_mode_at_the_stated_tolerance`).  test/reallife/effect.test.ts (the Effect-TS wrapper) and
test/fluent, test/math are outside the hot path."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PI = np.pi


def close_to(a, x, digits):  # vitest toBeCloseTo
    return abs(a - x) < 0.5 * 10.0 ** (-digits)


def wrapped(d):
    d = np.abs(d)
    return np.minimum(d, np.abs(d - 2 * PI))


@pytest.fixture
def f64_mode(pdsp):
    prev = pdsp.lib.pdsp_set_host_precision(64)
    yield
    pdsp.lib.pdsp_set_host_precision(prev)


class Case:
    def __init__(self, meta, arrays):
        self.name, self.kind, self.n, self.fs, self.params = meta["name"], meta["kind"], meta["n"], meta["sampleRate"], meta["params"]
        self.signal = arrays[self.name + "/signal"]
        self.re, self.im = arrays[self.name + "/fftRe"], arrays[self.name + "/fftIm"]
        self.magnitude, self.phase = np.hypot(self.re, self.im), np.arctan2(self.im, self.re)  # as gen_reallife_refs.py stores them


@pytest.fixture(scope="module")
def cases(reallife, manifest):
    return [Case(m, reallife) for m in manifest["reallife"]]


def family(cases, fam, manifest):
    names = {m["name"] for m in manifest["reallife"] if m["family"] == fam}
    return [c for c in cases if c.name in names]


def one(cases, pred):
    hits = [c for c in cases if pred(c)]
    assert hits, "fixture case missing"
    return hits[0]


OPTS = lambda c, sides="one": {"sampleRate": c.fs, "fftSize": c.n, "window": "rect", "sides": sides}  # noqa: E731


# ---- test/fft.test.ts -------------------------------------------------------------------------------

def test_fft_fixtures_matches_numpy_fft_for_each_random_normal_case(pdsp, f64_mode, v01, manifest):  # fft.test.ts:27
    n_cases = 0
    for m in manifest["v01_cases"]:
        if m["kind"] != "random_normal" or m["n"] not in (8, 16, 32):
            continue
        x = v01[f"case/{m['name']}/input"]
        out = pdsp.FFT(m["n"]).forward(x)
        assert np.abs(out.real - v01[f"case/{m['name']}/fftRe"]).max() <= 1e-6
        assert np.abs(out.imag - v01[f"case/{m['name']}/fftIm"]).max() <= 1e-6
        n_cases += 1
    assert n_cases == 15


def test_fft_fixtures_round_trips_each_random_normal_case(pdsp, f64_mode, v01, manifest):  # fft.test.ts:34
    for m in manifest["v01_cases"]:
        if m["kind"] != "random_normal" or m["n"] not in (8, 16, 32):
            continue
        x = v01[f"case/{m['name']}/input"]
        fft = pdsp.FFT(m["n"])
        back = fft.inverse(fft.forward(x))
        assert np.abs(back.real - x).max() <= 1e-6 and np.abs(back.imag).max() <= 1e-6


# ---- test/spectrum.test.ts --------------------------------------------------------------------------

def test_spectrum_returns_correct_peak_bin_frequency_amplitude(pdsp, f64_mode, v01, manifest):  # spectrum.test.ts:15
    m = next(c for c in manifest["v01_cases"] if c["kind"] == "sine_bin_centered")
    res = pdsp.spectrum(v01[f"case/{m['name']}/input"], {"sampleRate": m["sampleRate"], "fftSize": m["n"], "window": "rect", "sides": "one"})
    assert res.peak.index == m["meta"]["binCenteredK"]
    assert abs(res.peak.frequency - m["meta"]["expectedPeakHz"]) <= 1e-6
    assert abs(res.peak.amplitude - m["meta"]["amplitude"]) <= 1e-3


# ---- test/window.test.ts ----------------------------------------------------------------------------

def test_window_fixtures_matches_each_window(pdsp, v01, manifest):  # window.test.ts:22
    assert len(manifest["v01_windows"]) == 28
    for w in manifest["v01_windows"]:
        created = pdsp.createWindow(w["type"], w["n"])
        want = v01[w["key"]]
        assert len(created) == len(want) and np.abs(created - want).max() <= 1e-8, w["key"]


# ---- test/reallife/signals.test.ts ------------------------------------------------------------------

def test_signals_pure_sine_matches_numpy_fft(pdsp, f64_mode, cases, manifest):  # signals.test.ts:17
    fam = family(cases, "pure_sine", manifest)
    assert len(fam) == 23
    for c in fam:
        out = pdsp.FFT(c.n).forward(c.signal)
        assert np.abs(out.real - c.re).max() < 1e-10 and np.abs(out.imag - c.im).max() < 1e-10, c.name


def test_signals_pure_sine_magnitude_matches_numpy(pdsp, f64_mode, cases, manifest):  # signals.test.ts:26
    for c in family(cases, "pure_sine", manifest):
        mag = pdsp.magnitude(pdsp.FFT(c.n).forward(c.signal))
        assert np.abs(mag - c.magnitude).max() < 1e-10, c.name


def test_signals_pure_sine_phase_matches_numpy(pdsp, f64_mode, cases, manifest):  # signals.test.ts:34
    for c in family(cases, "pure_sine", manifest):
        ph = pdsp.phase(pdsp.FFT(c.n).forward(c.signal))
        mask = c.magnitude > 1e-6
        assert wrapped(ph[mask] - c.phase[mask]).max(initial=0) < 1e-10, c.name


def test_signals_pure_sine_round_trips_correctly(pdsp, f64_mode, cases, manifest):  # signals.test.ts:51
    for c in family(cases, "pure_sine", manifest):
        fft = pdsp.FFT(c.n)
        back = fft.inverse(fft.forward(c.signal))
        assert np.abs(back.real - c.signal).max() < 1e-10 and np.abs(back.imag).max() < 1e-10, c.name


def test_signals_multi_tone_matches_numpy_fft(pdsp, f64_mode, cases, manifest):  # signals.test.ts:71
    fam = family(cases, "multi_tone", manifest)
    assert len(fam) == 2
    for c in fam:
        out = pdsp.FFT(c.n).forward(c.signal)
        assert np.abs(out.real - c.re).max() < 1e-10 and np.abs(out.imag - c.im).max() < 1e-10, c.name


def test_signals_multi_tone_detects_correct_peaks(pdsp, f64_mode, cases, manifest):  # signals.test.ts:79
    for c in family(cases, "multi_tone", manifest):
        mag = pdsp.magnitude(pdsp.FFT(c.n).forward(c.signal))
        for b, a in zip(c.params["bin_indices"], c.params["amplitudes"]):
            assert close_to(mag[b], c.n * a / 2, 5), (c.name, b)


def test_signals_chirp_matches_numpy_fft(pdsp, f64_mode, cases, manifest):  # signals.test.ts:104
    fam = family(cases, "chirp", manifest)
    assert len(fam) == 1
    for c in fam:
        out = pdsp.FFT(c.n).forward(c.signal)
        assert np.abs(out.real - c.re).max() < 1e-10 and np.abs(out.imag - c.im).max() < 1e-10


def test_signals_chirp_round_trips_correctly(pdsp, f64_mode, cases, manifest):  # signals.test.ts:112
    for c in family(cases, "chirp", manifest):
        fft = pdsp.FFT(c.n)
        assert np.abs(fft.inverse(fft.forward(c.signal)).real - c.signal).max() < 1e-10


def test_signals_impulse_has_flat_magnitude_spectrum(pdsp, f64_mode, cases):  # signals.test.ts:127
    c = one(cases, lambda c: c.kind == "impulse")  # the first impulse case: position 0
    assert c.params["position"] == 0
    mag = pdsp.magnitude(pdsp.FFT(c.n).forward(c.signal))
    assert np.abs(mag - c.params["amplitude"]).max() < 0.5e-10


def test_signals_dc_signal_has_energy_only_in_bin_0(pdsp, f64_mode, cases):  # signals.test.ts:143
    c = one(cases, lambda c: c.kind == "dc")
    mag = pdsp.magnitude(pdsp.FFT(c.n).forward(c.signal))
    assert close_to(mag[0], c.n * c.params["level"], 10) and mag[1:].max() < 1e-10


def test_signals_nyquist_signal_has_energy_only_at_nyquist_bin(pdsp, f64_mode, cases):  # signals.test.ts:163
    c = one(cases, lambda c: c.kind == "nyquist")
    mag = pdsp.magnitude(pdsp.FFT(c.n).forward(c.signal))
    nb = c.n // 2
    assert close_to(mag[nb], c.n * c.params["amplitude"], 10) and np.delete(mag, nb).max() < 1e-10


def test_signals_zero_input_gives_zero_output(pdsp, f64_mode, cases):  # signals.test.ts:186
    c = one(cases, lambda c: c.kind == "zeros")
    out = pdsp.FFT(c.n).forward(c.signal)
    assert np.all(out.real == 0) and np.all(out.imag == 0)  # toBe(0)


# ---- test/reallife/phase.test.ts --------------------------------------------------------------------

def test_phase_cosine_leads_sine_by_90_degrees_at_peak_bin(pdsp, f64_mode, cases):  # phase.test.ts:20
    s = one(cases, lambda c: c.kind == "pure_sine_bin_centered" and c.params["bin_index"] == 8)
    k = one(cases, lambda c: c.kind == "cosine" and c.params["bin_index"] == 8)
    fft = pdsp.FFT(s.n)
    d = pdsp.phase(fft.forward(k.signal))[8] - pdsp.phase(fft.forward(s.signal))[8]
    while d > PI:
        d -= 2 * PI
    while d < -PI:
        d += 2 * PI
    assert abs(d - PI / 2) < 1e-6


def test_phase_is_correct_for_each_known_phase_signal(pdsp, f64_mode, cases):  # phase.test.ts:51
    fam = [c for c in cases if c.kind == "pure_sine_phase"]
    assert len(fam) >= 4
    for c in fam:
        b = c.params["bin_index"]
        ph = pdsp.phase(pdsp.FFT(c.n).forward(c.signal))
        assert wrapped(ph[b] - c.phase[b]) < 1e-10, c.name


def test_phase_spectrum_reports_correct_peak_phase(pdsp, f64_mode, cases):  # phase.test.ts:81
    for c in [c for c in cases if c.kind == "pure_sine_phase"]:
        res = pdsp.spectrum(c.signal, OPTS(c))
        b = c.params["bin_index"]
        assert res.peak.index == b and wrapped(res.peak.phase - c.phase[b]) < 1e-10, c.name


def test_phase_array_has_correct_length_for_one_sided_spectrum(pdsp, f64_mode, cases):  # phase.test.ts:103
    c = cases[0]
    assert len(pdsp.spectrum(c.signal, OPTS(c)).phase) == c.n // 2 + 1


def test_phase_array_has_correct_length_for_two_sided_spectrum(pdsp, f64_mode, cases):  # phase.test.ts:119
    c = cases[0]
    assert len(pdsp.spectrum(c.signal, OPTS(c, "two")).phase) == c.n


def test_phase_dc_phase_is_0_for_positive_dc_signal(pdsp, f64_mode):  # phase.test.ts:137
    assert close_to(pdsp.phase(pdsp.FFT(64).forward(np.full(64, 1.0)))[0], 0, 10)


def test_phase_dc_phase_is_pi_for_negative_dc_signal(pdsp, f64_mode):  # phase.test.ts:149
    assert close_to(abs(pdsp.phase(pdsp.FFT(64).forward(np.full(64, -1.0)))[0]), PI, 10)


# ---- test/reallife/scaling.test.ts ------------------------------------------------------------------

def centered(cases):
    fam = [c for c in cases if c.kind == "pure_sine_bin_centered"]
    assert len(fam) >= 6
    return fam


def test_scaling_returns_correct_amplitude_one_sided(pdsp, f64_mode, cases):  # scaling.test.ts:15
    for c in centered(cases):
        res = pdsp.spectrum(c.signal, OPTS(c))
        assert res.peak.index == c.params["bin_index"] and close_to(res.peak.amplitude, c.params["amplitude"], 2), c.name


def test_scaling_dc_bin_is_not_doubled(pdsp, f64_mode, cases):  # scaling.test.ts:35
    c = one(cases, lambda c: c.kind == "dc")
    assert close_to(pdsp.spectrum(c.signal, OPTS(c)).amplitude[0], c.params["level"], 6)


def test_scaling_nyquist_bin_is_not_doubled(pdsp, f64_mode, cases):  # scaling.test.ts:51
    c = one(cases, lambda c: c.kind == "nyquist")
    assert close_to(pdsp.spectrum(c.signal, OPTS(c)).amplitude[c.n // 2], c.params["amplitude"], 6)


def test_scaling_returns_correct_amplitude_two_sided(pdsp, f64_mode, cases):  # scaling.test.ts:81
    for c in centered(cases):
        amp = pdsp.spectrum(c.signal, OPTS(c, "two")).amplitude
        b, a = c.params["bin_index"], c.params["amplitude"]
        assert close_to(amp[b], a / 2, 2) and close_to(amp[c.n - b], a / 2, 2), c.name


def test_scaling_returns_full_n_bins_for_two_sided(pdsp, f64_mode, cases):  # scaling.test.ts:102
    c = cases[0]
    res = pdsp.spectrum(c.signal, OPTS(c, "two"))
    assert len(res.amplitude) == c.n and len(res.frequencies) == c.n and len(res.phase) == c.n


def test_scaling_peak_frequency_matches_expected(pdsp, f64_mode, cases):  # scaling.test.ts:125
    for c in centered(cases):
        assert close_to(pdsp.spectrum(c.signal, OPTS(c)).peak.frequency, c.params["frequency_hz"], 6), c.name


def test_scaling_frequency_axis_is_correctly_scaled(pdsp, f64_mode, cases):  # scaling.test.ts:138
    c = cases[0]
    f = pdsp.spectrum(c.signal, OPTS(c)).frequencies
    assert f[0] == 0
    assert np.abs(f - np.arange(len(f)) * (c.fs / c.n)).max() < 0.5e-10 and close_to(f[-1], c.fs / 2, 10)


def test_scaling_peak_ignores_dc_when_there_are_non_dc_components(pdsp, f64_mode, cases):  # scaling.test.ts:168
    c = one(cases, lambda c: c.kind == "dc_plus_sine")
    assert pdsp.spectrum(c.signal, OPTS(c)).peak.index == c.params["sine_bin"]


def test_scaling_peak_returns_dc_when_it_is_the_only_component(pdsp, f64_mode, cases):  # scaling.test.ts:185
    c = one(cases, lambda c: c.kind == "dc")
    assert pdsp.spectrum(c.signal, OPTS(c)).peak.index == 0


# ---- test/reallife/edge_cases.test.ts ---------------------------------------------------------------

def test_edge_fft_of_zeros_gives_zeros(pdsp, f64_mode, cases):  # edge_cases.test.ts:8
    c = one(cases, lambda c: c.kind == "zeros")
    out = pdsp.FFT(c.n).forward(c.signal)
    assert np.all(out.real == 0) and np.all(out.imag == 0)


def test_edge_spectrum_of_zeros_gives_zeros(pdsp, f64_mode):  # edge_cases.test.ts:22
    res = pdsp.spectrum(np.zeros(64), {"sampleRate": 48000, "fftSize": 64, "window": "rect", "sides": "one"})
    assert np.all(res.amplitude == 0) and res.peak.amplitude == 0


def test_edge_dc_signal_has_energy_only_in_bin_0(pdsp, f64_mode, cases):  # edge_cases.test.ts:42
    c = one(cases, lambda c: c.kind == "dc")
    mag = pdsp.magnitude(pdsp.FFT(c.n).forward(c.signal))
    assert close_to(mag[0], c.n * c.params["level"], 10) and mag[1:].max() < 1e-10


def test_edge_alternating_plus_minus_one_has_energy_only_at_nyquist(pdsp, f64_mode, cases):  # edge_cases.test.ts:65
    c = one(cases, lambda c: c.kind == "nyquist")
    mag = pdsp.magnitude(pdsp.FFT(c.n).forward(c.signal))
    assert close_to(mag[c.n // 2], c.n * c.params["amplitude"], 10) and np.delete(mag, c.n // 2).max() < 1e-10


def test_edge_impulse_at_position_0_gives_flat_magnitude_spectrum(pdsp, f64_mode, cases):  # edge_cases.test.ts:92
    c = one(cases, lambda c: c.kind == "impulse" and c.params["position"] == 0)
    assert np.abs(pdsp.magnitude(pdsp.FFT(c.n).forward(c.signal)) - c.params["amplitude"]).max() < 0.5e-10


def test_edge_impulse_at_middle_position_gives_correct_phase_pattern(pdsp, f64_mode, cases):  # edge_cases.test.ts:110
    c = one(cases, lambda c: c.kind == "impulse" and c.params["position"] > 0)
    assert np.abs(pdsp.magnitude(pdsp.FFT(c.n).forward(c.signal)) - c.params["amplitude"]).max() < 0.5e-10


def test_edge_handles_tiny_amplitude_signals_without_underflow(pdsp, f64_mode, cases):  # edge_cases.test.ts:130
    c = one(cases, lambda c: c.kind == "tiny")
    out = pdsp.FFT(c.n).forward(c.signal)
    assert np.all(np.isfinite(out.real)) and np.all(np.isfinite(out.imag))
    assert np.abs(out.real - c.re).max() < 1e-20


def test_edge_handles_large_amplitude_signals_without_overflow(pdsp, f64_mode, cases):  # edge_cases.test.ts:151
    c = one(cases, lambda c: c.kind == "large")
    out = pdsp.FFT(c.n).forward(c.signal)
    assert np.all(np.isfinite(out.real)) and np.all(np.isfinite(out.imag))
    big = np.abs(c.re) > 1
    assert (np.abs(out.real[big] - c.re[big]) / np.abs(c.re[big])).max(initial=0) < 1e-9
    assert np.abs(out.real[~big] - c.re[~big]).max(initial=0) < 1e-6


def test_edge_handles_input_shorter_than_fft_size(pdsp, f64_mode):  # edge_cases.test.ts:180
    res = pdsp.spectrum(np.array([1.0, 2, 3, 4]), {"sampleRate": 48000, "fftSize": 16, "window": "rect", "sides": "one"})
    assert len(res.amplitude) == 16 // 2 + 1 and np.isfinite(res.peak.amplitude) and np.isfinite(res.peak.frequency)


def test_edge_zero_padding_preserves_signal_content(pdsp, f64_mode):  # edge_cases.test.ts:199
    res = pdsp.spectrum(np.ones(4), {"sampleRate": 48000, "fftSize": 16, "window": "rect", "sides": "one"})
    assert close_to(res.amplitude[0], 4 / 16, 6)


def test_edge_ifft_of_fft_is_identity_for_all_special_signals(pdsp, f64_mode, cases, manifest):  # edge_cases.test.ts:217
    fam = family(cases, "special", manifest)
    assert len(fam) == 8
    for c in fam:
        fft = pdsp.FFT(c.n)
        back = fft.inverse(fft.forward(c.signal))
        assert np.abs(back.real - c.signal).max() < 1e-9 and np.abs(back.imag).max() < 1e-9, c.name


# ---- the same fixtures in f32 mode ------------------------------------------------------------------

def test_every_case_in_f32_mode_at_the_stated_tolerance(pdsp, cases):
    """pdsp_set_host_precision(32): the north-star's arithmetic.  Per case max|err| / max|X| <= 1e-5 against the
    NumPy goldens (transform, magnitude, round trip) and the exact-zero / peak-index assertions unchanged."""
    prev = pdsp.lib.pdsp_set_host_precision(32)
    try:
        for c in cases:
            fft = pdsp.FFT(c.n)
            out = fft.forward(c.signal)
            top = max(np.abs(c.re + 1j * c.im).max(), 1e-300)
            if c.kind == "zeros":
                assert np.all(out.real == 0) and np.all(out.imag == 0)
                continue
            assert max(np.abs(out.real - c.re).max(), np.abs(out.imag - c.im).max()) / top <= 1e-5, c.name
            assert np.abs(pdsp.magnitude(out) - c.magnitude).max() / top <= 1e-5, c.name
            back = fft.inverse(out)
            assert np.abs(back.real - c.signal).max() <= 1e-5 * max(np.abs(c.signal).max(), 1e-300), c.name
            if c.kind in ("pure_sine_bin_centered", "pure_sine_phase", "cosine", "dc", "dc_plus_sine", "nyquist"):
                want = {"dc": 0, "nyquist": c.n // 2, "dc_plus_sine": c.params.get("sine_bin")}.get(c.kind, c.params.get("bin_index"))
                assert pdsp.spectrum(c.signal, OPTS(c)).peak.index == want, c.name
    finally:
        pdsp.lib.pdsp_set_host_precision(prev)

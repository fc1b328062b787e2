"""Pins the CPU oracle (oracle/pdsp_oracle.c) on every golden vector, known-answer
test and fixture the reference holds for the hot path, at the
reference's own tolerances.  No GPU, no product code."""
import numpy as np
import pytest

TWO_PI = 2 * np.pi


def wrap(d):
    """phase difference modulo 2*pi (test/reallife/signals.test.ts:41-47)."""
    return np.abs((d + np.pi) % TWO_PI - np.pi)


# ---- test/reallife/signals.test.ts ------------------------------------------

def test_fft_matches_numpy_goldens_1e10(oracle_mod, reallife, manifest):
    plan = oracle_mod.Plan(1024)
    assert len(manifest["reallife"]) == 35
    for c in manifest["reallife"]:
        name = c["name"]
        re, im = plan.forward(reallife[name + "/signal"])
        gre, gim = reallife[name + "/fftRe"], reallife[name + "/fftIm"]
        if name == "large_amplitude":  # edge_cases.test.ts:166-175: relative 1e-9
            scale = max(np.abs(gre).max(), np.abs(gim).max())
            assert max(np.abs(re - gre).max(), np.abs(im - gim).max()) / scale < 1e-9
        elif name == "tiny_amplitude":  # edge_cases.test.ts:145-146: abs < 1e-20
            assert np.all(np.isfinite(re)) and max(np.abs(re - gre).max(), np.abs(im - gim).max()) < 1e-20
        else:  # signals.test.ts:22-23
            assert np.abs(re - gre).max() < 1e-10, name
            assert np.abs(im - gim).max() < 1e-10, name


def test_magnitude_and_phase_match_goldens(oracle_mod, reallife, manifest):
    plan = oracle_mod.Plan(1024)
    for c in manifest["reallife"]:
        name = c["name"]
        if name in ("large_amplitude",):
            continue
        re, im = plan.forward(reallife[name + "/signal"])
        gre, gim = reallife[name + "/fftRe"], reallife[name + "/fftIm"]
        gmag = np.sqrt(gre ** 2 + gim ** 2)  # scripts/gen_reallife_refs.py:139-141
        gph = np.arctan2(gim, gre)            # :144-146
        assert np.abs(oracle_mod.magnitude(re, im) - gmag).max() < 1e-10, name
        mask = gmag > 1e-6                    # signals.test.ts:41-47
        assert wrap(oracle_mod.phase(re, im)[mask] - gph[mask]).max(initial=0) < 1e-10, name
        # peak metadata the JSON carries
        assert abs(gmag[c["peakBin"]] - c["peakMagnitude"]) < 1e-9 * max(1, c["peakMagnitude"])


def test_round_trip_and_special_signals(oracle_mod, reallife, manifest):
    plan = oracle_mod.Plan(1024)
    for c in manifest["reallife"]:
        x = reallife[c["name"] + "/signal"]
        re, im = plan.forward(x)
        bre, bim = plan.inverse(re, im)
        scale = max(1.0, np.abs(x).max())
        assert np.abs(bre - x).max() / scale < 1e-9 and np.abs(bim).max() / scale < 1e-9
    re, im = plan.forward(reallife["impulse_pos0/signal"])       # flat |X| = 1
    assert np.abs(oracle_mod.magnitude(re, im) - 1).max() < 1e-10
    re, im = plan.forward(reallife["dc_level1/signal"])          # DC -> bin 0 = N * level
    assert abs(re[0] - 1024) < 1e-10 and np.abs(re[1:]).max() < 1e-10 and np.abs(im).max() < 1e-10
    re, im = plan.forward(reallife["nyquist/signal"])
    assert np.argmax(oracle_mod.magnitude(re, im)) == 512
    re, im = plan.forward(reallife["zeros/signal"])              # exact zeros (toBe(0))
    assert np.all(re == 0) and np.all(im == 0)


# ---- test/window.test.ts + windows_dsp.json ----------------------------------

def test_windows_match_scipy_goldens(oracle_mod, windows_dsp, v01, manifest):
    assert len(manifest["windows_dsp"]) == 16 and len(manifest["v01_windows"]) == 28
    for w in manifest["windows_dsp"]:
        vals = oracle_mod.create_window(w["type"], w["n"])
        assert np.abs(vals - windows_dsp[w["key"]]).max() < 1e-8  # window.test.ts:8
        assert abs(vals.mean() - w["coherentGain"]) < 1e-12
        assert abs(w["n"] * np.sum(vals ** 2) / np.sum(vals) ** 2 - w["enbw"]) < 1e-10
    for w in manifest["v01_windows"]:
        vals = oracle_mod.create_window(w["type"], w["n"])
        assert np.abs(vals - v01[w["key"]]).max() < 1e-8


def test_window_edges(oracle_mod):
    assert np.array_equal(oracle_mod.create_window("blackman", 1), [1.0])
    with pytest.raises(ValueError, match="Window size must be positive, got 0"):
        oracle_mod.create_window("hann", 0)
    with pytest.raises(ValueError, match="Unsupported window type: kaiser"):
        oracle_mod.create_window("kaiser", 8)
    with pytest.raises(ValueError, match="Window length must match input length."):
        oracle_mod.apply_window([1, 2, 3], [1, 2])


# ---- test/fft.test.ts + test/spectrum.test.ts (regenerated v0.1 fixture) --------

def test_v01_fixture_cases(oracle_mod, v01, manifest):
    assert abs(v01["case/rand_n8_0/input"][0] - 0.038268222830415852) < 1e-18  # seed-1337 determinism
    names = [c["name"] for c in manifest["v01_cases"]]
    assert sum(n.startswith("rand_n") for n in names) == 15
    for c in manifest["v01_cases"]:
        x = v01[f"case/{c['name']}/input"]
        plan = oracle_mod.Plan(c["n"])
        re, im = plan.forward(x)
        assert np.abs(re - v01[f"case/{c['name']}/fftRe"]).max() < 1e-6  # fft.test.ts:8
        assert np.abs(im - v01[f"case/{c['name']}/fftIm"]).max() < 1e-6
        assert np.abs(re - v01[f"case/{c['name']}/fftRe"]).max() < 1e-10 * max(1, c["n"] / 64)
        bre, bim = plan.inverse(re, im)
        assert np.abs(bre - x).max() < 1e-6 and np.abs(bim).max() < 1e-6


def test_spectrum_sine_bin_centered(oracle_mod, v01, manifest):
    (c,) = [c for c in manifest["v01_cases"] if c["kind"] == "sine_bin_centered"]
    r = oracle_mod.spectrum(v01[f"case/{c['name']}/input"], sample_rate=c["sampleRate"], fft_size=c["n"])
    assert r["peak"]["index"] == c["meta"]["binCenteredK"] == 32       # spectrum.test.ts:15-33
    assert abs(r["peak"]["frequency"] - c["meta"]["expectedPeakHz"]) <= 1e-6
    assert abs(r["peak"]["amplitude"] - c["meta"]["amplitude"]) <= 1e-3
    assert abs(r["peak"]["phase"] + np.pi / 2) < 1e-9


# ---- inline known answers ----------------------------------------

def test_spectrum_known_answers(oracle_mod):
    r = oracle_mod.spectrum([0, 1, 0, -1, 0, 1, 0, -1], sample_rate=48000)  # README.md:11, BASELINE config 1
    assert len(r["amplitude"]) == 5
    assert np.allclose(r["amplitude"], [0, 0, 1, 0, 0], atol=1e-15)
    assert r["peak"]["index"] == 2 and r["peak"]["frequency"] == 12000
    assert abs(r["peak"]["amplitude"] - 1) < 1e-15 and abs(r["peak"]["phase"] + np.pi / 2) < 1e-15
    r = oracle_mod.spectrum([1, 1, 1, 1], sample_rate=48000, fft_size=16)    # edge_cases.test.ts:199-213
    assert abs(r["amplitude"][0] - 0.25) < 1e-15
    r = oracle_mod.spectrum([1, 2, 3, 4], sample_rate=48000, fft_size=16)    # edge_cases.test.ts:180-197
    assert len(r["amplitude"]) == 9 and np.all(np.isfinite(r["amplitude"]))
    assert abs(r["amplitude"][0] - 0.625) < 1e-15 and r["peak"]["index"] == 1
    assert abs(r["peak"]["frequency"] - 3000) < 1e-9 and abs(r["peak"]["amplitude"] - 1.156321) < 1e-5
    r = oracle_mod.spectrum(np.zeros(64), sample_rate=48000)                  # edge_cases.test.ts:22-38
    assert np.all(r["amplitude"] == 0) and r["peak"]["amplitude"] == 0 and r["peak"]["index"] == 0
    re, im = oracle_mod.Plan(64).forward(np.ones(64))                         # phase.test.ts:137-160
    assert oracle_mod.phase(re, im)[0] == 0
    re, im = oracle_mod.Plan(64).forward(-np.ones(64))
    assert abs(abs(oracle_mod.phase(re, im)[0]) - np.pi) < 1e-15


def test_spectrum_scaling_rules(oracle_mod, reallife):
    x = reallife["sine_bin8_amp1.0/signal"]                                   # scaling.test.ts
    one = oracle_mod.spectrum(x, sample_rate=48000, fft_size=1024)
    assert one["peak"]["index"] == 8 and abs(one["peak"]["amplitude"] - 1) < 5e-3
    assert len(one["amplitude"]) == 513 and len(one["phase"]) == 513
    two = oracle_mod.spectrum(x, sample_rate=48000, fft_size=1024, sides="two")
    assert len(two["amplitude"]) == 1024 and len(two["phase"]) == 1024
    assert abs(two["amplitude"][8] - 0.5) < 5e-3 and abs(two["amplitude"][1016] - 0.5) < 5e-3
    assert np.allclose(one["frequencies"], np.arange(513) * 48000 / 1024, atol=1e-10)
    dc = oracle_mod.spectrum(reallife["dc_level1/signal"], sample_rate=48000, fft_size=1024)
    assert dc["peak"]["index"] == 0 and abs(dc["amplitude"][0] - 1) < 1e-6      # DC not doubled
    ny = oracle_mod.spectrum(reallife["nyquist/signal"], sample_rate=48000, fft_size=1024)
    assert ny["peak"]["index"] == 512 and abs(ny["amplitude"][512] - 1) < 1e-6  # Nyquist not doubled
    mix = oracle_mod.spectrum(reallife["dc_plus_sine_bin8/signal"], sample_rate=48000, fft_size=1024)
    assert mix["peak"]["index"] == 8                                           # DC skipped
    # truncation (spectrum.ts:38) and window over the padded length (:116-119)
    long_x = np.concatenate([x, 7 * np.ones(100)])
    assert np.array_equal(oracle_mod.spectrum(long_x, fft_size=1024)["amplitude"], oracle_mod.spectrum(x, fft_size=1024)["amplitude"])


def test_find_peak_rules(oracle_mod):
    assert oracle_mod.find_peak([5, 1, 3, 3, 2]) == 2      # DC ignored, first of the tie wins
    assert oracle_mod.find_peak([5, 0, 0, 0]) == 0         # nothing > 0 beyond DC
    assert oracle_mod.find_peak([0, 0, 0]) == 0
    assert oracle_mod.find_peak([1]) == 0
    assert oracle_mod.find_peak([0, 2, 1, 2]) == 1         # two-sided mirror tie: k wins over N-k


def test_index_helpers(oracle_mod):
    assert [oracle_mod.next_pow2(n) for n in (0, 1, 2, 3, 5, 1000, 1024, 1025)] == [1, 1, 2, 4, 8, 1024, 1024, 2048]
    assert [oracle_mod.is_pow2(n) for n in (0, 1, 2, 3, 4, 6, 1024, -4)] == [False, True, True, False, True, False, True, False]
    assert np.array_equal(oracle_mod.fft_shift([0, 1, 2, 3]), [2, 3, 0, 1])
    assert np.array_equal(oracle_mod.fft_shift([0, 1, 2, 3, 4]), [2, 3, 4, 0, 1])  # floor(n/2)
    assert np.array_equal(oracle_mod.bin_frequencies(8, 48000, "two"), np.arange(8) * 6000.0)
    with pytest.raises(ValueError, match="FFT size must be power of two, got 12"):
        oracle_mod.Plan(12)
    with pytest.raises(ValueError, match="FFT input length 5 != size 8"):
        oracle_mod.Plan(8).forward(np.zeros(5))


# ---- gaps the reference does not pin: NumPy cross-check (NOT reference-pinned) ----

@pytest.mark.parametrize("n", [1, 2, 4, 64, 2048, 4096, 16384])
def test_numpy_cross_check_complex_and_inverse(oracle_mod, n):
    rng = np.random.default_rng(n)
    z = rng.standard_normal((3, n)) + 1j * rng.standard_normal((3, n))
    plan = oracle_mod.Plan(n)
    re, im = plan.forward_complex(z.real, z.imag)
    want = np.fft.fft(z, axis=-1)
    assert np.abs(re + 1j * im - want).max() < 1e-10 * max(1, np.sqrt(n))
    bre, bim = plan.inverse(z.real, z.imag)
    assert np.abs(bre + 1j * bim - np.fft.ifft(z, axis=-1)).max() < 1e-12


@pytest.mark.parametrize("window", ["hann", "hamming", "blackman"])
def test_numpy_cross_check_windowed_spectrum(oracle_mod, window):
    from scipy.signal import windows as W
    rng = np.random.default_rng(7)
    x = rng.standard_normal(300)
    r = oracle_mod.spectrum(x, sample_rate=1000, fft_size=512, window=window)
    frame = np.zeros(512)
    frame[:300] = x
    X = np.fft.fft(frame * getattr(W, window)(512, sym=True))
    amp = np.abs(X[:257]) / 512
    amp[1:256] *= 2
    assert np.abs(r["amplitude"] - amp).max() < 1e-12
    plan = oracle_mod.Plan(512)
    a2, p2, k2 = plan.spectrum_batch(frame[None, :], window=oracle_mod.create_window(window, 512), want_phase=True, want_peak=True)
    assert np.array_equal(a2[0], r["amplitude"]) and np.array_equal(p2[0], r["phase"]) and k2[0] == r["peak"]["index"]


def test_complex_ops_known_answers(oracle_mod):
    """test/math/complex.test.ts known answers: (1+2i)(3+4i) = -5+10i etc."""
    r, i = oracle_mod.complex_op("mul", [1.0], [2.0], [3.0], [4.0])
    assert (r[0], i[0]) == (-5.0, 10.0)
    r, i = oracle_mod.complex_op("div", [-5.0], [10.0], [3.0], [4.0])
    assert abs(r[0] - 1) < 1e-15 and abs(i[0] - 2) < 1e-15
    r, i = oracle_mod.complex_op("conj", [1.0, -2.0], [3.0, 4.0])
    assert list(r) == [1.0, -2.0] and list(i) == [-3.0, -4.0]
    r, i = oracle_mod.complex_op("scale", [1.0, 2.0], [3.0, 4.0], s_re=2.0)
    assert list(r) == [2.0, 4.0] and list(i) == [6.0, 8.0]
    r, i = oracle_mod.complex_op("mulScalar", [1.0], [2.0], s_re=3.0, s_im=4.0)
    assert (r[0], i[0]) == (-5.0, 10.0)
    a = np.arange(6.0)
    r, i = oracle_mod.complex_op("add", a, -a, [1.0, 2.0], [10.0, 20.0])  # row broadcast, period 2
    assert list(r) == [1, 3, 3, 5, 5, 7] and list(i) == [10, 19, 8, 17, 6, 15]


# ---- oracle/pdsp_oracle.js: the "Node CPU path" timed by bench.py, pinned on the same goldens ----

def _node_fft(cases):
    import json, os, shutil, subprocess
    node = shutil.which("node")
    if node is None:
        pytest.skip("node not installed")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    req = {"cases": [{"n": len(c["re"]), "re": list(map(float, c["re"])),
                      "im": None if c.get("im") is None else list(map(float, c["im"])),
                      "inverse": bool(c.get("inverse"))} for c in cases]}
    p = subprocess.run([node, os.path.join(root, "oracle", "pdsp_oracle.js"), "fft"], input=json.dumps(req),
                       capture_output=True, text=True, timeout=120, check=True)
    return [(np.array(r["re"]), np.array(r["im"])) for r in json.loads(p.stdout)["results"]]


def test_node_oracle_matches_goldens_and_c_oracle(oracle_mod, reallife, v01, manifest):
    names = [c["name"] for c in manifest["reallife"] if c["name"] not in ("large_amplitude", "tiny_amplitude")]
    got = _node_fft([{"re": reallife[n + "/signal"]} for n in names])
    for n, (re, im) in zip(names, got):  # signals.test.ts:22-23
        assert np.abs(re - reallife[n + "/fftRe"]).max() < 1e-10, n
        assert np.abs(im - reallife[n + "/fftIm"]).max() < 1e-10, n
    # complex input and inverse: against the C oracle (same stage order => agreement to rounding)
    rng = np.random.default_rng(5)
    for n in (1, 2, 8, 64, 4096):
        z = rng.standard_normal((2, n))
        (fre, fim), (bre, bim) = _node_fft([{"re": z[0], "im": z[1]}, {"re": z[0], "im": z[1], "inverse": True}])
        plan = oracle_mod.Plan(n)
        wre, wim = plan.forward_complex(z[0], z[1])
        assert max(np.abs(fre - wre).max(), np.abs(fim - wim).max()) < 1e-12 * n
        wre, wim = plan.inverse(z[0], z[1])
        assert max(np.abs(bre - wre).max(), np.abs(bim - wim).max()) < 1e-14


def test_oracle_spectrum_peak_vs_golden_peak_metadata(oracle_mod, reallife, manifest):
    """findPeak + one-sided scaling of the oracle against the peak metadata the reference's generator
    stored for all 35 cases (peakBin / peakMagnitude / peakPhase, scripts/gen_reallife_refs.py:149-153,
    :503-517), as the reference's tests use them (scaling.test.ts:27-31, :150-201; phase.test.ts:92-97).
    The expectation helper is the one the GPU test uses (tests/test_gpu_golden_peaks.py)."""
    from test_gpu_golden_peaks import FS, N, expected, wrap
    for c in manifest["reallife"]:
        k, amp, ph, _ = expected(c, reallife)
        w = oracle_mod.spectrum(reallife[c["name"] + "/signal"], sample_rate=FS, fft_size=N)
        assert w["peak"]["index"] == k, c["name"]
        assert abs(w["peak"]["amplitude"] - amp) <= 1e-10 * max(1.0, amp), c["name"]
        assert w["peak"]["frequency"] == k * FS / N
        if float(c["peakMagnitude"]) > 1e-6:
            assert wrap(w["peak"]["phase"] - ph) < 1e-10, c["name"]


def test_node_spectrum_restatement_matches_c_oracle_and_known_answers(oracle_mod, reallife):
    """oracle/pdsp_oracle.js `spectrum()` -- the reference's one-shot call restated with its per-call plan
    and window builds, used only as the CPU latency baseline of bench_latency.js -- against the C oracle
    (itself pinned on the goldens) and the inline known answers of SURVEY 8(a)."""
    import json, os, shutil, subprocess
    node = shutil.which("node")
    if node is None:
        pytest.skip("node not installed")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rng = np.random.default_rng(9)
    cases = [
        ([0, 1, 0, -1, 0, 1, 0, -1], {"sampleRate": 48000}),
        (reallife["sine_440hz/signal"].tolist(), {"sampleRate": 48000, "fftSize": 1024, "window": "hann"}),
        (reallife["dc_plus_sine_bin8/signal"].tolist(), {"sampleRate": 48000, "fftSize": 1024, "sides": "two", "window": "blackman"}),
        ([1, 2, 3, 4], {"sampleRate": 48000, "fftSize": 16}),
        (np.zeros(64).tolist(), {"sampleRate": 48000}),
        (rng.standard_normal(300).tolist(), {"window": "hamming"}),
    ]
    req = {"cases": [{"samples": s, "options": o} for s, o in cases]}
    p = subprocess.run([node, os.path.join(root, "oracle", "pdsp_oracle.js"), "spectrum"], input=json.dumps(req),
                       capture_output=True, text=True, timeout=120, check=True)
    res = json.loads(p.stdout)["results"]
    for (s, o), g in zip(cases, res):
        w = oracle_mod.spectrum(s, sample_rate=o.get("sampleRate", 1), fft_size=o.get("fftSize"),
                                window=o.get("window", "rect"), sides=o.get("sides", "one"))
        assert np.array_equal(g["frequencies"], w["frequencies"])
        assert np.abs(np.array(g["amplitude"]) - w["amplitude"]).max() <= 1e-13 * max(1.0, w["amplitude"].max())
        assert g["peak"]["index"] == w["peak"]["index"] and abs(g["peak"]["amplitude"] - w["peak"]["amplitude"]) <= 1e-13
    assert res[0]["peak"] == {"index": 2, "frequency": 12000, "amplitude": 1, "phase": -np.pi / 2}  # README.md:11
    assert abs(res[3]["amplitude"][0] - 0.625) < 1e-15 and res[3]["peak"]["index"] == 1               # edge_cases.test.ts:180-197
    assert not any(res[4]["amplitude"]) and res[4]["peak"]["index"] == 0

"""The reference-language side of the boundary: the Node N-API addon + JS host
(pragma-dsp_amd/js) that stands in for pragma-dsp/core, /xform/fourier and the
root `spectrum` export.  CPU part: it loads, exports the reference's names and
throws the reference's error texts.  GPU part: results vs the oracle and the
reference goldens."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import rel_err

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ADDON = os.path.join(ROOT, "pragma-dsp_amd", "csrc", "pdsp_napi.node")
NODE = shutil.which("node")
needs_node = pytest.mark.skipif(NODE is None or not os.path.exists(ADDON), reason="node or the addon is not available")


def run_cases(cases, tmp_path):
    cin, cout = tmp_path / "cases.json", tmp_path / "out.json"
    cin.write_text(json.dumps(cases))
    subprocess.run([NODE, os.path.join(ROOT, "tests", "js", "run_cases.js"), str(cin), str(cout)],
                   check=True, timeout=120)
    return json.loads(cout.read_text())


@needs_node
def test_js_exports_and_error_texts(tmp_path):
    throws = {
        "size12": "FFT size must be power of two, got 12",
        "fft0": "FFT size must be power of two, got 0",
        "win0": "Window size must be positive, got 0",
        "winType": "Unsupported window type: kaiser",
        "winLen": "Window length must match input length.",
        "binSize": "FFT size must be positive, got 0",
        "binRate": "Sample rate must be positive, got -1",
        "specRate": "Sample rate must be positive, got 0",
        "specSize": "FFT size must be power of two, got 12",
        "specWin": "Unsupported window type: kaiser",
    }
    cases = [{"op": "exports"}, {"op": "misc"}, {"op": "createWindow", "type": "blackman", "size": 64}]
    cases += [{"op": "throws", "what": w} for w in throws]
    res = run_cases(cases, tmp_path)
    assert res[0]["root"] == ["spectrum", "spectrumBatch", "spectrumStream", "core", "fourier"]
    assert res[0]["core"] == ["createComplexArray", "isPowerOfTwo", "nextPowerOfTwo", "Radix2Fft"]
    assert res[0]["fourier"] == ["createWindow", "applyWindow", "FFT", "magnitude", "phase", "fftShift",
                                 "fftShiftComplex", "binFrequencies"]
    assert res[1] == {"next": [1, 1, 8, 1024, 2048], "pow2": [False, True, True, False, False],
                      "shift": [2, 3, 4, 0, 1], "freqs": [0, 6000, 12000, 18000, 24000, 30000, 36000, 42000],
                      "cfill": [7, 7, 7]}
    import oracle
    assert np.abs(np.array(res[2]) - oracle.create_window("blackman", 64)).max() <= 1e-15
    for r, (what, msg) in zip(res[3:], throws.items()):
        assert r["error"] == msg, what


@needs_node
@pytest.mark.gpu
def test_js_dropin_parity(tmp_path, oracle_mod, reallife, manifest):
    rng = np.random.default_rng(11)
    names = ["sine_bin8_amp1.0", "sine_440hz", "three_tone_bin4_bin16_bin48", "chirp_100hz_to_2000hz",
             "impulse_pos512", "dc_plus_sine_bin8", "zeros", "tiny_amplitude", "large_amplitude"]
    z = rng.standard_normal((2, 256))
    holes = [1.0, None, 3.0, None, 5.0, 6.0, None, 8.0]
    cases = [{"op": "forward", "n": 1024, "input": reallife[n + "/signal"].tolist()} for n in names]
    cases += [
        {"op": "forwardComplex", "n": 256, "real": z[0].tolist(), "imag": z[1].tolist()},
        {"op": "inverse", "n": 256, "real": z[0].tolist(), "imag": z[1].tolist()},
        {"op": "forward", "n": 8, "input": holes},
        {"op": "forwardTyped", "n": 8, "input": [0, 1, 0, -1, 0, 1, 0, -1]},
        {"op": "outIdentity", "n": 64, "input": rng.standard_normal(64).tolist()},
        {"op": "spectrum", "samples": [0, 1, 0, -1, 0, 1, 0, -1], "options": {"sampleRate": 48000}},
        {"op": "spectrum", "samples": reallife["sine_440hz/signal"].tolist(),
         "options": {"sampleRate": 48000, "fftSize": 1024, "window": "hann"}},
        {"op": "spectrum", "samples": [1, 2, 3, 4], "options": {"sampleRate": 48000, "fftSize": 16}},
        {"op": "spectrum", "samples": np.zeros(64).tolist(), "options": {"sampleRate": 48000}},
        {"op": "spectrum", "samples": reallife["sine_bin8_amp1.0/signal"].tolist(),
         "options": {"sampleRate": 48000, "fftSize": 1024, "sides": "two", "window": "blackman"}},
        {"op": "applyWindow", "input": z[0].tolist(), "window": oracle_mod.create_window("hamming", 256).tolist()},
        {"op": "magnitude", "real": z[0].tolist(), "imag": z[1].tolist()},
        {"op": "phase", "real": z[0].tolist(), "imag": z[1].tolist()},
        {"op": "throws", "what": "inputLen"},
    ]
    # spectrumBatch: runs of equal-length frames become one device batch each; results in input order
    t = np.arange(1024)
    frames = [(np.sin(2 * np.pi * (8 + i) * t / 1024) + 0.01 * rng.standard_normal(1024)).tolist() for i in range(5)]
    frames += [rng.standard_normal(300).tolist() for _ in range(3)] + [[0, 1, 0, -1, 0, 1, 0, -1]]
    batch_cases = [{"op": "spectrumBatch", "frames": frames, "options": {"sampleRate": 48000, "window": "hann"}},
                   {"op": "spectrumBatch", "frames": frames[:5],
                    "options": {"sampleRate": 48000, "fftSize": 2048, "sides": "two", "window": "blackman"}}]
    batch_cases += [dict(c, op="spectrumBatchFull") for c in batch_cases]
    # frames held as Float64Arrays (read in place through pdsp_spectrum_rows_host_f64), a mixed run, and a call large
    # enough to be cut into chunks on the library's workers (600 frames of 4096 samples, last frame's tone at bin 20)
    typed_cases = [dict(batch_cases[0], op="spectrumBatchTyped"), dict(batch_cases[1], op="spectrumBatchTyped", mixed=True),
                   {"op": "spectrumBatchBig", "n": 4096, "count": 600, "options": {"sampleRate": 48000, "window": "hann"}}]
    # spectrumStream: the frame-at-a-time contract (src/effect/index.ts:190-194) over an iterable whose producer
    # refills one buffer, batches of 4: nine results in order, yielded once their batch (4, 8, 9 frames drawn) ran
    typed_cases.append({"op": "spectrumStream", "frames": frames, "batchFrames": 4,
                        "options": {"sampleRate": 48000, "window": "hann"}})
    # FFT.forwardBatch / forwardComplexBatch / inverseBatch (the reference's batch loop, bench/reallife/signals.ts:264-270,
    # as one device batch): a small call and one large enough for the library's chunked path (300 rows of 8192)
    typed_cases += [{"op": "transformBatch", "n": 1024, "count": 9}, {"op": "transformBatch", "n": 8192, "count": 300}]
    typed_cases.append({"op": "spectrumBatchOverlap", "n": 1024, "options": {"sampleRate": 48000, "window": "hann"}})
    tres = run_cases(typed_cases, tmp_path)
    assert tres[6] == {"same": True, "count": 21}   # (6 n - n) / (n / 4) + 1 overlapping views
    for tb in tres[4:6]:
        assert tb["same"] is True and tb["roundTrip"] < 1e-12 and tb["empty"] == 0
        assert tb["threw"] == "FFT input length 3 != size " + str(1024 if tb["count"] == 9 else 8192)
        assert np.allclose(tb["sineBin2"], [0, 0, -4, 0, 0, 0, 4, 0], atol=1e-12)       # the README's N = 8 sine
        assert np.allclose(tb["hole"], np.fft.fft([1, 0, 1, 1, 1, 1, 1, 1]).real, atol=1e-12)  # a hole reads as 0 (`?? 0`)
    assert [tb["count"] for tb in tres[4:6]] == [9, 300]
    assert tres[0] == {"same": True, "count": 9} and tres[1] == {"same": True, "count": 5}
    assert tres[2] == {"same": True, "count": 600, "lastPeak": 3 + 599 % 97}
    assert tres[3] == {"same": True, "count": 9, "yieldedAfter": [4] * 4 + [8] * 4 + [9], "empty": 0,
                       "threw": "batchFrames must be >= 1, got 0"}
    bres = run_cases(batch_cases, tmp_path)
    assert bres[0]["same"] is True and bres[0]["count"] == 9 and bres[0]["empty"] == 0
    assert bres[0]["bins"] == [513] * 5 + [257] * 3 + [5] and bres[0]["peak0"]["index"] == 8
    assert bres[1]["same"] is True and bres[1]["bins"] == [2048] * 5
    # ... and every streamed result against the ORACLE's spectrum() directly (the map of
    # spectrumStream, src/effect/index.ts:143-194, test/reallife/effect.test.ts:34-45): default f64 mode
    for c, full in zip(batch_cases[:2], bres[2:]):
        opts = c["options"]
        assert len(full) == len(c["frames"])
        for frame, g in zip(c["frames"], full):
            w = oracle_mod.spectrum(frame, sample_rate=opts["sampleRate"], fft_size=opts.get("fftSize"),
                                    window=opts.get("window", "rect"), sides=opts.get("sides", "one"))
            top = max(w["amplitude"].max(), 1e-300)
            assert np.array_equal(g["frequencies"], w["frequencies"])
            assert np.abs(np.array(g["amplitude"]) - w["amplitude"]).max() <= 1e-12 * top
            mask = w["amplitude"] > 1e-4 * top
            dph = (np.array(g["phase"]) - w["phase"] + np.pi) % (2 * np.pi) - np.pi
            assert np.abs(dph[mask]).max(initial=0) <= 1e-9
            n = len(w["amplitude"])
            if opts.get("sides", "one") == "one":
                assert g["peak"]["index"] == w["peak"]["index"]
            else:
                assert g["peak"]["index"] in (w["peak"]["index"], (n - w["peak"]["index"]) % n)
            assert abs(g["peak"]["amplitude"] - w["peak"]["amplitude"]) <= 1e-12 * top
    res = run_cases(cases, tmp_path)
    for r in res[:-1]:
        assert "error" not in r or r.get("error") is None, r
    tol = 1e-5
    for name, r in zip(names, res):
        want = reallife[name + "/fftRe"] + 1j * reallife[name + "/fftIm"]
        got = np.array(r["real"]) + 1j * np.array(r["imag"])
        if name == "zeros":
            assert not got.any()
        else:
            assert rel_err(got, want) <= tol, name
    k = len(names)
    p256 = oracle_mod.Plan(256)
    wre, wim = p256.forward_complex(z[0], z[1])
    assert rel_err(np.array(res[k]["real"]) + 1j * np.array(res[k]["imag"]), wre + 1j * wim) <= tol
    wre, wim = p256.inverse(z[0], z[1])
    assert rel_err(np.array(res[k + 1]["real"]) + 1j * np.array(res[k + 1]["imag"]), wre + 1j * wim) <= tol
    wre, wim = oracle_mod.Plan(8).forward([0 if v is None else v for v in holes])   # `?? 0`
    assert rel_err(np.array(res[k + 2]["real"]) + 1j * np.array(res[k + 2]["imag"]), wre + 1j * wim) <= tol
    assert np.allclose(res[k + 3]["imag"], [0, 0, -4, 0, 0, 0, 4, 0], atol=1e-5)
    ident = res[k + 4]
    assert ident["same"] is True and ident["filled"] is True and ident["size"] == 64 and ident["fill"] == [2, 2]
    assert rel_err(np.array(ident["roundTrip"]), np.array(cases[k + 4]["input"])) <= tol
    s = res[k + 5]
    assert s["peak"]["index"] == 2 and s["peak"]["frequency"] == 12000 and abs(s["peak"]["amplitude"] - 1) < 1e-6
    assert abs(s["peak"]["phase"] + np.pi / 2) < 1e-6 and np.allclose(s["amplitude"], [0, 0, 1, 0, 0], atol=1e-6)
    for idx, (samples, opts) in zip(range(k + 6, k + 10), [(c["samples"], c["options"]) for c in cases[k + 6:k + 10]]):
        w = oracle_mod.spectrum(samples, sample_rate=opts["sampleRate"], fft_size=opts.get("fftSize"),
                                window=opts.get("window", "rect"), sides=opts.get("sides", "one"))
        g = res[idx]
        assert np.array_equal(g["frequencies"], w["frequencies"])
        if w["amplitude"].max() == 0:
            assert not np.any(g["amplitude"]) and g["peak"]["index"] == 0 and g["peak"]["amplitude"] == 0
        else:
            assert rel_err(np.array(g["amplitude"]), w["amplitude"]) <= tol
            n = len(w["amplitude"]) if opts.get("sides") == "two" else 0
            assert g["peak"]["index"] in (w["peak"]["index"], (n - w["peak"]["index"]) % max(n, 1))
    assert abs(res[k + 7]["amplitude"][0] - 0.625) < 1e-6 and res[k + 7]["peak"]["index"] == 1
    assert rel_err(np.array(res[k + 10]), oracle_mod.apply_window(z[0], oracle_mod.create_window("hamming", 256))) <= 1e-6
    assert rel_err(np.array(res[k + 11]), oracle_mod.magnitude(z[0], z[1])) <= 1e-6
    assert np.abs((np.array(res[k + 12]) - oracle_mod.phase(z[0], z[1]) + np.pi) % (2 * np.pi) - np.pi).max() <= 1e-5
    assert res[-1]["error"] == "FFT input length 3 != size 8"


@needs_node
@pytest.mark.gpu
def test_reference_suite_in_javascript_through_the_dropin(tmp_path, reallife, v01, manifest):
    """tests/js/reference_suite.js: every it(...) of the reference's hot-path tests (fft, spectrum, window,
    reallife/{signals, phase, scaling, edge_cases}.test.ts), with its assertions and tolerances, run under Node
    against pragma-dsp_amd/js -- the reference's own language on the drop-in (default f64 arithmetic).  The fixtures
    are the reference's goldens (tests/golden/*.npz) handed over as JSON."""
    fx = {"reallife": [], "v01": [], "windows": []}
    for m in manifest["reallife"]:
        re, im = reallife[m["name"] + "/fftRe"], reallife[m["name"] + "/fftIm"]
        fx["reallife"].append({**{k: m[k] for k in ("name", "kind", "family", "n", "sampleRate", "params")},
                               "signal": reallife[m["name"] + "/signal"].tolist(), "fftRe": re.tolist(), "fftIm": im.tolist(),
                               "magnitude": np.hypot(re, im).tolist(), "phase": np.arctan2(im, re).tolist()})
    for m in manifest["v01_cases"]:
        if m["n"] > 1024:
            continue  # the benchmark inputs: no test reads them
        fx["v01"].append({**{k: m[k] for k in ("name", "kind", "n", "sampleRate", "meta")},
                          "input": v01[f"case/{m['name']}/input"].tolist(), "fftRe": v01[f"case/{m['name']}/fftRe"].tolist(),
                          "fftIm": v01[f"case/{m['name']}/fftIm"].tolist()})
    for w in manifest["v01_windows"]:
        fx["windows"].append({"type": w["type"], "n": w["n"], "values": v01[w["key"]].tolist()})
    fin, fout = tmp_path / "fixtures.json", tmp_path / "results.json"
    fin.write_text(json.dumps(fx))
    subprocess.run([NODE, os.path.join(ROOT, "tests", "js", "reference_suite.js"), str(fin), str(fout)], check=True, timeout=300)
    res = json.loads(fout.read_text())
    bad = [r for r in res if not r["ok"]]
    assert not bad, bad[:5]
    # 15 x 2 random cases, 1 spectrum case, 28 windows, 23 x 4 pure sines, 2 x 2 multi-tone, 1 x 2 chirp, 4 special,
    # phase: 1 + 5 x 2 + 4, scaling: 15 x 3 + 6, edge cases: 11
    assert len(res) == 30 + 1 + 28 + 92 + 4 + 2 + 4 + 15 + 51 + 11 == 238, len(res)


@needs_node
def test_napi_addon_argument_handling_under_address_sanitizer(tmp_path):
    """The addon is the one piece of native code a JS caller reaches with arbitrary arguments.  An ASan + UBSan build of
    pdsp_napi.c (CPU only: sanitizers are not available on the GPU pool) is driven with short outputs, mismatched planes,
    bad sizes, wrong types and wrong arities: every case must end in a JS exception or a correct value, and the sanitizers
    must stay silent.  No GPU: every compute call stops at validation or at "no HIP device"."""
    gcc = shutil.which("gcc")
    if gcc is None or not os.path.exists("/usr/include/node/node_api.h"):
        pytest.skip("gcc or node headers missing")
    asan = subprocess.run([gcc, "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    ubsan = subprocess.run([gcc, "-print-file-name=libubsan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("libasan is not installed")
    csrc = os.path.join(ROOT, "pragma-dsp_amd", "csrc")
    addon = str(tmp_path / "pdsp_napi.node")
    subprocess.run([gcc, "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fPIC", "-shared", "-std=gnu11",
                    "-Wall", "-DNODE_GYP_MODULE_NAME=pdsp_napi", "-I/usr/include/node", "-I" + os.path.join(ROOT, "include"),
                    "-o", addon, os.path.join(csrc, "pdsp_napi.c"), "-L" + csrc, "-lpdsp_hip", "-Wl,-rpath," + csrc],
                   check=True, capture_output=True, text=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1",
               LD_PRELOAD=asan + (":" + ubsan if os.path.exists(ubsan) else ""))
    p = subprocess.run([NODE, os.path.join(ROOT, "tests", "js", "asan_cases.js"), addon], capture_output=True, text=True,
                       timeout=120, env=env)
    assert "AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr, p.stderr[-3000:]
    assert p.returncode == 0, (p.returncode, p.stderr[-2000:])
    log = {name: (kind, val) for name, kind, val in json.loads(p.stdout.strip().splitlines()[-1])}
    assert log["windowMake"][0] == "ok" and abs(float(log["windowMake"][1]) - 0.0008984113451827869) < 1e-15
    assert log["nextPow2"] == ("ok", "1,1,8,2048,4294967296")
    assert log["binFrequencies"] == ("ok", "0,6000,12000,18000,24000") and log["fftShift"] == ("ok", "2,3,4,0,1")
    assert log["planCreate 12"] == ("throws", "FFT size must be power of two, got 12")
    assert log["windowMake size 0"] == ("throws", "Window size must be positive, got 0")
    assert log["applyWindow length mismatch"] == ("throws", "Window length must match input length.")
    for name in ("windowMake short out", "binFrequencies short out", "fftShift short out", "magnitude short out",
                 "phase mismatched planes", "spectrum short out", "spectrumBatch short out", "transform wrong types",
                 "wrong argument count", "planCreate 0", "planCreate -8", "windowMake bad type", "binFrequencies rate 0",
                 "spectrum bad size", "spectrumRows not an array", "spectrumRows range past the end",
                 "spectrumRows negative start", "spectrumRows ragged frame", "spectrumRows plain-array frame",
                 "spectrumRows hole", "spectrumRows short out", "spectrumRows bad size", "spectrumRows bad rate",
                 "transformBatch wrong plan", "transformRows wrong plan", "transformRows not an array",
                 "spectrumRows mixed kinds", "spectrumRows Int32Array frame"):
        assert log[name][0] == "throws", (name, log[name])
    # well-formed Float32Array frames: computes where a GPU is present (zeros in, amplitude 0 out), else the device error
    assert log["spectrumRows f32 frames"] == ("ok", "0") or log["spectrumRows f32 frames"][0] == "throws"
    assert log["spectrumRows bad size"] == ("throws", "FFT size must be power of two, got 12")
    assert log["spectrumRows bad rate"] == ("throws", "Sample rate must be positive, got 0")

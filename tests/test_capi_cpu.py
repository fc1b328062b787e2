"""CPU-side checks of the product boundary: the C-ABI library loads and exports
every symbol include/pdsp_hip.h declares, host index math matches the oracle,
argument errors carry the reference's texts, and the product never routes through
the oracle or any CPU fallback.  No compute calls (no GPU here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols(name="pdsp_hip.h"):
    text = open(os.path.join(ROOT, "include", name)).read()
    return sorted(set(re.findall(r"PDSP_API\s+[\w\s\*]+?\b(pdsp_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol(pdsp):
    syms = header_symbols()
    dev = header_symbols("pdsp_hip_dev.h")
    assert len(syms) >= 25 and len(dev) >= 4
    raw = C.CDLL(pdsp.LIB_PATH)
    for s in syms + dev:
        assert hasattr(raw, s), f"{s} declared in include/*.h but not exported"
    # the boundary header carries no development switch: only the user-facing precision knob
    assert [s for s in syms if s.startswith("pdsp_set_")] == ["pdsp_set_host_precision"]
    assert all(s.startswith("pdsp_set_") for s in dev) and not set(dev) & set(syms)
    # and the ctypes binding covers all of them
    assert set(syms) | set(dev) == set(pdsp.lib._pdsp_symbols)
    assert pdsp.lib.pdsp_version() >= 100
    assert pdsp.lib.pdsp_max_size(4) == 1 << 28 and pdsp.lib.pdsp_max_size(8) == 1 << 26


def test_product_does_not_touch_the_oracle():
    pkg = os.path.join(ROOT, "pragma-dsp_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".c", ".cpp", ".js")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                code = "\n".join(l for l in src.splitlines() if not l.lstrip().startswith(("//", "#", "*", "/*")))
                assert not re.search(r"\b(import|from)\s+oracle\b|oracle/|pdsp_oracle|liboracle", code), \
                    f"{f} references the oracle"
                assert not re.search(r"np\.fft|numpy\.fft|torch\.fft|scipy\.fft|rocfft|hipfft", code, re.I), \
                    f"{f} has an FFT fallback"


def test_host_index_math_matches_oracle(pdsp, oracle_mod):
    for n in (-3, 0, 1, 2, 3, 7, 8, 1000, 4096, 4097, 1 << 20, (1 << 31) + 5):
        assert pdsp.lib.pdsp_next_pow2(n) == oracle_mod.next_pow2(n)
        assert bool(pdsp.lib.pdsp_is_pow2(n)) == oracle_mod.is_pow2(n)
    assert pdsp.nextPowerOfTwo(5) == 8 and pdsp.nextPowerOfTwo(0) == 1
    assert pdsp.isPowerOfTwo(1024) and not pdsp.isPowerOfTwo(0) and not pdsp.isPowerOfTwo(2.5)
    for kind in ("rect", "hann", "hamming", "blackman"):
        for n in (1, 2, 8, 64, 1024, 4096):
            assert np.abs(pdsp.createWindow(kind, n) - oracle_mod.create_window(kind, n)).max() <= 1e-15
    for sides in ("one", "two"):
        assert np.array_equal(pdsp.binFrequencies(1024, 48000, sides), oracle_mod.bin_frequencies(1024, 48000, sides))
    assert np.array_equal(pdsp.binFrequencies(1, 2.5), [0.0])
    rng = np.random.default_rng(3)
    for n in (1, 2, 3, 4, 5, 64, 129):
        x = rng.standard_normal(n)
        assert np.array_equal(pdsp.fftShift(x), oracle_mod.fft_shift(x))
    z = pdsp.fftShiftComplex(pdsp.ComplexArray(np.arange(4.0), -np.arange(4.0)))
    assert np.array_equal(z.real, [2, 3, 0, 1]) and np.array_equal(z.imag, [-2, -3, 0, -1])
    for _ in range(200):
        n = int(rng.integers(1, 40))
        a = np.abs(rng.integers(0, 4, n)).astype(np.float64)  # many ties and zeros
        assert pdsp.lib.pdsp_find_peak_f64(a.ctypes.data_as(C.POINTER(C.c_double)), n) == oracle_mod.find_peak(a)


def test_windows_against_reference_goldens(pdsp, windows_dsp, v01, manifest):
    # the product's createWindow against the reference's SciPy goldens (window.test.ts:8)
    for w in manifest["windows_dsp"]:
        assert np.abs(pdsp.createWindow(w["type"], w["n"]) - windows_dsp[w["key"]]).max() < 1e-8
    for w in manifest["v01_windows"]:
        assert np.abs(pdsp.createWindow(w["type"], w["n"]) - v01[w["key"]]).max() < 1e-8


def test_error_texts_match_reference(pdsp):
    E = pdsp.PdspError
    with pytest.raises(E, match=r"^FFT size must be power of two, got 12$"):
        pdsp.Radix2Fft(12)
    with pytest.raises(E, match=r"^FFT size must be power of two, got 0$"):
        pdsp.FFT(0)
    with pytest.raises(E, match=r"^FFT size must be power of two, got 2\.5$"):
        pdsp.Radix2Fft(2.5)
    with pytest.raises(E, match=r"^Window size must be positive, got 0$"):
        pdsp.createWindow("hann", 0)
    with pytest.raises(E, match=r"^Window size must be positive, got -3$"):
        pdsp.createWindow("rect", -3)
    with pytest.raises(E, match=r"^Unsupported window type: kaiser$"):
        pdsp.createWindow("kaiser", 8)
    assert np.array_equal(pdsp.createWindow("kaiser", 1), [1.0])  # size 1 returns before the type switch
    with pytest.raises(E, match=r"^Window length must match input length\.$"):
        pdsp.applyWindow([1, 2, 3], [1, 2])
    with pytest.raises(E, match=r"^FFT size must be positive, got 0$"):
        pdsp.binFrequencies(0, 48000)
    with pytest.raises(E, match=r"^Sample rate must be positive, got -1$"):
        pdsp.binFrequencies(8, -1)
    with pytest.raises(E, match=r"^Sample rate must be positive, got 0$"):
        pdsp.spectrum([1, 2, 3, 4], {"sampleRate": 0})
    with pytest.raises(E, match=r"^FFT size must be power of two, got 12$"):
        pdsp.spectrum([1, 2, 3, 4], {"fftSize": 12, "sampleRate": -1})  # FFT ctor throws first
    with pytest.raises(E, match=r"^Unsupported window type: kaiser$"):
        pdsp.spectrum([1, 2, 3, 4], {"window": "kaiser", "sampleRate": -1})  # then createWindow


def test_c_abi_status_codes_without_device(pdsp):
    from pragma_dsp_amd import _capi
    lib = pdsp.lib
    h = C.c_void_p()
    assert lib.pdsp_plan_create(12, -1, C.byref(h)) == _capi.ERR_SIZE_NOT_POW2
    assert lib.pdsp_last_error() == b"FFT size must be power of two, got 12"
    assert lib.pdsp_plan_create(1 << 29, -1, C.byref(h)) == _capi.ERR_UNSUPPORTED_SIZE
    out = np.empty(4)
    assert lib.pdsp_window_make(1, 0, out.ctypes.data_as(C.POINTER(C.c_double))) == _capi.ERR_WINDOW_SIZE
    assert lib.pdsp_last_error() == b"Window size must be positive, got 0"
    assert lib.pdsp_window_make(9, 4, out.ctypes.data_as(C.POINTER(C.c_double))) == _capi.ERR_WINDOW_TYPE
    assert lib.pdsp_bin_frequencies(0, 1.0, 0, None, None) == _capi.ERR_FFT_SIZE
    assert lib.pdsp_bin_frequencies(8, 0.0, 0, None, None) == _capi.ERR_SAMPLE_RATE
    x = np.ones(4)
    dp = C.POINTER(C.c_double)
    assert lib.pdsp_apply_window_host_f64(x.ctypes.data_as(dp), 4, x.ctypes.data_as(dp), 3, out.ctypes.data_as(dp)) \
        == _capi.ERR_WINDOW_LENGTH
    assert lib.pdsp_plan_destroy(None) == 0 and lib.pdsp_plan_size(None) == 0
    if lib.pdsp_device_count() == 0:
        # no CPU fallback: the product must fail loudly without a GPU
        assert lib.pdsp_plan_create(8, -1, C.byref(h)) == _capi.ERR_DEVICE
        assert b"no HIP device" in lib.pdsp_last_error()
        with pytest.raises(pdsp.PdspError, match="no HIP device"):
            pdsp.Radix2Fft(8)
        with pytest.raises(pdsp.PdspError, match="no HIP device"):
            pdsp.spectrum([0, 1, 0, -1])
        with pytest.raises(pdsp.PdspError, match="no HIP device"):
            pdsp.magnitude(pdsp.ComplexArray(np.ones(4), np.ones(4)))


def test_js_number_formatting():
    from pragma_dsp_amd.core import js_num
    assert js_num(12) == "12" and js_num(48000.0) == "48000" and js_num(2.5) == "2.5"
    assert js_num(-1.0) == "-1" and js_num(float("nan")) == "NaN" and js_num(float("inf")) == "Infinity"


def test_radix_plan_layout_is_consistent():
    """pdsp_radix.h restated: the pass radices multiply to N and the twiddle blocks tile the table."""
    for log2n in range(0, 15):
        n = 1 << log2n
        if log2n <= 4:
            radices = [n] if log2n else []
        else:
            radices = [16] * (log2n // 4) + ([1 << (log2n % 4)] if log2n % 4 else [])
        assert int(np.prod(radices)) == n if radices else n == 1
        assert len(radices) <= 4


def _build_c_consumer(tmp_path):
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    exe = str(tmp_path / "abi_smoke")
    csrc = os.path.join(ROOT, "pragma-dsp_amd", "csrc")
    subprocess.run(["gcc", "-std=c11", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"), "-o", exe,
                    os.path.join(ROOT, "tests", "c", "abi_smoke.c"), "-L" + csrc, "-lpdsp_hip", "-Wl,-rpath," + csrc, "-lm"],
                   check=True, capture_output=True, text=True)
    return exe


def test_headers_are_valid_c11_and_a_plain_c_consumer_links_and_fails_loudly_without_a_gpu(tmp_path):
    """include/pdsp_hip.h promises a C ABI: a C11 translation unit with -pedantic -Werror must compile against it (not
    only the C++ of the library itself), link to libpdsp_hip.so and -- here, without a GPU -- get PDSP_ERR_DEVICE with a
    message instead of a silent CPU fallback."""
    import subprocess
    import torch
    for h in ("pdsp_hip.h", "pdsp_hip_dev.h"):
        subprocess.run(["gcc", "-std=c11", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                        "-fsyntax-only", "-x", "c", os.path.join(ROOT, "include", h)], check=True, capture_output=True)
    exe = _build_c_consumer(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: tests/test_gpu_c_consumer.py runs it")
    p = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert p.returncode == 1 and "status 10" in p.stderr and "no HIP device available" in p.stderr and "no CPU fallback" in p.stderr

"""ctypes front-end of oracle/pdsp_oracle.c -- TEST INFRASTRUCTURE ONLY.

Never imported from pragma-dsp_amd/.  Reference lines are cited in the C file.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libpdsp_oracle.so")
_SRC = os.path.join(_HERE, "pdsp_oracle.c")

WINDOW_TYPES = {"rect": 0, "hann": 1, "hamming": 2, "blackman": 3}


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (seconds).  Returns the .so path."""
    stale = (not os.path.exists(_SO)) or os.path.getmtime(_SO) < os.path.getmtime(_SRC)
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libpdsp_oracle.so"])
    return _SO


class _Peak(C.Structure):
    _fields_ = [("index", C.c_int), ("frequency", C.c_double),
                ("amplitude", C.c_double), ("phase", C.c_double)]


_lib = None


def _L():
    global _lib
    if _lib is None:
        lib = C.CDLL(build())
        dp = C.POINTER(C.c_double)
        lib.oracle_is_pow2.argtypes = [C.c_longlong]
        lib.oracle_next_pow2.argtypes = [C.c_longlong]
        lib.oracle_next_pow2.restype = C.c_longlong
        lib.oracle_plan_create.argtypes = [C.c_int]
        lib.oracle_plan_create.restype = C.c_void_p
        lib.oracle_plan_destroy.argtypes = [C.c_void_p]
        lib.oracle_transform_batch.argtypes = [C.c_void_p, C.c_longlong, dp, dp, dp, dp, C.c_int]
        lib.oracle_create_window.argtypes = [C.c_int, C.c_int, dp]
        lib.oracle_apply_window.argtypes = [dp, dp, C.c_int, dp]
        lib.oracle_magnitude.argtypes = [dp, dp, C.c_longlong, dp]
        lib.oracle_phase.argtypes = [dp, dp, C.c_longlong, dp]
        lib.oracle_fft_shift.argtypes = [dp, C.c_int, dp]
        lib.oracle_bin_frequencies.argtypes = [C.c_int, C.c_double, C.c_int, dp]
        lib.oracle_find_peak.argtypes = [dp, C.c_int]
        lib.oracle_scale_amplitude.argtypes = [dp, C.c_int, C.c_int, dp]
        lib.oracle_spectrum.argtypes = [dp, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int,
                                        dp, dp, dp, C.POINTER(_Peak)]
        lib.oracle_spectrum_batch.argtypes = [C.c_void_p, C.c_longlong, dp, dp, C.c_int, dp, dp,
                                              C.POINTER(C.c_int)]
        lib.oracle_complex_op.argtypes = [C.c_int, C.c_longlong, dp, dp, dp, dp, C.c_longlong, C.c_double,
                                          C.c_double, dp, dp]
        lib.oracle_time_forward.argtypes = [C.c_void_p, C.c_longlong, C.c_int, dp, dp, dp]
        lib.oracle_time_forward.restype = C.c_double
        _lib = lib
    return _lib


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def is_pow2(n: int) -> bool:
    return bool(_L().oracle_is_pow2(int(n)))


def next_pow2(n: int) -> int:
    return int(_L().oracle_next_pow2(int(n)))


class Plan:
    """Radix2Fft restated (src/core/fft.ts:63-152); rows are transformed one by one."""

    def __init__(self, n: int):
        h = _L().oracle_plan_create(int(n))
        if not h:
            raise ValueError(f"FFT size must be power of two, got {n}")
        self._h = h
        self.size = int(n)

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.oracle_plan_destroy(self._h)
            self._h = None

    def _run(self, re, im, inverse):
        re = _f64(re)
        if re.shape[-1] != self.size:
            raise ValueError(f"FFT input length {re.shape[-1]} != size {self.size}")
        if im is not None:
            im = _f64(im)
            if im.shape != re.shape:
                raise ValueError(f"FFT input length {im.shape[-1]} != size {self.size}")
        batch = re.size // self.size if self.size else 0
        ore = np.empty_like(re)
        oim = np.empty_like(re)
        _L().oracle_transform_batch(self._h, batch, _p(re), _p(im), _p(ore), _p(oim), int(inverse))
        return ore, oim

    def forward(self, x):
        return self._run(x, None, False)

    def forward_complex(self, re, im):
        return self._run(re, im, False)

    def inverse(self, re, im):
        return self._run(re, im, True)

    def spectrum_batch(self, frames, window=None, two_sided=False, want_phase=False, want_peak=False):
        frames = _f64(frames)
        batch = frames.size // self.size
        bins = self.size if two_sided else self.size // 2 + 1
        win = _f64(window) if window is not None else None
        amp = np.empty(frames.shape[:-1] + (bins,), dtype=np.float64)
        ph = np.empty_like(amp) if want_phase else None
        pk = np.empty(batch, dtype=np.int32) if want_peak else None
        _L().oracle_spectrum_batch(self._h, batch, _p(frames), _p(win), int(two_sided), _p(amp), _p(ph),
                                   pk.ctypes.data_as(C.POINTER(C.c_int)) if pk is not None else None)
        return amp, ph, pk

    def time_forward(self, re, im=None, reps=1):
        re = _f64(re)
        im = _f64(im) if im is not None else None
        batch = re.size // self.size
        chk = C.c_double(0.0)
        sec = _L().oracle_time_forward(self._h, batch, int(reps), _p(re), _p(im), C.byref(chk))
        return sec, chk.value


def create_window(kind: str, size: int) -> np.ndarray:
    if size <= 0:
        raise ValueError(f"Window size must be positive, got {size}")
    if kind not in WINDOW_TYPES:
        raise ValueError(f"Unsupported window type: {kind}")
    out = np.empty(int(size), dtype=np.float64)
    _L().oracle_create_window(WINDOW_TYPES[kind], int(size), _p(out))
    return out


def apply_window(x, w) -> np.ndarray:
    x, w = _f64(x), _f64(w)
    if x.shape != w.shape:
        raise ValueError("Window length must match input length.")
    out = np.empty_like(x)
    _L().oracle_apply_window(_p(x), _p(w), x.size, _p(out))
    return out


def magnitude(re, im) -> np.ndarray:
    re, im = _f64(re), _f64(im)
    out = np.empty_like(re)
    _L().oracle_magnitude(_p(re), _p(im), re.size, _p(out))
    return out


def phase(re, im) -> np.ndarray:
    re, im = _f64(re), _f64(im)
    out = np.empty_like(re)
    _L().oracle_phase(_p(re), _p(im), re.size, _p(out))
    return out


def fft_shift(x) -> np.ndarray:
    x = _f64(x)
    out = np.empty_like(x)
    _L().oracle_fft_shift(_p(x), x.size, _p(out))
    return out


def bin_frequencies(size: int, sample_rate: float, sides: str = "one") -> np.ndarray:
    if size <= 0:
        raise ValueError(f"FFT size must be positive, got {size}")
    if not sample_rate > 0:
        raise ValueError(f"Sample rate must be positive, got {sample_rate}")
    out = np.empty(size if sides == "two" else size // 2 + 1, dtype=np.float64)
    _L().oracle_bin_frequencies(int(size), float(sample_rate), int(sides == "two"), _p(out))
    return out


def find_peak(amp) -> int:
    amp = _f64(amp)
    return int(_L().oracle_find_peak(_p(amp), amp.size))


def scale_amplitude(mag, size: int, sides: str = "one") -> np.ndarray:
    mag = _f64(mag)
    out = np.empty(size if sides == "two" else size // 2 + 1, dtype=np.float64)
    _L().oracle_scale_amplitude(_p(mag), int(size), int(sides == "two"), _p(out))
    return out


def spectrum(samples, sample_rate=1.0, fft_size=None, window="rect", sides="one") -> dict:
    """spectrum() restated (src/public/spectrum.ts:107-142)."""
    x = _f64(samples)
    n = int(fft_size) if fft_size is not None else next_pow2(x.size)
    if not is_pow2(n):
        raise ValueError(f"FFT size must be power of two, got {n}")
    if window not in WINDOW_TYPES:
        raise ValueError(f"Unsupported window type: {window}")
    if not sample_rate > 0:
        raise ValueError(f"Sample rate must be positive, got {sample_rate}")
    bins = n if sides == "two" else n // 2 + 1
    freq = np.empty(bins)
    amp = np.empty(n)
    ph = np.empty(n)
    pk = _Peak()
    rc = _L().oracle_spectrum(_p(x), x.size, float(sample_rate), n, WINDOW_TYPES[window],
                              int(sides == "two"), _p(freq), _p(amp), _p(ph), C.byref(pk))
    if rc < 0:
        raise RuntimeError(f"oracle_spectrum failed: {rc}")
    return {
        "frequencies": freq,
        "amplitude": amp[:bins].copy(),
        "phase": ph[:bins].copy(),
        "peak": {"index": pk.index, "frequency": pk.frequency,
                 "amplitude": pk.amplitude, "phase": pk.phase},
    }


COMPLEX_OPS = {"add": 0, "sub": 1, "mul": 2, "div": 3, "conj": 4, "scale": 5, "mulScalar": 6}


def complex_op(name, are, aim, bre=None, bim=None, s_re=0.0, s_im=0.0):
    """src/math/complex.ts restated; b broadcasts over the rows of a."""
    are, aim = _f64(are), _f64(aim)
    bre = _f64(bre) if bre is not None else None
    bim = _f64(bim) if bim is not None else None
    orr, oi = np.empty_like(are), np.empty_like(are)
    _L().oracle_complex_op(COMPLEX_OPS[name], are.size, _p(are), _p(aim), _p(bre), _p(bim),
                           bre.size if bre is not None else 1, float(s_re), float(s_im), _p(orr), _p(oi))
    return orr, oi

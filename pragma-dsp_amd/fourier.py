"""Host mirror of `pragma-dsp/xform/fourier` (src/xform/fourier.ts) over the HIP C ABI."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi
from ._capi import PdspError, check, dptr, lib
from .core import ComplexArray, Radix2Fft, as_f64, createComplexArray, isPowerOfTwo, js_num, _complex_planes

WindowType = ("rect", "hann", "hamming", "blackman")
FftSides = ("one", "two")


def createWindow(type: str, size) -> np.ndarray:
    """src/xform/fourier.ts:14-52 -- symmetric windows, built on the host in f64."""
    if size <= 0:
        raise PdspError(_capi.ERR_WINDOW_SIZE, f"Window size must be positive, got {js_num(size)}")
    size = int(size)
    if size == 1:
        return np.ones(1, dtype=np.float64)
    if type not in _capi.WINDOW_TYPES:
        raise PdspError(_capi.ERR_WINDOW_TYPE, f"Unsupported window type: {type}")
    out = np.empty(size, dtype=np.float64)
    check(lib.pdsp_window_make(_capi.WINDOW_TYPES[type], size, dptr(out)))
    return out


def applyWindow(input, window, out: np.ndarray | None = None) -> np.ndarray:
    """src/xform/fourier.ts:54-67."""
    if len(input) != len(window):
        raise PdspError(_capi.ERR_WINDOW_LENGTH, "Window length must match input length.")
    x, w = as_f64(input), as_f64(window)
    result = out if out is not None else np.empty(len(x), dtype=np.float64)
    tmp = result if (isinstance(result, np.ndarray) and result.dtype == np.float64
                     and result.flags.c_contiguous and len(result) == len(x)) else np.empty(len(x), dtype=np.float64)
    check(lib.pdsp_apply_window_host_f64(dptr(x), len(x), dptr(w), len(w), dptr(tmp)))
    if tmp is not result:
        result[:len(x)] = tmp
    return result


class FFT:
    """src/xform/fourier.ts:69-96 -- a pure delegate to Radix2Fft."""

    def __init__(self, size, device: int = -1):
        if not isPowerOfTwo(size):
            raise PdspError(_capi.ERR_SIZE_NOT_POW2, f"FFT size must be power of two, got {js_num(size)}")
        self.size = int(size)
        self._kernel = Radix2Fft(size, device)

    def forward(self, input, out: ComplexArray | None = None) -> ComplexArray:
        return self._kernel.forward(input, out)

    def forwardComplex(self, input, out: ComplexArray | None = None) -> ComplexArray:
        return self._kernel.forwardComplex(input, out)

    def inverse(self, input, out: ComplexArray | None = None) -> ComplexArray:
        return self._kernel.inverse(input, out)

    def createComplexArray(self, fill: float = 0) -> ComplexArray:
        return createComplexArray(self.size, fill)

    # extensions (see Radix2Fft): many rows in one device batch
    def forwardBatch(self, inputs) -> list:
        return self._kernel.forwardBatch(inputs)

    def forwardComplexBatch(self, inputs) -> list:
        return self._kernel.forwardComplexBatch(inputs)

    def inverseBatch(self, inputs) -> list:
        return self._kernel.inverseBatch(inputs)


def _polar(input, out, fn):
    re, im = _complex_planes(input)
    re, im = as_f64(re), as_f64(im)
    n = len(re)
    if len(im) < n:  # `input.imag[i] ?? 0`
        im = np.concatenate([im, np.zeros(n - len(im))])
    result = out if out is not None else np.empty(n, dtype=np.float64)
    tmp = result if (isinstance(result, np.ndarray) and result.dtype == np.float64
                     and result.flags.c_contiguous and len(result) == n) else np.empty(n, dtype=np.float64)
    check(fn(dptr(re), dptr(im[:n].copy() if len(im) != n else im), n, dptr(tmp)))
    if tmp is not result:
        result[:n] = tmp
    return result


def magnitude(input, out: np.ndarray | None = None) -> np.ndarray:
    """src/xform/fourier.ts:98-109."""
    return _polar(input, out, lib.pdsp_magnitude_host_f64)


def phase(input, out: np.ndarray | None = None) -> np.ndarray:
    """src/xform/fourier.ts:111-120."""
    return _polar(input, out, lib.pdsp_phase_host_f64)


def fftShift(input, out: np.ndarray | None = None) -> np.ndarray:
    """src/xform/fourier.ts:122-134 (host index math)."""
    x = as_f64(input)
    result = out if out is not None else np.empty(len(x), dtype=np.float64)
    tmp = np.empty(len(x), dtype=np.float64)
    check(lib.pdsp_fft_shift_f64(dptr(x), len(x), dptr(tmp)))
    result[:len(x)] = tmp
    return result


def fftShiftComplex(input, out: ComplexArray | None = None) -> ComplexArray:
    """src/xform/fourier.ts:136-145."""
    re, im = _complex_planes(input)
    result = out if out is not None else createComplexArray(len(re))
    result.real[:] = fftShift(re)
    result.imag[:] = fftShift(im)
    return result


def binFrequencies(size, sampleRate, sides: str = "one") -> np.ndarray:
    """src/xform/fourier.ts:147-165."""
    if size <= 0:
        raise PdspError(_capi.ERR_FFT_SIZE, f"FFT size must be positive, got {js_num(size)}")
    if sampleRate <= 0:
        raise PdspError(_capi.ERR_SAMPLE_RATE, f"Sample rate must be positive, got {js_num(sampleRate)}")
    size = int(size)
    out = np.empty(size // 2 + 1 if sides == "one" else size, dtype=np.float64)
    bins = C.c_longlong(0)
    check(lib.pdsp_bin_frequencies(size, float(sampleRate), 0 if sides == "one" else 1, dptr(out), C.byref(bins)))
    return out

"""Host mirror of `pragma-dsp` root export: spectrum() (src/public/spectrum.ts:107-142)."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _capi
from ._capi import PdspError, check, dptr, lib
from .core import as_f64, isPowerOfTwo, js_num, nextPowerOfTwo


@dataclass
class SpectrumPeak:  # src/public/spectrum.ts:15-20
    index: int
    frequency: float
    amplitude: float
    phase: float


@dataclass
class SpectrumResult:  # src/public/spectrum.ts:22-27
    frequencies: np.ndarray
    amplitude: np.ndarray
    phase: np.ndarray
    peak: SpectrumPeak


def spectrum(samples, options: dict | None = None, **kw) -> SpectrumResult:
    """One fused kernel launch: buildFrame -> window -> FFT -> magnitude/phase ->
    amplitude scaling; frequency axis and findPeak are host index math over the
    f64-promoted result.  Options as the reference's SpectrumOptions
    {sampleRate=1, fftSize=nextPow2(len), window="rect", sides="one"}."""
    opts = dict(options or {})
    opts.update(kw)
    sample_rate = opts["sampleRate"] if opts.get("sampleRate") is not None else 1
    sides = opts["sides"] if opts.get("sides") is not None else "one"
    x = as_f64(samples)
    target = opts["fftSize"] if opts.get("fftSize") is not None else nextPowerOfTwo(len(x))
    window = opts["window"] if opts.get("window") is not None else "rect"
    # error order of spectrum.ts:114-132: FFT ctor, createWindow, binFrequencies
    if not isPowerOfTwo(target):
        raise PdspError(_capi.ERR_SIZE_NOT_POW2, f"FFT size must be power of two, got {js_num(target)}")
    target = int(target)
    if target != 1 and window not in _capi.WINDOW_TYPES:
        raise PdspError(_capi.ERR_WINDOW_TYPE, f"Unsupported window type: {window}")
    if sample_rate <= 0:
        raise PdspError(_capi.ERR_SAMPLE_RATE, f"Sample rate must be positive, got {js_num(sample_rate)}")
    two = sides != "one"  # the reference treats anything but "one" as two-sided (spectrum.ts:124-131)
    bins = target if two else target // 2 + 1
    freq = np.empty(bins, dtype=np.float64)
    amp = np.empty(bins, dtype=np.float64)
    ph = np.empty(bins, dtype=np.float64)
    pk = _capi.Peak()
    nb = C.c_longlong(0)
    check(lib.pdsp_spectrum_host_f64(dptr(x), len(x), float(sample_rate), target,
                                     _capi.WINDOW_TYPES.get(window, 0), 1 if two else 0,
                                     dptr(freq), dptr(amp), dptr(ph), C.byref(pk), C.byref(nb)))
    return SpectrumResult(freq, amp, ph, SpectrumPeak(int(pk.index), float(pk.frequency),
                                                      float(pk.amplitude), float(pk.phase)))

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


def spectrumBatch(frames, options: dict | None = None, **kw) -> list:
    """Extension (as in the JS host, js/spectrum.js): the map of the reference's spectrumStream
    (src/effect/index.ts:190-194 -- one spectrum() result per frame, in order) as ONE device batch per run
    of equal-length frames.  Result i equals spectrum(frames[i], options) exactly; the frames of a run are
    read where they lie (pdsp_spectrum_rows_host_f64; float32 arrays through pdsp_spectrum_rows_host_f32in)."""
    opts = dict(options or {})
    opts.update(kw)
    sample_rate = opts["sampleRate"] if opts.get("sampleRate") is not None else 1
    sides = opts["sides"] if opts.get("sides") is not None else "one"
    window = opts["window"] if opts.get("window") is not None else "rect"
    two = sides != "one"
    out: list = []
    start, total = 0, len(frames)
    while start < total:
        length = len(frames[start])
        end = start + 1
        while end < total and len(frames[end]) == length:
            end += 1
        batch = end - start
        target = opts["fftSize"] if opts.get("fftSize") is not None else nextPowerOfTwo(length)
        if not isPowerOfTwo(target):
            raise PdspError(_capi.ERR_SIZE_NOT_POW2, f"FFT size must be power of two, got {js_num(target)}")
        target = int(target)
        if target != 1 and window not in _capi.WINDOW_TYPES:
            raise PdspError(_capi.ERR_WINDOW_TYPE, f"Unsupported window type: {window}")
        if sample_rate <= 0:
            raise PdspError(_capi.ERR_SAMPLE_RATE, f"Sample rate must be positive, got {js_num(sample_rate)}")
        bins = target if two else target // 2 + 1
        run = frames[start:end]
        f32 = all(isinstance(f, np.ndarray) and f.dtype == np.float32 for f in run)
        rows = [np.ascontiguousarray(f) if f32 else np.ascontiguousarray(as_f64(f)) for f in run]
        freq = np.empty(bins, dtype=np.float64)
        amp = np.empty((batch, bins), dtype=np.float64)
        ph = np.empty((batch, bins), dtype=np.float64)
        peaks = (_capi.Peak * batch)()
        if f32:
            fp = C.POINTER(C.c_float)
            ptrs = (fp * batch)(*[r.ctypes.data_as(fp) for r in rows])
            fn = lib.pdsp_spectrum_rows_host_f32in
        else:
            dp = C.POINTER(C.c_double)
            ptrs = (dp * batch)(*[dptr(r) for r in rows])
            fn = lib.pdsp_spectrum_rows_host_f64
        check(fn(ptrs, batch, length, float(sample_rate), target, _capi.WINDOW_TYPES.get(window, 0), 1 if two else 0,
                 dptr(freq), dptr(amp), dptr(ph), peaks, None))
        for b in range(batch):
            pk = peaks[b]
            out.append(SpectrumResult(freq if b == 0 else freq.copy(), amp[b], ph[b],
                                      SpectrumPeak(int(pk.index), float(pk.frequency), float(pk.amplitude), float(pk.phase))))
        start = end
    return out

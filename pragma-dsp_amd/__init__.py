"""pragma-dsp_amd -- MI355X (gfx950) engine behind pragma-dsp's spectrum(),
FFT.forward() and Radix2Fft surface.

Host mirror of the reference's three export surfaces that sit on the hot path
(`pragma-dsp/core`, `pragma-dsp/xform/fourier`, `pragma-dsp`), all computing on
the GPU through the C ABI of include/pdsp_hip.h.  Import as `pragma_dsp_amd`
(the repo-root shim maps the hyphenated directory name to a module name).
"""
from ._capi import LIB_PATH, PdspError, lib  # noqa: F401  (fails loudly if the .so is missing)
from .core import ComplexArray, Radix2Fft, createComplexArray, isPowerOfTwo, nextPowerOfTwo  # noqa: F401
from .fourier import (  # noqa: F401
    FFT,
    applyWindow,
    binFrequencies,
    createWindow,
    fftShift,
    fftShiftComplex,
    magnitude,
    phase,
)
from .spectrum import SpectrumPeak, SpectrumResult, spectrum, spectrumBatch  # noqa: F401

__all__ = [
    "ComplexArray", "Radix2Fft", "createComplexArray", "isPowerOfTwo", "nextPowerOfTwo",
    "FFT", "applyWindow", "binFrequencies", "createWindow", "fftShift", "fftShiftComplex",
    "magnitude", "phase", "spectrum", "spectrumBatch", "SpectrumPeak", "SpectrumResult", "PdspError",
]

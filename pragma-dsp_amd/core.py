"""Host mirror of `pragma-dsp/core` (src/core/fft.ts) over the HIP C ABI.

Same names, argument meaning and error texts as the reference so that tests read
like the reference's own: `Radix2Fft(size).forward(x, out?)` etc.  All numerics
run on the GPU through pdsp_fft_transform_host_f64; nothing is computed here.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi
from ._capi import PdspError, check, dptr, lib


def js_num(x) -> str:
    """Format a number the way a JS template literal would (48000, not 48000.0)."""
    if isinstance(x, bool):
        return "true" if x else "false"
    if isinstance(x, (int, np.integer)):
        return str(int(x))
    if isinstance(x, (float, np.floating)):
        f = float(x)
        if f != f:
            return "NaN"
        if f in (float("inf"), float("-inf")):
            return "Infinity" if f > 0 else "-Infinity"
        if f.is_integer() and abs(f) < 1e21:
            return str(int(f))
        return repr(f)
    return str(x)


def as_f64(values, name="input") -> np.ndarray:
    """ArrayLike<number> -> contiguous float64; missing entries (None) read as 0
    like the reference's `input[i] ?? 0` (src/core/fft.ts:112-113)."""
    try:
        return np.ascontiguousarray(values, dtype=np.float64)
    except (TypeError, ValueError):
        return np.ascontiguousarray([0.0 if v is None else float(v) for v in values], dtype=np.float64)


class ComplexArray:
    """`{real: Float64Array, imag: Float64Array}` (src/core/fft.ts:1-4)."""

    __slots__ = ("real", "imag")

    def __init__(self, real, imag):
        self.real = real
        self.imag = imag

    def __iter__(self):
        yield self.real
        yield self.imag

    def __repr__(self):
        return f"ComplexArray(size={len(self.real)})"


def createComplexArray(size: int, fill: float = 0) -> ComplexArray:
    """src/core/fft.ts:6-14 -- `fill` goes to BOTH planes."""
    return ComplexArray(np.full(int(size), float(fill), dtype=np.float64),
                        np.full(int(size), float(fill), dtype=np.float64))


def isPowerOfTwo(n) -> bool:
    """src/core/fft.ts:16; integers only (the reference's int32 coercion lets 2.5 through)."""
    try:
        if float(n) != int(n):
            return False
    except (TypeError, ValueError, OverflowError):
        return False
    return bool(lib.pdsp_is_pow2(int(n)))


def nextPowerOfTwo(n) -> int:
    """src/core/fft.ts:18-23."""
    import math
    return int(lib.pdsp_next_pow2(int(math.ceil(n))))


def _complex_planes(x, what="input"):
    if isinstance(x, ComplexArray):
        return x.real, x.imag
    if isinstance(x, dict):
        return x["real"], x["imag"]
    return x[0], x[1]


class Radix2Fft:
    """src/core/fft.ts:63-152.  A native plan; freed when the object dies (the
    reference has no destroy/close method)."""

    def __init__(self, size, device: int = -1):
        if not isPowerOfTwo(size):
            raise PdspError(_capi.ERR_SIZE_NOT_POW2, f"FFT size must be power of two, got {js_num(size)}")
        handle = C.c_void_p()
        check(lib.pdsp_plan_create(int(size), int(device), C.byref(handle)))
        self._h = handle
        self.size = int(size)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                lib.pdsp_plan_destroy(h)
            except Exception:
                pass
            self._h = None

    # -- the three public methods, fft.ts:77-87 ------------------------------
    def forward(self, input, out: ComplexArray | None = None) -> ComplexArray:
        return self._transform(input, None, out, False)

    def forwardComplex(self, input, out: ComplexArray | None = None) -> ComplexArray:
        re, im = _complex_planes(input)
        return self._transform(re, im, out, False)

    def inverse(self, input, out: ComplexArray | None = None) -> ComplexArray:
        re, im = _complex_planes(input)
        return self._transform(re, im, out, True)

    # -- extensions (as in the JS host, js/core.js): many rows in one call ----------
    def forwardBatch(self, inputs) -> list:
        """Element i equals forward(inputs[i]): the loop of the reference's batch idiom
        (bench/reallife/signals.ts:264-270) as one device batch; the ComplexArrays of a batch are
        views into one buffer per plane."""
        return self._transform_batch(inputs, False, False)

    def forwardComplexBatch(self, inputs) -> list:
        return self._transform_batch(inputs, True, False)

    def inverseBatch(self, inputs) -> list:
        return self._transform_batch(inputs, True, True)

    def _transform_batch(self, inputs, complex_rows: bool, inverse: bool) -> list:
        n, batch = self.size, len(inputs)
        re_rows, im_rows = [], []
        for item in inputs:  # the length checks of fft.ts:95-104, per row
            r, i = _complex_planes(item) if complex_rows else (item, None)
            if len(r) != n:
                raise PdspError(_capi.ERR_INPUT_LENGTH, f"FFT input length {len(r)} != size {n}")
            if i is not None and len(i) != n:
                raise PdspError(_capi.ERR_INPUT_LENGTH, f"FFT input length {len(i)} != size {n}")
            re_rows.append(np.ascontiguousarray(as_f64(r)))
            if complex_rows:
                im_rows.append(np.ascontiguousarray(as_f64(i)))
        ore = np.empty((batch, n), dtype=np.float64)
        oim = np.empty((batch, n), dtype=np.float64)
        if batch:
            dp = C.POINTER(C.c_double)
            rp = (dp * batch)(*[dptr(r) for r in re_rows])  # rows are read where they lie
            ip = (dp * batch)(*[dptr(r) for r in im_rows]) if complex_rows else None
            check(lib.pdsp_fft_transform_rows_host_f64(self._h, batch, n, rp, ip, dptr(ore), dptr(oim), int(bool(inverse))))
        return [ComplexArray(ore[b], oim[b]) for b in range(batch)]

    # -- fft.ts:89-151 -----------------------------------------------------------
    def _transform(self, input_real, input_imag, out, inverse: bool) -> ComplexArray:
        if len(input_real) != self.size:
            raise PdspError(_capi.ERR_INPUT_LENGTH, f"FFT input length {len(input_real)} != size {self.size}")
        if input_imag is not None and len(input_imag) != self.size:
            raise PdspError(_capi.ERR_INPUT_LENGTH, f"FFT input length {len(input_imag)} != size {self.size}")
        re = as_f64(input_real)
        im = as_f64(input_imag) if input_imag is not None else None
        result = out if out is not None else createComplexArray(self.size)
        ore, oim = result.real, result.imag
        direct = (isinstance(ore, np.ndarray) and isinstance(oim, np.ndarray) and ore.dtype == np.float64
                  and oim.dtype == np.float64 and ore.flags.c_contiguous and oim.flags.c_contiguous
                  and len(ore) == self.size and len(oim) == self.size)
        tre = ore if direct else np.empty(self.size, dtype=np.float64)
        tim = oim if direct else np.empty(self.size, dtype=np.float64)
        check(lib.pdsp_fft_transform_host_f64(self._h, 1, self.size, dptr(re), dptr(im), dptr(tre), dptr(tim),
                                              int(bool(inverse))))
        if not direct:
            ore[:] = tre
            oim[:] = tim
        return result  # the same object when `out` was given (chain.test.ts:68-75)

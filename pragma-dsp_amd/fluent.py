"""Device-resident counterpart of the reference's fluent pipeline
(`FluentFFT` in src/xform/fourier-fluent.ts:38-72, `ComplexChain` in
src/fluent/complex.ts:123-331): forward -> element-wise ops -> inverse on
batches of rows that never leave HBM.

Same method names and semantics as the reference's chain: every chainable op
mutates the chain's planes in place and returns the chain; `clone()` copies;
`mag()` / `arg()` are terminal projections; `inverse()` needs the FFT context a
`FluentBatchedFft.forward*()` binds (the reference gates that with a TypeScript
typestate; here it is a run-time `NoFftContext` error), `inverseChecked()`
returns a result record instead of raising.  f32 planes, [..., N] contiguous.
"""
from __future__ import annotations

import torch

from . import _capi, batch
from ._capi import PdspError


def assertNonZero(x) -> None:
    """fluent/complex.ts:88-93."""
    if x == 0:
        raise PdspError(_capi.ERR_BAD_ARG, "Expected non-zero number")


def asNonZero(x):
    """fluent/complex.ts:95-96."""
    return None if x == 0 else x


class DeviceChain:
    def __init__(self, re: torch.Tensor, im: torch.Tensor, plan: "batch.BatchedFft | None" = None):
        if re.shape != im.shape:
            raise PdspError(_capi.ERR_INPUT_LENGTH, "real and imag planes must have the same shape")
        self.re, self.im = re, im
        self._plan = plan

    # -- identity / accessors -------------------------------------------------
    def unwrap(self):
        return self.re, self.im

    @property
    def length(self) -> int:
        return int(self.re.shape[-1])

    def clone(self) -> "DeviceChain":
        return DeviceChain(self.re.clone(), self.im.clone(), self._plan)

    # -- chainable ops: in place, math/complex.ts:26-197 on the device ----------
    def _planes(self):
        return (self.re, self.im)

    @staticmethod
    def _operand(b):
        return b.unwrap() if isinstance(b, DeviceChain) else b

    def scale(self, s) -> "DeviceChain":
        batch.complex_scale(self._planes(), s, out=self._planes())
        return self

    def mul(self, b) -> "DeviceChain":
        batch.complex_mul(self._planes(), self._operand(b), out=self._planes())
        return self

    def mulScalar(self, re, im) -> "DeviceChain":
        batch.complex_mul_scalar(self._planes(), re, im, out=self._planes())
        return self

    def div(self, b) -> "DeviceChain":
        batch.complex_div(self._planes(), self._operand(b), out=self._planes())
        return self

    def divScalar(self, re, im) -> "DeviceChain":
        batch.complex_div_scalar(self._planes(), re, im, out=self._planes())
        return self

    def conj(self) -> "DeviceChain":
        batch.complex_conj(self._planes(), out=self._planes())
        return self

    def add(self, b) -> "DeviceChain":
        batch.complex_add(self._planes(), self._operand(b), out=self._planes())
        return self

    def sub(self, b) -> "DeviceChain":
        batch.complex_sub(self._planes(), self._operand(b), out=self._planes())
        return self

    # -- terminal projections ---------------------------------------------------
    def mag(self) -> torch.Tensor:
        return batch.magnitude(self.re, self.im)

    def arg(self) -> torch.Tensor:
        return batch.phase(self.re, self.im)

    # -- inverse ------------------------------------------------------------------
    def inverse(self, out=None):
        if self._plan is None:
            raise PdspError(_capi.ERR_BAD_ARG, "NoFftContext: the chain was not created by FluentBatchedFft.forward()")
        return self._plan.inverse(self.re, self.im, out=out)

    def inverseChecked(self, out=None) -> dict:
        """fluent/complex.ts:300-318: {'ok': True, 'value': (re, im)} or {'ok': False, 'error': {...}}."""
        if self._plan is None:
            return {"ok": False, "error": {"_tag": "NoFftContext"}}
        try:
            return {"ok": True, "value": self._plan.inverse(self.re, self.im, out=out)}
        except PdspError as e:
            return {"ok": False, "error": {"_tag": "NotInvertible", "reason": str(e)}}


def chain(re: torch.Tensor, im: torch.Tensor) -> DeviceChain:
    """Wrap existing planes without FFT context (fluent/complex.ts:331)."""
    return DeviceChain(re, im)


class FluentBatchedFft:
    """fourier-fluent.ts:38-72 over batch.BatchedFft: forward() returns a chain with the inverse bound."""

    def __init__(self, size, device=None):
        self._fft = batch.BatchedFft(size, device)
        self.size = self._fft.size

    def forward(self, rows: torch.Tensor, out=None) -> DeviceChain:
        re, im = self._fft.forward(rows, None, out=out)
        return DeviceChain(re, im, self._fft)

    def forwardComplex(self, re: torch.Tensor, im: torch.Tensor, out=None) -> DeviceChain:
        ore, oim = self._fft.forward(re, im, out=out)
        return DeviceChain(ore, oim, self._fft)

"""Streaming front-end: the contract of the reference's only streaming
caller, `spectrumStream(frames, opts)` (src/effect/index.ts:190-194: map each frame through
spectrum(), results in order), re-shaped for a GPU: frames are accumulated into device
batches, with two staging slots so the H2D copy and host-side f64->f32 conversion of batch
i+1 overlap the kernel and D2H of batch i.  Plans and windows are reused across batches
(FourierLive's Map<size, FFT> / Map<"type:size", window>, src/effect/index.ts:30-48).

Every result equals `spectrum(frame, opts)` bit for bit: the same kernel variant runs per
row, and findPeak runs on the host over the f64-promoted amplitudes exactly as there.
"""
from __future__ import annotations

import ctypes as C
from typing import Iterable, Iterator

import numpy as np
import torch

from . import _capi
from ._capi import PdspError, check, dptr, lib
from .batch import BatchedFft, _ptr, _stream_ptr
from .core import as_f64, isPowerOfTwo, js_num, nextPowerOfTwo
from .fourier import binFrequencies
from .spectrum import SpectrumPeak, SpectrumResult


class _Slot:
    def __init__(self, frames: int, n: int, bins: int, device, dtype):
        self.h_in = torch.empty((frames, n), dtype=dtype).pin_memory()
        self.d_in = torch.empty((frames, n), dtype=dtype, device=device)
        self.d_out = torch.empty((2, frames, bins), dtype=dtype, device=device)  # amp, phase
        self.h_out = torch.empty((2, frames, bins), dtype=dtype).pin_memory()
        self.done = torch.cuda.Event()
        self.count = 0
        self.in_flight = False


class SpectrumStream:
    """push(frame) -> results that became ready (in input order); flush() -> the rest."""

    def __init__(self, options: dict | None = None, batch_frames: int = 1024, device=None, **kw):
        opts = dict(options or {})
        opts.update(kw)
        self.sample_rate = opts["sampleRate"] if opts.get("sampleRate") is not None else 1
        self.sides = opts["sides"] if opts.get("sides") is not None else "one"
        self.fft_size = opts.get("fftSize")
        self.window = opts["window"] if opts.get("window") is not None else "rect"
        if self.fft_size is not None and not isPowerOfTwo(self.fft_size):
            raise PdspError(_capi.ERR_SIZE_NOT_POW2, f"FFT size must be power of two, got {js_num(self.fft_size)}")
        if self.sample_rate <= 0:
            raise PdspError(_capi.ERR_SAMPLE_RATE, f"Sample rate must be positive, got {js_num(self.sample_rate)}")
        if batch_frames < 1:
            raise ValueError("batch_frames must be >= 1")
        if not torch.cuda.is_available():
            raise PdspError(_capi.ERR_DEVICE, "no HIP device available (the pdsp engine has no CPU fallback)")
        self.batch_frames = int(batch_frames)
        # same arithmetic as spectrum(): the host precision (pdsp_set_host_precision), f64 by default;
        # per size it falls to f32 where the f64 tables do not reach, exactly as
        # pdsp_spectrum_batch_host_f64 does (_dtype_for)
        self.dtype = torch.float64 if lib.pdsp_set_host_precision(0) == 64 else torch.float32
        self.slot_bytes = int(opts.get("slotBytes", 1 << 30))  # cap on one slot's pinned input staging
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self._plans: dict[int, BatchedFft] = {}     # Map<size, FFT>
        self._freqs: dict[int, np.ndarray] = {}
        self._slots: dict[int, list[_Slot]] = {}    # two staging slots per size
        self._copy = torch.cuda.Stream(self.device)
        self._compute = torch.cuda.Stream(self.device)
        self._n = None          # size of the batch being filled
        self._fill = None       # slot being filled
        self._pending: list[tuple[int, _Slot]] = []  # submitted, not yet harvested (oldest first)

    # -- internals -------------------------------------------------------------
    def _size_for(self, length: int) -> int:
        n = self.fft_size if self.fft_size is not None else nextPowerOfTwo(length)
        if n != 1 and self.window not in _capi.WINDOW_TYPES:
            raise PdspError(_capi.ERR_WINDOW_TYPE, f"Unsupported window type: {self.window}")
        return int(n)

    def _bins(self, n: int) -> int:
        return n // 2 + 1 if self.sides == "one" else n

    def _dtype_for(self, n: int):
        """f64 only where the f64 device family covers the size (pdsp_max_size(8)); else f32 -- the
        rule of the one-shot spectrum(), so that every size it accepts streams too."""
        if self.dtype == torch.float64 and n > int(lib.pdsp_max_size(8)):
            return torch.float32
        return self.dtype

    def _frames_per_slot(self, n: int) -> int:
        """batch_frames, clamped so that one slot's pinned input buffer stays under slot_bytes (a slot of
        1024 frames of 2^20 f64 points would pin 8 GiB); at least one frame."""
        per_frame = n * (8 if self._dtype_for(n) == torch.float64 else 4)
        return max(1, min(self.batch_frames, self.slot_bytes // max(per_frame, 1)))

    def _free_slot(self, n: int) -> tuple[_Slot, list]:
        """A staging slot of size n that is neither being filled nor in flight.  When both slots of
        that size are busy the oldest pending batches are harvested -- whatever their size: results
        leave in input order -- until one of them comes free; every harvested result is returned."""
        ready: list = []
        slots = self._slots.setdefault(n, [])
        while True:
            for s in slots:
                if not s.in_flight and s.count == 0:
                    return s, ready
            if len(slots) < 2:
                s = _Slot(self._frames_per_slot(n), n, self._bins(n), self.device, self._dtype_for(n))
                slots.append(s)
                return s, ready
            ready += self._harvest_oldest()  # both busy: wait for the oldest batch in flight

    def _submit(self) -> None:
        n, s = self._n, self._fill
        if s is None or s.count == 0:
            return
        plan = self._plans.get(n)
        if plan is None:
            plan = self._plans[n] = BatchedFft(n, self.device, dtype=self._dtype_for(n))
        win = None if (self.window == "rect" or n == 1) else plan.window(self.window)
        cnt, bins = s.count, self._bins(n)
        with torch.cuda.stream(self._copy):
            s.d_in[:cnt].copy_(s.h_in[:cnt], non_blocking=True)
            uploaded = torch.cuda.Event()
            uploaded.record(self._copy)
        self._compute.wait_event(uploaded)
        with torch.cuda.stream(self._compute):
            check(getattr(lib, "pdsp_spectrum_" + plan._sfx)(plan._h, cnt, _ptr(s.d_in), n, n, _ptr(win), 0 if self.sides == "one" else 1,
                                        _ptr(s.d_out[0]), _ptr(s.d_out[1]), None, _stream_ptr(self.device)))
            s.h_out[:, :cnt].copy_(s.d_out[:, :cnt], non_blocking=True)
            s.done.record(self._compute)
        s.in_flight = True
        self._pending.append((n, s))
        self._fill = None

    def _harvest_oldest(self) -> list[SpectrumResult]:
        n, s = self._pending.pop(0)
        s.done.synchronize()
        freqs = self._freqs.get(n)
        if freqs is None:
            freqs = self._freqs[n] = binFrequencies(n, self.sample_rate, self.sides)
        out = []
        # fresh f64 arrays per batch (the staging slot is reused); each result owns its rows of them
        amp64 = s.h_out[0, :s.count].numpy().astype(np.float64)
        ph64 = s.h_out[1, :s.count].numpy().astype(np.float64)
        for r in range(s.count):
            a, p = amp64[r], ph64[r]
            k = int(lib.pdsp_find_peak_f64(dptr(a), len(a)))  # host findPeak, as in spectrum()
            out.append(SpectrumResult(freqs.copy(), a, p, SpectrumPeak(k, float(freqs[k]), float(a[k]), float(p[k]))))
        s.count = 0
        s.in_flight = False
        return out

    # -- public ------------------------------------------------------------------
    def push(self, frame) -> list[SpectrumResult]:
        x = as_f64(frame)
        n = self._size_for(len(x))
        if not isPowerOfTwo(n):
            raise PdspError(_capi.ERR_SIZE_NOT_POW2, f"FFT size must be power of two, got {js_num(n)}")
        ready: list[SpectrumResult] = []
        if self._fill is not None and n != self._n:  # a different size ends the current batch
            self._submit()
        if self._fill is None:
            self._n = n
            self._fill, r = self._free_slot(n)
            ready += r
        s = self._fill
        row = s.h_in[s.count].numpy()
        used = min(len(x), n)  # buildFrame: truncate or zero-pad (spectrum.ts:36-43)
        row[:used] = x[:used]
        row[used:] = 0.0
        s.count += 1
        if s.count == s.h_in.shape[0]:  # the slot is full (batch_frames, or fewer under the byte budget)
            self._submit()
            # keep one batch in flight: harvest everything older than the newest submission
            while len(self._pending) > 1:
                ready += self._harvest_oldest()
        return ready

    def flush(self) -> list[SpectrumResult]:
        self._submit()
        ready: list[SpectrumResult] = []
        while self._pending:
            ready += self._harvest_oldest()
        return ready


def spectrumStream(frames: Iterable, options: dict | None = None, batch_frames: int = 1024,
                   device=None) -> Iterator[SpectrumResult]:
    """Generator form: yields one SpectrumResult per input frame, in order (an empty input
    yields nothing, test/reallife/effect.test.ts:136-146)."""
    st = SpectrumStream(options, batch_frames, device)
    for f in frames:
        yield from st.push(f)
    yield from st.flush()

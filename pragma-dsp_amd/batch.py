"""Device-resident batched API (an extension: the reference has no batch shape,
only the caller loop of bench/reallife/signals.ts:264-270).  Row b of every call
means exactly `Radix2Fft.forward / forwardComplex / inverse` or the body of
`spectrum()` applied to row b.

torch is used for device memory and streams only; the arithmetic is the HIP
kernels behind include/pdsp_hip.h, reached with raw device pointers.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _capi
from ._capi import PdspError, check, lib
from .core import isPowerOfTwo, js_num
from .fourier import createWindow


def _stream_ptr(device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _ptr(t) -> C.c_void_p:
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class PlanWindow:
    """The plan's own device copy of createWindow(kind, N) (pdsp_plan_window_*): a borrowed device
    pointer, valid while the plan lives.  Passing it as `window` tells the engine which window it is,
    so the kernels that can fuse createWindow do (include/pdsp_hip.h); `.tensor()` gives a torch copy."""

    def __init__(self, plan: "BatchedFft", kind: str, ptr: int):
        self._plan = plan  # keeps the owner alive (the plan caches pointers, not these objects: no cycle)
        self._p = ptr
        self.kind = kind
        self.shape = (plan.size,)
        self.dtype = plan.dtype
        self.device = plan.device

    def data_ptr(self) -> int:
        return self._p

    def tensor(self) -> torch.Tensor:
        return torch.from_numpy(createWindow(self.kind, self.shape[0])).to(self.dtype).to(self.device)


class BatchedFft:
    """One plan, many rows.  Tensors are contiguous, shape [..., N], on the plan's GPU, of the
    plan's dtype: torch.float32 (default; every size up to 16384) or torch.float64 (complex
    transforms up to N = 8192, spectrum up to N = 16384)."""

    def __init__(self, size, device=None, dtype=torch.float32):
        if dtype not in (torch.float32, torch.float64):
            raise PdspError(_capi.ERR_BAD_ARG, f"unsupported dtype {dtype}")
        self.dtype = dtype
        self._sfx = "f32" if dtype == torch.float32 else "f64"
        if not isPowerOfTwo(size):
            raise PdspError(_capi.ERR_SIZE_NOT_POW2, f"FFT size must be power of two, got {js_num(size)}")
        if not torch.cuda.is_available():
            raise PdspError(_capi.ERR_DEVICE, "no HIP device available (the pdsp engine has no CPU fallback)")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.size = int(size)
        handle = C.c_void_p()
        check(lib.pdsp_plan_create(self.size, self.device.index, C.byref(handle)))
        self._h = handle
        self._windows = {}
        self.arena = None  # the allocation behind the planes of the last alloc_planes() call, if any

    def close(self):
        """Destroys the native plan now (also done when the object is collected).  For a size beyond the single-pass
        limit this also hands the engine's scratch planes -- as large as the largest such transform run -- back to
        the device, where the caller's allocator can use them again."""
        h = getattr(self, "_h", None)
        if h:
            try:
                lib.pdsp_plan_destroy(h)
            except Exception:
                pass
            self._h = None
            self._windows = {}

    def __del__(self):
        self.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    # -- helpers -------------------------------------------------------------
    def _check(self, t, name, last=None):
        if t.dtype != self.dtype or not t.is_cuda or not t.is_contiguous():
            raise PdspError(_capi.ERR_BAD_ARG, f"{name} must be a contiguous {self.dtype} tensor on {self.device}")
        if t.device != self.device:
            raise PdspError(_capi.ERR_BAD_ARG, f"{name} is on {t.device}, plan is on {self.device}")
        want = self.size if last is None else last
        if t.shape[-1] != want:
            raise PdspError(_capi.ERR_INPUT_LENGTH, f"FFT input length {t.shape[-1]} != size {want}")

    def _out(self, like, out):
        if out is not None:
            ore, oim = out
            self._check(ore, "out.real")
            self._check(oim, "out.imag")
            return ore, oim
        return torch.empty_like(like), torch.empty_like(like)

    def window(self, kind: str) -> PlanWindow:
        """The plan's device copy of createWindow(kind, N), cached per kind like FourierLive's window
        cache (src/effect/index.ts:39-48).  A window named by kind is known to the engine (fused
        createWindow where a kernel supports it); a caller's own tensor is read as a table."""
        p = self._windows.get(kind)
        if p is None:
            if kind not in _capi.WINDOW_TYPES:
                raise PdspError(_capi.ERR_WINDOW_TYPE, f"Unsupported window type: {kind}")
            out = C.c_void_p()
            check(getattr(lib, "pdsp_plan_window_" + self._sfx)(self._h, _capi.WINDOW_TYPES[kind], C.byref(out)))
            p = self._windows[kind] = out.value
        return PlanWindow(self, kind, p)

    # -- plane layout ----------------------------------------------------------
    # Where the four planes of a transform lie in HBM decides 8-14 % of the N = 4096 kernel's rate (DESIGN section 5,
    # "What the spread of configs[2] is made of"): inside one large allocation the card's address space behaves as
    # regions of 32 GiB, and the launch runs at its upper plateau (83-85 % of the HBM roofline against 76 % for four
    # planes back to back, 71-73 % for unlucky ones) when the two INPUT planes share a region and each OUTPUT plane
    # has a region of its own.  Offsets of 0 | plane | 40 GiB | 80 GiB put the outputs one and two regions beyond the
    # inputs whatever the phase of the allocation against the region grid (40 - plane > 32 > 40 - 32; 80 likewise).
    ARENA_OUT_GIB = (40, 80)

    def alloc_planes(self, batch: int, real_input: bool = False):
        """(re_in, im_in | None, re_out, im_out) for `batch` rows, carved out of ONE allocation of 80 GiB + one plane
        with the layout above -- the card has 288 GB; the gaps stay usable through `.arena` (a uint8 tensor) -- or,
        when that much memory is not free or a plane is below 256 MiB or above 8 GiB, four plain allocations (`.arena` is None).
        The tensors keep the allocation alive."""
        rows, n = int(batch), self.size
        esize = 4 if self.dtype == torch.float32 else 8
        plane = rows * n * esize
        gib = 1 << 30
        arena = None
        if (256 << 20) <= plane <= 8 * gib:  # planes below 256 MiB: nothing to gain, plain allocations
            try:
                free, _total = torch.cuda.mem_get_info(self.device)
                need = self.ARENA_OUT_GIB[1] * gib + plane
                if free >= need + 4 * gib:
                    arena = torch.empty(need, dtype=torch.uint8, device=self.device)
            except RuntimeError:
                arena = None
        self.arena = arena
        if arena is None:
            mk = lambda: torch.empty((rows, n), dtype=self.dtype, device=self.device)  # noqa: E731
            return mk(), (None if real_input else mk()), mk(), mk()
        pad = (plane + 255) & ~255  # the second input plane starts on a 256-byte boundary

        def view(off):
            return arena[off:off + plane].view(self.dtype).view(rows, n)

        return (view(0), (None if real_input else view(pad)), view(self.ARENA_OUT_GIB[0] * gib),
                view(self.ARENA_OUT_GIB[1] * gib))

    # -- transforms ------------------------------------------------------------
    def forward(self, re: torch.Tensor, im: torch.Tensor | None = None, out=None):
        """Rows of Radix2Fft.forward (im None) or .forwardComplex."""
        self._check(re, "input.real")
        batch = re.numel() // self.size if self.size else 0
        ore, oim = self._out(re, out)
        s = _stream_ptr(self.device)
        if im is None:
            check(getattr(lib, "pdsp_fft_forward_real_" + self._sfx)(self._h, batch, _ptr(re), _ptr(ore), _ptr(oim), s))
        else:
            self._check(im, "input.imag")
            check(getattr(lib, "pdsp_fft_forward_complex_" + self._sfx)(self._h, batch, _ptr(re), _ptr(im), _ptr(ore),
                                                                      _ptr(oim), s))
        return ore, oim

    def inverse(self, re: torch.Tensor, im: torch.Tensor, out=None):
        self._check(re, "input.real")
        self._check(im, "input.imag")
        batch = re.numel() // self.size
        ore, oim = self._out(re, out)
        check(getattr(lib, "pdsp_fft_inverse_" + self._sfx)(self._h, batch, _ptr(re), _ptr(im), _ptr(ore), _ptr(oim),
                                                          _stream_ptr(self.device)))
        return ore, oim

    def forward_interleaved(self, z: torch.Tensor, out: torch.Tensor | None = None, inverse: bool = False):
        """forwardComplex (or inverse) on rows of a complex64 / complex128 tensor [..., N] -- the
        interleaved (re, im) layout of I/Q streams.  Single-pass sizes only."""
        want = torch.complex64 if self.dtype == torch.float32 else torch.complex128
        if z.dtype != want or not z.is_cuda or not z.is_contiguous() or z.device != self.device:
            raise PdspError(_capi.ERR_BAD_ARG, f"input must be a contiguous {want} tensor on {self.device}")
        if z.shape[-1] != self.size:
            raise PdspError(_capi.ERR_INPUT_LENGTH, f"FFT input length {z.shape[-1]} != size {self.size}")
        if out is None:
            out = torch.empty_like(z)
        elif out.dtype != want or out.shape != z.shape or not out.is_contiguous() or out.device != self.device:
            raise PdspError(_capi.ERR_BAD_ARG, "out must match the input's dtype, shape and device")
        fn = getattr(lib, ("pdsp_fft_inverse_interleaved_" if inverse else "pdsp_fft_forward_interleaved_") + self._sfx)
        check(fn(self._h, z.numel() // self.size, _ptr(z), _ptr(out), _stream_ptr(self.device)))
        return out

    def inverse_interleaved(self, z: torch.Tensor, out: torch.Tensor | None = None):
        return self.forward_interleaved(z, out, inverse=True)

    def spectrum(self, frames: torch.Tensor, window="rect", sides: str = "one", want_phase: bool = False,
                 want_peak: bool = False, out=None):
        """Rows of spectrum()'s body: frames [..., L] (L <= N zero-padded, L > N
        truncated: spectrum.ts:36-43) -> amplitude [..., bins] (+ phase, + peak bin)."""
        if frames.dtype != self.dtype or not frames.is_cuda or not frames.is_contiguous():
            raise PdspError(_capi.ERR_BAD_ARG, f"frames must be a contiguous {self.dtype} CUDA tensor")
        length = frames.shape[-1]
        batch = frames.numel() // length if length else 0
        two = sides != "one"
        bins = self.size if two else self.size // 2 + 1
        if isinstance(window, str):
            if self.size != 1 and window not in _capi.WINDOW_TYPES:
                raise PdspError(_capi.ERR_WINDOW_TYPE, f"Unsupported window type: {window}")
            win = None if (window == "rect" or self.size == 1) else self.window(window)
        else:
            win = window
            if win is not None:
                if win.shape[-1] != self.size:
                    raise PdspError(_capi.ERR_WINDOW_LENGTH, "Window length must match input length.")
                if not isinstance(win, PlanWindow):
                    self._check(win, "window")
        shape = tuple(frames.shape[:-1])
        amp = out if out is not None else torch.empty(shape + (bins,), dtype=self.dtype, device=self.device)
        ph = torch.empty(shape + (bins,), dtype=self.dtype, device=self.device) if want_phase else None
        pk = torch.empty(shape, dtype=torch.int32, device=self.device) if want_peak else None
        check(getattr(lib, "pdsp_spectrum_" + self._sfx)(self._h, batch, _ptr(frames), min(length, self.size), length, _ptr(win),
                                    1 if two else 0, _ptr(amp), _ptr(ph), _ptr(pk), _stream_ptr(self.device)))
        return amp, ph, pk


    def stft(self, signal: torch.Tensor, hop: int, window="hann", sides: str = "one", want_phase: bool = False,
             want_peak: bool = False):
        """Short-time transform of one contiguous 1-D signal: frame b = signal[b*hop : b*hop + N], the
        body of spectrum() on every frame.  The frames are never materialised -- the kernels read the
        signal with row stride `hop`, so overlapping frames cost no extra HBM footprint.  Returns
        (amplitude [frames, bins], phase or None, peak bin or None)."""
        if signal.dim() != 1 or signal.dtype != self.dtype or not signal.is_cuda or not signal.is_contiguous():
            raise PdspError(_capi.ERR_BAD_ARG, f"signal must be a contiguous 1-D {self.dtype} CUDA tensor")
        if hop < 1:
            raise PdspError(_capi.ERR_BAD_ARG, f"hop must be >= 1, got {hop}")
        n = self.size
        if signal.numel() < n:
            raise PdspError(_capi.ERR_INPUT_LENGTH, f"signal length {signal.numel()} is shorter than one frame ({n})")
        frames = 1 + (signal.numel() - n) // hop
        two = sides != "one"
        bins = n if two else n // 2 + 1
        if self.size != 1 and window not in _capi.WINDOW_TYPES:
            raise PdspError(_capi.ERR_WINDOW_TYPE, f"Unsupported window type: {window}")
        win = None if (window == "rect" or n == 1) else self.window(window)
        amp = torch.empty((frames, bins), dtype=self.dtype, device=self.device)
        ph = torch.empty((frames, bins), dtype=self.dtype, device=self.device) if want_phase else None
        pk = torch.empty((frames,), dtype=torch.int32, device=self.device) if want_peak else None
        check(getattr(lib, "pdsp_spectrum_" + self._sfx)(self._h, frames, _ptr(signal), n, int(hop), _ptr(win), 1 if two else 0,
                                                         _ptr(amp), _ptr(ph), _ptr(pk), _stream_ptr(self.device)))
        return amp, ph, pk

    def spectrum_peaks(self, frames: torch.Tensor, window="rect", sides: str = "one", sample_rate: float = 1.0,
                       want_amp: bool = False, want_phase: bool = False):
        """Rows of the whole spectrum() tail on the device: one SpectrumPeak per frame
        (findPeak fused into the kernel).  Returns (index int32 [...], frequency, amplitude,
        phase float32 [...], amp-or-None, phase-or-None); with want_amp=False only 16 bytes
        per frame leave the kernel."""
        if self.dtype != torch.float32:
            raise PdspError(_capi.ERR_BAD_ARG, "spectrum_peaks is f32 only (16-byte f32 records)")
        if frames.dtype != torch.float32 or not frames.is_cuda or not frames.is_contiguous():
            raise PdspError(_capi.ERR_BAD_ARG, "frames must be a contiguous float32 CUDA tensor")
        if sample_rate <= 0:
            raise PdspError(_capi.ERR_SAMPLE_RATE, f"Sample rate must be positive, got {js_num(sample_rate)}")
        length = frames.shape[-1]
        batch = frames.numel() // length if length else 0
        two = sides != "one"
        bins = self.size if two else self.size // 2 + 1
        if isinstance(window, str):
            if self.size != 1 and window not in _capi.WINDOW_TYPES:
                raise PdspError(_capi.ERR_WINDOW_TYPE, f"Unsupported window type: {window}")
            win = None if (window == "rect" or self.size == 1) else self.window(window)
        else:
            win = window
        shape = tuple(frames.shape[:-1])
        amp = torch.empty(shape + (bins,), dtype=torch.float32, device=self.device) if (want_amp or want_phase) else None
        ph = torch.empty(shape + (bins,), dtype=torch.float32, device=self.device) if want_phase else None
        rec = torch.empty(shape + (4,), dtype=torch.int32, device=self.device)  # pdsp_peak32 records
        check(lib.pdsp_spectrum_peaks_f32(self._h, batch, _ptr(frames), min(length, self.size), length, _ptr(win),
                                          1 if two else 0, float(sample_rate), _ptr(amp), _ptr(ph), _ptr(rec),
                                          _stream_ptr(self.device)))
        f = rec.view(torch.float32)
        return rec[..., 0], f[..., 1], f[..., 2], f[..., 3], amp, ph


# -- stand-alone element-wise device helpers ----------------------------------

def apply_window(frames: torch.Tensor, window: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    if frames.shape[-1] != window.shape[-1]:
        raise PdspError(_capi.ERR_WINDOW_LENGTH, "Window length must match input length.")
    out = torch.empty_like(frames) if out is None else out
    n = frames.shape[-1]
    check(lib.pdsp_apply_window_f32(frames.numel() // n if n else 0, n, _ptr(frames), _ptr(window), _ptr(out),
                                    _stream_ptr(frames.device)))
    return out


def magnitude(re: torch.Tensor, im: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    out = torch.empty_like(re) if out is None else out
    check(lib.pdsp_magnitude_f32(re.numel(), _ptr(re), _ptr(im), _ptr(out), _stream_ptr(re.device)))
    return out


def phase(re: torch.Tensor, im: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    out = torch.empty_like(re) if out is None else out
    check(lib.pdsp_phase_f32(re.numel(), _ptr(re), _ptr(im), _ptr(out), _stream_ptr(re.device)))
    return out


# -- element-wise complex vector arithmetic (src/math/complex.ts) on device rows ------

def _complex_op(name, a, b=None, s_re=0.0, s_im=0.0, out=None):
    are, aim = a
    ore, oim = (torch.empty_like(are), torch.empty_like(aim)) if out is None else out
    bre = bim = None
    b_len = 0
    if b is not None:
        bre, bim = b
        b_len = bre.numel()
        if are.numel() % max(b_len, 1) != 0:
            raise PdspError(_capi.ERR_BAD_ARG, f"second operand length {b_len} must divide {are.numel()}")
    check(lib.pdsp_complex_op_f32(_capi.COMPLEX_OPS[name], are.numel(), _ptr(are), _ptr(aim), _ptr(bre), _ptr(bim),
                                  b_len, float(s_re), float(s_im), _ptr(ore), _ptr(oim), _stream_ptr(are.device)))
    return ore, oim


def complex_add(a, b, out=None):
    return _complex_op("add", a, b, out=out)


def complex_sub(a, b, out=None):
    return _complex_op("sub", a, b, out=out)


def complex_mul(a, b, out=None):
    """Hadamard product; `b` may be one row broadcast over the rows of `a`."""
    return _complex_op("mul", a, b, out=out)


def complex_div(a, b, out=None):
    return _complex_op("div", a, b, out=out)


def complex_conj(a, out=None):
    return _complex_op("conj", a, out=out)


def complex_scale(a, s, out=None):
    return _complex_op("scale", a, s_re=s, out=out)


def complex_mul_scalar(a, re, im, out=None):
    return _complex_op("mulScalar", a, s_re=re, s_im=im, out=out)


def complex_div_scalar(a, re, im, out=None):
    """complex.ts:176-186: multiply by the reciprocal computed on the host."""
    denom = re * re + im * im
    return _complex_op("mulScalar", a, s_re=re / denom, s_im=-im / denom, out=out)

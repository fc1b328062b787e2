"""ctypes binding of include/pdsp_hip.h (pragma-dsp_amd/csrc/libpdsp_hip.so).

This is the only door to compute: there is no Python/NumPy fallback.  If the
library has not been built, importing fails loudly; if there is no GPU, every
compute entry point raises PdspError(PDSP_ERR_DEVICE).
"""
from __future__ import annotations

import ctypes as C
import os

# ONE HIP runtime per process.  PyTorch-ROCm ships its own libamdhip64.so (SONAME
# libamdhip64.so.7) and pulls it in by file name; libpdsp_hip.so needs "libamdhip64.so.7".
# Loaded after torch, ours resolves onto torch's already-mapped copy by SONAME.  Loaded
# BEFORE torch, the system copy is mapped first and torch then maps a second runtime: device
# pointers would still work, but streams/events would not be shared, and whichever runtime
# initialises second can fail ("no HIP device").  So torch, when present, is imported first.
try:
    import torch  # noqa: F401
except ImportError:  # pure drop-in use without torch: the system runtime is the only one
    pass

_HERE = os.path.dirname(os.path.abspath(__file__))
# PDSP_LIB_PATH: another build of the same library (development A/B runs of compile-time kernel options)
LIB_PATH = os.environ.get("PDSP_LIB_PATH") or os.path.join(_HERE, "csrc", "libpdsp_hip.so")

# pdsp_status
OK = 0
ERR_SIZE_NOT_POW2 = 1
ERR_INPUT_LENGTH = 2
ERR_WINDOW_SIZE = 3
ERR_WINDOW_LENGTH = 4
ERR_WINDOW_TYPE = 5
ERR_FFT_SIZE = 6
ERR_SAMPLE_RATE = 7
ERR_UNSUPPORTED_SIZE = 8
ERR_BAD_ARG = 9
ERR_DEVICE = 10

WINDOW_TYPES = {"rect": 0, "hann": 1, "hamming": 2, "blackman": 3}
SIDES = {"one": 0, "two": 1}
COMPLEX_OPS = {"add": 0, "sub": 1, "mul": 2, "div": 3, "conj": 4, "scale": 5, "mulScalar": 6}


class PdspError(Exception):
    """Raised with the reference's message text (the JS API throws `Error(message)`)."""

    def __init__(self, code: int, message: str):
        super().__init__(message)
        self.code = code


class Peak(C.Structure):
    _fields_ = [("index", C.c_int32), ("frequency", C.c_double),
                ("amplitude", C.c_double), ("phase", C.c_double)]


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C pragma-dsp_amd/csrc`).  pragma-dsp_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    ll, i32, dbl, vp = C.c_longlong, C.c_int, C.c_double, C.c_void_p
    dp = C.POINTER(C.c_double)
    sigs = {
        "pdsp_version": ([], i32),
        "pdsp_last_error": ([], C.c_char_p),
        "pdsp_device_count": ([], i32),
        "pdsp_max_size": ([i32], i32),
        "pdsp_set_host_precision": ([i32], i32),
        "pdsp_set_split16k": ([i32], i32),
        "pdsp_set_staged_small": ([i32], i32),
        "pdsp_set_fused_window": ([i32], i32),
        "pdsp_set_twopass": ([i32], i32),
        "pdsp_set_real_packed": ([i32], i32),
        "pdsp_plan_window_f32": ([vp, i32, C.POINTER(vp)], i32),
        "pdsp_plan_window_f64": ([vp, i32, C.POINTER(vp)], i32),
        "pdsp_is_pow2": ([ll], i32),
        "pdsp_next_pow2": ([ll], ll),
        "pdsp_window_make": ([i32, ll, dp], i32),
        "pdsp_bin_frequencies": ([ll, dbl, i32, dp, C.POINTER(ll)], i32),
        "pdsp_fft_shift_f64": ([dp, ll, dp], i32),
        "pdsp_find_peak_f64": ([dp, ll], ll),
        "pdsp_plan_create": ([ll, i32, C.POINTER(vp)], i32),
        "pdsp_plan_destroy": ([vp], i32),
        "pdsp_plan_size": ([vp], ll),
        "pdsp_plan_cache_clear": ([], i32),
        "pdsp_plan_device": ([vp], i32),
        "pdsp_planes_alloc": ([vp, ll, i32, i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp),
                               C.POINTER(C.c_ulonglong)], i32),
        "pdsp_planes_free": ([vp], i32),
        "pdsp_fft_forward_real_f32": ([vp, ll, vp, vp, vp, vp], i32),
        "pdsp_fft_forward_complex_f32": ([vp, ll, vp, vp, vp, vp, vp], i32),
        "pdsp_fft_inverse_f32": ([vp, ll, vp, vp, vp, vp, vp], i32),
        "pdsp_fft_forward_interleaved_f32": ([vp, ll, vp, vp, vp], i32),
        "pdsp_fft_inverse_interleaved_f32": ([vp, ll, vp, vp, vp], i32),
        "pdsp_apply_window_f32": ([ll, ll, vp, vp, vp, vp], i32),
        "pdsp_magnitude_f32": ([ll, vp, vp, vp, vp], i32),
        "pdsp_phase_f32": ([ll, vp, vp, vp, vp], i32),
        "pdsp_complex_op_f32": ([i32, ll, vp, vp, vp, vp, ll, dbl, dbl, vp, vp, vp], i32),
        "pdsp_spectrum_f32": ([vp, ll, vp, ll, ll, vp, i32, vp, vp, vp, vp], i32),
        "pdsp_fft_forward_real_f64": ([vp, ll, vp, vp, vp, vp], i32),
        "pdsp_fft_forward_complex_f64": ([vp, ll, vp, vp, vp, vp, vp], i32),
        "pdsp_fft_inverse_f64": ([vp, ll, vp, vp, vp, vp, vp], i32),
        "pdsp_fft_forward_interleaved_f64": ([vp, ll, vp, vp, vp], i32),
        "pdsp_fft_inverse_interleaved_f64": ([vp, ll, vp, vp, vp], i32),
        "pdsp_apply_window_f64": ([ll, ll, vp, vp, vp, vp], i32),
        "pdsp_magnitude_f64": ([ll, vp, vp, vp, vp], i32),
        "pdsp_phase_f64": ([ll, vp, vp, vp, vp], i32),
        "pdsp_spectrum_f64": ([vp, ll, vp, ll, ll, vp, i32, vp, vp, vp, vp], i32),
        "pdsp_spectrum_peaks_f32": ([vp, ll, vp, ll, ll, vp, i32, dbl, vp, vp, vp, vp], i32),
        "pdsp_fft_transform_host_f64": ([vp, ll, ll, dp, dp, dp, dp, i32], i32),
        "pdsp_fft_transform_rows_host_f64": ([vp, ll, ll, C.POINTER(dp), C.POINTER(dp), dp, dp, i32], i32),
        "pdsp_apply_window_host_f64": ([dp, ll, dp, ll, dp], i32),
        "pdsp_magnitude_host_f64": ([dp, dp, ll, dp], i32),
        "pdsp_phase_host_f64": ([dp, dp, ll, dp], i32),
        "pdsp_spectrum_host_f64": ([dp, ll, dbl, ll, i32, i32, dp, dp, dp, C.POINTER(Peak), C.POINTER(ll)], i32),
        "pdsp_spectrum_batch_host_f64": ([dp, ll, ll, dbl, ll, i32, i32, dp, dp, dp, C.POINTER(Peak), C.POINTER(ll)], i32),
        "pdsp_spectrum_rows_host_f64": ([C.POINTER(dp), ll, ll, dbl, ll, i32, i32, dp, dp, dp, C.POINTER(Peak), C.POINTER(ll)], i32),
        "pdsp_spectrum_rows_host_f32in": ([C.POINTER(C.POINTER(C.c_float)), ll, ll, dbl, ll, i32, i32, dp, dp, dp, C.POINTER(Peak), C.POINTER(ll)], i32),
    }
    for name, (args, res) in sigs.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.argtypes = args
        fn.restype = res
    lib._pdsp_symbols = tuple(sigs)
    return lib


lib = _load()


def check(rc: int) -> None:
    if rc != OK:
        raise PdspError(rc, lib.pdsp_last_error().decode("utf-8", "replace"))


def dptr(a):
    """numpy float64 array -> double* (None passes NULL)."""
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None

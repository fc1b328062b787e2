"""CPU oracle for the pragma-dsp hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  See oracle/pdsp_oracle.c for the restated algorithm and the
reference lines each function follows.
"""
from .oracle import (  # noqa: F401
    Plan,
    WINDOW_TYPES,
    apply_window,
    bin_frequencies,
    build,
    complex_op,
    create_window,
    fft_shift,
    find_peak,
    is_pow2,
    magnitude,
    next_pow2,
    phase,
    scale_amplitude,
    spectrum,
)

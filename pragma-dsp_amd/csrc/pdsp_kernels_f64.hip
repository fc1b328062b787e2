// Kernel unit: every dispatcher in f64 (the reference's own precision and the default arithmetic of the host
// drop-in).  See pdsp_internal.h.
#include "pdsp_dispatch.inc"

namespace pdsp_host {
template int run_complex<double>(const pdsp_plan *, long long, const double *, const double *, double *, double *, double,
                                 hipStream_t);
template int run_interleaved<double>(const pdsp_plan *, long long, const double *, double *, bool, hipStream_t);
template int spectrum_impl<double>(const pdsp_plan *, long long, const double *, long long, long long, const double *, int,
                                   double *, double *, int32_t *, pdsp_peak32 *, double, hipStream_t);
template int apply_window_dev<double>(long long, long long, const double *, const double *, double *, hipStream_t);
template int polar_dev<double, false>(long long, const double *, const double *, double *, hipStream_t);
template int polar_dev<double, true>(long long, const double *, const double *, double *, hipStream_t);
}  // namespace pdsp_host

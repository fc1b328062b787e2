// Kernel unit: the fused f32 spectrum() body (buildFrame -> window -> FFT -> amplitude / phase / findPeak) for every
// size and call shape.  See pdsp_internal.h.
#include "pdsp_dispatch.inc"

namespace pdsp_host {
template int spectrum_impl<float>(const pdsp_plan *, long long, const float *, long long, long long, const float *, int,
                                  float *, float *, int32_t *, pdsp_peak32 *, double, hipStream_t);
}  // namespace pdsp_host

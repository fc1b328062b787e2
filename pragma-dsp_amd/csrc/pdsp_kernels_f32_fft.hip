// Kernel unit: the f32 transforms (rows of planar / interleaved complex points, every size and call shape) and
// the element-wise f32 kernels.  See pdsp_internal.h.
#include "pdsp_dispatch.inc"

namespace pdsp_host {
template int run_complex<float>(const pdsp_plan *, long long, const float *, const float *, float *, float *, float,
                                hipStream_t);
template int run_interleaved<float>(const pdsp_plan *, long long, const float *, float *, bool, hipStream_t);
template int apply_window_dev<float>(long long, long long, const float *, const float *, float *, hipStream_t);
template int polar_dev<float, false>(long long, const float *, const float *, float *, hipStream_t);
template int polar_dev<float, true>(long long, const float *, const float *, float *, hipStream_t);
int complex_op_f32(int op, long long count, const float *are, const float *aim, const float *bre, const float *bim,
                   long long b_len, float sre, float sim, float *ore, float *oim, hipStream_t s) {
  return complex_op_f32_switch(op, count, are, aim, bre, bim, b_len, sre, sim, ore, oim, s);
}
}  // namespace pdsp_host

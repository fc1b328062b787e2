// pdsp_internal.h -- what the translation units of libpdsp_hip.so share: the plan object and its device tables,
// error reporting, the stream-ordered scratch pool, the development switches, and the DECLARATIONS of the kernel
// dispatchers.  The library is built from four translation units so that (i) the kernels compile in parallel and
// (ii) a change to the host side of the boundary (pdsp_capi.hip: validation, plan tables, caches, staging, the
// chunked host calls, the extern "C" entry points -- no kernel is instantiated there) does not recompile them:
//   pdsp_capi.hip                 host side + extern "C"
//   pdsp_kernels_f32_fft.hip      run_complex<float>, run_interleaved<float>, element-wise f32 kernels
//   pdsp_kernels_f32_spectrum.hip spectrum_impl<float> (fused spectrum kernels, findPeak kernels)
//   pdsp_kernels_f64.hip          every dispatcher for double
// The dispatchers themselves are pdsp_dispatch.inc (templates on the scalar type), explicitly instantiated there.
// Not part of the boundary: nothing outside pragma-dsp_amd/csrc includes this file.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <mutex>
#include <string>

#include "../../include/pdsp_hip.h"
#include "../../include/pdsp_hip_dev.h"
#include "pdsp_fft_kernel.h"

namespace pdsp_host {

// last error text of the calling thread (pdsp_last_error); defined in pdsp_capi.hip
int fail(int code, const char *fmt, ...);

// development switches (tools / tests): 0 routes N = 16384 spectra to spectrum_packed_kernel<13>,
// and 32 <= N <= 256 transforms to the direct kernel instead of fft_staged_kernel
extern int g_split16k;
extern int g_fused_window;
extern int g_twopass;       // pdsp_set_twopass: 2^15 <= N <= 2^18 f32 transforms in two passes (balanced factors)  // pdsp_set_fused_window: plan-owned cosine-sum windows evaluated in the kernel
extern int g_split8k_f32;  // f32 N = 8192 rows on fft_split2_kernel too (A/B: pdsp_set_split16k bit 1)
extern int g_staged_small;
extern int g_real_packed;  // pdsp_set_real_packed: Radix2Fft.forward rows of 512 <= N <= 16384 on fft_real_kernel


#define PDSP_HIP_TRY(expr)                                                              \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess)                                                               \
      return fail(PDSP_ERR_DEVICE, "HIP error %d (%s) at %s", (int)e_, hipGetErrorString(e_), #expr); \
  } while (0)

// Stream-ordered scratch that is handed back on every exit path.
// Stream-ordered scratch planes of the multi-pass paths, from a pool of the engine's own per device whose release
// threshold is unlimited: the device's default pool hands its memory back at every synchronisation, so that a
// caller who synchronises between transforms (every host-f64 call does) paid a fresh 1-2 GiB allocation --
// a trip through the kernel driver, observed to stall for 0.5-1 s on a busy host -- on each call.  Here the planes
// of the largest transform seen stay with the engine until pdsp_plan_cache_clear() trims the pools.
hipMemPool_t scratch_pool();   // defined in pdsp_capi.hip
void trim_scratch_pools();

// Bytes of scratch planes this process has drawn from the pools since they were last trimmed: pdsp_plan_destroy()
// hands the pools' unused memory back to the device when a plan that needs scratch (N beyond the single-pass
// limit) goes away and anything was drawn -- otherwise GiBs of HBM stay pinned where the caller's allocator
// (PyTorch's, say) cannot see them, long after the last large transform.
extern std::atomic<unsigned long long> g_scratch_drawn;

struct StreamScratch {
  void *p = nullptr;
  hipStream_t s;
  explicit StreamScratch(hipStream_t stream) : s(stream) {}
  StreamScratch(const StreamScratch &) = delete;
  StreamScratch &operator=(const StreamScratch &) = delete;
  hipError_t alloc(size_t bytes) {
    g_scratch_drawn += bytes;
    if (hipMemPool_t pool = scratch_pool()) return hipMallocFromPoolAsync(&p, bytes, pool, s);
    return hipMallocAsync(&p, bytes, s);  // no pool of our own on this device: the default one
  }
  ~StreamScratch() {
    if (p) (void)hipFreeAsync(p, s);
  }
};

struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  hipError_t err = hipSuccess;
  explicit DeviceGuard(int dev) {
    err = hipGetDevice(&prev);
    if (err == hipSuccess && dev >= 0 && dev != prev) {
      err = hipSetDevice(dev);
      switched = (err == hipSuccess);
    }
  }
  ~DeviceGuard() {
    if (switched) (void)hipSetDevice(prev);
  }
};

inline int ilog2ll(long long n) {
  int l = 0;
  while ((1LL << l) < n) ++l;
  return l;
}

}  // namespace pdsp_host

// Device tables of one precision.
template <typename T>
struct Tables {
  using T2 = typename pdsp::vec2<T>::type;
  T2 *tw = nullptr;       // inter-pass twiddles of the N-point transform (pdsp_radix.h layout)
  // packed-real spectrum path (N >= 64): radix table of the N/2-point transform and the
  // split twiddles W_N^k, 0 <= k <= N/4
  T2 *tw_half = nullptr;
  T2 *twr = nullptr;
  T2 *tw12 = nullptr;  // N = 16384 only: radix table of the 4096-point sub-transforms (split kernels)
  T2 *tws4 = nullptr;  // rows of 16384 points (log2n2 == 14): W_16384^k, k < 768 (fft_split4_kernel)
  T2 *tws2 = nullptr;  // rows of 8192 points (log2n2 == 13): W_8192^k, k < 256 (fft_split2_kernel; uses tw12 too)
  T *win[4] = {nullptr, nullptr, nullptr, nullptr};  // createWindow(type, N), built on first use
  // N = 16384, f32: per-thread bases and per-q constants of the fused cosine-sum windows
  // (spectrum_dif16k_kernel, WinFused): cos / sin of f*(2 tid + e) and of f*512 q (+ 8192), f = 2 pi / (N - 1)
  float *wf_base = nullptr;
  float *wf_step = nullptr;
  // four-step path (N beyond the single-pass limit): `tw` then belongs to the N2-point rows,
  // N1 = N / N2, and W_N^m = twa[m >> 9] * twb[m & 511]
  int log2n2 = 0;  // log2 of the transform `tw` serves (== log2 N when single-pass)
  int log2n1 = 0;
  T2 *twa = nullptr;
  T2 *twb = nullptr;
  T2 *tw1 = nullptr;  // general four-step path (log2n1 > kMaxLog2N1): radix table of the N1-point rows
  // tile passes (f32): N = product of tp_np balanced factors 2^tp_l[i] (two for 2^15..2^18, three for
  // 2^19..2^27), radix table of each factor's transform
  int tp_np = 0;
  int tp_l[3] = {0, 0, 0};
  T2 *tp_tw[3] = {nullptr, nullptr, nullptr};
  T2 *tw8 = nullptr;  // radix table of the 256-point transform (tile_rows512_kernel's halves of a 512-point factor)
  // the same for the N/2-point transform of the packed-real spectrum path (2^15 <= N <= 2^27): it runs on
  // this plan's twa / twb with doubled exponents (TileGeom::tshift)
  int hp_np = 0;
  int hp_l[3] = {0, 0, 0};
  T2 *hp_tw[3] = {nullptr, nullptr, nullptr};
  float *hp_win = nullptr;  // angle-addition tables of the fused cosine-sum windows (TileGeom::wa ...): wa | wb | wstep | we
  size_t hp_win_a = 0;      // entries (cos, sin pairs) of wa
  void release() {
    if (tw12) (void)hipFree(tw12);
    tw12 = nullptr;
    if (wf_base) (void)hipFree(wf_base);
    if (wf_step) (void)hipFree(wf_step);
    wf_base = wf_step = nullptr;
    if (tws4) (void)hipFree(tws4);
    tws4 = nullptr;
    if (tws2) (void)hipFree(tws2);
    tws2 = nullptr;

    if (tw8) (void)hipFree(tw8);
    tw8 = nullptr;
    for (T2 *&q : tp_tw) {
      if (q) (void)hipFree(q);
      q = nullptr;
    }
    tp_np = 0;
    for (T2 *&q : hp_tw) {
      if (q) (void)hipFree(q);
      q = nullptr;
    }
    hp_np = 0;
    if (hp_win) (void)hipFree(hp_win);
    hp_win = nullptr;
    if (twa) (void)hipFree(twa);
    if (twb) (void)hipFree(twb);
    if (tw1) (void)hipFree(tw1);
    twa = twb = tw1 = nullptr;
    if (tw) (void)hipFree(tw);
    if (tw_half) (void)hipFree(tw_half);
    if (twr) (void)hipFree(twr);
    for (T *&w : win) {
      if (w) (void)hipFree(w);
      w = nullptr;
    }
    tw = tw_half = twr = nullptr;
  }
};

struct pdsp_plan {
  long long n = 0;
  int log2n = 0;
  int device = -1;
  Tables<float> t32;
  Tables<double> t64;  // present when the f64 single-pass kernels take this size
  // host-f64 entry points: one stream + growing staging buffers per plan
  std::mutex mu;
  hipStream_t stream = nullptr;
  void *h_stage = nullptr;  // pinned
  size_t h_bytes = 0;
  void *d_stage = nullptr;
  size_t d_bytes = 0;
  // batched host calls large enough to be cut into chunks (run_chunked): one stream per staging slot
  std::vector<hipStream_t> slot_streams;
};

template <typename T> Tables<T> &tables(pdsp_plan *p);
template <> inline Tables<float> &tables<float>(pdsp_plan *p) { return p->t32; }
template <> inline Tables<double> &tables<double>(pdsp_plan *p) { return p->t64; }
template <typename T> const Tables<T> &tables(const pdsp_plan *p) { return tables<T>(const_cast<pdsp_plan *>(p)); }

// Largest log2 N of the single-pass kernels: (N + N/16) complex values must fit 160 KiB of LDS.
template <typename T> constexpr int max_log2n() { return sizeof(T) == 4 ? pdsp::kMaxLog2N_f32 : pdsp::kMaxLog2N_f64; }

namespace pdsp_host {

// Factors of a three-pass transform of 2^lg points (2^18 < 2^lg <= 2^27): balanced, ascending.  (Tried and
// dropped: a 64-point first factor everywhere -- the widest tiles for the one pass that reads strided -- with
// 512-point factors behind it: 2^22 as 64 * 256 * 256 and 2^24 as 64 * 512 * 512 measured -3 % / +1 % against
// 128 * 128 * 256 and 256^3, the long-frame spectrum -2 ... -5 %: profiles/r02_experiments/sweep_large_factors.log.)
inline void three_factors(int lg, int *l) {
  l[0] = lg / 3, l[1] = (lg - l[0]) / 2, l[2] = lg - l[0] - l[1];
}
constexpr int tile_width(int l) { return l == 6 ? 64 : (l == 9 ? 16 : 32); }

inline int grid_for(long long total) {
  long long b = (total + 255) / 256;
  if (b > 2048) b = 2048;  // grid-stride the rest (256 CUs x 8)
  if (b < 1) b = 1;
  return (int)b;
}

inline int check_plan_batch(const pdsp_plan *plan, long long batch) {
  if (!plan) return fail(PDSP_ERR_BAD_ARG, "plan is null");
  if (batch < 0) return fail(PDSP_ERR_BAD_ARG, "batch must be >= 0, got %lld", batch);
  // grid.x limit; far beyond any HBM-resident batch
  if (batch > 0x7fffffffLL) return fail(PDSP_ERR_BAD_ARG, "batch too large: %lld", batch);
  return PDSP_OK;
}

// ---- kernel dispatchers: defined in pdsp_dispatch.inc, instantiated for float / double in the kernel units -------
// Rows of planar complex points -> rows (forward; the callers swap planes and pass 1/N for the inverse).
template <typename T>
int run_complex(const pdsp_plan *plan, long long batch, const T *re_in, const T *im_in, T *re_out, T *im_out, T scale,
                hipStream_t s);
template <typename T>
int run_interleaved(const pdsp_plan *plan, long long batch, const T *in, T *out, bool inverse, hipStream_t s);
// The batched body of spectrum() (pdsp_spectrum_f32 / _f64 / pdsp_spectrum_peaks_f32).
template <typename T>
int spectrum_impl(const pdsp_plan *plan, long long batch, const T *frames, long long frame_len, long long frame_stride,
                  const T *window, int sides, T *amp_out, T *phase_out, int32_t *peak_idx_out, pdsp_peak32 *peaks_out,
                  double sample_rate, hipStream_t stream);
template <typename T>
int apply_window_dev(long long batch, long long n, const T *in, const T *window, T *out, hipStream_t s);
template <typename T, bool PHASE>
int polar_dev(long long count, const T *re, const T *im, T *out, hipStream_t s);
// pdsp_complex_op_f32 after validation: op is a pdsp_complex_op
int complex_op_f32(int op, long long count, const float *are, const float *aim, const float *bre, const float *bim,
                   long long b_len, float sre, float sim, float *ore, float *oim, hipStream_t s);

}  // namespace pdsp_host

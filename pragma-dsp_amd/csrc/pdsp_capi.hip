// pdsp_capi.hip -- the C ABI of include/pdsp_hip.h: argument validation with the
// reference's error texts, plan objects, kernel dispatch by size, and the
// synchronous host-f64 entry points the JS drop-in binds.
//
// Product path only: nothing here touches oracle/, and there is no CPU fallback --
// without a HIP device every compute entry point fails with PDSP_ERR_DEVICE.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pdsp_hip.h"
#include "../../include/pdsp_hip_dev.h"
#include "pdsp_fft_kernel.h"

namespace {

thread_local std::string g_err;
// development switches (tools / tests): 0 routes N = 16384 spectra to spectrum_packed_kernel<13>,
// and 32 <= N <= 256 transforms to the direct kernel instead of fft_staged_kernel
int g_split16k = 1;
int g_fused_window = 1;
int g_twopass = 1;       // pdsp_set_twopass: 2^15 <= N <= 2^18 f32 transforms in two passes (balanced factors)  // pdsp_set_fused_window: plan-owned cosine-sum windows evaluated in the kernel
int g_split8k_f32 = 0;  // f32 N = 8192 rows on fft_split2_kernel too (A/B: pdsp_set_split16k bit 1)
int g_staged_small = 1;
int g_real_packed = 1;  // pdsp_set_real_packed: Radix2Fft.forward rows of 512 <= N <= 16384 on fft_real_kernel

int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

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
constexpr int kMaxPoolDevices = 64;
hipMemPool_t g_scratch_pool[kMaxPoolDevices] = {};
std::mutex g_scratch_pool_mu;
hipMemPool_t scratch_pool() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxPoolDevices) return nullptr;
  std::lock_guard<std::mutex> lk(g_scratch_pool_mu);
  if (!g_scratch_pool[dev]) {
    hipMemPoolProps props = {};
    props.allocType = hipMemAllocationTypePinned;
    props.location.type = hipMemLocationTypeDevice;
    props.location.id = dev;
    hipMemPool_t pool = nullptr;
    if (hipMemPoolCreate(&pool, &props) != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;
    }
    uint64_t keep = ~0ULL;
    (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
    g_scratch_pool[dev] = pool;
  }
  return g_scratch_pool[dev];
}
void trim_scratch_pools();

// Bytes of scratch planes this process has drawn from the pools since they were last trimmed: pdsp_plan_destroy()
// hands the pools' unused memory back to the device when a plan that needs scratch (N beyond the single-pass
// limit) goes away and anything was drawn -- otherwise GiBs of HBM stay pinned where the caller's allocator
// (PyTorch's, say) cannot see them, long after the last large transform.
std::atomic<unsigned long long> g_scratch_drawn{0};

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

void trim_scratch_pools() {
  std::lock_guard<std::mutex> lk(g_scratch_pool_mu);
  for (hipMemPool_t pool : g_scratch_pool)
    if (pool) (void)hipMemPoolTrimTo(pool, 0);
  g_scratch_drawn = 0;
}

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

int ilog2ll(long long n) {
  int l = 0;
  while ((1LL << l) < n) ++l;
  return l;
}

}  // namespace

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
template <> Tables<float> &tables<float>(pdsp_plan *p) { return p->t32; }
template <> Tables<double> &tables<double>(pdsp_plan *p) { return p->t64; }
template <typename T> const Tables<T> &tables(const pdsp_plan *p) { return tables<T>(const_cast<pdsp_plan *>(p)); }

// Largest log2 N of the single-pass kernels: (N + N/16) complex values must fit 160 KiB of LDS.
template <typename T> constexpr int max_log2n() { return sizeof(T) == 4 ? pdsp::kMaxLog2N_f32 : pdsp::kMaxLog2N_f64; }

namespace {

// Twiddle table for pdsp_radix.h's layout, built in f64 on the host
// (src/core/fft.ts:45-61 builds cos/sin of (-2*pi*k)/m per stage with Math.cos /
// Math.sin; same direct evaluation here, no recurrence), then rounded once.
template <typename T2>
std::vector<T2> build_twiddles(int log2n, int log2e = 4) {
  const pdsp::RadixPlan p = pdsp::make_radix_plan(log2n, log2e);
  std::vector<T2> tw((size_t)(p.twcount > 0 ? p.twcount : 1));
  for (int i = 0; i < p.np; ++i) {
    const int ns = p.ns[i], r = p.r[i];
    if (ns <= 1) continue;
    const double m = (double)ns * (double)r;
    for (int rr = 1; rr < r; ++rr)
      for (int k = 0; k < ns; ++k) {
        const double angle = (-2.0 * M_PI * (double)rr * (double)k) / m;
        T2 w;
        w.x = (decltype(w.x))std::cos(angle);
        w.y = (decltype(w.y))std::sin(angle);
        tw[(size_t)p.twoff[i] + (size_t)(rr - 1) * ns + k] = w;
      }
  }
  return tw;
}

// Factors of a three-pass transform of 2^lg points (2^18 < 2^lg <= 2^27): balanced, ascending.  (Tried and
// dropped: a 64-point first factor everywhere -- the widest tiles for the one pass that reads strided -- with
// 512-point factors behind it: 2^22 as 64 * 256 * 256 and 2^24 as 64 * 512 * 512 measured -3 % / +1 % against
// 128 * 128 * 256 and 256^3, the long-frame spectrum -2 ... -5 %: profiles/r02_experiments/sweep_large_factors.log.)
inline void three_factors(int lg, int *l) {
  l[0] = lg / 3, l[1] = (lg - l[0]) / 2, l[2] = lg - l[0] - l[1];
}

template <typename T, int LOG2N, class LD, class ST>
hipError_t launch_one(const LD &ld, const ST &st, const typename pdsp::vec2<T>::type *tw, long long batch,
                      hipStream_t s) {
  if constexpr (LOG2N > max_log2n<T>()) {
    return hipErrorInvalidValue;  // would not fit LDS; never instantiated
  } else {
    using TR = pdsp::FftTraits<LOG2N>;
    const long long blocks = (batch + TR::ROWS - 1) / TR::ROWS;
    hipLaunchKernelGGL((pdsp::fft_stockham_kernel<T, LOG2N, LD, ST>), dim3((unsigned)blocks), dim3(TR::WG), 0, s, ld,
                       st, tw, batch);
    return hipGetLastError();
  }
}

template <typename T, class LD, class ST>
hipError_t launch_fft(int log2n, const LD &ld, const ST &st, const typename pdsp::vec2<T>::type *tw, long long batch,
                      hipStream_t s) {
  switch (log2n) {
#define PDSP_CASE(L) \
  case L:            \
    return launch_one<T, L>(ld, st, tw, batch, s);
    PDSP_CASE(0) PDSP_CASE(1) PDSP_CASE(2) PDSP_CASE(3) PDSP_CASE(4) PDSP_CASE(5) PDSP_CASE(6) PDSP_CASE(7)
    PDSP_CASE(8) PDSP_CASE(9) PDSP_CASE(10) PDSP_CASE(11) PDSP_CASE(12) PDSP_CASE(13) PDSP_CASE(14)
#undef PDSP_CASE
    default:
      return hipErrorInvalidValue;
  }
}

// The same for N <= 32 only (spectrum() of frames below the packed-real path's sizes: LoadFrameWindowed /
// StoreAmplitude are not instantiated for the sizes that never take them).
template <typename T, class LD, class ST>
hipError_t launch_fft_small(int log2n, const LD &ld, const ST &st, const typename pdsp::vec2<T>::type *tw, long long batch,
                            hipStream_t s) {
  switch (log2n) {
#define PDSP_CASE(L) \
  case L:            \
    return launch_one<T, L>(ld, st, tw, batch, s);
    PDSP_CASE(0) PDSP_CASE(1) PDSP_CASE(2) PDSP_CASE(3) PDSP_CASE(4) PDSP_CASE(5)
#undef PDSP_CASE
    default:
      return hipErrorInvalidValue;
  }
}

// Rows of planar complex points: N = 16384 (f32) goes to fft_split4_kernel when the input planes
// allow 16-byte loads, everything else to the single-pass kernel of its size.
template <typename T, class LD, class ST>
hipError_t launch_rows(const Tables<T> &t, int log2n, const LD &ld, const ST &st, long long batch, hipStream_t s,
                       bool aligned16) {
  if constexpr (sizeof(T) == 4) {
    if (log2n == 14 && g_split16k && aligned16 && t.tws4 && t.tw12) {
      hipLaunchKernelGGL((pdsp::fft_split4_kernel<T, 12, LD, ST>), dim3((unsigned)batch), dim3(256), 0, s, ld, st,
                         t.tw12, t.tws4, batch);
      return hipGetLastError();
    }
  }
  // N = 8192: measured on one box, f64 C2C 4.35 -> 5.80 TB/s, f64 real-in 4.0 -> 5.6, f32 real-in 5.4 -> 5.7,
  // f32 C2C a wash (stays on the single-pass kernel)
  // (f64 real rows run on fft_real_kernel; their LoadReal form of this kernel spilled 37 registers and is not built)
  if constexpr (!(sizeof(T) == 8 && !LD::kHasIm)) {
    if (log2n == 13 && (sizeof(T) == 8 || !LD::kHasIm || g_split8k_f32) && g_split16k && aligned16 && t.tws2 && t.tw12) {
      hipLaunchKernelGGL((pdsp::fft_split2_kernel<T, LD, ST>), dim3((unsigned)batch), dim3(256), 0, s, ld, st, t.tw12,
                         t.tws2, batch);
      return hipGetLastError();
    }
  }
  return launch_fft<T>(log2n, ld, st, t.tw, batch, s);
}


template <typename T, int LOG2M>
hipError_t launch_packed_one(bool fast, const T *frames, const T *win, int wmode, pdsp::WinFused wf, long long frame_len,
                             long long stride, const typename pdsp::vec2<T>::type *tw,
                             const typename pdsp::vec2<T>::type *twr, T *amp, T *ph, int two_sided, T s_edge, T s_mid,
                             pdsp::PeakRec *peaks, T freq_scale, long long batch, hipStream_t s) {
  using TR = pdsp::FftTraits<LOG2M, pdsp::packed_log2e(LOG2M)>;
  const long long ngroups = (batch + TR::ROWS - 1) / TR::ROWS;
  const dim3 block(TR::WG);
#define PDSP_LAUNCH(F, W, P)                                                                                  \
  hipLaunchKernelGGL((pdsp::spectrum_packed_kernel<T, LOG2M, F, W, P>), dim3((unsigned)ngroups), block, 0, s, \
                     frames, win, wf, frame_len, stride, tw, twr, amp, ph, two_sided, s_edge, s_mid, peaks, freq_scale, \
                     batch)
#define PDSP_LAUNCH_FW(F, W)                            \
  do {                                                  \
    if constexpr (sizeof(T) == 4) {                     \
      if (peaks) PDSP_LAUNCH(F, W, true);               \
      else PDSP_LAUNCH(F, W, false);                    \
    } else {                                            \
      PDSP_LAUNCH(F, W, false); /* fused peaks: f32 only */ \
    }                                                   \
  } while (0)
  // fused cosine-sum windows (wmode 2 / 3): whole f32 frames of N = 1024 ... 8192 (the sizes whose plans
  // carry the angle-addition tables); everything else reads the window as a table
  if constexpr (sizeof(T) == 4 && LOG2M >= 9 && LOG2M <= 12) {
    if (fast && wmode == 2) {
      PDSP_LAUNCH_FW(true, 2);
      return hipGetLastError();
    }
    if (fast && wmode == 3) {
      PDSP_LAUNCH_FW(true, 3);
      return hipGetLastError();
    }
  }
  if (fast && win) PDSP_LAUNCH_FW(true, 1);
  else if (fast) PDSP_LAUNCH_FW(true, 0);
  else if (win) PDSP_LAUNCH_FW(false, 1);
  else PDSP_LAUNCH_FW(false, 0);
#undef PDSP_LAUNCH_FW
#undef PDSP_LAUNCH
  return hipGetLastError();
}

template <typename T, class... A>
hipError_t launch_packed(int log2m, A... a) {
  switch (log2m) {
#define PDSP_CASE(L) \
  case L:            \
    return launch_packed_one<T, L>(a...);
    PDSP_CASE(5) PDSP_CASE(6) PDSP_CASE(7) PDSP_CASE(8) PDSP_CASE(9) PDSP_CASE(10) PDSP_CASE(11) PDSP_CASE(12)
    PDSP_CASE(13)
#undef PDSP_CASE
    default:
      return hipErrorInvalidValue;
  }
}

template <typename T>
hipError_t launch_real(int log2m, const T *x, T *ore, T *oim, T scale, const typename pdsp::vec2<T>::type *tw,
                       const typename pdsp::vec2<T>::type *twr, long long batch, hipStream_t s) {
  switch (log2m) {
#define PDSP_CASE(L)                                                                                              \
  case L: {                                                                                                       \
    using TR = pdsp::FftTraits<L, 4>;                                                                             \
    hipLaunchKernelGGL((pdsp::fft_real_kernel<T, L>), dim3((unsigned)((batch + TR::ROWS - 1) / TR::ROWS)), dim3(TR::WG), \
                       0, s, x, ore, oim, scale, tw, twr, batch);                                                 \
    return hipGetLastError();                                                                                     \
  }
    PDSP_CASE(12) PDSP_CASE(13)
#undef PDSP_CASE
    default:
      return hipErrorInvalidValue;
  }
}

int check_plan_batch(const pdsp_plan *plan, long long batch) {
  if (!plan) return fail(PDSP_ERR_BAD_ARG, "plan is null");
  if (batch < 0) return fail(PDSP_ERR_BAD_ARG, "batch must be >= 0, got %lld", batch);
  // grid.x limit; far beyond any HBM-resident batch
  if (batch > 0x7fffffffLL) return fail(PDSP_ERR_BAD_ARG, "batch too large: %lld", batch);
  return PDSP_OK;
}

// Four-step transform of `batch` rows of N = N1*N2 points into scratch planes (pass A + B);
// the caller runs pass C.  REAL rows may carry a window and be shorter than N.
template <typename T>
int fourstep_ab(const pdsp_plan *plan, long long batch, const T *re_in, const T *im_in, const T *win,
                long long in_stride, long long frame_len, T *s_re, T *s_im, hipStream_t s) {
  const Tables<T> &t = tables<T>(plan);
  const int n2 = 1 << t.log2n2;
  const long long blocks = batch * (n2 / 256);
  if (blocks > 0x7fffffffLL) return fail(PDSP_ERR_BAD_ARG, "batch too large: %lld", batch);
  const pdsp::cx<T> *twa = reinterpret_cast<const pdsp::cx<T> *>(t.twa);
  const pdsp::cx<T> *twb = reinterpret_cast<const pdsp::cx<T> *>(t.twb);
#define PDSP_COLS(L, R, W)                                                                                        \
  hipLaunchKernelGGL((pdsp::fourstep_cols_kernel<T, L, R, W>), dim3((unsigned)blocks), dim3(256), 0, s, re_in, im_in, \
                     win, s_re, s_im, twa, twb, n2, in_stride, frame_len, batch)
#define PDSP_COLS_L(L)                     \
  do {                                     \
    if (im_in) PDSP_COLS(L, false, false); \
    else if (win) PDSP_COLS(L, true, true); \
    else PDSP_COLS(L, true, false);        \
  } while (0)
  switch (t.log2n1) {
    case 1: PDSP_COLS_L(1); break;
    case 2: PDSP_COLS_L(2); break;
    case 3: PDSP_COLS_L(3); break;
    case 4: PDSP_COLS_L(4); break;
    default: return fail(PDSP_ERR_UNSUPPORTED_SIZE, "unsupported four-step split");
  }
#undef PDSP_COLS_L
#undef PDSP_COLS
  PDSP_HIP_TRY(hipGetLastError());
  // pass B: the N1 * batch rows of N2 points, in place (each workgroup loads its row before it stores)
  pdsp::LoadComplex<T> ld{s_re, s_im, n2};
  pdsp::StoreComplex<T> st{s_re, s_im, n2, T(1)};
  PDSP_HIP_TRY(launch_rows<T>(t, t.log2n2, ld, st, batch << t.log2n1, s, true));  // scratch planes are aligned
  return PDSP_OK;
}

template <typename T, int MODE>
int fourstep_c(const pdsp_plan *plan, long long batch, const T *s_re, const T *s_im, T *o1, T *o2, T scale, int bins,
               int nyq, T s_edge, T s_mid, hipStream_t s) {
  const Tables<T> &t = tables<T>(plan);
  const int n2 = 1 << t.log2n2;
  const long long blocks = batch * (n2 / 256);
#define PDSP_OUT(L)                                                                                              \
  hipLaunchKernelGGL((pdsp::fourstep_out_kernel<T, L, MODE>), dim3((unsigned)blocks), dim3(256), 0, s, s_re, s_im, o1, \
                     o2, n2, scale, bins, nyq, s_edge, s_mid, batch)
  switch (t.log2n1) {
    case 1: PDSP_OUT(1); break;
    case 2: PDSP_OUT(2); break;
    case 3: PDSP_OUT(3); break;
    case 4: PDSP_OUT(4); break;
    default: return fail(PDSP_ERR_UNSUPPORTED_SIZE, "unsupported four-step split");
  }
#undef PDSP_OUT
  PDSP_HIP_TRY(hipGetLastError());
  return PDSP_OK;
}

// fft_staged_kernel on planar complex rows of 32 <= N <= 256 points (16-byte aligned planes).
template <typename T>
hipError_t launch_staged_complex(int log2n, const pdsp::LoadComplex<T> &ld, const pdsp::StoreComplex<T> &st,
                                 const typename pdsp::vec2<T>::type *tw, long long rows, hipStream_t s) {
  const long long blocks = ((rows << log2n) + 4095) / 4096;
#define PDSP_STAGED_C(L)                                                                                  \
  hipLaunchKernelGGL((pdsp::fft_staged_kernel<T, L, pdsp::LoadComplex<T>, pdsp::StoreComplex<T>>),         \
                     dim3((unsigned)blocks), dim3(256), 0, s, ld, st, tw, rows)
  switch (log2n) {
    case 5: PDSP_STAGED_C(5); break;
    case 6: PDSP_STAGED_C(6); break;
    case 7: PDSP_STAGED_C(7); break;
    case 8: PDSP_STAGED_C(8); break;
    default: return hipErrorInvalidValue;
  }
#undef PDSP_STAGED_C
  return hipGetLastError();
}

// findPeak over stored amplitude rows: index array and/or SpectrumPeak records.
template <typename T>
hipError_t launch_peaks(const T *amp, const T *ph, int bins, T freq_scale, int32_t *peak_idx, pdsp_peak32 *peaks,
                        long long batch, hipStream_t s) {
  pdsp::PeakRec *recs = reinterpret_cast<pdsp::PeakRec *>(peaks);
  if (bins <= 2048) {  // one wave per row
    hipLaunchKernelGGL((pdsp::peak_wave_kernel<T>), dim3((unsigned)((batch + 3) / 4)), dim3(256), 0, s, amp, ph, bins,
                       freq_scale, peak_idx, recs, batch);
    return hipGetLastError();
  }
  if (peak_idx) hipLaunchKernelGGL((pdsp::find_peak_kernel<T>), dim3((unsigned)batch), dim3(256), 0, s, amp, bins, peak_idx, batch);
  if (recs)
    hipLaunchKernelGGL((pdsp::peak_from_rows_kernel<T>), dim3((unsigned)batch), dim3(256), 0, s, amp, ph, bins, freq_scale,
                       recs, batch);
  return hipGetLastError();
}

// fft_tiny_staged_kernel for 2 <= N <= 16 (one thread per row, chunk staged through LDS).
template <typename T, bool AMP, class LD>
hipError_t launch_tiny(int log2n, const LD &ld, const T *win, T *o1, T *o2, T scale, int bins, int nyq, T s_edge,
                       T s_mid, long long batch, hipStream_t s) {
  const long long blocks = ((batch << log2n) + 4095) / 4096;
#define PDSP_TINY(L)                                                                                             \
  hipLaunchKernelGGL((pdsp::fft_tiny_staged_kernel<T, L, AMP, LD>), dim3((unsigned)blocks), dim3(256), 0, s, ld, win, \
                     o1, o2, scale, bins, nyq, s_edge, s_mid, batch)
  switch (log2n) {
    case 1: PDSP_TINY(1); break;
    case 2: PDSP_TINY(2); break;
    case 3: PDSP_TINY(3); break;
    case 4: PDSP_TINY(4); break;
    case 5:
      if constexpr (AMP) {  // N = 32 transforms have fft_staged_kernel; the spectrum of N = 32 frames comes here
        PDSP_TINY(5);
        break;
      }
      return hipErrorInvalidValue;
    default: return hipErrorInvalidValue;
  }
#undef PDSP_TINY
  return hipGetLastError();
}

// General four-step path (log2n1 > kMaxLog2N1), steps 1-4 of bigfft_transpose_kernel's header:
// transposes `in` into (a_re, a_im) = [n2][n1], N1-point rows in place, twiddled transpose into
// (b_re, b_im) = [k1][n2], N2-point rows in place.  Step 5 is bigfft_out.
template <typename T>
int bigfft_rows(const pdsp_plan *plan, long long batch, const T *re_in, const T *im_in, const T *win,
                long long in_stride, long long used, T *a_re, T *a_im, T *b_re, T *b_im, hipStream_t s) {
  const Tables<T> &t = tables<T>(plan);
  const int n1 = 1 << t.log2n1, n2 = 1 << t.log2n2;
  const long long tiles = batch * (plan->n / 1024);
  if (tiles >= (1LL << 31) || (batch << t.log2n2) >= (1LL << 31))
    return fail(PDSP_ERR_BAD_ARG, "batch %lld is too large for FFT size %lld", batch, plan->n);
  hipLaunchKernelGGL((pdsp::bigfft_transpose_kernel<T, false, false>), dim3((unsigned)tiles), dim3(256), 0, s, re_in,
                     im_in, win, used, in_stride, t.twa, t.twb, a_re, a_im, n1, n2, T(1), 0, 0, T(0), T(0));
  PDSP_HIP_TRY(hipGetLastError());
  {
    pdsp::LoadComplex<T> ld{a_re, a_im, n1};
    pdsp::StoreComplex<T> st{a_re, a_im, n1, T(1)};
    if (t.log2n1 == t.log2n2) PDSP_HIP_TRY(launch_rows<T>(t, t.log2n1, ld, st, batch << t.log2n2, s, true));
    else if (t.log2n1 <= (sizeof(T) == 4 ? 8 : 7) && g_staged_small)  // short rows: the staged kernel's coalesced I/O
      PDSP_HIP_TRY(launch_staged_complex<T>(t.log2n1, ld, st, t.tw1, batch << t.log2n2, s));
    else PDSP_HIP_TRY(launch_fft<T>(t.log2n1, ld, st, t.tw1, batch << t.log2n2, s));
  }
  hipLaunchKernelGGL((pdsp::bigfft_transpose_kernel<T, true, false>), dim3((unsigned)tiles), dim3(256), 0, s, a_re, a_im,
                     (const T *)nullptr, plan->n, plan->n, t.twa, t.twb, b_re, b_im, n2, n1, T(1), 0, 0, T(0), T(0));
  PDSP_HIP_TRY(hipGetLastError());
  pdsp::LoadComplex<T> ld{b_re, b_im, n2};
  pdsp::StoreComplex<T> st{b_re, b_im, n2, T(1)};
  PDSP_HIP_TRY(launch_rows<T>(t, t.log2n2, ld, st, batch << t.log2n1, s, true));
  return PDSP_OK;
}

// Step 5: [k1][k2] -> natural order.  AMP = false: complex planes (o1, o2) times `scale`;
// AMP = true: amplitude rows o1 (and phase rows o2 unless null) of `bins` values.
template <typename T, bool AMP>
int bigfft_out(const pdsp_plan *plan, long long batch, const T *b_re, const T *b_im, T *o1, T *o2, T scale, int bins,
               int nyq, T s_edge, T s_mid, hipStream_t s) {
  const Tables<T> &t = tables<T>(plan);
  const long long tiles = batch * (plan->n / 1024);
  hipLaunchKernelGGL((pdsp::bigfft_transpose_kernel<T, false, AMP>), dim3((unsigned)tiles), dim3(256), 0, s, b_re, b_im,
                     (const T *)nullptr, plan->n, plan->n, t.twa, t.twb, o1, o2, 1 << t.log2n1, 1 << t.log2n2, scale,
                     bins, nyq, s_edge, s_mid);
  PDSP_HIP_TRY(hipGetLastError());
  return PDSP_OK;
}

// One launch of tile_pass_kernel for a factor of 2^l points (tile width by factor: 64 / 32 / 32 / 16).
// real_in = tile_pass_kernel's IN: 0 complex planes, 1 real rows (in_im unused), 2 real rows times the window table
// in in_im, 3 / 4 the same for packed real rows (two samples per point)
template <typename T, bool COLS>
int tile_pass(int l, int real_in, const T *in_re, const T *in_im, T *out_re, T *out_im,
              const typename pdsp::vec2<T>::type *tw, const Tables<T> &t, pdsp::TileGeom g, T scale, long long batch,
              hipStream_t s) {
  const long long blocks = batch * g.nblk * g.tiles;
  if (blocks > 0x7fffffffLL) return fail(PDSP_ERR_BAD_ARG, "batch too large: %lld", batch);
  const pdsp::cx<T> *twa = reinterpret_cast<const pdsp::cx<T> *>(t.twa);
  const pdsp::cx<T> *twb = reinterpret_cast<const pdsp::cx<T> *>(t.twb);
  if constexpr (COLS && sizeof(T) == 4) {
    // a 512-point factor as a column pass: 32-column tiles through tile_cols512_kernel (128-byte strided segments)
    // instead of 16-column ones; pdsp_set_twopass bit 1 keeps the plain tiles (A/B tests)
    if (l == 9 && real_in <= 1 && t.tw8 && !(g_twopass & 2) && g.tiles % 2 == 0) {
      g.tiles /= 2;
      const long long wide = batch * g.nblk * g.tiles;
      if (real_in == 1)
        hipLaunchKernelGGL((pdsp::tile_cols512_kernel<T, 1>), dim3((unsigned)wide), dim3(256), 0, s, in_re, in_im, out_re,
                           out_im, t.tw8, twa, twb, g, batch);
      else
        hipLaunchKernelGGL((pdsp::tile_cols512_kernel<T, 0>), dim3((unsigned)wide), dim3(256), 0, s, in_re, in_im, out_re,
                           out_im, t.tw8, twa, twb, g, batch);
      PDSP_HIP_TRY(hipGetLastError());
      return PDSP_OK;
    }
  }
  if constexpr (!COLS && sizeof(T) == 4) {
    // a 512-point factor as the last pass: 32-row tiles through tile_rows512_kernel (128-byte output segments)
    // instead of 16-row ones; pdsp_set_twopass bit 1 keeps the plain tiles (A/B tests)
    if (l == 9 && t.tw8 && !(g_twopass & 2) && g.tiles % 2 == 0) {
      g.tiles /= 2;
      const long long wide = batch * g.tiles;
      hipLaunchKernelGGL((pdsp::tile_rows512_kernel<T>), dim3((unsigned)wide), dim3(256), 0, s, in_re, in_im, out_re, out_im,
                         t.tw8, twa, twb, g, scale, batch);
      PDSP_HIP_TRY(hipGetLastError());
      return PDSP_OK;
    }
  }
#define PDSP_TILE_IN(L, TILE, IN)                                                                                  \
  hipLaunchKernelGGL((pdsp::tile_pass_kernel<T, L, TILE, COLS, ((COLS && (IN < 5 || sizeof(T) == 4)) ? IN : 0)>),    \
                     dim3((unsigned)blocks), dim3(256),                                                               \
                     0, s, in_re, in_im, out_re, out_im, tw, twa, twb, g, scale, batch)
#define PDSP_TILE(L, TILE)                                 \
  do {                                                     \
    switch (COLS ? real_in : 0) {                          \
      case 1: PDSP_TILE_IN(L, TILE, 1); break;             \
      case 2: PDSP_TILE_IN(L, TILE, 2); break;             \
      case 3: PDSP_TILE_IN(L, TILE, 3); break;             \
      case 4: PDSP_TILE_IN(L, TILE, 4); break;             \
      case 5: PDSP_TILE_IN(L, TILE, 5); break;             \
      case 6: PDSP_TILE_IN(L, TILE, 6); break;             \
      default: PDSP_TILE_IN(L, TILE, 0); break;            \
    }                                                      \
  } while (0)
  switch (l) {
    case 6: PDSP_TILE(6, 64); break;
    case 7: PDSP_TILE(7, 32); break;
    case 8: PDSP_TILE(8, 32); break;
    case 9: PDSP_TILE(9, 16); break;
    default: return fail(PDSP_ERR_UNSUPPORTED_SIZE, "unsupported tile-pass factor 2^%d", l);
  }
#undef PDSP_TILE
#undef PDSP_TILE_IN
  PDSP_HIP_TRY(hipGetLastError());
  return PDSP_OK;
}
constexpr int tile_width(int l) { return l == 6 ? 64 : (l == 9 ? 16 : 32); }

// Two or three tile passes (tile_pass_kernel's header): 2^15 <= N <= 2^27, f32, 16-byte aligned
// planes.  s1 / s2: scratch plane pairs ((re, im) each); s2 is only used by the three-pass form.  Every
// pass reads one pair and writes another, so input and output may alias each other.
// `window` (real input only): applyWindow on the first pass's load; in_batch: distance between input rows.
// The pass chain on one set of factor tables: `n` points per transform, np factors 2^l[i] with radix tables tw[i];
// tshift = 1 when t.twa / t.twb belong to the 2n-point plan.  `first` = tile_pass_kernel's IN for the first pass
// (im_in then carries the window table or nothing); in_batch = distance between input rows (real samples for
// first >= 1).
template <typename T>
int tilepass_chain(const Tables<T> &t, long long n, int np, const int *l, typename pdsp::vec2<T>::type *const *tw,
                   unsigned tshift, int first, long long batch, const T *re_in, const T *im_in, long long in_batch,
                   T *re_out, T *im_out, T scale, T *s1_re, T *s1_im, T *s2_re, T *s2_im, hipStream_t s,
                   const pdsp::TileGeom *fused_win = nullptr) {
  auto with_window = [&](pdsp::TileGeom &g) {  // first = 5 / 6: the angle-addition tables and coefficients
    if (fused_win) {
      g.wa = fused_win->wa, g.wb = fused_win->wb, g.wstep = fused_win->wstep, g.we = fused_win->we;
      g.k0 = fused_win->k0, g.k1 = fused_win->k1, g.k2 = fused_win->k2;
    }
  };
  if (np == 2) {
    const long long a = 1LL << l[0], b = 1LL << l[1];
    pdsp::TileGeom g1{n, 1, (int)(b / tile_width(l[0])), 0, 0, b, b, 1u, in_batch, tshift};
    with_window(g1);
    if (int rc = tile_pass<T, true>(l[0], first, re_in, im_in, s1_re, s1_im, tw[0], t, g1, T(1), batch, s)) return rc;
    pdsp::TileGeom g2{n, 1, (int)(a / tile_width(l[1])), 0, 0, 0, a, 1u, n, tshift};
    return tile_pass<T, false>(l[1], 0, (const T *)s1_re, (const T *)s1_im, re_out, im_out, tw[1], t, g2, scale, batch, s);
  }
  const long long a = 1LL << l[0], b = 1LL << l[1], c = 1LL << l[2];
  pdsp::TileGeom g1{n, 1, (int)(b * c / tile_width(l[0])), 0, 0, b * c, b * c, 1u, in_batch, tshift};
  with_window(g1);
  pdsp::TileGeom g2{n, (int)a, (int)(c / tile_width(l[1])), b * c, c, c, a * c, (unsigned)a, n, tshift};
  if (!(g_twopass & 2)) {
    // the scratch planes between the first two passes tile-major (TileGeom::perm_*): the second pass reads its
    // [B][TILE] tiles as contiguous chunks; pdsp_set_twopass bit 1 keeps them in natural order (A/B tests)
    // log2 of the second pass's tile width (512-point columns: 32 on tile_cols512_kernel, which this mode implies)
    const int lt = l[1] == 6 ? 6 : ((l[1] == 9 && !(sizeof(T) == 4 && t.tw8)) ? 4 : 5);
    g1.perm_lc = l[2], g1.perm_lt = lt, g1.perm_b = (int)b;
    g2.in_tile = b << lt, g2.in_stride = 1LL << lt;
  }
  if (int rc = tile_pass<T, true>(l[0], first, re_in, im_in, s1_re, s1_im, tw[0], t, g1, T(1), batch, s)) return rc;
  if (int rc = tile_pass<T, true>(l[1], 0, (const T *)s1_re, (const T *)s1_im, s2_re, s2_im, tw[1], t, g2, T(1), batch, s))
    return rc;
  pdsp::TileGeom g3{n, 1, (int)(a * b / tile_width(l[2])), 0, 0, 0, a * b, 1u, n, tshift};
  return tile_pass<T, false>(l[2], 0, (const T *)s2_re, (const T *)s2_im, re_out, im_out, tw[2], t, g3, scale, batch, s);
}

// Two or three tile passes (tile_pass_kernel's header): 2^15 <= N <= 2^27, f32, 16-byte aligned
// planes.  s1 / s2: scratch plane pairs ((re, im) each); s2 is only used by the three-pass form.  Every
// pass reads one pair and writes another, so input and output may alias each other.
// `window` (real input only): applyWindow on the first pass's load; in_batch: distance between input rows.
template <typename T>
int tilepass_complex(const pdsp_plan *plan, long long batch, const T *re_in, const T *im_in, T *re_out, T *im_out,
                     T scale, T *s1_re, T *s1_im, T *s2_re, T *s2_im, hipStream_t s, const T *window = nullptr,
                     long long in_batch = 0) {
  const Tables<T> &t = tables<T>(plan);
  const int first = im_in ? 0 : (window ? 2 : 1);
  return tilepass_chain<T>(t, plan->n, t.tp_np, t.tp_l, t.tp_tw, 0u, first, batch, re_in, first == 2 ? window : im_in,
                           in_batch ? in_batch : plan->n, re_out, im_out, scale, s1_re, s1_im, s2_re, s2_im, s);
}

// Do any of the output planes share bytes with any of the input planes?  Byte ranges, not pointer equality: an
// output that starts one row into the input buffer overlaps it too.  The multi-pass paths use the output planes as
// their first scratch pair only when this is false.
template <typename T>
bool planes_overlap(const T *re_in, const T *im_in, const T *re_out, const T *im_out, size_t plane_bytes) {
  auto hit = [&](const T *a, const T *b) {
    return a && b && (const char *)a < (const char *)b + plane_bytes && (const char *)b < (const char *)a + plane_bytes;
  };
  return hit(re_in, re_out) || hit(re_in, im_out) || hit(im_in, re_out) || hit(im_in, im_out);
}

template <typename T>
int run_complex(const pdsp_plan *plan, long long batch, const T *re_in, const T *im_in, T *re_out, T *im_out, T scale,
                hipStream_t s) {
  if (int rc = check_plan_batch(plan, batch)) return rc;
  if (batch == 0) return PDSP_OK;
  if (!re_in || !re_out || !im_out) return fail(PDSP_ERR_BAD_ARG, "null buffer");
  const Tables<T> &t = tables<T>(plan);
  if (!t.tw)
    return fail(PDSP_ERR_UNSUPPORTED_SIZE, "FFT size %lld exceeds the %d-bit limit %d", plan->n, (int)(8 * sizeof(T)),
                pdsp_max_size((int)sizeof(T)));
  DeviceGuard g(plan->device);
  PDSP_HIP_TRY(g.err);
  // f64 Radix2Fft.forward rows (real input; f64 is the drop-in's default arithmetic) of N = 8192 and 16384: one
  // N/2-point packed-real transform per row and the split to X[k], X[k + N/2] (fft_real_kernel) -- half the
  // butterflies of the complex kernel on (x, 0), the same store streams, and N = 16384 stays in one pass.  These are
  // the sizes where the complex f64 kernel is short of registers (N = 8192: fft_split2_kernel's LoadReal form
  // spilled) or does not exist (N = 16384: four-step): tools/ab_real_packed.py --f64 on two boxes, N = 8192
  // 4.93 -> 6.57 and 3.97 -> 5.69 TB/s, N = 16384 1.60 -> 5.18 and 1.57 -> 5.07; one frame through the host drop-in
  // (tools/ab_single_frame_latency.py) 39.7 -> 36.2 us at 8192, but 47.3 -> 50.1 us at 16384 (one 512-thread
  // workgroup is a longer critical path than three short launches), hence the batch threshold there.  Below 8192
  // the same kernel measured +1 ... +9 % on one box and -9 ... +2 % on another (and 3-7 % slower for one frame): not
  // robust, not dispatched, not built.  In f32 it measured 0.98 ... 1.01 of the complex kernels: not built either.
  // Rows aligned to a sample pair.
  if constexpr (sizeof(T) == 8) {
    if (!im_in && g_real_packed && (plan->log2n == 13 || (plan->log2n == 14 && batch >= 8)) && t.tw_half && t.twr &&
        ((uintptr_t)re_in & (2 * sizeof(T) - 1)) == 0) {
      PDSP_HIP_TRY(launch_real<T>(plan->log2n - 1, re_in, re_out, im_out, scale, t.tw_half, t.twr, batch, s));
      return PDSP_OK;
    }
  }
  if constexpr (sizeof(T) == 4) {
    // N = 2^15 / 2^16 out of place: ONE pass over HBM by 2 / 4 sibling workgroups per transform that share their
    // XCD's L2 (fft_paired_kernel).  In place the siblings would overwrite each other's input: tile passes then.
    // pdsp_set_twopass: any value but 1 keeps the tile passes (5: their current form) -- A/B tests.
    if ((plan->log2n == 15 || plan->log2n == 16) && g_twopass == 1 && t.tw12 && t.tws4 && t.twa && t.twb &&
        (((uintptr_t)re_in | (uintptr_t)im_in | (uintptr_t)re_out | (uintptr_t)im_out) & 15) == 0) {
      if (!planes_overlap(re_in, im_in, re_out, im_out, (size_t)batch * (size_t)plan->n * sizeof(T))) {
        const int lp = plan->log2n - 14;
        const long long blocks = ((batch + 7) / 8) * 8 * (1LL << lp);
        if (blocks > 0x7fffffffLL) return fail(PDSP_ERR_BAD_ARG, "batch too large: %lld", batch);
        const pdsp::cx<T> *twa = reinterpret_cast<const pdsp::cx<T> *>(t.twa);
        const pdsp::cx<T> *twb = reinterpret_cast<const pdsp::cx<T> *>(t.twb);
#define PDSP_PAIRED(LP, REAL)                                                                                        \
  hipLaunchKernelGGL((pdsp::fft_paired_kernel<T, LP, REAL>), dim3((unsigned)blocks), dim3(256), 0, s, re_in, im_in, re_out, \
                     im_out, t.tw12, t.tws4, twa, twb, scale, batch, pdsp::PairedPacked{})
        if (lp == 1) {
          if (im_in) PDSP_PAIRED(1, false);
          else PDSP_PAIRED(1, true);
        } else {
          if (im_in) PDSP_PAIRED(2, false);
          else PDSP_PAIRED(2, true);
        }
#undef PDSP_PAIRED
        PDSP_HIP_TRY(hipGetLastError());
        return PDSP_OK;
      }
    }
    // tile passes with balanced factors (two for 2^15..2^18, three for 2^19..2^27) where the tables exist and
    // every plane is 16-byte aligned; pdsp_set_twopass(0) keeps round 1's four-step forms (A/B tests)
    if (t.tp_np && (g_twopass & 1) &&
        (((uintptr_t)re_in | (uintptr_t)im_in | (uintptr_t)re_out | (uintptr_t)im_out) & 15) == 0) {
      const size_t plane = (size_t)batch * (size_t)plan->n;
      const bool aliased = planes_overlap(re_in, im_in, re_out, im_out, plane * sizeof(T));
      // two passes: one scratch pair.  Three passes: the output planes double as the first scratch pair
      // unless they share bytes with the input (equal pointers or a partial overlap).
      const int pairs = t.tp_np == 2 ? 1 : (aliased ? 2 : 1);
      StreamScratch mem(s);
      PDSP_HIP_TRY(mem.alloc((size_t)pairs * 2 * plane * sizeof(T)));
      T *const sc = (T *)mem.p;
      if (t.tp_np == 2)
        return tilepass_complex<T>(plan, batch, re_in, im_in, re_out, im_out, scale, sc, sc + plane, nullptr, nullptr, s);
      T *s1_re = aliased ? sc + 2 * plane : re_out, *s1_im = aliased ? sc + 3 * plane : im_out;
      return tilepass_complex<T>(plan, batch, re_in, im_in, re_out, im_out, scale, s1_re, s1_im, sc, sc + plane, s);
    }
  }
  if (t.log2n1 > pdsp::kMaxLog2N1) {  // general four-step: the output planes double as the first scratch pair
    const size_t plane = (size_t)batch * (size_t)plan->n;
    const bool aliased = planes_overlap(re_in, im_in, re_out, im_out, plane * sizeof(T));
    StreamScratch mem(s);
    PDSP_HIP_TRY(mem.alloc((aliased ? 4 : 2) * plane * sizeof(T)));
    T *const scratch = (T *)mem.p;
    T *a_re = aliased ? scratch + 2 * plane : re_out, *a_im = aliased ? scratch + 3 * plane : im_out;
    int rc = bigfft_rows<T>(plan, batch, re_in, im_in, nullptr, plan->n, plan->n, a_re, a_im, scratch, scratch + plane, s);
    if (!rc) rc = bigfft_out<T, false>(plan, batch, scratch, scratch + plane, re_out, im_out, scale, 0, 0, T(0), T(0), s);
    return rc;
  }
  if (t.log2n1 > 0) {  // beyond the single-pass limit: four-step through stream-ordered scratch planes
    const size_t plane = (size_t)batch * (size_t)plan->n;
    StreamScratch mem(s);
    PDSP_HIP_TRY(mem.alloc(2 * plane * sizeof(T)));
    T *const scratch = (T *)mem.p;
    int rc = fourstep_ab<T>(plan, batch, re_in, im_in, nullptr, plan->n, plan->n, scratch, scratch + plane, s);
    if (!rc) rc = fourstep_c<T, 0>(plan, batch, scratch, scratch + plane, re_out, im_out, scale, 0, 0, T(0), T(0), s);
    return rc;
  }
  hipError_t e;
  const bool planes16 =
      (((uintptr_t)re_in | (uintptr_t)im_in | (uintptr_t)re_out | (uintptr_t)im_out) & (4 * sizeof(T) - 1)) == 0;
  if (plan->log2n >= 1 && plan->log2n <= 4 && g_staged_small && planes16) {  // 2 <= N <= 16: one thread per row
    if (im_in) {
      pdsp::LoadComplex<T> ld{re_in, im_in, plan->n};
      e = launch_tiny<T, false>(plan->log2n, ld, (const T *)nullptr, re_out, im_out, scale, 0, 0, T(0), T(0), batch, s);
    } else {
      pdsp::LoadReal<T> ld{re_in, plan->n};
      e = launch_tiny<T, false>(plan->log2n, ld, (const T *)nullptr, re_out, im_out, scale, 0, 0, T(0), T(0), batch, s);
    }
    PDSP_HIP_TRY(e);
    return PDSP_OK;
  }
  pdsp::StoreComplex<T> st{re_out, im_out, plan->n, scale};
  // (f64 at N = 256: 69.6 KB of LDS per workgroup, the direct kernel measures 12 % faster)
  if (plan->log2n >= 5 && plan->log2n <= (sizeof(T) == 4 ? 8 : 7) && g_staged_small &&
      (((uintptr_t)re_in | (uintptr_t)im_in | (uintptr_t)re_out | (uintptr_t)im_out) & (4 * sizeof(T) - 1)) == 0) {
    // small N: coalesced 16-byte I/O staged through LDS (fft_staged_kernel)
    const long long blocks = (batch * plan->n + 4095) / 4096;
#define PDSP_STAGED(L)                                                                                         \
  do {                                                                                                         \
    if (im_in) {                                                                                               \
      pdsp::LoadComplex<T> ld{re_in, im_in, plan->n};                                                          \
      hipLaunchKernelGGL((pdsp::fft_staged_kernel<T, L, pdsp::LoadComplex<T>, pdsp::StoreComplex<T>>),          \
                         dim3((unsigned)blocks), dim3(256), 0, s, ld, st, t.tw, batch);                        \
    } else {                                                                                                   \
      pdsp::LoadReal<T> ld{re_in, plan->n};                                                                    \
      hipLaunchKernelGGL((pdsp::fft_staged_kernel<T, L, pdsp::LoadReal<T>, pdsp::StoreComplex<T>>),             \
                         dim3((unsigned)blocks), dim3(256), 0, s, ld, st, t.tw, batch);                        \
    }                                                                                                          \
  } while (0)
    switch (plan->log2n) {
      case 5: PDSP_STAGED(5); break;
      case 6: PDSP_STAGED(6); break;
      case 7: PDSP_STAGED(7); break;
      default: PDSP_STAGED(8); break;
    }
#undef PDSP_STAGED
    PDSP_HIP_TRY(hipGetLastError());
    return PDSP_OK;
  }
  const bool aligned16 = (((uintptr_t)re_in | (uintptr_t)im_in) & 15) == 0;
  if (im_in) {
    pdsp::LoadComplex<T> ld{re_in, im_in, plan->n};
    e = launch_rows<T>(t, plan->log2n, ld, st, batch, s, aligned16);
  } else {
    pdsp::LoadReal<T> ld{re_in, plan->n};
    e = launch_rows<T>(t, plan->log2n, ld, st, batch, s, aligned16);
  }
  PDSP_HIP_TRY(e);
  return PDSP_OK;
}

// Interleaved complex rows (single-pass sizes): forward, or inverse = conj . forward . conj with 1/N.
template <typename T>
int run_interleaved(const pdsp_plan *plan, long long batch, const T *in, T *out, bool inverse, hipStream_t s) {
  if (int rc = check_plan_batch(plan, batch)) return rc;
  if (batch == 0) return PDSP_OK;
  if (!in || !out) return fail(PDSP_ERR_BAD_ARG, "null buffer");
  if ((((uintptr_t)in | (uintptr_t)out) & (2 * sizeof(T) - 1)) != 0)
    return fail(PDSP_ERR_BAD_ARG, "interleaved rows must be aligned to one (re, im) pair");
  const Tables<T> &t = tables<T>(plan);
  if (!t.tw || t.log2n1 > 0)
    return fail(PDSP_ERR_UNSUPPORTED_SIZE, "interleaved rows are single-pass only: FFT size %lld exceeds %d", plan->n,
                1 << max_log2n<T>());
  DeviceGuard g(plan->device);
  PDSP_HIP_TRY(g.err);
  const pdsp::cx<T> *zin = reinterpret_cast<const pdsp::cx<T> *>(in);
  pdsp::cx<T> *zout = reinterpret_cast<pdsp::cx<T> *>(out);
  if (inverse) {
    pdsp::LoadInterleaved<T, true> ld{zin, plan->n};
    pdsp::StoreInterleaved<T, true> st{zout, plan->n, T(1) / (T)plan->n};
    PDSP_HIP_TRY(launch_fft<T>(plan->log2n, ld, st, t.tw, batch, s));
  } else {
    pdsp::LoadInterleaved<T, false> ld{zin, plan->n};
    pdsp::StoreInterleaved<T, false> st{zout, plan->n, T(1)};
    PDSP_HIP_TRY(launch_fft<T>(plan->log2n, ld, st, t.tw, batch, s));
  }
  return PDSP_OK;
}

int grid_for(long long total) {
  long long b = (total + 255) / 256;
  if (b > 2048) b = 2048;  // grid-stride the rest (256 CUs x 8)
  if (b < 1) b = 1;
  return (int)b;
}

int ensure_stage(pdsp_plan *plan, size_t bytes) {
  if (!plan->stream) PDSP_HIP_TRY(hipStreamCreateWithFlags(&plan->stream, hipStreamNonBlocking));
  if (plan->h_bytes < bytes) {
    if (plan->h_stage) (void)hipHostFree(plan->h_stage);
    plan->h_stage = nullptr;
    plan->h_bytes = 0;
    PDSP_HIP_TRY(hipHostMalloc(&plan->h_stage, bytes, hipHostMallocDefault));
    plan->h_bytes = bytes;
  }
  if (plan->d_bytes < bytes) {
    if (plan->d_stage) (void)hipFree(plan->d_stage);
    plan->d_stage = nullptr;
    plan->d_bytes = 0;
    PDSP_HIP_TRY(hipMalloc(&plan->d_stage, bytes));
    plan->d_bytes = bytes;
  }
  return PDSP_OK;
}

// One-frame calls are latency-bound (two small copies + one kernel + one sync).  Up to 1 MiB of staging
// the kernel reads the frame from, and writes the result to, the pinned staging buffer itself
// (hipHostMalloc memory is mapped into the device's address space): no copy commands at all.
// PDSP_ZERO_COPY=0 in the environment restores the staged copies (A/B, tests); a value > 1 sets the limit.
long long g_zero_copy_bytes = -1;
bool zero_copy(size_t bytes) {
  if (g_zero_copy_bytes < 0) {
    const char *e = getenv("PDSP_ZERO_COPY");  // 0 = off, 1 / unset = default limit, > 1 = limit in bytes
    const long long v = e ? atoll(e) : 1;
    g_zero_copy_bytes = v <= 0 ? 0 : (v == 1 ? 1024 * 1024 : v);
  }
  return (long long)bytes <= g_zero_copy_bytes;
}
template <typename T>
T *stage_device_view(pdsp_plan *plan) {  // device-side address of the pinned staging buffer
  void *dp = nullptr;
  if (hipHostGetDevicePointer(&dp, plan->h_stage, 0) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return (T *)dp;
}

// Scratch (plan-less) staging for the element-wise host entry points.
template <typename T>
struct Scratch {
  T *h = nullptr, *d = nullptr;
  ~Scratch() {
    if (h) (void)hipHostFree(h);
    if (d) (void)hipFree(d);
  }
  int reserve(size_t count) {
    PDSP_HIP_TRY(hipHostMalloc((void **)&h, count * sizeof(T), hipHostMallocDefault));
    PDSP_HIP_TRY(hipMalloc((void **)&d, count * sizeof(T)));
    return PDSP_OK;
  }
};

// Plans for the one-shot host entry points, keyed by (size, device): the idea of
// FourierLive's `Map<size, FFT>` and `Map<"type:size", window>` caches
// (src/effect/index.ts:30-48).  The reference's spectrum() rebuilds both on every call
// (spectrum.ts:114-116) -- its dominant one-shot cost; here a repeat call costs no table
// build, no hipMalloc and no upload.  Leaked on purpose at exit (the HIP runtime may
// already be gone when static destructors run); pdsp_plan_cache_clear() frees it.
// Bounded: at most kMaxCachedPlans entries, least recently used evicted first (an entry a call is
// still running on is pinned and never evicted), so a long-running host that calls spectrum() with
// ever-changing lengths keeps a bounded set of tables, streams and staging buffers.
constexpr size_t kMaxCachedPlans = 16;
struct PlanCache {
  struct Entry {
    pdsp_plan *plan = nullptr;
    unsigned long long last_use = 0;
    int pins = 0;
  };
  std::mutex mu;
  std::map<std::pair<long long, int>, Entry> plans;
  unsigned long long tick = 0;
};
PlanCache &plan_cache() {
  static PlanCache *c = new PlanCache();
  return *c;
}

// RAII pin of a cached plan for the duration of one host call.
struct CachedPlan {
  pdsp_plan *plan = nullptr;
  std::pair<long long, int> key{0, 0};
  CachedPlan() = default;
  CachedPlan(const CachedPlan &) = delete;
  CachedPlan &operator=(const CachedPlan &) = delete;
  ~CachedPlan() {
    if (!plan) return;
    PlanCache &c = plan_cache();
    std::lock_guard<std::mutex> lk(c.mu);
    auto it = c.plans.find(key);
    if (it != c.plans.end() && it->second.plan == plan) --it->second.pins;
  }
};

int cached_plan(long long n, CachedPlan *out) {
  int dev = 0;
  PDSP_HIP_TRY(hipGetDevice(&dev));
  PlanCache &c = plan_cache();
  std::lock_guard<std::mutex> lk(c.mu);
  const std::pair<long long, int> key{n, dev};
  auto it = c.plans.find(key);
  if (it == c.plans.end()) {
    // make room first: drop least-recently-used entries nobody is running on
    while (c.plans.size() >= kMaxCachedPlans) {
      auto victim = c.plans.end();
      for (auto jt = c.plans.begin(); jt != c.plans.end(); ++jt)
        if (jt->second.pins == 0 && (victim == c.plans.end() || jt->second.last_use < victim->second.last_use)) victim = jt;
      if (victim == c.plans.end()) break;  // every entry is in use: grow past the bound rather than block
      pdsp_plan_destroy(victim->second.plan);
      c.plans.erase(victim);
    }
    pdsp_plan *p = nullptr;
    if (int rc = pdsp_plan_create(n, dev, &p)) return rc;
    it = c.plans.emplace(key, PlanCache::Entry{p, 0, 0}).first;
  }
  it->second.last_use = ++c.tick;
  ++it->second.pins;
  out->plan = it->second.plan;
  out->key = key;
  return PDSP_OK;
}

// Staging above this size is handed back after the call that needed it (a one-off long frame or
// large batch must not pin host memory and HBM for the life of the plan); smaller staging stays, so
// repeat calls of ordinary sizes still cost no allocation.  Caller holds plan->mu.
constexpr size_t kStageKeepBytes = (size_t)64 << 20;
void trim_stage(pdsp_plan *plan) {
  if (plan->h_bytes > kStageKeepBytes) {
    (void)hipHostFree(plan->h_stage);
    plan->h_stage = nullptr;
    plan->h_bytes = 0;
  }
  if (plan->d_bytes > kStageKeepBytes) {
    (void)hipFree(plan->d_stage);
    plan->d_stage = nullptr;
    plan->d_bytes = 0;
  }
}

// Device copy of createWindow(type, N), built once per plan and precision (the window is
// always computed in f64 on the host and rounded once).  Caller holds plan->mu.
template <typename T>
int plan_window(pdsp_plan *plan, int type, const T **out) {
  Tables<T> &t = tables<T>(plan);
  if (!t.win[type]) {
    std::vector<double> w((size_t)plan->n);
    if (int rc = pdsp_window_make(type, plan->n, w.data())) return rc;
    std::vector<T> wt(w.begin(), w.end());
    T *d = nullptr;
    PDSP_HIP_TRY(hipMalloc((void **)&d, wt.size() * sizeof(T)));
    hipError_t e = hipMemcpy(d, wt.data(), wt.size() * sizeof(T), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
      (void)hipFree(d);
      PDSP_HIP_TRY(e);
    }
    t.win[type] = d;
  }
  *out = t.win[type];
  return PDSP_OK;
}

// Precision of the host-f64 entry points: 64 (default) computes in f64 wherever the single-pass
// kernels hold the size (complex N <= 8192, real spectrum N <= 16384) and in f32 beyond; 32 always
// computes in f32 (the north-star's contract).  PDSP_HOST_PRECISION=32 in the environment presets it.
int g_host_precision = 0;
int host_precision() {
  if (g_host_precision == 0) {
    const char *e = getenv("PDSP_HOST_PRECISION");
    g_host_precision = (e && atoi(e) == 32) ? 32 : 64;
  }
  return g_host_precision;
}

template <typename T>
hipError_t upload_tables(Tables<T> &t, int log2n, long long size, bool full, bool half) {
  using T2 = typename pdsp::vec2<T>::type;
  hipError_t e = hipSuccess;
  if (full) {
    t.log2n2 = log2n > max_log2n<T>() ? max_log2n<T>() : log2n;
    t.log2n1 = log2n - t.log2n2;
    const std::vector<T2> tw = build_twiddles<T2>(t.log2n2);
    e = hipMalloc((void **)&t.tw, tw.size() * sizeof(T2));
    if (e == hipSuccess) e = hipMemcpy(t.tw, tw.data(), tw.size() * sizeof(T2), hipMemcpyHostToDevice);
    if (e == hipSuccess && t.log2n2 == 14) {  // fft_split4_kernel: 4096-point radix table + W_16384^k
      const std::vector<T2> t12 = build_twiddles<T2>(12);
      std::vector<T2> w(768);
      for (size_t k = 0; k < w.size(); ++k) {
        const double angle = (-2.0 * M_PI * (double)k) / 16384.0;
        w[k].x = (T)std::cos(angle);
        w[k].y = (T)std::sin(angle);
      }
      e = hipMalloc((void **)&t.tw12, t12.size() * sizeof(T2));
      if (e == hipSuccess) e = hipMemcpy(t.tw12, t12.data(), t12.size() * sizeof(T2), hipMemcpyHostToDevice);
      if (e == hipSuccess) e = hipMalloc((void **)&t.tws4, w.size() * sizeof(T2));
      if (e == hipSuccess) e = hipMemcpy(t.tws4, w.data(), w.size() * sizeof(T2), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess && t.log2n2 == 13) {  // fft_split2_kernel: 4096-point radix table + W_8192^k
      const std::vector<T2> t12 = build_twiddles<T2>(12);
      std::vector<T2> w(256);
      for (size_t k = 0; k < w.size(); ++k) {
        const double angle = (-2.0 * M_PI * (double)k) / 8192.0;
        w[k].x = (T)std::cos(angle);
        w[k].y = (T)std::sin(angle);
      }
      e = hipMalloc((void **)&t.tw12, t12.size() * sizeof(T2));
      if (e == hipSuccess) e = hipMemcpy(t.tw12, t12.data(), t12.size() * sizeof(T2), hipMemcpyHostToDevice);
      if (e == hipSuccess) e = hipMalloc((void **)&t.tws2, w.size() * sizeof(T2));
      if (e == hipSuccess) e = hipMemcpy(t.tws2, w.data(), w.size() * sizeof(T2), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess && t.log2n1 > 0) {  // W_N^m = twa[m >> 9] * twb[m & 511]
      std::vector<T2> a((size_t)(size >> 9)), b(512);
      for (size_t i = 0; i < a.size(); ++i) {
        const double angle = (-2.0 * M_PI * (double)(i << 9)) / (double)size;
        a[i].x = (T)std::cos(angle);
        a[i].y = (T)std::sin(angle);
      }
      for (size_t i = 0; i < 512; ++i) {
        const double angle = (-2.0 * M_PI * (double)i) / (double)size;
        b[i].x = (T)std::cos(angle);
        b[i].y = (T)std::sin(angle);
      }
      e = hipMalloc((void **)&t.twa, a.size() * sizeof(T2));
      if (e == hipSuccess) e = hipMemcpy(t.twa, a.data(), a.size() * sizeof(T2), hipMemcpyHostToDevice);
      if (e == hipSuccess) e = hipMalloc((void **)&t.twb, b.size() * sizeof(T2));
      if (e == hipSuccess) e = hipMemcpy(t.twb, b.data(), b.size() * sizeof(T2), hipMemcpyHostToDevice);
      // tile passes, balanced factors (tile_pass_kernel's header), ascending so that the widest tiles
      // serve the passes with two strided streams
      if (e == hipSuccess && sizeof(T) == 4 && log2n >= 15 && log2n <= 27) {
        if (log2n <= 18) {  // 2^18 = 512 * 512: both factors on 32-wide tiles (tile_cols512_kernel, tile_rows512_kernel)
          t.tp_np = 2;
          t.tp_l[0] = log2n / 2, t.tp_l[1] = log2n - t.tp_l[0];
        } else {
          t.tp_np = 3;
          three_factors(log2n, t.tp_l);
        }
        if (e == hipSuccess) {
          const std::vector<T2> t8 = build_twiddles<T2>(8);
          e = hipMalloc((void **)&t.tw8, t8.size() * sizeof(T2));
          if (e == hipSuccess) e = hipMemcpy(t.tw8, t8.data(), t8.size() * sizeof(T2), hipMemcpyHostToDevice);
        }
        for (int i = 0; i < t.tp_np && e == hipSuccess; ++i) {
          const std::vector<T2> tf = build_twiddles<T2>(t.tp_l[i]);
          e = hipMalloc((void **)&t.tp_tw[i], tf.size() * sizeof(T2));
          if (e == hipSuccess) e = hipMemcpy(t.tp_tw[i], tf.data(), tf.size() * sizeof(T2), hipMemcpyHostToDevice);
        }
        // the N/2-point transform of the packed-real spectrum path: 2^14 ... 2^18 in two factors (2^18 = 512 * 512:
        // the packed first pass on 16-column tiles still reads 128-byte segments, eight samples per lane), above in three
        const int lm = log2n - 1;
        if (lm <= 18) {
          t.hp_np = 2;
          t.hp_l[0] = lm / 2, t.hp_l[1] = lm - t.hp_l[0];
        } else {
          t.hp_np = 3;
          three_factors(lm, t.hp_l);
        }
        for (int i = 0; i < t.hp_np && e == hipSuccess; ++i) {
          const std::vector<T2> tf = build_twiddles<T2>(t.hp_l[i]);
          e = hipMalloc((void **)&t.hp_tw[i], tf.size() * sizeof(T2));
          if (e == hipSuccess) e = hipMemcpy(t.hp_tw[i], tf.data(), tf.size() * sizeof(T2), hipMemcpyHostToDevice);
        }
        if (e == hipSuccess) {
          // fused createWindow on the packed first pass (tile_pass_kernel IN = 5 / 6; fourier.ts:14-52:
          // f = 2 pi / (size - 1)): cs(f n) by angle addition, tables built in f64
          const double f = 2.0 * M_PI / (double)(size - 1);
          const long long in_stride = (size / 2) >> t.hp_l[0];              // points between the rows of a column
          const int spi = 1024 / tile_width(t.hp_l[0]);                     // tile_pass_kernel's SPI
          t.hp_win_a = (size_t)((size / 8 + 511) / 512);
          std::vector<float> w(2 * (t.hp_win_a + 512 + 8 + 8 + 16));  // + wq (fft_split4_kernel's packed loader, N = 2^15)
          size_t o = 0;
          for (size_t i = 0; i < t.hp_win_a; ++i, o += 2)
            w[o] = (float)std::cos(f * 4096.0 * (double)i), w[o + 1] = (float)std::sin(f * 4096.0 * (double)i);
          for (int j = 0; j < 512; ++j, o += 2) w[o] = (float)std::cos(f * 8.0 * j), w[o + 1] = (float)std::sin(f * 8.0 * j);
          for (int ic = 0; ic < 8; ++ic, o += 2) {
            const double a = f * 2.0 * (double)in_stride * (double)spi * (double)ic;
            w[o] = (float)std::cos(a), w[o + 1] = (float)std::sin(a);
          }
          for (int ee = 0; ee < 8; ++ee, o += 2) w[o] = (float)std::cos(f * ee), w[o + 1] = (float)std::sin(f * ee);
          for (int q = 0; q < 16; ++q, o += 2) w[o] = (float)std::cos(f * 2048.0 * q), w[o + 1] = (float)std::sin(f * 2048.0 * q);
          e = hipMalloc((void **)&t.hp_win, w.size() * sizeof(float));
          if (e == hipSuccess) e = hipMemcpy(t.hp_win, w.data(), w.size() * sizeof(float), hipMemcpyHostToDevice);
        }
      }
      if (e == hipSuccess && t.log2n1 > pdsp::kMaxLog2N1) {
        const std::vector<T2> t1 = build_twiddles<T2>(t.log2n1);
        e = hipMalloc((void **)&t.tw1, t1.size() * sizeof(T2));
        if (e == hipSuccess) e = hipMemcpy(t.tw1, t1.data(), t1.size() * sizeof(T2), hipMemcpyHostToDevice);
      }
    }
  }
  if (e == hipSuccess && half) {
    const std::vector<T2> twh = build_twiddles<T2>(log2n - 1, pdsp::packed_log2e(log2n - 1));
    std::vector<T2> twr((size_t)(size / 4 + 1));
    for (long long k = 0; k <= size / 4; ++k) {
      const double angle = (-2.0 * M_PI * (double)k) / (double)size;
      twr[(size_t)k].x = (T)std::cos(angle);
      twr[(size_t)k].y = (T)std::sin(angle);
    }
    e = hipMalloc((void **)&t.tw_half, twh.size() * sizeof(T2));
    if (e == hipSuccess) e = hipMemcpy(t.tw_half, twh.data(), twh.size() * sizeof(T2), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void **)&t.twr, twr.size() * sizeof(T2));
    if (e == hipSuccess) e = hipMemcpy(t.twr, twr.data(), twr.size() * sizeof(T2), hipMemcpyHostToDevice);
    if (e == hipSuccess && log2n == 14 && !t.tw12) {
      const std::vector<T2> t12 = build_twiddles<T2>(12);
      e = hipMalloc((void **)&t.tw12, t12.size() * sizeof(T2));
      if (e == hipSuccess) e = hipMemcpy(t.tw12, t12.data(), t12.size() * sizeof(T2), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess && log2n >= 10 && log2n <= 13 && sizeof(T) == 4) {
      // fused createWindow for spectrum_packed_kernel (N = 1024 ... 8192): sample n = (2 tid + e) + 2 TP q
      const double f = 2.0 * M_PI / (double)(size - 1);
      const int tp = (int)(size / 32);
      std::vector<float> base((size_t)tp * 4), step(32);
      for (int tdx = 0; tdx < tp; ++tdx)
        for (int ee = 0; ee < 2; ++ee) {
          base[4 * tdx + 2 * ee] = (float)std::cos(f * (2 * tdx + ee));
          base[4 * tdx + 2 * ee + 1] = (float)std::sin(f * (2 * tdx + ee));
        }
      for (int q = 0; q < 16; ++q) {
        step[2 * q] = (float)std::cos(f * 2.0 * tp * q);
        step[2 * q + 1] = (float)std::sin(f * 2.0 * tp * q);
      }
      e = hipMalloc((void **)&t.wf_base, base.size() * sizeof(float));
      if (e == hipSuccess) e = hipMemcpy(t.wf_base, base.data(), base.size() * sizeof(float), hipMemcpyHostToDevice);
      if (e == hipSuccess) e = hipMalloc((void **)&t.wf_step, step.size() * sizeof(float));
      if (e == hipSuccess) e = hipMemcpy(t.wf_step, step.data(), step.size() * sizeof(float), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess && log2n == 14 && sizeof(T) == 4) {
      // fused createWindow (fourier.ts:14-52: f = 2 pi i / (size - 1)): angle-addition tables, built in f64
      const double f = 2.0 * M_PI / (double)(size - 1);
      std::vector<float> base(256 * 4), step(64);
      for (int tdx = 0; tdx < 256; ++tdx)
        for (int ee = 0; ee < 2; ++ee) {
          base[4 * tdx + 2 * ee] = (float)std::cos(f * (2 * tdx + ee));
          base[4 * tdx + 2 * ee + 1] = (float)std::sin(f * (2 * tdx + ee));
        }
      for (int q = 0; q < 16; ++q) {
        step[2 * q] = (float)std::cos(f * 512 * q);
        step[2 * q + 1] = (float)std::sin(f * 512 * q);
        step[32 + 2 * q] = (float)std::cos(f * (512 * q + 8192));
        step[32 + 2 * q + 1] = (float)std::sin(f * (512 * q + 8192));
      }
      e = hipMalloc((void **)&t.wf_base, base.size() * sizeof(float));
      if (e == hipSuccess) e = hipMemcpy(t.wf_base, base.data(), base.size() * sizeof(float), hipMemcpyHostToDevice);
      if (e == hipSuccess) e = hipMalloc((void **)&t.wf_step, step.size() * sizeof(float));
      if (e == hipSuccess) e = hipMemcpy(t.wf_step, step.data(), step.size() * sizeof(float), hipMemcpyHostToDevice);
    }
  }
  return e;
}

int require_device() {
  int count = 0;
  const hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) {
    (void)hipGetLastError();
    return fail(PDSP_ERR_DEVICE, "no HIP device available (the pdsp engine has no CPU fallback) [hipGetDeviceCount: %s, %d]",
                hipGetErrorString(e), count);
  }
  return PDSP_OK;
}

template <int OP>
int launch_complex_op(long long count, const float *are, const float *aim, const float *bre, const float *bim,
                      long long b_len, float sre, float sim, float *ore, float *oim, hipStream_t s) {
  const bool binary = OP <= pdsp::kDiv;
  const uintptr_t align = (uintptr_t)are | (uintptr_t)aim | (uintptr_t)ore | (uintptr_t)oim |
                          (binary ? ((uintptr_t)bre | (uintptr_t)bim) : 0);
  const bool vec4 = (align & 15) == 0 && count % 4 == 0 && (!binary || b_len % 4 == 0);
  if (vec4)
    hipLaunchKernelGGL((pdsp::complex_op_kernel<float, OP, 4>), dim3(grid_for(count / 4)), dim3(256), 0, s, are, aim,
                       bre, bim, sre, sim, ore, oim, count, b_len);
  else
    hipLaunchKernelGGL((pdsp::complex_op_kernel<float, OP, 1>), dim3(grid_for(count)), dim3(256), 0, s, are, aim, bre,
                       bim, sre, sim, ore, oim, count, b_len);
  PDSP_HIP_TRY(hipGetLastError());
  return PDSP_OK;
}

template <typename T>
int spectrum_impl(const pdsp_plan *plan, long long batch, const T *frames, long long frame_len, long long frame_stride,
                  const T *window, int sides, T *amp_out, T *phase_out, int32_t *peak_idx_out, pdsp_peak32 *peaks_out,
                  double sample_rate, hipStream_t stream) {
  if (int rc = check_plan_batch(plan, batch)) return rc;
  if (sides != PDSP_SIDES_ONE && sides != PDSP_SIDES_TWO) return fail(PDSP_ERR_BAD_ARG, "bad sides %d", sides);
  // frame_stride < frame_len = overlapping frames of one signal (an STFT with hop = frame_stride): rows are only read
  if (frame_len < 0 || frame_stride < 1) return fail(PDSP_ERR_BAD_ARG, "bad frame_len/frame_stride");
  if (peaks_out && sample_rate <= 0)
    return fail(PDSP_ERR_SAMPLE_RATE, "Sample rate must be positive, got %.17g", sample_rate);
  if (batch == 0) return PDSP_OK;
  if (!frames || (!amp_out && !peaks_out)) return fail(PDSP_ERR_BAD_ARG, "null buffer");
  if ((phase_out || peak_idx_out) && !amp_out) return fail(PDSP_ERR_BAD_ARG, "phase/peak index output needs amp_out");
  const Tables<T> &t = tables<T>(plan);
  if (!t.tw && !t.tw_half)
    return fail(PDSP_ERR_UNSUPPORTED_SIZE, "FFT size %lld exceeds the %d-bit limit %d", plan->n, (int)(8 * sizeof(T)),
                pdsp_max_size((int)sizeof(T)));
  DeviceGuard g(plan->device);
  PDSP_HIP_TRY(g.err);
  static_assert(sizeof(pdsp_peak32) == sizeof(pdsp::PeakRec), "peak record layout");
  const long long n = plan->n;
  const int bins = (int)(sides == PDSP_SIDES_ONE ? n / 2 + 1 : n);
  const long long used = frame_len < n ? frame_len : n;
  const T freq_scale = peaks_out ? (T)(sample_rate / (double)n) : T(0);
  if (used == 0) {  // an empty frame is all zeros: amplitude 0, atan2(0, 0) = 0, peak 0
    if (amp_out) PDSP_HIP_TRY(hipMemsetAsync(amp_out, 0, (size_t)batch * bins * sizeof(T), stream));
    if (phase_out) PDSP_HIP_TRY(hipMemsetAsync(phase_out, 0, (size_t)batch * bins * sizeof(T), stream));
    if (peak_idx_out) PDSP_HIP_TRY(hipMemsetAsync(peak_idx_out, 0, (size_t)batch * sizeof(int32_t), stream));
    if (peaks_out) PDSP_HIP_TRY(hipMemsetAsync(peaks_out, 0, (size_t)batch * sizeof(pdsp_peak32), stream));
    return PDSP_OK;
  }
  const T s_edge = T(1) / (T)n, s_mid = (sides == PDSP_SIDES_ONE ? T(2) : T(1)) / (T)n;
  if constexpr (sizeof(T) == 4) {
    // N beyond the single-pass limit, whole 16-byte aligned frames: the packed-real form on tile passes.
    // z[m] = (x*w)[2m] + i (x*w)[2m+1] is read straight from the frame (and the window table) by the first
    // pass; two (N <= 2^18) or three passes of the N/2-point transform (one pass of fft_split4_kernel at
    // N = 2^15); split_amp_rows_kernel undoes the packing on the way to the amplitude (+ phase) rows.  HBM
    // bytes per sample: 4+4, 4+4 (, 4+4), 4+2 = 22 (30) where the four-step forms on (x*w, 0) move 38 (70).  The four-step forms stay for partial /
    // unaligned frames and f64.
    if (t.log2n1 > 0 && t.hp_np && (g_twopass & 1) && used == n && (frame_stride & 3) == 0 &&
        (((uintptr_t)frames | (uintptr_t)window) & 15) == 0) {
      T *amp = amp_out, *ph = phase_out;
      const long long m = n / 2;
      const size_t plane = (size_t)batch * (size_t)m, rows = (size_t)batch * bins;
      const size_t extra = (peaks_out && !amp ? rows : 0) + (peaks_out && !ph ? rows : 0);
      StreamScratch mem(stream);
      PDSP_HIP_TRY(mem.alloc((4 * plane + extra) * sizeof(T)));
      T *const sc = (T *)mem.p;
      if (peaks_out && !amp) amp = sc + 4 * plane;
      if (peaks_out && !ph) ph = sc + 4 * plane + (amp_out ? 0 : rows);
      // pass chain: frames -> s1 (-> s2) -> Z; two passes: Z = s2; three passes: Z = s1 again
      T *const s1_re = sc, *const s1_im = sc + plane, *const s2_re = sc + 2 * plane, *const s2_im = sc + 3 * plane;
      T *const z_re = t.hp_np == 2 ? s2_re : s1_re, *const z_im = t.hp_np == 2 ? s2_im : s1_im;
      // a window that is one of the plan's own tables (pdsp_plan_window_f32) is known by kind: the cosine sum is
      // then evaluated in the first pass instead of being read back (4 more bytes per sample).
      // first = tile_pass_kernel's IN: 3 rect, 4 window table, 5 / 6 fused two- / three-term window
      int first = window ? 4 : 3;
      pdsp::TileGeom fw{};
      int kind = -1;
      for (int k = 0; k < 4; ++k)
        if (window && window == t.win[k]) kind = k;
      if (kind == PDSP_WIN_RECT) first = 3;  // createWindow("rect") is all ones
      if (t.hp_win && g_fused_window && kind > PDSP_WIN_RECT) {
        fw.wa = t.hp_win, fw.wb = fw.wa + 2 * t.hp_win_a, fw.wstep = fw.wb + 2 * 512, fw.we = fw.wstep + 2 * 8;
        if (kind == PDSP_WIN_HANN) first = 5, fw.k0 = 0.5f, fw.k1 = -0.5f;
        else if (kind == PDSP_WIN_HAMMING) first = 5, fw.k0 = 0.54f, fw.k1 = -0.46f;
        else if (kind == PDSP_WIN_BLACKMAN) first = 6, fw.k0 = 0.42f - 0.08f, fw.k1 = -0.5f, fw.k2 = 2 * 0.08f;
      }
      if (plan->log2n == 15 && t.tws4 && t.tw12 && g_split16k) {
        // N = 32768: the 16384-point transform is one pass of fft_split4_kernel (14 bytes per sample in all)
        const pdsp::StoreComplex<T> st{z_re, z_im, m, T(1)};
#define PDSP_SPLIT4_PACKED(W)                                                                                         \
  hipLaunchKernelGGL((pdsp::fft_split4_kernel<T, 12, pdsp::LoadPackedFrames<T, W>, pdsp::StoreComplex<T>>),           \
                     dim3((unsigned)batch), dim3(256), 0, stream,                                                     \
                     pdsp::LoadPackedFrames<T, W>{frames, window, frame_stride, fw.wb, fw.we + 2 * 8, fw.we, fw.k0,   \
                                                  fw.k1, fw.k2},                                                      \
                     st, t.tw12, t.tws4, batch)
        switch (first) {
          case 3: PDSP_SPLIT4_PACKED(0); break;
          case 4: PDSP_SPLIT4_PACKED(1); break;
          case 5: PDSP_SPLIT4_PACKED(2); break;
          default: PDSP_SPLIT4_PACKED(3); break;
        }
#undef PDSP_SPLIT4_PACKED
        PDSP_HIP_TRY(hipGetLastError());
      } else if (plan->log2n == 16 && g_twopass == 1 && t.tws4 && t.tw12) {
        // N = 65536: the 32768-point transform in ONE pass by two sibling workgroups per frame that share an XCD's
        // L2 (fft_paired_kernel, packed loader): 14 bytes per sample in all, where the two tile passes move 22
        const long long blocks = ((batch + 7) / 8) * 8 * 2;
        if (blocks > 0x7fffffffLL) return fail(PDSP_ERR_BAD_ARG, "batch too large: %lld", batch);
        const pdsp::PairedPacked pk{frame_stride, fw.wb, fw.we + 2 * 8, fw.we, fw.k0, fw.k1, fw.k2};
        const pdsp::cx<T> *twa = reinterpret_cast<const pdsp::cx<T> *>(t.twa);
        const pdsp::cx<T> *twb = reinterpret_cast<const pdsp::cx<T> *>(t.twb);
#define PDSP_PAIRED_PACKED(PK)                                                                                       \
  hipLaunchKernelGGL((pdsp::fft_paired_kernel<T, 1, false, PK>), dim3((unsigned)blocks), dim3(256), 0, stream, frames,   \
                     first == 4 ? window : (const T *)nullptr, z_re, z_im, t.tw12, t.tws4, twa, twb, T(1), batch, pk)
        switch (first) {
          case 3: PDSP_PAIRED_PACKED(1); break;
          case 4: PDSP_PAIRED_PACKED(2); break;
          case 5: PDSP_PAIRED_PACKED(3); break;
          default: PDSP_PAIRED_PACKED(4); break;
        }
#undef PDSP_PAIRED_PACKED
        PDSP_HIP_TRY(hipGetLastError());
      } else {
        if (int rc = tilepass_chain<T>(t, m, t.hp_np, t.hp_l, t.hp_tw, 1u, first, batch, frames, first == 4 ? window : nullptr,
                                       frame_stride, z_re, z_im, T(1), s1_re, s1_im, s2_re, s2_im, stream,
                                       first >= 5 ? &fw : nullptr))
          return rc;
      }
      const long long chunks = m / 2048;  // 256 lanes of four pairs each
      if (batch * chunks > 0x7fffffffLL) return fail(PDSP_ERR_BAD_ARG, "batch too large: %lld", batch);
      hipLaunchKernelGGL((pdsp::split_amp_rows_kernel<T>), dim3((unsigned)(batch * chunks)), dim3(256), 0, stream,
                         (const T *)z_re, (const T *)z_im, amp, ph, reinterpret_cast<const pdsp::cx<T> *>(t.twa),
                         reinterpret_cast<const pdsp::cx<T> *>(t.twb), (int)m, bins, s_edge, s_mid, batch);
      PDSP_HIP_TRY(hipGetLastError());
      if ((peaks_out || peak_idx_out) &&
          launch_peaks<T>(amp, ph, bins, freq_scale, peak_idx_out, peaks_out, batch, stream) != hipSuccess)
        return fail(PDSP_ERR_DEVICE, "peak kernel launch failed");
      return PDSP_OK;
    }
  }
  // N beyond the single-pass limit: four-step on (x*w, 0), amplitude rows in the last pass.  (Not where the packed-real
  // tables exist: f64 frames of N = 16384 are ONE 8192-point packed transform -- spectrum_packed_kernel<double, 13> --
  // although the complex f64 transform of that size is a four-step one.  Round 2 sent them through the four-step
  // path by this test's order.)
  if (t.log2n1 > 0 && !t.tw_half) {
    const bool big = t.log2n1 > pdsp::kMaxLog2N1;  // general path: two scratch pairs
    T *amp = amp_out, *ph = phase_out;
    const size_t plane = (size_t)batch * (size_t)n, rows = (size_t)batch * bins, planes = big ? 4 : 2;
    const size_t extra = (peaks_out && !amp ? rows : 0) + (peaks_out && !ph ? rows : 0);
    StreamScratch mem(stream);
    PDSP_HIP_TRY(mem.alloc((planes * plane + extra) * sizeof(T)));
    T *const scratch = (T *)mem.p;
    if (peaks_out && !amp) amp = scratch + planes * plane;
    if (peaks_out && !ph) ph = scratch + planes * plane + (amp_out ? 0 : rows);
    const int nyq = (sides == PDSP_SIDES_ONE) ? (int)(n / 2) : -1;
    int rc;
    if (big) {
      rc = bigfft_rows<T>(plan, batch, frames, nullptr, window, frame_stride, used, scratch + 2 * plane,
                          scratch + 3 * plane, scratch, scratch + plane, stream);
      if (!rc) rc = bigfft_out<T, true>(plan, batch, scratch, scratch + plane, amp, ph, T(1), bins, nyq, s_edge, s_mid, stream);
    } else {
      rc = fourstep_ab<T>(plan, batch, frames, nullptr, window, frame_stride, used, scratch, scratch + plane, stream);
      if (!rc) rc = fourstep_c<T, 1>(plan, batch, scratch, scratch + plane, amp, ph, T(1), bins, nyq, s_edge, s_mid, stream);
    }
    if (!rc && (peaks_out || peak_idx_out) &&
        launch_peaks<T>(amp, ph, bins, freq_scale, peak_idx_out, peaks_out, batch, stream) != hipSuccess)
      rc = fail(PDSP_ERR_DEVICE, "peak kernel launch failed");
    return rc;
  }
  constexpr uintptr_t kPairMask = 2 * sizeof(T) - 1;  // alignment of one (re, im) pair
  if (t.tw_half) {
    // packed-real path (N >= 64): N/2-point complex transform + Hermitian split (+ findPeak) fused with the store.
    // fast variant: whole pair-aligned frames (and window), one-sided, no phase rows (config 4's shape); the
    // general variant takes any frame length, stride and alignment of frames and window
    const bool fast = ((uintptr_t)frames & kPairMask) == 0 && (frame_stride & 1) == 0 && used == n &&
                      sides == PDSP_SIDES_ONE && phase_out == nullptr && ((uintptr_t)window & kPairMask) == 0;
    // 64 <= N <= 512, amplitude only: contiguous frames staged in / amplitude rows staged out through LDS
    // (f32 only: in f64 the two LDS regions take 102 KB, one workgroup per CU, and measure slower than the direct kernel)
    if (sizeof(T) == 4 && fast && !peaks_out && !peak_idx_out && plan->log2n >= 6 && plan->log2n <= 9 && g_staged_small &&
        frame_stride == n &&
        ((uintptr_t)frames & (4 * sizeof(T) - 1)) == 0 && ((uintptr_t)window & (4 * sizeof(T) - 1)) == 0) {
      const long long blocks = (batch * (n / 2) + 4095) / 4096;
#define PDSP_SSTAGED(LM)                                                                                            \
  do {                                                                                                              \
    if (window)                                                                                                     \
      hipLaunchKernelGGL((pdsp::spectrum_staged_kernel<T, LM, true>), dim3((unsigned)blocks), dim3(256), 0, stream,  \
                         frames, window, t.tw_half, t.twr, amp_out, s_edge, s_mid, batch);                          \
    else                                                                                                            \
      hipLaunchKernelGGL((pdsp::spectrum_staged_kernel<T, LM, false>), dim3((unsigned)blocks), dim3(256), 0, stream, \
                         frames, window, t.tw_half, t.twr, amp_out, s_edge, s_mid, batch);                          \
  } while (0)
      switch (plan->log2n - 1) {
        case 5: PDSP_SSTAGED(5); break;
        case 6: PDSP_SSTAGED(6); break;
        case 7: PDSP_SSTAGED(7); break;
        default: PDSP_SSTAGED(8); break;
      }
#undef PDSP_SSTAGED
      PDSP_HIP_TRY(hipGetLastError());
      return PDSP_OK;
    }
    bool launched = false;
    // A window that is one of the PLAN'S OWN tables (pdsp_plan_window_f32) is known by kind: the kernels
    // that can (whole f32 frames, N = 1024 ... 16384) then evaluate the cosine sum in registers.
    // wmode: 0 rect, 1 table, 2 / 3 fused two- / three-term cosine sum.
    pdsp::WinFused wf{nullptr, nullptr, 0.f, 0.f, 0.f, 1.f};
    int wmode = window ? 1 : 0;
    if constexpr (sizeof(T) == 4) {
      int kind = -1;  // -1: caller's table
      for (int k = 0; k < 4; ++k)
        if (window && window == t.win[k]) kind = k;
      if (kind == PDSP_WIN_RECT) wmode = 0;  // createWindow("rect") is all ones
      if (t.wf_base && g_fused_window && fast) {
        wf.base = t.wf_base, wf.step = t.wf_step;
        if (kind == PDSP_WIN_HANN) wmode = 2, wf.k0 = 0.5f, wf.k1 = -0.5f;
        else if (kind == PDSP_WIN_HAMMING) wmode = 2, wf.k0 = 0.54f, wf.k1 = -0.46f;
        else if (kind == PDSP_WIN_BLACKMAN) wmode = 3, wf.k0 = 0.42f - 0.08f, wf.k1 = -0.5f, wf.k2 = 2 * 0.08f;
        if (wmode >= 2) {  // the kernels take the fused coefficients pre-scaled by s_mid / 2 (a power of two: exact)
          const float g = 0.5f * (float)s_mid;
          wf.k0 *= g, wf.k1 *= g, wf.k2 *= g;
          wf.edge_ratio = (float)(s_edge / s_mid);
        }
      }
    }
    if constexpr (sizeof(T) == 4) {
      // N = 16384: two 4096-point sub-transforms per 256-thread workgroup (3 frames per CU instead of 2),
      // decimation in frequency on top (spectrum_dif16k_kernel).  A window that is one of the PLAN'S OWN
      // tables (pdsp_plan_window_f32) is known by kind, and createWindow is fused into the kernel: the
      // reference's windows are cosine sums (fourier.ts:14-52), evaluated in registers instead of being
      // read back, 64 KB per frame, from L2.  Any other window pointer is read as a table.
      if (fast && plan->log2n == 14 && g_split16k && t.wf_base) {
        pdsp::PeakRec *pk = reinterpret_cast<pdsp::PeakRec *>(peaks_out);
        const int mode = wmode;
#define PDSP_DIF(W, P)                                                                                              \
  hipLaunchKernelGGL((pdsp::spectrum_dif16k_kernel<T, W, P>), dim3((unsigned)batch), dim3(256), 0, stream, frames,  \
                     window, wf, frame_stride, t.tw12, t.twr, amp_out, s_edge, s_mid, pk, freq_scale, batch)
#define PDSP_DIF_P(W)    \
  do {                   \
    if (pk) PDSP_DIF(W, true); \
    else PDSP_DIF(W, false);   \
  } while (0)
        switch (mode) {
          case 0: PDSP_DIF_P(0); break;
          case 1: PDSP_DIF_P(1); break;
          case 2: PDSP_DIF_P(2); break;
          default: PDSP_DIF_P(3); break;
        }
#undef PDSP_DIF_P
#undef PDSP_DIF
        PDSP_HIP_TRY(hipGetLastError());
        launched = true;  // a requested peak-index array is filled by the common tail below
      }
    }
    if (!launched)
      PDSP_HIP_TRY(launch_packed<T>(plan->log2n - 1, fast, frames, wmode == 0 ? (const T *)nullptr : window,
                                    (wmode >= 2 && plan->log2n == 14) ? 1 : wmode, wf, used, frame_stride, t.tw_half, t.twr,
                                    amp_out, phase_out, sides == PDSP_SIDES_TWO ? 1 : 0, s_edge, s_mid,
                                    reinterpret_cast<pdsp::PeakRec *>(peaks_out), freq_scale, batch, stream));
  } else {
    // complex kernel on (x, 0) for N < 64 (the sizes without packed-real tables); peaks come from the stored rows
    if (!t.tw || plan->log2n > 5) return fail(PDSP_ERR_UNSUPPORTED_SIZE, "no spectrum tables for size %lld", plan->n);
    if (plan->log2n >= 1 && plan->log2n <= 5 && g_staged_small && used == n && frame_stride == n && amp_out &&
        !phase_out && !peaks_out && ((uintptr_t)frames & (4 * sizeof(T) - 1)) == 0) {
      // 2 <= N <= 32, whole contiguous frames, amplitude only: one thread per frame, chunk staged through LDS
      pdsp::LoadReal<T> ld{frames, n};
      PDSP_HIP_TRY((launch_tiny<T, true>(plan->log2n, ld, window, amp_out, (T *)nullptr, T(1), bins,
                                         sides == PDSP_SIDES_ONE ? (int)(n / 2) : -1, s_edge, s_mid, batch, stream)));
      if (peak_idx_out)
        PDSP_HIP_TRY(launch_peaks<T>(amp_out, (const T *)nullptr, bins, T(0), peak_idx_out, nullptr, batch, stream));
      return PDSP_OK;
    }
    T *amp = amp_out, *ph = phase_out;
    const size_t row_bytes = (size_t)batch * bins * sizeof(T);
    StreamScratch tmp_amp(stream), tmp_ph(stream);  // peaks-only output: the rows live in scratch
    if (peaks_out && !amp) {
      PDSP_HIP_TRY(tmp_amp.alloc(row_bytes));
      amp = (T *)tmp_amp.p;
    }
    if (peaks_out && !ph) {
      PDSP_HIP_TRY(tmp_ph.alloc(row_bytes));
      ph = (T *)tmp_ph.p;
    }
    pdsp::StoreAmplitude<T> st{amp, ph, bins,
                               // scaleAmplitudeOneSided: `nyquist = size % 2 === 0 ? size/2 : -1`; N = 1 is odd
                               (sides == PDSP_SIDES_ONE && n % 2 == 0) ? (int)(n / 2) : -1, s_edge, s_mid};
    if (window) {
      pdsp::LoadFrameWindowed<T, true> ld{frames, window, used, frame_stride};
      PDSP_HIP_TRY(launch_fft_small<T>(plan->log2n, ld, st, t.tw, batch, stream));
    } else {
      pdsp::LoadFrameWindowed<T, false> ld{frames, window, used, frame_stride};
      PDSP_HIP_TRY(launch_fft_small<T>(plan->log2n, ld, st, t.tw, batch, stream));
    }
    if (peaks_out) PDSP_HIP_TRY(launch_peaks<T>(amp, ph, bins, freq_scale, nullptr, peaks_out, batch, stream));
  }
  if (peak_idx_out)
    PDSP_HIP_TRY(launch_peaks<T>(amp_out, (const T *)nullptr, bins, T(0), peak_idx_out, nullptr, batch, stream));
  return PDSP_OK;
}

// ---- chunked host calls -------------------------------------------------------------------------------------
// A batched host-f64 call moves every sample through the CPU twice (the caller's f64 rows <-> pinned staging), over
// PCIe twice, and through a kernel that needs a few percent of that time.  As ONE stage -> copy -> launch -> copy ->
// unstage sequence each step waits for the one before it and one core does all the staging: ~0.4 GSample/s at
// N = 4096 whatever the card does.  Calls above kChunkedMinBytes of staging are cut into chunks of ~kChunkInBytes of
// input rows instead; K workers (the calling thread and K - 1 helpers that live for the call) each own one staging
// slot and one stream and draw chunks from a shared counter: fill the slot, H2D, the same kernels the one-shot path
// launches, D2H, wait for the stream, unstage (+ findPeak).  Inside a worker the steps stay in order; across workers
// staging, both PCIe directions and the kernels overlap.  Row b of the result is the one-shot result of row b bit
// for bit (the kernels work row by row and the variant does not depend on the row count).
// PDSP_HOST_THREADS sets K (1 = the one-shot sequence; default: half the cores this process may run on, 2 ... 6).
constexpr size_t kChunkInBytes = (size_t)2 << 20;
constexpr size_t kChunkedMinBytes = (size_t)8 << 20;

// multipass: the size runs on the multi-pass paths, whose scratch planes come from the engine's stream-ordered pool --
// planes freed on one stream are not reusable on another before a synchronisation, so many streams grow the pool
// through the driver instead of overlapping (f64 N = 16384 rows: 20 ms on 2 workers, 49-57 ms on 4-8): two workers.
int host_workers(bool multipass = false) {
  long v = 0;
  if (const char *e = getenv("PDSP_HOST_THREADS")) v = atol(e);
  if (v <= 0) {
    const unsigned hc = std::thread::hardware_concurrency();
    v = hc ? (long)(hc / 2) : 2;
    if (v < 2) v = 2;
    if (v > 6) v = 6;
  }
  if (multipass && v > 2) v = 2;
  return (int)(v > 16 ? 16 : v);
}

inline bool host_ranges_overlap(const void *a, size_t a_bytes, const void *b, size_t b_bytes) {
  const uintptr_t a0 = (uintptr_t)a, b0 = (uintptr_t)b;
  return a && b && a_bytes && b_bytes && a0 < b0 + b_bytes && b0 < a0 + a_bytes;
}

struct ChunkJob {
  long long first = 0, count = 0;  // rows [first, first + count) of the call
  int slot = 0;                    // staging slot of the worker that runs it
  hipStream_t stream = nullptr;
};

// body(job) -> pdsp status, error text in the running thread's g_err.  Caller holds plan->mu and has staged
// `workers` slots (ensure_stage).  The first failure stops the hand-out of chunks and is what the call returns.
template <class Body>
int run_chunked(pdsp_plan *plan, long long rows, long long rows_per_chunk, int workers, Body body) {
  const long long nchunks = (rows + rows_per_chunk - 1) / rows_per_chunk;
  if (workers > nchunks) workers = (int)nchunks;
  while ((long long)plan->slot_streams.size() < workers) {
    hipStream_t st = nullptr;
    PDSP_HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    plan->slot_streams.push_back(st);
  }
  std::atomic<long long> next{0};
  std::atomic<int> status{PDSP_OK};
  std::mutex err_mu;
  std::string err_text;
  auto report = [&](int rc) {
    std::lock_guard<std::mutex> lk(err_mu);
    if (status.load() == PDSP_OK) {
      err_text = g_err;
      status.store(rc);
    }
  };
  auto worker = [&](int w) {
    const hipStream_t st = plan->slot_streams[(size_t)w];
    const hipError_t e = hipSetDevice(plan->device);  // helpers start on device 0
    if (e != hipSuccess) {
      report(fail(PDSP_ERR_DEVICE, "HIP error %d (%s) at hipSetDevice", (int)e, hipGetErrorString(e)));
      return;
    }
    while (status.load() == PDSP_OK) {
      const long long c = next.fetch_add(1);
      if (c >= nchunks) break;
      ChunkJob job;
      job.first = c * rows_per_chunk;
      job.count = rows - job.first < rows_per_chunk ? rows - job.first : rows_per_chunk;
      job.slot = w;
      job.stream = st;
      if (const int rc = body(job)) {
        report(rc);
        break;
      }
    }
    (void)hipStreamSynchronize(st);  // nothing of this call stays in flight on the slot, on any exit
  };
  std::vector<std::thread> helpers;
  helpers.reserve((size_t)workers);
  for (int w = 1; w < workers; ++w) {
    try {
      helpers.emplace_back(worker, w);
    } catch (...) {
      break;  // no more threads to be had: the workers that did start share the chunks
    }
  }
  worker(0);
  for (std::thread &t : helpers) t.join();
  if (status.load() != PDSP_OK) {
    g_err = err_text;
    return status.load();
  }
  return PDSP_OK;
}

template <typename T>
inline void rows_to_stage(T *dst, const double *src, size_t count) {
  if constexpr (sizeof(T) == sizeof(double)) std::memcpy(dst, src, count * sizeof(double));
  else
    for (size_t i = 0; i < count; ++i) dst[i] = (T)src[i];
}
template <typename T>
inline void stage_to_rows(double *dst, const T *src, size_t count) {
  if constexpr (sizeof(T) == sizeof(double)) std::memcpy(dst, src, count * sizeof(double));
  else
    for (size_t i = 0; i < count; ++i) dst[i] = (double)src[i];
}

// Radix2Fft.transform for `batch` host rows in precision T (f64 at the boundary either way).  Input rows either
// contiguous (re_in / im_in) or one pointer per row (re_rows / im_rows: pdsp_fft_transform_rows_host_f64).
template <typename T>
int transform_host(pdsp_plan *plan, long long batch, const double *re_in, const double *im_in,
                   const double *const *re_rows, const double *const *im_rows, double *re_out, double *im_out,
                   int inverse) {
  const size_t n = (size_t)plan->n, cnt = (size_t)batch * n;
  const bool has_im = im_in || im_rows;
  auto re_row = [&](long long r) { return re_rows ? re_rows[r] : re_in + (size_t)r * n; };
  auto im_row = [&](long long r) { return im_rows ? im_rows[r] : im_in + (size_t)r * n; };
  // rows [first, first + count) into staging planes of `count` rows each
  auto stage_in = [&](T *h_re, T *h_im, long long first, long long count) {
    if (!re_rows) rows_to_stage<T>(h_re, re_in + (size_t)first * n, (size_t)count * n);
    else
      for (long long r = 0; r < count; ++r) rows_to_stage<T>(h_re + (size_t)r * n, re_rows[first + r], n);
    if (!has_im) return;
    if (!im_rows) rows_to_stage<T>(h_im, im_in + (size_t)first * n, (size_t)count * n);
    else
      for (long long r = 0; r < count; ++r) rows_to_stage<T>(h_im + (size_t)r * n, im_rows[first + r], n);
  };
  {
    // many rows: chunks on several workers (run_chunked); planes that overlap each other in host memory keep the
    // one-shot sequence, which has read every input before it writes any output
    const size_t row_bytes = n * sizeof(T), out_bytes = cnt * sizeof(double), in_row_bytes = n * sizeof(double);
    const long long per_chunk = (long long)(kChunkInBytes / ((has_im ? 2 : 1) * row_bytes));
    const int workers = host_workers(tables<T>(plan).log2n1 > 0);
    bool overlap = false;
    if (re_rows || im_rows) {
      for (long long r = 0; r < batch && !overlap; ++r)
        overlap = host_ranges_overlap(re_row(r), in_row_bytes, re_out, out_bytes) ||
                  host_ranges_overlap(re_row(r), in_row_bytes, im_out, out_bytes) ||
                  (has_im && (host_ranges_overlap(im_row(r), in_row_bytes, re_out, out_bytes) ||
                              host_ranges_overlap(im_row(r), in_row_bytes, im_out, out_bytes)));
    } else {
      overlap = host_ranges_overlap(re_in, out_bytes, re_out, out_bytes) || host_ranges_overlap(re_in, out_bytes, im_out, out_bytes) ||
                host_ranges_overlap(im_in, out_bytes, re_out, out_bytes) || host_ranges_overlap(im_in, out_bytes, im_out, out_bytes);
    }
    if (workers >= 2 && per_chunk >= 1 && batch >= 2 * per_chunk && 4 * cnt * sizeof(T) >= kChunkedMinBytes && !overlap) {
      const size_t slot = 4 * (size_t)per_chunk * n;  // elements: re | im | out re | out im
      const int k = (long long)workers < (batch + per_chunk - 1) / per_chunk ? workers : (int)((batch + per_chunk - 1) / per_chunk);
      if (int rc = ensure_stage(plan, (size_t)k * slot * sizeof(T))) return rc;
      return run_chunked(plan, batch, per_chunk, k, [&](const ChunkJob &job) -> int {
        const size_t c = (size_t)job.count * n, off = (size_t)job.first * n;
        T *h = (T *)plan->h_stage + (size_t)job.slot * slot, *d = (T *)plan->d_stage + (size_t)job.slot * slot;
        stage_in(h, h + c, job.first, job.count);
        PDSP_HIP_TRY(hipMemcpyAsync(d, h, (has_im ? 2 : 1) * c * sizeof(T), hipMemcpyHostToDevice, job.stream));
        int rc;
        if (inverse) rc = run_complex<T>(plan, job.count, d + c, d, d + 3 * c, d + 2 * c, T(1) / (T)plan->n, job.stream);
        else rc = run_complex<T>(plan, job.count, d, has_im ? d + c : nullptr, d + 2 * c, d + 3 * c, T(1), job.stream);
        if (rc) return rc;
        PDSP_HIP_TRY(hipMemcpyAsync(h + 2 * c, d + 2 * c, 2 * c * sizeof(T), hipMemcpyDeviceToHost, job.stream));
        PDSP_HIP_TRY(hipStreamSynchronize(job.stream));
        stage_to_rows<T>(re_out + off, h + 2 * c, c);
        stage_to_rows<T>(im_out + off, h + 3 * c, c);
        return PDSP_OK;
      });
    }
  }
  if (int rc = ensure_stage(plan, 4 * cnt * sizeof(T))) return rc;
  T *h_re = (T *)plan->h_stage, *h_im = h_re + cnt, *h_ore = h_im + cnt, *h_oim = h_ore + cnt;
  T *d_re = (T *)plan->d_stage, *d_im = d_re + cnt, *d_ore = d_im + cnt, *d_oim = d_ore + cnt;
  stage_in(h_re, h_im, 0, batch);
  hipStream_t s = plan->stream;
  T *const z = zero_copy(4 * cnt * sizeof(T)) ? stage_device_view<T>(plan) : nullptr;
  if (z) {  // the kernels work on the pinned buffer itself
    d_re = z, d_im = z + cnt, d_ore = z + 2 * cnt, d_oim = z + 3 * cnt;
  } else {
    PDSP_HIP_TRY(hipMemcpyAsync(d_re, h_re, (has_im ? 2 : 1) * cnt * sizeof(T), hipMemcpyHostToDevice, s));
  }
  int rc;
  // inverse: conj(FFT(conj(z))) == swap(FFT(swap(z))) -- the conjugated-twiddle sweep of fft.ts:122
  // is the forward kernel with the planes exchanged on the way in and out; 1/N rides on the store
  if (inverse) rc = run_complex<T>(plan, batch, d_im, d_re, d_oim, d_ore, T(1) / (T)plan->n, s);
  else rc = run_complex<T>(plan, batch, d_re, has_im ? d_im : nullptr, d_ore, d_oim, T(1), s);
  if (rc) return rc;
  if (!z) PDSP_HIP_TRY(hipMemcpyAsync(h_ore, d_ore, 2 * cnt * sizeof(T), hipMemcpyDeviceToHost, s));
  PDSP_HIP_TRY(hipStreamSynchronize(s));
  stage_to_rows<T>(re_out, h_ore, cnt);
  stage_to_rows<T>(im_out, h_oim, cnt);
  return PDSP_OK;
}

template <typename T>
int apply_window_dev(long long batch, long long n, const T *in, const T *window, T *out, hipStream_t s) {
  const long long total = batch * n;
  if (total == 0) return PDSP_OK;
  if (!in || !window || !out) return fail(PDSP_ERR_BAD_ARG, "null buffer");
  hipLaunchKernelGGL((pdsp::apply_window_kernel<T>), dim3(grid_for(total)), dim3(256), 0, s, in, window, out, total, n);
  PDSP_HIP_TRY(hipGetLastError());
  return PDSP_OK;
}

template <typename T, bool PHASE>
int polar_dev(long long count, const T *re, const T *im, T *out, hipStream_t s) {
  if (count < 0) return fail(PDSP_ERR_BAD_ARG, "negative size");
  if (count == 0) return PDSP_OK;
  if (!re || !im || !out) return fail(PDSP_ERR_BAD_ARG, "null buffer");
  hipLaunchKernelGGL((pdsp::polar_kernel<T, PHASE>), dim3(grid_for(count)), dim3(256), 0, s, re, im, out, count);
  PDSP_HIP_TRY(hipGetLastError());
  return PDSP_OK;
}

template <typename T>
int apply_window_host(const double *in, long long n_, const double *window, double *out) {
  Scratch<T> sc;
  const size_t n = (size_t)n_;
  if (int rc = sc.reserve(3 * n)) return rc;
  for (size_t i = 0; i < n; ++i) sc.h[i] = (T)in[i];
  for (size_t i = 0; i < n; ++i) sc.h[n + i] = (T)window[i];
  PDSP_HIP_TRY(hipMemcpy(sc.d, sc.h, 2 * n * sizeof(T), hipMemcpyHostToDevice));
  if (int rc = apply_window_dev<T>(1, n_, sc.d, sc.d + n, sc.d + 2 * n, nullptr)) return rc;
  PDSP_HIP_TRY(hipMemcpy(sc.h, sc.d + 2 * n, n * sizeof(T), hipMemcpyDeviceToHost));
  for (size_t i = 0; i < n; ++i) out[i] = (double)sc.h[i];
  return PDSP_OK;
}

template <typename T>
int polar_host_t(const double *re, const double *im, long long n_, double *out, bool want_phase) {
  Scratch<T> sc;
  const size_t n = (size_t)n_;
  if (int rc = sc.reserve(3 * n)) return rc;
  for (size_t i = 0; i < n; ++i) sc.h[i] = (T)re[i];
  for (size_t i = 0; i < n; ++i) sc.h[n + i] = (T)im[i];
  PDSP_HIP_TRY(hipMemcpy(sc.d, sc.h, 2 * n * sizeof(T), hipMemcpyHostToDevice));
  const int rc = want_phase ? polar_dev<T, true>(n_, sc.d, sc.d + n, sc.d + 2 * n, nullptr)
                            : polar_dev<T, false>(n_, sc.d, sc.d + n, sc.d + 2 * n, nullptr);
  if (rc) return rc;
  PDSP_HIP_TRY(hipMemcpy(sc.h, sc.d + 2 * n, n * sizeof(T), hipMemcpyDeviceToHost));
  for (size_t i = 0; i < n; ++i) out[i] = (double)sc.h[i];
  return PDSP_OK;
}

// `batch` frames of spectrum() (each `len` samples, contiguous) in precision T; plan->mu held by the
// caller.  One frame (the drop-in spectrum()) and many (spectrumBatch) run the same kernel variant per
// row, so row b of a batch equals the one-frame call on frame b bit for bit.
// findPeak on the host over the f64 amplitudes of rows [first, first + count): exact strict-'>' and first-wins
// behaviour (spectrum.ts:74-105), peak.frequency from the call's one frequency axis.
inline void host_peaks(const double *freq, const double *amp_out, const double *phase_out, long long bins,
                       long long first, long long count, pdsp_peak *peak_out) {
  for (long long b = first; b < first + count; ++b) {
    const double *a = amp_out + (size_t)b * (size_t)bins, *p = phase_out + (size_t)b * (size_t)bins;
    const long long pk = pdsp_find_peak_f64(a, bins);
    peak_out[b].index = (int32_t)pk;
    peak_out[b].frequency = freq[pk];
    peak_out[b].amplitude = a[pk];
    peak_out[b].phase = p[pk];
  }
}

// Frame b starts at row_ptrs[b] when row_ptrs is given (pdsp_spectrum_rows_host_f64), else at samples + b * len.
template <typename T>
int spectrum_host_t(pdsp_plan *plan, const double *samples, const double *const *row_ptrs, long long len, int window,
                    int sides, double *amp_out, double *phase_out, long long batch, const double *freq,
                    pdsp_peak *peak_out) {
  auto frame = [&](long long b) { return row_ptrs ? row_ptrs[b] : samples + (size_t)b * (size_t)len; };
  const long long n = plan->n;
  const long long bins = sides == PDSP_SIDES_ONE ? n / 2 + 1 : n;
  const long long used = len < n ? len : n;
  {
    // many frames: chunks on several workers (run_chunked)
    const size_t frame_bytes = (size_t)n * sizeof(T);
    const long long per_chunk = (long long)(kChunkInBytes / frame_bytes);
    const int workers = host_workers(tables<T>(plan).log2n1 > 0 && !tables<T>(plan).tw_half);  // packed-real frames: one pass
    const size_t in_b = (size_t)batch * (size_t)(len > 0 ? len : 1) * sizeof(double), out_b = (size_t)batch * (size_t)bins * sizeof(double);
    bool overlap = false;
    if (row_ptrs) {
      for (long long b = 0; b < batch && !overlap; ++b)
        overlap = host_ranges_overlap(row_ptrs[b], (size_t)len * sizeof(double), amp_out, out_b) ||
                  host_ranges_overlap(row_ptrs[b], (size_t)len * sizeof(double), phase_out, out_b);
    } else {
      overlap = host_ranges_overlap(samples, in_b, amp_out, out_b) || host_ranges_overlap(samples, in_b, phase_out, out_b);
    }
    if (workers >= 2 && per_chunk >= 1 && batch >= 2 * per_chunk && len > 0 &&
        (size_t)batch * (size_t)(n + 2 * bins) * sizeof(T) >= kChunkedMinBytes && !overlap) {
      const T *d_window = nullptr;
      if (n != 1 && window != PDSP_WIN_RECT) {
        if (int rc = plan_window<T>(plan, window, &d_window)) return rc;  // built once, before the workers read it
      }
      // slot (elements): [per_chunk frames of n][per_chunk rows of amp][per_chunk rows of phase], 16-byte aligned parts
      const size_t amp_off = ((size_t)per_chunk * (size_t)n + 3) & ~(size_t)3;
      const size_t ph_off = (amp_off + (size_t)per_chunk * (size_t)bins + 3) & ~(size_t)3;
      const size_t slot = (ph_off + (size_t)per_chunk * (size_t)bins + 3) & ~(size_t)3;
      const long long nchunks = (batch + per_chunk - 1) / per_chunk;
      const int k = (long long)workers < nchunks ? workers : (int)nchunks;
      if (int rc = ensure_stage(plan, (size_t)k * slot * sizeof(T))) return rc;
      return run_chunked(plan, batch, per_chunk, k, [&](const ChunkJob &job) -> int {
        T *h = (T *)plan->h_stage + (size_t)job.slot * slot, *d = (T *)plan->d_stage + (size_t)job.slot * slot;
        for (long long b = 0; b < job.count; ++b) {
          T *dst = h + (size_t)b * (size_t)n;
          rows_to_stage<T>(dst, frame(job.first + b), (size_t)used);
          if (used < n) std::memset(dst + used, 0, (size_t)(n - used) * sizeof(T));
        }
        const size_t rows = (size_t)job.count * (size_t)bins;
        PDSP_HIP_TRY(hipMemcpyAsync(d, h, (size_t)job.count * frame_bytes, hipMemcpyHostToDevice, job.stream));
        if (int rc = spectrum_impl<T>(plan, job.count, d, n, n, d_window, sides, d + amp_off, d + ph_off, nullptr, nullptr,
                                      1.0, job.stream))
          return rc;
        PDSP_HIP_TRY(hipMemcpyAsync(h + amp_off, d + amp_off, rows * sizeof(T), hipMemcpyDeviceToHost, job.stream));
        PDSP_HIP_TRY(hipMemcpyAsync(h + ph_off, d + ph_off, rows * sizeof(T), hipMemcpyDeviceToHost, job.stream));
        PDSP_HIP_TRY(hipStreamSynchronize(job.stream));
        stage_to_rows<T>(amp_out + (size_t)job.first * (size_t)bins, h + amp_off, rows);
        stage_to_rows<T>(phase_out + (size_t)job.first * (size_t)bins, h + ph_off, rows);
        if (peak_out) host_peaks(freq, amp_out, phase_out, bins, job.first, job.count, peak_out);
        return PDSP_OK;
      });
    }
  }
  // staging: [batch frames of n][batch rows of amp][batch rows of phase]; rows start 16-byte aligned
  const size_t rows = (size_t)batch * (size_t)bins, frames_sz = (size_t)batch * (size_t)n;
  const size_t amp_off = (frames_sz + 3) & ~(size_t)3, ph_off = (amp_off + rows + 3) & ~(size_t)3;
  const size_t total = ph_off + rows;
  if (int rc = ensure_stage(plan, total * sizeof(T))) return rc;
  T *h = (T *)plan->h_stage, *d = (T *)plan->d_stage;
  for (long long b = 0; b < batch; ++b) {
    const double *src = frame(b);
    T *dst = h + (size_t)b * (size_t)n;
    for (long long i = 0; i < used; ++i) dst[i] = (T)src[i];
    for (long long i = used; i < n; ++i) dst[i] = T(0);
  }
  const T *d_window = nullptr;
  if (n != 1 && window != PDSP_WIN_RECT) {
    if (int rc = plan_window<T>(plan, window, &d_window)) return rc;
  }
  hipStream_t s = plan->stream;
  T *const z = zero_copy(total * sizeof(T)) ? stage_device_view<T>(plan) : nullptr;
  if (z) d = z;  // the kernel works on the pinned buffer itself
  else PDSP_HIP_TRY(hipMemcpyAsync(d, h, frames_sz * sizeof(T), hipMemcpyHostToDevice, s));
  if (int rc = spectrum_impl<T>(plan, batch, d, n, n, d_window, sides, d + amp_off, d + ph_off, nullptr, nullptr, 1.0, s))
    return rc;
  if (!z) PDSP_HIP_TRY(hipMemcpyAsync(h + amp_off, d + amp_off, (total - amp_off) * sizeof(T), hipMemcpyDeviceToHost, s));
  PDSP_HIP_TRY(hipStreamSynchronize(s));
  for (size_t i = 0; i < rows; ++i) amp_out[i] = (double)h[amp_off + i];
  for (size_t i = 0; i < rows; ++i) phase_out[i] = (double)h[ph_off + i];
  if (peak_out) host_peaks(freq, amp_out, phase_out, bins, 0, batch, peak_out);
  return PDSP_OK;
}

int spectrum_frames_host(const double *samples, const double *const *rows, long long batch, long long len,
                         double sample_rate, long long fft_size, int window, int sides, double *freq_out,
                         double *amp_out, double *phase_out, pdsp_peak *peak_out, long long *bins_out);

}  // namespace

extern "C" {

int pdsp_version(void) { return 100; }

const char *pdsp_last_error(void) { return g_err.c_str(); }

int pdsp_device_count(void) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return count;
}

int pdsp_max_size(int scalar_bytes) {  // incl. the four-step paths
  if (scalar_bytes == 4) return 1 << pdsp::kMaxLog2Big_f32;
  if (scalar_bytes == 8) return 1 << pdsp::kMaxLog2Big_f64;
  return 0;
}

int pdsp_set_split16k(int enabled) {
  const int prev = g_split16k;
  g_split16k = enabled ? 1 : 0;
  g_split8k_f32 = (enabled & 2) ? 1 : 0;
  return prev;
}

int pdsp_set_twopass(int enabled) {
  const int prev = g_twopass;
  g_twopass = enabled & 7;  // bit 0: tile passes; bit 1: their first form; bit 2 (or any value but 1): no fft_paired_kernel
  return prev;
}

int pdsp_set_fused_window(int enabled) {
  const int prev = g_fused_window;
  g_fused_window = enabled ? 1 : 0;
  return prev;
}

#define PDSP_DEFINE_PLAN_WINDOW(SUFFIX, T)                                                                  \
  int pdsp_plan_window_##SUFFIX(pdsp_plan *plan, int type, const T **window_out) {                          \
    if (!plan || !window_out) return fail(PDSP_ERR_BAD_ARG, "null plan or output");                        \
    *window_out = nullptr;                                                                                 \
    if (type < PDSP_WIN_RECT || type > PDSP_WIN_BLACKMAN)                                                  \
      return fail(PDSP_ERR_WINDOW_TYPE, "Unsupported window type: %d", type);                              \
    std::lock_guard<std::mutex> lk(plan->mu);                                                              \
    DeviceGuard g(plan->device);                                                                           \
    PDSP_HIP_TRY(g.err);                                                                                   \
    return plan_window<T>(plan, type, window_out);                                                         \
  }
PDSP_DEFINE_PLAN_WINDOW(f32, float)
PDSP_DEFINE_PLAN_WINDOW(f64, double)
#undef PDSP_DEFINE_PLAN_WINDOW

int pdsp_set_staged_small(int enabled) {
  const int prev = g_staged_small;
  g_staged_small = enabled ? 1 : 0;
  return prev;
}

int pdsp_set_real_packed(int enabled) {
  const int prev = g_real_packed;
  g_real_packed = enabled ? 1 : 0;
  return prev;
}

int pdsp_set_host_precision(int bits) {
  const int prev = host_precision();
  if (bits == 32 || bits == 64) g_host_precision = bits;
  return prev;
}

/* ---- host index math ---------------------------------------------------- */

int pdsp_is_pow2(long long n) { return n > 0 && (n & (n - 1)) == 0; }

long long pdsp_next_pow2(long long n) {
  if (n <= 1) return 1;
  long long p = 1;
  while (p < n && p < (1LL << 62)) p <<= 1;
  return p;
}

int pdsp_window_make(int type, long long size, double *out) {
  if (size <= 0) return fail(PDSP_ERR_WINDOW_SIZE, "Window size must be positive, got %lld", size);
  if (!out) return fail(PDSP_ERR_BAD_ARG, "out is null");
  if (size == 1) {  // fourier.ts:18-20 returns [1] before looking at the type
    out[0] = 1.0;
    return PDSP_OK;
  }
  if (type < PDSP_WIN_RECT || type > PDSP_WIN_BLACKMAN)
    return fail(PDSP_ERR_WINDOW_TYPE, "Unsupported window type: %d", type);
  const double denom = (double)(size - 1);
  for (long long i = 0; i < size; ++i) {
    const double f = (2.0 * M_PI * (double)i) / denom;
    double v = 1.0;
    if (type == PDSP_WIN_HANN) v = 0.5 * (1.0 - std::cos(f));
    else if (type == PDSP_WIN_HAMMING) v = 0.54 - 0.46 * std::cos(f);
    else if (type == PDSP_WIN_BLACKMAN) v = 0.42 - 0.5 * std::cos(f) + 0.08 * std::cos(2.0 * f);
    out[i] = v;
  }
  return PDSP_OK;
}

int pdsp_bin_frequencies(long long size, double sample_rate, int sides, double *out, long long *bins_out) {
  if (size <= 0) return fail(PDSP_ERR_FFT_SIZE, "FFT size must be positive, got %lld", size);
  if (sample_rate <= 0) return fail(PDSP_ERR_SAMPLE_RATE, "Sample rate must be positive, got %.17g", sample_rate);
  const long long bins = sides == PDSP_SIDES_ONE ? size / 2 + 1 : size;
  if (bins_out) *bins_out = bins;
  if (out) {
    const double scale = sample_rate / (double)size;
    for (long long i = 0; i < bins; ++i) out[i] = (double)i * scale;
  }
  return PDSP_OK;
}

int pdsp_fft_shift_f64(const double *in, long long n, double *out) {
  if (n < 0 || (n > 0 && (!in || !out))) return fail(PDSP_ERR_BAD_ARG, "bad fftShift arguments");
  const long long mid = n / 2;
  for (long long i = 0; i < n; ++i) out[i] = in[(i + mid) % n];
  return PDSP_OK;
}

long long pdsp_find_peak_f64(const double *amp, long long bins) {
  if (!amp || bins <= 0) return 0;
  long long max_i = 0, nondc_i = 0;
  double max_v = amp[0], nondc_v = 0.0;
  bool has_nondc = false;
  for (long long i = 1; i < bins; ++i) {
    const double v = amp[i];
    if (v > nondc_v) {
      nondc_v = v;
      nondc_i = i;
    }
    if (v > 0) has_nondc = true;
    if (v > max_v) {
      max_v = v;
      max_i = i;
    }
  }
  return has_nondc ? nondc_i : max_i;
}

/* ---- plan ----------------------------------------------------------------- */

int pdsp_plan_create(long long size, int device, pdsp_plan **plan_out) {
  if (!plan_out) return fail(PDSP_ERR_BAD_ARG, "plan_out is null");
  *plan_out = nullptr;
  if (!pdsp_is_pow2(size)) return fail(PDSP_ERR_SIZE_NOT_POW2, "FFT size must be power of two, got %lld", size);
  const int log2n = ilog2ll(size);
  if (log2n > pdsp::kMaxLog2Big_f32)
    return fail(PDSP_ERR_UNSUPPORTED_SIZE, "FFT size %lld exceeds the supported maximum %d", size,
                1 << pdsp::kMaxLog2Big_f32);
  if (int rc = require_device()) return rc;
  int count = 0;
  PDSP_HIP_TRY(hipGetDeviceCount(&count));
  if (device < 0) PDSP_HIP_TRY(hipGetDevice(&device));
  if (device >= count) return fail(PDSP_ERR_BAD_ARG, "device %d out of range (%d visible)", device, count);
  DeviceGuard g(device);
  PDSP_HIP_TRY(g.err);
  pdsp_plan *p = new (std::nothrow) pdsp_plan();
  if (!p) return fail(PDSP_ERR_BAD_ARG, "out of host memory");
  p->n = size;
  p->log2n = log2n;
  p->device = device;
  // f32: single-pass up to 2^14, four-step with fused columns (N1 <= 16) up to 2^18, general
  // four-step (N1 x 2^14) up to 2^28; the packed-real spectrum tables exist for the single-pass sizes
  hipError_t e = upload_tables<float>(p->t32, log2n, size, true, log2n >= 6 && log2n <= pdsp::kMaxLog2N_f32);
  // f64: the complex transform single-pass up to 2^13 and four-step up to 2^26; the packed-real
  // spectrum (an N/2-point transform) up to N = 2^14
  if (e == hipSuccess)
    e = upload_tables<double>(p->t64, log2n, size, log2n <= pdsp::kMaxLog2Big_f64,
                              log2n >= 6 && log2n - 1 <= pdsp::kMaxLog2N_f64);
  if (e != hipSuccess) {
    p->t32.release();
    p->t64.release();
    delete p;
    return fail(PDSP_ERR_DEVICE, "HIP error %d (%s) while uploading the twiddle tables", (int)e, hipGetErrorString(e));
  }
  *plan_out = p;
  return PDSP_OK;
}

int pdsp_plan_cache_clear(void) {
  PlanCache &c = plan_cache();
  std::lock_guard<std::mutex> lk(c.mu);
  // entries a call is still running on stay (their pin is released by that call)
  for (auto it = c.plans.begin(); it != c.plans.end();) {
    if (it->second.pins == 0) {
      pdsp_plan_destroy(it->second.plan);
      it = c.plans.erase(it);
    } else {
      ++it;
    }
  }
  trim_scratch_pools();  // the multi-pass paths' scratch planes (StreamScratch) go back to the device
  return PDSP_OK;
}

int pdsp_plan_destroy(pdsp_plan *plan) {
  if (!plan) return PDSP_OK;
  {
    DeviceGuard g(plan->device);
    if (plan->stream) {
      (void)hipStreamSynchronize(plan->stream);
      (void)hipStreamDestroy(plan->stream);
    }
    for (hipStream_t st : plan->slot_streams) {
      (void)hipStreamSynchronize(st);
      (void)hipStreamDestroy(st);
    }
    // a plan of a multi-pass size that drew scratch planes: the freed planes go back to the device with it
    const bool multipass = plan->t32.log2n1 > 0 || plan->t64.log2n1 > 0;
    plan->t32.release();
    plan->t64.release();
    if (plan->d_stage) (void)hipFree(plan->d_stage);
    if (plan->h_stage) (void)hipHostFree(plan->h_stage);
    if (multipass && g_scratch_drawn.load() > 0) {
      (void)hipDeviceSynchronize();  // stream-ordered frees complete before the pool can let go of them
      trim_scratch_pools();
    }
  }
  delete plan;
  return PDSP_OK;
}

long long pdsp_plan_size(const pdsp_plan *plan) { return plan ? plan->n : 0; }
int pdsp_plan_device(const pdsp_plan *plan) { return plan ? plan->device : -1; }

/* ---- device-pointer transforms --------------------------------------------- */

#define PDSP_DEFINE_TRANSFORMS(SUFFIX, T)                                                                          \
  int pdsp_fft_forward_real_##SUFFIX(const pdsp_plan *plan, long long batch, const T *re_in, T *re_out, T *im_out, \
                                     pdsp_stream stream) {                                                         \
    return run_complex<T>(plan, batch, re_in, nullptr, re_out, im_out, T(1), (hipStream_t)stream);                 \
  }                                                                                                                \
  int pdsp_fft_forward_complex_##SUFFIX(const pdsp_plan *plan, long long batch, const T *re_in, const T *im_in,    \
                                        T *re_out, T *im_out, pdsp_stream stream) {                                \
    if (batch > 0 && !im_in) return fail(PDSP_ERR_BAD_ARG, "null buffer");                                         \
    return run_complex<T>(plan, batch, re_in, im_in, re_out, im_out, T(1), (hipStream_t)stream);                   \
  }                                                                                                                \
  /* conj(FFT(conj(z))) == swap(FFT(swap(z))): the conjugated-twiddle sweep of fft.ts:122 is the forward */        \
  /* kernel with the planes exchanged on the way in and out; the 1/N of fft.ts:142-148 rides on the store */       \
  int pdsp_fft_inverse_##SUFFIX(const pdsp_plan *plan, long long batch, const T *re_in, const T *im_in, T *re_out, \
                                T *im_out, pdsp_stream stream) {                                                   \
    if (!plan) return fail(PDSP_ERR_BAD_ARG, "plan is null");                                                      \
    if (batch > 0 && !re_in) return fail(PDSP_ERR_BAD_ARG, "null buffer");                                         \
    return run_complex<T>(plan, batch, im_in, re_in, im_out, re_out, T(1) / (T)plan->n, (hipStream_t)stream);      \
  }                                                                                                                \
  int pdsp_fft_forward_interleaved_##SUFFIX(const pdsp_plan *plan, long long batch, const T *in, T *out,           \
                                            pdsp_stream stream) {                                                  \
    return run_interleaved<T>(plan, batch, in, out, false, (hipStream_t)stream);                                   \
  }                                                                                                                \
  int pdsp_fft_inverse_interleaved_##SUFFIX(const pdsp_plan *plan, long long batch, const T *in, T *out,           \
                                            pdsp_stream stream) {                                                  \
    return run_interleaved<T>(plan, batch, in, out, true, (hipStream_t)stream);                                    \
  }                                                                                                                \
  int pdsp_apply_window_##SUFFIX(long long batch, long long n, const T *in, const T *window, T *out,               \
                                 pdsp_stream stream) {                                                             \
    if (batch < 0 || n < 0) return fail(PDSP_ERR_BAD_ARG, "negative size");                                        \
    return apply_window_dev<T>(batch, n, in, window, out, (hipStream_t)stream);                                    \
  }                                                                                                                \
  int pdsp_magnitude_##SUFFIX(long long count, const T *re, const T *im, T *out, pdsp_stream stream) {             \
    return polar_dev<T, false>(count, re, im, out, (hipStream_t)stream);                                           \
  }                                                                                                                \
  int pdsp_phase_##SUFFIX(long long count, const T *re, const T *im, T *out, pdsp_stream stream) {                 \
    return polar_dev<T, true>(count, re, im, out, (hipStream_t)stream);                                            \
  }                                                                                                                \
  int pdsp_spectrum_##SUFFIX(const pdsp_plan *plan, long long batch, const T *frames, long long frame_len,         \
                             long long frame_stride, const T *window, int sides, T *amp_out, T *phase_out,         \
                             int32_t *peak_out, pdsp_stream stream) {                                              \
    if (batch > 0 && !amp_out) return fail(PDSP_ERR_BAD_ARG, "null buffer");                                       \
    return spectrum_impl<T>(plan, batch, frames, frame_len, frame_stride, window, sides, amp_out, phase_out,       \
                            peak_out, nullptr, 1.0, (hipStream_t)stream);                                          \
  }

PDSP_DEFINE_TRANSFORMS(f32, float)
PDSP_DEFINE_TRANSFORMS(f64, double)
#undef PDSP_DEFINE_TRANSFORMS

int pdsp_complex_op_f32(int op, long long count, const float *a_re, const float *a_im, const float *b_re,
                        const float *b_im, long long b_len, double s_re, double s_im, float *out_re, float *out_im,
                        pdsp_stream stream) {
  if (count < 0) return fail(PDSP_ERR_BAD_ARG, "negative size");
  if (op < PDSP_CX_ADD || op > PDSP_CX_MUL_SCALAR) return fail(PDSP_ERR_BAD_ARG, "unknown complex op %d", op);
  if (count == 0) return PDSP_OK;
  if (!a_re || !a_im || !out_re || !out_im) return fail(PDSP_ERR_BAD_ARG, "null buffer");
  const bool binary = op <= PDSP_CX_DIV;
  if (binary) {
    if (!b_re || !b_im) return fail(PDSP_ERR_BAD_ARG, "null buffer");
    if (b_len <= 0 || count % b_len != 0)
      return fail(PDSP_ERR_BAD_ARG, "second operand length %lld must divide %lld", b_len, count);
  }
  hipStream_t s = (hipStream_t)stream;
  const float sr = (float)s_re, si = (float)s_im;
  switch (op) {
    case PDSP_CX_ADD: return launch_complex_op<pdsp::kAdd>(count, a_re, a_im, b_re, b_im, b_len, sr, si, out_re, out_im, s);
    case PDSP_CX_SUB: return launch_complex_op<pdsp::kSub>(count, a_re, a_im, b_re, b_im, b_len, sr, si, out_re, out_im, s);
    case PDSP_CX_MUL: return launch_complex_op<pdsp::kMul>(count, a_re, a_im, b_re, b_im, b_len, sr, si, out_re, out_im, s);
    case PDSP_CX_DIV: return launch_complex_op<pdsp::kDiv>(count, a_re, a_im, b_re, b_im, b_len, sr, si, out_re, out_im, s);
    case PDSP_CX_CONJ: return launch_complex_op<pdsp::kConj>(count, a_re, a_im, b_re, b_im, b_len, sr, si, out_re, out_im, s);
    case PDSP_CX_SCALE: return launch_complex_op<pdsp::kScale>(count, a_re, a_im, b_re, b_im, b_len, sr, si, out_re, out_im, s);
    default: return launch_complex_op<pdsp::kMulScalar>(count, a_re, a_im, b_re, b_im, b_len, sr, si, out_re, out_im, s);
  }
}

/* ---- fused spectrum: peaks ---------------------------------------------------- */

int pdsp_spectrum_peaks_f32(const pdsp_plan *plan, long long batch, const float *frames, long long frame_len,
                            long long frame_stride, const float *window, int sides, double sample_rate,
                            float *amp_out, float *phase_out, pdsp_peak32 *peaks_out, pdsp_stream stream) {
  if (batch > 0 && !peaks_out) return fail(PDSP_ERR_BAD_ARG, "null buffer");
  return spectrum_impl<float>(plan, batch, frames, frame_len, frame_stride, window, sides, amp_out, phase_out, nullptr,
                              peaks_out, sample_rate, (hipStream_t)stream);
}

/* ---- host f64 drop-in entry points ------------------------------------------ */

int pdsp_fft_transform_host_f64(pdsp_plan *plan, long long batch, long long in_len, const double *re_in,
                                const double *im_in, double *re_out, double *im_out, int inverse) {
  if (int rc = check_plan_batch(plan, batch)) return rc;
  if (in_len != plan->n) return fail(PDSP_ERR_INPUT_LENGTH, "FFT input length %lld != size %lld", in_len, plan->n);
  if (batch == 0) return PDSP_OK;
  if (!re_in || !re_out || !im_out) return fail(PDSP_ERR_BAD_ARG, "null buffer");
  if (inverse && !im_in) return fail(PDSP_ERR_BAD_ARG, "inverse needs an imaginary plane");
  std::lock_guard<std::mutex> lk(plan->mu);
  DeviceGuard g(plan->device);
  PDSP_HIP_TRY(g.err);
  const int rc = (host_precision() == 64 && plan->t64.tw)
                     ? transform_host<double>(plan, batch, re_in, im_in, nullptr, nullptr, re_out, im_out, inverse)
                     : transform_host<float>(plan, batch, re_in, im_in, nullptr, nullptr, re_out, im_out, inverse);
  trim_stage(plan);
  return rc;
}

int pdsp_fft_transform_rows_host_f64(pdsp_plan *plan, long long batch, long long in_len, const double *const *re_rows,
                                     const double *const *im_rows, double *re_out, double *im_out, int inverse) {
  if (int rc = check_plan_batch(plan, batch)) return rc;
  if (in_len != plan->n) return fail(PDSP_ERR_INPUT_LENGTH, "FFT input length %lld != size %lld", in_len, plan->n);
  if (batch == 0) return PDSP_OK;
  if (!re_rows || !re_out || !im_out) return fail(PDSP_ERR_BAD_ARG, "null buffer");
  if (inverse && !im_rows) return fail(PDSP_ERR_BAD_ARG, "inverse needs an imaginary plane");
  for (long long r = 0; r < batch; ++r)
    if (!re_rows[r] || (im_rows && !im_rows[r])) return fail(PDSP_ERR_BAD_ARG, "null buffer");
  std::lock_guard<std::mutex> lk(plan->mu);
  DeviceGuard g(plan->device);
  PDSP_HIP_TRY(g.err);
  const int rc = (host_precision() == 64 && plan->t64.tw)
                     ? transform_host<double>(plan, batch, nullptr, nullptr, re_rows, im_rows, re_out, im_out, inverse)
                     : transform_host<float>(plan, batch, nullptr, nullptr, re_rows, im_rows, re_out, im_out, inverse);
  trim_stage(plan);
  return rc;
}

int pdsp_apply_window_host_f64(const double *in, long long in_len, const double *window, long long window_len,
                               double *out) {
  if (in_len != window_len) return fail(PDSP_ERR_WINDOW_LENGTH, "Window length must match input length.");
  if (in_len == 0) return PDSP_OK;
  if (in_len < 0 || !in || !window || !out) return fail(PDSP_ERR_BAD_ARG, "bad applyWindow arguments");
  if (int rc = require_device()) return rc;
  return host_precision() == 64 ? apply_window_host<double>(in, in_len, window, out)
                                : apply_window_host<float>(in, in_len, window, out);
}

static int polar_host(const double *re, const double *im, long long n, double *out, bool want_phase) {
  if (n == 0) return PDSP_OK;
  if (n < 0 || !re || !im || !out) return fail(PDSP_ERR_BAD_ARG, "bad magnitude/phase arguments");
  if (int rc = require_device()) return rc;
  return host_precision() == 64 ? polar_host_t<double>(re, im, n, out, want_phase)
                                : polar_host_t<float>(re, im, n, out, want_phase);
}

int pdsp_magnitude_host_f64(const double *re, const double *im, long long n, double *out) {
  return polar_host(re, im, n, out, false);
}

int pdsp_phase_host_f64(const double *re, const double *im, long long n, double *out) {
  return polar_host(re, im, n, out, true);
}

int pdsp_spectrum_host_f64(const double *samples, long long len, double sample_rate, long long fft_size, int window,
                           int sides, double *freq_out, double *amp_out, double *phase_out, pdsp_peak *peak_out,
                           long long *bins_out) {
  return pdsp_spectrum_batch_host_f64(samples, 1, len, sample_rate, fft_size, window, sides, freq_out, amp_out, phase_out,
                                      peak_out, bins_out);
}

int pdsp_spectrum_batch_host_f64(const double *samples, long long batch, long long len, double sample_rate,
                                 long long fft_size, int window, int sides, double *freq_out, double *amp_out,
                                 double *phase_out, pdsp_peak *peak_out, long long *bins_out) {
  return spectrum_frames_host(samples, nullptr, batch, len, sample_rate, fft_size, window, sides, freq_out, amp_out,
                              phase_out, peak_out, bins_out);
}

int pdsp_spectrum_rows_host_f64(const double *const *rows, long long batch, long long len, double sample_rate,
                                long long fft_size, int window, int sides, double *freq_out, double *amp_out,
                                double *phase_out, pdsp_peak *peak_out, long long *bins_out) {
  if (batch > 0 && len > 0) {
    if (!rows) return fail(PDSP_ERR_BAD_ARG, "bad samples");
    for (long long b = 0; b < batch; ++b)
      if (!rows[b]) return fail(PDSP_ERR_BAD_ARG, "bad samples");
  }
  return spectrum_frames_host(nullptr, len > 0 ? rows : nullptr, batch, len, sample_rate, fft_size, window, sides,
                              freq_out, amp_out, phase_out, peak_out, bins_out);
}

}  // extern "C"

namespace {

int spectrum_frames_host(const double *samples, const double *const *rows, long long batch, long long len,
                         double sample_rate, long long fft_size, int window, int sides, double *freq_out,
                         double *amp_out, double *phase_out, pdsp_peak *peak_out, long long *bins_out) {
  if (batch < 0) return fail(PDSP_ERR_BAD_ARG, "batch must be >= 0, got %lld", batch);
  if (len < 0 || (len > 0 && batch > 0 && !samples && !rows)) return fail(PDSP_ERR_BAD_ARG, "bad samples");
  if (sides != PDSP_SIDES_ONE && sides != PDSP_SIDES_TWO) return fail(PDSP_ERR_BAD_ARG, "bad sides %d", sides);
  // Error order of spectrum.ts:113-132: FFT ctor (power of two) -> createWindow
  // (type; N == 1 returns before the type switch) -> ... -> binFrequencies (rate).
  const long long n = fft_size >= 0 ? fft_size : pdsp_next_pow2(len);  // < 0: options.fftSize absent
  if (!pdsp_is_pow2(n)) return fail(PDSP_ERR_SIZE_NOT_POW2, "FFT size must be power of two, got %lld", n);
  if (n != 1 && (window < PDSP_WIN_RECT || window > PDSP_WIN_BLACKMAN))
    return fail(PDSP_ERR_WINDOW_TYPE, "Unsupported window type: %d", window);
  if (sample_rate <= 0) return fail(PDSP_ERR_SAMPLE_RATE, "Sample rate must be positive, got %.17g", sample_rate);
  if (!freq_out || !amp_out || !phase_out) return fail(PDSP_ERR_BAD_ARG, "null output");
  if (int rc = require_device()) return rc;
  CachedPlan pin;
  if (int rc = cached_plan(n, &pin)) return rc;  // plan + window are cached per (size, device), LRU-bounded
  pdsp_plan *const plan = pin.plan;
  std::lock_guard<std::mutex> lk(plan->mu);
  DeviceGuard g(plan->device);
  PDSP_HIP_TRY(g.err);
  const long long bins = sides == PDSP_SIDES_ONE ? n / 2 + 1 : n;
  if (int rc = pdsp_bin_frequencies(n, sample_rate, sides, freq_out, nullptr)) return rc;
  if (bins_out) *bins_out = bins;
  if (batch == 0) return PDSP_OK;
  const bool f64 = host_precision() == 64 && (plan->t64.tw_half || plan->t64.tw);
  static const double kNoSample = 0.0;  // len == 0: every frame is all zero padding
  const double *src = len > 0 ? samples : &kNoSample;
  // findPeak runs on the host over the f64 amplitudes (host_peaks), inside the call's staging loop
  const int rc_run =
      f64 ? spectrum_host_t<double>(plan, src, rows, len, window, sides, amp_out, phase_out, batch, freq_out, peak_out)
          : spectrum_host_t<float>(plan, src, rows, len, window, sides, amp_out, phase_out, batch, freq_out, peak_out);
  trim_stage(plan);
  return rc_run;
}

}  // namespace

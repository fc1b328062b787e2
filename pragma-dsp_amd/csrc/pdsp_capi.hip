// pdsp_capi.hip -- the C ABI of include/pdsp_hip.h: argument validation with the
// reference's error texts, plan objects, kernel dispatch by size, and the
// synchronous host-f64 entry points the JS drop-in binds.
//
// Host side only: no kernel is instantiated in this translation unit (the dispatchers it calls -- run_complex,
// spectrum_impl, ... -- are declared in pdsp_internal.h and live in the pdsp_kernels_*.hip units).
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

#include "pdsp_internal.h"

namespace pdsp_host {

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
std::atomic<unsigned long long> g_scratch_drawn{0};

void trim_scratch_pools() {
  std::lock_guard<std::mutex> lk(g_scratch_pool_mu);
  for (hipMemPool_t pool : g_scratch_pool)
    if (pool) (void)hipMemPoolTrimTo(pool, 0);
  g_scratch_drawn = 0;
}

}  // namespace pdsp_host

using namespace pdsp_host;

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

// ---- chunked host calls -------------------------------------------------------------------------------------
// A batched host-f64 call moves every sample through the CPU twice (the caller's f64 rows <-> pinned staging), over
// PCIe twice, and through a kernel that needs a few percent of that time.  As ONE stage -> copy -> launch -> copy ->
// unstage sequence each step waits for the one before it and one core does all the staging: ~0.4 GSample/s at
// N = 4096 whatever the card does.  Calls above kChunkedMinBytes of staging are cut into chunks of ~kChunkInBytes of
// input rows instead; K workers (the calling thread and K - 1 helpers that live for the call) each own one staging
// slot and one stream and draw chunks from a shared counter: fill the slot, H2D, the same kernels the one-shot path
// launches, D2H, wait for the stream, unstage (+ findPeak).  Inside a worker the steps stay in order; across workers
// staging, both PCIe directions and the kernels overlap.  Row b of the result is the one-shot result of row b bit
// for bit (the kernels work row by row and the variant does not depend on the row count).  A worker keeps one
// chunk on the card while it fills the next (two slots and streams per worker).
// PDSP_HOST_THREADS sets K (1 = the one-shot sequence; default: half the cores this process may run on, 2 ... 6).
constexpr size_t kChunkInBytes = (size_t)2 << 20;
constexpr size_t kChunkedMinBytes = (size_t)4 << 20;
// Input bytes per chunk for a call that stages `in_total` bytes of input on `workers` workers: 2 MiB, less for small
// calls so that every worker still gets two chunks, not below 256 KiB.  (N = 1024 frames, us per call, one-shot /
// fixed 2-MiB chunks / these: 256 frames 327 / - / 304, 512 frames 597 / 445 / 406, 1,024 frames 1,331 / 652 / 536,
// 4,096 frames 14,663 / 1,405 / 1,527; 128 frames stay one-shot: 188 against 201.)
size_t chunk_in_bytes(size_t in_total, int workers) {
  size_t c = in_total / (size_t)(2 * (workers > 0 ? workers : 1));
  if (c > kChunkInBytes) c = kChunkInBytes;
  if (c < ((size_t)256 << 10)) c = (size_t)256 << 10;
  return c;
}

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

// submit(job) fills the job's slot and enqueues its copies and kernels on the job's stream without waiting;
// finish(job) is called after that stream has drained and unstages the results.  Both return a pdsp status (error
// text in the running thread's g_err).  Each worker owns TWO slots and streams and keeps one chunk on the card while
// it fills the next (slot = 2 * worker + parity), so the caller has staged 2 * `workers` slots (ensure_stage) and
// holds plan->mu.  The first failure stops the hand-out of chunks and is what the call returns.
constexpr int kSlotsPerWorker = 2;
template <class Submit, class Finish>
int run_chunked(pdsp_plan *plan, long long rows, long long rows_per_chunk, int workers, Submit submit, Finish finish) {
  const long long nchunks = (rows + rows_per_chunk - 1) / rows_per_chunk;
  if (workers > nchunks) workers = (int)nchunks;
  while ((long long)plan->slot_streams.size() < (long long)kSlotsPerWorker * workers) {
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
    const hipError_t e = hipSetDevice(plan->device);  // helpers start on device 0
    if (e != hipSuccess) {
      report(fail(PDSP_ERR_DEVICE, "HIP error %d (%s) at hipSetDevice", (int)e, hipGetErrorString(e)));
      return;
    }
    ChunkJob in_flight;  // the chunk this worker has on the card (count == 0: none)
    auto complete = [&]() -> int {  // wait for the chunk in flight and unstage it
      if (in_flight.count == 0) return PDSP_OK;
      const hipError_t se = hipStreamSynchronize(in_flight.stream);
      int rc = PDSP_OK;
      if (se != hipSuccess) rc = fail(PDSP_ERR_DEVICE, "HIP error %d (%s) at hipStreamSynchronize", (int)se, hipGetErrorString(se));
      else rc = finish(in_flight);
      in_flight.count = 0;
      return rc;
    };
    int parity = 0;
    while (status.load() == PDSP_OK) {
      const long long c = next.fetch_add(1);
      if (c >= nchunks) break;
      ChunkJob job;
      job.first = c * rows_per_chunk;
      job.count = rows - job.first < rows_per_chunk ? rows - job.first : rows_per_chunk;
      job.slot = kSlotsPerWorker * w + parity;
      job.stream = plan->slot_streams[(size_t)job.slot];
      if (const int rc = submit(job)) {  // fills the OTHER slot while in_flight's chunk is on the card
        report(rc);
        break;
      }
      if (const int rc = complete()) {
        report(rc);
        in_flight = job;  // drained below
        break;
      }
      in_flight = job;
      parity ^= 1;
    }
    if (status.load() == PDSP_OK) {
      if (const int rc = complete()) report(rc);
    }
    // nothing of this call stays in flight on the worker's slots, on any exit
    for (int q = 0; q < kSlotsPerWorker; ++q) (void)hipStreamSynchronize(plan->slot_streams[(size_t)(kSlotsPerWorker * w + q)]);
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

// f64 staging copies INTO the pinned slots (the caller's rows -> slot) with non-temporal 16-byte stores: the destination
// lines are not read again by this core, so the read-for-ownership a plain store pays is pure host-memory traffic --
// and host memory traffic (48 bytes per sample between the staging copies and the DMA engines) is what bounds the
// chunked calls.  Measured, 6 workers: transforms +11 ... +15 %, spectrum at N = 4096 +11 ... +21 %, at N = 16384 +-0.
// The closing sfence orders the stores before the copy command that reads the slot.  PDSP_NT_COPY=0: plain memcpy.
inline bool nt_copy() {
  static const bool on = [] { const char *e = getenv("PDSP_NT_COPY"); return !(e && atoi(e) == 0); }();
  return on;
}
// `nt`: the chunked paths only -- a one-frame call's few KiB are read by its caller next, and should stay in cache.
inline void copy_f64(double *dst, const double *src, size_t count, bool nt) {
#if defined(__x86_64__)
  typedef double v2d __attribute__((vector_size(16), aligned(16)));
  typedef double v2du __attribute__((vector_size(16), aligned(8)));
  if (nt && nt_copy() && count >= 64 && !((uintptr_t)dst & 7) && !((uintptr_t)src & 7)) {
    size_t i = 0;
    if ((uintptr_t)dst & 15) dst[i] = src[i], ++i;
    for (; i + 2 <= count; i += 2) __builtin_nontemporal_store(*(const v2du *)(src + i), (v2d *)(dst + i));
    for (; i < count; ++i) dst[i] = src[i];
    __builtin_ia32_sfence();
    return;
  }
#endif
  std::memcpy(dst, src, count * sizeof(double));
}
template <typename T>
inline void rows_to_stage(T *dst, const double *src, size_t count, bool nt = false) {
  if constexpr (sizeof(T) == sizeof(double)) copy_f64(dst, src, count, nt);
  else
    for (size_t i = 0; i < count; ++i) dst[i] = (T)src[i];
}
// (results go out with ordinary stores: a caller's fresh result arrays are in cache right after their first touch,
// where a non-temporal store is the slower one; measured with reused arrays: no difference either way)
template <typename T>
inline void stage_to_rows(double *dst, const T *src, size_t count) {
  if constexpr (sizeof(T) == sizeof(double)) copy_f64(dst, src, count, false);
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
  auto stage_in = [&](T *h_re, T *h_im, long long first, long long count, bool nt) {
    if (!re_rows) rows_to_stage<T>(h_re, re_in + (size_t)first * n, (size_t)count * n, nt);
    else
      for (long long r = 0; r < count; ++r) rows_to_stage<T>(h_re + (size_t)r * n, re_rows[first + r], n, nt);
    if (!has_im) return;
    if (!im_rows) rows_to_stage<T>(h_im, im_in + (size_t)first * n, (size_t)count * n, nt);
    else
      for (long long r = 0; r < count; ++r) rows_to_stage<T>(h_im + (size_t)r * n, im_rows[first + r], n, nt);
  };
  {
    // many rows: chunks on several workers (run_chunked); planes that overlap each other in host memory keep the
    // one-shot sequence, which has read every input before it writes any output
    const size_t row_bytes = n * sizeof(T), out_bytes = cnt * sizeof(double), in_row_bytes = n * sizeof(double);
    // (two planes' worth whether or not there is an imaginary input: the slot -- re | im | out re | out im -- stays
    // at 2 x kChunkInBytes, so 2 slots x 6 workers fit the staging a plan keeps between calls)
    const int workers = host_workers(tables<T>(plan).log2n1 > 0);
    const long long per_chunk = (long long)(chunk_in_bytes(2 * cnt * sizeof(T), workers) / (2 * row_bytes));
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
      if (int rc = ensure_stage(plan, (size_t)kSlotsPerWorker * k * slot * sizeof(T))) return rc;
      return run_chunked(
          plan, batch, per_chunk, k,
          [&](const ChunkJob &job) -> int {  // submit
            const size_t c = (size_t)job.count * n;
            T *h = (T *)plan->h_stage + (size_t)job.slot * slot, *d = (T *)plan->d_stage + (size_t)job.slot * slot;
            stage_in(h, h + c, job.first, job.count, true);
            PDSP_HIP_TRY(hipMemcpyAsync(d, h, (has_im ? 2 : 1) * c * sizeof(T), hipMemcpyHostToDevice, job.stream));
            int rc;
            if (inverse) rc = run_complex<T>(plan, job.count, d + c, d, d + 3 * c, d + 2 * c, T(1) / (T)plan->n, job.stream);
            else rc = run_complex<T>(plan, job.count, d, has_im ? d + c : nullptr, d + 2 * c, d + 3 * c, T(1), job.stream);
            if (rc) return rc;
            PDSP_HIP_TRY(hipMemcpyAsync(h + 2 * c, d + 2 * c, 2 * c * sizeof(T), hipMemcpyDeviceToHost, job.stream));
            return PDSP_OK;
          },
          [&](const ChunkJob &job) -> int {  // finish
            const size_t c = (size_t)job.count * n, off = (size_t)job.first * n;
            const T *h = (const T *)plan->h_stage + (size_t)job.slot * slot;
            stage_to_rows<T>(re_out + off, h + 2 * c, c);
            stage_to_rows<T>(im_out + off, h + 3 * c, c);
            return PDSP_OK;
          });
    }
  }
  if (int rc = ensure_stage(plan, 4 * cnt * sizeof(T))) return rc;
  T *h_re = (T *)plan->h_stage, *h_im = h_re + cnt, *h_ore = h_im + cnt, *h_oim = h_ore + cnt;
  T *d_re = (T *)plan->d_stage, *d_im = d_re + cnt, *d_ore = d_im + cnt, *d_oim = d_ore + cnt;
  stage_in(h_re, h_im, 0, batch, false);
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
// findPeak on the host over `count` rows of f64 amplitudes: exact strict-'>' and first-wins
// behaviour (spectrum.ts:74-105), peak.frequency from the call's one frequency axis.
inline void host_peaks(const double *freq, const double *amp_rows, const double *phase_rows, long long bins,
                       long long count, pdsp_peak *peak_out) {
  for (long long b = 0; b < count; ++b) {
    const double *a = amp_rows + (size_t)b * (size_t)bins, *p = phase_rows + (size_t)b * (size_t)bins;
    const long long pk = pdsp_find_peak_f64(a, bins);
    peak_out[b].index = (int32_t)pk;
    peak_out[b].frequency = freq[pk];
    peak_out[b].amplitude = a[pk];
    peak_out[b].phase = p[pk];
  }
}

// Where the frames of a host spectrum call lie: contiguous f64 (pdsp_spectrum_batch_host_f64), one pointer per frame
// in f64 (pdsp_spectrum_rows_host_f64) or in f32 (pdsp_spectrum_rows_host_f32in: Float32Array audio frames).
struct FrameSource {
  const double *samples = nullptr;
  const double *const *rows64 = nullptr;
  const float *const *rows32 = nullptr;
  long long len = 0;
  size_t frame_bytes() const { return (size_t)len * (rows32 ? sizeof(float) : sizeof(double)); }
  const void *frame(long long b) const {
    if (rows32) return rows32[b];
    return rows64 ? rows64[b] : samples + (size_t)b * (size_t)len;
  }
  // the first `used` samples of frame b into staging of precision T
  template <typename T>
  void stage(T *dst, long long b, size_t used, bool nt = false) const {
    if (rows32) {
      const float *src = rows32[b];
      if constexpr (sizeof(T) == sizeof(float)) std::memcpy(dst, src, used * sizeof(float));
      else
        for (size_t i = 0; i < used; ++i) dst[i] = (T)src[i];
    } else {
      rows_to_stage<T>(dst, (const double *)frame(b), used, nt);
    }
  }
  bool overlaps(const void *out, size_t out_bytes, long long batch) const {
    if (!rows64 && !rows32) return host_ranges_overlap(samples, (size_t)batch * frame_bytes(), out, out_bytes);
    for (long long b = 0; b < batch; ++b)
      if (host_ranges_overlap(frame(b), frame_bytes(), out, out_bytes)) return true;
    return false;
  }
};

template <typename T>
int spectrum_host_t(pdsp_plan *plan, const FrameSource &in, int window, int sides, double *amp_out, double *phase_out,
                    long long batch, const double *freq, pdsp_peak *peak_out) {
  const long long len = in.len;
  const long long n = plan->n;
  const long long bins = sides == PDSP_SIDES_ONE ? n / 2 + 1 : n;
  const long long used = len < n ? len : n;
  {
    // many frames: chunks on several workers (run_chunked)
    const size_t frame_bytes = (size_t)n * sizeof(T);
    const int workers = host_workers(tables<T>(plan).log2n1 > 0 && !tables<T>(plan).tw_half);  // packed-real frames: one pass
    const long long per_chunk = (long long)(chunk_in_bytes((size_t)batch * frame_bytes, workers) / frame_bytes);
    const size_t out_b = (size_t)batch * (size_t)bins * sizeof(double);
    const bool overlap = len > 0 && (in.overlaps(amp_out, out_b, batch) || in.overlaps(phase_out, out_b, batch));
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
      if (int rc = ensure_stage(plan, (size_t)kSlotsPerWorker * k * slot * sizeof(T))) return rc;
      return run_chunked(
          plan, batch, per_chunk, k,
          [&](const ChunkJob &job) -> int {  // submit
            T *h = (T *)plan->h_stage + (size_t)job.slot * slot, *d = (T *)plan->d_stage + (size_t)job.slot * slot;
            for (long long b = 0; b < job.count; ++b) {
              T *dst = h + (size_t)b * (size_t)n;
              in.stage<T>(dst, job.first + b, (size_t)used, true);
              if (used < n) std::memset(dst + used, 0, (size_t)(n - used) * sizeof(T));
            }
            const size_t rows = (size_t)job.count * (size_t)bins;
            PDSP_HIP_TRY(hipMemcpyAsync(d, h, (size_t)job.count * frame_bytes, hipMemcpyHostToDevice, job.stream));
            if (int rc = spectrum_impl<T>(plan, job.count, d, n, n, d_window, sides, d + amp_off, d + ph_off, nullptr,
                                          nullptr, 1.0, job.stream))
              return rc;
            PDSP_HIP_TRY(hipMemcpyAsync(h + amp_off, d + amp_off, rows * sizeof(T), hipMemcpyDeviceToHost, job.stream));
            PDSP_HIP_TRY(hipMemcpyAsync(h + ph_off, d + ph_off, rows * sizeof(T), hipMemcpyDeviceToHost, job.stream));
            return PDSP_OK;
          },
          [&](const ChunkJob &job) -> int {  // finish
            const T *h = (const T *)plan->h_stage + (size_t)job.slot * slot;
            const size_t rows = (size_t)job.count * (size_t)bins;
            // (host_peaks below reads the f64 rows from the slot, which is still in cache, not from the rows just streamed out)
            stage_to_rows<T>(amp_out + (size_t)job.first * (size_t)bins, h + amp_off, rows);
            stage_to_rows<T>(phase_out + (size_t)job.first * (size_t)bins, h + ph_off, rows);
            if (peak_out) {
              if constexpr (sizeof(T) == sizeof(double))
                host_peaks(freq, (const double *)(h + amp_off), (const double *)(h + ph_off), bins, job.count, peak_out + job.first);
              else
                host_peaks(freq, amp_out + (size_t)job.first * (size_t)bins, phase_out + (size_t)job.first * (size_t)bins, bins,
                           job.count, peak_out + job.first);
            }
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
    T *dst = h + (size_t)b * (size_t)n;
    if (used > 0) in.stage<T>(dst, b, (size_t)used);
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
  if (peak_out) host_peaks(freq, amp_out, phase_out, bins, batch, peak_out);
  return PDSP_OK;
}

int spectrum_frames_host(const FrameSource &in, long long batch, double sample_rate, long long fft_size, int window,
                         int sides, double *freq_out, double *amp_out, double *phase_out, pdsp_peak *peak_out,
                         long long *bins_out);

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

/* ---- plane layout ------------------------------------------------------------ */

struct pdsp_arena {
  int device = -1;
  void *parts[4] = {nullptr, nullptr, nullptr, nullptr};  // one entry (the arena) or up to four plain allocations
};

int pdsp_planes_alloc(const pdsp_plan *plan, long long batch, int scalar_bytes, int real_input, void **re_in,
                      void **im_in, void **re_out, void **im_out, pdsp_arena **arena, unsigned long long *arena_bytes) {
  if (!plan) return fail(PDSP_ERR_BAD_ARG, "plan is null");
  if (!re_in || !im_in || !re_out || !im_out || !arena) return fail(PDSP_ERR_BAD_ARG, "null output");
  if (batch <= 0) return fail(PDSP_ERR_BAD_ARG, "batch must be > 0, got %lld", batch);
  if (scalar_bytes != 4 && scalar_bytes != 8) return fail(PDSP_ERR_BAD_ARG, "scalar_bytes must be 4 or 8, got %d", scalar_bytes);
  *re_in = *im_in = *re_out = *im_out = nullptr;
  *arena = nullptr;
  if (arena_bytes) *arena_bytes = 0;
  DeviceGuard g(plan->device);
  PDSP_HIP_TRY(g.err);
  const size_t gib = (size_t)1 << 30;
  const size_t plane = (size_t)batch * (size_t)plan->n * (size_t)scalar_bytes;
  pdsp_arena *a = new (std::nothrow) pdsp_arena();
  if (!a) return fail(PDSP_ERR_BAD_ARG, "out of host memory");
  a->device = plan->device;
  if (plane >= ((size_t)256 << 20) && plane <= 8 * gib) {  // smaller planes have nothing to gain: plain allocations
    size_t free_b = 0, total_b = 0;
    const size_t need = 80 * gib + plane;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b >= need + 4 * gib) {
      void *base = nullptr;
      if (hipMalloc(&base, need) == hipSuccess) {
        a->parts[0] = base;
        char *b = (char *)base;
        *re_in = b;
        *im_in = real_input ? nullptr : b + ((plane + 255) & ~(size_t)255);
        *re_out = b + 40 * gib;
        *im_out = b + 80 * gib;
        *arena = a;
        if (arena_bytes) *arena_bytes = (unsigned long long)need;
        return PDSP_OK;
      }
      (void)hipGetLastError();
    }
  }
  // no room for the layout: four plain allocations
  void **outs[4] = {re_in, im_in, re_out, im_out};
  for (int i = 0; i < 4; ++i) {
    if (i == 1 && real_input) continue;
    const hipError_t e = hipMalloc(&a->parts[i], plane);
    if (e != hipSuccess) {
      for (void *p : a->parts)
        if (p) (void)hipFree(p);
      delete a;
      *re_in = *im_in = *re_out = *im_out = nullptr;
      return fail(PDSP_ERR_DEVICE, "HIP error %d (%s) at hipMalloc of a plane of %zu bytes", (int)e, hipGetErrorString(e), plane);
    }
    *outs[i] = a->parts[i];
  }
  *arena = a;
  return PDSP_OK;
}

int pdsp_planes_free(pdsp_arena *arena) {
  if (!arena) return PDSP_OK;
  DeviceGuard g(arena->device);
  for (void *p : arena->parts)
    if (p) (void)hipFree(p);
  delete arena;
  return PDSP_OK;
}

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
  return complex_op_f32(op, count, a_re, a_im, b_re, b_im, b_len, sr, si, out_re, out_im, s);
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
  if (len > 0 && batch > 0 && !samples) return fail(PDSP_ERR_BAD_ARG, "bad samples");
  FrameSource in;
  in.samples = samples;
  in.len = len;
  return spectrum_frames_host(in, batch, sample_rate, fft_size, window, sides, freq_out, amp_out, phase_out, peak_out,
                              bins_out);
}

int pdsp_spectrum_rows_host_f64(const double *const *rows, long long batch, long long len, double sample_rate,
                                long long fft_size, int window, int sides, double *freq_out, double *amp_out,
                                double *phase_out, pdsp_peak *peak_out, long long *bins_out) {
  if (batch > 0 && len > 0) {
    if (!rows) return fail(PDSP_ERR_BAD_ARG, "bad samples");
    for (long long b = 0; b < batch; ++b)
      if (!rows[b]) return fail(PDSP_ERR_BAD_ARG, "bad samples");
  }
  FrameSource in;
  in.rows64 = len > 0 ? rows : nullptr;
  in.len = len;
  return spectrum_frames_host(in, batch, sample_rate, fft_size, window, sides, freq_out, amp_out, phase_out, peak_out,
                              bins_out);
}

int pdsp_spectrum_rows_host_f32in(const float *const *rows, long long batch, long long len, double sample_rate,
                                  long long fft_size, int window, int sides, double *freq_out, double *amp_out,
                                  double *phase_out, pdsp_peak *peak_out, long long *bins_out) {
  if (batch > 0 && len > 0) {
    if (!rows) return fail(PDSP_ERR_BAD_ARG, "bad samples");
    for (long long b = 0; b < batch; ++b)
      if (!rows[b]) return fail(PDSP_ERR_BAD_ARG, "bad samples");
  }
  FrameSource in;
  in.rows32 = len > 0 ? rows : nullptr;
  in.len = len;
  return spectrum_frames_host(in, batch, sample_rate, fft_size, window, sides, freq_out, amp_out, phase_out, peak_out,
                              bins_out);
}

}  // extern "C"

namespace {

int spectrum_frames_host(const FrameSource &in, long long batch, double sample_rate, long long fft_size, int window,
                         int sides, double *freq_out, double *amp_out, double *phase_out, pdsp_peak *peak_out,
                         long long *bins_out) {
  const long long len = in.len;
  if (batch < 0) return fail(PDSP_ERR_BAD_ARG, "batch must be >= 0, got %lld", batch);
  if (len < 0) return fail(PDSP_ERR_BAD_ARG, "bad samples");
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
  // (len == 0: every frame is all zero padding; no frame is read)
  // findPeak runs on the host over the f64 amplitudes (host_peaks), inside the call's staging loop
  const int rc_run = f64 ? spectrum_host_t<double>(plan, in, window, sides, amp_out, phase_out, batch, freq_out, peak_out)
                         : spectrum_host_t<float>(plan, in, window, sides, amp_out, phase_out, batch, freq_out, peak_out);
  trim_stage(plan);
  return rc_run;
}

}  // namespace

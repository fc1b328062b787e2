// tools/kbench.hip -- standalone kernel micro-benchmark (development tool, not product).
// Times the N=4096 x 65536 C2C kernel against copy kernels with the same and with
// the ideal access pattern, interleaved in one process (guide rule 24).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I pragma-dsp_amd/csrc tools/kbench.hip -o tools/kbench
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#include "pdsp_fft_kernel.h"

#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e = (x);                                                             \
    if (e != hipSuccess) {                                                          \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
      exit(1);                                                                      \
    }                                                                               \
  } while (0)

// copy with the FFT kernel's access pattern: one 256-thread WG per row, 16 dword
// loads per plane per thread at tid + 256*q
template <bool NT>
__global__ void __launch_bounds__(256) copy_rowpattern(const float *__restrict__ re, const float *__restrict__ im,
                                                       float *__restrict__ ore, float *__restrict__ oim) {
  const size_t base = (size_t)blockIdx.x * 4096 + threadIdx.x;
  float a[16], b[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    a[q] = NT ? __builtin_nontemporal_load(re + base + 256 * q) : re[base + 256 * q];
    b[q] = NT ? __builtin_nontemporal_load(im + base + 256 * q) : im[base + 256 * q];
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    if (NT) {
      __builtin_nontemporal_store(a[q], ore + base + 256 * q);
      __builtin_nontemporal_store(b[q], oim + base + 256 * q);
    } else {
      ore[base + 256 * q] = a[q];
      oim[base + 256 * q] = b[q];
    }
  }
}

// classic float4 grid-stride copy of both planes
__global__ void __launch_bounds__(256) copy_float4(const float4 *__restrict__ in, float4 *__restrict__ out, size_t n4) {
  const size_t step = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += step) out[i] = in[i];
}

static void fill_tw(int log2n, std::vector<float2> &tw, int log2e = 4) {
  const pdsp::RadixPlan p = pdsp::make_radix_plan(log2n, log2e);
  tw.assign(p.twcount > 0 ? p.twcount : 1, make_float2(1, 0));
  for (int i = 0; i < p.np; ++i) {
    if (p.ns[i] <= 1) continue;
    for (int r = 1; r < p.r[i]; ++r)
      for (int k = 0; k < p.ns[i]; ++k) {
        const double ang = -2.0 * M_PI * r * k / ((double)p.ns[i] * p.r[i]);
        tw[p.twoff[i] + (r - 1) * p.ns[i] + k] = make_float2((float)cos(ang), (float)sin(ang));
      }
  }
}

// N=16384 fused spectrum (configs[3]) on a chunk of frames
static int spec_main(long long frames, int rounds) {
  const int n = 16384, m = n / 2, bins = m + 1;
  float *x, *amp, *win;
  CK(hipMalloc(&x, (size_t)frames * n * 4));
  CK(hipMalloc(&amp, (size_t)frames * bins * 4));
  CK(hipMalloc(&win, n * 4));
  {
    std::vector<float> h((size_t)frames * n);
    unsigned s = 777;
    for (auto &v : h) {
      s = s * 1664525u + 1013904223u;
      v = ((s >> 8) & 0xffff) / 32768.0f - 1.0f;
    }
    CK(hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> w(n);
    for (int i = 0; i < n; ++i) w[i] = (float)(0.5 * (1 - cos(2 * M_PI * i / (n - 1))));
    CK(hipMemcpy(win, w.data(), n * 4, hipMemcpyHostToDevice));
  }
  std::vector<float2> tw, twr(n / 4 + 1);
  fill_tw(13, tw, pdsp::packed_log2e(13));
  for (int k = 0; k <= n / 4; ++k) twr[k] = make_float2((float)cos(-2 * M_PI * k / n), (float)sin(-2 * M_PI * k / n));
  float2 *dtw, *dtwr;
  CK(hipMalloc(&dtw, tw.size() * 8));
  CK(hipMemcpy(dtw, tw.data(), tw.size() * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&dtwr, twr.size() * 8));
  CK(hipMemcpy(dtwr, twr.data(), twr.size() * 8, hipMemcpyHostToDevice));
  using TR = pdsp::FftTraits<13, pdsp::packed_log2e(13)>;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::vector<float> ms;
  std::vector<float2> tw12;
  fill_tw(12, tw12);
  float2 *dtw12;
  CK(hipMalloc(&dtw12, tw12.size() * 8));
  CK(hipMemcpy(dtw12, tw12.data(), tw12.size() * 8, hipMemcpyHostToDevice));
  const bool split = getenv("KB_SPLIT") != nullptr;
  struct SV {
    const char *name;
    std::function<void()> run;
    std::vector<float> ms;
  };
  std::vector<SV> svs;
  svs.push_back({"packed<13>", [&] {
    hipLaunchKernelGGL((pdsp::spectrum_packed_kernel<float, 13, true, 1, false>), dim3((frames + TR::ROWS - 1) / TR::ROWS),
                       dim3(TR::WG), 0, 0, x, win, pdsp::WinFused{nullptr, nullptr, 0.f, 0.f, 0.f, 1.f}, (long long)n, (long long)n, dtw, dtwr, amp, (float *)nullptr, 0,
                       1.0f / n, 2.0f / n, (pdsp::PeakRec *)nullptr, 0.0f, frames); }, {}});
  pdsp::WinFused wfz{nullptr, nullptr, 0.f, 0.f, 0.f, 1.f}, wfh = wfz;
  {
    const double f = 2 * M_PI / (n - 1);
    std::vector<float> hb(256 * 4), hs(64);
    for (int t = 0; t < 256; ++t)
      for (int e = 0; e < 2; ++e) {
        hb[4 * t + 2 * e] = (float)cos(f * (2 * t + e));
        hb[4 * t + 2 * e + 1] = (float)sin(f * (2 * t + e));
      }
    for (int q = 0; q < 16; ++q) {
      hs[2 * q] = (float)cos(f * 512 * q), hs[2 * q + 1] = (float)sin(f * 512 * q);
      hs[32 + 2 * q] = (float)cos(f * (512 * q + 8192)), hs[32 + 2 * q + 1] = (float)sin(f * (512 * q + 8192));
    }
    float *db, *dsx;
    CK(hipMalloc(&db, hb.size() * 4));
    CK(hipMalloc(&dsx, hs.size() * 4));
    CK(hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dsx, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
    wfh = pdsp::WinFused{db, dsx, 0.5f / n, -0.5f / n, 0.0f, 0.5f};  // pre-scaled by s_mid / 2 = 1/N (one-sided)
  }
  svs.push_back({"dif16k (x2 ld, x2 nt st)", [&] {
    hipLaunchKernelGGL((pdsp::spectrum_dif16k_kernel<float, 1, false>), dim3(frames), dim3(256), 0, 0, x, win, wfz,
                       (long long)n, dtw12, dtwr, amp, 1.0f / n, 2.0f / n, (pdsp::PeakRec *)nullptr, 0.0f, frames); }, {}});
  svs.push_back({"dif16k fused hann (2-term)", [&] {
    hipLaunchKernelGGL((pdsp::spectrum_dif16k_kernel<float, 2, false>), dim3(frames), dim3(256), 0, 0, x, win, wfh,
                       (long long)n, dtw12, dtwr, amp, 1.0f / n, 2.0f / n, (pdsp::PeakRec *)nullptr, 0.0f, frames); }, {}});
  svs.push_back({"dif16k fused 3-term (hann coefficients)", [&] {
    hipLaunchKernelGGL((pdsp::spectrum_dif16k_kernel<float, 3, false>), dim3(frames), dim3(256), 0, 0, x, win, wfh,
                       (long long)n, dtw12, dtwr, amp, 1.0f / n, 2.0f / n, (pdsp::PeakRec *)nullptr, 0.0f, frames); }, {}});
  svs.push_back({"dif16k rect window", [&] {
    hipLaunchKernelGGL((pdsp::spectrum_dif16k_kernel<float, 0, false>), dim3(frames), dim3(256), 0, 0, x, win, wfz,
                       (long long)n, dtw12, dtwr, amp, 1.0f / n, 2.0f / n, (pdsp::PeakRec *)nullptr, 0.0f, frames); }, {}});
  auto run = [&] { svs[split ? 1 : 0].run(); };
  for (int i = 0; i < 10; ++i)
    for (auto &v : svs) v.run();
  CK(hipDeviceSynchronize());
  for (int r = 0; r < rounds; ++r)
    for (auto &v : svs) {
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < 5; ++i) v.run();
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float t;
      CK(hipEventElapsedTime(&t, e0, e1));
      v.ms.push_back(t / 5);
    }
  CK(hipGetLastError());
  {
    // agreement of the variants' outputs at the fp32 tolerance of the path (the parity tests proper are
    // tests/test_gpu_spectrum.py): every dif16k variant against the packed kernel's rows
    std::vector<float> ref((size_t)64 * bins), got((size_t)64 * bins);
    svs[0].run();
    CK(hipMemcpy(ref.data(), amp + (size_t)(frames - 64) * bins, ref.size() * 4, hipMemcpyDeviceToHost));
    for (size_t vi = 1; vi + 1 < svs.size(); ++vi) {
      CK(hipMemset(amp, 0xff, (size_t)frames * bins * 4));
      svs[vi].run();
      CK(hipMemcpy(got.data(), amp + (size_t)(frames - 64) * bins, got.size() * 4, hipMemcpyDeviceToHost));
      double worst = 0;
      for (int r = 0; r < 64; ++r) {
        double mx = 0, err = 0;
        for (int i = 0; i < bins; ++i) {
          mx = std::max(mx, (double)ref[(size_t)r * bins + i]);
          const double d = std::fabs((double)ref[(size_t)r * bins + i] - (double)got[(size_t)r * bins + i]);
          err = std::max(err, std::isnan(d) ? 1e30 : d);
        }
        worst = std::max(worst, err / mx);
      }
      printf("%s vs packed<13> rows: max |diff| / max = %.3e\n", svs[vi].name, worst);
    }
  }
  const double sbytes = (4.0 * n + 4.0 * bins) * frames;
  for (auto &v : svs) {
    std::sort(v.ms.begin(), v.ms.end());
    printf("%-44s med %.4f ms  min %.4f ms  med %.0f GB/s  max %.0f GB/s\n", v.name, v.ms[v.ms.size() / 2], v.ms[0],
           sbytes / v.ms[v.ms.size() / 2] / 1e6, sbytes / v.ms[0] / 1e6);
  }
  ms = svs[1].ms;
  (void)run;
#ifdef PDSP_STAMPS
  {
    unsigned long long z[64] = {0}, acc[64];
    CK(hipMemcpyToSymbol(HIP_SYMBOL(pdsp::pdsp_stamp_acc), z, sizeof(z)));
    run();
    CK(hipDeviceSynchronize());
    CK(hipMemcpyFromSymbol(acc, HIP_SYMBOL(pdsp::pdsp_stamp_acc), sizeof(acc)));
    const double wgs = (double)((frames + TR::ROWS - 1) / TR::ROWS);
    const char *names[] = {"load+window", "passes total", "split+stores"};
    double tot = 0;
    for (int i = 0; i < 3; ++i) tot += acc[i] / wgs;
    for (int i = 0; i < 3; ++i) printf("  %-14s %9.0f cycles/WG  %5.1f %%\n", names[i], acc[i] / wgs, 100.0 * acc[i] / wgs / tot);
    if (split) {
      const char *sn[] = {"wait frame+window", "sub-transform 1", "sub-transform 2", "combine+LDS", "split+mag+stores"};
      for (int i = 0; i < 5; ++i) printf("  %-18s %9.0f cycles/WG\n", sn[i], acc[32 + i] / (double)frames);
    }
    for (int p = 0; p < 5; ++p)
      if (acc[8 + 4 * p])
        printf("    pass %d: compute+scatter %8.0f   barrier %8.0f   readback+barrier %8.0f\n", p, acc[8 + 4 * p] / wgs,
               acc[9 + 4 * p] / wgs, acc[10 + 4 * p] / wgs);
  }
#endif
  std::sort(ms.begin(), ms.end());
  const double bytes = (4.0 * n + 4.0 * bins) * frames;
  printf("spectrum16k paired=%d frames=%lld  med %.4f ms  min %.4f ms  med %.1f GB/s  max %.1f GB/s\n", 0, frames, ms[ms.size() / 2], ms[0], bytes / ms[ms.size() / 2] / 1e6, bytes / ms[0] / 1e6);
  return 0;
}

// N=16384 C2C rows on fft_split4_kernel (argv: rows rounds c16k)
static int c16k_main(long long rows, int rounds) {
  const int n = 16384;
  const size_t cnt = (size_t)rows * n;
  float *re, *im, *ore, *oim;
  CK(hipMalloc(&re, cnt * 4));
  CK(hipMalloc(&im, cnt * 4));
  CK(hipMalloc(&ore, cnt * 4));
  CK(hipMalloc(&oim, cnt * 4));
  {
    std::vector<float> h(cnt);
    unsigned s = 4242;
    for (auto &v : h) {
      s = s * 1664525u + 1013904223u;
      v = ((s >> 8) & 0xffff) / 32768.0f - 1.0f;
    }
    CK(hipMemcpy(re, h.data(), cnt * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(im, h.data() + 1, (cnt - 1) * 4, hipMemcpyHostToDevice));
  }
  std::vector<float2> tw12, tws(768);
  fill_tw(12, tw12);
  for (int k = 0; k < 768; ++k) tws[k] = make_float2((float)cos(-2 * M_PI * k / n), (float)sin(-2 * M_PI * k / n));
  float2 *dtw12, *dtws;
  CK(hipMalloc(&dtw12, tw12.size() * 8));
  CK(hipMemcpy(dtw12, tw12.data(), tw12.size() * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&dtws, tws.size() * 8));
  CK(hipMemcpy(dtws, tws.data(), tws.size() * 8, hipMemcpyHostToDevice));
  pdsp::LoadComplex<float> ld{re, im, n};
  pdsp::StoreComplex<float> st{ore, oim, n, 1.0f};
  auto run = [&] {
    hipLaunchKernelGGL((pdsp::fft_split4_kernel<float, 12, decltype(ld), decltype(st)>), dim3(rows), dim3(256), 0, 0, ld,
                       st, dtw12, dtws, rows);
  };
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::vector<float> ms;
  for (int i = 0; i < 20; ++i) run();
  CK(hipDeviceSynchronize());
  for (int r = 0; r < rounds; ++r) {
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 5; ++i) run();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float t;
    CK(hipEventElapsedTime(&t, e0, e1));
    ms.push_back(t / 5);
  }
  CK(hipGetLastError());
  std::sort(ms.begin(), ms.end());
  const double bytes = 16.0 * cnt;
  printf("c2c16k rows=%lld paired=%d  med %.4f ms  min %.4f ms  med %.1f GB/s  max %.1f GB/s\n", rows, 0,
         ms[ms.size() / 2], ms[0], bytes / ms[ms.size() / 2] / 1e6, bytes / ms[0] / 1e6);
  return 0;
}

// HBM ceilings by read:write mix (argv: MiB rounds mix): float4 grid-stride kernels, non-temporal
template <int R, int W, bool SCALAR_ST = false>
__global__ void __launch_bounds__(256) mix_kernel(const float4 *__restrict__ in_, float4 *__restrict__ out_, size_t n4) {
  typedef float V4 __attribute__((ext_vector_type(4)));
  const V4 *in = reinterpret_cast<const V4 *>(in_);
  V4 *out = reinterpret_cast<V4 *>(out_);
  // each step reads R float4 (from R planes of n4) and writes W float4 (to W planes of n4)
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    V4 acc = V4{0, 0, 0, 0};
    for (int r = 0; r < R; ++r) acc += __builtin_nontemporal_load(in + (size_t)r * n4 + i);
    if (W == 0) {
      if (acc.x == 12345.678f) out[0] = acc;  // never true: keeps the loads alive
    } else if (SCALAR_ST) {
      // dword stores, unit stride across the lanes (the FFT kernels' store shape): 4 per float4
      float *o = reinterpret_cast<float *>(out);
      const size_t blk = (i / 64) * 256, lane = i % 64;
      for (int w = 0; w < W; ++w)
        for (int j = 0; j < 4; ++j) __builtin_nontemporal_store(acc[j], o + (size_t)w * n4 * 4 + blk + 64 * j + lane);
    } else {
      for (int w = 0; w < W; ++w) __builtin_nontemporal_store(acc, out + (size_t)w * n4 + i);
    }
  }
}

template <int R, int W, bool SCALAR_ST = false>
static void mix_run(const float4 *in, float4 *out, size_t n4, int rounds, int grid) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::vector<float> ms;
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((mix_kernel<R, W, SCALAR_ST>), dim3(grid), dim3(256), 0, 0, in, out, n4);
  CK(hipDeviceSynchronize());
  for (int r = 0; r < rounds; ++r) {
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((mix_kernel<R, W, SCALAR_ST>), dim3(grid), dim3(256), 0, 0, in, out, n4);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float t;
    CK(hipEventElapsedTime(&t, e0, e1));
    ms.push_back(t / 5);
  }
  std::sort(ms.begin(), ms.end());
  const double bytes = 16.0 * n4 * (R + W);
  printf("mix read:write %d:%d%s grid %5d  med %.4f ms  %.0f GB/s (max %.0f)\n", R, W, SCALAR_ST ? " (dword stores)" : "", grid, ms[ms.size() / 2],
         bytes / ms[ms.size() / 2] / 1e6, bytes / ms[0] / 1e6);
}

static int mix_main(long long mib, int rounds) {
  const size_t n4 = (size_t)mib * 1024 * 1024 / 16;  // float4 per plane
  float4 *in, *out;
  CK(hipMalloc(&in, n4 * 16 * 4));
  CK(hipMalloc(&out, n4 * 16 * 2));
  CK(hipMemset(in, 0, n4 * 16 * 4));
  for (int grid : {2048, 8192, 65536}) {
    mix_run<1, 0>(in, out, n4 * 4, rounds, grid);
    mix_run<1, 1>(in, out, n4 * 2, rounds, grid);
    mix_run<2, 1>(in, out, n4 * 2, rounds, grid);
    mix_run<2, 1, true>(in, out, n4 * 2, rounds, grid);
    mix_run<1, 1, true>(in, out, n4 * 2, rounds, grid);
    mix_run<4, 1>(in, out, n4, rounds, grid);
    mix_run<1, 2>(in, out, n4, rounds, grid);
  }
  return 0;
}

// XCD-locality probe (argv: rows rounds xcd): the N=4096 C2C kernel with the workgroup -> row map
//   row = 8P*g + P*((xcd + c) % 8) + j,   xcd = blockIdx % 8, j = (blockIdx % 8P) / 8, g = blockIdx / 8P
// i.e. every run of P consecutive rows goes to one XCD, rotated by c.  If HBM stacks have an affinity
// to XCDs, some (P, c) stands out.
template <int P>
__global__ void __launch_bounds__(256)
fft_remap_kernel(const float *__restrict__ re, const float *__restrict__ im, float *__restrict__ ore,
                 float *__restrict__ oim, const float2 *__restrict__ tw, const int c, const long long batch) {
  using namespace pdsp;
  using TR = FftTraits<12>;
  constexpr int E = 16, TP = 256;
  __shared__ cx<float> lds[TR::LDS_ELEMS];
  const int tid = (int)threadIdx.x;
  const long long b = blockIdx.x;
  const long long g = b / (8 * P);
  const int w = (int)(b % (8 * P)), xcd = w % 8, j = w / 8;
  const long long row = uniform_row<TP>(g * (8 * P) + P * ((xcd + c) % 8) + j);
  LoadComplex<float> ld{re, im, 4096};
  StoreComplex<float> st{ore, oim, 4096, 1.0f};
  cx<float> x[E];
  static_for<E>([&](auto q) { x[q] = ld(row, TP * q, tid); });
  RegTwiddles<float, 12> twf;
  twf.load(reinterpret_cast<const cx<float> *>(tw), tid);
  fft_passes<float, 12, false>(x, lds, twf, tid);
  static_for<E>([&](auto q) { st(row, TP * q, tid, x[q]); });
}

template <int P>
static void xcd_run(const float *re, const float *im, float *ore, float *oim, const float2 *dtw, long long batch, int rounds) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  printf("P=%2d:", P);
  for (int c = 0; c < 8; ++c) {
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((fft_remap_kernel<P>), dim3(batch), dim3(256), 0, 0, re, im, ore, oim, dtw, c, batch);
    CK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int r = 0; r < rounds; ++r) {
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((fft_remap_kernel<P>), dim3(batch), dim3(256), 0, 0, re, im, ore, oim, dtw, c, batch);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float t;
      CK(hipEventElapsedTime(&t, e0, e1));
      best = std::min(best, t / 5);
    }
    printf(" %5.0f", 16.0 * batch * 4096 / best / 1e6);
  }
  printf("  GB/s for c = 0..7\n");
}

static int xcd_main(long long batch, int rounds) {
  const size_t cnt = (size_t)batch * 4096;
  float *re, *im, *ore, *oim;
  CK(hipMalloc(&re, cnt * 4));
  CK(hipMalloc(&im, cnt * 4));
  CK(hipMalloc(&ore, cnt * 4));
  CK(hipMalloc(&oim, cnt * 4));
  CK(hipMemset(re, 0, cnt * 4));
  CK(hipMemset(im, 0, cnt * 4));
  std::vector<float2> tw;
  fill_tw(12, tw);
  float2 *dtw;
  CK(hipMalloc(&dtw, tw.size() * 8));
  CK(hipMemcpy(dtw, tw.data(), tw.size() * 8, hipMemcpyHostToDevice));
  printf("planes at %p %p %p %p\n", (void *)re, (void *)im, (void *)ore, (void *)oim);
  for (int rep = 0; rep < 2; ++rep) {
    xcd_run<1>(re, im, ore, oim, dtw, batch, rounds);
    xcd_run<2>(re, im, ore, oim, dtw, batch, rounds);
    xcd_run<4>(re, im, ore, oim, dtw, batch, rounds);
    xcd_run<8>(re, im, ore, oim, dtw, batch, rounds);
    xcd_run<16>(re, im, ore, oim, dtw, batch, rounds);
    xcd_run<64>(re, im, ore, oim, dtw, batch, rounds);
  }
  return 0;
}

// I/O skeleton of the N=16384 spectrum kernel (argv: frames rounds skel): 64 KB frame in with 16-byte
// loads, 8193 amplitudes out with dword stores in the kernel's two directions, no transform.
//   THREADS per frame-workgroup, NT loads / NT stores, HALVES = 1: all loads first; 2: two load/store rounds
template <int THREADS, bool NT_LD, bool NT_ST, int HALVES, int PITCH = 8193>
__global__ void __launch_bounds__(THREADS)
skel_kernel(const float *__restrict__ frames, float *__restrict__ amp, long long nframes) {
  typedef float V4 __attribute__((ext_vector_type(4)));
  constexpr int N = 16384, M = 8192, PER = N / 4 / THREADS;  // V4 loads per thread
  const int tid = (int)threadIdx.x;
  const long long row = blockIdx.x;
  if (row >= nframes) return;
  const V4 *x4 = reinterpret_cast<const V4 *>(frames + (size_t)row * N);
  float *arow = amp + (size_t)row * PITCH;
  #pragma unroll
  for (int h = 0; h < HALVES; ++h) {
    V4 v[PER / HALVES];
    #pragma unroll
    for (int q = 0; q < PER / HALVES; ++q) {
      const V4 *p = x4 + THREADS * (q + h * (PER / HALVES)) + tid;
      v[q] = NT_LD ? __builtin_nontemporal_load(p) : *p;
    }
    #pragma unroll
    for (int q = 0; q < PER / HALVES; ++q) {
      const int k = tid + THREADS * (q + h * (PER / HALVES));  // < 4096
      const float ma = sqrtf(v[q].x * v[q].x + v[q].y * v[q].y), mb = sqrtf(v[q].z * v[q].z + v[q].w * v[q].w);
      if (NT_ST) {
        __builtin_nontemporal_store(ma, arow + k);
        __builtin_nontemporal_store(mb, arow + (M - k));
      } else {
        arow[k] = ma;
        arow[M - k] = mb;
      }
    }
  }
  if (tid == 0) arow[4096] = 1.0f;
}

template <int THREADS, bool NT_LD, bool NT_ST, int HALVES, int PITCH = 8193>
static void skel_run(const float *x, float *amp, long long frames, int rounds) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::vector<float> ms;
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((skel_kernel<THREADS, NT_LD, NT_ST, HALVES, PITCH>), dim3(frames), dim3(THREADS), 0, 0, x, amp, frames);
  CK(hipDeviceSynchronize());
  for (int r = 0; r < rounds; ++r) {
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((skel_kernel<THREADS, NT_LD, NT_ST, HALVES, PITCH>), dim3(frames), dim3(THREADS), 0, 0, x, amp, frames);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float t;
    CK(hipEventElapsedTime(&t, e0, e1));
    ms.push_back(t / 5);
  }
  std::sort(ms.begin(), ms.end());
  const double bytes = (4.0 * 16384 + 4.0 * 8193) * frames;
  printf("skel pitch=%d threads=%4d ntld=%d ntst=%d halves=%d  med %.4f ms  %.0f GB/s (max %.0f)\n", PITCH, THREADS, NT_LD, NT_ST, HALVES,
         ms[ms.size() / 2], bytes / ms[ms.size() / 2] / 1e6, bytes / ms[0] / 1e6);
}

static int skel_main(long long frames, int rounds) {
  float *x, *amp;
  CK(hipMalloc(&x, (size_t)frames * 16384 * 4));
  CK(hipMalloc(&amp, (size_t)frames * 8320 * 4));
  CK(hipMemset(x, 0, (size_t)frames * 16384 * 4));
  for (int rep = 0; rep < 2; ++rep) {
    skel_run<256, true, false, 1>(x, amp, frames, rounds);
    skel_run<256, true, false, 1, 8208>(x, amp, frames, rounds);  // rows padded to a multiple of 64 B
    skel_run<256, true, false, 1, 8256>(x, amp, frames, rounds);  // ... of 256 B
    skel_run<256, true, true, 1, 8256>(x, amp, frames, rounds);
    skel_run<256, false, false, 1>(x, amp, frames, rounds);
    skel_run<256, true, true, 1>(x, amp, frames, rounds);
    skel_run<256, true, false, 2>(x, amp, frames, rounds);
    skel_run<256, true, false, 4>(x, amp, frames, rounds);
    skel_run<512, true, false, 1>(x, amp, frames, rounds);
    skel_run<1024, true, false, 1>(x, amp, frames, rounds);
    skel_run<128, true, false, 1>(x, amp, frames, rounds);
    skel_run<128, true, false, 4>(x, amp, frames, rounds);
  }
  return 0;
}

// Skeleton of the packed spectrum kernel at mid sizes (argv: MiB rounds skelmid): N-sample frames read
// with 8-byte loads by TP = N/32 threads each (16 loads per thread), 256/TP frames per workgroup,
// N/2+1 amplitudes per frame out with dword stores in two directions.
template <int N>
__global__ void __launch_bounds__(256) skelmid_kernel(const float *__restrict__ frames, float *__restrict__ amp, long long nframes) {
  typedef float V2 __attribute__((ext_vector_type(2)));
  constexpr int M = N / 2, TP = M / 16, ROWS = 256 / TP;
  const int tid = (int)threadIdx.x % TP, rloc = (int)threadIdx.x / TP;
  const long long row = (long long)blockIdx.x * ROWS + rloc;
  if (row >= nframes) return;
  const V2 *x2 = reinterpret_cast<const V2 *>(frames + (size_t)row * N);
  float *arow = amp + (size_t)row * (M + 1);
  V2 v[16];
  #pragma unroll
  for (int q = 0; q < 16; ++q) v[q] = __builtin_nontemporal_load(x2 + TP * q + tid);
  #pragma unroll
  for (int q = 0; q < 9; ++q) {  // bins k and M-k for k = tid + TP*q <= M/2
    const int k = tid + TP * q;
    if (q < 8 || tid == 0) {
      const float ma = sqrtf(v[q % 16].x * v[q % 16].x + v[q % 16].y * v[q % 16].y);
      const float mb = sqrtf(v[(q + 8) % 16].x * v[(q + 8) % 16].x + v[(q + 8) % 16].y * v[(q + 8) % 16].y);
      arow[k] = ma;
      if (M - k != k) arow[M - k] = mb;
    }
  }
}

template <int N>
static void skelmid_run(const float *x, float *amp, long long mib, int rounds) {
  const long long frames = mib * 1024 * 1024 / 4 / N;
  constexpr int ROWS = 256 / (N / 32);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::vector<float> ms;
  const unsigned blocks = (unsigned)((frames + ROWS - 1) / ROWS);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((skelmid_kernel<N>), dim3(blocks), dim3(256), 0, 0, x, amp, frames);
  CK(hipDeviceSynchronize());
  for (int r = 0; r < rounds; ++r) {
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((skelmid_kernel<N>), dim3(blocks), dim3(256), 0, 0, x, amp, frames);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float t;
    CK(hipEventElapsedTime(&t, e0, e1));
    ms.push_back(t / 5);
  }
  std::sort(ms.begin(), ms.end());
  const double bytes = (4.0 * N + 4.0 * (N / 2 + 1)) * frames;
  printf("skelmid N=%5d  med %.4f ms  %.0f GB/s (max %.0f)\n", N, ms[ms.size() / 2], bytes / ms[ms.size() / 2] / 1e6, bytes / ms[0] / 1e6);
}

static int skelmid_main(long long mib, int rounds) {
  float *x, *amp;
  CK(hipMalloc(&x, (size_t)mib << 20));
  CK(hipMalloc(&amp, ((size_t)mib << 19) + (1 << 20)));
  CK(hipMemset(x, 0, (size_t)mib << 20));
  for (int rep = 0; rep < 2; ++rep) {
    skelmid_run<1024>(x, amp, mib, rounds);
    skelmid_run<2048>(x, amp, mib, rounds);
    skelmid_run<4096>(x, amp, mib, rounds);
    skelmid_run<8192>(x, amp, mib, rounds);
  }
  return 0;
}

int main(int argc, char **argv) {
  if (argc > 3 && std::string(argv[3]) == "skelmid") return skelmid_main(atoll(argv[1]), atoi(argv[2]));
  if (argc > 3 && std::string(argv[3]) == "skel") return skel_main(atoll(argv[1]), atoi(argv[2]));
  if (argc > 3 && std::string(argv[3]) == "xcd") return xcd_main(atoll(argv[1]), atoi(argv[2]));
  if (argc > 3 && std::string(argv[3]) == "mix") return mix_main(atoll(argv[1]), atoi(argv[2]));
  if (argc > 3 && std::string(argv[3]) == "c16k") return c16k_main(atoll(argv[1]), atoi(argv[2]));
  if (argc > 3 && std::string(argv[3]) == "spec") return spec_main(atoll(argv[1]), atoi(argv[2]));
  const int n = 4096;
  const long long batch = argc > 1 ? atoll(argv[1]) : 65536;
  const int rounds = argc > 2 ? atoi(argv[2]) : 10;
  const size_t cnt = (size_t)batch * n;
  float *re, *im, *ore, *oim;
  CK(hipMalloc(&re, cnt * 4));
  CK(hipMalloc(&im, cnt * 4));
  CK(hipMalloc(&ore, cnt * 4));
  CK(hipMalloc(&oim, cnt * 4));
  {
    std::vector<float> h(cnt);
    unsigned s = 12345;
    for (size_t i = 0; i < cnt; ++i) {
      s = s * 1664525u + 1013904223u;
      h[i] = ((s >> 8) & 0xffff) / 32768.0f - 1.0f;
    }
    CK(hipMemcpy(re, h.data(), cnt * 4, hipMemcpyHostToDevice));
    for (size_t i = 0; i < cnt; ++i) {
      s = s * 1664525u + 1013904223u;
      h[i] = ((s >> 8) & 0xffff) / 32768.0f - 1.0f;
    }
    CK(hipMemcpy(im, h.data(), cnt * 4, hipMemcpyHostToDevice));
  }
  // twiddles
  const pdsp::RadixPlan p = pdsp::make_radix_plan(12);
  std::vector<float2> tw(p.twcount);
  for (int i = 0; i < p.np; ++i) {
    if (p.ns[i] <= 1) continue;
    for (int r = 1; r < p.r[i]; ++r)
      for (int k = 0; k < p.ns[i]; ++k) {
        const double ang = -2.0 * M_PI * r * k / ((double)p.ns[i] * p.r[i]);
        tw[p.twoff[i] + (r - 1) * p.ns[i] + k] = make_float2((float)cos(ang), (float)sin(ang));
      }
  }
  float2 *dtw;
  CK(hipMalloc(&dtw, tw.size() * sizeof(float2)));
  CK(hipMemcpy(dtw, tw.data(), tw.size() * sizeof(float2), hipMemcpyHostToDevice));

  using TR = pdsp::FftTraits<12>;
  struct Variant {
    std::string name;
    std::function<void()> run;
    std::vector<float> ms;
  };
  std::vector<Variant> vs;
  vs.push_back({"copy_float4", [&] {
                  hipLaunchKernelGGL(copy_float4, dim3(2048), dim3(256), 0, 0, (const float4 *)re, (float4 *)ore, cnt / 4);
                  hipLaunchKernelGGL(copy_float4, dim3(2048), dim3(256), 0, 0, (const float4 *)im, (float4 *)oim, cnt / 4);
                }});
  vs.push_back({"copy_rowpattern", [&] {
                  hipLaunchKernelGGL(copy_rowpattern<false>, dim3(batch), dim3(256), 0, 0, re, im, ore, oim);
                }});
  vs.push_back({"copy_rowpattern_nt", [&] {
                  hipLaunchKernelGGL(copy_rowpattern<true>, dim3(batch), dim3(256), 0, 0, re, im, ore, oim);
                }});
  vs.push_back({"fft_v1", [&] {
                  pdsp::LoadComplex<float> ld{re, im, n};
                  pdsp::StoreComplex<float> st{ore, oim, n, 1.0f};
                  hipLaunchKernelGGL((pdsp::fft_stockham_kernel<float, 12, decltype(ld), decltype(st)>),
                                     dim3((batch + TR::ROWS - 1) / TR::ROWS), dim3(TR::WG), 0, 0, ld, st, dtw, batch);
                }});
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (auto &v : vs) v.run();  // warm-up
  CK(hipDeviceSynchronize());
  for (int r = 0; r < rounds; ++r) {
    for (auto &v : vs) {
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < 5; ++i) v.run();
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      v.ms.push_back(ms / 5);
    }
  }
  CK(hipGetLastError());
  const double bytes = 16.0 * cnt;
  printf("%-22s %10s %10s %10s %10s\n", "variant", "med_ms", "min_ms", "med_GB/s", "max_GB/s");
  for (auto &v : vs) {
    std::sort(v.ms.begin(), v.ms.end());
    const float med = v.ms[v.ms.size() / 2], mn = v.ms[0];
    printf("%-22s %10.4f %10.4f %10.1f %10.1f\n", v.name.c_str(), med, mn, bytes / med / 1e6, bytes / mn / 1e6);
  }
  return 0;
}

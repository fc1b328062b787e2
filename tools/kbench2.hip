// tools/kbench2.hip -- round-2 micro-benchmarks (development tool, not product): I/O skeletons of the
// N=16384 fused spectrum kernel (64 KB frame in, 8193 amplitudes out, no transform) in the shapes round 2
// asks about -- wide stores staged through LDS, LDS-DMA frame loads, persistent workgroups with the next
// frame in flight -- next to the shipped shape and to grid-stride copies of the same read:write mix, all
// interleaved in one process (guide rule 24).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/kbench2.hip -o tools/kbench2
//   tools/kbench2 <frames> <rounds>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e = (x);                                                             \
    if (e != hipSuccess) {                                                          \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
      exit(1);                                                                      \
    }                                                                               \
  } while (0)

typedef float V4 __attribute__((ext_vector_type(4)));
constexpr int N = 16384, M = 8192, H = 4096, BINS = M + 1;

__device__ __forceinline__ float mag2(float a, float b) { return __builtin_amdgcn_sqrtf(a * a + b * b); }

// MODE 0: the shipped shape -- 16 non-temporal 16-byte loads per thread, all issued first; plain dword
//         stores arow[k] (forward) and arow[M-k] (lanes reversed).
// MODE 1: both store streams forward (arow[k], arow[H+k]).
// MODE 2: amplitudes staged through an LDS row (shifted by the row's misalignment), then aligned
//         16-byte global stores; the ragged first / last group by dword stores.
// MODE 3: as 2 with non-temporal 16-byte stores.
// LD 0: 16 x4 loads at V4 index tid + 256q (shipped).  LD 1: 64 dword loads at tid + 256j.  LD 2: 32 x2 loads
// at V2 index tid + 256j.  LD 3: 16 x4 loads, each WAVE streaming its own contiguous 16 KB quarter.
template <int MODE, int LD = 0>
__global__ void __launch_bounds__(256) skel_kernel(const float *__restrict__ frames, float *__restrict__ amp,
                                                   long long nframes) {
  typedef float V2 __attribute__((ext_vector_type(2)));
  const int tid = (int)threadIdx.x;
  const long long row = blockIdx.x;
  if (row >= nframes) return;
  const V4 *x4 = reinterpret_cast<const V4 *>(frames + (size_t)row * N);
  float *arow = amp + (size_t)row * BINS;
  V4 v[16];
  if constexpr (LD == 0) {
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = __builtin_nontemporal_load(x4 + 256 * q + tid);
  } else if constexpr (LD == 1) {
    const float *x1 = frames + (size_t)row * N;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      v[q].x = __builtin_nontemporal_load(x1 + 256 * (4 * q + 0) + tid);
      v[q].y = __builtin_nontemporal_load(x1 + 256 * (4 * q + 1) + tid);
      v[q].z = __builtin_nontemporal_load(x1 + 256 * (4 * q + 2) + tid);
      v[q].w = __builtin_nontemporal_load(x1 + 256 * (4 * q + 3) + tid);
    }
  } else if constexpr (LD == 2) {
    const V2 *x2 = reinterpret_cast<const V2 *>(frames + (size_t)row * N);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const V2 a = __builtin_nontemporal_load(x2 + 256 * (2 * q) + tid), b = __builtin_nontemporal_load(x2 + 256 * (2 * q + 1) + tid);
      v[q] = V4{a.x, a.y, b.x, b.y};
    }
  } else {
    const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = __builtin_nontemporal_load(x4 + 1024 * wave + 64 * q + lane);
  }
  if constexpr (MODE == 4 || MODE == 5) {
    // the store shape of a decimation-in-frequency top split: a thread owns bins (2k, 2k+1) and their
    // mirrors (8191-2k, 8192-2k), k = tid + 256q, q < 8: two 8-byte stores per k, no LDS row
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int k = tid + 256 * q;
      const V2 lo = V2{mag2(v[q].x, v[q].y), mag2(v[q].z, v[q].w)};
      const V2 hi = V2{mag2(v[q + 8].z, v[q + 8].w), mag2(v[q + 8].x, v[q + 8].y)};
      if (MODE == 5) {
        __builtin_nontemporal_store(lo, reinterpret_cast<V2 *>(arow + 2 * k));
        __builtin_nontemporal_store(hi, reinterpret_cast<V2 *>(arow + (M - 1 - 2 * k)));
      } else {
        *reinterpret_cast<V2 *>(arow + 2 * k) = lo;
        *reinterpret_cast<V2 *>(arow + (M - 1 - 2 * k)) = hi;
      }
    }
    if (tid == 0) arow[H] = 1.0f;
  } else if constexpr (MODE <= 1) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int k = tid + 256 * q;
      const float ma = mag2(v[q].x, v[q].y), mb = mag2(v[q].z, v[q].w);
      arow[k] = ma;
      if (MODE == 0) arow[M - k] = mb;
      else arow[H + 1 + k] = mb;
    }
    if (tid == 0) arow[H] = 1.0f;
  } else {
    __shared__ float lrow[BINS + 8];
    const int o = (int)(((size_t)row * BINS) & 3);  // floats the row start lies past a 16-byte boundary
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int k = tid + 256 * q;
      lrow[o + k] = mag2(v[q].x, v[q].y);
      lrow[o + M - k] = mag2(v[q].z, v[q].w);
    }
    if (tid == 0) lrow[o + H] = 1.0f;
    __syncthreads();
    float *const abase = arow - o;                // 16-byte aligned
    const int groups = (o + BINS + 3) / 4;        // 2049 or 2050
#pragma unroll
    for (int r = 0; r < 9; ++r) {
      const int g = tid + 256 * r;
      if (g < groups) {
        const V4 w = *reinterpret_cast<const V4 *>(lrow + 4 * g);
        if (4 * g >= o && 4 * g + 3 < o + BINS) {
          if (MODE == 3) __builtin_nontemporal_store(w, reinterpret_cast<V4 *>(abase + 4 * g));
          else *reinterpret_cast<V4 *>(abase + 4 * g) = w;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (4 * g + j >= o && 4 * g + j < o + BINS) abase[4 * g + j] = w[j];
        }
      }
    }
  }
}

// Dword loads (the C2C kernel's load shape: 64 per thread at tid + 256*j), dword stores as MODE 0.
__global__ void __launch_bounds__(256) skel_dword_kernel(const float *__restrict__ frames, float *__restrict__ amp,
                                                         long long nframes) {
  const int tid = (int)threadIdx.x;
  const long long row = blockIdx.x;
  if (row >= nframes) return;
  const float *x = frames + (size_t)row * N;
  float *arow = amp + (size_t)row * BINS;
  float v[64];
#pragma unroll
  for (int j = 0; j < 64; ++j) v[j] = __builtin_nontemporal_load(x + 256 * j + tid);
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int k = tid + 256 * q;
    arow[k] = mag2(v[q], v[q + 16]);
    arow[M - k] = mag2(v[q + 32], v[q + 48]);
  }
  if (tid == 0) arow[H] = 1.0f;
}

// LDS-DMA frame loads: 16 global_load_lds_dwordx4 per thread land the 64 KB frame in LDS (no VGPRs), one
// vmcnt(0) + barrier, then ds_read_b128 + the MODE 0 stores.  64 KB of LDS: 2 workgroups per CU.
__global__ void __launch_bounds__(256) skel_dma_kernel(const float *__restrict__ frames, float *__restrict__ amp,
                                                       long long nframes) {
  __shared__ V4 stage[N / 4];
  const int tid = (int)threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long long row = blockIdx.x;
  if (row >= nframes) return;
  const V4 *x4 = reinterpret_cast<const V4 *>(frames + (size_t)row * N);
  float *arow = amp + (size_t)row * BINS;
#pragma unroll
  for (int q = 0; q < 16; ++q)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(x4 + 256 * q + tid),
                                     (__attribute__((address_space(3))) void *)(stage + 256 * q + 64 * wave), 16, 0, 2 /* nt */);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int k = tid + 256 * q;
    const V4 w = stage[256 * q + tid];
    arow[k] = mag2(w.x, w.y);
    arow[M - k] = mag2(w.z, w.w);
  }
  if (tid == 0) arow[H] = 1.0f;
}

// Persistent workgroups: `gridDim.x` workgroups walk the frames (row = blockIdx.x + i*gridDim.x); the next
// frame's 16 loads are issued BEFORE the current frame's stores, so every workgroup always has a frame in
// flight (two register sets, named: no run-time indexing).
__global__ void __launch_bounds__(256) skel_persist_kernel(const float *__restrict__ frames, float *__restrict__ amp,
                                                           long long nframes) {
  const int tid = (int)threadIdx.x;
  long long row = blockIdx.x;
  if (row >= nframes) return;
  V4 a[16], b[16];
  {
    const V4 *x4 = reinterpret_cast<const V4 *>(frames + (size_t)row * N);
#pragma unroll
    for (int q = 0; q < 16; ++q) a[q] = __builtin_nontemporal_load(x4 + 256 * q + tid);
  }
  for (;;) {
    const long long nxt = row + gridDim.x;
    const bool more = nxt < nframes;
    {
      const V4 *x4 = reinterpret_cast<const V4 *>(frames + (size_t)(more ? nxt : row) * N);
#pragma unroll
      for (int q = 0; q < 16; ++q) b[q] = __builtin_nontemporal_load(x4 + 256 * q + tid);
    }
    {
      float *arow = amp + (size_t)row * BINS;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int k = tid + 256 * q;
        arow[k] = mag2(a[q].x, a[q].y);
        arow[M - k] = mag2(a[q].z, a[q].w);
      }
      if (tid == 0) arow[H] = 1.0f;
    }
    if (!more) break;
    row = nxt;
#pragma unroll
    for (int q = 0; q < 16; ++q) a[q] = b[q];
  }
}

// Grid-stride copies with the kernel's byte mix (2 floats read : 1 float written), 16-byte accesses.
template <bool NT_ST>
__global__ void __launch_bounds__(256) mix21_kernel(const V4 *__restrict__ in, V4 *__restrict__ out, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const V4 a = __builtin_nontemporal_load(in + i), b = __builtin_nontemporal_load(in + n4 + i);
    const V4 r = a + b;
    if (NT_ST) __builtin_nontemporal_store(r, out + i);
    else out[i] = r;
  }
}
// read-only and 1:1 for the scale
__global__ void __launch_bounds__(256) read_kernel(const V4 *__restrict__ in, V4 *__restrict__ out, size_t n4) {
  V4 acc = V4{0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256)
    acc += __builtin_nontemporal_load(in + i);
  if (acc.x == 12345.678f) out[0] = acc;
}

// I/O skeleton of the planar C2C kernel at N = 4096: one 256-thread workgroup per row, both planes in,
// both planes out, W floats per lane per access (W = 1: the shipped dword shape tid + 256q).
template <int W, bool NT_ST>
__global__ void __launch_bounds__(256) c2c_skel_kernel(const float *__restrict__ re, const float *__restrict__ im,
                                                       float *__restrict__ ore, float *__restrict__ oim) {
  typedef float VW __attribute__((ext_vector_type(W)));
  constexpr int PER = 16 / W;  // accesses per plane per thread
  const size_t base = (size_t)blockIdx.x * 4096;
  const VW *r = reinterpret_cast<const VW *>(re + base), *i = reinterpret_cast<const VW *>(im + base);
  VW *orr = reinterpret_cast<VW *>(ore + base), *oi = reinterpret_cast<VW *>(oim + base);
  VW a[PER], b[PER];
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    a[q] = __builtin_nontemporal_load(r + 256 * q + threadIdx.x);
    b[q] = __builtin_nontemporal_load(i + 256 * q + threadIdx.x);
  }
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const VW x = a[q] + b[q], y = a[q] - b[q];
    if (NT_ST) {
      __builtin_nontemporal_store(x, orr + 256 * q + threadIdx.x);
      __builtin_nontemporal_store(y, oi + 256 * q + threadIdx.x);
    } else {
      orr[256 * q + threadIdx.x] = x;
      oi[256 * q + threadIdx.x] = y;
    }
  }
}
// one element per thread (tiny workgroups), 16-byte accesses: the 1:1 ceiling of the part
template <bool NT_ST>
__global__ void __launch_bounds__(256) copy11_kernel(const V4 *__restrict__ in, V4 *__restrict__ out, size_t n4) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n4) {
    const V4 v = __builtin_nontemporal_load(in + i);
    if (NT_ST) __builtin_nontemporal_store(v, out + i);
    else out[i] = v;
  }
}

int main(int argc, char **argv) {
  const long long frames = argc > 1 ? atoll(argv[1]) : 16384;
  const int rounds = argc > 2 ? atoi(argv[2]) : 10;
  float *x, *amp;
  CK(hipMalloc(&x, (size_t)frames * N * 4));
  CK(hipMalloc(&amp, (size_t)frames * (BINS + 64) * 4));
  {
    std::vector<float> h((size_t)frames * N);
    unsigned s = 777;
    for (auto &v : h) {
      s = s * 1664525u + 1013904223u;
      v = ((s >> 8) & 0xffff) / 32768.0f - 1.0f;
    }
    CK(hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  }
  struct Variant {
    std::string name;
    std::function<void()> run;
    double bytes;
    std::vector<float> ms;
  };
  const double fb = (4.0 * N + 4.0 * BINS) * frames;
  std::vector<Variant> vs;
#define SK(NAME, ...) vs.push_back({NAME, [&] { __VA_ARGS__; }, fb, {}})
  SK("skel shipped (x4 nt ld, dword st fwd+rev)", hipLaunchKernelGGL(skel_kernel<0>, dim3(frames), dim3(256), 0, 0, x, amp, frames));
  SK("skel dword st fwd+fwd", hipLaunchKernelGGL(skel_kernel<1>, dim3(frames), dim3(256), 0, 0, x, amp, frames));
  SK("skel x4 st via LDS row", hipLaunchKernelGGL(skel_kernel<2>, dim3(frames), dim3(256), 0, 0, x, amp, frames));
  SK("skel x4 nt st via LDS row", hipLaunchKernelGGL(skel_kernel<3>, dim3(frames), dim3(256), 0, 0, x, amp, frames));
  SK("skel dword ld (64/thread)", hipLaunchKernelGGL(skel_dword_kernel, dim3(frames), dim3(256), 0, 0, x, amp, frames));
  SK("skel dword ld + x4 nt st via LDS", hipLaunchKernelGGL((skel_kernel<3, 1>), dim3(frames), dim3(256), 0, 0, x, amp, frames));
  SK("skel x2 ld + x4 nt st via LDS", hipLaunchKernelGGL((skel_kernel<3, 2>), dim3(frames), dim3(256), 0, 0, x, amp, frames));
  SK("skel x2 ld + dword st", hipLaunchKernelGGL((skel_kernel<0, 2>), dim3(frames), dim3(256), 0, 0, x, amp, frames));
  SK("skel x2 ld + x2 st direct (DIF shape)", hipLaunchKernelGGL((skel_kernel<4, 2>), dim3(frames), dim3(256), 0, 0, x, amp, frames));
  SK("skel x2 ld + x2 nt st direct (DIF shape)", hipLaunchKernelGGL((skel_kernel<5, 2>), dim3(frames), dim3(256), 0, 0, x, amp, frames));
  SK("skel x4 ld + x2 st direct", hipLaunchKernelGGL((skel_kernel<4, 0>), dim3(frames), dim3(256), 0, 0, x, amp, frames));
  SK("skel x2 ld + x4 st via LDS (plain)", hipLaunchKernelGGL((skel_kernel<2, 2>), dim3(frames), dim3(256), 0, 0, x, amp, frames));
  SK("skel x4 ld wave-contiguous + x4 nt st", hipLaunchKernelGGL((skel_kernel<3, 3>), dim3(frames), dim3(256), 0, 0, x, amp, frames));
  SK("skel x4 ld wave-contiguous + dword st", hipLaunchKernelGGL((skel_kernel<0, 3>), dim3(frames), dim3(256), 0, 0, x, amp, frames));
  SK("skel LDS-DMA ld", hipLaunchKernelGGL(skel_dma_kernel, dim3(frames), dim3(256), 0, 0, x, amp, frames));
  SK("skel persistent 768 WGs", hipLaunchKernelGGL(skel_persist_kernel, dim3(768), dim3(256), 0, 0, x, amp, frames));
  const size_t n4 = (size_t)frames * N / 4 / 2;  // two input planes of n4 V4 = the whole frame buffer
  const double mb = 16.0 * 3 * n4;
  vs.push_back({"mix 2:1 x4 grid-stride, plain st, 2048 WGs", [&] { hipLaunchKernelGGL(mix21_kernel<false>, dim3(2048), dim3(256), 0, 0, (const V4 *)x, (V4 *)amp, n4); }, mb, {}});
  vs.push_back({"mix 2:1 x4 grid-stride, nt st, 2048 WGs", [&] { hipLaunchKernelGGL(mix21_kernel<true>, dim3(2048), dim3(256), 0, 0, (const V4 *)x, (V4 *)amp, n4); }, mb, {}});
  vs.push_back({"mix 2:1 x4, plain st, one pass (grid = n4/256)", [&] { hipLaunchKernelGGL(mix21_kernel<false>, dim3((unsigned)(n4 / 256)), dim3(256), 0, 0, (const V4 *)x, (V4 *)amp, n4); }, mb, {}});
  vs.push_back({"mix 2:1 x4, nt st, one pass (grid = n4/256)", [&] { hipLaunchKernelGGL(mix21_kernel<true>, dim3((unsigned)(n4 / 256)), dim3(256), 0, 0, (const V4 *)x, (V4 *)amp, n4); }, mb, {}});
  vs.push_back({"read-only x4 grid-stride 2048 WGs", [&] { hipLaunchKernelGGL(read_kernel, dim3(2048), dim3(256), 0, 0, (const V4 *)x, (V4 *)amp, 2 * n4); }, 16.0 * 2 * n4, {}});

  if (argc > 3 && std::string(argv[3]) == "c2c") {
    // planar C2C skeletons: rows = frames * 4 rows of 4096 points in each of two input planes
    vs.clear();
    const long long rows = frames * 2;  // x holds frames*16384 floats = two planes of rows*4096
    float *ore = amp, *oim;
    CK(hipMalloc(&oim, (size_t)rows * 4096 * 4));
    float *ore2;
    CK(hipMalloc(&ore2, (size_t)rows * 4096 * 4));
    ore = ore2;
    const float *re = x, *im = x + (size_t)rows * 4096;
    const double cb = 16.0 * rows * 4096;
#define C2(NAME, ...) vs.push_back({NAME, [=] { __VA_ARGS__; }, cb, {}})  /* by value: these locals end with the block */
    C2("c2c skel dword ld, dword nt st (shipped shape)", hipLaunchKernelGGL((c2c_skel_kernel<1, true>), dim3(rows), dim3(256), 0, 0, re, im, ore, oim));
    C2("c2c skel x2 ld, x2 nt st", hipLaunchKernelGGL((c2c_skel_kernel<2, true>), dim3(rows), dim3(256), 0, 0, re, im, ore, oim));
    C2("c2c skel x4 ld, x4 nt st", hipLaunchKernelGGL((c2c_skel_kernel<4, true>), dim3(rows), dim3(256), 0, 0, re, im, ore, oim));
    C2("c2c skel x2 ld, x2 plain st", hipLaunchKernelGGL((c2c_skel_kernel<2, false>), dim3(rows), dim3(256), 0, 0, re, im, ore, oim));
    C2("c2c skel dword ld, dword plain st", hipLaunchKernelGGL((c2c_skel_kernel<1, false>), dim3(rows), dim3(256), 0, 0, re, im, ore, oim));
    // the shipped shape with occupancy capped by a dynamic LDS allocation (bytes in flight per CU)
    for (int wgs : {1, 2, 3, 4, 6}) {
      const int ldsb = 160 * 1024 / wgs - 512;
      hipFuncSetAttribute(reinterpret_cast<const void *>(&c2c_skel_kernel<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      static std::string names[8];
      names[wgs] = "c2c skel dword/dword nt, " + std::to_string(wgs) + " workgroup(s) per CU (LDS cap)";
      vs.push_back({names[wgs], [=] { hipLaunchKernelGGL((c2c_skel_kernel<1, true>), dim3(rows), dim3(256), ldsb, 0, re, im, ore, oim); }, cb, {}});
    }
    const size_t c4 = (size_t)rows * 4096 / 4;
    C2("copy 1:1 x4 one element per thread, nt st (2 launches)", hipLaunchKernelGGL(copy11_kernel<true>, dim3((unsigned)(c4 / 256)), dim3(256), 0, 0, (const V4 *)re, (V4 *)ore, c4); hipLaunchKernelGGL(copy11_kernel<true>, dim3((unsigned)(c4 / 256)), dim3(256), 0, 0, (const V4 *)im, (V4 *)oim, c4));
    C2("copy 1:1 x4 one element per thread, plain st", hipLaunchKernelGGL(copy11_kernel<false>, dim3((unsigned)(c4 / 256)), dim3(256), 0, 0, (const V4 *)re, (V4 *)ore, c4); hipLaunchKernelGGL(copy11_kernel<false>, dim3((unsigned)(c4 / 256)), dim3(256), 0, 0, (const V4 *)im, (V4 *)oim, c4));
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w)
    for (auto &v : vs) v.run();
  CK(hipDeviceSynchronize());
  CK(hipGetLastError());
  for (int r = 0; r < rounds; ++r)
    for (auto &v : vs) {
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < 5; ++i) v.run();
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float t;
      CK(hipEventElapsedTime(&t, e0, e1));
      v.ms.push_back(t / 5);
    }
  CK(hipGetLastError());
  printf("%-52s %9s %9s %9s %9s\n", "variant", "med_ms", "min_ms", "med_GB/s", "max_GB/s");
  for (auto &v : vs) {
    std::sort(v.ms.begin(), v.ms.end());
    const float med = v.ms[v.ms.size() / 2], mn = v.ms[0];
    printf("%-52s %9.4f %9.4f %9.0f %9.0f\n", v.name.c_str(), med, mn, v.bytes / med / 1e6, v.bytes / mn / 1e6);
  }
  return 0;
}

// tools/kbench3.hip -- what can a COLUMN tile pass reach?  (development tool, not product)
// The two-pass transforms read their first pass strided: a 256-thread workgroup owns a tile of W columns of a
// [rows][PITCH] matrix (two planes, like the planar complex rows), i.e. rows x (W*4)-byte segments PITCH*4 bytes
// apart, 64 KB per workgroup.  This skeleton moves such tiles with no transform in between -- all loads issued
// first, 16 bytes per lane, non-temporal, then the stores -- in the four combinations of strided / contiguous
// reads and writes, by segment width and by workgroups per CU (a dummy LDS allocation sets the occupancy).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/kbench3.hip -o tools/kbench3 && tools/kbench3
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e = (x);                                                             \
    if (e != hipSuccess) {                                                          \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
      exit(1);                                                                      \
    }                                                                               \
  } while (0)

typedef float V4 __attribute__((ext_vector_type(4)));

// One workgroup = 8192 floats per plane (32 KB), W columns wide: R = 8192 / W rows.
// RS / WS: 1 = strided tile (row r of the tile at r*pitch + tile*W), 0 = one contiguous 32 KB chunk.
template <int W, bool RS, bool WS>
__global__ void __launch_bounds__(256)
tile_move(const float *__restrict__ in_re, const float *__restrict__ in_im, float *__restrict__ out_re,
          float *__restrict__ out_im, const int pitch, const int tiles_per_row, const int lds_bytes,
          const int spread = 1) {
  extern __shared__ float dummy[];
  constexpr int TS = W / 4, SPI = 256 / TS, R = 8192 / W, NIT = R / SPI;
  const int t = (int)threadIdx.x;
  const size_t blk = blockIdx.x;
  const size_t band = blk / tiles_per_row, tile = blk % tiles_per_row;  // a band = R rows of the matrix
  const size_t sbase = band * (size_t)R * (size_t)pitch + tile * W;       // strided tile origin
  // contiguous chunk origin; spread S > 1: workgroups b, b+1, ... take chunks gridDim/S apart (S concurrent fronts)
  const size_t cbase = (spread > 1 ? (blk % spread) * (gridDim.x / spread) + blk / spread : blk) * 8192;
  const int seg = t / TS, j4 = (t % TS) * 4;
  V4 r[NIT], m[NIT];
#pragma unroll
  for (int ic = 0; ic < NIT; ++ic) {
    const size_t gi = RS ? sbase + (size_t)(seg + SPI * ic) * (size_t)pitch + j4 : cbase + 4 * (t + 256 * ic);
    r[ic] = __builtin_nontemporal_load(reinterpret_cast<const V4 *>(in_re + gi));
    m[ic] = __builtin_nontemporal_load(reinterpret_cast<const V4 *>(in_im + gi));
  }
  if (lds_bytes < 0) dummy[t] = r[0].x;  // never: keeps the allocation
#pragma unroll
  for (int ic = 0; ic < NIT; ++ic) {
    const size_t go = WS ? sbase + (size_t)(seg + SPI * ic) * (size_t)pitch + j4 : cbase + 4 * (t + 256 * ic);
    __builtin_nontemporal_store(r[ic] + m[ic], reinterpret_cast<V4 *>(out_re + go));
    __builtin_nontemporal_store(r[ic] - m[ic], reinterpret_cast<V4 *>(out_im + go));
  }
}

template <int W, bool RS, bool WS>
double run(const float *a, const float *b, float *c, float *d, size_t floats, int pitch, int lds, int rounds) {
  const int blocks = (int)(floats / 8192);
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&tile_move<W, RS, WS>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i)
    hipLaunchKernelGGL((tile_move<W, RS, WS>), dim3(blocks), dim3(256), lds, 0, a, b, c, d, pitch, pitch / W, lds);
  CK(hipEventRecord(e0));
  for (int i = 0; i < rounds; ++i)
    hipLaunchKernelGGL((tile_move<W, RS, WS>), dim3(blocks), dim3(256), lds, 0, a, b, c, d, pitch, pitch / W, lds);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGetLastError());
  return 16.0 * (double)floats / (ms / rounds * 1e-3) / 1e12;  // TB/s: 8 B read + 8 B written per float pair
}

// contiguous 32 KB chunks per plane (the shape of a row kernel's workgroup), consecutive workgroups `spread` fronts apart
double run_spread(const float *a, const float *b, float *c, float *d, size_t floats, int spread, int lds, int rounds) {
  const int blocks = (int)(floats / 8192);
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&tile_move<256, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i)
    hipLaunchKernelGGL((tile_move<256, false, false>), dim3(blocks), dim3(256), lds, 0, a, b, c, d, 256, 1, lds, spread);
  CK(hipEventRecord(e0));
  for (int i = 0; i < rounds; ++i)
    hipLaunchKernelGGL((tile_move<256, false, false>), dim3(blocks), dim3(256), lds, 0, a, b, c, d, 256, 1, lds, spread);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return 16.0 * (double)floats / (ms / rounds * 1e-3) / 1e12;
}

int main(int argc, char **argv) {
  const size_t floats = (size_t)1 << 27;  // per plane: 512 MiB, 2 GiB of traffic per launch
  const int rounds = argc > 1 ? atoi(argv[1]) : 10;
  float *a, *b, *c, *d;
  CK(hipMalloc(&a, floats * 4));
  CK(hipMalloc(&b, floats * 4));
  CK(hipMalloc(&c, floats * 4));
  CK(hipMalloc(&d, floats * 4));
  CK(hipMemset(a, 0, floats * 4));
  CK(hipMemset(b, 0, floats * 4));
  printf("contiguous 32 KB chunks, consecutive workgroups on S fronts gridDim/S chunks apart (TB/s), 4 workgroups per CU\n");
  for (int rep = 0; rep < 2; ++rep)
    for (int spread : {1, 2, 8, 16, 64, 256, 1024, 4096, 16384})
      printf("  S = %5d  %6.2f\n", spread, run_spread(a, b, c, d, floats, spread, 36 * 1024, rounds));
  if (argc > 2) return 0;
  printf("TB/s of traffic, 2^27 complex points per launch; rows of a tile are `pitch` floats apart\n");
  printf("%6s %6s %9s | %8s %8s %8s %8s\n", "pitch", "wg/CU", "segment", "rs+ws", "rs+wc", "rc+ws", "rc+wc");
  for (int pitch : {256, 4096, 65536}) {
    for (int lds : {36 * 1024, 70 * 1024, 16 * 1024}) {  // 4, 2 and 8+ workgroups per CU
#define ROW(W)                                                                                        \
  printf("%6d %6d %7d B | %8.2f %8.2f %8.2f %8.2f\n", pitch, 160 * 1024 / lds > 8 ? 8 : 160 * 1024 / lds, W * 4, \
         run<W, true, true>(a, b, c, d, floats, pitch, lds, rounds), run<W, true, false>(a, b, c, d, floats, pitch, lds, rounds), \
         run<W, false, true>(a, b, c, d, floats, pitch, lds, rounds), run<W, false, false>(a, b, c, d, floats, pitch, lds, rounds)); \
  fflush(stdout)
      ROW(16);
      ROW(32);
      ROW(64);
      ROW(128);
      ROW(256);
#undef ROW
    }
  }
  return 0;
}

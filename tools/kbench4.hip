// tools/kbench4.hip -- what would a RESIDENT kernel buy a one-frame call?  (development tool, not product)
// a one-frame FFT.forward through the drop-in costs one launch + one completion (~10 us on this
// stack) before any arithmetic.  This measures the alternative's floor: one 64-thread workgroup stays on the card
// and polls a doorbell word in pinned host memory; the host writes a request (a sequence number, 8 KB of input
// already in pinned memory), the kernel copies the 8 KB input to an 16 KB output (standing in for the N = 1024
// transform: two LDS passes, ~1 us) and acknowledges with the sequence number; the host spins on the ack.
// Reported: host-to-host round trip per request, and for comparison an empty kernel's launch + synchronize.
// Every wave reaches an exit: the loop ends on the quit flag, or after a bounded number of polls.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/kbench4.hip -o tools/kbench4 && tools/kbench4
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e = (x);                                                             \
    if (e != hipSuccess) {                                                          \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
      exit(1);                                                                      \
    }                                                                               \
  } while (0)

struct Mailbox {
  volatile unsigned req;   // host -> card: sequence number of the newest request
  volatile unsigned quit;  // host -> card
  unsigned pad0[14];
  volatile unsigned ack;   // card -> host: sequence number of the last request served
  volatile unsigned alive; // card -> host: 1 while the kernel runs, 2 when it has left
  unsigned pad1[14];
};

__global__ void __launch_bounds__(64) resident(Mailbox *mb, const float *in, float *out, const unsigned long long max_polls) {
  const int t = (int)threadIdx.x;
  unsigned served = 0;
  if (t == 0) __hip_atomic_store(&mb->alive, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  for (unsigned long long polls = 0; polls < max_polls; ++polls) {
    unsigned want = 0, q = 0;
    if (t == 0) {
      want = __hip_atomic_load(&mb->req, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
      q = __hip_atomic_load(&mb->quit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    want = __builtin_amdgcn_readfirstlane(want);
    q = __builtin_amdgcn_readfirstlane(q);
    if (q) break;
    if (want == served) continue;
    // "the transform": 2048 floats in (N = 1024 f64-as-2-floats or a complex f32 row), 4096 floats out
    float4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = reinterpret_cast<const float4 *>(in)[t + 64 * i];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      reinterpret_cast<float4 *>(out)[t + 64 * i] = v[i];
      reinterpret_cast<float4 *>(out)[512 + t + 64 * i] = v[i];
    }
    __threadfence_system();
    __syncthreads();
    served = want;
    if (t == 0) __hip_atomic_store(&mb->ack, served, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (t == 0) __hip_atomic_store(&mb->alive, 2u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void empty_kernel(float *out) {
  if (threadIdx.x == 1024) out[0] = 1.f;
}

static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
  Mailbox *mb;
  float *in, *out;
  CK(hipHostMalloc((void **)&mb, sizeof(Mailbox), hipHostMallocDefault));
  CK(hipHostMalloc((void **)&in, 2048 * 4, hipHostMallocDefault));
  CK(hipHostMalloc((void **)&out, 4096 * 4, hipHostMallocDefault));
  *mb = Mailbox{};
  for (int i = 0; i < 2048; ++i) in[i] = (float)i;
  Mailbox *dmb;
  float *din, *dout;
  CK(hipHostGetDevicePointer((void **)&dmb, mb, 0));
  CK(hipHostGetDevicePointer((void **)&din, in, 0));
  CK(hipHostGetDevicePointer((void **)&dout, out, 0));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));

  // launch + synchronize of an empty kernel (what every one-frame call pays today)
  std::vector<double> ls;
  for (int i = 0; i < 2200; ++i) {
    const double t0 = now_us();
    hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s, dout);
    CK(hipStreamSynchronize(s));
    if (i >= 200) ls.push_back(now_us() - t0);
  }
  std::sort(ls.begin(), ls.end());
  printf("empty kernel, launch + hipStreamSynchronize: median %.2f us, p10 %.2f, p90 %.2f\n", ls[ls.size() / 2],
         ls[ls.size() / 10], ls[ls.size() * 9 / 10]);

  // the resident kernel: bounded at 2e7 polls (~30 s at most), ended by the quit flag
  hipLaunchKernelGGL(resident, dim3(1), dim3(64), 0, s, dmb, din, dout, 20000000ULL);
  CK(hipGetLastError());
  double t_wait = now_us();
  while (mb->alive == 0 && now_us() - t_wait < 5e6) {
  }
  if (mb->alive == 0) {
    printf("resident kernel did not start\n");
    mb->quit = 1;
    CK(hipStreamSynchronize(s));
    return 1;
  }
  std::vector<double> rt;
  unsigned seq = 0;
  bool ok = true;
  for (int i = 0; i < 20200 && ok; ++i) {
    in[0] = (float)i;
    const double t0 = now_us();
    __atomic_store_n(&mb->req, ++seq, __ATOMIC_RELEASE);
    while (__atomic_load_n(&mb->ack, __ATOMIC_ACQUIRE) != seq) {
      if (now_us() - t0 > 2e6) {
        ok = false;
        break;
      }
    }
    const double dt = now_us() - t0;
    if (ok && (out[0] != (float)i || out[2048] != (float)i)) {
      printf("stale output at request %d: %g %g\n", i, out[0], out[2048]);
      ok = false;
    }
    if (i >= 200) rt.push_back(dt);
  }
  mb->quit = 1;
  CK(hipStreamSynchronize(s));
  if (!ok || rt.empty()) {
    printf("resident round trip FAILED (timeout or stale data)\n");
    return 1;
  }
  std::sort(rt.begin(), rt.end());
  printf("resident kernel, doorbell -> 8 KB in, 16 KB out over PCIe -> ack: median %.2f us, p10 %.2f, p90 %.2f, max %.1f (n = %zu)\n",
         rt[rt.size() / 2], rt[rt.size() / 10], rt[rt.size() * 9 / 10], rt.back(), rt.size());
  printf("kernel left cleanly: alive = %u\n", mb->alive);
  return 0;
}

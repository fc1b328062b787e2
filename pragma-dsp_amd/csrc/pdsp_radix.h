// pdsp_radix.h -- the pass decomposition shared by the host twiddle builder and
// the device kernels (both index the same table, so both derive it from here).
//
// A length-N transform (N = 2^L) is run by TP = N/E cooperating threads, each
// holding E points in registers.  It is a Stockham autosort factorisation
// N = R0*R1*...: pass p combines radix-Rp butterflies (each one is log2(Rp)
// radix-2 stages of the reference's loop nest, src/core/fft.ts:116-140, done
// in registers), and the autosort addressing absorbs the bit-reversal scatter
// of src/core/fft.ts:110-114.
#pragma once

namespace pdsp {

constexpr int kMaxPasses = 5;
constexpr int kMaxLog2N_f32 = 14;  // (N + N/16) * 8 B of LDS <= 160 KiB
constexpr int kMaxLog2N_f64 = 13;  // (N + N/16) * 16 B
constexpr int kMaxLog2N1 = 4;      // four-step path: N = N1 * N2, N1 <= 16 columns per thread
// general four-step path (N1 > 16: both factors run on the single-pass row kernels)
constexpr int kMaxLog2Big_f32 = 2 * kMaxLog2N_f32;  // 2^28
constexpr int kMaxLog2Big_f64 = 2 * kMaxLog2N_f64;  // 2^26

struct RadixPlan {
  int log2n;
  int n;
  int e;               // points per thread
  int tp;              // threads per transform
  int np;              // passes
  int r[kMaxPasses];   // radix of each pass
  int ns[kMaxPasses];  // product of the earlier radices (sub-transform length so far)
  int twoff[kMaxPasses];  // offset of the pass's twiddle block in the table
  int twcount;         // total table entries (complex)
};

// log2e: points per thread (4 = sixteen, the default; 3 = eight: half the registers, twice the
// waves, one more LDS pass -- what the VALU-issue-bound packed-real kernel at M = 8192 wants).
constexpr RadixPlan make_radix_plan(int log2n, int log2e = 4) {
  RadixPlan p{};
  p.log2n = log2n;
  p.n = 1 << log2n;
  if (log2n <= log2e) {
    p.e = p.n;
    p.tp = 1;
    p.np = log2n == 0 ? 0 : 1;
    p.r[0] = p.n;
  } else {
    p.e = 1 << log2e;
    p.tp = p.n >> log2e;
    const int full = log2n / log2e, rem = log2n % log2e;
    p.np = full + (rem ? 1 : 0);
    for (int i = 0; i < full; ++i) p.r[i] = 1 << log2e;
    if (rem) p.r[full] = 1 << rem;
  }
  int ns = 1, off = 0;
  for (int i = 0; i < p.np; ++i) {
    p.ns[i] = ns;
    p.twoff[i] = off;
    // pass i multiplies input r (1 <= r < R) of the butterfly at position k
    // (0 <= k < Ns) by W_{Ns*R}^{r*k}; stored as tw[off + (r-1)*Ns + k].
    // The first pass has Ns = 1 (all twiddles are 1) and stores nothing.
    if (ns > 1) off += (p.r[i] - 1) * ns;
    ns *= p.r[i];
  }
  p.twcount = off;
  return p;
}

// Points per thread of the packed-real spectrum kernel, by log2 of its N/2-point transform.
#ifndef PDSP_PACKED_LOG2E_13
#define PDSP_PACKED_LOG2E_13 4  /* 3 (eight points per thread, 8 waves/SIMD) measured 6 % slower: tools/kbench */
#endif
constexpr int packed_log2e(int log2m) { return log2m == 13 ? PDSP_PACKED_LOG2E_13 : 4; }

}  // namespace pdsp

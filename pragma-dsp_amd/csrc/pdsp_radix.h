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

constexpr int kMaxPasses = 4;
constexpr int kMaxLog2N_f32 = 14;  // (N + N/16) * 8 B of LDS <= 160 KiB
constexpr int kMaxLog2N_f64 = 13;  // (N + N/16) * 16 B
constexpr int kMaxLog2N1 = 4;      // four-step path: N = N1 * N2, N1 <= 16 columns per thread

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

constexpr RadixPlan make_radix_plan(int log2n) {
  RadixPlan p{};
  p.log2n = log2n;
  p.n = 1 << log2n;
  if (log2n <= 4) {
    p.e = p.n;
    p.tp = 1;
    p.np = log2n == 0 ? 0 : 1;
    p.r[0] = p.n;
  } else {
    p.e = 16;
    p.tp = p.n / 16;
    const int full = log2n / 4, rem = log2n % 4;
    p.np = full + (rem ? 1 : 0);
    for (int i = 0; i < full; ++i) p.r[i] = 16;
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

}  // namespace pdsp

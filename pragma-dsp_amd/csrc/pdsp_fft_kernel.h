// pdsp_fft_kernel.h -- single-pass, LDS-resident Stockham autosort FFT for gfx950.
//
// What it replaces: the body of Radix2Fft.transform, src/core/fft.ts:89-151 of
// pragma-dsp (bit-reversal scatter + log2 N radix-2 sweeps over memory + the
// inverse 1/N sweep), applied to `batch` independent rows, with the caller's
// pre/post element-wise steps (applyWindow, magnitude, phase, amplitude scaling:
// src/xform/fourier.ts:54-120, src/public/spectrum.ts:45-72) folded into the
// first load and the last store.  HBM is touched once each way; every butterfly
// stage runs out of registers and LDS.
//
// Shape: a transform is owned by TP = N/E threads with E points each in VGPRs.
// Pass p does radix-Rp butterflies in registers (log2 Rp radix-2 stages with
// compile-time constant twiddles), multiplies by the inter-pass twiddles
// W_{Ns*Rp}^{r*k} from a host-built f64->f32 table, and hands the points to the
// next pass through LDS in autosort order.  Global loads and stores are always
// `row*N + tid + TP*q` -- unit stride across the lanes of a wave -- so the
// bit-reversed scatter of the reference never happens in memory.
//
// No MFMA: an FFT is not a dense contraction; the kernel is bound by HBM
// (16 B/sample at 3.75 flop/B), not by VALU.
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>
#include <utility>

#include "pdsp_radix.h"

namespace pdsp {

// Development-only ablation switches (tools/kbench): 0 in every product build.
//   1 = inter-pass twiddles from a constant instead of the table
//   2 = no window load      4 = no Hermitian split (store |Z| of the half-size transform)
#ifndef PDSP_EXPERIMENT
#define PDSP_EXPERIMENT 0
#endif
constexpr int kExp = PDSP_EXPERIMENT;

template <typename T> struct vec2;
template <> struct vec2<float> { using type = float2; };
template <> struct vec2<double> { using type = double2; };

template <int... Is, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F &&f) {
  (f(std::integral_constant<int, Is>{}), ...);
}
// Fully unrolled loop whose index is a compile-time constant (register arrays
// must never be indexed at run time: they would go to scratch).
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
  static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F &&>(f));
}

constexpr int ilog2(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}
constexpr int bitrev(int x, int bits) {
  int y = 0;
  for (int b = 0; b < bits; ++b) {
    y = (y << 1) | (x & 1);
    x >>= 1;
  }
  return y;
}

// (re + i*im) *= W16^M,  W16 = e^{-2*pi*i/16},  0 <= M < 8.
template <typename T, int M>
__device__ __forceinline__ void mul_w16(T &re, T &im) {
  constexpr T C1 = T(0.92387953251128673848);  // cos(pi/8)
  constexpr T C2 = T(0.70710678118654752440);  // cos(pi/4)
  constexpr T C3 = T(0.38268343236508977173);  // sin(pi/8)
  if constexpr (M == 0) {
  } else if constexpr (M == 4) {  // -i
    const T t = re;
    re = im;
    im = -t;
  } else if constexpr (M == 2) {  // (1 - i)/sqrt2
    const T a = re, b = im;
    re = (a + b) * C2;
    im = (b - a) * C2;
  } else if constexpr (M == 6) {  // (-1 - i)/sqrt2
    const T a = re, b = im;
    re = (b - a) * C2;
    im = -(a + b) * C2;
  } else {
    constexpr T c = M == 1 ? C1 : M == 3 ? C3 : M == 5 ? -C3 : -C1;
    constexpr T s = M == 1 ? -C3 : M == 3 ? -C1 : M == 5 ? -C1 : -C3;
    const T a = re, b = im;
    re = a * c - b * s;
    im = a * s + b * c;
  }
}

// In-register radix-R DFT (R = 2, 4, 8, 16) as log2 R decimation-in-frequency
// radix-2 stages.  Output k ends up in slot bitrev(k).
template <typename T, int R>
__device__ __forceinline__ void fft_reg(T (&ar)[R], T (&ai)[R]) {
  static_assert(R >= 1 && R <= 16 && (R & (R - 1)) == 0, "radix");
  static_for<ilog2(R)>([&](auto stc) {
    constexpr int s = R >> (stc + 1);  // half length of this stage's sub-transforms
    static_for<R / 2>([&](auto ic) {
      constexpr int g = (ic / s) * 2 * s, k = ic % s;
      constexpr int i0 = g + k, i1 = i0 + s;
      const T ur = ar[i0], ui = ai[i0], vr = ar[i1], vi = ai[i1];
      ar[i0] = ur + vr;
      ai[i0] = ui + vi;
      T dr = ur - vr, di = ui - vi;
      mul_w16<T, k *(8 / s)>(dr, di);  // W_{2s}^k
      ar[i1] = dr;
      ai[i1] = di;
    });
  });
}

template <int LOG2N>
struct FftTraits {
  static constexpr RadixPlan P = make_radix_plan(LOG2N);
  static constexpr int N = P.n, E = P.e, TP = P.tp, NP = P.np;
  static constexpr int WG = TP >= 256 ? TP : 256;  // threads per workgroup
  static constexpr int ROWS = WG / TP;             // transforms per workgroup
  // one pad element every 16: the radix-16 scatter (stride 16 complex) would
  // otherwise put a whole ds_write lane group on one bank
  static constexpr int LROW = N + N / 16;
  static constexpr int LDS_ELEMS = NP > 1 ? ROWS * LROW : 1;
};

__device__ __forceinline__ int lds_pad(int i) { return i + (i >> 4); }
// pad(a + c) == pad(a) + cpad(c) when c is a multiple of 16 (or a is and c < 16)
constexpr int cpad(int c) { return c + c / 16; }

// A transform owned by >= 64 threads has one row per wave: tell the compiler the row
// is wave-uniform so row bases live in SGPRs (batch < 2^31 is checked on the host).
template <int TP>
__device__ __forceinline__ long long uniform_row(long long row) {
  if constexpr (TP >= 64) return (long long)__builtin_amdgcn_readfirstlane((int)row);
  else return row;
}

// Streamed rows are touched exactly once: non-temporal loads/stores keep them from
// displacing the twiddle/window tables in L2 (+4..10 % on the row-pattern copy and on
// the C2C kernel, tools/kbench).
template <typename T>
__device__ __forceinline__ T ld_stream(const T *p) { return __builtin_nontemporal_load(p); }
template <typename T>
__device__ __forceinline__ void st_stream(T v, T *p) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ float2 ld_stream2(const float2 *p) {
  const double d = __builtin_nontemporal_load(reinterpret_cast<const double *>(p));  // one 8-byte nt load
  float2 r;
  __builtin_memcpy(&r, &d, sizeof(r));
  return r;
}

// |z| for the fused amplitude stores: one v_sqrt_f32 (1 ulp) instead of the ~10-instruction
// correctly rounded sequence -- 17 of them per thread sit in the VALU-bound epilogue of the
// N=16384 spectrum kernel.  1 ulp = 6e-8 relative, far inside the 1e-5 contract.
__device__ __forceinline__ float mag2(float re, float im) { return __builtin_amdgcn_sqrtf(re * re + im * im); }
__device__ __forceinline__ double mag2(double re, double im) { return sqrt(re * re + im * im); }

// ---- load / store policies ------------------------------------------------
// ld(row, off, lane, re, im): fetch point off + lane of row `row` (row < batch).
// st(row, off, lane, re, im): write point off + lane.
// `row` and `off` are wave-uniform whenever a transform spans whole waves, so
// `base + row*N + off` stays in SGPRs and the only per-lane address register is
// `lane` (global_load ... v_lane, s[base:base+1] offset:imm).

template <typename T>
struct LoadComplex {  // forwardComplex / inverse (planes swapped by the caller)
  const T *__restrict__ re;
  const T *__restrict__ im;
  long long n;
  __device__ __forceinline__ void operator()(long long row, int off, int lane, T &a, T &b) const {
    const size_t o = (size_t)row * (size_t)n + (size_t)off;
    a = ld_stream(re + o + (unsigned)lane);
    b = ld_stream(im + o + (unsigned)lane);
  }
};

template <typename T>
struct LoadReal {  // Radix2Fft.forward: imaginary part is zero
  const T *__restrict__ re;
  long long n;
  __device__ __forceinline__ void operator()(long long row, int off, int lane, T &a, T &b) const {
    a = ld_stream(re + (size_t)row * (size_t)n + (size_t)off + (unsigned)lane);
    b = T(0);
  }
};

// Loads are unconditional (clamped address + select): a per-element `if` around a
// load makes hipcc branch and drain vmcnt per element (cdna guide, section 5 item 4c).
template <typename T, bool HAS_WIN>
struct LoadFrameWindowed {  // buildFrame + applyWindow, spectrum.ts:36-43, :116-119
  const T *__restrict__ x;
  const T *__restrict__ win;  // N values when HAS_WIN (rect otherwise)
  long long frame_len;        // 1 <= samples used per row <= N; the rest reads as zero
  long long stride;
  __device__ __forceinline__ void operator()(long long row, int off, int lane, T &a, T &b) const {
    const int i = off + lane;
    const int last = (int)frame_len - 1;
    T v = ld_stream(x + (size_t)row * (size_t)stride + (unsigned)(i < last ? i : last));
    v = i <= last ? v : T(0);
    if constexpr (HAS_WIN) v *= win[i];
    a = v;
    b = T(0);
  }
};

template <typename T>
struct StoreComplex {
  T *__restrict__ re;
  T *__restrict__ im;
  long long n;
  T scale;  // 1 forward, 1/N inverse (fft.ts:142-148); power of two => exact
  __device__ __forceinline__ void operator()(long long row, int off, int lane, T a, T b) const {
    const size_t o = (size_t)row * (size_t)n + (size_t)off;
    st_stream(a * scale, re + o + (unsigned)lane);
    st_stream(b * scale, im + o + (unsigned)lane);
  }
};

template <typename T>
struct StoreAmplitude {  // magnitude + scaleAmplitude{One,Two}Sided [+ phase]
  T *__restrict__ amp;
  T *__restrict__ ph;  // may be null
  int bins;            // N/2+1 or N
  int nyq;             // N/2 for one-sided (that bin is not doubled), -1 for two-sided
  T s_edge;            // 1/N
  T s_mid;             // 2/N one-sided, 1/N two-sided
  __device__ __forceinline__ void operator()(long long row, int off, int lane, T a, T b) const {
    const int i = off + lane;
    if (i < bins) {
      const size_t o = (size_t)row * (size_t)bins;
      const T m = mag2(a, b);
      st_stream(m * ((i == 0 || i == nyq) ? s_edge : s_mid), amp + o + (unsigned)i);
      if (ph) st_stream(T(atan2(b, a)), ph + o + (unsigned)i);
    }
  }
};

// ---- twiddle providers ---------------------------------------------------------

// Reads W_{Ns*R}^{r*k} from the host-built table (layout: pdsp_radix.h), every use.
template <typename T, int LOG2N>
struct TableTwiddles {
  const typename vec2<T>::type *__restrict__ tw;
  template <int p, int r, int b>
  __device__ __forceinline__ typename vec2<T>::type get(const int j) const {
    constexpr int Ns = FftTraits<LOG2N>::P.ns[p];
    return tw[FftTraits<LOG2N>::P.twoff[p] + (r - 1) * Ns + (j & (Ns - 1))];
  }
};

// ---- the passes --------------------------------------------------------------

// Runs every pass of the length-2^LOG2N transform on the E points each of the TP
// cooperating threads holds.  LAST_TO_LDS = false: the result comes back in the
// registers, X[tid + TP*q] in slot q.  LAST_TO_LDS = true: the last pass also
// scatters to LDS, in natural order (X[k] at lds_pad(k)), for a consumer that needs
// other threads' bins; the caller must __syncthreads() before reading it.
//
// TWF supplies the inter-pass twiddles: twf.template get<p, r, b>(j) = W_{Ns*R}^{r*(j mod Ns)}
// for input r of butterfly j = tid + b*TP of pass p.
template <typename T, int LOG2N, bool LAST_TO_LDS, class TWF>
__device__ __forceinline__ void fft_passes(T (&xr)[FftTraits<LOG2N>::E], T (&xi)[FftTraits<LOG2N>::E],
                                           typename vec2<T>::type *const lrow, const TWF &twf, const int tid) {
  using TR = FftTraits<LOG2N>;
  using T2 = typename vec2<T>::type;
  constexpr int E = TR::E, TP = TR::TP, NP = TR::NP;
  // For N >= 256 every LDS address is (a thread-only base) + (a compile-time offset),
  // so there is one address register per pass instead of one per element.
  constexpr bool kConstOffsets = (TP % 16 == 0);

  static_for<NP>([&](auto pc) {
    constexpr int p = pc;
    constexpr int R = TR::P.r[p], Ns = TR::P.ns[p], EB = E / R, LR = ilog2(R);
    constexpr bool last = (p == NP - 1);
    constexpr bool to_lds = !last || LAST_TO_LDS;

    static_for<EB>([&](auto bc) {
      constexpr int b = bc;
      T ar[R], ai[R];
      static_for<R>([&](auto rc) {
        ar[rc] = xr[b + rc * EB];
        ai[rc] = xi[b + rc * EB];
      });
      const int j = tid + b * TP;  // butterfly index within the pass, 0 <= j < N/R
      if constexpr (Ns > 1) {
        static_for<R - 1>([&](auto rc) {
          constexpr int r = rc + 1;
          T2 w;
          if constexpr (kExp & 1) {
            w.x = T(0.999) + T(r) * T(1e-4);
            w.y = T(0.03);
          } else {
            w = twf.template get<p, r, b>(j);
          }
          const T a = ar[r], c = ai[r];
          ar[r] = a * w.x - c * w.y;
          ai[r] = a * w.y + c * w.x;
        });
      }
      fft_reg<T, R>(ar, ai);
      if constexpr (!to_lds) {
        // Ns*R == N: output r of butterfly j is X[j + r*N/R] = slot b + r*EB
        static_for<R>([&](auto rc) {
          xr[b + rc * EB] = ar[bitrev(rc, LR)];
          xi[b + rc * EB] = ai[bitrev(rc, LR)];
        });
      } else {
        // autosort scatter; for the last pass (Ns*R == N) this is the natural order
        if constexpr (kConstOffsets && (Ns % 16 == 0 || (Ns == 1 && R == 16))) {
          // j = tid + b*TP: the part of the index that depends on b and rc is constant
          constexpr int cb = Ns <= TP ? b * TP * R : b * TP;
          const int j0t = Ns <= TP ? ((tid >> ilog2(Ns)) << ilog2(Ns * R)) + (tid & (Ns - 1)) : tid;
          T2 *const wbase = lrow + lds_pad(j0t);
          static_for<R>([&](auto rc) {
            T2 v;
            v.x = ar[bitrev(rc, LR)];
            v.y = ai[bitrev(rc, LR)];
            wbase[Ns == 1 ? cpad(cb) + rc : cpad(cb + rc * Ns)] = v;
          });
        } else {
          const int j0 = ((j >> ilog2(Ns)) << ilog2(Ns * R)) + (j & (Ns - 1));
          static_for<R>([&](auto rc) {
            T2 v;
            v.x = ar[bitrev(rc, LR)];
            v.y = ai[bitrev(rc, LR)];
            lrow[lds_pad(j0 + rc * Ns)] = v;
          });
        }
      }
    });

    if constexpr (!last) {
      __syncthreads();
      if constexpr (kConstOffsets) {
        const T2 *const rbase = lrow + lds_pad(tid);
        static_for<E>([&](auto q) {
          const T2 v = rbase[cpad(TP * q)];
          xr[q] = v.x;
          xi[q] = v.y;
        });
      } else {
        static_for<E>([&](auto q) {
          const T2 v = lrow[lds_pad(tid + TP * q)];
          xr[q] = v.x;
          xi[q] = v.y;
        });
      }
      // the next pass writes LDS again (every pass but a register-resident last one)
      if constexpr (p + 1 < NP - 1 || LAST_TO_LDS) __syncthreads();
    }
  });
}

// ---- the kernels ---------------------------------------------------------------

template <typename T, int LOG2N, class LD, class ST>
__global__ void __launch_bounds__(FftTraits<LOG2N>::WG)
fft_stockham_kernel(const LD ld, const ST st, const typename vec2<T>::type *__restrict__ tw,
                    const long long batch) {
  using TR = FftTraits<LOG2N>;
  using T2 = typename vec2<T>::type;
  constexpr int E = TR::E, TP = TR::TP, NP = TR::NP;

  __shared__ T2 lds[TR::LDS_ELEMS];

  const int tid = TP == 1 ? 0 : (int)(threadIdx.x % TP);
  const int rloc = (int)(threadIdx.x / TP);
  const long long row_raw = (long long)blockIdx.x * TR::ROWS + rloc;
  const bool live = row_raw < batch;
  // dead rows of the last workgroup recompute the last live row and skip the
  // store, so that every thread reaches every barrier without predicated loads
  const long long row = uniform_row<TP>(live ? row_raw : batch - 1);
  T2 *const lrow = lds + (NP > 1 ? rloc * TR::LROW : 0);

  T xr[E], xi[E];
  static_for<E>([&](auto q) { ld(row, TP * q, tid, xr[q], xi[q]); });
  fft_passes<T, LOG2N, false>(xr, xi, lrow, TableTwiddles<T, LOG2N>{tw}, tid);
  if (live) {
    static_for<E>([&](auto q) { st(row, TP * q, tid, xr[q], xi[q]); });
  }
}

// Fused body of spectrum() for real frames, one frame per row, via the packed-real
// identity: z[m] = x[2m] + i*x[2m+1] (a plain float2 view of the windowed frame),
// Z = FFT_M(z) with M = N/2, then for each pair (k, M-k)
//     E = (Z[k] + conj Z[M-k]) / 2,  O = (Z[k] - conj Z[M-k]) / (2i),  t = W_N^k * O,
//     X[k] = E + t,   X[M-k] = conj(E - t),          (k = 0 also yields X[M] = Nyquist)
// and only |X| (scaled as scaleAmplitude{One,Two}Sided, spectrum.ts:45-72) and
// optionally atan2 leave the chip.  Half the butterflies, half the LDS and half the
// loads per thread of running the complex kernel on (x, 0).
//   LOG2M = log2(N/2) >= 5.   twr[k] = e^{-2*pi*i*k/N}, 0 <= k <= M/2.
template <typename T, int LOG2M, bool VEC2, bool HAS_WIN>
__global__ void __launch_bounds__(FftTraits<LOG2M>::WG)
spectrum_packed_kernel(const T *__restrict__ frames, const T *__restrict__ win, const long long frame_len,
                       const long long stride, const typename vec2<T>::type *__restrict__ tw,
                       const typename vec2<T>::type *__restrict__ twr, T *__restrict__ amp,
                       T *__restrict__ ph, const int two_sided, const T s_edge, const T s_mid,
                       const long long batch) {
  using TR = FftTraits<LOG2M>;
  using T2 = typename vec2<T>::type;
  constexpr int E = TR::E, TP = TR::TP, M = TR::N;
  static_assert(LOG2M >= 5, "packed path needs TP >= 2");

  __shared__ T2 lds[TR::LDS_ELEMS];

  const int tid = (int)(threadIdx.x % TP);
  const int rloc = (int)(threadIdx.x / TP);
  const long long row_raw = (long long)blockIdx.x * TR::ROWS + rloc;
  const bool live = row_raw < batch;
  const long long row = uniform_row<TP>(live ? row_raw : batch - 1);
  T2 *const lrow = lds + rloc * TR::LROW;

  // buildFrame + applyWindow (spectrum.ts:36-43, :116-119) on load.  Unconditional
  // clamped loads + selects (no per-element branches); the host guarantees
  // 1 <= frame_len <= N, and an even frame_len and 8-byte aligned rows when VEC2.
  const T *const x = frames + (size_t)row * (size_t)stride;
  const int flen = (int)frame_len;
  T xr[E], xi[E];
  static_for<E>([&](auto q) {
    const int i0 = 2 * (tid + TP * q);
    T a, b;
    if constexpr (VEC2) {
      const int c = i0 < flen - 2 ? i0 : flen - 2;
      const T2 v = ld_stream2(reinterpret_cast<const T2 *>(x + (unsigned)c));
      a = i0 < flen ? v.x : T(0);
      b = i0 < flen ? v.y : T(0);
    } else {
      const int c0 = i0 < flen - 1 ? i0 : flen - 1, c1 = i0 + 1 < flen - 1 ? i0 + 1 : flen - 1;
      const T v0 = ld_stream(x + (unsigned)c0), v1 = ld_stream(x + (unsigned)c1);
      a = i0 < flen ? v0 : T(0);
      b = i0 + 1 < flen ? v1 : T(0);
    }
    if constexpr (HAS_WIN && !(kExp & 2)) {
      const T2 w = *reinterpret_cast<const T2 *>(win + (unsigned)i0);
      a *= w.x;
      b *= w.y;
    }
    xr[q] = a;
    xi[q] = b;
  });

  fft_passes<T, LOG2M, true>(xr, xi, lrow, TableTwiddles<T, LOG2M>{tw}, tid);
  __syncthreads();

  if (!live) return;
  const int bins = two_sided ? 2 * M : M + 1;
  T *const arow = amp + (size_t)row * (size_t)bins;
  T *const prow = ph ? ph + (size_t)row * (size_t)bins : nullptr;
  // pairs k = tid + TP*q, q < E/2 (k < M/2); k = M/2 is one more pair for tid == 0
  if constexpr (kExp & 4) {
    static_for<E / 2 + 1>([&](auto qc) {
      constexpr int q = qc;
      if (q < E / 2 || tid == 0) {
        const int k = tid + TP * q;
        const T2 z = lrow[lds_pad(k)], zp = lrow[lds_pad((M - k) & (M - 1))];
        arow[k] = sqrt(z.x * z.x + z.y * z.y);
        arow[M - k] = sqrt(zp.x * zp.x + zp.y * zp.y);
      }
    });
    return;
  }
  static_for<E / 2 + 1>([&](auto qc) {
    constexpr int q = qc;
    if (q < E / 2 || tid == 0) {
      const int k = tid + TP * q;
      const int kp = (M - k) & (M - 1);  // Z[M] == Z[0]
      const T2 z = lrow[lds_pad(k)], zp = lrow[lds_pad(kp)];
      const T2 w = twr[k];
      const T er = T(0.5) * (z.x + zp.x), ei = T(0.5) * (z.y - zp.y);  // E
      const T orr = T(0.5) * (z.y + zp.y), oi = T(0.5) * (zp.x - z.x);  // O = (Z - conj Zp)/(2i)
      const T tr = orr * w.x - oi * w.y, ti = orr * w.y + oi * w.x;     // t = W_N^k O
      const T ar = er + tr, ai = ei + ti;                                // X[k]
      const T br = er - tr, bi = -(ei - ti);                             // X[M-k]
      const int k2 = M - k;
      // bins 0 (DC, from k = 0) and M (Nyquist, the partner of k = 0) are not doubled
      const T sc = (k == 0) ? s_edge : s_mid;
      const T ma = mag2(ar, ai) * sc;
      const T mb = mag2(br, bi) * sc;
      st_stream(ma, arow + k);
      if (k2 != k) st_stream(mb, arow + k2);
      if (two_sided && k != 0) {  // X[N-k] = conj X[k]
        st_stream(ma, arow + (2 * M - k));
        if (k2 != k) st_stream(mb, arow + (2 * M - k2));
      }
      if (prow) {
        st_stream(T(atan2(ai, ar)), prow + k);
        if (k2 != k) st_stream(T(atan2(bi, br)), prow + k2);
        if (two_sided && k != 0) {
          st_stream(T(atan2(-ai, ar)), prow + (2 * M - k));
          if (k2 != k) st_stream(T(atan2(-bi, br)), prow + (2 * M - k2));
        }
      }
    }
  });
}

// ---- element-wise kernels (stand-alone applyWindow / magnitude / phase) -----

template <typename T>
__global__ void __launch_bounds__(256)
apply_window_kernel(const T *__restrict__ in, const T *__restrict__ win, T *__restrict__ out,
                    long long total, long long n) {
  const long long step = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += step)
    out[i] = in[i] * win[i % n];
}

template <typename T, bool PHASE>
__global__ void __launch_bounds__(256)
polar_kernel(const T *__restrict__ re, const T *__restrict__ im, T *__restrict__ out, long long total) {
  const long long step = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += step) {
    const T a = re[i], b = im[i];
    out[i] = PHASE ? atan2(b, a) : sqrt(a * a + b * b);
  }
}

// findPeak per frame, spectrum.ts:74-105: among bins >= 1 the first bin holding
// the largest value that is > 0; bin 0 when there is none (amplitudes are >= 0,
// so the reference's global-max fallback can only land on bin 0).
template <typename T>
__global__ void __launch_bounds__(256)
find_peak_kernel(const T *__restrict__ amp, int bins, int *__restrict__ peak, long long batch) {
  __shared__ T sv[256];
  __shared__ int si[256];
  const long long row = blockIdx.x;
  if (row >= batch) return;
  const T *a = amp + (size_t)row * (size_t)bins;
  T bv = T(0);
  int bi = 0;
  for (int i = 1 + (int)threadIdx.x; i < bins; i += 256) {
    const T v = a[i];
    if (v > bv) {  // strict: the earliest index wins inside a thread's stride
      bv = v;
      bi = i;
    }
  }
  sv[threadIdx.x] = bv;
  si[threadIdx.x] = bi;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      const T ov = sv[threadIdx.x + s];
      const int oi = si[threadIdx.x + s];
      const T mv = sv[threadIdx.x];
      const int mi = si[threadIdx.x];
      // larger value wins; equal values: smaller index (first-wins); index 0 = "none"
      if (ov > mv || (ov == mv && oi != 0 && (mi == 0 || oi < mi))) {
        sv[threadIdx.x] = ov;
        si[threadIdx.x] = oi;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) peak[row] = si[0];
}

}  // namespace pdsp

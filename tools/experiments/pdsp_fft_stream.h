// EXPERIMENT, NOT PRODUCT -- a record, not a buildable file: it was written against the scalar
// (re[], im[]) kernel API of commit "Non-temporal streaming accesses, scalar row bases ..." and
// does not compile against the current vector (cx) headers.  What survived of it in the product:
// RegTwiddles (register-resident twiddle bases), now inside pdsp_fft_kernel.h.
// (DESIGN.md section 5 tells the story.)
// built, parity-tested (67 GPU tests passed with it in the dispatch) and measured in round 1;
// it equals the one-row-per-workgroup kernel within 0.5 % once that kernel got non-temporal
// accesses, so the simpler kernel ships.
//
// pdsp_fft_stream.h -- persistent ("streaming") variant of the fused real-frame
// spectrum kernel for N >= 2048.
//
// Same arithmetic as pdsp_fft_kernel.h; what changes is how a workgroup spends its
// time.  The one-row-per-workgroup kernel runs load -> passes -> store strictly in
// sequence; at N = 16384 only two such workgroups fit a CU (LDS), so the memory pipe
// idles whenever both are in their butterfly phases.  Here a workgroup stays resident,
// walks rows g, g + gridDim.x, ..., and
//   * issues the loads of its NEXT row into a second register set before it starts the
//     passes of the current one (the loads fly during the whole compute phase);
//   * keeps the row-invariant inter-pass twiddles in registers for the life of the
//     workgroup: per pass six bases {w, w^2, w^3, w^4, w^8, w^12} of its own k (the other
//     nine of a radix-16 butterfly are one complex product each), so no table traffic
//     and no L2 round trip sits between two LDS exchanges;
//   * (spectrum) parks the window in the registers that are dead while the row is in LDS.
// vmcnt is an in-order counter, so nothing newer than the prefetch may be waited on
// during the passes -- that is why the twiddles must not be loaded per row.
//
// Measured (tools/kbench, N=16384 x 16384 frames): 3.04 TB/s vs 2.89 TB/s for the
// one-row kernel.  The same treatment of the C2C kernel was built and measured SLOWER
// (5.18 vs 5.67 TB/s at N=4096: 3 instead of 4 workgroups per CU, and that kernel already
// sits at the row-pattern copy ceiling), so C2C/R2C stay on fft_stockham_kernel.
#pragma once

#include "pdsp_fft_kernel.h"  // -I pragma-dsp_amd/csrc

namespace pdsp {

// (re + i*im) *= e^{-2*pi*i*NUM/32}, NUM compile-time, any integer.
template <typename T, int NUM>
__device__ __forceinline__ void mul_w32(T &re, T &im) {
  constexpr int m = ((NUM % 32) + 32) % 32;
  if constexpr (m % 2 == 0) {
    constexpr int h = m / 2;  // W16^h
    if constexpr (h >= 8) {
      mul_w16<T, h - 8>(re, im);
      re = -re;
      im = -im;
    } else {
      mul_w16<T, h>(re, im);
    }
  } else {
    // cos(pi*q/16), q = 0..8
    constexpr double C[9] = {1.0, 0.98078528040323044913, 0.92387953251128675613, 0.83146961230254523708,
                             0.70710678118654752440, 0.55557023301960222474, 0.38268343236508977173,
                             0.19509032201612826785, 0.0};
    constexpr int q = m % 16;  // angle = pi*m/16; fold by quadrant
    constexpr bool neg = m >= 16;
    constexpr double cq = q <= 8 ? C[q] : -C[16 - q];
    constexpr double sq = q <= 8 ? C[8 - q] : C[q - 8];
    constexpr T c = T(neg ? -cq : cq), s = T(neg ? sq : -sq);  // W = cos - i sin
    const T a = re, b = im;
    re = a * c - b * s;
    im = a * s + b * c;
  }
}

// Forces the values of a register array to be materialised here (no instruction emitted).
template <typename T, int E, int I = 0>
__device__ __forceinline__ void pin_array(T (&a)[E]) {
  if constexpr (I < E) {
    asm volatile("" : "+v"(a[I]));
    pin_array<T, E, I + 1>(a);
  }
}

// Register-resident twiddle bases of one thread.
template <typename T, int LOG2N>
struct RegTwiddles {
  using TR = FftTraits<LOG2N>;
  using T2 = typename vec2<T>::type;
  static constexpr int nb(int p) { return TR::P.ns[p] > 1 ? (TR::P.r[p] == 16 ? 6 : TR::P.r[p] - 1) : 0; }
  static constexpr int off(int p) {
    int o = 0;
    for (int i = 0; i < p; ++i) o += nb(i);
    return o;
  }
  static constexpr int TOTAL = off(TR::NP) > 0 ? off(TR::NP) : 1;
  // which power r the i-th base of a radix-R pass holds: 16 -> 1,2,3,4,8,12; else 1..R-1
  static constexpr int base_r(int R, int i) { return R == 16 ? (i < 4 ? i + 1 : (i - 2) * 4) : i + 1; }

  T2 tb[TOTAL];

  __device__ __forceinline__ void load(const T2 *__restrict__ tw, const int tid) {
    static_for<TR::NP>([&](auto pc) {
      constexpr int p = pc;
      constexpr int Ns = TR::P.ns[p], R = TR::P.r[p];
      if constexpr (Ns > 1) {
        // Ns <= TP for every pass but a short last one, where j = tid + b*TP < Ns
        const int k0 = Ns <= TR::TP ? (tid & (Ns - 1)) : tid;
        static_for<nb(p)>([&](auto ic) {
          constexpr int r = base_r(R, ic);
          tb[off(p) + ic] = tw[TR::P.twoff[p] + (r - 1) * Ns + k0];
        });
      }
    });
  }

  // Called once per row: makes the bases opaque to loop-invariant code motion, which
  // would otherwise hoist all the derived products out of the row loop and hold them
  // (and spill) for the life of the workgroup.  Emits no instruction.
  __device__ __forceinline__ void pin() {
    static_for<TOTAL>([&](auto ic) { asm volatile("" : "+v"(tb[ic].x), "+v"(tb[ic].y)); });
  }

  template <int p, int r, int b>
  __device__ __forceinline__ T2 get(const int) const {
    constexpr int Ns = TR::P.ns[p], R = TR::P.r[p], O = off(p);
    T2 w;
    if constexpr (R == 16) {
      constexpr int hi = r & 12, lo = r & 3;
      if constexpr (hi == 0) {
        w = tb[O + lo - 1];
      } else if constexpr (lo == 0) {
        w = tb[O + 2 + hi / 4];
      } else {  // w^(hi+lo) = w^hi * w^lo: one product of two table-exact values
        const T2 a = tb[O + 2 + hi / 4], c = tb[O + lo - 1];
        w.x = a.x * c.x - a.y * c.y;
        w.y = a.x * c.y + a.y * c.x;
      }
    } else {
      w = tb[O + r - 1];
    }
    if constexpr (Ns > TR::TP && b > 0) {
      // short last pass: k = tid + b*TP, and W_N^{r*b*TP} = W16^{r*b} because TP = N/16
      mul_w32<T, 2 * ((r * b) % 16)>(w.x, w.y);
    }
    return w;
  }
};

// Occupancy targets (second __launch_bounds__ argument = waves per SIMD; a workgroup of
// WG threads puts WG/256 waves on each SIMD).  They cap the register allocator: left
// alone it takes 250+ VGPRs and halves the resident workgroups.
#ifndef PDSP_STREAM_WAVES_256
#define PDSP_STREAM_WAVES_256 3  // 256-thread workgroups: 3 per CU, <= 168 VGPRs
#endif
#ifndef PDSP_STREAM_WAVES_512
#define PDSP_STREAM_WAVES_512 2  // 512-thread workgroups: 1 per CU, <= 256 VGPRs (128 spills, and a
                                 // spill reload is a VMEM op: its wait drains the prefetch too)
#endif
template <int LOG2N>
struct StreamTraits {
  using TR = FftTraits<LOG2N>;
  static constexpr int MINW = TR::WG == 256 ? PDSP_STREAM_WAVES_256 : TR::WG == 512 ? PDSP_STREAM_WAVES_512 : 4;
};

// ---- fused real-frame spectrum (packed-real, see spectrum_packed_kernel) -------------
// Full frames only (frame_len == N, even stride, 8-byte aligned rows); anything else runs
// on spectrum_packed_kernel.  Two register sets, not three: set A holds the current row
// during the passes and, once the row sits in LDS, receives the window for the NEXT row;
// set B is the prefetch target.  The next row is then formed in place: A = B * A.

// GENERAL = false: one-sided amplitude only (the config-4 shape), which keeps the atan2 and
// mirror-store code out of the persistent loop; GENERAL = true adds phase / two-sided.
template <typename T, int LOG2M, bool HAS_WIN, bool GENERAL>
__global__ void __launch_bounds__(FftTraits<LOG2M>::WG, StreamTraits<LOG2M>::MINW)
spectrum_stream_kernel(const T *__restrict__ frames, const T *__restrict__ win, const long long stride,
                       const typename vec2<T>::type *__restrict__ tw,
                       const typename vec2<T>::type *__restrict__ twr, T *__restrict__ amp, T *__restrict__ ph,
                       const int two_sided, const T s_edge, const T s_mid, const long long batch) {
  using TR = FftTraits<LOG2M>;
  using T2 = typename vec2<T>::type;
  constexpr int E = TR::E, TP = TR::TP, M = TR::N, ROWS = TR::ROWS;
  static_assert(M == 16 * TP, "W_N^{TP*q} = W32^q needs N = 32*TP");

  __shared__ T2 lds[TR::LDS_ELEMS];

  const int tid = (int)(threadIdx.x % TP);
  const int rloc = (int)(threadIdx.x / TP);
  T2 *const lrow = lds + rloc * TR::LROW;
  const long long ngroups = (batch + ROWS - 1) / ROWS;
  const int bins = two_sided ? 2 * M : M + 1;

  RegTwiddles<T, LOG2M> twf;
  twf.load(tw, tid);
  twf.pin();  // land the table loads here, once: no wait inside the loop may cover a load
  T2 twb = twr[tid];  // W_N^tid; W_N^(tid + TP*q) = twb * W32^q
  asm volatile("" : "+v"(twb.x), "+v"(twb.y));

  long long g = blockIdx.x;
  if (g >= ngroups) return;
  auto row_of = [&](long long grp) {
    const long long r = grp * ROWS + rloc;
    return uniform_row<TP>(r < batch ? r : batch - 1);
  };
  // z[m] = x[2m] + i*x[2m+1]: a float2 view of the row; m = tid + TP*q
  auto load_row = [&](long long row, T (&a)[E], T (&b)[E]) {
    const T2 *const x2 = reinterpret_cast<const T2 *>(frames + (size_t)row * (size_t)stride);
    static_for<E>([&](auto q) {
      const T2 v = ld_stream2(x2 + TP * q + (unsigned)tid);
      a[q] = v.x;
      b[q] = v.y;
    });
  };
  auto load_win = [&](T (&a)[E], T (&b)[E]) {
    const T2 *const w2 = reinterpret_cast<const T2 *>(win);
    static_for<E>([&](auto q) {
      const T2 v = (w2 + TP * q)[(unsigned)tid];
      a[q] = v.x;
      b[q] = v.y;
    });
  };

  T xr[E], xi[E];  // set A
  T nr[E], ni[E];  // set B
  load_row(row_of(g), nr, ni);
  if constexpr (HAS_WIN) load_win(xr, xi);
  // applyWindow (spectrum.ts:116-119), in place: A = B * A
  auto form_row = [&]() {
    static_for<E>([&](auto q) {
      xr[q] = HAS_WIN ? nr[q] * xr[q] : nr[q];
      xi[q] = HAS_WIN ? ni[q] * xi[q] : ni[q];
    });
  };
  form_row();
  pin_array<T, E>(xr);  // land the first row before the loop
  pin_array<T, E>(xi);

  for (; g < ngroups; g += gridDim.x) {
    // straight-line body, see fft_stream_kernel: the last iteration re-reads its own row
    const long long gn = g + gridDim.x;
    load_row(row_of(gn < ngroups ? gn : g), nr, ni);  // prefetch, in flight through all passes

    twf.pin();
    T2 twk = twb;  // opaque per row, or LICM hoists (and spills) the nine twb * W32^q products
    asm volatile("" : "+v"(twk.x), "+v"(twk.y));
    fft_passes<T, LOG2M, true>(xr, xi, lrow, twf, tid);
    __syncthreads();
    // the row now lives in LDS: set A takes the window for the next row
    if constexpr (HAS_WIN) load_win(xr, xi);

    {  // unconditional: dead rows of the last group duplicate the clamped last row's stores
      const long long row_st = row_of(g);
      T *const arow = amp + (size_t)row_st * (size_t)bins;
      T *const prow = ph ? ph + (size_t)row_st * (size_t)bins : nullptr;
      // Hermitian split, pairs k = tid + TP*q (q < E/2) and, for tid == 0, k = M/2.
      // LDS addresses are (thread base) +/- (constant): Z[k] at pad(tid) + q*cpad(TP),
      // Z[M-k] at pad(M - tid) - q*cpad(TP), with Z[M] == Z[0] for the k = 0 pair.
      const T2 *const zlo = lrow + lds_pad(tid);
      const T2 *const zhi = lrow + lds_pad(M - tid);
      const T2 *const zhi0 = lrow + lds_pad((M - tid) & (M - 1));
      static_for<E / 2 + 1>([&](auto qc) {
        constexpr int q = qc;
        if (q < E / 2 || tid == 0) {
          const int k = tid + TP * q;
          const T2 z = zlo[cpad(TP * q)];
          const T2 zp = q == 0 ? zhi0[0] : *(zhi - cpad(TP * q));
          T2 w = twk;
          mul_w32<T, q>(w.x, w.y);  // W_N^k
          const T er = T(0.5) * (z.x + zp.x), ei = T(0.5) * (z.y - zp.y);   // E = (Z + conj Zp)/2
          const T orr = T(0.5) * (z.y + zp.y), oi = T(0.5) * (zp.x - z.x);  // O = (Z - conj Zp)/(2i)
          const T tr = orr * w.x - oi * w.y, ti = orr * w.y + oi * w.x;     // t = W_N^k O
          const T ar = er + tr, ai = ei + ti;                                // X[k]
          const T br = er - tr, bi = -(ei - ti);                             // X[M-k]
          const int k2 = M - k;
          const T sc = (k == 0) ? s_edge : s_mid;  // DC and (its partner) Nyquist are not doubled
          const T ma = mag2(ar, ai) * sc;
          const T mb = mag2(br, bi) * sc;
          st_stream(ma, arow + (unsigned)k);
          if (k2 != k) st_stream(mb, arow + (unsigned)k2);
          if constexpr (GENERAL) {
          if (two_sided && k != 0) {  // X[N-k] = conj X[k]
            st_stream(ma, arow + (unsigned)(2 * M - k));
            if (k2 != k) st_stream(mb, arow + (unsigned)(2 * M - k2));
          }
          if (prow) {
            st_stream(T(atan2(ai, ar)), prow + (unsigned)k);
            if (k2 != k) st_stream(T(atan2(bi, br)), prow + (unsigned)k2);
            if (two_sided && k != 0) {
              st_stream(T(atan2(-ai, ar)), prow + (unsigned)(2 * M - k));
              if (k2 != k) st_stream(T(atan2(-bi, br)), prow + (unsigned)(2 * M - k2));
            }
          }
          }
        }
      });
    }
    // consume the prefetch AFTER the stores were issued (see fft_stream_kernel)
    form_row();
    __syncthreads();  // the next row's first pass overwrites the exchange buffer
  }
}

}  // namespace pdsp

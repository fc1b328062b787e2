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
// Arithmetic: a complex point is ONE 2-wide vector register pair cx = (re, im), so
// complex add/sub, scaling and the two halves of a complex multiply are single
// v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 instructions (swaps and sign flips ride
// on the op_sel / neg modifiers).  The N=16384 spectrum kernel is VALU-issue bound, so
// instruction count, not flops, is what this layout buys.
//
// No MFMA: an FFT is not a dense contraction (3.75 flop/B is far below the VALU roof).
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>
#include <utility>

#include "pdsp_radix.h"

namespace pdsp {

// Diagnostic build only (tools/kbench -DPDSP_STAMPS): wave 0 of each workgroup adds the shader
// cycles between phase boundaries into pdsp_stamp_acc[slot]; nothing is emitted otherwise.
#ifdef PDSP_STAMPS
__device__ unsigned long long pdsp_stamp_acc[64];
#define PDSP_STAMP_INIT() unsigned long long stamp_last_ = clock64()
#define PDSP_STAMP(slot)                                                          \
  do {                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                            \
    const unsigned long long t_ = clock64();                                      \
    if (threadIdx.x == 0) atomicAdd(&pdsp_stamp_acc[slot], t_ - stamp_last_);     \
    stamp_last_ = clock64();                                                      \
    __builtin_amdgcn_sched_barrier(0);                                            \
  } while (0)
#else
#define PDSP_STAMP_INIT() ((void)0)
#define PDSP_STAMP(slot) ((void)0)
#endif

template <typename T> struct vec2;
template <> struct vec2<float> { using type = float2; };
template <> struct vec2<double> { using type = double2; };

// A complex value as a native 2-wide vector: .x = re, .y = im.
template <typename T>
using cx = T __attribute__((ext_vector_type(2)));

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

// Empty asm that claims to modify every element: the values become opaque at this point (keeps loads
// from sinking past it in the stamped build; keeps loop-invariant values from being hoisted with
// everything derived from them in the multi-frame kernels).
template <typename T, int E, int I = 0>
__device__ __forceinline__ void pin_regs(T __attribute__((ext_vector_type(2))) (&a)[E]) {
  if constexpr (I < E) {
    asm volatile("" : "+v"(a[I]));
    pin_regs<T, E, I + 1>(a);
  }
}

// ---- complex helpers on cx ---------------------------------------------------

template <typename T>
__device__ __forceinline__ cx<T> mul_neg_i(const cx<T> a) {  // a * (-i) = (im, -re)
  return cx<T>{a.y, -a.x};
}
template <typename T>
__device__ __forceinline__ cx<T> conj(const cx<T> a) {
  return cx<T>{a.x, -a.y};
}
// a + (-i)*b = (a.re + b.im, a.im - b.re)  and  a + i*b = (a.re - b.im, a.im + b.re).
// ONE packed instruction each.  hipcc does not fold a per-half negation into v_pk_add_f32's neg_lo / neg_hi (written as
// plain arithmetic it materialises (-i)*b with a v_xor + v_mov first: 21 % of the N=16384 spectrum kernel's vector
// instructions were such pairs in round 1), so rounds 1-2 spelled the add in inline asm with op_sel / neg modifiers.
// Round 3: hipcc DOES fold the half swap of a v_pk_fma_f32 operand into op_sel, and a multiply by (1, -1) is exact, so
// fma(b.yx, (1, -1), a) is the same value bit for bit, one v_pk_fma_f32 with the constant in an SGPR pair -- and no
// inline asm: every asm statement is an opaque instruction that hipcc pads with an s_nop against its neighbours
// (68 of the N=4096 complex kernel's 673 instructions were such pads) and cannot schedule around.
#ifndef PDSP_CMUL_ONE_ASM
#define PDSP_CMUL_ONE_ASM 1
#endif
#ifndef PDSP_ROT_ASM
#define PDSP_ROT_ASM 0  /* 1: rounds 1-2's inline-asm v_pk_add_f32 forms (A/B builds) */
#endif
template <typename T>
__device__ __forceinline__ cx<T> add_mul_neg_i(const cx<T> a, const cx<T> b) {
  if constexpr (std::is_same_v<T, float> && PDSP_ROT_ASM) {
    cx<float> r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
  } else if constexpr (std::is_same_v<T, float>) {
    return __builtin_elementwise_fma(__builtin_shufflevector(b, b, 1, 0), cx<float>{1.0f, -1.0f}, a);
  } else {
    return cx<T>{a.x + b.y, a.y - b.x};
  }
}
template <typename T>
__device__ __forceinline__ cx<T> add_mul_pos_i(const cx<T> a, const cx<T> b) {
  if constexpr (std::is_same_v<T, float> && PDSP_ROT_ASM) {
    cx<float> r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
  } else if constexpr (std::is_same_v<T, float>) {
    return __builtin_elementwise_fma(__builtin_shufflevector(b, b, 1, 0), cx<float>{-1.0f, 1.0f}, a);
  } else {
    return cx<T>{a.x - b.y, a.y + b.x};
  }
}
// a * w = a.re*(w.re, w.im) + a.im*(-w.im, w.re): one pk_mul + one pk_fma (f32: spelled out for the
// same reason -- the second product's swapped, half-negated operand is an op_sel / neg_lo pattern)
template <typename T>
__device__ __forceinline__ cx<T> cmul(const cx<T> a, const cx<T> w) {
  if constexpr (std::is_same_v<T, float>) {
#if PDSP_CMUL_ONE_ASM
    // both instructions in ONE statement (the product accumulates in the result register): half the asm boundaries
    // for hipcc to pad, one register less
    cx<float> r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
        : "=&v"(r)
        : "v"(a), "v"(w));
    return r;
#else
    cx<float> t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
#endif
  } else {
    return a.xx * w + a.yy * cx<T>{-w.y, w.x};
  }
}

// a * W16^M,  W16 = e^{-2*pi*i/16},  0 <= M < 8.
template <typename T, int M>
__device__ __forceinline__ cx<T> mul_w16(const cx<T> a) {
  constexpr T C1 = T(0.92387953251128673848);  // cos(pi/8)
  constexpr T C2 = T(0.70710678118654752440);  // cos(pi/4)
  constexpr T C3 = T(0.38268343236508977173);  // sin(pi/8)
  if constexpr (M == 0) {
    return a;
  } else if constexpr (M == 4) {  // -i
    return mul_neg_i(a);
  } else if constexpr (M == 2) {  // (1 - i)/sqrt2: (re + im, im - re) * C2
    return add_mul_neg_i(a, a) * C2;
  } else if constexpr (M == 6) {  // (-1 - i)/sqrt2 = -(1 + i)/sqrt2: -(re - im, im + re) * C2
    return add_mul_pos_i(a, a) * (-C2);
  } else {
    constexpr T c = M == 1 ? C1 : M == 3 ? C3 : M == 5 ? -C3 : -C1;
    constexpr T s = M == 1 ? -C3 : M == 3 ? -C1 : M == 5 ? -C1 : -C3;  // W = c + i*s
    return a.xx * cx<T>{c, s} + a.yy * cx<T>{-s, c};
  }
}

template <typename T, int NUM>
__device__ __forceinline__ cx<T> mul_w32(const cx<T> a);

// In-register radix-R DFT (R = 2, 4, 8, 16; 32 for whole tiny rows) as log2 R decimation-in-frequency
// radix-2 stages -- each one a stage of the reference's loop nest
// (src/core/fft.ts:116-140) with a compile-time twiddle.  Output k lands in slot bitrev(k).
// Radix-4 DIF butterfly on four registers, outputs in natural order y0..y3.  The two odd outputs are
// (x0 - x2) -/+ i (x1 - x3): one fused add each, no materialised rotation.
template <typename T>
__device__ __forceinline__ void bfly4(cx<T> &a0, cx<T> &a1, cx<T> &a2, cx<T> &a3) {
  const cx<T> t0 = a0 + a2, t1 = a0 - a2, t2 = a1 + a3, t3 = a1 - a3;
  a0 = t0 + t2;
  a2 = t0 - t2;
  a1 = add_mul_neg_i(t1, t3);
  a3 = add_mul_pos_i(t1, t3);
}
// The same with input x2 still carrying a pending factor -i (x2 = -i * d): t0 = x0 - i d, t1 = x0 + i d.
template <typename T>
__device__ __forceinline__ void bfly4_x2_rot(cx<T> &a0, cx<T> &a1, cx<T> &d, cx<T> &a3) {
  const cx<T> t0 = add_mul_neg_i(a0, d), t1 = add_mul_pos_i(a0, d), t2 = a1 + a3, t3 = a1 - a3;
  a0 = t0 + t2;
  d = t0 - t2;
  a1 = add_mul_neg_i(t1, t3);
  a3 = add_mul_pos_i(t1, t3);
}

template <typename T, int R>
__device__ __forceinline__ void fft_reg(cx<T> (&a)[R]) {
  static_assert(R >= 1 && R <= 32 && (R & (R - 1)) == 0, "radix");
  if constexpr (R == 4) {
    bfly4(a[0], a[1], a[2], a[3]);  // X[q] in slot q; bitrev order wants X[1] <-> X[2] swapped
    const cx<T> x1 = a[1];
    a[1] = a[2];
    a[2] = x1;
    return;
  } else if constexpr (R == 8) {
    // one radix-2 step (W8^k on the odd half; W8^2 = -i stays pending), then a radix-4 on each half
    static_for<4>([&](auto kc) {
      constexpr int k = kc;
      const cx<T> u = a[k], v = a[k + 4];
      a[k] = u + v;
      if constexpr (k == 2) a[k + 4] = u - v;  // times -i inside bfly4_x2_rot
      else a[k + 4] = mul_w32<T, 4 * k>(u - v);
    });
    bfly4(a[0], a[1], a[2], a[3]);            // X[2m] in a[m]
    bfly4_x2_rot(a[4], a[5], a[6], a[7]);     // X[2m+1] in a[4+m]
    cx<T> o[8];
    static_for<8>([&](auto kc) {
      constexpr int k = kc;
      o[bitrev(k, 3)] = a[(k & 1) * 4 + (k >> 1)];
    });
    static_for<8>([&](auto kc) { a[kc] = o[kc]; });
    return;
  } else if constexpr (R == 16) {
    // two radix-4 stages: 64 fused adds + 8 twiddle products, no rotation is ever materialised.
    // Stage A over stride 4: y_q of column j lands in a[j + 4q], then times W16^(j*q).
    static_for<4>([&](auto jc) {
      constexpr int j = jc;
      bfly4(a[j], a[j + 4], a[j + 8], a[j + 12]);
      if constexpr (j > 0) {
        a[j + 4] = mul_w32<T, 2 * j>(a[j + 4]);                           // W16^j
        if constexpr (j != 2) a[j + 8] = mul_w32<T, (4 * j) % 32>(a[j + 8]);  // W16^2j (j = 2: the -i stays pending)
        a[j + 12] = mul_w32<T, (6 * j) % 32>(a[j + 12]);                  // W16^3j
      }
    });
    // Stage B on each group of four: X[q + 4*q2] lands in a[4q + q2]
    bfly4(a[0], a[1], a[2], a[3]);
    bfly4(a[4], a[5], a[6], a[7]);
    bfly4_x2_rot(a[8], a[9], a[10], a[11]);  // a[10] = y_2 of column 2, still to be multiplied by W16^4 = -i
    bfly4(a[12], a[13], a[14], a[15]);
    // callers expect output k in slot bitrev(k, 4)
    cx<T> o[16];
    static_for<16>([&](auto kc) {
      constexpr int k = kc;
      o[bitrev(k, 4)] = a[4 * (k & 3) + (k >> 2)];
    });
    static_for<16>([&](auto kc) { a[kc] = o[kc]; });
    return;
  }
  static_for<ilog2(R)>([&](auto stc) {
    constexpr int s = R >> (stc + 1);  // half length of this stage's sub-transforms
    static_for<R / 2>([&](auto ic) {
      constexpr int g = (ic / s) * 2 * s, k = ic % s;
      constexpr int i0 = g + k, i1 = i0 + s;
      const cx<T> u = a[i0], v = a[i1];
      a[i0] = u + v;
      a[i1] = mul_w32<T, k *(16 / s)>(u - v);  // W_{2s}^k (even multiples resolve to the W16 forms)
    });
  });
}

template <int LOG2N, int LOG2E = 4>
struct FftTraits {
  static constexpr RadixPlan P = make_radix_plan(LOG2N, LOG2E);
  static constexpr int N = P.n, E = P.e, TP = P.tp, NP = P.np;
  static constexpr int WG = TP >= 256 ? TP : 256;  // threads per workgroup
  static constexpr int ROWS = WG / TP;             // transforms per workgroup
  // one pad element every 16: the radix-16 scatter (stride 16 complex) would
  // otherwise put a whole ds_write lane group on one bank
  static constexpr int LROW = N + N / 16;
  static constexpr int LDS_ELEMS = NP > 1 ? ROWS * LROW : 1;
};

// comma-free spelling for the __launch_bounds__ macro argument
template <int LOG2M>
constexpr int kPackedWG = FftTraits<LOG2M, packed_log2e(LOG2M)>::WG;

__device__ __forceinline__ int lds_pad(int i) { return i + (i >> 4); }
// pad(a + c) == pad(a) + cpad(c) when c is a multiple of 16 (or a is and c < 16)
constexpr int cpad(int c) { return c + c / 16; }

// Bank conflicts of this layout, measured where the SQ counters do not saturate (2,048 rows per launch;
// profiles/r03_lds_bank_conflicts_unsaturated.txt): the pad keeps the radix-16 SCATTER (ds_write_b64: 16-lane groups
// over 32 banks) conflict-free but costs every READ-BACK a 2-way conflict -- a ds_read_b64 is served in two groups of
// 32 lanes over 64 banks, and 32 consecutive points of a padded row span 33 slots, so lanes 0 and 31 of each group
// meet on one bank pair: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = exactly 0.25 on fft_stockham_kernel<12> (64 of
// 256 LDS-array cycles per wave), 0.20 on spectrum_dif16k_kernel.  No additive pad that is constant over 16-point
// blocks can serve both sides (the scatter needs the pads of blocks 2s and 2s+1 to differ mod 16, the read-back
// needs them equal mod 32).  An XOR swizzle, i ^ ((i >> 4) & 15), does -- built in round 3 for configs[2]'s kernel:
// conflicts 524,288 -> 0 and LDS-array cycles -25 % per launch, at +59 vector instructions per thread (the
// scatter's addresses are no longer base + constant) and 97 instead of 70 VGPRs; time 0.6681 vs 0.6689 ms, board
// power 1,390 vs 1,388 W (tools' one-process A/B, profiles/r03_experiments/ab_lds_swizzle_*): the conflict cycles
// are neither on the critical path nor in the power budget of an HBM-bound kernel, so the padded form stays and
// the swizzled one was removed again.
// A transform owned by >= 64 threads has one row per wave: tell the compiler the row
// is wave-uniform so row bases live in SGPRs (batch < 2^31 is checked on the host).
template <int TP>
__device__ __forceinline__ long long uniform_row(long long row) {
  if constexpr (TP >= 64) return (long long)__builtin_amdgcn_readfirstlane((int)row);
  else return row;
}

// Order of a kernel's first loads: 1 = the L2-resident tables a thread needs (twiddle bases, split
// twiddles, window values) are requested BEFORE its streamed row (HBM).  The memory counter retires in
// issue order and the vector-memory pipeline is a queue: table loads issued behind the row loads sit
// behind 16-64 KB of streaming requests and their data is not usable until every older HBM load is back.
// Round 1 issued them "right behind the row loads".  Measured A/B (tools/kbench, tools/sweep.py with
// PDSP_LIB_PATH): the fused spectrum kernels, which carry a window's worth of table loads, gain
// (N=16384 decimation-in-time kernel 5.07-5.12 -> 5.40 TB/s; packed kernel N=1024 / 2048 / 4096 / 8192
// +3 / +4 / +1 / +1 %), the N=16384 complex kernel gains +2...7 %; the plain transforms, whose only
// tables are 12-14 twiddle bases, LOSE 1 % (complex) to 3.5 % (real input) -- there the bases stay
// behind the row loads (PDSP_TABLES_FIRST_C2C 0).  0 / 1 are compile-time A/B switches.
#ifndef PDSP_TABLES_FIRST
#define PDSP_TABLES_FIRST 1      /* spectrum_packed_kernel, fft_split4_kernel, fft_split2_kernel */
#endif
#ifndef PDSP_TABLES_FIRST_C2C
#define PDSP_TABLES_FIRST_C2C 0  /* fft_stockham_kernel */
#endif
__device__ __forceinline__ void load_order_fence() { __builtin_amdgcn_sched_barrier(0); }

// Streamed rows are touched exactly once: non-temporal loads/stores keep them from
// displacing the twiddle/window tables in L2 (+4..10 % on the row-pattern copy and on
// the C2C kernel, tools/kbench).
template <typename T>
__device__ __forceinline__ T ld_stream(const T *p) { return __builtin_nontemporal_load(p); }
template <typename T>
__device__ __forceinline__ void st_stream(T v, T *p) { __builtin_nontemporal_store(v, p); }
// Amplitude / phase rows are N/2+1 floats long, so a wave's 256-byte store straddles
// cache lines that the neighbouring waves complete: plain stores let L2 combine them
// (PMC: non-temporal stores wrote 13 % more than the algorithmic bytes here).
#ifndef PDSP_AMP_STORE_NT
#define PDSP_AMP_STORE_NT 0
#endif
template <typename T>
__device__ __forceinline__ void st_rowtail(T v, T *p) {
  if constexpr (PDSP_AMP_STORE_NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}

// |z| for the fused amplitude stores: one v_sqrt_f32 (1 ulp) instead of the ~10-instruction
// correctly rounded sequence -- 17 of them per thread sit in the VALU-bound epilogue of the
// N=16384 spectrum kernel.  1 ulp = 6e-8 relative, far inside the 1e-5 contract.
__device__ __forceinline__ float mag(const cx<float> z) {
  const cx<float> s = z * z;
  return __builtin_amdgcn_sqrtf(s.x + s.y);
}
// f64 is the drop-in's default arithmetic and the reference's magnitude() is Math.hypot
// (src/xform/fourier.ts:106): no overflow above 1e154, no underflow below 1e-162.
// (Round 3 tried sqrt(fma(re, re, im * im)) with hypot only outside [2^-900, 2^1000]: the fast path is ~15 f64
// instructions against hypot's ~35, 17 times per thread -- and the f64 spectrum sweep moved by +4 % at N = 1024,
// +-1 % at 2048 ... 8192 and -6 ... -8 % at N = 128 ... 512: no gain, hypot stays.
// profiles/r03_experiments/sweep_f64_hypot_vs_fastmag.txt)
__device__ __forceinline__ double mag(const cx<double> z) { return hypot(z.x, z.y); }

// ---- load / store policies ------------------------------------------------
// Eight consecutive values of a cosine-sum window (createWindow, fourier.ts:14-52) by angle addition, two per
// packed instruction: c0 = (cos t, sin t) of the first sample, we[e] = (cos f e, sin f e) wave-uniform;
// w[e] = k0 + k1 cos(t + f e), or k0 + c (k1 + k2 c) for the three-term window (cos 2t = 2 c^2 - 1 folded in).
template <typename T, bool THREE>
__device__ __forceinline__ void fused_window_pairs(const cx<T> c0, const cx<T> *__restrict__ we, const T k0, const T k1,
                                                   const T k2, cx<T> (&w)[4]) {
  static_for<4>([&](auto h) {
    const cx<T> e0 = we[2 * h], e1 = we[2 * h + 1];
    const cx<T> c = cx<T>{e0.x, e1.x} * c0.x - cx<T>{e0.y, e1.y} * c0.y;  // cos(t + f e), e = 2h, 2h + 1
    if constexpr (THREE) w[h] = c * (c * k2 + k1) + k0;
    else w[h] = c * k1 + k0;
  });
}

// ld(row, off, lane) -> cx: fetch point off + lane of row `row` (row < batch).
// st(row, off, lane, cx): write point off + lane.
// `row` and `off` are wave-uniform whenever a transform spans whole waves, so
// `base + row*N + off` stays in SGPRs and the only per-lane address register is
// `lane` (global_load ... v_lane, s[base:base+1] offset:imm).

template <typename T>
struct LoadComplex {  // forwardComplex / inverse (planes swapped by the caller)
  const T *__restrict__ re;
  const T *__restrict__ im;
  long long n;
  __device__ __forceinline__ cx<T> operator()(long long row, int off, int lane) const {
    const size_t o = (size_t)row * (size_t)n + (size_t)off;
    return cx<T>{ld_stream(re + o + (unsigned)lane), ld_stream(im + o + (unsigned)lane)};
  }
  static constexpr bool kPlanar = true;  // rows are contiguous planes: eligible for staged I/O
  static constexpr bool kHasIm = true;
  static constexpr bool kPacked = false;
  __device__ __forceinline__ const T *plane_re() const { return re; }
  __device__ __forceinline__ const T *plane_im() const { return im; }
};

template <typename T>
struct LoadReal {  // Radix2Fft.forward: imaginary part is zero
  const T *__restrict__ re;
  long long n;
  __device__ __forceinline__ cx<T> operator()(long long row, int off, int lane) const {
    return cx<T>{ld_stream(re + (size_t)row * (size_t)n + (size_t)off + (unsigned)lane), T(0)};
  }
  static constexpr bool kPlanar = true;
  static constexpr bool kHasIm = false;
  static constexpr bool kPacked = false;
  __device__ __forceinline__ const T *plane_re() const { return re; }
  __device__ __forceinline__ const T *plane_im() const { return nullptr; }
};

// Packed real frames for fft_split4_kernel: point m of the row is (x[2m], x[2m+1]) * (w[2m], w[2m+1]) of a
// frame of 2N samples, `stride` samples apart (16-byte aligned) -- the first pass of the packed-real
// spectrum path at N = 32768 (split_amp_rows_kernel undoes the packing).
// WIN: 0 rect, 1 window table (2N values), 2 / 3 two- / three-term cosine-sum window evaluated in registers
// (createWindow fused, fourier.ts:14-52): sample n = 8 (tid + 256 q) + e has, with f = 2 pi / (2N - 1) and
// cs(t) = (cos t, sin t),  cs(f n) = wb[tid] * wq[q] * we[e]  (wb[j] = cs(8 f j), wq[q] = cs(2048 f q),
// we[e] = cs(f e); f64-built), w = k0 + k1 c or k0 + c (k1 + k2 c), c = cos(f n).
template <typename T, int WIN>
struct LoadPackedFrames {
  const T *__restrict__ x;
  const T *__restrict__ win;
  long long stride;
  const float *wb, *wq, *we;
  float k0, k1, k2;
  static constexpr bool kPlanar = true;  // whole rows, vector loads
  static constexpr bool kHasIm = true;
  static constexpr bool kPacked = true;
  static constexpr int kWin = WIN;
};

// Interleaved (re, im) rows -- the layout of I/Q streams and of complex64 tensors; one 8-byte
// access per point.  CONJ turns the forward kernel into the inverse: conj on the way in and out.
template <typename T, bool CONJ>
struct LoadInterleaved {
  const cx<T> *__restrict__ z;
  long long n;
  __device__ __forceinline__ cx<T> operator()(long long row, int off, int lane) const {
    cx<T> v = ld_stream(z + (size_t)row * (size_t)n + (size_t)off + (unsigned)lane);
    if constexpr (CONJ) v.y = -v.y;
    return v;
  }
  static constexpr bool kPlanar = false;
  static constexpr bool kHasIm = true;
};

template <typename T, bool CONJ>
struct StoreInterleaved {
  cx<T> *__restrict__ z;
  long long n;
  T scale;
  __device__ __forceinline__ void operator()(long long row, int off, int lane, cx<T> v) const {
    v = v * scale;
    if constexpr (CONJ) v.y = -v.y;
    st_stream(v, z + (size_t)row * (size_t)n + (size_t)off + (unsigned)lane);
  }
  static constexpr bool kPlanar = false;
};

// Loads are unconditional (clamped address + select): a per-element `if` around a
// load makes hipcc branch and drain vmcnt per element (cdna guide, section 5 item 4c).
template <typename T, bool HAS_WIN>
struct LoadFrameWindowed {  // buildFrame + applyWindow, spectrum.ts:36-43, :116-119
  const T *__restrict__ x;
  const T *__restrict__ win;  // N values when HAS_WIN (rect otherwise)
  long long frame_len;        // 1 <= samples used per row <= N; the rest reads as zero
  long long stride;
  __device__ __forceinline__ cx<T> operator()(long long row, int off, int lane) const {
    const int i = off + lane;
    const int last = (int)frame_len - 1;
    T v = ld_stream(x + (size_t)row * (size_t)stride + (unsigned)(i < last ? i : last));
    v = i <= last ? v : T(0);
    if constexpr (HAS_WIN) v *= win[i];
    return cx<T>{v, T(0)};
  }
};

template <typename T>
struct StoreComplex {
  T *__restrict__ re;
  T *__restrict__ im;
  long long n;
  T scale;  // 1 forward, 1/N inverse (fft.ts:142-148); power of two => exact
  __device__ __forceinline__ void operator()(long long row, int off, int lane, cx<T> v) const {
    const size_t o = (size_t)row * (size_t)n + (size_t)off;
    v = v * scale;
    st_stream(v.x, re + o + (unsigned)lane);
    st_stream(v.y, im + o + (unsigned)lane);
  }
  static constexpr bool kPlanar = true;
};

template <typename T>
struct StoreAmplitude {  // magnitude + scaleAmplitude{One,Two}Sided [+ phase]
  T *__restrict__ amp;
  T *__restrict__ ph;  // may be null
  int bins;            // N/2+1 or N
  int nyq;             // N/2 for one-sided (that bin is not doubled), -1 for two-sided
  T s_edge;            // 1/N
  T s_mid;             // 2/N one-sided, 1/N two-sided
  __device__ __forceinline__ void operator()(long long row, int off, int lane, cx<T> v) const {
    const int i = off + lane;
    if (i < bins) {
      const size_t o = (size_t)row * (size_t)bins;
      st_stream(mag(v) * ((i == 0 || i == nyq) ? s_edge : s_mid), amp + o + (unsigned)i);
      if (ph) st_stream(T(atan2(v.y, v.x)), ph + o + (unsigned)i);
    }
  }
};

// ---- twiddle providers ---------------------------------------------------------
// twf.template get<p, r, b>(j) = W_{Ns*R}^{r*(j mod Ns)} for input r of butterfly j = tid + b*TP of pass p.

// Reads the host-built table (layout: pdsp_radix.h) at every use: uniform table base per r
// (SGPR) + one 32-bit lane offset k.
template <typename T, int LOG2N, int LOG2E = 4>
struct TableTwiddles {
  const cx<T> *__restrict__ tw;
  template <int p, int r, int b>
  __device__ __forceinline__ cx<T> get(const int j) const {
    using TR = FftTraits<LOG2N, LOG2E>;
    constexpr int Ns = TR::P.ns[p];
    return (tw + (TR::P.twoff[p] + (r - 1) * Ns))[(unsigned)(j & (Ns - 1))];
  }
};

// a * e^{-2*pi*i*NUM/32}, NUM compile-time
template <typename T, int NUM>
__device__ __forceinline__ cx<T> mul_w32(const cx<T> a) {
  constexpr int m = ((NUM % 32) + 32) % 32;
  if constexpr (m % 2 == 0) {
    constexpr int h = m / 2;  // W16^h
    if constexpr (h >= 8) return -mul_w16<T, h - 8>(a);
    else return mul_w16<T, h>(a);
  } else {
    // cos(pi*q/16), q = 0..8
    constexpr double C[9] = {1.0, 0.98078528040323044913, 0.92387953251128675613, 0.83146961230254523708,
                             0.70710678118654752440, 0.55557023301960222474, 0.38268343236508977173,
                             0.19509032201612826785, 0.0};
    constexpr int q = m % 16;
    constexpr bool neg = m >= 16;
    constexpr int lo = q <= 8 ? q : 16 - q;  // fold onto the first quarter turn
    constexpr double cq = q <= 8 ? C[lo] : -C[lo];
    constexpr double sq = C[8 - lo];
    constexpr T c = T(neg ? -cq : cq), s = T(neg ? sq : -sq);  // W = c + i*s
    return a.xx * cx<T>{c, s} + a.yy * cx<T>{-s, c};
  }
}

// Per-thread twiddle BASES held in registers.  A thread's twiddles depend only on its
// position in the transform, so they are fetched once, right behind the frame loads (their
// L2 latency hides under the HBM wait) instead of inside every pass, where a stamped build
// showed each table fetch exposed for thousands of cycles behind the streaming traffic.
// Per radix-16 pass six bases {w, w^2, w^3, w^4, w^8, w^12} of the thread's own k; the other
// nine are one product of two table-exact values.  A short last pass (Ns > TP) needs only
// k = tid: W_N^{r*(tid + b*TP)} = base_r * W16^{r*b} because TP = N/16.
template <typename T, int LOG2N, int LOG2E = 4>
struct RegTwiddles {
  using TR = FftTraits<LOG2N, LOG2E>;
  static_assert(LOG2E == 4, "the W16 correction of the short last pass assumes TP = N/16");
  static constexpr int nb(int p) { return TR::P.ns[p] > 1 ? (TR::P.r[p] == 16 ? 6 : TR::P.r[p] - 1) : 0; }
  static constexpr int off(int p) {
    int o = 0;
    for (int i = 0; i < p; ++i) o += nb(i);
    return o;
  }
  static constexpr int TOTAL = off(TR::NP) > 0 ? off(TR::NP) : 1;
  // which power r the i-th base of a radix-R pass holds: 16 -> 1,2,3,4,8,12; else 1..R-1
  static constexpr int base_r(int R, int i) { return R == 16 ? (i < 4 ? i + 1 : (i - 2) * 4) : i + 1; }

  cx<T> tb[TOTAL];

  __device__ __forceinline__ void load(const cx<T> *__restrict__ tw, const int tid) {
    static_for<TR::NP>([&](auto pc) {
      constexpr int p = pc;
      constexpr int Ns = TR::P.ns[p], R = TR::P.r[p];
      if constexpr (Ns > 1) {
        // Ns <= TP for every pass but a short last one, where j = tid + b*TP < Ns
        const unsigned k0 = (unsigned)(Ns <= TR::TP ? (tid & (Ns - 1)) : tid);
        static_for<nb(p)>([&](auto ic) {
          constexpr int r = base_r(R, ic);
          tb[off(p) + ic] = (tw + (TR::P.twoff[p] + (r - 1) * Ns))[k0];
        });
      }
    });
  }

  template <int p, int r, int b>
  __device__ __forceinline__ cx<T> get(const int) const {
    constexpr int Ns = TR::P.ns[p], R = TR::P.r[p], O = off(p);
    cx<T> w;
    if constexpr (R == 16) {
      constexpr int hi = r & 12, lo = r & 3;
      if constexpr (hi == 0) w = tb[O + lo - 1];
      else if constexpr (lo == 0) w = tb[O + 2 + hi / 4];
      else w = cmul(tb[O + 2 + hi / 4], tb[O + lo - 1]);  // w^(hi+lo) = w^hi * w^lo
    } else {
      w = tb[O + r - 1];
    }
    if constexpr (Ns > TR::TP && b > 0) w = mul_w32<T, 2 * ((r * b) % 16)>(w);  // * W16^{r*b}
    return w;
  }
};

// ---- the passes --------------------------------------------------------------

// Pass p for one thread's E points: twiddle, radix-R butterflies, then either the autosort scatter
// to LDS or (register-resident last pass) back into x in natural order.
template <typename T, int LOG2N, bool LAST_TO_LDS, int LOG2E, int p, class TWF, int EE>
__device__ __forceinline__ void fft_pass_compute(cx<T> (&x)[EE], cx<T> *const lrow, const TWF &twf, const int tid) {
  using TR = FftTraits<LOG2N, LOG2E>;
  constexpr int E = TR::E, TP = TR::TP, NP = TR::NP;
  static_assert(EE == E, "register array size must be the plan's points per thread");
  // For N >= 256 every LDS address is (a thread-only base) + (a compile-time offset),
  // so there is one address register per pass instead of one per element.
  constexpr bool kConstOffsets = (TP % 16 == 0);
  constexpr int R = TR::P.r[p], Ns = TR::P.ns[p], EB = E / R, LR = ilog2(R);
  constexpr bool last = (p == NP - 1);
  constexpr bool to_lds = !last || LAST_TO_LDS;

  static_for<EB>([&](auto bc) {
    constexpr int b = bc;
    cx<T> a[R];
    static_for<R>([&](auto rc) { a[rc] = x[b + rc * EB]; });
    const int j = tid + b * TP;  // butterfly index within the pass, 0 <= j < N/R
    if constexpr (Ns > 1) {
      static_for<R - 1>([&](auto rc) {
        constexpr int r = rc + 1;
        a[r] = cmul(a[r], twf.template get<p, r, b>(j));
      });
    }
    fft_reg<T, R>(a);
    if constexpr (!to_lds) {
      // Ns*R == N: output r of butterfly j is X[j + r*N/R] = slot b + r*EB
      static_for<R>([&](auto rc) { x[b + rc * EB] = a[bitrev(rc, LR)]; });
    } else if constexpr (kConstOffsets) {
      // autosort scatter (natural order when Ns*R == N).  j = tid + b*TP: the part of the
      // index that depends on b and rc is the constant c = cb + rc*Ns, and
      // pad(j0t + c) == pad(j0t) + cpad(c): cb is a multiple of 16, and (j0t & 15) + (rc*Ns & 15)
      // never carries (for Ns < 16 the first is < Ns, the second a multiple of Ns below 16)
      constexpr int cb = Ns <= TP ? b * TP * R : b * TP;
      const int j0t = Ns <= TP ? ((tid >> ilog2(Ns)) << ilog2(Ns * R)) + (tid & (Ns - 1)) : tid;
      cx<T> *const wbase = lrow + lds_pad(j0t);
      static_for<R>([&](auto rc) { wbase[cpad(cb + rc * Ns)] = a[bitrev(rc, LR)]; });
    } else {
      const int j0 = ((j >> ilog2(Ns)) << ilog2(Ns * R)) + (j & (Ns - 1));
      static_for<R>([&](auto rc) { lrow[lds_pad(j0 + rc * Ns)] = a[bitrev(rc, LR)]; });
    }
  });
}

// The thread's E inputs of the next pass, x[q] = row[tid + TP*q], after the barrier behind a scatter.
template <typename T, int LOG2N, int LOG2E, int EE>
__device__ __forceinline__ void fft_pass_readback(cx<T> (&x)[EE], const cx<T> *const lrow, const int tid) {
  using TR = FftTraits<LOG2N, LOG2E>;
  constexpr int E = TR::E, TP = TR::TP;
  // (For TP <= 128 hipcc pairs these read-backs into ds_read2_b64, which the LDS serves at half ds_read_b64's rate.
  // Keeping them apart -- round-robin opaque bases, +4 instructions -- measured +-0 on every size from 256 to 8192,
  // C2C and fused spectrum (profiles/r03_experiments/ab_lds_read2_vs_split.log): LDS time is not what bounds these
  // kernels, same finding as the bank-conflict experiment at lds_pad.)
  if constexpr (TP % 16 == 0) {
    const cx<T> *const rbase = lrow + lds_pad(tid);
    static_for<E>([&](auto q) { x[q] = rbase[cpad(TP * q)]; });
  } else {
    static_for<E>([&](auto q) { x[q] = lrow[lds_pad(tid + TP * q)]; });
  }
}

// Runs every pass of the length-2^LOG2N transform on the E points each of the TP
// cooperating threads holds.  LAST_TO_LDS = false: the result comes back in the
// registers, X[tid + TP*q] in slot q.  LAST_TO_LDS = true: the last pass also
// scatters to LDS, in natural order (X[k] at lds_pad(k)), for a consumer that needs
// other threads' bins; the caller must __syncthreads() before reading it.
// Inter-pass twiddles W_{Ns*R}^{r*k} come from the provider `twf` (table or registers).
template <typename T, int LOG2N, bool LAST_TO_LDS, int LOG2E = 4, class TWF = void, int EE = 0>
__device__ __forceinline__ void fft_passes(cx<T> (&x)[EE], cx<T> *const lrow, const TWF &twf, const int tid) {
  PDSP_STAMP_INIT();
  constexpr int NP = FftTraits<LOG2N, LOG2E>::NP;
  static_for<NP>([&](auto pc) {
    constexpr int p = pc;
    fft_pass_compute<T, LOG2N, LAST_TO_LDS, LOG2E, p>(x, lrow, twf, tid);
    PDSP_STAMP(8 + 4 * p);  // twiddle loads + butterflies + LDS scatter issued
    if constexpr (p != NP - 1) {
      __syncthreads();
      PDSP_STAMP(9 + 4 * p);  // barrier (incl. draining the scatter)
      fft_pass_readback<T, LOG2N, LOG2E>(x, lrow, tid);
      // the next pass writes LDS again (every pass but a register-resident last one)
      if constexpr (p + 1 < NP - 1 || LAST_TO_LDS) __syncthreads();
      PDSP_STAMP(10 + 4 * p);  // LDS read-back issued + barrier
    }
  });
}

// ---- findPeak on the device ------------------------------------------------------
// SpectrumPeak per frame (src/public/spectrum.ts:15-20) in f32; layout == pdsp_peak32.
struct PeakRec {
  int index;
  float frequency;
  float amplitude;
  float phase;
};

// Running arg-max with findPeak's rules (spectrum.ts:74-105) over bins >= 1: a larger
// value wins; an equal value wins only with a smaller index ("strict >, first wins",
// made order-independent); only values > 0 ever enter (index 0 = "no bin > 0 yet").
template <typename T>
struct PeakBest {
  T v;
  int i;
  cx<T> x;  // the complex bin, so the phase is one atan2 at the very end
  __device__ __forceinline__ void consider(const T ov, const int oi, const cx<T> ox) {
    if (ov > v || (ov == v && ov > T(0) && oi < i)) {
      v = ov;
      i = oi;
      x = ox;
    }
  }
  // A thread that visits its OWN bins in a known index order does not need the order-independent rule: in ASCENDING
  // order the strict '>' alone is "first wins" (an equal value at a larger index never replaces), in DESCENDING order
  // '>=' is (an equal value at a smaller index does replace).  One compare + four selects per bin where the rule
  // above costs four compares and their mask logic -- 32 bins per thread in the N = 16384 kernel, on a peaks-only
  // path that is bound by instruction issue, not by HBM.  A descending run may pick up a bin of value 0; the
  // order-independent merge that follows drops it (only values > 0 enter).  NaNs never enter, as before.
  __device__ __forceinline__ void consider_ascending(const T ov, const int oi, const cx<T> ox) {
    if (ov > v) {
      v = ov;
      i = oi;
      x = ox;
    }
  }
  __device__ __forceinline__ void consider_descending(const T ov, const int oi, const cx<T> ox) {
    if (ov >= v) {
      v = ov;
      i = oi;
      x = ox;
    }
  }
};

// ---- the kernels ---------------------------------------------------------------

template <typename T, int LOG2N, class LD, class ST>
__global__ void __launch_bounds__(FftTraits<LOG2N>::WG)
fft_stockham_kernel(const LD ld, const ST st, const typename vec2<T>::type *__restrict__ tw,
                    const long long batch) {
  using TR = FftTraits<LOG2N>;
  constexpr int E = TR::E, TP = TR::TP, NP = TR::NP;

  __shared__ cx<T> lds[TR::LDS_ELEMS];

  const int tid = TP == 1 ? 0 : (int)(threadIdx.x % TP);
  const int rloc = (int)(threadIdx.x / TP);
  const long long row_raw = (long long)blockIdx.x * TR::ROWS + rloc;
  const bool live = row_raw < batch;
  // dead rows of the last workgroup recompute the last live row and skip the
  // store, so that every thread reaches every barrier without predicated loads
  const long long row = uniform_row<TP>(live ? row_raw : batch - 1);
  cx<T> *const lrow = lds + (NP > 1 ? rloc * TR::LROW : 0);

  cx<T> x[E];
#ifndef PDSP_C2C_TABLE_TWIDDLES
  constexpr bool kRegTw = (TP >= 16 && NP > 1);  // twiddle bases in registers, fetched with the row loads
#else
  constexpr bool kRegTw = false;
#endif
  std::conditional_t<kRegTw, RegTwiddles<T, LOG2N>, TableTwiddles<T, LOG2N>> twf;
  if constexpr (kRegTw && PDSP_TABLES_FIRST_C2C) {
    twf.load(reinterpret_cast<const cx<T> *>(tw), tid);
    load_order_fence();
  }
  static_for<E>([&](auto q) { x[q] = ld(row, TP * q, tid); });
  if constexpr (kRegTw && !PDSP_TABLES_FIRST_C2C) twf.load(reinterpret_cast<const cx<T> *>(tw), tid);
  if constexpr (!kRegTw) twf.tw = reinterpret_cast<const cx<T> *>(tw);
  fft_passes<T, LOG2N, false>(x, lrow, twf, tid);
  if (live) {
    static_for<E>([&](auto q) { st(row, TP * q, tid, x[q]); });
  }
}

// Small transforms (32 <= N <= 256): only N/16 <= 16 threads own a row, so the direct kernel's
// wave-level loads are 16..64-byte fragments of many rows (N = 64 measured 15 % of the HBM
// roofline).  But the 4096/N rows of a workgroup are ONE contiguous 4096-point chunk per plane:
// stage it through LDS with perfectly coalesced 16-byte accesses on both sides -- one extra LDS
// round trip each way, cheap next to the few passes such a size needs.
//   LD/ST must be the planar policies (kPlanar) over 16-byte aligned planes.
template <typename T, int LOG2N, class LD, class ST>
__global__ void __launch_bounds__(256)
fft_staged_kernel(const LD ld, const ST st, const typename vec2<T>::type *__restrict__ tw, const long long batch) {
  using TR = FftTraits<LOG2N>;
  constexpr int N = TR::N, E = TR::E, TP = TR::TP, ROWS = TR::ROWS, WG = 256;
  static_assert(TR::WG == WG && TR::NP > 1 && N % 16 == 0, "staged path: 32 <= N <= 256");
  constexpr int CHUNK = ROWS * N;  // 4096 points per plane
  typedef T V4 __attribute__((ext_vector_type(4)));
  __shared__ cx<T> lds[TR::LDS_ELEMS];

  const int t = (int)threadIdx.x;
  const int tid = t % TP, rloc = t / TP;
  cx<T> *const lrow = lds + rloc * TR::LROW;
  const size_t base = (size_t)blockIdx.x * CHUNK;
  const size_t limit = (size_t)batch * N;  // points per plane; a multiple of 4
  const T *const pre = ld.plane_re();
  const T *const pim = ld.plane_im();

  // chunk -> LDS in natural order: point p of the chunk is element p % N of local row p / N
  static_for<CHUNK / 4 / WG>([&](auto ic) {
    const int p = 4 * (t + WG * ic);
    size_t g = base + (size_t)p;
    g = g + 4 <= limit ? g : limit - 4;  // the tail workgroup re-reads valid points; its dead rows never store
    const V4 r = ld_stream(reinterpret_cast<const V4 *>(pre + g));
    V4 m = V4{T(0), T(0), T(0), T(0)};
    if constexpr (LD::kHasIm) m = ld_stream(reinterpret_cast<const V4 *>(pim + g));  // no run-time branch around a load
    cx<T> *const d = lds + (p / N) * TR::LROW + lds_pad(p % N);  // 4 points never straddle a 16-block
    d[0] = cx<T>{r.x, m.x};
    d[1] = cx<T>{r.y, m.y};
    d[2] = cx<T>{r.z, m.z};
    d[3] = cx<T>{r.w, m.w};
  });
  __syncthreads();
  cx<T> x[E];
  {
    const cx<T> *const rbase = lrow + lds_pad(tid);
    static_for<E>([&](auto q) { x[q] = rbase[cpad(TP * q)]; });
  }
  __syncthreads();  // the first pass scatters into the same buffer

  RegTwiddles<T, LOG2N> twf;
  twf.load(reinterpret_cast<const cx<T> *>(tw), tid);
  fft_passes<T, LOG2N, true>(x, lrow, twf, tid);  // result in LDS, natural order per row
  __syncthreads();

  static_for<CHUNK / 4 / WG>([&](auto ic) {
    const int p = 4 * (t + WG * ic);
    const size_t g = base + (size_t)p;
    if (g + 4 <= limit) {
      const cx<T> *const d = lds + (p / N) * TR::LROW + lds_pad(p % N);
      const cx<T> v0 = d[0] * st.scale, v1 = d[1] * st.scale, v2 = d[2] * st.scale, v3 = d[3] * st.scale;
      st_stream(V4{v0.x, v1.x, v2.x, v3.x}, reinterpret_cast<V4 *>(st.re + g));
      st_stream(V4{v0.y, v1.y, v2.y, v3.y}, reinterpret_cast<V4 *>(st.im + g));
    }
  });
}

// Tiny transforms (2 <= N <= 16; the amplitude spectrum also N = 32): one thread owns a whole row, so in the direct kernel a lane's
// accesses are N*4 bytes apart (N = 16 measured 4 % of the HBM roofline).  Same cure as
// fft_staged_kernel: the workgroup's 4096-point chunk goes through LDS with coalesced 16-byte
// accesses on both sides; each thread transforms 16/N rows out of LDS in registers (fft_reg).
//   AMP = false: planar complex rows out (times st.scale) -- forward / forwardComplex / inverse.
//   AMP = true:  real frames (whole, contiguous) times an optional window in, one- or two-sided
//                amplitude rows of `bins` values out (the body of spectrum(), complex kernel on (x*w, 0)).
template <typename T, int LOG2N, bool AMP, class LD>
__global__ void __launch_bounds__(256)
fft_tiny_staged_kernel(const LD ld, const T *__restrict__ win, T *__restrict__ o1, T *__restrict__ o2, const T scale,
                       const int bins, const int nyq, const T s_edge, const T s_mid, const long long batch) {
  constexpr int N = 1 << LOG2N, WG = 256, CHUNK = 4096, ROWS = CHUNK / N, RPT = ROWS >= WG ? ROWS / WG : 1;
  static_assert(LOG2N >= 1 && LOG2N <= 5, "tiny path: 2 <= N <= 32 (N = 32: half the threads hold a row)");
  typedef T V4 __attribute__((ext_vector_type(4)));
  __shared__ cx<T> lds[CHUNK + CHUNK / 16];
  const int t = (int)threadIdx.x;
  const size_t base = (size_t)blockIdx.x * CHUNK;
  const size_t limit = (size_t)batch * N;  // points per plane
  const T *const pre = ld.plane_re();
  const T *const pim = ld.plane_im();

  static_for<CHUNK / 4 / WG>([&](auto ic) {
    const int p = 4 * (t + WG * ic);
    size_t g = base + (size_t)p;
    if (limit >= 4) g = g + 4 <= limit ? g : limit - 4;  // the tail workgroup re-reads valid points
    V4 r, m = V4{T(0), T(0), T(0), T(0)};
    if (limit >= 4) {
      r = ld_stream(reinterpret_cast<const V4 *>(pre + g));
      if constexpr (LD::kHasIm) m = ld_stream(reinterpret_cast<const V4 *>(pim + g));
    } else {  // a single N = 2 row: two points in all
      r = V4{pre[0], pre[1], T(0), T(0)};
      if constexpr (LD::kHasIm) m = V4{pim[0], pim[1], T(0), T(0)};
    }
    if constexpr (N == 2) {
      // an odd number of 2-point rows: the group that straddles the end was read two points early
      const size_t g0 = base + (size_t)p;
      if (limit >= 4 && g0 < limit && g0 + 4 > limit) {
        r = V4{r.z, r.w, T(0), T(0)};
        m = V4{m.z, m.w, T(0), T(0)};
      }
    }
    cx<T> *const d = lds + lds_pad(p);
    d[0] = cx<T>{r.x, m.x};
    d[1] = cx<T>{r.y, m.y};
    d[2] = cx<T>{r.z, m.z};
    d[3] = cx<T>{r.w, m.w};
  });
  __syncthreads();

  cx<T> x[RPT][N];
  const bool holds_row = ROWS >= WG || t < ROWS;
  static_for<RPT>([&](auto rc) {
    const int row = holds_row ? t + WG * rc : ROWS - 1;  // local row; its points are lds[pad(row*N + q)]
    static_for<N>([&](auto q) {
      cx<T> v = lds[lds_pad(row * N + q)];
      if constexpr (AMP) {
        if (win) v = v * win[q];  // uniform across the lanes: scalar loads
      }
      x[rc][q] = v;
    });
    fft_reg<T, N>(x[rc]);  // X[k] in slot bitrev(k)
  });
  __syncthreads();  // every thread has its inputs: the buffer now takes the outputs

  if constexpr (!AMP) {
    static_for<RPT>([&](auto rc) {
      const int row = t + WG * rc;
      if (holds_row) static_for<N>([&](auto k) { lds[lds_pad(row * N + k)] = x[rc][bitrev(k, LOG2N)] * scale; });
    });
    __syncthreads();
    static_for<CHUNK / 4 / WG>([&](auto ic) {
      const int p = 4 * (t + WG * ic);
      const size_t g = base + (size_t)p;
      const cx<T> *const d = lds + lds_pad(p);
      if (g + 4 <= limit) {
        st_stream(V4{d[0].x, d[1].x, d[2].x, d[3].x}, reinterpret_cast<V4 *>(o1 + g));
        st_stream(V4{d[0].y, d[1].y, d[2].y, d[3].y}, reinterpret_cast<V4 *>(o2 + g));
      } else if (g < limit) {  // N = 2, odd tail: fewer than four points left
        for (int j = 0; j < 4 && g + j < limit; ++j) {
          o1[g + j] = d[j].x;
          o2[g + j] = d[j].y;
        }
      }
    });
  } else {
    T *const ampf = reinterpret_cast<T *>(lds);  // ROWS rows of `bins` <= N amplitudes: at most 4096 values
    static_for<RPT>([&](auto rc) {
      const int row = t + WG * rc;
      static_for<N>([&](auto k) {
        if (holds_row && k < bins)
          ampf[row * bins + k] = mag(x[rc][bitrev(k, LOG2N)]) * ((k == 0 || k == nyq) ? s_edge : s_mid);
      });
    });
    __syncthreads();
    const size_t out_base = (size_t)blockIdx.x * ROWS * (size_t)bins, out_limit = (size_t)batch * (size_t)bins;
    const int out_count = ROWS * bins;
    for (int i = t; i < out_count; i += WG)
      if (out_base + (size_t)i < out_limit) o1[out_base + (size_t)i] = ampf[i];
  }
}

// Tables of a FUSED cosine-sum window (createWindow of src/xform/fourier.ts:14-52 evaluated in the kernel
// instead of read from an N-value table): a thread's sample indices are n = (2 tid + e) + 2 TP q (+ N/2 in
// the N = 16384 kernel), so cos(f n), f = 2 pi / (N - 1), is one angle addition from a per-thread base and a
// per-q constant; w = k0 + c (k1 + k2 c) with (k0, k1, k2) = (a0 - a2, -a1, 2 a2)  [cos 2x = 2 c^2 - 1].
struct WinFused {
  const float *base;  // [TP][4]: cos, sin of f*(2 tid), cos, sin of f*(2 tid + 1)
  const float *step;  // [16][2]: cos, sin of f*2 TP q (q < 16); N = 16384 kernel: [32][2], the second half for + N/2
  float k0, k1, k2;
  // the coefficients above already carry the amplitude scale s_mid / 2 (a power of two: exact), so the
  // Hermitian split multiplies by nothing; edge_ratio = s_edge / s_mid fixes up the two bins that are
  // not doubled (DC, Nyquist)
  float edge_ratio;
};

// Fused body of spectrum() for real frames, one frame per row, via the packed-real
// identity: z[m] = x[2m] + i*x[2m+1] (a plain 2-wide view of the windowed frame),
// Z = FFT_M(z) with M = N/2, then for each pair (k, M-k)
//     E = (Z[k] + conj Z[M-k]) / 2,  O = (Z[k] - conj Z[M-k]) / (2i),  t = W_N^k * O,
//     X[k] = E + t,   X[M-k] = conj(E - t),          (k = 0 also yields X[M] = Nyquist)
// and only |X| (scaled as scaleAmplitude{One,Two}Sided, spectrum.ts:45-72) and
// optionally atan2 leave the chip.  Half the butterflies, half the LDS and half the
// loads per thread of running the complex kernel on (x, 0).
//   LOG2M = log2(N/2) >= 5.   twr[k] = e^{-2*pi*i*k/N}, 0 <= k <= M/2.
//   FAST: whole, 8-byte aligned frames (frame_len == N, even stride), one-sided
//         amplitude only -- the config-4 shape: 8-byte loads, no selects, and the phase /
//         mirror code stays out of the (VALU-issue bound) instruction stream.
//   !FAST: any frame_len >= 1 / alignment (scalar clamped loads + selects), phase and
//         two-sided output at run time.
//   PEAK: also reduce each frame to its SpectrumPeak (findPeak fused; `amp` may then be
//         null = peaks-only output, 16 B per frame instead of 2*(N/2+1) B).
//   WIN: 0 = rect, 1 = window table (N values), 2 / 3 = two- / three-term cosine-sum window FUSED (WinFused;
//        FAST f32 only): no table traffic, 3-4 packed instructions per pair of samples.
template <typename T, int LOG2M, bool FAST, int WIN, bool PEAK>
__global__ void __launch_bounds__(kPackedWG<LOG2M>)
spectrum_packed_kernel(const T *__restrict__ frames, const T *__restrict__ win, const WinFused wf, const long long frame_len,
                       const long long stride, const typename vec2<T>::type *__restrict__ tw,
                       const typename vec2<T>::type *__restrict__ twr, T *__restrict__ amp,
                       T *__restrict__ ph, const int two_sided, const T s_edge, const T s_mid,
                       PeakRec *__restrict__ peaks, const T freq_scale, const long long batch) {
  constexpr int LOG2E = packed_log2e(LOG2M);
  using TR = FftTraits<LOG2M, LOG2E>;
  constexpr int E = TR::E, TP = TR::TP, M = TR::N;
  static_assert(LOG2M >= 5, "packed path needs TP >= 2");
  constexpr bool HAS_WIN = WIN == 1;
  static_assert(WIN <= 1 || (FAST && sizeof(T) == 4 && LOG2E == 4), "fused windows: whole f32 frames");
#ifndef PDSP_PACKED_ADJ
#define PDSP_PACKED_ADJ 1  /* 0: round-1 split (bins tid + TP*q, dword stores in two directions): A/B builds */
#endif
  // adjacent-bin 8-byte non-temporal stores pay when a wave's stores cover whole cache lines of one row
  // (TP >= 32: N >= 1024).  Below that a wave spans many short rows and each lane's 8 / 16 bytes would be
  // a partial-line streaming write: measured 1.8 -> 0.8 TB/s at N = 64 (f32, staged path off) and
  // 36 -> 22 % in f64 -- those sizes keep round 1's split (dword stores that L2 merges).
  constexpr bool kAdj = FAST && LOG2E == 4 && TP >= 32 && PDSP_PACKED_ADJ;
  typedef T V4 __attribute__((ext_vector_type(4)));

  __shared__ cx<T> lds[TR::LDS_ELEMS];

  const int tid = (int)(threadIdx.x % TP);
  const int rloc = (int)(threadIdx.x / TP);
  const long long row_raw = (long long)blockIdx.x * TR::ROWS + rloc;
  const bool live = row_raw < batch;
  const long long row = uniform_row<TP>(live ? row_raw : batch - 1);
  cx<T> *const lrow = lds + rloc * TR::LROW;

  // buildFrame + applyWindow (spectrum.ts:36-43, :116-119) on load; m = tid + TP*q
  const T *const xrow = frames + (size_t)row * (size_t)stride;
  cx<T> x[E];
  // tables first (PDSP_TABLES_FIRST): twiddle bases, the split twiddle, the thread's window values
  constexpr bool kRegTw = (LOG2E == 4 && TP >= 16);
  std::conditional_t<kRegTw, RegTwiddles<T, LOG2M, LOG2E>, TableTwiddles<T, LOG2M, LOG2E>> twf;
  cx<T> twk0;
  V4 tb = V4{T(0), T(0), T(0), T(0)};  // kAdj: (W_N^(2 tid), W_N^(2 tid + 1)), the split's two bases
  V4 wb4 = V4{T(0), T(0), T(0), T(0)};  // fused window: the thread's base (cos, sin) pairs
  cx<T> wv[(HAS_WIN && PDSP_TABLES_FIRST) ? E : 1];
  if constexpr (PDSP_TABLES_FIRST) {
    if constexpr (kRegTw) twf.load(reinterpret_cast<const cx<T> *>(tw), tid);
    if constexpr (kAdj) tb = reinterpret_cast<const V4 *>(twr)[(unsigned)tid];
    else twk0 = reinterpret_cast<const cx<T> *>(twr)[(unsigned)tid];
    if constexpr (WIN >= 2) wb4 = reinterpret_cast<const V4 *>(wf.base)[(unsigned)tid];
    if constexpr (HAS_WIN && FAST) {
      const cx<T> *const w2 = reinterpret_cast<const cx<T> *>(win);
      static_for<E>([&](auto q) { wv[q] = (w2 + TP * q)[(unsigned)tid]; });
    } else if constexpr (HAS_WIN) {
      // the general variant takes a window at ANY alignment (a view one value into a tensor): two scalar loads
      static_for<E>([&](auto q) { wv[q] = cx<T>{(win + 2 * TP * q)[2 * (unsigned)tid], (win + 2 * TP * q + 1)[2 * (unsigned)tid]}; });
    }
    load_order_fence();
  }
  // (Buffer-form addressing -- one descriptor per row, scalar q offsets -- gains 4-5 % on the N = 16384
  // kernel, where it removes ~100 address instructions; here, A/B with tools/sweep.py, it measured +-0 at
  // N = 2048 / 8192 and -2 % at 4096, so the flat form stays.)
  if constexpr (FAST) {
    const cx<T> *const x2 = reinterpret_cast<const cx<T> *>(xrow);
    static_for<E>([&](auto q) { x[q] = ld_stream(x2 + TP * q + (unsigned)tid); });
  } else {
    // unconditional clamped loads + selects (no per-element branches); 1 <= frame_len <= N
    const int flen = (int)frame_len;
    static_for<E>([&](auto q) {
      const int i0 = 2 * (tid + TP * q);
      const int c0 = i0 < flen - 1 ? i0 : flen - 1, c1 = i0 + 1 < flen - 1 ? i0 + 1 : flen - 1;
      const T v0 = ld_stream(xrow + (unsigned)c0), v1 = ld_stream(xrow + (unsigned)c1);
      x[q] = cx<T>{i0 < flen ? v0 : T(0), i0 + 1 < flen ? v1 : T(0)};
    });
  }
  PDSP_STAMP_INIT();
  if constexpr (!kRegTw) twf.tw = reinterpret_cast<const cx<T> *>(tw);
  if constexpr (PDSP_TABLES_FIRST) {
    load_order_fence();
    if constexpr (HAS_WIN) static_for<E>([&](auto q) { x[q] = x[q] * wv[q]; });
    if constexpr (WIN >= 2) {
      static_assert(WIN < 2 || kAdj, "the fused windows' pre-scaled frames are split by the adjacent-bin code");
      const cx<T> cb{wb4.x, wb4.z}, sb{wb4.y, wb4.w};  // (e = 0, e = 1)
      const T k0 = wf.k0, k1 = wf.k1, k2 = wf.k2;        // pre-scaled by s_mid / 2 on the host (WinFused)
      const cx<T> kc = cb * k1, ks = sb * k1;             // two-term form: w = k0 + kc cos_q - ks sin_q
      static_for<E>([&](auto qc) {
        constexpr int q = qc;
        const T cq = wf.step[2 * q], sq = wf.step[2 * q + 1];  // wave-uniform: scalar loads
        if constexpr (WIN == 3) {
          const cx<T> c = cb * cq - sb * sq;                   // cos(f n), n = 2 (tid + TP q) + e
          x[q] = x[q] * (k0 + c * (k1 + k2 * c));
        } else {
          x[q] = x[q] * ((k0 + kc * cq) - ks * sq);
        }
      });
    }
  } else {
    // round-1 order: twiddle bases right behind the frame loads, consumed passes later
    if constexpr (kRegTw) twf.load(reinterpret_cast<const cx<T> *>(tw), tid);
    // W_N^tid: the Hermitian split needs W_N^(tid + TP*q) = twk0 * W32^q (N = 32*TP)
    if constexpr (kAdj) tb = reinterpret_cast<const V4 *>(twr)[(unsigned)tid];
    else twk0 = reinterpret_cast<const cx<T> *>(twr)[(unsigned)tid];
    static_assert(WIN <= 1 || PDSP_TABLES_FIRST, "fused windows are written for the tables-first order");
    if constexpr (HAS_WIN && FAST) {
      const cx<T> *const w2 = reinterpret_cast<const cx<T> *>(win);
      static_for<E>([&](auto q) { x[q] = x[q] * (w2 + TP * q)[(unsigned)tid]; });
    } else if constexpr (HAS_WIN) {
      static_for<E>([&](auto q) { x[q] = x[q] * cx<T>{(win + 2 * TP * q)[2 * (unsigned)tid], (win + 2 * TP * q + 1)[2 * (unsigned)tid]}; });
    }
  }
  // The adjacent-bin split takes a frame PRE-SCALED by g = s_mid / 2 (a power of two: exact) and multiplies by
  // nothing: the fused windows carry g in their coefficients (WinFused); the rect and table variants multiply here --
  // one packed multiply per pair of samples instead of four per pair of bins.  edge = s_edge / s_mid fixes up DC and
  // Nyquist, which are not doubled.
  const T edge = WIN >= 2 ? T(wf.edge_ratio) : s_edge / s_mid;
  if constexpr (kAdj && WIN <= 1) {
    const T g = T(0.5) * s_mid;
    static_for<E>([&](auto q) { x[q] = x[q] * g; });
  }
#ifdef PDSP_STAMPS
  pin_regs<T, E>(x);  // land the frame + window here
#endif
  PDSP_STAMP(0);  // frame + window loads

  fft_passes<T, LOG2M, true, LOG2E>(x, lrow, twf, tid);
  __syncthreads();
  PDSP_STAMP(1);  // all passes (detail in slots 8..)

  const int bins = (!FAST && two_sided) ? 2 * M : M + 1;
  const bool store_amp = live && (!PEAK || amp != nullptr);
  T *const arow = amp + (size_t)row * (size_t)bins;
  T *const prow = (!FAST && ph) ? ph + (size_t)row * (size_t)bins : nullptr;
  // (the ordered runs of spectrum_dif16k_kernel -- PeakBest::consider_ascending / _descending -- measured +-0 at
  // N = 4096 and -2.3 % at N = 1024 here, where a thread owns 16 bins, not 32: the order-independent rule stays)
  PeakBest<T> best{T(0), 0, cx<T>{T(0), T(0)}};
  T dc_amp = T(0);
  cx<T> dc_x{T(0), T(0)};
  if constexpr (kAdj) {
    // FAST split, round 2: a thread takes ADJACENT bins -- k0 = 2 (tid + TP q'), k1 = k0 + 1, q' < 4 -- so
    // that its outputs are the pairs (k0, k0+1) and (M-k0-1, M-k0): 8-byte non-temporal stores (the shape
    // that measured 6.0 vs 5.6 TB/s on the N = 16384 load / store skeleton, tools/kbench2) instead of
    // dword stores in two directions.  W_N^(k0 + e) = twr[2 tid + e] * W_16^q'  (N = 32 TP).
    typedef T V2 __attribute__((ext_vector_type(2)));
    const cx<T> tw0{tb.x, tb.y}, tw1{tb.z, tb.w};
    static_for<4>([&](auto qc) {
      constexpr int q = qc;
      const int k0 = 2 * (tid + TP * q);  // even, < M/2
      const cx<T> z0 = lrow[lds_pad(k0)], z1 = lrow[lds_pad(k0) + 1];          // k0 % 16 <= 14: same block of 16
      const cx<T> zp0 = lrow[lds_pad((M - k0) & (M - 1))], zp1 = lrow[lds_pad(M - k0 - 1)];
      const cx<T> w0 = mul_w32<T, 2 * q>(tw0), w1 = mul_w32<T, 2 * q>(tw1);
      // the frame came in pre-scaled by s_mid / 2: no multiplies here; DC and Nyquist (k0 = 0: not doubled) are
      // fixed up on xa0 / xb0 below
      const cx<T> e0 = z0 + conj(zp0), p0 = cmul(z0 - conj(zp0), w0);
      const cx<T> e1 = z1 + conj(zp1), p1 = cmul(z1 - conj(zp1), w1);
      cx<T> xa0 = add_mul_neg_i(e0, p0), xb0 = conj(add_mul_pos_i(e0, p0));        // X[k0], X[M - k0]
      const cx<T> xa1 = add_mul_neg_i(e1, p1), xb1 = conj(add_mul_pos_i(e1, p1));  // X[k0 + 1], X[M - k0 - 1]
      if constexpr (q == 0) {
        const T r = tid == 0 ? edge : T(1);  // k0 = 0 lives in thread 0's first pair
        xa0 = xa0 * r;
        xb0 = xb0 * r;
      }
      const T ma0 = mag(xa0), mb0 = mag(xb0), ma1 = mag(xa1), mb1 = mag(xb1);
      if constexpr (PEAK) {
        if (k0 == 0) {
          dc_amp = ma0;
          dc_x = xa0;
        } else {
          best.consider(ma0, k0, xa0);
        }
        best.consider(ma1, k0 + 1, xa1);
        best.consider(mb1, M - k0 - 1, xb1);
        best.consider(mb0, M - k0, xb0);
      }
      if (store_amp) {
        __builtin_nontemporal_store(V2{ma0, ma1}, reinterpret_cast<V2 *>(arow + (unsigned)k0));
        __builtin_nontemporal_store(V2{mb1, mb0}, reinterpret_cast<V2 *>(arow + (unsigned)(M - k0 - 1)));
      }
    });
    if (tid == 0) {  // the middle bin M/2 pairs with itself: X[M/2] = conj(Z[M/2]) scaled
      const cx<T> xm = conj(lrow[lds_pad(M / 2)]) * T(2);  // the pre-scaled frame carries s_mid / 2; one value, not a sum
      const T mm = mag(xm);
      if constexpr (PEAK) best.consider(mm, M / 2, xm);
      if (store_amp) st_rowtail(mm, arow + (unsigned)(M / 2));
    }
  }
  // pairs k = tid + TP*q, q < E/2 (k < M/2); k = M/2 is one more pair for tid == 0.
  // LDS: Z[k] at pad(tid) + q*cpad(TP); Z[M-k] at pad(M - tid) - q*cpad(TP); Z[M] == Z[0].
  const cx<T> *const zlo = lrow + lds_pad(tid);
  const cx<T> *const zhi = lrow + lds_pad(M - tid);
  const cx<T> *const zhi0 = lrow + lds_pad((M - tid) & (M - 1));
  static_for<kAdj ? 0 : E / 2 + 1>([&](auto qc) {
    constexpr int q = qc;
    if (q < E / 2 || tid == 0) {
      const int k = tid + TP * q, k2 = M - k;
      cx<T> z, zp;
      if constexpr (TP % 16 == 0) {
        z = zlo[cpad(TP * q)];
        zp = q == 0 ? zhi0[0] : *(zhi - cpad(TP * q));
      } else {
        z = lrow[lds_pad(k)];
        zp = lrow[lds_pad(k2 & (M - 1))];
      }
      cx<T> w;                                             // W_N^k
      if constexpr (LOG2E == 4) w = mul_w32<T, q>(twk0);   // N = 32*TP: W_N^(TP*q) = W32^q
      else w = (reinterpret_cast<const cx<T> *>(twr) + TP * q)[(unsigned)tid];
      // the amplitude scale rides on the 1/2 of the split; bins 0 (DC, from k = 0) and M (Nyquist, the
      // partner of k = 0) are not doubled.  A positive scale leaves the phases untouched.
      const T h = T(0.5) * ((k == 0) ? s_edge : s_mid);
      const cx<T> e = (z + conj(zp)) * h;                  // scaled E
      const cx<T> p = cmul(z - conj(zp), w) * h;           // scaled i * O * W, O = (Z - conj Zp)/(2i)
      const cx<T> xa = add_mul_neg_i(e, p);                // scaled X[k] = E + W*O
      const cx<T> xb = conj(add_mul_pos_i(e, p));          // scaled X[M-k] = conj(E - W*O)
      const T ma = mag(xa), mb = mag(xb);
      if constexpr (PEAK) {
        // the mirrored two-sided bins N-k carry identical values at larger indices, so
        // the strict-'>' search never selects them: bins 1..M decide
        if (k == 0) {
          dc_amp = ma;
          dc_x = xa;
        } else {
          best.consider(ma, k, xa);
        }
        if (k2 != k) best.consider(mb, k2, xb);
      }
      if (store_amp) {
        st_rowtail(ma, arow + (unsigned)k);
        if (k2 != k) st_rowtail(mb, arow + (unsigned)k2);
        if constexpr (!FAST) {
          if (two_sided && k != 0) {  // X[N-k] = conj X[k]
            st_rowtail(ma, arow + (unsigned)(2 * M - k));
            if (k2 != k) st_rowtail(mb, arow + (unsigned)(2 * M - k2));
          }
          if (prow) {
            st_rowtail(T(atan2(xa.y, xa.x)), prow + (unsigned)k);
            if (k2 != k) st_rowtail(T(atan2(xb.y, xb.x)), prow + (unsigned)k2);
            if (two_sided && k != 0) {
              st_rowtail(T(atan2(-xa.y, xa.x)), prow + (unsigned)(2 * M - k));
              if (k2 != k) st_rowtail(T(atan2(-xb.y, xb.x)), prow + (unsigned)(2 * M - k2));
            }
          }
        }
      }
    }
  });

  PDSP_STAMP(2);  // Hermitian split + stores issued
  if constexpr (PEAK) {
    // row-wide arg-max: butterflies inside the wave, then one LDS hop across the row's waves
    constexpr int WSPAN = TP < 64 ? TP : 64;
    static_for<ilog2(WSPAN)>([&](auto sc) {
      constexpr int off = WSPAN >> (sc + 1);
      PeakBest<T> o;
      o.v = __shfl_xor(best.v, off, 64);
      o.i = __shfl_xor(best.i, off, 64);
      o.x = cx<T>{__shfl_xor(best.x.x, off, 64), __shfl_xor(best.x.y, off, 64)};
      best.consider(o.v, o.i, o.x);
    });
    if constexpr (TP > 64) {
      constexpr int WAVES = TP / 64;  // waves per row; ROWS == 1 or WG/TP rows of WAVES waves
      __shared__ T pk_v[TR::WG / 64];
      __shared__ int pk_i[TR::WG / 64];
      __shared__ cx<T> pk_x[TR::WG / 64];
      const int wave = (int)threadIdx.x / 64;
      if ((threadIdx.x & 63) == 0) {
        pk_v[wave] = best.v;
        pk_i[wave] = best.i;
        pk_x[wave] = best.x;
      }
      __syncthreads();
      if (tid == 0) {
        static_for<WAVES - 1>([&](auto wc) {
          const int w = rloc * WAVES + wc + 1;
          best.consider(pk_v[w], pk_i[w], pk_x[w]);
        });
      }
    }
    if (live && tid == 0) {
      // nothing > 0 beyond DC: findPeak falls back to the global max, which is bin 0
      const bool none = best.i == 0;
      const cx<T> px = none ? dc_x : best.x;
      PeakRec r;
      r.index = best.i;
      r.frequency = (float)(T(best.i) * freq_scale);
      r.amplitude = (float)(none ? dc_amp : best.v);
      r.phase = (float)atan2(px.y, px.x);
      peaks[row] = r;
    }
  }
}

// Radix2Fft.forward(real input) (src/core/fft.ts:77-79: imaginary part taken as zero) on rows of N = 2M real
// samples through the packed-real identity, with the FULL complex spectrum written to both planes.
//   z[m] = x[2m] + i x[2m+1] (an 8-byte view of the row),  Z = FFT_M(z)  -- half the butterflies, LDS traffic and
//   twiddles of running the complex kernel on (x, 0), and the row arrives as 8-byte instead of 4-byte loads;
//   E[k] = (Z[k] + conj Z[M-k]) / 2,  O[k] = (Z[k] - conj Z[M-k]) / (2i)   (transforms of the even / odd samples)
//   X[k] = E[k] + W_N^k O[k],  X[k + M] = E[k] - W_N^k O[k],   0 <= k < M.
// Every thread forms X[k], X[k + M] for its OWN sixteen k = tid + TP q from (Z[k], Z[M-k]) read back from LDS: each
// pair (k, M-k) is split twice, once from either side -- six packed instructions per k -- so that both planes
// leave as the same forward unit-stride streams `row N + tid + TP q` the complex kernel writes.  (Round 2 tried the
// split that writes X[k] and its mirror X[N-k] = conj X[k] from one side: four store streams per plane in two
// directions, slower than the complex kernel at every size)
//   tw = radix table of the M-point transform (Tables::tw_half), twr[k] = W_N^k (Tables::twr), scale = 1.
// Rows must be aligned to one pair of samples.  In place row for row is fine: a workgroup loads its rows before it
// stores.  Dispatched for f64 rows (the drop-in's default arithmetic) of N = 8192 and N = 16384, the sizes where the
// complex kernel is short of registers or LDS (N = 8192: 50-62 -> 71-82 % of 8 TB/s by box; N = 16384: one pass
// instead of a four-step transform, 20 -> 63-65 %).  Below 8192 it measured +-: +1 ... +9 % on one box, -9 ... +2 %
// on another; in f32 both forms run at the box's copy ceiling (A/B 0.98 ... 1.01): neither is built.
template <typename T, int LOG2M>
__global__ void __launch_bounds__(kPackedWG<LOG2M>)
fft_real_kernel(const T *__restrict__ xin, T *__restrict__ ore, T *__restrict__ oim, const T scale,
                const typename vec2<T>::type *__restrict__ tw, const typename vec2<T>::type *__restrict__ twr,
                const long long batch) {
  constexpr int LOG2E = 4;
  using TR = FftTraits<LOG2M, LOG2E>;
  constexpr int E = TR::E, TP = TR::TP, M = TR::N;
  static_assert(LOG2M >= 8 && TP % 16 == 0, "N >= 512: register twiddle bases, constant LDS offsets");
  static_assert(packed_log2e(LOG2M) == LOG2E, "Tables::tw_half is built for sixteen points per thread");
  __shared__ cx<T> lds[TR::LDS_ELEMS];

  const int tid = (int)(threadIdx.x % TP);
  const int rloc = (int)(threadIdx.x / TP);
  const long long row_raw = (long long)blockIdx.x * TR::ROWS + rloc;
  const bool live = row_raw < batch;
  const long long row = uniform_row<TP>(live ? row_raw : batch - 1);
  cx<T> *const lrow = lds + rloc * TR::LROW;

  RegTwiddles<T, LOG2M, LOG2E> twf;
  cx<T> twk0;
  if constexpr (PDSP_TABLES_FIRST_C2C) {
    twf.load(reinterpret_cast<const cx<T> *>(tw), tid);
    twk0 = reinterpret_cast<const cx<T> *>(twr)[(unsigned)tid];
    load_order_fence();
  }
  const cx<T> *const x2 = reinterpret_cast<const cx<T> *>(xin + (size_t)row * (size_t)(2 * M));
  cx<T> x[E];
  static_for<E>([&](auto q) { x[q] = ld_stream(x2 + TP * q + (unsigned)tid); });
  if constexpr (!PDSP_TABLES_FIRST_C2C) {
    twf.load(reinterpret_cast<const cx<T> *>(tw), tid);
    twk0 = reinterpret_cast<const cx<T> *>(twr)[(unsigned)tid];  // W_N^tid; W_N^(tid + TP q) = twk0 * W_32^q (N = 32 TP)
  }

  fft_passes<T, LOG2M, true, LOG2E>(x, lrow, twf, tid);  // Z in LDS, natural order
  __syncthreads();

  // LDS: Z[k] at pad(tid) + q cpad(TP); Z[M-k] at pad(M - tid) - q cpad(TP); Z[M] == Z[0]
  const cx<T> *const zlo = lrow + lds_pad(tid);
  const cx<T> *const zhi = lrow + lds_pad(M - tid);
  const cx<T> *const zhi0 = lrow + lds_pad((M - tid) & (M - 1));
  T *const rre = ore + (size_t)row * (size_t)(2 * M), *const rim = oim + (size_t)row * (size_t)(2 * M);
  const T h = T(0.5) * scale;
  static_for<E>([&](auto qc) {
    constexpr int q = qc;
    const cx<T> z = zlo[cpad(TP * q)];
    const cx<T> zp = q == 0 ? zhi0[0] : *(zhi - cpad(TP * q));
    const cx<T> w = mul_w32<T, q>(twk0);
    const cx<T> s = z + conj(zp);
    const cx<T> p = cmul(z - conj(zp), w);             // 2i W^k O[k]
    const cx<T> xa = add_mul_neg_i(s, p) * h;          // X[k]     = (S - i P) / 2
    const cx<T> xb = add_mul_pos_i(s, p) * h;          // X[k + M] = (S + i P) / 2
    if (live) {
      st_stream(xa.x, rre + TP * q + (unsigned)tid);
      st_stream(xa.y, rim + TP * q + (unsigned)tid);
      st_stream(xb.x, rre + M + TP * q + (unsigned)tid);
      st_stream(xb.y, rim + M + TP * q + (unsigned)tid);
    }
  });
}

// ---- four-step path for N beyond the single-pass LDS limit --------------------
// N = N1 * N2 with N2 = the largest single-pass size and 2 <= N1 <= 16.  Input index
// n = n1*N2 + n2, output index k = k1 + N1*k2.
//   A  (this kernel)  for every column n2: the N1-point DFT over n1 (stride N2), times
//      W_N^{n2*k1}, written back in the same [k1][n2] layout.  One thread per column, so
//      both its loads and stores are unit-stride across the lanes.
//   B  fft_stockham_kernel on the N1 rows of N2 points, in place.
//   C  (fourstep_out_kernel) transposes [k1][k2] -> k1 + N1*k2 with the 1/N of the inverse,
//      or reduces to amplitude / phase rows for the spectrum path.
// W_N^m = twa[m >> 9] * twb[m & 511]: two small host-built f64 tables, one product.
template <typename T, int LOG2N1, bool REAL_IN, bool HAS_WIN>
__global__ void __launch_bounds__(256)
fourstep_cols_kernel(const T *__restrict__ re, const T *__restrict__ im, const T *__restrict__ win,
                     T *__restrict__ sre, T *__restrict__ sim, const cx<T> *__restrict__ twa,
                     const cx<T> *__restrict__ twb, int n2, long long in_stride, long long frame_len,
                     long long batch) {
  constexpr int N1 = 1 << LOG2N1;
  const int cblocks = n2 / 256;
  const long long b = blockIdx.x / cblocks;
  const int col = (int)(blockIdx.x % cblocks) * 256 + (int)threadIdx.x;
  if (b >= batch) return;
  const size_t n = (size_t)N1 * (size_t)n2;
  const T *const xr = re + (size_t)b * (size_t)in_stride;
  const T *const xi = REAL_IN ? nullptr : im + (size_t)b * (size_t)in_stride;
  const long long last = frame_len - 1;  // REAL_IN rows may be shorter than N (buildFrame zero-padding)
  cx<T> a[N1];
  static_for<N1>([&](auto q) {
    const long long i = (long long)q * n2 + col;
    if constexpr (REAL_IN) {
      T v = ld_stream(xr + (size_t)(i < last ? i : last));
      v = i <= last ? v : T(0);
      if constexpr (HAS_WIN) v *= win[i];
      a[q] = cx<T>{v, T(0)};
    } else {
      a[q] = cx<T>{ld_stream(xr + (size_t)i), ld_stream(xi + (size_t)i)};
    }
  });
  fft_reg<T, N1>(a);
  T *const orow = sre + (size_t)b * n, *const irow = sim + (size_t)b * n;
  static_for<N1>([&](auto k1) {
    cx<T> v = a[bitrev(k1, LOG2N1)];
    if constexpr (k1 > 0) {
      const unsigned m = (unsigned)col * (unsigned)k1;  // < N <= 2^18
      v = cmul(v, cmul(twa[m >> 9], twb[m & 511]));
    }
    orow[(size_t)k1 * n2 + col] = v.x;
    irow[(size_t)k1 * n2 + col] = v.y;
  });
}

// Pass C.  MODE 0: complex planes out (times `scale`); MODE 1: amplitude (+ phase) rows of
// spectrum(), bins = N/2+1 or N.  One thread per k2, N1 consecutive outputs each.
template <typename T, int LOG2N1, int MODE>
__global__ void __launch_bounds__(256)
fourstep_out_kernel(const T *__restrict__ sre, const T *__restrict__ sim, T *__restrict__ ore, T *__restrict__ oim,
                    int n2, T scale, int bins, int nyq, T s_edge, T s_mid, long long batch) {
  constexpr int N1 = 1 << LOG2N1;
  const int cblocks = n2 / 256;
  const long long b = blockIdx.x / cblocks;
  const int k2 = (int)(blockIdx.x % cblocks) * 256 + (int)threadIdx.x;
  if (b >= batch) return;
  const size_t n = (size_t)N1 * (size_t)n2;
  const T *const rr = sre + (size_t)b * n, *const ri = sim + (size_t)b * n;
  T vr[N1], vi[N1];
  static_for<N1>([&](auto k1) {
    vr[k1] = rr[(size_t)k1 * n2 + k2];
    vi[k1] = ri[(size_t)k1 * n2 + k2];
  });
  if constexpr (MODE == 0) {
    T *const o1 = ore + (size_t)b * n + (size_t)k2 * N1, *const o2 = oim + (size_t)b * n + (size_t)k2 * N1;
    static_for<N1>([&](auto k1) {
      o1[k1] = vr[k1] * scale;
      o2[k1] = vi[k1] * scale;
    });
  } else {
    T *const arow = ore + (size_t)b * (size_t)bins, *const prow = oim ? oim + (size_t)b * (size_t)bins : nullptr;
    static_for<N1>([&](auto k1) {
      const int k = k2 * N1 + k1;
      if (k < bins) {
        arow[k] = mag(cx<T>{vr[k1], vi[k1]}) * ((k == 0 || k == nyq) ? s_edge : s_mid);
        if (prow) prow[k] = T(atan2(vi[k1], vr[k1]));
      }
    });
  }
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
    if constexpr (PHASE) out[i] = T(atan2(b, a));
    else if constexpr (sizeof(T) == 8) out[i] = hypot(a, b);  // Math.hypot, fourier.ts:106 (range-safe)
    else out[i] = sqrt(a * a + b * b);                        // f32: |x| within ~1e-19 .. 1e19 (header)
  }
}

// Batched element-wise complex arithmetic on planar rows (src/math/complex.ts:26-197:
// add, sub, mul, div, conj, scale, mulScalar), so forward -> mul -> inverse pipelines
// (FFT-domain convolution, test/fluent/chain.test.ts:287-317) stay in HBM.  The second
// operand may be one row of `b_len` values broadcast over the batch (a filter response).
enum ComplexOp { kAdd = 0, kSub = 1, kMul = 2, kDiv = 3, kConj = 4, kScale = 5, kMulScalar = 6 };

template <typename T, int OP>
__device__ __forceinline__ void complex_op1(T ar, T ai, T br, T bi, T &orr, T &oi) {
  if constexpr (OP == kAdd) {
    orr = ar + br;
    oi = ai + bi;
  } else if constexpr (OP == kSub) {
    orr = ar - br;
    oi = ai - bi;
  } else if constexpr (OP == kMul || OP == kMulScalar) {  // (ac - bd) + i(ad + bc)
    orr = ar * br - ai * bi;
    oi = ar * bi + ai * br;
  } else if constexpr (OP == kDiv) {  // ((ac + bd) + i(bc - ad)) / (c^2 + d^2), complex.ts:150-161
    const T denom = br * br + bi * bi;
    orr = (ar * br + ai * bi) / denom;
    oi = (ai * br - ar * bi) / denom;
  } else if constexpr (OP == kConj) {
    orr = ar;
    oi = -ai;
  } else {  // kScale: real scalar in br
    orr = ar * br;
    oi = ai * br;
  }
}

template <typename T, int V>
__device__ __forceinline__ void ld_vec(const T *p, long long i, T (&v)[V]) {
  if constexpr (V == 4) {
    typedef T V4 __attribute__((ext_vector_type(4)));
    const V4 t = reinterpret_cast<const V4 *>(p)[i];
    v[0] = t.x, v[1] = t.y, v[2] = t.z, v[3] = t.w;
  } else {
    v[0] = p[i];
  }
}
template <typename T, int V>
__device__ __forceinline__ void st_vec(T *p, long long i, const T (&v)[V]) {
  if constexpr (V == 4) {
    typedef T V4 __attribute__((ext_vector_type(4)));
    reinterpret_cast<V4 *>(p)[i] = V4{v[0], v[1], v[2], v[3]};
  } else {
    p[i] = v[0];
  }
}

// No __restrict__: `out` may alias `a` (the fluent chain works in place) and `b` may be `out` too
// (chain.mul(chain)); every thread reads its own index of a and b before it writes that index.  A
// broadcast b (b_len < count) must not overlap `out`.
template <typename T, int OP, int V>  // V values per thread per step (4 = 16-byte accesses)
__global__ void __launch_bounds__(256)
complex_op_kernel(const T *are, const T *aim, const T *bre, const T *bim, T sre, T sim, T *ore, T *oim,
                  long long count, long long b_len) {
  constexpr bool kBinary = OP <= kDiv;
  const long long nvec = count / V;
  const long long step = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += step) {
    T ar[V], ai[V], br[V], bi[V], orr[V], oi[V];
    ld_vec<T, V>(are, i, ar);
    ld_vec<T, V>(aim, i, ai);
    if constexpr (kBinary) {
      const long long j = b_len == count ? i : ((i * V) % b_len) / V;  // b_len % V == 0 on this path
      ld_vec<T, V>(bre, j, br);
      ld_vec<T, V>(bim, j, bi);
    }
#pragma unroll
    for (int v = 0; v < V; ++v)
      complex_op1<T, OP>(ar[v], ai[v], kBinary ? br[v] : sre, kBinary ? bi[v] : sim, orr[v], oi[v]);
    st_vec<T, V>(ore, i, orr);
    st_vec<T, V>(oim, i, oi);
  }
}

// Fused spectrum for small frames (64 <= N <= 512), the staged counterpart of
// spectrum_packed_kernel's FAST variant: whole contiguous 16-byte aligned frames, one-sided
// amplitude.  The workgroup's 4096/M frames are one contiguous 8192-float chunk: staged in with
// 16-byte loads (window applied on the way), transformed and split as in the packed kernel, and
// the amplitude rows -- N/2+1 floats each, so no row starts aligned -- are staged out through an
// LDS row buffer and leave as one linear, fully coalesced stream.  Measured (tools/sweep.py):
// N = 64: 1.9 -> 5.6 TB/s, 128: 2.8 -> 5.9, 256: 3.9 -> 5.7, 512: 4.9 -> 5.4; from N = 1024 up the
// extra LDS round trip and workgroup-wide barrier cost more than they save (4.6 vs 5.0 TB/s).
template <typename T, int LOG2M, bool HAS_WIN>
__global__ void __launch_bounds__(256)
spectrum_staged_kernel(const T *__restrict__ frames, const T *__restrict__ win,
                       const typename vec2<T>::type *__restrict__ tw, const typename vec2<T>::type *__restrict__ twr,
                       T *__restrict__ amp, const T s_edge, const T s_mid, const long long batch) {
  using TR = FftTraits<LOG2M>;
  constexpr int M = TR::N, N = 2 * M, E = TR::E, TP = TR::TP, ROWS = TR::ROWS, WG = 256;
  static_assert(TR::WG == WG && TR::NP > 1 && LOG2M >= 5 && LOG2M <= 8, "staged spectrum: 64 <= N <= 512");
  constexpr int IN_FLOATS = ROWS * N;         // 8192
  constexpr int OUT_FLOATS = ROWS * (M + 1);  // 4096 + ROWS
  typedef T V4 __attribute__((ext_vector_type(4)));
  __shared__ cx<T> lds[TR::LDS_ELEMS];
  __shared__ T ampbuf[OUT_FLOATS];

  const int t = (int)threadIdx.x;
  const int tid = t % TP, rloc = t / TP;
  cx<T> *const lrow = lds + rloc * TR::LROW;
  const size_t in_base = (size_t)blockIdx.x * IN_FLOATS, in_limit = (size_t)batch * N;

  // frames -> LDS: float 4j.. of the chunk are points m, m+1 (m even) of local row (4j)/N
  static_for<IN_FLOATS / 4 / WG>([&](auto ic) {
    const int p = 4 * (t + WG * ic);
    size_t g = in_base + (size_t)p;
    g = g + 4 <= in_limit ? g : in_limit - 4;  // tail workgroup: re-read valid floats; dead rows never store
    V4 v = ld_stream(reinterpret_cast<const V4 *>(frames + g));
    if constexpr (HAS_WIN) v = v * reinterpret_cast<const V4 *>(win)[(p % N) / 4];  // applyWindow
    cx<T> *const d = lds + (p / N) * TR::LROW + lds_pad((p % N) / 2);
    d[0] = cx<T>{v.x, v.y};
    d[1] = cx<T>{v.z, v.w};
  });
  __syncthreads();
  cx<T> x[E];
  static_for<E>([&](auto q) { x[q] = lrow[lds_pad(tid + TP * q)]; });
  __syncthreads();  // the first pass scatters into the same buffer

  RegTwiddles<T, LOG2M> twf;
  twf.load(reinterpret_cast<const cx<T> *>(tw), tid);
  const cx<T> twk0 = reinterpret_cast<const cx<T> *>(twr)[(unsigned)tid];  // W_N^tid
  fft_passes<T, LOG2M, true>(x, lrow, twf, tid);
  __syncthreads();

  // Hermitian split (see spectrum_packed_kernel) into the LDS row buffer
  T *const arow = ampbuf + rloc * (M + 1);
  static_for<E / 2 + 1>([&](auto qc) {
    constexpr int q = qc;
    if (q < E / 2 || tid == 0) {
      const int k = tid + TP * q, k2 = M - k;
      const cx<T> z = lrow[lds_pad(k)], zp = lrow[lds_pad(k2 & (M - 1))];
      const cx<T> w = mul_w32<T, q>(twk0);  // W_N^k: N = 32*TP
      // the amplitude scale rides on the 1/2 of the split: DC and Nyquist (k = 0) are not doubled
      const T h = T(0.5) * ((k == 0) ? s_edge : s_mid);
      const cx<T> e = (z + conj(zp)) * h;
      const cx<T> p = cmul(z - conj(zp), w) * h;  // i * W * O
      arow[k] = mag(add_mul_neg_i(e, p));
      if (k2 != k) arow[k2] = mag(add_mul_pos_i(e, p));  // |conj(.)| = |.|
    }
  });
  __syncthreads();

  const size_t out_base = (size_t)blockIdx.x * OUT_FLOATS, out_limit = (size_t)batch * (M + 1);
  static_for<(OUT_FLOATS + WG - 1) / WG>([&](auto ic) {
    const int i = t + WG * ic;
    if (i < OUT_FLOATS && out_base + (size_t)i < out_limit) amp[out_base + (size_t)i] = ampbuf[i];
  });
}

// a * e^{-2*pi*i*NUM/64}, NUM compile-time
template <typename T, int NUM>
__device__ __forceinline__ cx<T> mul_w64(const cx<T> a) {
  static_assert(NUM >= 0 && NUM < 64, "one turn");
  if constexpr (NUM >= 32) {
    return -mul_w64<T, NUM - 32>(a);
  } else if constexpr (NUM % 2 == 0) {
    return mul_w32<T, NUM / 2>(a);
  } else {
    // cos(pi*j/32), j = 0..16
    constexpr double C[17] = {1.0,
                              0.99518472667219688624, 0.98078528040323044913, 0.95694033573220886494,
                              0.92387953251128675613, 0.88192126434835502971, 0.83146961230254523708,
                              0.77301045336273696081, 0.70710678118654752440, 0.63439328416364549822,
                              0.55557023301960222474, 0.47139673682599764856, 0.38268343236508977173,
                              0.29028467725446236764, 0.19509032201612826785, 0.09801714032956060199,
                              0.0};
    constexpr int lo = NUM <= 16 ? NUM : 32 - NUM;  // fold onto the first quarter turn
    constexpr double cq = NUM <= 16 ? C[lo] : -C[lo];
    constexpr double sq = C[16 - lo];
    constexpr T c = T(cq), s = T(-sq);  // W = cos - i sin
    return a.xx * cx<T>{c, s} + a.yy * cx<T>{-s, c};
  }
}

// forward / forwardComplex / inverse at N = 16384 (f32), re-cut so that TWO workgroups fit a CU.
// fft_stockham_kernel<14> needs 139 KB of LDS: one 1024-thread workgroup per CU whose load, compute
// and store phases cannot overlap anything (52 % of the HBM roofline, against 77 % at N = 4096).
// Here the transform is one radix-4 decimation-in-time step over four 4096-point transforms that
// the SAME 256 threads run back to back through one 4096-point LDS buffer:
//   one 16-byte load per plane gives x[4m..4m+3] = (s0[m], s1[m], s2[m], s3[m]);  Fj = FFT_4096(sj);
//   X[k + 4096q] = sum_j (-i)^{jq} W_16384^{jk} Fj[k]   in registers (thread tid holds k = tid + 256e
//   of all four).
// 34.8 KB of LDS and ~200 VGPRs per thread: two workgroups per CU, 16-byte loads, and one
// workgroup's HBM phases overlap the other's butterflies.
//   tw12 = radix table of the 4096-point transform; tws[k] = W_16384^k, k < 768.
template <typename T, int LOG2S, class LD, class ST>
__global__ void __launch_bounds__((1 << LOG2S) / 16, 2)
fft_split4_kernel(const LD ld, const ST st, const typename vec2<T>::type *__restrict__ tw12,
                  const typename vec2<T>::type *__restrict__ tws, const long long batch) {
  using TR = FftTraits<LOG2S>;
  constexpr int E = 16, TP = TR::TP, H = TR::N, N = 4 * H;
  static_assert(LD::kPlanar && ST::kPlanar, "planar rows");
  static_assert(TP >= 64 && TP % 64 == 0, "whole waves per row");
  typedef T V4 __attribute__((ext_vector_type(4)));
  __shared__ cx<T> lds[TR::LROW];

  const int tid = (int)threadIdx.x;
  const long long row = uniform_row<TP>((long long)blockIdx.x);
  if (row >= batch) return;

  cx<T> a[E], b[E], c[E], d[E];
  RegTwiddles<T, LOG2S> twf;
  const cx<T> *const twn = reinterpret_cast<const cx<T> *>(tws);
  cx<T> w1, w2, w3;
  if constexpr (PDSP_TABLES_FIRST) {
    twf.load(reinterpret_cast<const cx<T> *>(tw12), tid);
    w1 = twn[(unsigned)tid], w2 = twn[(unsigned)(2 * tid)], w3 = twn[(unsigned)(3 * tid)];
    load_order_fence();
  }
  if constexpr (LD::kPacked) {
    // points 4m .. 4m+3 = samples 8m .. 8m+7 of the frame: two 16-byte loads (+ two of the window table)
    const V4 *const f4 = reinterpret_cast<const V4 *>(ld.x + (size_t)row * (size_t)ld.stride);
    cx<T> wcs{T(1), T(0)};
    if constexpr (LD::kWin >= 2) wcs = reinterpret_cast<const cx<T> *>(ld.wb)[(unsigned)tid];
    static_for<E>([&](auto q) {
      // plain loads: the two halves of a lane's 32 bytes come from one cache line in two instructions, and
      // the second finds it in L1 (non-temporal: -8 %, tools/ab_long_spectrum.py at N = 32768)
      V4 r0 = f4[2 * (TP * q + (unsigned)tid)], r1 = f4[2 * (TP * q + (unsigned)tid) + 1];
      if constexpr (LD::kWin == 1) {
        const V4 *const w4 = reinterpret_cast<const V4 *>(ld.win);
        r0 = r0 * w4[2 * (TP * q + (unsigned)tid)];
        r1 = r1 * w4[2 * (TP * q + (unsigned)tid) + 1];
      }
      if constexpr (LD::kWin >= 2) {
        static_assert(LD::kWin < 2 || (sizeof(T) == 4 && TP == 256), "fused windows: f32 rows of 16384 points");
        const cx<T> c0 = cmul(wcs, reinterpret_cast<const cx<T> *>(ld.wq)[q]);  // wave-uniform: scalar loads
        cx<T> w[4];
        fused_window_pairs<T, LD::kWin == 3>(c0, reinterpret_cast<const cx<T> *>(ld.we), ld.k0, ld.k1, ld.k2, w);
        r0 = r0 * V4{w[0].x, w[0].y, w[1].x, w[1].y};
        r1 = r1 * V4{w[2].x, w[2].y, w[3].x, w[3].y};
      }
      a[q] = cx<T>{r0.x, r0.y};
      b[q] = cx<T>{r0.z, r0.w};
      c[q] = cx<T>{r1.x, r1.y};
      d[q] = cx<T>{r1.z, r1.w};
      // window table: a step holds four 16-byte loads; left alone hipcc hoists all sixteen steps' loads above the
      // first product (256 registers in flight beside the 128 of a..d: 10 spilled).  Four steps at a time.
      if constexpr (LD::kWin == 1 && q % 4 == 3) load_order_fence();
    });
  } else if constexpr (LD::kHasIm) {
    const V4 *const r4 = reinterpret_cast<const V4 *>(ld.plane_re() + (size_t)row * N);
    const V4 *const i4 = reinterpret_cast<const V4 *>(ld.plane_im() + (size_t)row * N);
    static_for<E>([&](auto q) {
      const V4 r = ld_stream(r4 + TP * q + (unsigned)tid);
      const V4 m = ld_stream(i4 + TP * q + (unsigned)tid);
      a[q] = cx<T>{r.x, m.x};
      b[q] = cx<T>{r.y, m.y};
      c[q] = cx<T>{r.z, m.z};
      d[q] = cx<T>{r.w, m.w};
    });
  } else {
    const V4 *const r4 = reinterpret_cast<const V4 *>(ld.plane_re() + (size_t)row * N);
    static_for<E>([&](auto q) {
      const V4 r = ld_stream(r4 + TP * q + (unsigned)tid);
      a[q] = cx<T>{r.x, T(0)};
      b[q] = cx<T>{r.y, T(0)};
      c[q] = cx<T>{r.z, T(0)};
      d[q] = cx<T>{r.w, T(0)};
    });
  }
  if constexpr (!PDSP_TABLES_FIRST) {
    twf.load(reinterpret_cast<const cx<T> *>(tw12), tid);
    w1 = twn[(unsigned)tid], w2 = twn[(unsigned)(2 * tid)], w3 = twn[(unsigned)(3 * tid)];
  }

  fft_passes<T, LOG2S, false>(a, lds, twf, tid);  // a[e] = F0[tid + TP*e]
  __syncthreads();                               // the buffer is reused by the next transform
  fft_passes<T, LOG2S, false>(b, lds, twf, tid);
  __syncthreads();
  fft_passes<T, LOG2S, false>(c, lds, twf, tid);
  __syncthreads();
  fft_passes<T, LOG2S, false>(d, lds, twf, tid);

  // radix-4 combine; W_N^{j(tid + TP*e)} = wj * W_64^{je}  (N = 64*TP)
  static_for<E>([&](auto ec) {
    constexpr int e = ec;
    const cx<T> t1 = cmul(b[e], mul_w64<T, e>(w1));
    const cx<T> t2 = cmul(c[e], mul_w64<T, (2 * e) % 64>(w2));
    const cx<T> t3 = cmul(d[e], mul_w64<T, (3 * e) % 64>(w3));
    const cx<T> s0 = a[e] + t2, s1 = a[e] - t2, s2 = t1 + t3, d13 = t1 - t3;
    st(row, 0 * H + TP * e, tid, s0 + s2);
    st(row, 1 * H + TP * e, tid, add_mul_neg_i(s1, d13));  // s1 - i (t1 - t3)
    st(row, 2 * H + TP * e, tid, s0 - s2);
    st(row, 3 * H + TP * e, tid, add_mul_pos_i(s1, d13));  // s1 + i (t1 - t3)
  });
}

// N = P * 16384 (P = 2, 4) in ONE pass over HBM, by workgroups that are siblings in one XCD's L2.
// Decimation in frequency by P in front of fft_split4_kernel's body: with S = 16384 and n < S,
//   y_h[n] = W_N^(h n) * sum_{p < P} x[n + p S] W_P^(p h),        X[P k + h] = FFT_S(y_h)[k],     h < P,
// so transform t is P independent 16384-point transforms -- P workgroups -- each of which reads ALL of x (P
// contiguous segments of S points) and writes every P-th output.  Taken alone that is P times the read traffic
// and partial-line writes; but an XCD's workgroups share its 4 MiB L2, and the dispatcher deals workgroup ids
// round-robin to the eight XCDs: sibling h of transform t sits at blockIdx (t / 8) * 8 P + 8 h + (t % 8), so the
// P siblings have the same id mod 8 -- the same XCD -- and leave the dispatcher one round apart.  They run the
// same instruction stream over the same addresses: the first sibling's loads bring a line into that L2, the others
// hit it; each line of the output is completed there by the P siblings' stores before it is written back.  HBM
// then sees 8 B read + 8 B written per sample -- one pass -- where the tile passes make two (32 B).
// (If the dispatcher did not pair them the results would still be right: every workgroup is self-sufficient.)
//   tw12 = radix table of the 4096-point transform; tws[k] = W_16384^k, k < 768 (fft_split4_kernel's);
//   W_N^m = twa[m >> 9] * twb[m & 511]; REAL: Radix2Fft.forward rows (no imaginary plane).
// PK > 0: the rows are PACKED real frames (LoadPackedFrames' layout: point m = (x[2m], x[2m+1]) of a frame of 2N
// samples, `pk.stride` samples apart) -- the one transform pass of spectrum() on frames of 2 * P * 16384 samples,
// split_amp_rows_kernel undoes the packing -- times the window: PK - 1 = 0 rect, 1 window table (in_im), 2 / 3
// two- / three-term cosine-sum window evaluated in registers (LoadPackedFrames' tables; segment p adds the angle
// 32768 f p = 2 * (8 * 2048 f)).  twa / twb then belong to the plan of 2N points: W_N^m = W_2N^(2m).
struct PairedPacked {
  long long stride;
  const float *wb, *wq, *we;
  float k0, k1, k2;
};
template <typename T, int LOG2P, bool REAL, int PK = 0>
__global__ void __launch_bounds__(256, 2)
fft_paired_kernel(const T *__restrict__ in_re, const T *__restrict__ in_im, T *__restrict__ out_re, T *__restrict__ out_im,
                  const typename vec2<T>::type *__restrict__ tw12, const typename vec2<T>::type *__restrict__ tws,
                  const cx<T> *__restrict__ twa, const cx<T> *__restrict__ twb, const T scale, const long long batch,
                  const PairedPacked pk) {
  static_assert(PK == 0 || (!REAL && sizeof(T) == 4 && LOG2P == 1), "packed frames: f32, two siblings");
  constexpr unsigned TSH = PK ? 1u : 0u;
  using TR = FftTraits<12>;
  constexpr int E = 16, TP = 256, H = 4096, S = 4 * H, P = 1 << LOG2P, N = P * S;
  static_assert(LOG2P == 1 || LOG2P == 2, "two or four siblings");
  typedef T V4 __attribute__((ext_vector_type(4)));
  __shared__ cx<T> lds[TR::LROW];

  const int tid = (int)threadIdx.x;
  const unsigned blk = (unsigned)__builtin_amdgcn_readfirstlane((int)blockIdx.x);
  const long long t = (long long)(blk / (8u * P)) * 8 + (blk & 7u);
  const unsigned h = (blk >> 3) & (unsigned)(P - 1);
  if (t >= batch) return;

  RegTwiddles<T, 12> twf;
  twf.load(reinterpret_cast<const cx<T> *>(tw12), tid);
  const cx<T> *const twn = reinterpret_cast<const cx<T> *>(tws);
  const cx<T> w1 = twn[(unsigned)tid], w2 = twn[(unsigned)(2 * tid)], w3 = twn[(unsigned)(3 * tid)];
  // W_N^(h n), n = 4 (tid + 256 q) + j:  wt[j] = W_N^(h (4 tid + j)) per thread, wq = W_N^(1024 h q) per q (wave-uniform)
  auto look = [&](unsigned m) { return cmul(twa[(m << TSH) >> 9], twb[(m << TSH) & 511]); };
  cx<T> wt[4];
  static_for<4>([&](auto j) { wt[j] = look(h * (unsigned)(4 * tid + j)); });
  cx<T> wcs{T(1), T(0)}, wseg{T(1), T(0)};  // PK >= 3: cs(8 f tid); cs(32768 f) = cs(8 * 2048 f)^2
  if constexpr (PK >= 3) {
    wcs = reinterpret_cast<const cx<T> *>(pk.wb)[(unsigned)tid];
    const cx<T> w8 = reinterpret_cast<const cx<T> *>(pk.wq)[8];
    wseg = cmul(w8, w8);
  }
  load_order_fence();

  const size_t base = (size_t)t * (size_t)N;
  cx<T> a[E], b[E], c[E], d[E];
  // The loads run AHEAD of the arithmetic by DEPTH steps of q, in slots of their own: left to itself hipcc waits
  // for each step's loads (vmcnt(0)) before it issues the next step's -- sixteen exposed round trips per workgroup,
  // on a kernel that is bound by one workgroup's critical path, not by HBM.
#ifndef PDSP_PAIRED_DEPTH2
#define PDSP_PAIRED_DEPTH2 5  /* steps of q in flight, two siblings (3: -6 %, 7: no better; tools/ab_paired.py) */
#endif
#ifndef PDSP_PAIRED_DEPTH4
#define PDSP_PAIRED_DEPTH4 2  /* four siblings, twice the registers per step (1: -4 %; 3 spills) */
#endif
  constexpr int DEPTH = P == 2 ? PDSP_PAIRED_DEPTH2 : PDSP_PAIRED_DEPTH4, SLOTS = DEPTH + 1;
  V4 l0[SLOTS][P], l1[SLOTS][P];  // planar: re / im of four points; packed: samples 8m .. 8m+3 / 8m+4 .. 8m+7
  auto issue = [&](auto qc, auto sc) {
    constexpr int q = qc, sl = sc;
    static_for<P>([&](auto pc) {
      // plain (not non-temporal) accesses on both sides: the siblings' reuse lives in the XCD's L2
      if constexpr (PK > 0) {
        // points 4m .. 4m+3 of segment p = samples 8m .. 8m+7 of the frame's half p: two 16-byte loads
        const size_t so = (size_t)t * (size_t)pk.stride + (size_t)pc * (2 * S) + 8 * (size_t)(TP * q + tid);
        l0[sl][pc] = *reinterpret_cast<const V4 *>(in_re + so);
        l1[sl][pc] = *(reinterpret_cast<const V4 *>(in_re + so) + 1);
      } else {
        const size_t o = base + (size_t)pc * S + 4 * (size_t)(TP * q + tid);
        l0[sl][pc] = *reinterpret_cast<const V4 *>(in_re + o);
        if constexpr (!REAL) l1[sl][pc] = *reinterpret_cast<const V4 *>(in_im + o);
      }
    });
  };
  static_for<DEPTH>([&](auto qc) { issue(qc, std::integral_constant<int, qc % SLOTS>{}); });
  static_for<E>([&](auto qc) {
    constexpr int q = qc, sl = q % SLOTS;
    if constexpr (q + DEPTH < E) {
      issue(std::integral_constant<int, q + DEPTH>{}, std::integral_constant<int, (q + DEPTH) % SLOTS>{});
      load_order_fence();
    }
    cx<T> x[P][4];
    static_for<P>([&](auto pc) {
      if constexpr (PK > 0) {
        V4 r0 = l0[sl][pc], r1 = l1[sl][pc];
        if constexpr (PK == 2) {
          const size_t wo = (size_t)pc * (2 * S) + 8 * (size_t)(TP * q + tid);
          r0 = r0 * *reinterpret_cast<const V4 *>(in_im + wo);
          r1 = r1 * *(reinterpret_cast<const V4 *>(in_im + wo) + 1);
        }
        if constexpr (PK >= 3) {
          cx<T> c0 = cmul(wcs, reinterpret_cast<const cx<T> *>(pk.wq)[q]);  // wave-uniform: scalar loads
          if constexpr (pc == 1) c0 = cmul(c0, wseg);
          cx<T> w[4];
          fused_window_pairs<T, PK == 4>(c0, reinterpret_cast<const cx<T> *>(pk.we), pk.k0, pk.k1, pk.k2, w);
          r0 = r0 * V4{w[0].x, w[0].y, w[1].x, w[1].y};
          r1 = r1 * V4{w[2].x, w[2].y, w[3].x, w[3].y};
        }
        x[pc][0] = cx<T>{r0.x, r0.y}, x[pc][1] = cx<T>{r0.z, r0.w}, x[pc][2] = cx<T>{r1.x, r1.y}, x[pc][3] = cx<T>{r1.z, r1.w};
      } else {
        const V4 r = l0[sl][pc];
        V4 m = V4{T(0), T(0), T(0), T(0)};
        if constexpr (!REAL) m = l1[sl][pc];
        x[pc][0] = cx<T>{r.x, m.x}, x[pc][1] = cx<T>{r.y, m.y}, x[pc][2] = cx<T>{r.z, m.z}, x[pc][3] = cx<T>{r.w, m.w};
      }
    });
    cx<T> y[4];
    static_for<4>([&](auto j) {
      if constexpr (P == 2) {
        y[j] = h ? x[0][j] - x[1][j] : x[0][j] + x[1][j];
      } else {
        // output h of the 4-point DFT over the segments: sum_p x_p (-i)^(p h)
        const cx<T> s02 = (h & 1) ? x[0][j] - x[2][j] : x[0][j] + x[2][j];
        const cx<T> s13 = (h & 1) ? x[1][j] - x[3][j] : x[1][j] + x[3][j];
        y[j] = h == 0 ? s02 + s13 : h == 2 ? s02 - s13 : h == 1 ? add_mul_neg_i(s02, s13) : add_mul_pos_i(s02, s13);
      }
    });
    if (h) {  // wave-uniform
      const cx<T> wq = look(h * 1024u * (unsigned)q);
      static_for<4>([&](auto j) { y[j] = cmul(y[j], cmul(wt[j], wq)); });
    }
    a[q] = y[0], b[q] = y[1], c[q] = y[2], d[q] = y[3];
  });

  fft_passes<T, 12, false>(a, lds, twf, tid);  // a[e] = F0[tid + TP*e]
  __syncthreads();                            // the buffer is reused by the next transform
  fft_passes<T, 12, false>(b, lds, twf, tid);
  __syncthreads();
  fft_passes<T, 12, false>(c, lds, twf, tid);
  __syncthreads();
  fft_passes<T, 12, false>(d, lds, twf, tid);

  // fft_split4_kernel's radix-4 combine; bin k of this sibling is X[P k + h]
  T *const ore = out_re + base + h, *const oim = out_im + base + h;
  auto put = [&](int k, cx<T> v) {
    v = v * scale;
    ore[(size_t)P * (size_t)(unsigned)k] = v.x;
    oim[(size_t)P * (size_t)(unsigned)k] = v.y;
  };
  static_for<E>([&](auto ec) {
    constexpr int e = ec;
    const cx<T> t1 = cmul(b[e], mul_w64<T, e>(w1));
    const cx<T> t2 = cmul(c[e], mul_w64<T, (2 * e) % 64>(w2));
    const cx<T> t3 = cmul(d[e], mul_w64<T, (3 * e) % 64>(w3));
    const cx<T> s0 = a[e] + t2, s1 = a[e] - t2, s2 = t1 + t3, d13 = t1 - t3;
    put(0 * H + TP * e + tid, s0 + s2);
    put(1 * H + TP * e + tid, add_mul_neg_i(s1, d13));
    put(2 * H + TP * e + tid, s0 - s2);
    put(3 * H + TP * e + tid, add_mul_pos_i(s1, d13));
  });
}

// The same cut one size down, as a radix-2 step: N = 8192 rows as two 4096-point transforms of the
// same 256 threads (x[2m], x[2m+1] arrive in one 2-wide load per plane) and
//   X[k] = E[k] + W_8192^k O[k],   X[k + 4096] = E[k] - W_8192^k O[k]    in registers.
// This is the f64 kernel of the largest single-pass f64 size (fft_stockham_kernel<double, 13> holds
// 139 KB of LDS, one workgroup per CU; here 69.6 KB, two), and with it the row pass of every f64
// four-step transform.   tws[k] = W_8192^k, k < 256.
template <typename T, class LD, class ST>
__global__ void __launch_bounds__(256, 2)
fft_split2_kernel(const LD ld, const ST st, const typename vec2<T>::type *__restrict__ tw12,
                  const typename vec2<T>::type *__restrict__ tws, const long long batch) {
  using TR = FftTraits<12>;
  constexpr int E = 16, TP = 256, H = 4096, N = 8192;
  static_assert(LD::kPlanar && ST::kPlanar, "planar rows");
  __shared__ cx<T> lds[TR::LROW];

  const int tid = (int)threadIdx.x;
  const long long row = uniform_row<TP>((long long)blockIdx.x);
  if (row >= batch) return;

  const cx<T> *const r2 = reinterpret_cast<const cx<T> *>(ld.plane_re() + (size_t)row * N);
  cx<T> a[E], b[E];
  // f64 complex rows: 64 data registers + 24 twiddle-base registers + the butterfly's temporaries do not
  // fit 256 VGPRs (81 spilled); reading the twiddles from the L2-resident table at each use instead
  // measured 4.36 -> 5.80 TB/s.  (f64 REAL rows no longer come here: with the bases in registers they spilled 37
  // registers in round 2's build; they run on fft_real_kernel, one 4096-point packed transform, at 6.5 TB/s.)
  constexpr bool kTableTw = sizeof(T) == 8 && LD::kHasIm;
  std::conditional_t<kTableTw, TableTwiddles<T, 12>, RegTwiddles<T, 12>> twf;
  cx<T> w1;
  if constexpr (PDSP_TABLES_FIRST) {
    if constexpr (!kTableTw) twf.load(reinterpret_cast<const cx<T> *>(tw12), tid);
    w1 = reinterpret_cast<const cx<T> *>(tws)[(unsigned)tid];
    load_order_fence();
  }
  if constexpr (LD::kHasIm) {
    const cx<T> *const i2 = reinterpret_cast<const cx<T> *>(ld.plane_im() + (size_t)row * N);
    static_for<E>([&](auto q) {
      const cx<T> r = ld_stream(r2 + TP * q + (unsigned)tid);
      const cx<T> m = ld_stream(i2 + TP * q + (unsigned)tid);
      a[q] = cx<T>{r.x, m.x};
      b[q] = cx<T>{r.y, m.y};
    });
  } else {
    static_for<E>([&](auto q) {
      const cx<T> r = ld_stream(r2 + TP * q + (unsigned)tid);
      a[q] = cx<T>{r.x, T(0)};
      b[q] = cx<T>{r.y, T(0)};
    });
  }
  if constexpr (kTableTw) twf.tw = reinterpret_cast<const cx<T> *>(tw12);
  if constexpr (!PDSP_TABLES_FIRST) {
    if constexpr (!kTableTw) twf.load(reinterpret_cast<const cx<T> *>(tw12), tid);
    w1 = reinterpret_cast<const cx<T> *>(tws)[(unsigned)tid];
  }

  fft_passes<T, 12, false>(a, lds, twf, tid);  // a[e] = E[tid + 256e]
  __syncthreads();                            // the buffer is reused by the second transform
  fft_passes<T, 12, false>(b, lds, twf, tid);  // b[e] = O[tid + 256e]

  static_for<E>([&](auto ec) {
    constexpr int e = ec;
    const cx<T> t = cmul(b[e], mul_w32<T, e>(w1));  // W_8192^(tid + 256e) = w1 * W_32^e
    st(row, TP * e, tid, a[e] + t);
    st(row, H + TP * e, tid, a[e] - t);
  });
}

#ifndef PDSP_DIF16K_WAVES
#define PDSP_DIF16K_WAVES 3  // three workgroups per CU (~160 VGPRs fit the 168 budget)
#endif
// spectrum() body for N = 16384 real frames, the config-4 shape (whole 8-byte-aligned frames, one-sided
// amplitude, optional fused findPeak), cut so that THREE frames fit a CU: the packed transform
// Z = FFT_8192(z), z[m] = x[2m] + i*x[2m+1], is one radix-2 decimation-in-FREQUENCY step over two
// 4096-point transforms that the same 256 threads run back to back through one 4096-point LDS buffer
// (spectrum_packed_kernel<13> needs 69.6 KB of LDS per frame and 5 full-size LDS round trips; here
// 34.8 KB and 2.5).  Round 1 made the cut by decimation in TIME (16-byte loads, Z[k] / Z[k+4096] in
// registers, dword stores): same arithmetic, 4.65-5.05 TB/s by box; this cut, with the fused Hann window,
// buffer addressing and the scale folded into the window: 5.6-6.0.
//   u[m] = z[m] + z[m+4096],  v[m] = (z[m] - z[m+4096]) * W_8192^m,   m < 4096;
//   U = FFT_4096(u) = the even bins Z[2k],  V = FFT_4096(v) = the odd bins Z[2k+1].
// The radix-2 step sits in front of the sub-transforms, which shapes the memory accesses at both ends
// (tools/kbench2: the load / store skeleton of this shape runs at 5.9-6.0 TB/s against 5.5-5.6 for the
// decimation-in-time shape, on a part whose 2:1 read:write copy tops out at 6.1-6.3):
//   * a thread needs z[m] and z[m+4096], m = tid + 256q: plain 8-byte loads at unit stride across the
//     lanes (32 per thread) instead of 16-byte loads that drag two sub-sequences along;
//   * the Hermitian split pairs Z[j] with Z[8192-j]: even with even, odd with odd, so thread tid ends up
//     owning bins (2k, 2k+1) and their mirrors (8191-2k, 8192-2k), k = tid + 256q, q < 8 -- ADJACENT
//     bins: 16 non-temporal 8-byte stores per thread instead of 33 dword stores in two directions.
//     (Non-temporal matters: plain 8-byte stores measured 5.4, non-temporal 6.0 TB/s; dword stores are
//     the other way round because a wave's 256 unaligned bytes are completed by its neighbours in L2.)
//   * partners: U[4096-k] and V[4095-k] live in the upper eight registers of thread 256-tid / 255-tid:
//     every thread parks its upper eight U and V values in LDS (32 KB, unit stride, no padding needed)
//     and reads eight of each back, lanes reversed.
// 34.8 KB of LDS (the sub-transforms' exchange buffer), three workgroups per CU.
//   tw12 = radix table of the 4096-point transform; twr[k] = W_16384^k, k <= 4096.
//   WIN: 0 = rect; 1 = window table (N values, 8-byte loads: 64 KB of L2 reads per frame -- measured
//   11 % of the kernel's time, 5.32 vs 5.97 TB/s); 2 / 3 = createWindow FUSED (fourier.ts:14-52): the
//   reference's windows are cosine sums a0 - a1 cos(f n) + a2 cos(2 f n), f = 2 pi / (N - 1), and a
//   thread's sample indices are n = (2 tid + e) + 512 q (+ 8192), so cos(f n) is one angle addition from
//   a per-thread base (one 16-byte load: cos / sin of f (2 tid + e), e = 0, 1) and a per-q constant
//   (wave-uniform: scalar loads): 3-4 packed instructions per pair of samples and no table traffic.
//   (WinFused above; values agree with the f64-built, f32-rounded table to ~2e-7 absolute.)
template <typename T, int WIN, bool PEAK>
__global__ void __launch_bounds__(256, PDSP_DIF16K_WAVES)
spectrum_dif16k_kernel(const T *__restrict__ frames, const T *__restrict__ win, const WinFused wf, const long long stride,
                       const typename vec2<T>::type *__restrict__ tw12,
                       const typename vec2<T>::type *__restrict__ twr, T *__restrict__ amp, const T s_edge,
                       const T s_mid, PeakRec *__restrict__ peaks, const T freq_scale, const long long batch) {
  constexpr bool HAS_WIN = WIN == 1;
  static_assert(WIN <= 1 || sizeof(T) == 4, "the fused window is the f32 path's (f64 keeps the f64-built table)");
  using TR = FftTraits<12>;
  constexpr int E = 16, TP = 256, H = 4096, M = 8192, Q = 2048;
  typedef T V4 __attribute__((ext_vector_type(4)));
  __shared__ cx<T> lds[TR::LROW];

  const int tid = (int)threadIdx.x;
  const long long row = uniform_row<TP>((long long)blockIdx.x);
  if (row >= batch) return;
  const cx<T> *const tw = reinterpret_cast<const cx<T> *>(tw12);

  // tables first (PDSP_TABLES_FIRST): twiddle bases, the two split bases, the thread's window values
  RegTwiddles<T, 12> twf;
  twf.load(tw, tid);
  // (W_16384^(2 tid), W_16384^(2 tid + 1)) in one 16-byte load: the first is W_8192^tid, base of the
  // decimation twiddle W_8192^(tid+256q) = wc0 * W32^q AND of the even bins' split twiddle W_16384^(2k);
  // the second is the base of the odd bins' W_16384^(2k+1) = ws1 * W32^q
  const V4 wcs = reinterpret_cast<const V4 *>(twr)[(unsigned)tid];
  const cx<T> wc0{wcs.x, wcs.y}, ws1{wcs.z, wcs.w};
  cx<T> wlo[HAS_WIN ? E : 1], whi[HAS_WIN ? E : 1];
  if constexpr (HAS_WIN) {
    const cx<T> *const w2 = reinterpret_cast<const cx<T> *>(win);
    static_for<E>([&](auto q) {
      wlo[q] = (w2 + TP * q)[(unsigned)tid];
      whi[q] = (w2 + H + TP * q)[(unsigned)tid];
    });
  }
  V4 wb4 = V4{T(0), T(0), T(0), T(0)};
  if constexpr (WIN >= 2) wb4 = reinterpret_cast<const V4 *>(wf.base)[(unsigned)tid];
  load_order_fence();

  // z[m], z[m + 4096], m = tid + 256q: 8-byte non-temporal loads, unit stride across the lanes.
  // Buffer form (PDSP_DIF_BUFFER): one descriptor for the frame, ONE per-lane offset register (8 tid) and
  // the 2048 q (+ 32768) part as a scalar offset -- with flat addresses hipcc built a 64-bit vector address
  // per load (v_add_co / s_nop / v_addc: the q offsets exceed the 13-bit immediate), ~100 of the kernel's
  // ~1300 issue slots, on a kernel whose vector issue is its busiest resource.
#ifndef PDSP_DIF_BUFFER
#define PDSP_DIF_BUFFER 1
#endif
  const cx<T> *const z2 = reinterpret_cast<const cx<T> *>(frames + (size_t)row * (size_t)stride);
  cx<T> a[E], b[E];
  if constexpr (PDSP_DIF_BUFFER && sizeof(T) == 4) {
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<cx<T> *>(z2), 0, 2 * M * (int)sizeof(T), 0x00020000);
    const int vo = tid * 8;
    static_for<E>([&](auto q) {
      a[q] = __builtin_bit_cast(cx<T>, __builtin_amdgcn_raw_buffer_load_b64(rs, vo, 2048 * q, 2 /* nt */));
      b[q] = __builtin_bit_cast(cx<T>, __builtin_amdgcn_raw_buffer_load_b64(rs, vo, 32768 + 2048 * q, 2));
    });
  } else {
    static_for<E>([&](auto q) {
      a[q] = ld_stream(z2 + TP * q + (unsigned)tid);
      b[q] = ld_stream(z2 + H + TP * q + (unsigned)tid);
    });
  }
  load_order_fence();
  // Every variant hands the sub-transforms a frame PRE-SCALED by g = s_mid / 2 (a power of two: exact), so that the
  // Hermitian split multiplies by nothing: the fused windows carry g in their coefficients (WinFused), the table
  // and rect variants multiply here (one packed multiply per pair of samples, against four per bin pair in the
  // split).  DC and Nyquist, which are not doubled, get edge = s_edge / s_mid on their two values.
  const T edge = WIN >= 2 ? T(wf.edge_ratio) : s_edge / s_mid;
  if constexpr (WIN <= 1) {
    const T g = T(0.5) * s_mid;
    static_for<E>([&](auto q) {
      if constexpr (HAS_WIN) {
        a[q] = a[q] * (wlo[q] * g);
        b[q] = b[q] * (whi[q] * g);
      } else {
        a[q] = a[q] * g;
        b[q] = b[q] * g;
      }
    });
  }
  if constexpr (WIN >= 2) {
    const cx<T> cb{wb4.x, wb4.z}, sb{wb4.y, wb4.w};  // (e = 0, e = 1)
    const T k0 = wf.k0, k1 = wf.k1, k2 = wf.k2;        // pre-scaled by s_mid / 2 on the host (WinFused)
    const cx<T> kc = cb * k1, ks = sb * k1;             // two-term form: w = k0 + kc cos_q - ks sin_q
    static_for<E>([&](auto qc) {
      constexpr int q = qc;
      const T cl = wf.step[2 * q], sl = wf.step[2 * q + 1], ch = wf.step[32 + 2 * q], sh = wf.step[32 + 2 * q + 1];
      if constexpr (WIN == 3) {
        const cx<T> c_lo = cb * cl - sb * sl, c_hi = cb * ch - sb * sh;  // cos(f n), n = 2m + e and + 8192
        a[q] = a[q] * (k0 + c_lo * (k1 + k2 * c_lo));
        b[q] = b[q] * (k0 + c_hi * (k1 + k2 * c_hi));
      } else {
        a[q] = a[q] * ((k0 + kc * cl) - ks * sl);
        b[q] = b[q] * ((k0 + kc * ch) - ks * sh);
      }
    });
  }
  // decimation in frequency: a <- u, b <- v
  static_for<E>([&](auto q) {
    const cx<T> d = a[q] - b[q];
    a[q] = a[q] + b[q];
    b[q] = cmul(d, mul_w32<T, q>(wc0));
  });

  fft_passes<T, 12, false>(a, lds, twf, tid);  // a[q] = U[tid + 256q] = Z[2(tid + 256q)]
  __syncthreads();                            // the buffer is reused by the second transform
  fft_passes<T, 12, false>(b, lds, twf, tid);  // b[q] = V[tid + 256q] = Z[2(tid + 256q) + 1]

  // upper halves (k >= 2048) -> LDS: U[k] at k - 2048, V[k] at 2048 + (k - 2048)
  __syncthreads();
  static_for<E / 2>([&](auto q) {
    lds[tid + TP * q] = a[q + E / 2];
    lds[Q + tid + TP * q] = b[q + E / 2];
  });
  const cx<T> umid = a[E / 2];  // thread 0: U[2048] = Z[4096], the bin that pairs with itself
  __syncthreads();

  T *const arow = amp + (size_t)row * (size_t)(M + 1);
  const bool store_amp = !PEAK || amp != nullptr;
  // best: the forward bins, visited in ascending order; bestm: the mirrored bins, visited in descending order
  PeakBest<T> best{T(0), 0, cx<T>{T(0), T(0)}}, bestm{T(0), 0, cx<T>{T(0), T(0)}};
  T dc_amp = T(0);
  cx<T> dc_x{T(0), T(0)};
  typedef T V2 __attribute__((ext_vector_type(2)));
  // amplitude row as a buffer: pairs (2k, 2k+1) at byte 8 tid + 2048 q, mirrors (8191-2k, 8192-2k) at
  // byte 4 (M-1) - 8 tid - 2048 q = [4 (M-1) - 2048*7 - 8 tid] + 2048 (7 - q): two per-lane offsets, scalar rest
  typedef unsigned U2s __attribute__((ext_vector_type(2)));
  const __amdgpu_buffer_rsrc_t ws = __builtin_amdgcn_make_buffer_rsrc(arow, 0, (M + 1) * (int)sizeof(T), 0x00020000);
  const int vo_lo = tid * 8, vo_hi = 4 * (M - 1) - 2048 * 7 - tid * 8;
  static_for<E / 2>([&](auto qc) {
    constexpr int q = qc;
    const int k = tid + TP * q;  // < 2048
    // even bins: Z[2k] = U[k] with Z[8192 - 2k] = U[4096 - k] (k = 0: DC and Nyquist, paired with itself)
    const cx<T> ze = a[q];
    cx<T> zpe;
    if constexpr (q == 0) zpe = tid == 0 ? ze : lds[Q - tid];
    else zpe = lds[Q - TP * q - tid];
    // odd bins: Z[2k+1] = V[k] with Z[8191 - 2k] = V[4095 - k]
    const cx<T> zo = b[q];
    const cx<T> zpo = lds[Q + (Q - 1) - TP * q - tid];
    const cx<T> we = mul_w32<T, q>(wc0);  // W_16384^(2k)
    const cx<T> wo = mul_w32<T, q>(ws1);  // W_16384^(2k+1)
    // the frame came in pre-scaled by s_mid / 2: no multiplies here; DC and Nyquist (k = 0: not doubled) are fixed
    // up on their two values below
    const cx<T> ee = ze + conj(zpe), pe = cmul(ze - conj(zpe), we);
    const cx<T> eo = zo + conj(zpo), po = cmul(zo - conj(zpo), wo);
    cx<T> xae = add_mul_neg_i(ee, pe);        // scaled X[2k]
    cx<T> xbe = conj(add_mul_pos_i(ee, pe));  // scaled X[8192 - 2k]
    const cx<T> xao = add_mul_neg_i(eo, po);        // scaled X[2k + 1]
    const cx<T> xbo = conj(add_mul_pos_i(eo, po));  // scaled X[8191 - 2k]
    if constexpr (q == 0) {
      const T r = tid == 0 ? edge : T(1);  // k = 0 lives in thread 0's first pair
      xae = xae * r;
      xbe = xbe * r;
    }
    const T mae = mag(xae), mbe = mag(xbe), mao = mag(xao), mbo = mag(xbo);
    if constexpr (PEAK) {
      if (k == 0) {
        dc_amp = mae;
        dc_x = xae;
      } else {
        best.consider_ascending(mae, 2 * k, xae);
      }
      best.consider_ascending(mao, 2 * k + 1, xao);
      bestm.consider_descending(mbe, M - 2 * k, xbe);
      bestm.consider_descending(mbo, M - 1 - 2 * k, xbo);
    }
    if (store_amp) {
      if constexpr (PDSP_DIF_BUFFER && sizeof(T) == 4) {
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(U2s, V2{mae, mao}), ws, vo_lo, 2048 * q, 2 /* nt */);
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(U2s, V2{mbo, mbe}), ws, vo_hi, 2048 * (7 - q), 2);
      } else {
        __builtin_nontemporal_store(V2{mae, mao}, reinterpret_cast<V2 *>(arow + (unsigned)(2 * k)));
        __builtin_nontemporal_store(V2{mbo, mbe}, reinterpret_cast<V2 *>(arow + (unsigned)(M - 1 - 2 * k)));
      }
    }
  });
  // the middle bin 4096 = 2*2048 pairs with itself: X[4096] = conj(Z[4096]) = conj(U[2048])
  if (tid == 0) {
    const cx<T> xm = conj(umid);
    const T mm = mag(xm) * T(2);  // the pre-scaled frame carries s_mid / 2; this bin is one value, not a sum of two
    if constexpr (PEAK) best.consider(mm, H, xm);
    if (store_amp) st_rowtail(mm, arow + (unsigned)H);
  }

  if constexpr (PEAK) {
    best.consider(bestm.v, bestm.i, bestm.x);  // the thread's two runs, by the order-independent rule
    // Wave-wide arg-max in two scalar reductions instead of six rounds of the full rule on four shuffled values:
    // the largest value, then the smallest index among the lanes that hold it; that lane -- unique, a bin has one
    // owner -- hands its record to LDS itself, so the complex bin is never shuffled.  (Values are never NaN here: a
    // NaN never enters a PeakBest.)
    T vmax = best.v;
    static_for<6>([&](auto sc) { vmax = fmaxf(vmax, __shfl_xor(vmax, 32 >> sc, 64)); });
    const bool cand = vmax > T(0) && best.v == vmax;
    int imin = cand ? best.i : 0x7fffffff;
    static_for<6>([&](auto sc) { imin = min(imin, __shfl_xor(imin, 32 >> sc, 64)); });
    __shared__ T pk_v[4];
    __shared__ int pk_i[4];
    __shared__ cx<T> pk_x[4];
    const int wave = tid / 64;
    if (cand && best.i == imin) {
      pk_v[wave] = best.v;
      pk_i[wave] = best.i;
      pk_x[wave] = best.x;
    } else if ((tid & 63) == 0 && !(vmax > T(0))) {  // nothing > 0 in this wave's bins
      pk_v[wave] = T(0);
      pk_i[wave] = 0;
      pk_x[wave] = cx<T>{T(0), T(0)};
    }
    __syncthreads();
    if (tid == 0) {
      best = PeakBest<T>{pk_v[0], pk_i[0], pk_x[0]};
      static_for<3>([&](auto wc) { best.consider(pk_v[wc + 1], pk_i[wc + 1], pk_x[wc + 1]); });
      const bool none = best.i == 0;
      const cx<T> px = none ? dc_x : best.x;
      PeakRec r;
      r.index = best.i;
      r.frequency = (float)(T(best.i) * freq_scale);
      r.amplitude = (float)(none ? dc_amp : best.v);
      r.phase = (float)atan2(px.y, px.x);
      peaks[row] = r;
    }
  }
}

// ---- tile passes: two passes for 2^15 <= N <= 2^18, three for 2^19 <= N <= 2^27 (f32) -----------
// N is cut into BALANCED factors of 64 ... 512 points instead of N1 * 16384, and every pass is one launch
// of tile_pass_kernel: a 256-thread workgroup owns TILE transforms of one factor, moves them between HBM
// and LDS in 128 ... 256-byte segments, runs them in LDS (fft_passes on the tile's LDS rows, 256/TP rows
// per round), and writes them back -- the transposition that a multi-pass transform needs happens in LDS.
//
// Two passes, N = A*B: n = n1*B + n2, k = k1 + A*k2,
//   X[k1 + A k2] = sum_n2 W_B^(n2 k2) * [ W_N^(n2 k1) * sum_n1 x[n1 B + n2] W_A^(n1 k1) ].
//   pass 1 (COLS): A-point transforms down the columns n2, times W_N^(n2 k1), written back in place of
//                  the matrix [k1][n2];
//   pass 2 (ROWS): B-point transforms along the rows k1, written TRANSPOSED to k1 + A*k2.
// Three passes, N = A*B*C: n = (n1*B + n2)*C + n3, k = k1 + A*(k2 + B*k3); with n' = n2*C + n3,
//   W_N^(n k) = W_A^(n1 k1) * W_N^(n' k1) * W_B^(n2 k2) * W_BC^(n3 k2) * W_C^(n3 k3):
//   pass 1 (COLS): A-point transforms over n1 (stride B*C) for every column n', times W_N^(n' k1), in place;
//   pass 2 (COLS): for every k1, B-point transforms over n2 (stride C) for every n3, times
//                  W_BC^(n3 k2) = W_N^(A n3 k2), written to row k2*A + k1 (so that pass 3 finds rows of
//                  consecutive k1 next to each other);
//   pass 3 (ROWS): C-point transforms along the rows, written transposed to (k2*A + k1) + A*B*k3.
// HBM traffic is 16 B per sample per pass (32 / 48 B for 16 algorithmic) where round 1's forms made three
// passes (N1 <= 16 columns / 16384-point rows / transposing copy) up to 2^18 and five above.
//
// Geometry of one launch (TileGeom): blockIdx -> (batch row b, block blk < nblk, tile < tiles);
//   COLS: element (p, j), p < L, j < TILE:  in[b*N + blk*in_blk + tile*TILE + p*in_stride + j]
//                                          out[b*N + blk*out_blk + tile*TILE + p*out_stride + j]
//         times W_N^(tmul * (tile*TILE + j) * p);
//   ROWS: TILE whole rows of L contiguous points from in[b*N + tile*TILE*L]; row i, output p goes to
//         out[b*N + tile*TILE + i + p*out_stride], times `scale`.
// LDS rows are LROWX = LROW + (2 - LROW) mod 8 elements apart: a lane quad writes one element to each of
// four rows, and four rows of LROW (a multiple of 8 for L >= 128) would land on the same banks.
//   tw = radix table of the L-point transform; W_N^m = twa[m >> 9] * twb[m & 511].
struct TileGeom {
  long long n;          // N: distance between batch rows on the output side (and on the input side unless in_batch)
  int nblk, tiles;      // blocks per transform, tiles per block
  long long in_blk, out_blk, in_stride, out_stride;
  unsigned tmul;        // twiddle exponent multiplier (1, or A in the middle pass of three)
  long long in_batch;   // distance between batch rows of the input (frames of spectrum(): the caller's stride,
                        // in real samples)
  unsigned tshift;      // 0, or 1 when twa / twb belong to a plan of 2N points (W_N^m = W_2N^(2m): the
                        // N/2-point transform of the packed-real spectrum path on its N-point plan's tables)
  // IN = 5 / 6: createWindow fused into the packed first pass (cosine-sum windows, fourier.ts:14-52, evaluated by
  // angle addition instead of being read back, 4 bytes per sample): with f = 2 pi / (frame length - 1) and
  // cs(t) = (cos t, sin t), sample n = 8 u + 2 in_stride SPI ic + e has
  //   cs(f n) = wa[u >> 9] * wb[u & 511] * wstep[ic] * we[e]     (wa[i] = cs(4096 f i), wb[j] = cs(8 f j),
  //   wstep[ic] = cs(2 f in_stride SPI ic), we[e] = cs(f e); all built in f64 on the host)
  // and w = k0 + k1 c (IN = 5) or k0 + c (k1 + k2 c) (IN = 6), c = cos(f n).
  const float *wa, *wb, *wstep, *we;
  float k0, k1, k2;
  // Scratch planes between the first two of three passes, TILE-MAJOR (0 = natural order everywhere): a strided
  // tile costs its READER (the column passes run at 4.9-5.7 TB/s, the row passes -- contiguous reads, strided
  // writes -- at 6.1-6.3), so the first pass writes column n' = n2 C + n3 of every row block to
  //   ((n3 >> perm_lt) * perm_b + n2) << perm_lt | (n3 & (2^perm_lt - 1))       (perm_lc = log2 C, perm_b = B)
  // i.e. the [B][2^perm_lt] tile the second pass needs becomes one contiguous chunk (in_tile = B << perm_lt apart,
  // in_stride = 2^perm_lt), and the first pass still writes segments of the same width at the same stride.
  int perm_lc, perm_lt, perm_b;  // output side; perm_lt == 0: no permutation
  long long in_tile;            // input side: distance between the origins of consecutive tiles (0: TILE)
};

// IN (first pass only; in_im then carries the window table or nothing):
//   0  complex planes
//   1  real rows: the first pass of Radix2Fft.forward / spectrum(), no imaginary plane
//   2  real rows times the window table (applyWindow on load, window value at the sample's index in the frame)
//   3  PACKED real rows: point m of the N-point transform is (x[2m], x[2m+1]) of a 2N-sample frame -- the
//      packed-real form of spectrum() (split_amp_rows_kernel undoes the packing)
//   4  packed real rows times the window table
//   5  packed real rows times a two-term cosine-sum window evaluated in registers (TileGeom::wa ...)
//   6  the same, three-term (blackman)
template <typename T, int LOG2L, int TILE, bool COLS, int IN = 0>
__global__ void __launch_bounds__(256)
tile_pass_kernel(const T *__restrict__ in_re, const T *__restrict__ in_im, T *__restrict__ out_re, T *__restrict__ out_im,
                 const typename vec2<T>::type *__restrict__ tw, const cx<T> *__restrict__ twa,
                 const cx<T> *__restrict__ twb, const TileGeom g, const T scale, const long long batch) {
  static_assert(IN == 0 || COLS, "real / packed / windowed input rides on the first (column) pass");
  constexpr bool REAL_IN = IN == 1 || IN == 2, WINDOWED = IN == 2 || IN == 4, PACKED = IN >= 3, FUSEDWIN = IN >= 5;
  static_assert(!FUSEDWIN || sizeof(T) == 4, "fused windows: f32");
  using TR = FftTraits<LOG2L>;
  constexpr int L = TR::N, E = TR::E, TP = TR::TP, RPR = 256 / TP, ROUNDS = TILE / RPR;
  static_assert(LOG2L >= 6 && LOG2L <= 9 && TILE % RPR == 0 && TILE % 4 == 0 && ROUNDS >= 1, "tile of whole rounds");
  constexpr int LROWX = TR::LROW + ((2 - TR::LROW % 8) + 8) % 8;
  constexpr int TS = TILE / 4, SPI = 256 / TS;  // threads per tile segment, segments per wave of accesses
  static_assert(L % SPI == 0, "whole accesses");
  typedef T V4 __attribute__((ext_vector_type(4)));
  __shared__ cx<T> lds[TILE * LROWX];

  const int t = (int)threadIdx.x;
  const long long per = (long long)g.nblk * g.tiles;
  const long long b = (long long)blockIdx.x / per;
  const int rem = (int)((long long)blockIdx.x % per);
  const int blk = rem / g.tiles, t0 = (rem % g.tiles) * TILE;
  if (b >= batch) return;
  const size_t base = (size_t)b * (size_t)g.n;
  const int seg = t / TS, j4 = (t % TS) * 4;

  // COLS: the inter-pass twiddles W_N^(tmul (t0 + j4 + j) p), p = seg + SPI ic.  Tables first (the lesson of
  // the single-pass kernels, PDSP_TABLES_FIRST): four two-level lookups per lane up here, their L2 latency
  // under the tile's HBM loads, and one product per iteration below -- instead of four gathers per iteration
  // between the transforms and the stores.  With q = tmul << tshift, J = t0 + j4:
  //   w(ic)  = W^(q p J) = W^(q seg J) * (W^(q SPI J))^ic          first element of the lane's four
  //   ws(ic) = W^(q p)   = W^(q seg)   * (W^(q SPI))^ic            step from element j to j + 1
  cx<T> tw_w{T(1), T(0)}, tw_wv{T(1), T(0)}, tw_s{T(1), T(0)}, tw_su{T(1), T(0)};
  if constexpr (COLS) {
    const unsigned q = g.tmul << g.tshift, J = (unsigned)(t0 + j4);
    const unsigned m0 = q * (unsigned)seg * J, mv = q * (unsigned)SPI * J;  // < N <= 2^27 (of the tables' plan)
    const unsigned s0 = q * (unsigned)seg, su = q * (unsigned)SPI;
    tw_w = cmul(twa[m0 >> 9], twb[m0 & 511]);
    tw_wv = cmul(twa[mv >> 9], twb[mv & 511]);
    tw_s = cmul(twa[s0 >> 9], twb[s0 & 511]);
    tw_su = cmul(twa[su >> 9], twb[su & 511]);
  }
  if constexpr (COLS) {
    // strided tile in: element (p, j) -> LDS row j, position p
    const size_t in_off = (size_t)blk * (size_t)g.in_blk +
                          (g.in_tile ? (size_t)(t0 / TILE) * (size_t)g.in_tile + (size_t)j4 : (size_t)(t0 + j4));  // index inside the frame
    const size_t ibase = (size_t)b * (size_t)g.in_batch + in_off;
    cx<T> wcs{T(1), T(0)};  // FUSEDWIN: cs(f n) of the lane's first sample at ic = 0
    if constexpr (FUSEDWIN) {
      const unsigned u = (unsigned)((in_off + (size_t)seg * (size_t)g.in_stride) >> 2);  // sample index / 8
      wcs = cmul(reinterpret_cast<const cx<T> *>(g.wa)[u >> 9], reinterpret_cast<const cx<T> *>(g.wb)[u & 511]);
    }
    static_for<L / SPI>([&](auto ic) {
      const int p = seg + SPI * ic;
      if constexpr (PACKED) {
        // four points = eight consecutive samples of the frame (32 bytes per lane, TILE * 8 per segment)
        const size_t fo = 2 * (in_off + (size_t)p * (size_t)g.in_stride);  // sample index inside the frame
        const T *const src = in_re + (size_t)b * (size_t)g.in_batch + fo;
        V4 r0 = ld_stream(reinterpret_cast<const V4 *>(src)), r1 = ld_stream(reinterpret_cast<const V4 *>(src) + 1);
        if constexpr (WINDOWED) {
          r0 = r0 * *reinterpret_cast<const V4 *>(in_im + fo);
          r1 = r1 * *(reinterpret_cast<const V4 *>(in_im + fo) + 1);
        }
        if constexpr (FUSEDWIN) {
          const cx<T> *const wst = reinterpret_cast<const cx<T> *>(g.wstep), *const wee = reinterpret_cast<const cx<T> *>(g.we);
          const cx<T> c0 = cmul(wcs, wst[ic]);  // wave-uniform step: scalar loads
          cx<T> w[4];
          fused_window_pairs<T, IN == 6>(c0, wee, g.k0, g.k1, g.k2, w);
          r0 = r0 * V4{w[0].x, w[0].y, w[1].x, w[1].y};
          r1 = r1 * V4{w[2].x, w[2].y, w[3].x, w[3].y};
        }
        cx<T> *const d = lds + j4 * LROWX + lds_pad(p);
        d[0 * LROWX] = cx<T>{r0.x, r0.y};
        d[1 * LROWX] = cx<T>{r0.z, r0.w};
        d[2 * LROWX] = cx<T>{r1.x, r1.y};
        d[3 * LROWX] = cx<T>{r1.z, r1.w};
        return;
      }
      const size_t gi = ibase + (size_t)p * (size_t)g.in_stride;
      V4 r = ld_stream(reinterpret_cast<const V4 *>(in_re + gi));
      V4 m = V4{T(0), T(0), T(0), T(0)};
      if constexpr (!REAL_IN) m = ld_stream(reinterpret_cast<const V4 *>(in_im + gi));
      // WINDOWED: in_im is the window table (N values), indexed like the frame
      if constexpr (WINDOWED) r = r * *reinterpret_cast<const V4 *>(in_im + in_off + (size_t)p * (size_t)g.in_stride);
      cx<T> *const d = lds + j4 * LROWX + lds_pad(p);
      d[0 * LROWX] = cx<T>{r.x, m.x};
      d[1 * LROWX] = cx<T>{r.y, m.y};
      d[2 * LROWX] = cx<T>{r.z, m.z};
      d[3 * LROWX] = cx<T>{r.w, m.w};
    });
  } else {
    // TILE whole rows, one contiguous chunk per plane
    const size_t cbase = base + (size_t)t0 * L;
    static_for<TILE * L / 4 / 256>([&](auto ic) {
      const int e = 4 * (t + 256 * ic);
      const V4 r = ld_stream(reinterpret_cast<const V4 *>(in_re + cbase + e));
      const V4 m = ld_stream(reinterpret_cast<const V4 *>(in_im + cbase + e));
      cx<T> *const d = lds + (e / L) * LROWX + lds_pad(e % L);  // 4 points never straddle a 16-block
      d[0] = cx<T>{r.x, m.x};
      d[1] = cx<T>{r.y, m.y};
      d[2] = cx<T>{r.z, m.z};
      d[3] = cx<T>{r.w, m.w};
    });
  }
  __syncthreads();

  const int tid = t % TP, rloc = t / TP;
  RegTwiddles<T, LOG2L> twf;
  twf.load(reinterpret_cast<const cx<T> *>(tw), tid);
  static_for<ROUNDS>([&](auto rc) {
    cx<T> *const lrow = lds + (rc * RPR + rloc) * LROWX;
    cx<T> x[E];
    static_for<E>([&](auto q) { x[q] = lrow[lds_pad(tid + TP * q)]; });
    __syncthreads();  // the first pass scatters into the same rows
    fft_passes<T, LOG2L, true>(x, lrow, twf, tid);  // natural order in LDS
  });
  __syncthreads();

  // strided tile out: element (p, j) -> out[... + p*out_stride + j]
  size_t ocol = (size_t)(t0 + j4);
  if constexpr (COLS) {
    if (g.perm_lt) {  // tile-major scratch (TileGeom): the lane's four columns stay adjacent (2^perm_lt >= 16)
      const unsigned np = (unsigned)(t0 + j4), n2 = np >> g.perm_lc, n3 = np & ((1u << g.perm_lc) - 1u);
      ocol = ((size_t)((n3 >> g.perm_lt) * (unsigned)g.perm_b + n2) << g.perm_lt) | (size_t)(n3 & ((1u << g.perm_lt) - 1u));
    }
  }
  const size_t obase = COLS ? base + (size_t)blk * (size_t)g.out_blk + ocol : base + ocol;
  static_for<L / SPI>([&](auto ic) {
    const int p = seg + SPI * ic;
    const cx<T> *const d = lds + j4 * LROWX + lds_pad(p);
    cx<T> v[4] = {d[0 * LROWX], d[1 * LROWX], d[2 * LROWX], d[3 * LROWX]};
    if constexpr (COLS) {
      // W_N^(tmul (t0 + j4 + j) p), j = 0..3: w(ic) for j = 0, then steps of ws(ic) (both advanced below)
      cx<T> w = tw_w;
      v[0] = cmul(v[0], w);
      static_for<3>([&](auto jc) {
        w = cmul(w, tw_s);
        v[jc + 1] = cmul(v[jc + 1], w);
      });
      if constexpr (ic + 1 < L / SPI) {
        tw_w = cmul(tw_w, tw_wv);
        tw_s = cmul(tw_s, tw_su);
      }
    } else {
      static_for<4>([&](auto jc) { v[jc] = v[jc] * scale; });
    }
    const size_t go = obase + (size_t)p * (size_t)g.out_stride;
    st_stream(V4{v[0].x, v[1].x, v[2].x, v[3].x}, reinterpret_cast<V4 *>(out_re + go));
    st_stream(V4{v[0].y, v[1].y, v[2].y, v[3].y}, reinterpret_cast<V4 *>(out_im + go));
  });
}

// tile_pass_kernel's COLS pass for a 512-point factor on tiles of 32 columns (plain tiles: 16 columns, 64-byte
// strided segments -- the width at which the memory system collapses, tools/kbench3).  One radix-2
// decimation-in-FREQUENCY step over two 256-point halves that the same 256 threads run back to back through the
// same 70 KB of LDS: a lane loads x[p] and x[p + 256] of its four columns, u[p] = x[p] + x[p + 256] goes to LDS,
// v[p] = (x[p] - x[p + 256]) W_512^p waits in 64 registers; FFT_256(u) = the even output rows k1 = 2k,
// FFT_256(v) = the odd ones k1 = 2k + 1.  Geometry (tiles = columns / 32), the tile-major scratch options and the
// inter-pass twiddles W_N^(tmul col k1) (hoisted two-level lookups) as tile_pass_kernel's COLS; IN = 0 complex
// planes, 1 real rows.  tw = radix table of the 256-point transform.
template <typename T, int IN>
__global__ void __launch_bounds__(256, 2)
tile_cols512_kernel(const T *__restrict__ in_re, const T *__restrict__ in_im, T *__restrict__ out_re, T *__restrict__ out_im,
                    const typename vec2<T>::type *__restrict__ tw, const cx<T> *__restrict__ twa,
                    const cx<T> *__restrict__ twb, const TileGeom g, const long long batch) {
  static_assert(IN == 0 || IN == 1, "complex planes or real rows");
  constexpr bool REAL_IN = IN == 1;
  using TR = FftTraits<8>;
  constexpr int TILE = 32, H = 256, E = TR::E, TP = TR::TP, RPR = 256 / TP, ROUNDS = TILE / RPR;
  constexpr int LROWX = TR::LROW + ((2 - TR::LROW % 8) + 8) % 8;
  constexpr int TS = TILE / 4, SPI = 256 / TS, NIT = H / SPI;  // 8 lanes per segment, 32 rows per access, 8 accesses
  typedef T V4 __attribute__((ext_vector_type(4)));
  __shared__ cx<T> lds[TILE * LROWX];

  const int t = (int)threadIdx.x;
  const long long per = (long long)g.nblk * g.tiles;
  const long long b = (long long)blockIdx.x / per;
  const int rem = (int)((long long)blockIdx.x % per);
  const int blk = rem / g.tiles, t0 = (rem % g.tiles) * TILE;
  if (b >= batch) return;
  const size_t base = (size_t)b * (size_t)g.n;
  const int seg = t / TS, j4 = (t % TS) * 4;
  const int tid = t % TP, rloc = t / TP;

  // tables first.  Output row k1 = 2 (seg + SPI ic) + o (o = 0: even half, 1: odd half); q = tmul << tshift, J = t0 + j4:
  //   first column of the lane's four: W^(q J k1) = W^(q J 2 seg) * (W^(q J 2 SPI))^ic   [* W^(q J) for o = 1]
  //   step to the next column:         W^(q k1)   = W^(q 2 seg)   * (W^(q 2 SPI))^ic     [* W^q     for o = 1]
  RegTwiddles<T, 8> twf;
  twf.load(reinterpret_cast<const cx<T> *>(tw), tid);
  const unsigned q = g.tmul << g.tshift, J = (unsigned)(t0 + j4);
  auto look = [&](unsigned m) { return cmul(twa[m >> 9], twb[m & 511]); };
  const cx<T> w_first0 = look(q * J * (unsigned)(2 * seg)), w_first_step = look(q * J * (unsigned)(2 * SPI));
  const cx<T> w_col0 = look(q * (unsigned)(2 * seg)), w_col_step = look(q * (unsigned)(2 * SPI));
  const cx<T> w_odd_first = look(q * J), w_odd_col = look(q);
  const cx<T> ws = look(((unsigned)(g.n >> 9) << g.tshift) * (unsigned)seg);  // W_512^seg; W_512^(seg + 32 ic) = ws W_16^ic
  cx<T> wv[NIT];
  static_for<NIT>([&](auto ic) {
    constexpr int i = ic;
    wv[i] = mul_w16<T, i>(ws);
  });

  // strided tile in: rows p and p + 256 of the lane's four columns; every load first
  const size_t in_off = (size_t)blk * (size_t)g.in_blk +
                        (g.in_tile ? (size_t)(t0 / TILE) * (size_t)g.in_tile + (size_t)j4 : (size_t)(t0 + j4));
  const size_t ibase = (size_t)b * (size_t)g.in_batch + in_off;
  V4 ra[NIT], ma[NIT], rb[NIT], mb[NIT];
  static_for<NIT>([&](auto ic) {
    const size_t gi = ibase + (size_t)(seg + SPI * ic) * (size_t)g.in_stride, gj = gi + (size_t)H * (size_t)g.in_stride;
    ra[ic] = ld_stream(reinterpret_cast<const V4 *>(in_re + gi));
    rb[ic] = ld_stream(reinterpret_cast<const V4 *>(in_re + gj));
    if constexpr (!REAL_IN) {
      ma[ic] = ld_stream(reinterpret_cast<const V4 *>(in_im + gi));
      mb[ic] = ld_stream(reinterpret_cast<const V4 *>(in_im + gj));
    }
  });
  cx<T> v[NIT][4];
  static_for<NIT>([&](auto ic) {
    const V4 z = V4{T(0), T(0), T(0), T(0)};
    const V4 r1 = ra[ic], r2 = rb[ic], m1 = REAL_IN ? z : ma[ic], m2 = REAL_IN ? z : mb[ic];
    const cx<T> a[4] = {cx<T>{r1.x, m1.x}, cx<T>{r1.y, m1.y}, cx<T>{r1.z, m1.z}, cx<T>{r1.w, m1.w}};
    const cx<T> c[4] = {cx<T>{r2.x, m2.x}, cx<T>{r2.y, m2.y}, cx<T>{r2.z, m2.z}, cx<T>{r2.w, m2.w}};
    cx<T> *const d = lds + j4 * LROWX + lds_pad(seg + SPI * ic);
    static_for<4>([&](auto jc) {
      v[ic][jc] = cmul(a[jc] - c[jc], wv[ic]);
      d[jc * LROWX] = a[jc] + c[jc];
    });
  });
  __syncthreads();

  size_t ocol = (size_t)(t0 + j4);
  if (g.perm_lt) {  // tile-major scratch (TileGeom)
    const unsigned np = (unsigned)(t0 + j4), n2 = np >> g.perm_lc, n3 = np & ((1u << g.perm_lc) - 1u);
    ocol = ((size_t)((n3 >> g.perm_lt) * (unsigned)g.perm_b + n2) << g.perm_lt) | (size_t)(n3 & ((1u << g.perm_lt) - 1u));
  }
  const size_t obase = base + (size_t)blk * (size_t)g.out_blk + ocol;
  static_for<2>([&](auto oc) {
    constexpr int o = oc;
    if constexpr (o == 1) {  // the odd half: v from the registers into the same rows
      __syncthreads();       // (the even half's tile has been read out)
      static_for<NIT>([&](auto ic) {
        cx<T> *const d = lds + j4 * LROWX + lds_pad(seg + SPI * ic);
        static_for<4>([&](auto jc) { d[jc * LROWX] = v[ic][jc]; });
      });
      __syncthreads();
    }
    static_for<ROUNDS>([&](auto rc) {
      cx<T> *const lrow = lds + (rc * RPR + rloc) * LROWX;
      cx<T> x[E];
      static_for<E>([&](auto qq) { x[qq] = lrow[lds_pad(tid + TP * qq)]; });
      __syncthreads();  // the first pass scatters into the same rows
      fft_passes<T, 8, true>(x, lrow, twf, tid);  // natural order in LDS
    });
    __syncthreads();
    // strided tile out: element (k, j) -> out[... + (2 k + o) * out_stride + j], times W_N^(tmul col k1)
    cx<T> tw_w = o == 1 ? cmul(w_first0, w_odd_first) : w_first0, tw_s = o == 1 ? cmul(w_col0, w_odd_col) : w_col0;
    static_for<NIT>([&](auto ic) {
      const int k1 = 2 * (seg + SPI * ic) + o;
      const cx<T> *const d = lds + j4 * LROWX + lds_pad(seg + SPI * ic);
      cx<T> y[4] = {d[0 * LROWX], d[1 * LROWX], d[2 * LROWX], d[3 * LROWX]};
      cx<T> w = tw_w;
      y[0] = cmul(y[0], w);
      static_for<3>([&](auto jc) {
        w = cmul(w, tw_s);
        y[jc + 1] = cmul(y[jc + 1], w);
      });
      if constexpr (ic + 1 < NIT) {
        tw_w = cmul(tw_w, w_first_step);
        tw_s = cmul(tw_s, w_col_step);
      }
      const size_t go = obase + (size_t)k1 * (size_t)g.out_stride;
      st_stream(V4{y[0].x, y[1].x, y[2].x, y[3].x}, reinterpret_cast<V4 *>(out_re + go));
      st_stream(V4{y[0].y, y[1].y, y[2].y, y[3].y}, reinterpret_cast<V4 *>(out_im + go));
    });
  });
}

// tile_pass_kernel's ROWS pass for a 512-point factor on tiles of 32 rows.  As a plain tile a 512-point factor
// fits 16 rows (70 KB of LDS), i.e. 64-byte output segments: 5.0 TB/s per pass, against 6.1 for the 32-row
// tiles of a 256-point factor (rocprofv3, profiles/r02_experiments/tile_pass_times_per_kernel.csv).  Here the
// factor is one radix-2 decimation-in-TIME step over two 256-point transforms that the same 256 threads run back
// to back through the same 70 KB (spectrum_dif16k_kernel's idea, the other way round): a lane's 16-byte loads
// bring x[p .. p+3] of a row, the even samples go to LDS, the odd ones wait in 64 registers;
//   E = FFT_256(x[2m]) is read back in OUTPUT layout (four rows per lane, 64 registers), O = FFT_256(x[2m+1])
//   follows through the same rows, and X[k] = E[k] + W_512^k O[k], X[k + 256] = E[k] - W_512^k O[k]
// leave as 128-byte segments (32 consecutive rows) at k and k + 256.  Geometry as tile_pass_kernel's ROWS
// (tiles = rows / 32); tw = radix table of the 256-point transform; W_512^k from the plan's two-level table.
template <typename T>
__global__ void __launch_bounds__(256, 2)
tile_rows512_kernel(const T *__restrict__ in_re, const T *__restrict__ in_im, T *__restrict__ out_re, T *__restrict__ out_im,
                    const typename vec2<T>::type *__restrict__ tw, const cx<T> *__restrict__ twa,
                    const cx<T> *__restrict__ twb, const TileGeom g, const T scale, const long long batch) {
  using TR = FftTraits<8>;
  constexpr int L = 512, H = 256, TILE = 32, E = TR::E, TP = TR::TP, RPR = 256 / TP, ROUNDS = TILE / RPR;
  constexpr int LROWX = TR::LROW + ((2 - TR::LROW % 8) + 8) % 8;
  constexpr int TS = TILE / 4, SPI = 256 / TS, NIN = TILE * L / 4 / 256, NOUT = H / SPI;  // 16 loads, 8 output rounds
  typedef T V4 __attribute__((ext_vector_type(4)));
  __shared__ cx<T> lds[TILE * LROWX];

  const int t = (int)threadIdx.x;
  const long long b = (long long)blockIdx.x / g.tiles;
  const int t0 = (int)((long long)blockIdx.x % g.tiles) * TILE;
  if (b >= batch) return;
  const size_t base = (size_t)b * (size_t)g.n;
  const int seg = t / TS, j4 = (t % TS) * 4;
  const int tid = t % TP, rloc = t / TP;

  // tables first: the 256-point transform's bases and W_512^seg (W_512^(seg + 32 ic) = W_512^seg * W_16^ic)
  RegTwiddles<T, 8> twf;
  twf.load(reinterpret_cast<const cx<T> *>(tw), tid);
  const unsigned m512 = ((unsigned)(g.n >> 9) << g.tshift) * (unsigned)seg;
  const cx<T> wseg = cmul(twa[m512 >> 9], twb[m512 & 511]);

  // TILE whole rows, one contiguous chunk per plane: lane t, load ic holds points p .. p+3 of row (t >> 7) + 2 ic,
  // p = 4 (t & 127): even samples 2m, 2m + 2 -> E's inputs m, m + 1 (m = 2 (t & 127)); odd samples -> O's
  const size_t cbase = base + (size_t)t0 * L;
  V4 r[NIN], mi[NIN];
  static_for<NIN>([&](auto ic) {
    const int e = 4 * (t + 256 * ic);
    r[ic] = ld_stream(reinterpret_cast<const V4 *>(in_re + cbase + e));
    mi[ic] = ld_stream(reinterpret_cast<const V4 *>(in_im + cbase + e));
  });
  const int me = 2 * (t & 127), lrow0 = t >> 7;
  cx<T> od[NIN][2];
  static_for<NIN>([&](auto ic) {
    cx<T> *const d = lds + (lrow0 + 2 * ic) * LROWX + lds_pad(me);  // me is even: me, me + 1 share a 16-block
    d[0] = cx<T>{r[ic].x, mi[ic].x};
    d[1] = cx<T>{r[ic].z, mi[ic].z};
    od[ic][0] = cx<T>{r[ic].y, mi[ic].y};
    od[ic][1] = cx<T>{r[ic].w, mi[ic].w};
  });
  __syncthreads();

  auto transform_rows = [&]() {
    static_for<ROUNDS>([&](auto rc) {
      cx<T> *const lrow = lds + (rc * RPR + rloc) * LROWX;
      cx<T> x[E];
      static_for<E>([&](auto q) { x[q] = lrow[lds_pad(tid + TP * q)]; });
      __syncthreads();  // the first pass scatters into the same rows
      fft_passes<T, 8, true>(x, lrow, twf, tid);  // natural order in LDS
    });
    __syncthreads();
  };
  transform_rows();
  // E in output layout: element (k, j), k = seg + SPI ic, rows j4 .. j4 + 3
  cx<T> ev[NOUT][4];
  static_for<NOUT>([&](auto ic) {
    const cx<T> *const d = lds + j4 * LROWX + lds_pad(seg + SPI * ic);
    static_for<4>([&](auto jc) { ev[ic][jc] = d[jc * LROWX]; });
  });
  __syncthreads();
  static_for<NIN>([&](auto ic) {
    cx<T> *const d = lds + (lrow0 + 2 * ic) * LROWX + lds_pad(me);
    d[0] = od[ic][0];
    d[1] = od[ic][1];
  });
  __syncthreads();
  transform_rows();

  // row i = j4 + jc, output k goes to out[base + t0 + i + k * out_stride]: 32 consecutive rows = 128-byte segments
  const size_t obase = base + (size_t)(t0 + j4);
  static_for<NOUT>([&](auto ic) {
    constexpr int i = ic;
    const int k = seg + SPI * i;
    const cx<T> *const d = lds + j4 * LROWX + lds_pad(k);
    const cx<T> w = mul_w16<T, i>(wseg);  // W_512^(seg + 32 i)
    cx<T> lo[4], hi[4];
    static_for<4>([&](auto jc) {
      const cx<T> tt = cmul(d[jc * LROWX], w);
      lo[jc] = (ev[i][jc] + tt) * scale;
      hi[jc] = (ev[i][jc] - tt) * scale;
    });
    const size_t go = obase + (size_t)k * (size_t)g.out_stride, gh = go + (size_t)H * (size_t)g.out_stride;
    st_stream(V4{lo[0].x, lo[1].x, lo[2].x, lo[3].x}, reinterpret_cast<V4 *>(out_re + go));
    st_stream(V4{lo[0].y, lo[1].y, lo[2].y, lo[3].y}, reinterpret_cast<V4 *>(out_im + go));
    st_stream(V4{hi[0].x, hi[1].x, hi[2].x, hi[3].x}, reinterpret_cast<V4 *>(out_re + gh));
    st_stream(V4{hi[0].y, hi[1].y, hi[2].y, hi[3].y}, reinterpret_cast<V4 *>(out_im + gh));
  });
}

// The tail of the packed-real long-frame spectrum path: Z = the M-point transform (natural order, planar)
// of z[m] = x[2m] + i x[2m+1], M = N/2.  One lane takes the pair (k, M - k), 0 <= k <= M/2:
//   X[k]     =      (Z[k] + conj(Z[M-k])) / 2 - i W_N^k (Z[k] - conj(Z[M-k])) / 2
//   X[M - k] = conj((Z[k] + conj(Z[M-k])) / 2 + i W_N^k (Z[k] - conj(Z[M-k])) / 2),   Z[M] = Z[0]
// -- the same split spectrum_packed_kernel runs from LDS -- then magnitude + scaleAmplitude{One,Two}Sided
// (+ phase), spectrum.ts:45-72, :121-131; two-sided rows get X[N - k] = conj(X[k]) as well.
//   W_N^k = twa[k >> 9] * twb[k & 511] (the N-point plan's two-level table); bins = M + 1 or N.
// One lane takes k = 4i .. 4i+3 (M/8 lanes per frame, M >= 2048): 16-byte loads of Z[4i ..] and of
// Z[M-4i-4 .. M-4i-1] (the mirrors of 4i+1 .. 4i+3; Z[M-4i] is one more dword, Z[0] for lane 0), 16-byte
// stores of bins 4i .. 4i+3 and M-4i-3 .. M-4i (rows of M + 1 bins: 4-byte aligned); lane 0 adds bin M/2.
template <typename T>
__global__ void __launch_bounds__(256)
split_amp_rows_kernel(const T *__restrict__ zre, const T *__restrict__ zim, T *__restrict__ amp, T *__restrict__ ph,
                      const cx<T> *__restrict__ twa, const cx<T> *__restrict__ twb, const int M, const int bins,
                      const T s_edge, const T s_mid, const long long batch) {
  typedef T V4 __attribute__((ext_vector_type(4)));
  typedef T V4u __attribute__((ext_vector_type(4), aligned(4)));  // rows of M + 1 bins start anywhere
  const int chunks = M / 8 / 256;
  const long long b = (long long)blockIdx.x / chunks;
  const int i = (int)((long long)blockIdx.x % chunks) * 256 + (int)threadIdx.x;
  if (b >= batch) return;
  const T *const re = zre + (size_t)b * (size_t)M, *const im = zim + (size_t)b * (size_t)M;
  T *const arow = amp + (size_t)b * (size_t)bins;
  T *const prow = ph ? ph + (size_t)b * (size_t)bins : nullptr;
  const bool two = bins > M + 1;
  const int k0 = 4 * i;
  const V4 fr = ld_stream(reinterpret_cast<const V4 *>(re + k0)), fi = ld_stream(reinterpret_cast<const V4 *>(im + k0));
  const V4 mr = ld_stream(reinterpret_cast<const V4 *>(re + (M - k0 - 4)));
  const V4 mi = ld_stream(reinterpret_cast<const V4 *>(im + (M - k0 - 4)));
  const int km = (M - k0) & (M - 1);
  const cx<T> m0{ld_stream(re + km), ld_stream(im + km)};
  const cx<T> zh{re[M / 2], im[M / 2]};  // lane 0's extra bin (every lane loads it: no branch around a load)
  const cx<T> wa = twa[k0 >> 9];
  const V4 wb01 = *reinterpret_cast<const V4 *>(twb + (k0 & 511)), wb23 = *(reinterpret_cast<const V4 *>(twb + (k0 & 511)) + 1);
  const cx<T> z[4] = {cx<T>{fr.x, fi.x}, cx<T>{fr.y, fi.y}, cx<T>{fr.z, fi.z}, cx<T>{fr.w, fi.w}};
  const cx<T> zp[4] = {m0, cx<T>{mr.w, mi.w}, cx<T>{mr.z, mi.z}, cx<T>{mr.y, mi.y}};
  const cx<T> wb[4] = {cx<T>{wb01.x, wb01.y}, cx<T>{wb01.z, wb01.w}, cx<T>{wb23.x, wb23.y}, cx<T>{wb23.z, wb23.w}};
  cx<T> xa[4], xb[4];
  T ma[4], mb[4];
  static_for<4>([&](auto j) {
    const cx<T> w = cmul(wa, wb[j]);
    const cx<T> e = (z[j] + conj(zp[j])) * T(0.5), p = cmul(z[j] - conj(zp[j]), w) * T(0.5);
    xa[j] = add_mul_neg_i(e, p), xb[j] = conj(add_mul_pos_i(e, p));  // X[k], X[M - k]
    const T sc = (j == 0 && i == 0) ? s_edge : s_mid;                // k = 0 <-> bins 0 and M: DC and Nyquist
    ma[j] = mag(xa[j]) * sc, mb[j] = mag(xb[j]) * sc;
  });
  *reinterpret_cast<V4u *>(arow + k0) = V4{ma[0], ma[1], ma[2], ma[3]};
  *reinterpret_cast<V4u *>(arow + (M - k0 - 3)) = V4{mb[3], mb[2], mb[1], mb[0]};
  if (prow) {
    *reinterpret_cast<V4u *>(prow + k0) =
        V4{T(atan2(xa[0].y, xa[0].x)), T(atan2(xa[1].y, xa[1].x)), T(atan2(xa[2].y, xa[2].x)), T(atan2(xa[3].y, xa[3].x))};
    *reinterpret_cast<V4u *>(prow + (M - k0 - 3)) =
        V4{T(atan2(xb[3].y, xb[3].x)), T(atan2(xb[2].y, xb[2].x)), T(atan2(xb[1].y, xb[1].x)), T(atan2(xb[0].y, xb[0].x))};
  }
  if (two) {  // X[N - k] = conj(X[k]) (k > 0), X[M + k] = conj(X[M - k])
    static_for<4>([&](auto j) {
      const int k = k0 + j;
      if (k > 0) {
        arow[2 * M - k] = ma[j];
        arow[M + k] = mb[j];
        if (prow) {
          prow[2 * M - k] = T(atan2(-xa[j].y, xa[j].x));
          prow[M + k] = T(atan2(-xb[j].y, xb[j].x));
        }
      }
    });
  }
  if (i == 0) {  // X[M/2] = conj(Z[M/2]) pairs with itself
    arow[M / 2] = mag(zh) * s_mid;
    if (prow) prow[M / 2] = T(atan2(-zh.y, zh.x));
    if (two) {
      arow[M + M / 2] = mag(zh) * s_mid;
      if (prow) prow[M + M / 2] = T(atan2(zh.y, zh.x));
    }
  }
}

// ---- general four-step path: N = N1 * N2 with 32 <= N1 <= N2 = the largest single-pass size ---
// Input index n = n1*N2 + n2, output index k = k1 + N1*k2; both factors run on the row kernels,
// so the data is transposed between them (Bailey's four-step):
//   1  [n1][n2] -> [n2][n1]                       (this kernel; window / zero padding of real frames here)
//   2  N1-point rows, in place                     (fft_stockham_kernel)
//   3  [n2][k1] -> [k1][n2], times W_N^{n2*k1}     (this kernel, TW)
//   4  N2-point rows, in place
//   5  [k1][k2] -> [k2][k1] = natural order        (this kernel; 1/N of the inverse, or AMP: the
//      amplitude / phase rows of spectrum())
// Five passes over HBM -- a completeness path for long one-shot transforms, not a bench configuration.
// One 256-thread workgroup moves one 32x32 tile of both planes through LDS, so the loads are
// unit-stride along the input rows and the stores along the output rows.
//   in: [batch][R][C] planar, element n = r*C + c; reads as zero for n >= used (real frames shorter
//   than N); in_im may be null (real input); win (N values) may be null.
template <typename T, bool TW, bool AMP>
__global__ void __launch_bounds__(256)
bigfft_transpose_kernel(const T *__restrict__ in_re, const T *__restrict__ in_im, const T *__restrict__ win,
                        const long long used, const long long in_stride,
                        const typename vec2<T>::type *__restrict__ twa, const typename vec2<T>::type *__restrict__ twb,
                        T *__restrict__ o1, T *__restrict__ o2, const int R, const int C, const T scale, const int bins,
                        const int nyq, const T s_edge, const T s_mid) {
  __shared__ T tre[32][33], tim[32][33];
  const int tiles_c = C / 32, tiles_r = R / 32;
  const long long blk = (long long)blockIdx.x;
  const long long b = blk / ((long long)tiles_r * tiles_c);
  const int t = (int)(blk % ((long long)tiles_r * tiles_c));
  const int tr = t / tiles_c, tc = t % tiles_c;
  const int tx = (int)threadIdx.x & 31, ty = (int)threadIdx.x >> 5;
  const T *const bre = in_re + (size_t)b * (size_t)in_stride;
  const T *const bim = in_im ? in_im + (size_t)b * (size_t)in_stride : nullptr;
  static_for<4>([&](auto ic) {
    const int r = tr * 32 + ty + 8 * ic, c = tc * 32 + tx;
    const long long n = (long long)r * C + c;
    const long long nc = n < used ? n : used - 1;  // unconditional clamped loads + select
    cx<T> v{bre[nc], bim ? bim[nc] : T(0)};
    if (win) v = v * win[n];
    if (n >= used) v = cx<T>{T(0), T(0)};
    if constexpr (TW) {
      const unsigned m = (unsigned)r * (unsigned)c;  // < N <= 2^28
      v = cmul(v, cmul(reinterpret_cast<const cx<T> *>(twa)[m >> 9], reinterpret_cast<const cx<T> *>(twb)[m & 511]));
    }
    tre[ty + 8 * ic][tx] = v.x;
    tim[ty + 8 * ic][tx] = v.y;
  });
  __syncthreads();
  static_for<4>([&](auto ic) {
    const int orow = tc * 32 + ty + 8 * ic, ocol = tr * 32 + tx;  // output matrix is [C][R]
    const cx<T> v{tre[tx][ty + 8 * ic], tim[tx][ty + 8 * ic]};
    const long long k = (long long)orow * R + ocol;
    if constexpr (AMP) {
      if (k < bins) {
        const size_t o = (size_t)b * (size_t)bins + (size_t)k;
        o1[o] = mag(v) * ((k == 0 || k == nyq) ? s_edge : s_mid);
        if (o2) o2[o] = T(atan2(v.y, v.x));
      }
    } else {
      const size_t o = (size_t)b * (size_t)R * (size_t)C + (size_t)k;
      o1[o] = v.x * scale;
      o2[o] = v.y * scale;
    }
  });
}

// SpectrumPeak per frame from stored amplitude (and phase) rows: the fallback of the fused
// PEAK path for sizes / alignments the packed kernel does not take.  One workgroup per row.
template <typename T>
__global__ void __launch_bounds__(256)
peak_from_rows_kernel(const T *__restrict__ amp, const T *__restrict__ ph, int bins, T freq_scale,
                      PeakRec *__restrict__ peaks, long long batch) {
  __shared__ T sv[256];
  __shared__ int si[256];
  const long long row = blockIdx.x;
  if (row >= batch) return;
  const T *a = amp + (size_t)row * (size_t)bins;
  T bv = T(0);
  int bi = 0;
  for (int i = 1 + (int)threadIdx.x; i < bins; i += 256) {
    const T v = a[i];
    if (v > bv) {
      bv = v;
      bi = i;
    }
  }
  sv[threadIdx.x] = bv;
  si[threadIdx.x] = bi;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      const T ov = sv[threadIdx.x + s], mv = sv[threadIdx.x];
      const int oi = si[threadIdx.x + s], mi = si[threadIdx.x];
      if (ov > mv || (ov == mv && ov > T(0) && oi < mi)) {
        sv[threadIdx.x] = ov;
        si[threadIdx.x] = oi;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const int idx = si[0];
    PeakRec r;
    r.index = idx;
    r.frequency = (float)(T(idx) * freq_scale);
    r.amplitude = (float)a[idx];
    r.phase = ph ? (float)ph[(size_t)row * (size_t)bins + idx] : 0.0f;
    peaks[row] = r;
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

// Short rows (bins <= 2048): one WAVE per row, four rows per workgroup -- no LDS, no barriers, and
// a launch of `batch/4` workgroups instead of `batch` (N = 64: the workgroup-per-row form took
// 14x the time of the spectrum kernel it follows).  Same rules as find_peak_kernel /
// peak_from_rows_kernel; `ph` and `peaks` may be null / `peak` may be null (one of the outputs is set).
template <typename T>
__global__ void __launch_bounds__(256)
peak_wave_kernel(const T *__restrict__ amp, const T *__restrict__ ph, int bins, T freq_scale, int *__restrict__ peak,
                 PeakRec *__restrict__ peaks, long long batch) {
  const int lane = (int)threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + ((int)threadIdx.x >> 6);
  if (row >= batch) return;
  const T *a = amp + (size_t)row * (size_t)bins;
  T bv = T(0);
  int bi = 0;
  for (int i = 1 + lane; i < bins; i += 64) {
    const T v = a[i];
    if (v > bv) {  // strict: the earliest index wins inside a lane's stride
      bv = v;
      bi = i;
    }
  }
  static_for<6>([&](auto sc) {
    constexpr int off = 32 >> sc;
    const T ov = __shfl_xor(bv, off, 64);
    const int oi = __shfl_xor(bi, off, 64);
    // larger value wins; equal values: smaller index (first-wins); index 0 = "none"
    if (ov > bv || (ov == bv && oi != 0 && (bi == 0 || oi < bi))) {
      bv = ov;
      bi = oi;
    }
  });
  if (lane == 0) {
    if (peak) peak[row] = bi;
    if (peaks) {
      PeakRec r;
      r.index = bi;
      r.frequency = (float)(T(bi) * freq_scale);
      r.amplitude = (float)a[bi];
      r.phase = ph ? (float)ph[(size_t)row * (size_t)bins + bi] : 0.0f;
      peaks[row] = r;
    }
  }
}

}  // namespace pdsp

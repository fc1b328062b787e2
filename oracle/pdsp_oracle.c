/*
 * pdsp_oracle.c -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * A scalar, single-threaded IEEE-f64 restatement of the pragma-dsp hot path
 * (radix-2 FFT + window / magnitude / phase helpers + spectrum()).  It is the
 * checker that tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline`
 * leg compare the HIP path against (and time beside it).  Nothing under
 * pragma-dsp_amd/ may include, link or call this file.
 *
 * Parity pin: this restatement is checked by tests/test_oracle_golden.py
 * against (1) the reference's own NumPy/SciPy goldens
 * test/reallife/references/{...}.json (all 35 signal cases + 16 window cases) and
 * (2) the fixture file regenerated with the reference's own
 * scripts/gen_fixtures.py (seed 1337), both committed in trimmed binary form
 * under tests/golden/ by tests/golden/make_golden.py.
 *
 * The reference implementation itself is TypeScript (no TS toolchain in the
 * image), so there is no oracle/_ref build: see DESIGN.md "Oracle".
 *
 * Each function cites the reference lines (relative to /root/reference) whose
 * behaviour it restates.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define ORACLE_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ sizes */

/* src/core/fft.ts:16 -- n > 0 and a single bit set. */
ORACLE_API int oracle_is_pow2(long long n) { return n > 0 && (n & (n - 1)) == 0; }

/* src/core/fft.ts:18-23 -- smallest power of two >= n; n <= 1 -> 1.
 * (The reference overflows int32 above 2^30; 64-bit here) */
ORACLE_API long long oracle_next_pow2(long long n) {
  long long p = 1;
  if (n <= 1) return 1;
  while (p < n) p <<= 1;
  return p;
}

/* ------------------------------------------------------------------- plan */

typedef struct {
  int n;
  int stages;
  uint32_t *rev;  /* src/core/fft.ts:25-38 */
  double **cosv;  /* src/core/fft.ts:45-61: one table per stage, m/2 entries */
  double **sinv;
} oracle_plan;

ORACLE_API void oracle_plan_destroy(oracle_plan *p) {
  if (!p) return;
  if (p->cosv) for (int s = 0; s < p->stages; ++s) free(p->cosv[s]);
  if (p->sinv) for (int s = 0; s < p->stages; ++s) free(p->sinv[s]);
  free(p->cosv);
  free(p->sinv);
  free(p->rev);
  free(p);
}

/* src/core/fft.ts:68-75 (ctor), :25-38 (bit reverse), :45-61 (twiddles).
 * Returns NULL when n is not a power of two (the reference throws
 * "FFT size must be power of two, got ${size}"). */
ORACLE_API oracle_plan *oracle_plan_create(int n) {
  if (!oracle_is_pow2(n)) return NULL;
  oracle_plan *p = (oracle_plan *)calloc(1, sizeof(*p));
  int bits = 0;
  while ((1 << bits) < n) ++bits;
  p->n = n;
  p->stages = bits;
  p->rev = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)n);
  for (int i = 0; i < n; ++i) {
    uint32_t x = (uint32_t)i, y = 0;
    for (int b = 0; b < bits; ++b) {
      y = (y << 1) | (x & 1u);
      x >>= 1;
    }
    p->rev[i] = y;
  }
  p->cosv = (double **)calloc((size_t)(bits ? bits : 1), sizeof(double *));
  p->sinv = (double **)calloc((size_t)(bits ? bits : 1), sizeof(double *));
  for (int s = 1; s <= bits; ++s) {
    int m = 1 << s, half = m >> 1;
    double *c = (double *)malloc(sizeof(double) * (size_t)half);
    double *d = (double *)malloc(sizeof(double) * (size_t)half);
    for (int k = 0; k < half; ++k) {
      /* angle = (-2*pi*k)/m, direct cos/sin, no recurrence (fft.ts:54-56) */
      double angle = (-2.0 * M_PI * (double)k) / (double)m;
      c[k] = cos(angle);
      d[k] = sin(angle);
    }
    p->cosv[s - 1] = c;
    p->sinv[s - 1] = d;
  }
  return p;
}

ORACLE_API int oracle_plan_size(const oracle_plan *p) { return p->n; }

/* src/core/fft.ts:89-151 -- transform(): bit-reversal scatter (:110-114),
 * log2 N in-place DIT stages (:116-140), inverse-only 1/N scale (:142-148).
 * im_in == NULL means "imaginary part is zero" (Radix2Fft.forward, :77-79).
 * Operation order inside the butterfly follows :125-138 exactly so that the
 * f64 result is bit-comparable with the TS path (JS has no FMA contraction;
 * this file is compiled with -ffp-contract=off). */
ORACLE_API void oracle_transform(const oracle_plan *p, const double *re_in,
                                 const double *im_in, double *re_out,
                                 double *im_out, int inverse) {
  const int n = p->n;
  for (int i = 0; i < n; ++i) {
    uint32_t j = p->rev[i];
    re_out[j] = re_in[i];
    im_out[j] = im_in ? im_in[i] : 0.0;
  }
  const double sgn = inverse ? -1.0 : 1.0;
  for (int s = 0; s < p->stages; ++s) {
    const int m = 1 << (s + 1), half = m >> 1;
    const double *c = p->cosv[s], *d = p->sinv[s];
    for (int k = 0; k < n; k += m) {
      for (int j = 0; j < half; ++j) {
        const int lo = k + j, hi = lo + half;
        const double tr = c[j] * re_out[hi] - sgn * d[j] * im_out[hi];
        const double ti = sgn * d[j] * re_out[hi] + c[j] * im_out[hi];
        const double ur = re_out[lo], ui = im_out[lo];
        re_out[lo] = ur + tr;
        im_out[lo] = ui + ti;
        re_out[hi] = ur - tr;
        im_out[hi] = ui - ti;
      }
    }
  }
  if (inverse) {
    const double scale = 1.0 / (double)n;
    for (int i = 0; i < n; ++i) {
      re_out[i] *= scale;
      im_out[i] *= scale;
    }
  }
}

/* Row-by-row batch: what the reference's caller loops do
 * (bench/reallife/signals.ts:264-270, bench/run.ts:18-26). */
ORACLE_API void oracle_transform_batch(const oracle_plan *p, long long batch,
                                       const double *re_in, const double *im_in,
                                       double *re_out, double *im_out,
                                       int inverse) {
  const size_t n = (size_t)p->n;
  for (long long b = 0; b < batch; ++b) {
    oracle_transform(p, re_in + (size_t)b * n, im_in ? im_in + (size_t)b * n : NULL,
                     re_out + (size_t)b * n, im_out + (size_t)b * n, inverse);
  }
}

/* ---------------------------------------------------------------- windows */

enum { ORACLE_WIN_RECT = 0, ORACLE_WIN_HANN = 1, ORACLE_WIN_HAMMING = 2, ORACLE_WIN_BLACKMAN = 3 };

/* src/xform/fourier.ts:14-52 -- symmetric windows, denominator size-1,
 * size==1 -> [1]; returns -1 for size <= 0 ("Window size must be positive"),
 * -2 for an unknown type ("Unsupported window type"). */
ORACLE_API int oracle_create_window(int type, int size, double *out) {
  if (size <= 0) return -1;
  if (type < 0 || type > 3) return -2;
  if (size == 1) {
    out[0] = 1.0;
    return 0;
  }
  for (int i = 0; i < size; ++i) {
    const double f = (2.0 * M_PI * (double)i) / (double)(size - 1);
    switch (type) {
      case ORACLE_WIN_RECT: out[i] = 1.0; break;
      case ORACLE_WIN_HANN: out[i] = 0.5 * (1.0 - cos(f)); break;
      case ORACLE_WIN_HAMMING: out[i] = 0.54 - 0.46 * cos(f); break;
      default: out[i] = 0.42 - 0.5 * cos(f) + 0.08 * cos(2.0 * f); break;
    }
  }
  return 0;
}

/* src/xform/fourier.ts:54-67 */
ORACLE_API void oracle_apply_window(const double *in, const double *win, int n, double *out) {
  for (int i = 0; i < n; ++i) out[i] = in[i] * win[i];
}

/* src/xform/fourier.ts:98-109 -- Math.hypot */
ORACLE_API void oracle_magnitude(const double *re, const double *im, long long n, double *out) {
  for (long long i = 0; i < n; ++i) out[i] = hypot(re[i], im[i]);
}

/* src/xform/fourier.ts:111-120 -- Math.atan2(im, re) */
ORACLE_API void oracle_phase(const double *re, const double *im, long long n, double *out) {
  for (long long i = 0; i < n; ++i) out[i] = atan2(im[i], re[i]);
}

/* src/xform/fourier.ts:122-134 -- out[i] = in[(i + floor(n/2)) % n] */
ORACLE_API void oracle_fft_shift(const double *in, int n, double *out) {
  const int mid = n / 2;
  for (int i = 0; i < n; ++i) out[i] = in[(i + mid) % n];
}

/* src/xform/fourier.ts:147-165 -- i * (fs / size); returns the bin count,
 * -1 for size <= 0, -2 for sample_rate <= 0. */
ORACLE_API int oracle_bin_frequencies(int size, double sample_rate, int two_sided, double *out) {
  if (size <= 0) return -1;
  if (sample_rate <= 0) return -2; /* `sampleRate <= 0` exactly (NaN passes, as in JS) */
  const int bins = two_sided ? size : size / 2 + 1;
  const double scale = sample_rate / (double)size;
  for (int i = 0; i < bins; ++i) out[i] = (double)i * scale;
  return bins;
}

/* --------------------------------------------------------------- spectrum */

typedef struct {
  int index;
  double frequency;
  double amplitude;
  double phase;
} oracle_peak;

/* src/public/spectrum.ts:74-105 -- arg-max over bins >= 1 with strict '>'
 * starting from 0; falls back to the global max (bin 0 wins ties) only when
 * no bin >= 1 is > 0. */
ORACLE_API int oracle_find_peak(const double *amp, int bins) {
  int max_i = 0, nondc_i = 0, has_nondc = 0;
  double max_v = bins > 0 ? amp[0] : 0.0, nondc_v = 0.0;
  for (int i = 1; i < bins; ++i) {
    const double v = amp[i];
    if (v > nondc_v) { nondc_v = v; nondc_i = i; }
    if (v > 0) has_nondc = 1;
    if (v > max_v) { max_v = v; max_i = i; }
  }
  return has_nondc ? nondc_i : max_i;
}

/* src/public/spectrum.ts:45-72 -- one-sided: bins 0 and N/2 scaled 1/N,
 * others 2/N; two-sided: 1/N.  `(2 * mag) / size` order kept. */
ORACLE_API void oracle_scale_amplitude(const double *mag, int size, int two_sided, double *out) {
  if (two_sided) {
    for (int k = 0; k < size; ++k) out[k] = mag[k] / (double)size;
    return;
  }
  const int bins = size / 2 + 1;
  const int nyq = (size % 2 == 0) ? size / 2 : -1;
  for (int k = 0; k < bins; ++k) {
    if (k == 0 || k == nyq) out[k] = mag[k] / (double)size;
    else out[k] = (2.0 * mag[k]) / (double)size;
  }
}

/* src/public/spectrum.ts:107-142.  fft_size <= 0 means "nextPowerOfTwo(len)"
 * (:113).  Frame = first min(N, len) samples zero padded (:36-43); the window
 * spans the padded length (:116-119).  Returns the bin count, or a negative
 * code: -1 fft size not a power of two, -2 bad window type, -3 bad rate. */
ORACLE_API int oracle_spectrum(const double *samples, int len, double sample_rate,
                               int fft_size, int window_type, int two_sided,
                               double *freq_out, double *amp_out, double *phase_out,
                               oracle_peak *peak) {
  const int n = fft_size > 0 ? fft_size : (int)oracle_next_pow2(len);
  oracle_plan *p = oracle_plan_create(n);
  if (!p) return -1;
  double *win = (double *)malloc(sizeof(double) * (size_t)n * 6);
  double *frame = win + n, *re = frame + n, *im = re + n, *mag = im + n, *ang = mag + n;
  if (oracle_create_window(window_type, n, win) != 0) { free(win); oracle_plan_destroy(p); return -2; }
  memset(frame, 0, sizeof(double) * (size_t)n);
  for (int i = 0; i < (len < n ? len : n); ++i) frame[i] = samples[i];
  oracle_apply_window(frame, win, n, frame);
  oracle_transform(p, frame, NULL, re, im, 0);
  oracle_magnitude(re, im, n, mag);
  oracle_phase(re, im, n, ang);
  const int bins = oracle_bin_frequencies(n, sample_rate, two_sided, freq_out);
  if (bins < 0) { free(win); oracle_plan_destroy(p); return -3; }
  oracle_scale_amplitude(mag, n, two_sided, amp_out);
  memcpy(phase_out, ang, sizeof(double) * (size_t)bins);
  const int pk = oracle_find_peak(amp_out, bins);
  peak->index = pk;
  peak->frequency = freq_out[pk];
  peak->amplitude = amp_out[pk];
  peak->phase = phase_out[pk];
  free(win);
  oracle_plan_destroy(p);
  return bins;
}

/* Batched fused spectrum amplitude (the row-by-row meaning of the device
 * kernel pdsp_spectrum_*): frames[batch][n] -> amp[batch][bins] with an
 * optional phase[batch][bins] and peak index per frame. */
ORACLE_API void oracle_spectrum_batch(const oracle_plan *p, long long batch,
                                      const double *frames, const double *win,
                                      int two_sided, double *amp_out,
                                      double *phase_out, int *peak_out) {
  const int n = p->n;
  const int bins = two_sided ? n : n / 2 + 1;
  double *buf = (double *)malloc(sizeof(double) * (size_t)n * 5);
  double *re = buf + n, *im = re + n, *mag = im + n, *ang = mag + n;
  for (long long b = 0; b < batch; ++b) {
    const double *x = frames + (size_t)b * (size_t)n;
    if (win) oracle_apply_window(x, win, n, buf);
    else memcpy(buf, x, sizeof(double) * (size_t)n);
    oracle_transform(p, buf, NULL, re, im, 0);
    oracle_magnitude(re, im, n, mag);
    oracle_scale_amplitude(mag, n, two_sided, amp_out + (size_t)b * (size_t)bins);
    if (phase_out) {
      oracle_phase(re, im, n, ang);
      memcpy(phase_out + (size_t)b * (size_t)bins, ang, sizeof(double) * (size_t)bins);
    }
    if (peak_out) peak_out[b] = oracle_find_peak(amp_out + (size_t)b * (size_t)bins, bins);
  }
  free(buf);
}

/* ---------------------------------------------------------- complex vectors */

/* src/math/complex.ts:26-197 -- op: 0 add, 1 sub, 2 mul, 3 div, 4 conj, 5 scale(s_re),
 * 6 mulScalar(s_re, s_im).  b is broadcast with period b_len (the device API's row
 * broadcast; b_len == n is the reference's element-wise case). */
ORACLE_API void oracle_complex_op(int op, long long n, const double *ar, const double *ai, const double *br,
                                  const double *bi, long long b_len, double s_re, double s_im, double *orr,
                                  double *oi) {
  for (long long i = 0; i < n; ++i) {
    const double a = ar[i], b = ai[i];
    const double c = op <= 3 ? br[i % b_len] : s_re, d = op <= 3 ? bi[i % b_len] : s_im;
    switch (op) {
      case 0: orr[i] = a + c; oi[i] = b + d; break;
      case 1: orr[i] = a - c; oi[i] = b - d; break;
      case 2: case 6: orr[i] = a * c - b * d; oi[i] = a * d + b * c; break;
      case 3: { const double den = c * c + d * d; orr[i] = (a * c + b * d) / den; oi[i] = (b * c - a * d) / den; break; }
      case 4: orr[i] = a; oi[i] = -b; break;
      default: orr[i] = a * c; oi[i] = b * c; break;
    }
  }
}

/* ------------------------------------------------------- cpu_baseline leg */

/* Times `reps` passes of the reference's caller loop (plan and `out` reused,
 * checksum accumulated so the work cannot be elided: bench/run.ts:13-26) over
 * `batch` distinct rows.  Returns seconds; *checksum receives the guard. */
ORACLE_API double oracle_time_forward(const oracle_plan *p, long long batch, int reps,
                                      const double *re_in, const double *im_in,
                                      double *checksum) {
  const size_t n = (size_t)p->n;
  double *re = (double *)malloc(sizeof(double) * n * 2), *im = re + n;
  double acc = 0.0;
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int r = 0; r < reps; ++r) {
    for (long long b = 0; b < batch; ++b) {
      oracle_transform(p, re_in + (size_t)b * n, im_in ? im_in + (size_t)b * n : NULL, re, im, 0);
      acc += re[1] + im[n - 1];
    }
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  free(re);
  if (checksum) *checksum = acc;
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

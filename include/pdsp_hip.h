/*
 * pdsp_hip.h -- C ABI of the MI355X (gfx950) batched FFT / spectrum engine that
 * re-backs pragma-dsp's hot path.  Plain pointers and sizes only; no torch, no
 * C++ types.  Built as pragma-dsp_amd/csrc/libpdsp_hip.so.
 *
 * Every entry point cites the reference interface (relative to the pragma-dsp
 * repository) whose work it replaces.  The reference has no FFI today (it is
 * pure TypeScript); the binding a maintainer would add (N-API addon) is in
 * pragma-dsp_amd/csrc/pdsp_napi.c and is described in INTEGRATION.md.
 *
 * Conventions (src/core/fft.ts:89-151, PLAN.md:104-128):
 *   forward  X[k] = sum_n x[n] e^{-j 2 pi k n / N}   (no normalisation)
 *   inverse  x[n] = (1/N) sum_k X[k] e^{+j 2 pi k n / N}
 *   complex data is PLANAR {real[], imag[]} like the reference's ComplexArray;
 *   a batch is `batch` rows of N contiguous values: re[b*N + i], im[b*N + i].
 *
 * Two families:
 *   *_f32 / *_f64 with `pdsp_stream`  -- DEVICE pointers, asynchronous on the
 *       given HIP stream (0 = the null stream).  This is the throughput path
 *       (bench, batched callers).
 *   *_host_f64                        -- HOST f64 pointers, synchronous.  This
 *       is what the JS drop-in (`Radix2Fft`, `FFT`, `spectrum`) binds: the
 *       reference API is Float64Array in / Float64Array out.
 *
 * Return value: 0 (PDSP_OK) or a pdsp_status; pdsp_last_error() then returns a
 * thread-local message.  For the argument errors the reference throws on, the
 * message is the reference's exact text, and validation happens before any
 * device work.
 */
#ifndef PDSP_HIP_H
#define PDSP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PDSP_API __attribute__((visibility("default")))

typedef struct pdsp_plan pdsp_plan; /* replaces a Radix2Fft instance, src/core/fft.ts:63-75 */
typedef void *pdsp_stream;          /* a hipStream_t, passed as an opaque pointer */

typedef enum pdsp_status {
  PDSP_OK = 0,
  PDSP_ERR_SIZE_NOT_POW2 = 1,   /* "FFT size must be power of two, got ${size}"  fft.ts:69-71, fourier.ts:74-76 */
  PDSP_ERR_INPUT_LENGTH = 2,    /* "FFT input length ${len} != size ${N}"         fft.ts:95-104 */
  PDSP_ERR_WINDOW_SIZE = 3,     /* "Window size must be positive, got ${size}"    fourier.ts:15-17 */
  PDSP_ERR_WINDOW_LENGTH = 4,   /* "Window length must match input length."       fourier.ts:59-61 */
  PDSP_ERR_WINDOW_TYPE = 5,     /* "Unsupported window type: ${t}"                fourier.ts:47-50 */
  PDSP_ERR_FFT_SIZE = 6,        /* "FFT size must be positive, got ${size}"       fourier.ts:152-154 */
  PDSP_ERR_SAMPLE_RATE = 7,     /* "Sample rate must be positive, got ${sr}"      fourier.ts:155-157 */
  PDSP_ERR_UNSUPPORTED_SIZE = 8,/* power of two, but beyond pdsp_max_size() for that precision */
  PDSP_ERR_BAD_ARG = 9,         /* null pointer / negative batch */
  PDSP_ERR_DEVICE = 10          /* no GPU, or a HIP runtime error (message has the HIP text) */
} pdsp_status;

typedef enum pdsp_window {      /* WindowType, src/xform/fourier.ts:11 */
  PDSP_WIN_RECT = 0,
  PDSP_WIN_HANN = 1,
  PDSP_WIN_HAMMING = 2,
  PDSP_WIN_BLACKMAN = 3
} pdsp_window;

typedef enum pdsp_sides {       /* FftSides, src/xform/fourier.ts:12 */
  PDSP_SIDES_ONE = 0,
  PDSP_SIDES_TWO = 1
} pdsp_sides;

typedef struct pdsp_peak {      /* SpectrumPeak, src/public/spectrum.ts:15-20 */
  int32_t index;
  double frequency;
  double amplitude;
  double phase;
} pdsp_peak;

typedef struct pdsp_peak32 {    /* SpectrumPeak of one frame as the device writes it (16 bytes) */
  int32_t index;
  float frequency;
  float amplitude;
  float phase;
} pdsp_peak32;

/* ---- library ---------------------------------------------------------- */

PDSP_API int pdsp_version(void);
PDSP_API const char *pdsp_last_error(void);
/* Number of visible HIP devices (0 when there is none); never fails. */
PDSP_API int pdsp_device_count(void);
/* Largest N for 4-byte / 8-byte scalars: 2^28 / 2^26.  Up to 16384 / 8192 (16384 for the f64
 * real-frame spectrum) one LDS-resident pass; up to 16x that a three-pass four-step transform
 * (columns fused into one kernel); above, the general five-pass four-step (both factors on the
 * row kernels, tiled transposes between them). */
PDSP_API int pdsp_max_size(int scalar_bytes);

/* Arithmetic of the *_host_f64 entry points (process-wide; returns the previous value).
 *   64 (default): f64 on the device for every size up to 2^26 -- the drop-in then meets the
 *       reference's own test tolerances (1e-10 against NumPy, signals.test.ts:22-23), not just the
 *       f32 contract; N = 2^27, 2^28 compute in f32.
 *   32: always f32 (the north-star's stated contract, max|err|/max|X| <= 1e-5).
 * PDSP_HOST_PRECISION=32 in the environment presets it. */
PDSP_API int pdsp_set_host_precision(int bits);

/* ---- host-side index math (no device) --------------------------------- */

/* isPowerOfTwo, src/core/fft.ts:16 (integers only). */
PDSP_API int pdsp_is_pow2(long long n);
/* nextPowerOfTwo, src/core/fft.ts:18-23 (no int32 overflow). */
PDSP_API long long pdsp_next_pow2(long long n);
/* createWindow, src/xform/fourier.ts:14-52.  Host, f64 (an f32 window misses
 * the reference's 1e-8 window tolerance). */
PDSP_API int pdsp_window_make(int type, long long size, double *out);
/* binFrequencies, src/xform/fourier.ts:147-165.  `out` holds size/2+1 (one) or
 * size (two) values; *bins_out receives the count. */
PDSP_API int pdsp_bin_frequencies(long long size, double sample_rate, int sides,
                                  double *out, long long *bins_out);
/* fftShift, src/xform/fourier.ts:122-134: out[i] = in[(i + floor(n/2)) % n]. */
PDSP_API int pdsp_fft_shift_f64(const double *in, long long n, double *out);
/* findPeak, src/public/spectrum.ts:74-105 (strict '>', DC skipped unless no
 * other bin is > 0).  Returns the index (>= 0). */
PDSP_API long long pdsp_find_peak_f64(const double *amplitude, long long bins);

/* ---- plan -------------------------------------------------------------- */

/* new Radix2Fft(size) / new FFT(size): src/core/fft.ts:68-75,
 * src/xform/fourier.ts:73-79.  Builds the twiddle tables on the host in f64 and
 * uploads them to `device` (-1 = the current HIP device).  The bit-reversal
 * table of fft.ts:25-38 has no counterpart: the Stockham kernel autosorts. */
PDSP_API int pdsp_plan_create(long long size, int device, pdsp_plan **plan_out);
PDSP_API int pdsp_plan_destroy(pdsp_plan *plan);
PDSP_API long long pdsp_plan_size(const pdsp_plan *plan);
/* pdsp_spectrum_host_f64 keeps its plans and device windows in a process-wide cache keyed
 * by (size, device) -- the FourierLive idea, src/effect/index.ts:30-48 (the reference's
 * spectrum() rebuilds plan and window on every call, spectrum.ts:114-116).  This frees it, and hands the
 * scratch planes the multi-pass paths (N > 16384) keep in the engine's own stream-ordered memory pool -- as large
 * as the largest such transform run so far -- back to the device. */
PDSP_API int pdsp_plan_cache_clear(void);
PDSP_API int pdsp_plan_device(const pdsp_plan *plan);

/* The plan's own device copy of createWindow(type, N) (src/xform/fourier.ts:14-52; built in f64 on
 * the host, rounded once, uploaded on first use, owned by the plan and valid for its lifetime) -- the
 * Map<"type:size", window> of FourierLive (src/effect/index.ts:39-48) per plan.  Passing this pointer
 * as `window` to pdsp_spectrum_f32 / pdsp_spectrum_peaks_f32 tells the engine WHICH window it is: for
 * N = 16384 f32 frames the cosine-sum window a0 - a1 cos(2 pi n/(N-1)) + a2 cos(4 pi n/(N-1)) is then
 * evaluated in registers (createWindow + applyWindow fused into the frame load; values agree with
 * the table to ~2e-7 absolute) instead of being re-read, 64 KB per frame, from L2.  Any other
 * pointer is used as a table of N values. */
PDSP_API int pdsp_plan_window_f32(pdsp_plan *plan, int type, const float **window_out);
PDSP_API int pdsp_plan_window_f64(pdsp_plan *plan, int type, const double **window_out);

/* ---- plane layout ------------------------------------------------------- */

/* Device memory for the four planes of a batched transform, laid out for this card's memory system: inside one large
 * allocation the MI355X address space behaves as regions of 32 GiB, and a transform that streams two input planes
 * and two output planes runs fastest -- 79-83 % of the HBM roofline at N = 4096, repeatable, against 71-84 % by
 * lottery for four separate allocations -- with both INPUT planes in one region and each OUTPUT plane in a region
 * of its own.  One allocation of 80 GiB + a plane: re_in at 0, im_in right behind it, re_out
 * 40 GiB in, im_out 80 GiB in (any phase of the allocation against the region grid then puts the outputs one and two
 * regions beyond the inputs).  When that much memory is not free, or a plane is below 256 MiB (nothing to gain) or
 * above 8 GiB: four plain allocations.
 *   scalar_bytes  4 (float planes) or 8 (double planes); each plane holds batch * N scalars
 *   real_input    non-zero: no imaginary input plane (*im_in = NULL)
 *   *arena        opaque handle for pdsp_planes_free (which frees all four planes); *arena_bytes (may be NULL)
 *                 receives the size of the one allocation, 0 for the plain-allocation fallback
 * Nothing in the reference corresponds to this: it is a property of the device, not of the algorithm. */
typedef struct pdsp_arena pdsp_arena;
PDSP_API int pdsp_planes_alloc(const pdsp_plan *plan, long long batch, int scalar_bytes, int real_input,
                               void **re_in, void **im_in, void **re_out, void **im_out,
                               pdsp_arena **arena, unsigned long long *arena_bytes);
PDSP_API int pdsp_planes_free(pdsp_arena *arena);

/* ---- device-pointer batched transforms (f32) --------------------------- */

/* Radix2Fft.forward(input) row by row, src/core/fft.ts:77-79: real input,
 * imaginary part taken as zero.  re_in[batch*N] -> re_out, im_out[batch*N]. */
PDSP_API int pdsp_fft_forward_real_f32(const pdsp_plan *plan, long long batch,
                                       const float *re_in, float *re_out, float *im_out,
                                       pdsp_stream stream);
/* Radix2Fft.forwardComplex, src/core/fft.ts:81-83.
 * Where the four planes lie matters on this card (8-14 % at N = 4096): inside one large allocation the address space
 * behaves as regions of 32 GiB, and the launch is fastest -- and repeatable -- with both input planes in one region
 * and each output plane in a region of its own, e.g. offsets 0 / plane / 40 GiB / 80 GiB of one allocation
 * (pragma_dsp_amd.batch.BatchedFft.alloc_planes does exactly that). */
PDSP_API int pdsp_fft_forward_complex_f32(const pdsp_plan *plan, long long batch,
                                          const float *re_in, const float *im_in,
                                          float *re_out, float *im_out, pdsp_stream stream);
/* The same forwardComplex / inverse on INTERLEAVED rows: in/out hold batch*N (re, im) pairs
 * (2*N scalars per row) -- the layout of I/Q streams and complex64 tensors.  An extension: the
 * reference's ComplexArray is planar (src/core/fft.ts:1-4).  Single-pass sizes only (N <= 16384
 * in f32, 8192 in f64; PDSP_ERR_UNSUPPORTED_SIZE beyond); out may alias in row for row. */
PDSP_API int pdsp_fft_forward_interleaved_f32(const pdsp_plan *plan, long long batch, const float *in,
                                              float *out, pdsp_stream stream);
PDSP_API int pdsp_fft_inverse_interleaved_f32(const pdsp_plan *plan, long long batch, const float *in,
                                              float *out, pdsp_stream stream);
/* Radix2Fft.inverse (conjugate twiddles + 1/N), src/core/fft.ts:85-87, :142-148. */
PDSP_API int pdsp_fft_inverse_f32(const pdsp_plan *plan, long long batch,
                                  const float *re_in, const float *im_in,
                                  float *re_out, float *im_out, pdsp_stream stream);

/* ---- device-pointer elementwise helpers (f32) -------------------------- */

/* applyWindow row by row, src/xform/fourier.ts:54-67: out[b][i] = in[b][i]*win[i]. */
PDSP_API int pdsp_apply_window_f32(long long batch, long long n, const float *in,
                                   const float *window, float *out, pdsp_stream stream);
/* magnitude, src/xform/fourier.ts:98-109 (sqrt(re^2+im^2) in f32; not the
 * overflow-safe hypot: fine for |X| < 1e19). */
PDSP_API int pdsp_magnitude_f32(long long count, const float *re, const float *im,
                                float *out, pdsp_stream stream);
/* phase, src/xform/fourier.ts:111-120: atan2(im, re). */
PDSP_API int pdsp_phase_f32(long long count, const float *re, const float *im,
                            float *out, pdsp_stream stream);

/* Element-wise complex vector arithmetic on planar device rows, src/math/complex.ts:26-197
 * (scaleInto, addInto, subInto, mulInto, mulScalarInto, divInto, conjInto), so FFT-domain
 * pipelines (forward -> mul -> inverse, test/fluent/chain.test.ts:287-317) stay in HBM.
 * out = a OP b for the binary ops, where b holds b_len values and is broadcast over the
 * count/b_len rows of a when b_len < count (b_len must divide count); out = a OP (s_re, s_im)
 * for SCALE (real s_re) and MUL_SCALAR; out = conj(a) for CONJ.  divScalar is MUL_SCALAR by the
 * host-computed reciprocal, as complex.ts:176-186 does.  out may alias a, and b may alias out when
 * b_len == count (every element is read before it is written); a broadcast b (b_len < count) must
 * not overlap out. */
typedef enum pdsp_complex_op {
  PDSP_CX_ADD = 0, PDSP_CX_SUB = 1, PDSP_CX_MUL = 2, PDSP_CX_DIV = 3,
  PDSP_CX_CONJ = 4, PDSP_CX_SCALE = 5, PDSP_CX_MUL_SCALAR = 6
} pdsp_complex_op;
PDSP_API int pdsp_complex_op_f32(int op, long long count, const float *a_re, const float *a_im,
                                 const float *b_re, const float *b_im, long long b_len,
                                 double s_re, double s_im, float *out_re, float *out_im,
                                 pdsp_stream stream);

/* ---- fused spectrum (f32) ---------------------------------------------- */

/* The batched body of spectrum(), src/public/spectrum.ts:116-131, one frame per
 * row, fused in one kernel: buildFrame (zero-pad / truncate to N, :36-43) ->
 * applyWindow -> forward FFT -> magnitude -> one-/two-sided amplitude scaling
 * (:45-72) [-> phase, :122,128-131] [-> findPeak index, :74-105].
 *   frames      [batch][frame_stride] real samples; the first
 *               min(frame_len, N) of each row are used, the rest is zero.
 *               frame_stride >= 1: a stride below frame_len reads OVERLAPPING frames of one
 *               signal (a short-time transform with hop = frame_stride; frame b starts at
 *               sample b*frame_stride and the buffer must hold (batch-1)*frame_stride +
 *               min(frame_len, N) samples).
 *   window      N device floats, or NULL for "rect".
 *   amp_out     [batch][bins], bins = N/2+1 (one-sided) or N (two-sided).
 *   phase_out   same shape, or NULL.
 *   peak_out    [batch] int32 peak bin per frame, or NULL. */
PDSP_API int pdsp_spectrum_f32(const pdsp_plan *plan, long long batch,
                               const float *frames, long long frame_len, long long frame_stride,
                               const float *window, int sides,
                               float *amp_out, float *phase_out, int32_t *peak_out,
                               pdsp_stream stream);

/* The whole tail of spectrum() on the device (src/public/spectrum.ts:116-134), one
 * SpectrumPeak per frame: findPeak (:74-105: bins >= 1, strict '>', the first of equal
 * values wins, bin 0 when no other bin is > 0) is fused into the spectrum kernel, with
 * peak.frequency = index * sample_rate / N and peak.phase = atan2 of that bin.
 *   peaks_out   [batch] records (16 B each).
 *   amp_out / phase_out   optional [batch][bins] rows as in pdsp_spectrum_f32; NULL for
 *               peaks-only output: HBM traffic is then 4 B/sample in + 16 B/frame out, and a
 *               multi-GPU gather moves KiBs instead of GiBs.
 * In two-sided mode bins k and N-k carry identical values here, so the peak is the lower
 * index k (the reference picks whichever f64 rounding favours). */
PDSP_API int pdsp_spectrum_peaks_f32(const pdsp_plan *plan, long long batch,
                                     const float *frames, long long frame_len, long long frame_stride,
                                     const float *window, int sides, double sample_rate,
                                     float *amp_out, float *phase_out, pdsp_peak32 *peaks_out,
                                     pdsp_stream stream);

/* ---- the same device-pointer family in f64 ------------------------------ */
/* Identical contracts with double rows, N <= 2^26 (PDSP_ERR_UNSUPPORTED_SIZE beyond).
 * ~1e-15 relative to max vs the f64 reference. */
PDSP_API int pdsp_fft_forward_real_f64(const pdsp_plan *plan, long long batch,
                                       const double *re_in, double *re_out, double *im_out,
                                       pdsp_stream stream);
PDSP_API int pdsp_fft_forward_complex_f64(const pdsp_plan *plan, long long batch,
                                          const double *re_in, const double *im_in,
                                          double *re_out, double *im_out, pdsp_stream stream);
PDSP_API int pdsp_fft_inverse_f64(const pdsp_plan *plan, long long batch,
                                  const double *re_in, const double *im_in,
                                  double *re_out, double *im_out, pdsp_stream stream);
PDSP_API int pdsp_fft_forward_interleaved_f64(const pdsp_plan *plan, long long batch, const double *in,
                                              double *out, pdsp_stream stream);
PDSP_API int pdsp_fft_inverse_interleaved_f64(const pdsp_plan *plan, long long batch, const double *in,
                                              double *out, pdsp_stream stream);
PDSP_API int pdsp_apply_window_f64(long long batch, long long n, const double *in,
                                   const double *window, double *out, pdsp_stream stream);
/* f64 magnitude is hypot(re, im) like the reference's Math.hypot (fourier.ts:106): no overflow /
 * underflow of the squares at the ends of the double range. */
PDSP_API int pdsp_magnitude_f64(long long count, const double *re, const double *im,
                                double *out, pdsp_stream stream);
PDSP_API int pdsp_phase_f64(long long count, const double *re, const double *im,
                            double *out, pdsp_stream stream);
PDSP_API int pdsp_spectrum_f64(const pdsp_plan *plan, long long batch,
                               const double *frames, long long frame_len, long long frame_stride,
                               const double *window, int sides,
                               double *amp_out, double *phase_out, int32_t *peak_out,
                               pdsp_stream stream);

/* ---- host f64 drop-in entry points (synchronous) ----------------------- */

/* Radix2Fft.transform, src/core/fft.ts:89-151, for `batch` rows.  im_in may be
 * NULL (= forward(real)).  in_len is the caller's row length and must equal N
 * (PDSP_ERR_INPUT_LENGTH otherwise, message as fft.ts:95-104).  Computes on the
 * device in the precision pdsp_set_host_precision() selects (f64 at the boundary either way). */
PDSP_API int pdsp_fft_transform_host_f64(pdsp_plan *plan, long long batch, long long in_len,
                                         const double *re_in, const double *im_in,
                                         double *re_out, double *im_out, int inverse);
/* The same with one pointer per input row (re_rows[b], and im_rows[b] unless im_rows is NULL, point at N values
 * each, anywhere in host memory): a JS array of Float64Arrays / ComplexArrays taken where it lies.  Outputs are
 * contiguous planes of batch*N values as above. */
PDSP_API int pdsp_fft_transform_rows_host_f64(pdsp_plan *plan, long long batch, long long in_len,
                                              const double *const *re_rows, const double *const *im_rows,
                                              double *re_out, double *im_out, int inverse);
/* applyWindow / magnitude / phase on host arrays (fourier.ts:54-67, :98-120).
 * Small inputs: done by launching the same device kernels. */
PDSP_API int pdsp_apply_window_host_f64(const double *in, long long in_len, const double *window,
                                        long long window_len, double *out);
PDSP_API int pdsp_magnitude_host_f64(const double *re, const double *im, long long n, double *out);
PDSP_API int pdsp_phase_host_f64(const double *re, const double *im, long long n, double *out);
/* spectrum(samples, options), src/public/spectrum.ts:107-142.  fft_size < 0
 * means "absent" = nextPowerOfTwo(len) (0 is rejected like `new FFT(0)`).  freq/amp/phase hold bins values (N/2+1 or N);
 * *bins_out receives the count.  Peak search runs on the host over the
 * f64-promoted amplitudes so the strict-'>' / first-wins rules are exact. */
PDSP_API int pdsp_spectrum_host_f64(const double *samples, long long len, double sample_rate,
                                    long long fft_size, int window, int sides,
                                    double *freq_out, double *amp_out, double *phase_out,
                                    pdsp_peak *peak_out, long long *bins_out);
/* The same on `batch` frames of `len` samples each (contiguous) in ONE call -- the map of the
 * reference's spectrumStream (src/effect/index.ts:190-194) batched onto the device.  freq_out holds
 * the one frequency axis (bins values); amp_out / phase_out batch*bins; peak_out batch records.
 * Row b equals pdsp_spectrum_host_f64 on frame b bit for bit.  Calls with 4 MiB or more of staging are cut into
 * chunks that several host threads (PDSP_HOST_THREADS; default half the cores, 2 ... 6; 1 = none) stage, copy and
 * launch concurrently on streams of their own -- the results do not depend on it. */
PDSP_API int pdsp_spectrum_batch_host_f64(const double *frames, long long batch, long long len,
                                          double sample_rate, long long fft_size, int window, int sides,
                                          double *freq_out, double *amp_out, double *phase_out,
                                          pdsp_peak *peak_out, long long *bins_out);

/* The same with ONE POINTER PER FRAME: rows[b] points at the `len` samples of frame b, anywhere in host memory --
 * the Iterable<ArrayLike<number>> of spectrumStream (src/effect/index.ts:190-194) as a JS caller holds it, an array
 * of Float64Arrays, taken as it stands instead of being flattened into one buffer first (a copy of 8 bytes per
 * sample on the caller's one thread).  Same outputs, same bit-for-bit rule. */
PDSP_API int pdsp_spectrum_rows_host_f64(const double *const *rows, long long batch, long long len,
                                         double sample_rate, long long fft_size, int window, int sides,
                                         double *freq_out, double *amp_out, double *phase_out,
                                         pdsp_peak *peak_out, long long *bins_out);

/* The same for frames held as 32-bit floats (Float32Array: what Web Audio, decoders and capture APIs hand a JS
 * caller; `ArrayLike<number>` in the reference's signature): rows[b] points at `len` floats.  The samples are widened
 * exactly (float -> double is exact) where the call computes in f64, and taken as they are in f32 mode, so the
 * results equal those of the f64 entry points on the widened frames bit for bit; outputs stay f64. */
PDSP_API int pdsp_spectrum_rows_host_f32in(const float *const *rows, long long batch, long long len,
                                           double sample_rate, long long fft_size, int window, int sides,
                                           double *freq_out, double *amp_out, double *phase_out,
                                           pdsp_peak *peak_out, long long *bins_out);

#ifdef __cplusplus
}
#endif
#endif /* PDSP_HIP_H */

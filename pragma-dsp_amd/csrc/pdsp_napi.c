/*
 * pdsp_napi.c -- Node N-API addon: the binding a pragma-dsp maintainer adds to
 * re-back src/core/fft.ts and src/xform/fourier.ts with libpdsp_hip.so.
 *
 * Thin by design: typed-array pointers in, the C ABI of include/pdsp_hip.h
 * called, status mapped to `throw new Error(pdsp_last_error())` (the reference's
 * error convention).  ArrayLike flattening (`?? 0`), `out` identity and
 * option defaults live in the JS host (pragma-dsp_amd/js/), not here.
 *
 * Native plans are wrapped in a napi external with a finalizer: the reference's
 * Radix2Fft has no destroy() method, so the GC frees the device tables.
 */
#include <node_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pdsp_hip.h"

#define NAPI_OK_OR_THROW(env, call)                                       \
  do {                                                                    \
    if ((call) != napi_ok) {                                              \
      napi_throw_error((env), NULL, "pdsp_napi: N-API call failed: " #call); \
      return NULL;                                                        \
    }                                                                     \
  } while (0)

static napi_value throw_pdsp(napi_env env) {
  napi_throw_error(env, NULL, pdsp_last_error());
  return NULL;
}

static int get_args(napi_env env, napi_callback_info info, size_t want, napi_value *argv) {
  size_t argc = want;
  if (napi_get_cb_info(env, info, &argc, argv, NULL, NULL) != napi_ok || argc < want) {
    napi_throw_type_error(env, NULL, "pdsp_napi: wrong number of arguments");
    return 0;
  }
  return 1;
}

/* Float64Array -> (double*, length).  null/undefined -> (NULL, 0). */
static int f64_array(napi_env env, napi_value v, double **data, size_t *len) {
  napi_valuetype t;
  *data = NULL;
  *len = 0;
  if (napi_typeof(env, v, &t) != napi_ok) return 0;
  if (t == napi_null || t == napi_undefined) return 1;
  bool is_ta = false;
  if (napi_is_typedarray(env, v, &is_ta) != napi_ok || !is_ta) {
    napi_throw_type_error(env, NULL, "pdsp_napi: expected a Float64Array");
    return 0;
  }
  napi_typedarray_type tt;
  void *p = NULL;
  size_t n = 0;
  if (napi_get_typedarray_info(env, v, &tt, &n, &p, NULL, NULL) != napi_ok || tt != napi_float64_array) {
    napi_throw_type_error(env, NULL, "pdsp_napi: expected a Float64Array");
    return 0;
  }
  *data = (double *)p;
  *len = n;
  return 1;
}

/* Float64Array or Float32Array -> (data, length, is_f32); anything else throws a TypeError. */
static int f64_or_f32_array(napi_env env, napi_value v, void **data, size_t *len, int *is_f32) {
  bool is_ta = false;
  napi_typedarray_type tt;
  void *p = NULL;
  size_t n = 0;
  *data = NULL;
  *len = 0;
  if (napi_is_typedarray(env, v, &is_ta) != napi_ok || !is_ta ||
      napi_get_typedarray_info(env, v, &tt, &n, &p, NULL, NULL) != napi_ok ||
      (tt != napi_float64_array && tt != napi_float32_array)) {
    napi_throw_type_error(env, NULL, "pdsp_napi: expected a Float64Array or a Float32Array");
    return 0;
  }
  *data = p;
  *len = n;
  *is_f32 = tt == napi_float32_array;
  return 1;
}

static int get_i64(napi_env env, napi_value v, int64_t *out) {
  if (napi_get_value_int64(env, v, out) != napi_ok) {
    napi_throw_type_error(env, NULL, "pdsp_napi: expected a number");
    return 0;
  }
  return 1;
}

static int get_f64(napi_env env, napi_value v, double *out) {
  if (napi_get_value_double(env, v, out) != napi_ok) {
    napi_throw_type_error(env, NULL, "pdsp_napi: expected a number");
    return 0;
  }
  return 1;
}

static void plan_finalize(napi_env env, void *data, void *hint) {
  (void)env;
  (void)hint;
  pdsp_plan_destroy((pdsp_plan *)data);
}

/* planCreate(size) -> external   [new Radix2Fft(size), src/core/fft.ts:68-75] */
static napi_value PlanCreate(napi_env env, napi_callback_info info) {
  napi_value argv[1];
  int64_t size;
  if (!get_args(env, info, 1, argv) || !get_i64(env, argv[0], &size)) return NULL;
  pdsp_plan *plan = NULL;
  if (pdsp_plan_create(size, -1, &plan) != PDSP_OK) return throw_pdsp(env);
  napi_value ext;
  NAPI_OK_OR_THROW(env, napi_create_external(env, plan, plan_finalize, NULL, &ext));
  return ext;
}

/* transform(plan, re, imOrNull, outRe, outIm, inverse)   [Radix2Fft.transform, fft.ts:89-151] */
static napi_value Transform(napi_env env, napi_callback_info info) {
  napi_value argv[6];
  if (!get_args(env, info, 6, argv)) return NULL;
  void *plan = NULL;
  NAPI_OK_OR_THROW(env, napi_get_value_external(env, argv[0], &plan));
  double *re, *im, *ore, *oim;
  size_t nre, nim, nore, noim;
  if (!f64_array(env, argv[1], &re, &nre) || !f64_array(env, argv[2], &im, &nim) ||
      !f64_array(env, argv[3], &ore, &nore) || !f64_array(env, argv[4], &oim, &noim))
    return NULL;
  bool inverse = false;
  NAPI_OK_OR_THROW(env, napi_get_value_bool(env, argv[5], &inverse));
  const long long n = pdsp_plan_size((pdsp_plan *)plan);
  /* the length checks of fft.ts:95-104, real plane first, then imaginary */
  long long bad = -1;
  if ((long long)nre != n) bad = (long long)nre;
  else if (im && (long long)nim != n) bad = (long long)nim;
  if (bad >= 0) {
    char msg[128];
    snprintf(msg, sizeof(msg), "FFT input length %lld != size %lld", bad, n);
    napi_throw_error(env, NULL, msg);
    return NULL;
  }
  if ((long long)nore != n || (long long)noim != n) {
    napi_throw_error(env, NULL, "pdsp_napi: output planes must have the plan's size");
    return NULL;
  }
  if (pdsp_fft_transform_host_f64((pdsp_plan *)plan, 1, n, re, im, ore, oim, inverse ? 1 : 0) != PDSP_OK)
    return throw_pdsp(env);
  return NULL;
}

/* transformBatch(plan, batch, re, imOrNull, outRe, outIm, inverse): `batch` rows of the plan's size, planes of
 * batch*size values   [Radix2Fft.transform row by row, fft.ts:89-151; the loop of bench/reallife/signals.ts:264-270
 * as one call] */
static napi_value TransformBatch(napi_env env, napi_callback_info info) {
  napi_value argv[7];
  if (!get_args(env, info, 7, argv)) return NULL;
  void *plan = NULL;
  NAPI_OK_OR_THROW(env, napi_get_value_external(env, argv[0], &plan));
  int64_t batch;
  double *re, *im, *ore, *oim;
  size_t nre, nim, nore, noim;
  if (!get_i64(env, argv[1], &batch) || !f64_array(env, argv[2], &re, &nre) || !f64_array(env, argv[3], &im, &nim) ||
      !f64_array(env, argv[4], &ore, &nore) || !f64_array(env, argv[5], &oim, &noim))
    return NULL;
  bool inverse = false;
  NAPI_OK_OR_THROW(env, napi_get_value_bool(env, argv[6], &inverse));
  const long long n = pdsp_plan_size((pdsp_plan *)plan);
  if (batch < 0 || n <= 0 || batch > (int64_t)(((size_t)-1) / 16 / (size_t)n)) {
    napi_throw_error(env, NULL, "pdsp_napi: transformBatch bad batch");
    return NULL;
  }
  const size_t want = (size_t)batch * (size_t)n;
  if (nre != want || (im && nim != want) || nore != want || noim != want) {
    napi_throw_error(env, NULL, "pdsp_napi: transformBatch planes must hold batch * size values");
    return NULL;
  }
  if (inverse && !im && batch > 0) {
    napi_throw_error(env, NULL, "pdsp_napi: transformBatch inverse needs an imaginary plane");
    return NULL;
  }
  if (pdsp_fft_transform_host_f64((pdsp_plan *)plan, batch, n, re, im, ore, oim, inverse ? 1 : 0) != PDSP_OK)
    return throw_pdsp(env);
  return NULL;
}

/* transformRows(plan, inputs, complex, outRe, outIm, inverse): inputs is a JS array of Float64Arrays (complex =
 * false: real rows) or of { real, imag } pairs of Float64Arrays, every plane of the plan's size -- read where they
 * lie (pdsp_fft_transform_rows_host_f64); outRe / outIm hold inputs.length * size values. */
static napi_value TransformRows(napi_env env, napi_callback_info info) {
  napi_value argv[6];
  if (!get_args(env, info, 6, argv)) return NULL;
  void *plan = NULL;
  NAPI_OK_OR_THROW(env, napi_get_value_external(env, argv[0], &plan));
  bool is_array = false, complex = false, inverse = false;
  uint32_t batch = 0;
  if (napi_is_array(env, argv[1], &is_array) != napi_ok || !is_array ||
      napi_get_array_length(env, argv[1], &batch) != napi_ok) {
    napi_throw_type_error(env, NULL, "pdsp_napi: transformRows expects an array of rows");
    return NULL;
  }
  NAPI_OK_OR_THROW(env, napi_get_value_bool(env, argv[2], &complex));
  double *ore, *oim;
  size_t nore, noim;
  if (!f64_array(env, argv[3], &ore, &nore) || !f64_array(env, argv[4], &oim, &noim)) return NULL;
  NAPI_OK_OR_THROW(env, napi_get_value_bool(env, argv[5], &inverse));
  const long long n = pdsp_plan_size((pdsp_plan *)plan);
  if (n <= 0 || (size_t)batch > ((size_t)-1) / 16 / (size_t)n || nore != (size_t)batch * (size_t)n ||
      noim != (size_t)batch * (size_t)n) {
    napi_throw_error(env, NULL, "pdsp_napi: transformRows output planes must hold rows * size values");
    return NULL;
  }
  if (inverse && !complex && batch > 0) {
    napi_throw_error(env, NULL, "pdsp_napi: transformRows inverse needs complex rows");
    return NULL;
  }
  const double **re = (const double **)malloc(sizeof(double *) * (size_t)(batch ? batch : 1));
  const double **im = complex ? (const double **)malloc(sizeof(double *) * (size_t)(batch ? batch : 1)) : NULL;
  if (!re || (complex && !im)) {
    free(re);
    free(im);
    napi_throw_error(env, NULL, "pdsp_napi: out of memory");
    return NULL;
  }
  for (uint32_t b = 0; b < batch; ++b) {
    napi_handle_scope scope;
    if (napi_open_handle_scope(env, &scope) != napi_ok) {
      napi_throw_error(env, NULL, "pdsp_napi: N-API call failed: napi_open_handle_scope");
      free(re);
      free(im);
      return NULL;
    }
    napi_value el, v;
    double *p = NULL, *q = NULL;
    size_t np = 0, nq = 0;
    int ok = napi_get_element(env, argv[1], b, &el) == napi_ok;
    if (ok && !complex) {
      ok = f64_array(env, el, &p, &np);
    } else if (ok) {
      napi_valuetype t;
      ok = napi_typeof(env, el, &t) == napi_ok && t == napi_object;
      if (!ok) napi_throw_type_error(env, NULL, "pdsp_napi: transformRows expects { real, imag } rows");
      ok = ok && napi_get_named_property(env, el, "real", &v) == napi_ok && f64_array(env, v, &p, &np) &&
           napi_get_named_property(env, el, "imag", &v) == napi_ok && f64_array(env, v, &q, &nq);
    }
    bool pending = false;
    if (!ok && napi_is_exception_pending(env, &pending) == napi_ok && !pending)
      napi_throw_error(env, NULL, "pdsp_napi: N-API call failed while reading a row");
    if (ok && ((long long)np != n || (complex && (long long)nq != n) || !p || (complex && !q))) {
      napi_throw_error(env, NULL, "pdsp_napi: transformRows rows must be Float64Arrays of the plan's size");
      ok = 0;
    }
    napi_close_handle_scope(env, scope);
    if (!ok) {
      free(re);
      free(im);
      return NULL;
    }
    re[b] = p;
    if (complex) im[b] = q;
  }
  const int rc = pdsp_fft_transform_rows_host_f64((pdsp_plan *)plan, (long long)batch, n, re, im, ore, oim, inverse ? 1 : 0);
  free(re);
  free(im);
  if (rc != PDSP_OK) return throw_pdsp(env);
  return NULL;
}

/* windowMake(type, size, out)   [createWindow, fourier.ts:14-52] */
static napi_value WindowMake(napi_env env, napi_callback_info info) {
  napi_value argv[3];
  int64_t type, size;
  double *out;
  size_t n;
  if (!get_args(env, info, 3, argv) || !get_i64(env, argv[0], &type) || !get_i64(env, argv[1], &size) ||
      !f64_array(env, argv[2], &out, &n))
    return NULL;
  if ((int64_t)n < size) {
    napi_throw_error(env, NULL, "pdsp_napi: window output too small");
    return NULL;
  }
  if (pdsp_window_make((int)type, size, out) != PDSP_OK) return throw_pdsp(env);
  return NULL;
}

/* applyWindow(in, win, out)   [fourier.ts:54-67] */
static napi_value ApplyWindow(napi_env env, napi_callback_info info) {
  napi_value argv[3];
  double *in, *win, *out;
  size_t nin, nwin, nout;
  if (!get_args(env, info, 3, argv) || !f64_array(env, argv[0], &in, &nin) || !f64_array(env, argv[1], &win, &nwin) ||
      !f64_array(env, argv[2], &out, &nout))
    return NULL;
  if (nout < nin && nin == nwin) {
    napi_throw_error(env, NULL, "pdsp_napi: applyWindow output too small");
    return NULL;
  }
  if (pdsp_apply_window_host_f64(in, (long long)nin, win, (long long)nwin, out) != PDSP_OK) return throw_pdsp(env);
  return NULL;
}

static napi_value polar(napi_env env, napi_callback_info info, int want_phase) {
  napi_value argv[3];
  double *re, *im, *out;
  size_t nre, nim, nout;
  if (!get_args(env, info, 3, argv) || !f64_array(env, argv[0], &re, &nre) || !f64_array(env, argv[1], &im, &nim) ||
      !f64_array(env, argv[2], &out, &nout))
    return NULL;
  if (nim < nre || nout < nre) {
    napi_throw_error(env, NULL, "pdsp_napi: magnitude/phase planes too small");
    return NULL;
  }
  const int rc = want_phase ? pdsp_phase_host_f64(re, im, (long long)nre, out)
                            : pdsp_magnitude_host_f64(re, im, (long long)nre, out);
  if (rc != PDSP_OK) return throw_pdsp(env);
  return NULL;
}
/* magnitude(re, im, out) / phase(re, im, out)   [fourier.ts:98-120] */
static napi_value Magnitude(napi_env env, napi_callback_info info) { return polar(env, info, 0); }
static napi_value Phase(napi_env env, napi_callback_info info) { return polar(env, info, 1); }

/* spectrum(samples, sampleRate, fftSizeOrMinus1, window, sides, freq, amp, phase) -> peak
 * [spectrum(), src/public/spectrum.ts:107-142] */
static napi_value Spectrum(napi_env env, napi_callback_info info) {
  napi_value argv[8];
  if (!get_args(env, info, 8, argv)) return NULL;
  double *x, *freq, *amp, *ph, rate;
  size_t nx, nf, na, np;
  int64_t fft_size, window, sides;
  if (!f64_array(env, argv[0], &x, &nx) || !get_f64(env, argv[1], &rate) || !get_i64(env, argv[2], &fft_size) ||
      !get_i64(env, argv[3], &window) || !get_i64(env, argv[4], &sides) || !f64_array(env, argv[5], &freq, &nf) ||
      !f64_array(env, argv[6], &amp, &na) || !f64_array(env, argv[7], &ph, &np))
    return NULL;
  const long long n = fft_size >= 0 ? fft_size : pdsp_next_pow2((long long)nx);
  const long long bins = sides == PDSP_SIDES_ONE ? n / 2 + 1 : n;
  if (pdsp_is_pow2(n) && ((long long)nf < bins || (long long)na < bins || (long long)np < bins)) {
    napi_throw_error(env, NULL, "pdsp_napi: spectrum outputs too small");
    return NULL;
  }
  pdsp_peak pk;
  long long got = 0;
  static double dummy = 0.0;
  if (pdsp_spectrum_host_f64(nx ? x : &dummy, (long long)nx, rate, fft_size, (int)window, (int)sides, freq, amp, ph,
                             &pk, &got) != PDSP_OK)
    return throw_pdsp(env);
  napi_value obj, v;
  NAPI_OK_OR_THROW(env, napi_create_object(env, &obj));
  NAPI_OK_OR_THROW(env, napi_create_int32(env, pk.index, &v));
  NAPI_OK_OR_THROW(env, napi_set_named_property(env, obj, "index", v));
  NAPI_OK_OR_THROW(env, napi_create_double(env, pk.frequency, &v));
  NAPI_OK_OR_THROW(env, napi_set_named_property(env, obj, "frequency", v));
  NAPI_OK_OR_THROW(env, napi_create_double(env, pk.amplitude, &v));
  NAPI_OK_OR_THROW(env, napi_set_named_property(env, obj, "amplitude", v));
  NAPI_OK_OR_THROW(env, napi_create_double(env, pk.phase, &v));
  NAPI_OK_OR_THROW(env, napi_set_named_property(env, obj, "phase", v));
  return obj;
}

/* spectrumBatch(frames, batch, frameLen, sampleRate, fftSizeOrMinus1, window, sides, freq, amp, phase, peaks)
 * frames: batch*frameLen samples; amp/phase: batch*bins; peaks: 4 doubles per frame
 * (index, frequency, amplitude, phase)   [the map of spectrumStream, src/effect/index.ts:190-194] */
static napi_value SpectrumBatch(napi_env env, napi_callback_info info) {
  napi_value argv[11];
  if (!get_args(env, info, 11, argv)) return NULL;
  double *x, *freq, *amp, *ph, *pk, rate;
  size_t nx, nf, na, np, npk;
  int64_t batch, len, fft_size, window, sides;
  if (!f64_array(env, argv[0], &x, &nx) || !get_i64(env, argv[1], &batch) || !get_i64(env, argv[2], &len) ||
      !get_f64(env, argv[3], &rate) || !get_i64(env, argv[4], &fft_size) || !get_i64(env, argv[5], &window) ||
      !get_i64(env, argv[6], &sides) || !f64_array(env, argv[7], &freq, &nf) || !f64_array(env, argv[8], &amp, &na) ||
      !f64_array(env, argv[9], &ph, &np) || !f64_array(env, argv[10], &pk, &npk))
    return NULL;
  if (batch < 0 || len < 0 || (long long)nx < batch * len) {
    napi_throw_error(env, NULL, "pdsp_napi: spectrumBatch frames buffer too small");
    return NULL;
  }
  const long long n = fft_size >= 0 ? fft_size : pdsp_next_pow2((long long)len);
  const long long bins = sides == PDSP_SIDES_ONE ? n / 2 + 1 : n;
  if (pdsp_is_pow2(n) && ((long long)nf < bins || (long long)na < batch * bins || (long long)np < batch * bins ||
                          (long long)npk < 4 * batch)) {
    napi_throw_error(env, NULL, "pdsp_napi: spectrumBatch outputs too small");
    return NULL;
  }
  pdsp_peak *recs = (pdsp_peak *)malloc(sizeof(pdsp_peak) * (size_t)(batch > 0 ? batch : 1));
  if (!recs) {
    napi_throw_error(env, NULL, "pdsp_napi: out of memory");
    return NULL;
  }
  static double dummy = 0.0;
  const int rc = pdsp_spectrum_batch_host_f64(nx ? x : &dummy, batch, len, rate, fft_size, (int)window, (int)sides, freq,
                                              amp, ph, recs, NULL);
  if (rc == PDSP_OK)
    for (int64_t b = 0; b < batch; ++b) {
      pk[4 * b + 0] = (double)recs[b].index;
      pk[4 * b + 1] = recs[b].frequency;
      pk[4 * b + 2] = recs[b].amplitude;
      pk[4 * b + 3] = recs[b].phase;
    }
  free(recs);
  if (rc != PDSP_OK) return throw_pdsp(env);
  return NULL;
}

/* spectrumRows(frames, start, batch, frameLen, sampleRate, fftSizeOrMinus1, window, sides, freq, amp, phase, peaks)
 * frames: a JS array whose elements start .. start+batch-1 are all Float64Arrays or all Float32Arrays of frameLen
 * samples each -- taken where they lie (pdsp_spectrum_rows_host_f64 / _f32in), not flattened; outputs as
 * spectrumBatch. */
static napi_value SpectrumRows(napi_env env, napi_callback_info info) {
  napi_value argv[12];
  if (!get_args(env, info, 12, argv)) return NULL;
  double *freq, *amp, *ph, *pk, rate;
  size_t nf, na, np, npk;
  int64_t start, batch, len, fft_size, window, sides;
  bool is_array = false;
  uint32_t nframes = 0;
  if (napi_is_array(env, argv[0], &is_array) != napi_ok || !is_array ||
      napi_get_array_length(env, argv[0], &nframes) != napi_ok) {
    napi_throw_type_error(env, NULL, "pdsp_napi: spectrumRows expects an array of Float64Arrays");
    return NULL;
  }
  if (!get_i64(env, argv[1], &start) || !get_i64(env, argv[2], &batch) || !get_i64(env, argv[3], &len) ||
      !get_f64(env, argv[4], &rate) || !get_i64(env, argv[5], &fft_size) || !get_i64(env, argv[6], &window) ||
      !get_i64(env, argv[7], &sides) || !f64_array(env, argv[8], &freq, &nf) || !f64_array(env, argv[9], &amp, &na) ||
      !f64_array(env, argv[10], &ph, &np) || !f64_array(env, argv[11], &pk, &npk))
    return NULL;
  if (start < 0 || batch < 0 || len < 0 || start + batch > (int64_t)nframes) {
    napi_throw_error(env, NULL, "pdsp_napi: spectrumRows frame range out of bounds");
    return NULL;
  }
  const long long n = fft_size >= 0 ? fft_size : pdsp_next_pow2((long long)len);
  const long long bins = sides == PDSP_SIDES_ONE ? n / 2 + 1 : n;
  if (pdsp_is_pow2(n) && ((long long)nf < bins || (long long)na < batch * bins || (long long)np < batch * bins ||
                          (long long)npk < 4 * batch)) {
    napi_throw_error(env, NULL, "pdsp_napi: spectrumRows outputs too small");
    return NULL;
  }
  const void **rows = (const void **)malloc(sizeof(void *) * (size_t)(batch > 0 ? batch : 1));
  pdsp_peak *recs = (pdsp_peak *)malloc(sizeof(pdsp_peak) * (size_t)(batch > 0 ? batch : 1));
  if (!rows || !recs) {
    free(rows);
    free(recs);
    napi_throw_error(env, NULL, "pdsp_napi: out of memory");
    return NULL;
  }
  static double dummy = 0.0;
  int run_f32 = -1; /* the run's element type: every frame a Float64Array, or every frame a Float32Array */
  for (int64_t b = 0; b < batch; ++b) {
    napi_handle_scope scope;
    napi_value el;
    void *p = NULL;
    size_t plen = 0;
    int f32 = 0;
    int ok = napi_open_handle_scope(env, &scope) == napi_ok;
    if (ok) {
      ok = napi_get_element(env, argv[0], (uint32_t)(start + b), &el) == napi_ok &&
           f64_or_f32_array(env, el, &p, &plen, &f32);
      if (ok && (int64_t)plen != len) {
        napi_throw_error(env, NULL, "pdsp_napi: spectrumRows frames must all have frameLen samples");
        ok = 0;
      }
      if (ok && run_f32 >= 0 && run_f32 != f32) {
        napi_throw_type_error(env, NULL, "pdsp_napi: spectrumRows frames must all be of one typed-array kind");
        ok = 0;
      }
      napi_close_handle_scope(env, scope);
    } else {
      napi_throw_error(env, NULL, "pdsp_napi: N-API call failed: napi_open_handle_scope");
    }
    if (!ok) {
      free(rows);
      free(recs);
      return NULL;
    }
    run_f32 = f32;
    rows[b] = p ? p : (void *)&dummy; /* a zero-length typed array has no data pointer */
  }
  const int rc = run_f32 == 1
                     ? pdsp_spectrum_rows_host_f32in((const float *const *)rows, batch, len, rate, fft_size, (int)window,
                                                     (int)sides, freq, amp, ph, recs, NULL)
                     : pdsp_spectrum_rows_host_f64((const double *const *)rows, batch, len, rate, fft_size, (int)window,
                                                   (int)sides, freq, amp, ph, recs, NULL);
  if (rc == PDSP_OK)
    for (int64_t b = 0; b < batch; ++b) {
      pk[4 * b + 0] = (double)recs[b].index;
      pk[4 * b + 1] = recs[b].frequency;
      pk[4 * b + 2] = recs[b].amplitude;
      pk[4 * b + 3] = recs[b].phase;
    }
  free(rows);
  free(recs);
  if (rc != PDSP_OK) return throw_pdsp(env);
  return NULL;
}

/* binFrequencies(size, sampleRate, sides, out)   [fourier.ts:147-165] */
static napi_value BinFrequencies(napi_env env, napi_callback_info info) {
  napi_value argv[4];
  int64_t size, sides;
  double rate, *out;
  size_t n;
  if (!get_args(env, info, 4, argv) || !get_i64(env, argv[0], &size) || !get_f64(env, argv[1], &rate) ||
      !get_i64(env, argv[2], &sides) || !f64_array(env, argv[3], &out, &n))
    return NULL;
  /* size / rate errors keep the reference's texts (the C call validates before it writes); a valid request must
   * fit the caller's array: size/2 + 1 values one-sided, size two-sided */
  if (size > 0 && rate > 0 && n < (size_t)(sides == PDSP_SIDES_ONE ? size / 2 + 1 : size)) {
    napi_throw_error(env, NULL, "pdsp_napi: binFrequencies output too small");
    return NULL;
  }
  if (pdsp_bin_frequencies(size, rate, (int)sides, out, NULL) != PDSP_OK) return throw_pdsp(env);
  return NULL;
}

/* fftShift(in, out)   [fourier.ts:122-134] */
static napi_value FftShift(napi_env env, napi_callback_info info) {
  napi_value argv[2];
  double *in, *out;
  size_t nin, nout;
  if (!get_args(env, info, 2, argv) || !f64_array(env, argv[0], &in, &nin) || !f64_array(env, argv[1], &out, &nout))
    return NULL;
  if (nout < nin) {
    napi_throw_error(env, NULL, "pdsp_napi: fftShift output too small");
    return NULL;
  }
  if (pdsp_fft_shift_f64(in, (long long)nin, out) != PDSP_OK) return throw_pdsp(env);
  return NULL;
}

static napi_value NextPow2(napi_env env, napi_callback_info info) {
  napi_value argv[1], out;
  int64_t n;
  if (!get_args(env, info, 1, argv) || !get_i64(env, argv[0], &n)) return NULL;
  NAPI_OK_OR_THROW(env, napi_create_int64(env, pdsp_next_pow2(n), &out));
  return out;
}

static napi_value DeviceCount(napi_env env, napi_callback_info info) {
  (void)info;
  napi_value out;
  NAPI_OK_OR_THROW(env, napi_create_int32(env, pdsp_device_count(), &out));
  return out;
}

static napi_value Init(napi_env env, napi_value exports) {
  const struct {
    const char *name;
    napi_callback fn;
  } fns[] = {
      {"planCreate", PlanCreate}, {"transform", Transform},   {"windowMake", WindowMake},
      {"transformBatch", TransformBatch}, {"transformRows", TransformRows},
      {"applyWindow", ApplyWindow}, {"magnitude", Magnitude}, {"phase", Phase},
      {"spectrum", Spectrum},     {"binFrequencies", BinFrequencies}, {"fftShift", FftShift},
      {"spectrumBatch", SpectrumBatch}, {"spectrumRows", SpectrumRows},
      {"nextPow2", NextPow2},     {"deviceCount", DeviceCount},
  };
  for (size_t i = 0; i < sizeof(fns) / sizeof(fns[0]); ++i) {
    napi_value f;
    if (napi_create_function(env, fns[i].name, NAPI_AUTO_LENGTH, fns[i].fn, NULL, &f) != napi_ok) return NULL;
    if (napi_set_named_property(env, exports, fns[i].name, f) != napi_ok) return NULL;
  }
  napi_value ver;
  if (napi_create_int32(env, pdsp_version(), &ver) == napi_ok) napi_set_named_property(env, exports, "version", ver);
  return exports;
}

NAPI_MODULE(NODE_GYP_MODULE_NAME, Init)

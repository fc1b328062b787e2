/* A plain C11 consumer of include/pdsp_hip.h: no HIP headers, no C++, host-pointer entry points only -- what a cgo / JNI /
 * N-API / ctypes binding sees.  Built by tests/test_capi_cpu.py with `gcc -std=c11 -pedantic -Wall -Werror` (the header
 * must be valid C, not just C++), run on the GPU box by tests/test_gpu_c_consumer.py.
 *
 * It runs BASELINE configs[0] -- spectrum([0,1,0,-1,0,1,0,-1], {sampleRate: 48000}) of the reference's README (README.md:11,
 * test/fluent/chain.test.ts:27) -- and one Radix2Fft.forward / inverse round trip, and prints the results as one line of JSON.
 * Exit status: 0 ok, 1 a call failed (message on stderr), 2 a result is off. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "pdsp_hip.h"

static int check(int rc, const char *what) {
  if (rc != PDSP_OK) fprintf(stderr, "%s failed: status %d: %s\n", what, rc, pdsp_last_error());
  return rc;
}

int main(void) {
  const double x[8] = {0, 1, 0, -1, 0, 1, 0, -1};
  double freq[5], amp[5], phase[5];
  pdsp_peak peak;
  long long bins = 0;
  /* fft_size < 0 = absent = nextPowerOfTwo(len); window rect, one-sided: the reference's defaults (spectrum.ts:111-115) */
  if (check(pdsp_spectrum_host_f64(x, 8, 48000.0, -1, PDSP_WIN_RECT, PDSP_SIDES_ONE, freq, amp, phase, &peak, &bins), "spectrum")) return 1;

  pdsp_plan *plan = NULL;
  if (check(pdsp_plan_create(8, -1, &plan), "plan_create")) return 1;
  double re[8], im[8], bre[8], bim[8];
  if (check(pdsp_fft_transform_host_f64(plan, 1, 8, x, NULL, re, im, 0), "forward")) return 1;
  if (check(pdsp_fft_transform_host_f64(plan, 1, 8, re, im, bre, bim, 1), "inverse")) return 1;
  double rt = 0;
  for (int i = 0; i < 8; ++i) rt = fmax(rt, fmax(fabs(bre[i] - x[i]), fabs(bim[i])));
  /* an argument error carries the reference's text (fft.ts:69-71) */
  pdsp_plan *bad = NULL;
  const int rc12 = pdsp_plan_create(12, -1, &bad);
  printf("{\"version\": %d, \"bins\": %lld, \"amplitude\": [%.17g, %.17g, %.17g, %.17g, %.17g], \"peak\": {\"index\": %d, "
         "\"frequency\": %.17g, \"amplitude\": %.17g, \"phase\": %.17g}, \"X2\": [%.17g, %.17g], \"round_trip_err\": %.3g, "
         "\"size12_status\": %d, \"size12_message\": \"%s\", \"next_pow2_1000\": %lld}\n",
         pdsp_version(), bins, amp[0], amp[1], amp[2], amp[3], amp[4], (int)peak.index, peak.frequency, peak.amplitude,
         peak.phase, re[2], im[2], rt, rc12, pdsp_last_error(), pdsp_next_pow2(1000));
  pdsp_plan_destroy(plan);
  const double pi = 3.14159265358979323846;
  const int ok = bins == 5 && peak.index == 2 && fabs(peak.frequency - 12000.0) < 1e-9 && fabs(peak.amplitude - 1.0) < 1e-12 &&
                 fabs(peak.phase + pi / 2) < 1e-12 && fabs(amp[2] - 1.0) < 1e-12 && fabs(amp[0]) < 1e-12 && fabs(re[2]) < 1e-12 &&
                 fabs(im[2] + 4.0) < 1e-12 && rt < 1e-12 && rc12 == PDSP_ERR_SIZE_NOT_POW2;
  return ok ? 0 : 2;
}

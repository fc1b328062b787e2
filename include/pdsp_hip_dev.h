/*
 * pdsp_hip_dev.h -- development switches of libpdsp_hip.so.  NOT part of the drop-in boundary: the reference has
 * nothing like them (src/core/fft.ts and src/xform/fourier.ts select no algorithms), a binding of pdsp_hip.h never
 * includes this file, and every default is the production path.  They exist so that the parity tests can run the
 * same input through two kernels of the engine and hold both to the oracle (tests/test_gpu_*.py), and so that
 * the scripts under tools/ can time one against the other inside one process.  Process-wide; each returns the previous value.
 */
#ifndef PDSP_HIP_DEV_H
#define PDSP_HIP_DEV_H

#include "pdsp_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Kernel selection switch for A/B tests (process-wide; returns the previous value): 1 (default)
 * runs whole pair-aligned one-sided N = 16384 f32 spectra on spectrum_dif16k_kernel (two 4096-point
 * sub-transforms per workgroup, decimation in frequency on top) and 16-byte aligned N = 16384 f32
 * complex/real rows on fft_split4_kernel (four 4096-point sub-transforms per workgroup), and N = 8192
 * rows (f64; f32 real input) on fft_split2_kernel; 0 on spectrum_packed_kernel<13> and the
 * single-pass fft_stockham_kernel.  Same results within rounding.  (Bit 1 set also routes f32 complex
 * N = 8192 rows to fft_split2_kernel: a development A/B switch.) */
PDSP_API int pdsp_set_split16k(int enabled);
/* Same kind of switch for 32 <= N <= 256 transforms on 16-byte aligned planes: 1 (default) =
 * fft_staged_kernel (the workgroup's contiguous 4096-point chunk staged through LDS with coalesced
 * 16-byte accesses), 0 = the direct kernel. */
PDSP_API int pdsp_set_staged_small(int enabled);
/* 1 (default): a window argument that IS one of the plan's own tables (pdsp_plan_window_f32) is
 * evaluated inside the N = 16384 f32 spectrum kernel (createWindow fused, see below); 0: it is read
 * as a table like any caller-supplied window.  A/B switch for the parity tests; returns the
 * previous value. */
PDSP_API int pdsp_set_fused_window(int enabled);
/* 1 (default): f32 transforms of 2^15 <= N <= 2^27 on 16-byte aligned planes run as tile passes over
 * balanced factors of 64 ... 512 points (tile_pass_kernel): TWO passes over HBM up to 2^18, THREE above;
 * 0: round 1's four-step forms (N1 <= 16 columns, 16384-point rows, transposing copy: three passes up to
 * 2^18, five above); 3: tile passes in their first form -- the scratch planes between the first two of three
 * passes in natural order instead of tile-major (bit-identical results), 512-point factors on 16-wide tiles
 * instead of the 32-wide ones of tile_rows512_kernel / tile_cols512_kernel (same results within rounding).
 * With the value 1 only, 2^15 and 2^16 out of place run in ONE pass over HBM on fft_paired_kernel (2 / 4 sibling
 * workgroups per transform sharing an XCD's L2); 5 = the tile passes' current form without it.
 * A/B switch, returns the previous value. */
PDSP_API int pdsp_set_twopass(int enabled);

/* 1 (default): f64 Radix2Fft.forward rows (real input, src/core/fft.ts:77-79) of N = 8192, and of N = 16384 from 8
 * rows up, run as ONE N/2-point packed-real transform per row + the split to X[k], X[k + N/2] (fft_real_kernel);
 * 0: as the complex kernel of their size on (x, 0) (fft_stockham_kernel with LoadReal; N = 16384: the four-step
 * path).  Same results within rounding.  Other sizes and f32 real rows always take the complex kernels
 * (DESIGN 4.1c). */
PDSP_API int pdsp_set_real_packed(int enabled);

#ifdef __cplusplus
}
#endif
#endif /* PDSP_HIP_DEV_H */

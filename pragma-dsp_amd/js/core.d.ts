// Types of the drop-in for `pragma-dsp/core` (reference src/core/fft.ts:1-87): planar complex buffers and the
// radix-2 plan.  Declarations only -- the arithmetic is the HIP engine behind the N-API addon.
export type ComplexArray = { real: Float64Array; imag: Float64Array };

/** fft.ts:6-14 -- `fill` goes to BOTH planes. */
export function createComplexArray(size: number, fill?: number): ComplexArray;
/** fft.ts:16, plus an integer check (2.5 is not a power of two here). */
export function isPowerOfTwo(n: number): boolean;
/** fft.ts:18-23 without the int32 overflow above 2^30; n <= 1 gives 1. */
export function nextPowerOfTwo(n: number): number;

/** fft.ts:63-87.  Throws `FFT size must be power of two, got ${size}` / `FFT input length ${len} != size ${N}`.
 *  A supplied `out` is written in place and returned (the same object). */
export class Radix2Fft {
  constructor(size: number);
  readonly size: number;
  forward(input: ArrayLike<number>, out?: ComplexArray): ComplexArray;
  forwardComplex(input: ComplexArray, out?: ComplexArray): ComplexArray;
  inverse(input: ComplexArray, out?: ComplexArray): ComplexArray;
  /** Extensions: element i equals forward / forwardComplex / inverse of inputs[i]; one device batch per call
   *  (the loop of bench/reallife/signals.ts:264-270), results are views into one buffer per plane. */
  forwardBatch(inputs: ReadonlyArray<ArrayLike<number>>): ComplexArray[];
  forwardComplexBatch(inputs: ReadonlyArray<ComplexArray>): ComplexArray[];
  inverseBatch(inputs: ReadonlyArray<ComplexArray>): ComplexArray[];
}

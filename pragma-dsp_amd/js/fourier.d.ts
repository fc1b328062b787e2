// Types of the drop-in for `pragma-dsp/xform/fourier` (reference src/xform/fourier.ts:11-165).
import { ComplexArray } from './core';

export type WindowType = 'rect' | 'hann' | 'hamming' | 'blackman';

/** fourier.ts:14-52: symmetric windows, denominator size - 1; built in f64 on the host. */
export function createWindow(type: WindowType, size: number): Float64Array;
/** fourier.ts:54-67.  Throws `Window length must match input length.` */
export function applyWindow(input: ArrayLike<number>, window: ArrayLike<number>, out?: Float64Array): Float64Array;

/** fourier.ts:69-96: delegates 1:1 to Radix2Fft. */
export class FFT {
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
  createComplexArray(fill?: number): ComplexArray;
}

/** fourier.ts:98-109 (hypot per bin). */
export function magnitude(input: ComplexArray, out?: Float64Array): Float64Array;
/** fourier.ts:111-120 (atan2 per bin). */
export function phase(input: ComplexArray, out?: Float64Array): Float64Array;
/** fourier.ts:122-134: rotation by floor(n / 2). */
export function fftShift(input: ArrayLike<number>, out?: Float64Array): Float64Array;
/** fourier.ts:136-145. */
export function fftShiftComplex(input: ComplexArray, out?: ComplexArray): ComplexArray;
/** fourier.ts:147-165: i * sampleRate / size for size/2 + 1 (one-sided, default) or size bins. */
export function binFrequencies(size: number, sampleRate: number, sides?: 'one' | 'two'): Float64Array;

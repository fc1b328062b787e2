// Types of the drop-in for the root export `spectrum` (reference src/public/spectrum.ts:15-34, 107-142).
import { WindowType } from './fourier';

export type SpectrumOptions = {
  /** default 1 */
  sampleRate?: number;
  /** default nextPowerOfTwo(samples.length); longer input is truncated, shorter zero-padded */
  fftSize?: number;
  /** default 'rect' */
  window?: WindowType;
  /** default 'one': bins 0 .. N/2, DC and Nyquist not doubled */
  sides?: 'one' | 'two';
};

export type SpectrumPeak = { index: number; frequency: number; amplitude: number; phase: number };

export type SpectrumResult = {
  frequencies: Float64Array;
  amplitude: Float64Array;
  phase: Float64Array;
  peak: SpectrumPeak;
};

export function spectrum(samples: ArrayLike<number>, options?: SpectrumOptions): SpectrumResult;

/** Extension: the map of the reference's spectrumStream (src/effect/index.ts:190-194) as one device batch per
 *  run of equal-length frames; result i equals spectrum(frames[i], options) exactly. */
export function spectrumBatch(frames: ReadonlyArray<ArrayLike<number>>, options?: SpectrumOptions): SpectrumResult[];

/** Extension: spectrumStream's frame-at-a-time contract (src/effect/index.ts:190-194; one result per frame, in
 *  order) as a synchronous generator that runs the frames as device batches of `batchFrames` (default 256). */
export function spectrumStream(frames: Iterable<ArrayLike<number>>, options?: SpectrumOptions,
  batchFrames?: number): Generator<SpectrumResult, void, undefined>;

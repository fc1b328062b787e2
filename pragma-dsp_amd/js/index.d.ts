// Export map of the host module: the three reference surfaces that sit on the hot path.
import * as coreNs from './core';
import * as fourierNs from './fourier';

export { spectrum, spectrumBatch, spectrumStream, SpectrumOptions, SpectrumPeak, SpectrumResult } from './spectrum';
export { ComplexArray } from './core';
export { WindowType } from './fourier';

export const core: {
  createComplexArray: typeof coreNs.createComplexArray;
  isPowerOfTwo: typeof coreNs.isPowerOfTwo;
  nextPowerOfTwo: typeof coreNs.nextPowerOfTwo;
  Radix2Fft: typeof coreNs.Radix2Fft;
};
export const fourier: {
  createWindow: typeof fourierNs.createWindow;
  applyWindow: typeof fourierNs.applyWindow;
  FFT: typeof fourierNs.FFT;
  magnitude: typeof fourierNs.magnitude;
  phase: typeof fourierNs.phase;
  fftShift: typeof fourierNs.fftShift;
  fftShiftComplex: typeof fourierNs.fftShiftComplex;
  binFrequencies: typeof fourierNs.binFrequencies;
};

'use strict';
// Drop-in for the root export `spectrum` (reference src/public/spectrum.ts:107-142):
// one fused kernel launch (frame build -> window -> FFT -> magnitude/phase -> amplitude
// scaling); the frequency axis and findPeak are host index math inside the addon.
const native = require('./native');
const core = require('./core');
const fourier = require('./fourier');

function spectrum(samples, options) {
  const opts = options || {};
  const sampleRate = opts.sampleRate === undefined || opts.sampleRate === null ? 1 : opts.sampleRate;
  const sides = opts.sides === undefined || opts.sides === null ? 'one' : opts.sides;
  const targetSize = opts.fftSize === undefined || opts.fftSize === null
    ? core.nextPowerOfTwo(samples.length) : opts.fftSize;
  const windowType = opts.window === undefined || opts.window === null ? 'rect' : opts.window;
  // error order of spectrum.ts:114-132: FFT ctor, createWindow, binFrequencies
  if (!core.isPowerOfTwo(targetSize)) {
    throw new Error('FFT size must be power of two, got ' + targetSize);
  }
  if (targetSize !== 1 && !Object.prototype.hasOwnProperty.call(fourier._WINDOW_IDS, windowType)) {
    throw new Error('Unsupported window type: ' + windowType);
  }
  if (sampleRate <= 0) {
    throw new Error('Sample rate must be positive, got ' + sampleRate);
  }
  const one = sides === 'one';
  const bins = one ? Math.floor(targetSize / 2) + 1 : targetSize;
  const frequencies = new Float64Array(bins);
  const amplitude = new Float64Array(bins);
  const phase = new Float64Array(bins);
  const id = fourier._WINDOW_IDS[windowType];
  const peak = native.spectrum(core._toF64(samples), sampleRate, targetSize, id === undefined ? 0 : id,
    one ? 0 : 1, frequencies, amplitude, phase);
  return { frequencies: frequencies, amplitude: amplitude, phase: phase, peak: peak };
}

module.exports = { spectrum };

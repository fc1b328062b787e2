'use strict';
// Drop-in for `pragma-dsp/xform/fourier` (reference src/xform/fourier.ts).
const native = require('./native');
const core = require('./core');

const WINDOW_IDS = { rect: 0, hann: 1, hamming: 2, blackman: 3 };
const toF64 = core._toF64;

// fourier.ts:14-52 -- host f64 (an f32 window would miss the 1e-8 golden tolerance)
function createWindow(type, size) {
  if (size <= 0) {
    throw new Error('Window size must be positive, got ' + size);
  }
  if (size === 1) return new Float64Array([1]);
  if (!Object.prototype.hasOwnProperty.call(WINDOW_IDS, type)) {
    throw new Error('Unsupported window type: ' + type);
  }
  const out = new Float64Array(size);
  native.windowMake(WINDOW_IDS[type], size, out);
  return out;
}

function intoOut(out, n, fill) {
  const direct = out instanceof Float64Array && out.length >= n;
  const tmp = direct ? out : new Float64Array(n);
  fill(tmp);
  if (!direct && out) for (let i = 0; i < n; i += 1) out[i] = tmp[i];
  return out || tmp;
}

// fourier.ts:54-67
function applyWindow(input, window, out) {
  if (input.length !== window.length) {
    throw new Error('Window length must match input length.');
  }
  const x = toF64(input);
  const w = toF64(window);
  return intoOut(out, x.length, function (dst) { native.applyWindow(x, w, dst); });
}

// fourier.ts:69-96
class FFT {
  constructor(size) {
    if (!core.isPowerOfTwo(size)) {
      throw new Error('FFT size must be power of two, got ' + size);
    }
    Object.defineProperty(this, 'size', { value: size, enumerable: true, writable: false });
    this._kernel = new core.Radix2Fft(size);
  }
  forward(input, out) { return this._kernel.forward(input, out); }
  forwardComplex(input, out) { return this._kernel.forwardComplex(input, out); }
  inverse(input, out) { return this._kernel.inverse(input, out); }
  // extensions (see Radix2Fft): many rows in one device batch
  forwardBatch(inputs) { return this._kernel.forwardBatch(inputs); }
  forwardComplexBatch(inputs) { return this._kernel.forwardComplexBatch(inputs); }
  inverseBatch(inputs) { return this._kernel.inverseBatch(inputs); }
  createComplexArray(fill) { return core.createComplexArray(this.size, fill === undefined ? 0 : fill); }
}

function planes(input) {
  const re = toF64(input.real);
  let im = toF64(input.imag);
  if (im.length < re.length) { // `input.imag[i] ?? 0`
    const padded = new Float64Array(re.length);
    padded.set(im);
    im = padded;
  }
  return [re, im];
}

// fourier.ts:98-109
function magnitude(input, out) {
  const p = planes(input);
  return intoOut(out, p[0].length, function (dst) { native.magnitude(p[0], p[1], dst); });
}

// fourier.ts:111-120
function phase(input, out) {
  const p = planes(input);
  return intoOut(out, p[0].length, function (dst) { native.phase(p[0], p[1], dst); });
}

// fourier.ts:122-134 (host index math)
function fftShift(input, out) {
  const x = toF64(input);
  return intoOut(out, x.length, function (dst) { native.fftShift(x, dst); });
}

// fourier.ts:136-145
function fftShiftComplex(input, out) {
  const n = input.real.length;
  const result = out === undefined || out === null ? core.createComplexArray(n) : out;
  result.real.set(fftShift(input.real));
  result.imag.set(fftShift(input.imag));
  return result;
}

// fourier.ts:147-165
function binFrequencies(size, sampleRate, sides) {
  if (size <= 0) {
    throw new Error('FFT size must be positive, got ' + size);
  }
  if (sampleRate <= 0) {
    throw new Error('Sample rate must be positive, got ' + sampleRate);
  }
  const one = sides === undefined || sides === 'one';
  const out = new Float64Array(one ? Math.floor(size / 2) + 1 : size);
  native.binFrequencies(size, sampleRate, one ? 0 : 1, out);
  return out;
}

module.exports = {
  createWindow, applyWindow, FFT, magnitude, phase, fftShift, fftShiftComplex, binFrequencies,
  _WINDOW_IDS: WINDOW_IDS,
};

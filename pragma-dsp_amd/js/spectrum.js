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

// spectrumBatch(frames, options): the map of the reference's spectrumStream (src/effect/index.ts:190-194
// -- one spectrum() result per frame, in order) as ONE device batch per run of equal-length frames
// instead of one launch + one sync per frame.  Every result equals spectrum(frame, options) exactly;
// amplitude and phase of a run are views into one buffer each, frequencies is a fresh copy per result.
function spectrumBatch(frames, options) {
  const opts = options || {};
  const sampleRate = opts.sampleRate === undefined || opts.sampleRate === null ? 1 : opts.sampleRate;
  const sides = opts.sides === undefined || opts.sides === null ? 'one' : opts.sides;
  const windowType = opts.window === undefined || opts.window === null ? 'rect' : opts.window;
  const one = sides === 'one';
  const id = fourier._WINDOW_IDS[windowType];
  const out = [];
  let start = 0;
  while (start < frames.length) {
    const len = frames[start].length;
    let end = start + 1;
    while (end < frames.length && frames[end].length === len) end++;
    const batch = end - start;
    const targetSize = opts.fftSize === undefined || opts.fftSize === null ? core.nextPowerOfTwo(len) : opts.fftSize;
    // error order of spectrum.ts:114-132: FFT ctor, createWindow, binFrequencies
    if (!core.isPowerOfTwo(targetSize)) throw new Error('FFT size must be power of two, got ' + targetSize);
    if (targetSize !== 1 && !Object.prototype.hasOwnProperty.call(fourier._WINDOW_IDS, windowType)) {
      throw new Error('Unsupported window type: ' + windowType);
    }
    if (sampleRate <= 0) throw new Error('Sample rate must be positive, got ' + sampleRate);
    const bins = one ? Math.floor(targetSize / 2) + 1 : targetSize;
    const frequencies = new Float64Array(bins);
    const amplitude = new Float64Array(batch * bins);
    const phase = new Float64Array(batch * bins);
    const peaks = new Float64Array(4 * batch);
    // a run of Float64Arrays -- or of Float32Arrays, the usual form of audio frames -- is read where it lies (one
    // pointer per frame); anything else -- plain arrays, other typed arrays, mixed kinds, holes read as 0 -- is
    // flattened to f64 first
    const Kind = frames[start] instanceof Float32Array ? Float32Array : Float64Array;
    let inPlace = Array.isArray(frames);
    for (let b = start; b < end && inPlace; b++) inPlace = frames[b] instanceof Kind;
    if (inPlace) {
      native.spectrumRows(frames, start, batch, len, sampleRate, targetSize, id === undefined ? 0 : id, one ? 0 : 1,
        frequencies, amplitude, phase, peaks);
    } else {
      const flat = new Float64Array(batch * len);
      for (let b = 0; b < batch; b++) flat.set(core._toF64(frames[start + b]), b * len);
      native.spectrumBatch(flat, batch, len, sampleRate, targetSize, id === undefined ? 0 : id, one ? 0 : 1,
        frequencies, amplitude, phase, peaks);
    }
    for (let b = 0; b < batch; b++) {
      out.push({
        frequencies: b === 0 ? frequencies : frequencies.slice(),
        amplitude: amplitude.subarray(b * bins, (b + 1) * bins),
        phase: phase.subarray(b * bins, (b + 1) * bins),
        peak: { index: peaks[4 * b], frequency: peaks[4 * b + 1], amplitude: peaks[4 * b + 2], phase: peaks[4 * b + 3] },
      });
    }
    start = end;
  }
  return out;
}

// spectrumStream(frames, options, batchFrames): the frame-at-a-time contract of the reference's only streaming
// caller (src/effect/index.ts:190-194: `Stream.map(frames, spectrum)` -- one result per frame, in input order; an
// empty input yields nothing, test/reallife/effect.test.ts:136-146) as a synchronous generator over any iterable
// of frames.  Frames are gathered into batches of `batchFrames` (default 256) and each batch is one spectrumBatch()
// call -- one device batch per run of equal lengths, plans and windows cached by the engine (the Map<size, FFT> /
// Map<"type:size", window> of FourierLive, index.ts:30-48) -- so a result is yielded at most batchFrames frames
// after its input was drawn.  Each frame is copied when it is drawn (a producer may refill one buffer per frame).
function* spectrumStream(frames, options, batchFrames) {
  const limit = batchFrames === undefined || batchFrames === null ? 256 : batchFrames;
  if (!(limit >= 1)) throw new Error('batchFrames must be >= 1, got ' + limit);
  let pending = [];
  for (const f of frames) {
    const x = core._toF64(f);
    pending.push(x === f ? x.slice() : x);  // _toF64 already made a fresh array unless f is a Float64Array
    if (pending.length >= limit) {
      const out = spectrumBatch(pending, options);
      pending = [];
      yield* out;
    }
  }
  if (pending.length) yield* spectrumBatch(pending, options);
}

module.exports = { spectrum, spectrumBatch, spectrumStream };

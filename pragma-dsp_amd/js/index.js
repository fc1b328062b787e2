'use strict';
// Export map mirroring the three reference surfaces that sit on the hot path:
//   require('.../js')            -> { spectrum }            (pragma-dsp)
//   require('.../js').core       -> pragma-dsp/core
//   require('.../js').fourier    -> pragma-dsp/xform/fourier
const core = require('./core');
const fourier = require('./fourier');
const s = require('./spectrum');

module.exports = {
  spectrum: s.spectrum,
  spectrumBatch: s.spectrumBatch,  // extension: spectrumStream's map as one device batch
  spectrumStream: s.spectrumStream,  // extension: the same, frame at a time over any iterable (batched inside)
  core: {
    createComplexArray: core.createComplexArray,
    isPowerOfTwo: core.isPowerOfTwo,
    nextPowerOfTwo: core.nextPowerOfTwo,
    Radix2Fft: core.Radix2Fft,
  },
  fourier: {
    createWindow: fourier.createWindow,
    applyWindow: fourier.applyWindow,
    FFT: fourier.FFT,
    magnitude: fourier.magnitude,
    phase: fourier.phase,
    fftShift: fourier.fftShift,
    fftShiftComplex: fourier.fftShiftComplex,
    binFrequencies: fourier.binFrequencies,
  },
};

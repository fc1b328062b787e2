'use strict';
// Driver for tests/test_js_host.py: runs a list of JSON-described calls through the
// JS drop-in (pragma-dsp_amd/js) and writes the results as JSON.
const fs = require('fs');
const path = require('path');
const p = require(path.join(__dirname, '..', '..', 'pragma-dsp_amd', 'js'));

const cases = JSON.parse(fs.readFileSync(process.argv[2], 'utf8'));
const arr = (a) => Array.from(a);
const cplx = (c) => ({ real: arr(c.real), imag: arr(c.imag) });
const plans = {};
const plan = (n) => plans[n] || (plans[n] = new p.fourier.FFT(n));

const results = cases.map((c) => {
  try {
    switch (c.op) {
      case 'exports':
        return { root: Object.keys(p), core: Object.keys(p.core), fourier: Object.keys(p.fourier) };
      case 'forward':
        return cplx(plan(c.n).forward(c.input));
      case 'forwardTyped':
        return cplx(new p.core.Radix2Fft(c.n).forward(Float32Array.from(c.input)));
      case 'forwardComplex':
        return cplx(plan(c.n).forwardComplex({ real: Float64Array.from(c.real), imag: Float64Array.from(c.imag) }));
      case 'inverse':
        return cplx(plan(c.n).inverse({ real: c.real, imag: c.imag }));
      case 'outIdentity': {
        const fft = plan(c.n);
        const out = fft.createComplexArray();
        const r = fft.forward(c.input, out);
        const back = fft.inverse(r);
        return { same: r === out, filled: out.real.some((v) => v !== 0), roundTrip: arr(back.real),
                 size: fft.size, fill: arr(fft.createComplexArray(2).imag).slice(0, 2) };
      }
      case 'spectrum': {
        const r = p.spectrum(c.samples, c.options);
        return { frequencies: arr(r.frequencies), amplitude: arr(r.amplitude), phase: arr(r.phase), peak: r.peak };
      }
      case 'spectrumBatch': {
        // every result of the batch call must equal spectrum(frame, options) exactly, in order
        const rs = p.spectrumBatch(c.frames, c.options);
        let same = rs.length === c.frames.length;
        const eq = (a, b) => a.length === b.length && a.every((v, i) => Object.is(v, b[i]) || v === b[i]);
        for (let i = 0; same && i < rs.length; i++) {
          const one = p.spectrum(c.frames[i], c.options);
          same = eq(rs[i].frequencies, one.frequencies) && eq(rs[i].amplitude, one.amplitude) &&
            eq(rs[i].phase, one.phase) && rs[i].peak.index === one.peak.index &&
            rs[i].peak.frequency === one.peak.frequency && rs[i].peak.amplitude === one.peak.amplitude &&
            rs[i].peak.phase === one.peak.phase;
        }
        return { same: same, count: rs.length, bins: rs.map((r) => r.amplitude.length),
                 peak0: rs.length ? rs[0].peak : null, empty: p.spectrumBatch([], c.options).length };
      }
      case 'spectrumBatchFull':
        // every result in full, for a direct comparison with the CPU oracle on the Python side
        return p.spectrumBatch(c.frames, c.options).map((r) => ({
          frequencies: arr(r.frequencies), amplitude: arr(r.amplitude), phase: arr(r.phase), peak: r.peak }));
      case 'createWindow':
        return arr(p.fourier.createWindow(c.type, c.size));
      case 'applyWindow':
        return arr(p.fourier.applyWindow(c.input, c.window));
      case 'magnitude':
        return arr(p.fourier.magnitude({ real: c.real, imag: c.imag }));
      case 'phase':
        return arr(p.fourier.phase({ real: c.real, imag: c.imag }));
      case 'misc':
        return {
          next: [0, 1, 5, 1000, 1025].map(p.core.nextPowerOfTwo),
          pow2: [0, 1, 8, 12, 2.5].map(p.core.isPowerOfTwo),
          shift: arr(p.fourier.fftShift([0, 1, 2, 3, 4])),
          freqs: arr(p.fourier.binFrequencies(8, 48000, 'two')),
          cfill: arr(p.core.createComplexArray(3, 7).imag),
        };
      case 'throws':
        switch (c.what) {
          case 'size12': new p.core.Radix2Fft(12); break;
          case 'fft0': new p.fourier.FFT(0); break;
          case 'win0': p.fourier.createWindow('hann', 0); break;
          case 'winType': p.fourier.createWindow('kaiser', 8); break;
          case 'winLen': p.fourier.applyWindow([1, 2, 3], [1, 2]); break;
          case 'binSize': p.fourier.binFrequencies(0, 1); break;
          case 'binRate': p.fourier.binFrequencies(8, -1); break;
          case 'specRate': p.spectrum([1, 2, 3, 4], { sampleRate: 0 }); break;
          case 'specSize': p.spectrum([1, 2, 3, 4], { fftSize: 12, sampleRate: -1 }); break;
          case 'specWin': p.spectrum([1, 2, 3, 4], { window: 'kaiser', sampleRate: -1 }); break;
          case 'inputLen': plan(8).forward([1, 2, 3]); break;
          default: throw new Error('unknown throws case');
        }
        return { error: null };
      default:
        throw new Error('unknown op ' + c.op);
    }
  } catch (e) {
    return { error: e.message };
  }
});
fs.writeFileSync(process.argv[3], JSON.stringify(results));

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
      case 'spectrumBatchTyped': {
        // frames held as Float64Arrays are read where they lie (native.spectrumRows); `mixed` keeps every third one a
        // plain array, which sends its run through the flattening path: same results either way
        const frames = c.frames.map((f, i) => (c.mixed && i % 3 === 1 ? f : Float64Array.from(f)));
        const rs = p.spectrumBatch(frames, c.options);
        const plain = p.spectrumBatch(c.frames, c.options);
        const eq = (a, b) => a.length === b.length && a.every((v, i) => Object.is(v, b[i]) || v === b[i]);
        let same = rs.length === plain.length;
        for (let i = 0; same && i < rs.length; i++) {
          same = eq(rs[i].frequencies, plain[i].frequencies) && eq(rs[i].amplitude, plain[i].amplitude) &&
            eq(rs[i].phase, plain[i].phase) && rs[i].peak.index === plain[i].peak.index &&
            rs[i].peak.amplitude === plain[i].peak.amplitude && rs[i].peak.phase === plain[i].peak.phase;
        }
        // Float32Array frames (audio) are read where they lie too: results equal those of the same values as plain
        // arrays (widening float -> double is exact); a run that mixes the two kinds falls back to flattening
        const asF32 = c.frames.map((f) => Float32Array.from(f));
        const r32 = p.spectrumBatch(asF32, c.options);
        const widened = p.spectrumBatch(asF32.map((f) => Array.from(f)), c.options);
        const mixedKinds = p.spectrumBatch(asF32.map((f, i) => (i % 2 ? Float64Array.from(f) : f)), c.options);
        for (let i = 0; same && i < r32.length; i++) {
          same = eq(r32[i].amplitude, widened[i].amplitude) && eq(r32[i].phase, widened[i].phase) &&
            r32[i].peak.index === widened[i].peak.index && eq(mixedKinds[i].amplitude, widened[i].amplitude);
        }
        return { same: same, count: rs.length };
      }
      case 'spectrumBatchOverlap': {
        // a short-time transform the JS way: frames are OVERLAPPING subarray views of one signal (hop = n / 4); the
        // addon reads each view where it lies -- no copies -- and every result equals spectrum(view)
        const sig = new Float64Array(c.n * 6);
        for (let i = 0; i < sig.length; i++) sig[i] = Math.sin(2 * Math.PI * (40 + i / 997) * i / c.n) + 0.01 * ((i * 7919) % 13);
        const hop = c.n / 4, frames = [];
        for (let o = 0; o + c.n <= sig.length; o += hop) frames.push(sig.subarray(o, o + c.n));
        const rs = p.spectrumBatch(frames, c.options);
        const eq = (a, b) => a.length === b.length && a.every((v, i) => Object.is(v, b[i]) || v === b[i]);
        let same = rs.length === frames.length;
        for (let i = 0; same && i < rs.length; i++) {
          const one = p.spectrum(frames[i], c.options);
          same = eq(rs[i].amplitude, one.amplitude) && eq(rs[i].phase, one.phase) && rs[i].peak.index === one.peak.index;
        }
        return { same: same, count: rs.length };
      }
      case 'spectrumBatchBig': {
        // enough frames for the library to cut the call into chunks on several workers: every sampled result must
        // still equal the one-frame spectrum() exactly, from Float64Array frames and from plain arrays
        let seed = 12345;
        const rnd = () => { seed = (seed * 1103515245 + 12345) % 2147483648; return seed / 2147483648 - 0.5; };
        const frames = [];
        for (let b = 0; b < c.count; b++) {
          const f = new Float64Array(c.n);
          const k = 3 + (b % 97);
          for (let i = 0; i < c.n; i++) f[i] = Math.sin(2 * Math.PI * k * i / c.n) + 0.1 * rnd();
          frames.push(f);
        }
        const eq = (a, b) => a.length === b.length && a.every((v, i) => Object.is(v, b[i]) || v === b[i]);
        const rs = p.spectrumBatch(frames, c.options);
        const plain = p.spectrumBatch(frames.map((f) => Array.from(f)), c.options);
        let same = rs.length === c.count && plain.length === c.count;
        const step = Math.max(1, Math.floor(c.count / 23));
        for (let i = 0; same && i < c.count; i += (i + step < c.count || i === c.count - 1 ? step : c.count - 1 - i)) {
          const one = p.spectrum(frames[i], c.options);
          same = eq(rs[i].amplitude, one.amplitude) && eq(rs[i].phase, one.phase) && rs[i].peak.index === one.peak.index &&
            eq(plain[i].amplitude, one.amplitude) && eq(plain[i].phase, one.phase) && plain[i].peak.index === one.peak.index;
        }
        const f32frames = frames.map((f) => Float32Array.from(f));
        const r32 = p.spectrumBatch(f32frames, c.options);
        for (const i of [0, 1, c.count >> 1, c.count - 1]) {
          const one = p.spectrum(f32frames[i], c.options);
          same = same && eq(r32[i].amplitude, one.amplitude) && eq(r32[i].phase, one.phase) && r32[i].peak.index === one.peak.index;
        }
        return { same: same, count: rs.length, lastPeak: rs[c.count - 1].peak.index };
      }
      case 'transformBatch': {
        // forwardBatch / forwardComplexBatch / inverseBatch: element i equals the one-row call on inputs[i] exactly
        // (c.n <= 8192: the kernel does not depend on the row count), also when the call is large enough to be cut
        // into chunks on the library's workers (c.count rows generated here)
        const fft = new p.fourier.FFT(c.n);
        let seed = 4242;
        const rnd = () => { seed = (seed * 1103515245 + 12345) % 2147483648; return seed / 2147483648 - 0.5; };
        const rows = [];
        for (let b = 0; b < c.count; b++) { const f = new Float64Array(c.n); for (let i = 0; i < c.n; i++) f[i] = rnd(); rows.push(f); }
        const eq = (a, b) => a.length === b.length && a.every((v, i) => Object.is(v, b[i]) || v === b[i]);
        const fwd = fft.forwardBatch(rows);
        const pick = [0, 1, Math.floor(c.count / 2), c.count - 1];
        let same = fwd.length === c.count;
        for (const i of pick) { const one = fft.forward(rows[i]); same = same && eq(fwd[i].real, one.real) && eq(fwd[i].imag, one.imag); }
        // Float64Array rows are read where they lie (native.transformRows); a batch with plain arrays in it is
        // flattened first (native.transformBatch): same values
        const nm = Math.min(c.count, 40);
        const mixed = fft.forwardBatch(rows.slice(0, nm).map((r, i) => (i % 2 ? Array.from(r) : r)));
        for (const i of [0, 1, nm - 1]) same = same && eq(mixed[i].real, fwd[i].real) && eq(mixed[i].imag, fwd[i].imag);
        const cplx = rows.map((r, i) => ({ real: r, imag: rows[(i + 1) % c.count] }));
        const fc = fft.forwardComplexBatch(cplx);
        for (const i of pick) { const one = fft.forwardComplex(cplx[i]); same = same && eq(fc[i].real, one.real) && eq(fc[i].imag, one.imag); }
        const back = fft.inverseBatch(fc);
        let worst = 0;
        for (const i of pick) {
          const one = fft.inverse(fc[i]);
          same = same && eq(back[i].real, one.real) && eq(back[i].imag, one.imag);
          for (let k = 0; k < c.n; k++) worst = Math.max(worst, Math.abs(back[i].real[k] - cplx[i].real[k]), Math.abs(back[i].imag[k] - cplx[i].imag[k]));
        }
        let threw = null;
        try { fft.forwardBatch([rows[0], new Float64Array(3)]); } catch (e) { threw = e.message; }
        const plain = new p.core.Radix2Fft(8).forwardBatch([[0, 1, 0, -1, 0, 1, 0, -1], [1, , 1, 1, 1, 1, 1, 1]]);
        return { same: same, count: fwd.length, roundTrip: worst, threw: threw, empty: fft.forwardBatch([]).length,
                 sineBin2: arr(plain[0].imag), hole: arr(plain[1].real) };
      }
      case 'spectrumStream': {
        // a producer that refills ONE buffer per frame (frames are copied when drawn), batches of c.batchFrames:
        // result i must equal spectrum(frames[i], options) exactly, in order; an empty iterable yields nothing
        const eq = (a, b) => a.length === b.length && a.every((v, i) => Object.is(v, b[i]) || v === b[i]);
        let drawn = 0;
        const yieldedAfter = [];
        function* producer() {
          const buf = new Float64Array(c.frames[0].length);
          for (const f of c.frames) {
            if (f.length === buf.length) { buf.set(f); drawn++; yield buf; } else { drawn++; yield f; }
          }
        }
        let same = true, count = 0;
        for (const r of p.spectrumStream(producer(), c.options, c.batchFrames)) {
          const one = p.spectrum(c.frames[count], c.options);
          same = same && eq(r.amplitude, one.amplitude) && eq(r.phase, one.phase) && eq(r.frequencies, one.frequencies) &&
            r.peak.index === one.peak.index && r.peak.amplitude === one.peak.amplitude;
          yieldedAfter.push(drawn);
          count++;
        }
        let threw = null;
        try { Array.from(p.spectrumStream(c.frames, c.options, 0)); } catch (e) { threw = e.message; }
        return { same: same, count: count, yieldedAfter: yieldedAfter,
                 empty: Array.from(p.spectrumStream([], c.options)).length, threw: threw };
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

// pdsp_oracle.js -- TEST INFRASTRUCTURE ONLY (never imported by the product).
//
// A Node-12-compatible restatement of the reference's transform, used for one purpose: timing
// "the Node CPU path" next to the GPU numbers (SURVEY 8(d): the reference itself is TypeScript and
// cannot be loaded by this image's Node 12).  Same algorithm and the same kind of code the
// reference runs under V8 -- f64 typed arrays, a bit-reversal scatter, log2(N) in-place radix-2
// DIT stages with per-stage cos/sin tables filled by Math.cos / Math.sin, 1/N on the inverse
// (src/core/fft.ts:25-38, 45-61, 110-148) -- written from that description, not copied.
// Pinned against the reference's goldens by tests/test_oracle_golden.py::test_node_oracle_*.
//
// Also the whole one-shot spectrum() call (src/public/spectrum.ts:107-142) restated WITH its per-call
// costs -- a fresh plan (bit-reversal table + 2(N-1) cos/sin: fft.ts:25-61, rebuilt by `new FFT` on
// every call, spectrum.ts:114), a fresh window (fourier.ts:14-52), Math.hypot / Math.atan2 per bin
// (fourier.ts:98-120), the scaling, frequency axis and findPeak -- so that the drop-in's one-frame
// latency has the reference's own one-frame cost beside it, not just the bare transform.
//
//   node oracle/pdsp_oracle.js fft            < {"cases":[{"n","re","im"|null,"inverse"}]}  > outputs
//   node oracle/pdsp_oracle.js spectrum       < {"cases":[{"samples","options"}]}           > results
//   node oracle/pdsp_oracle.js time N ROWS S  -> {"transforms","seconds","checksum","node"}
'use strict';

function makePlan(n) {
  if (!(n > 0) || (n & (n - 1)) !== 0) throw new Error('FFT size must be power of two, got ' + n);
  let bits = 0;
  while ((1 << bits) < n) bits++;
  const rev = new Uint32Array(n);
  for (let i = 0; i < n; i++) {
    let r = 0;
    for (let b = 0; b < bits; b++) if (i & (1 << b)) r |= 1 << (bits - 1 - b);
    rev[i] = r;
  }
  const stages = [];  // stage s (span m = 2^s): cos/sin of -2*pi*k/m, k < m/2
  for (let m = 2; m <= n; m *= 2) {
    const half = m / 2, c = new Float64Array(half), s = new Float64Array(half);
    for (let k = 0; k < half; k++) {
      const a = (-2 * Math.PI * k) / m;
      c[k] = Math.cos(a);
      s[k] = Math.sin(a);
    }
    stages.push({ m: m, c: c, s: s });
  }
  return { n: n, rev: rev, stages: stages };
}

// im may be null (real input).  outRe/outIm must not alias the inputs.
function transform(plan, re, im, outRe, outIm, inverse) {
  const n = plan.n, rev = plan.rev;
  if (re.length !== n) throw new Error('FFT input length ' + re.length + ' != size ' + n);
  if (im && im.length !== n) throw new Error('FFT input length ' + im.length + ' != size ' + n);
  for (let i = 0; i < n; i++) {
    const j = rev[i];
    outRe[j] = re[i];
    outIm[j] = im ? im[i] : 0;
  }
  const sign = inverse ? -1 : 1;
  for (let si = 0; si < plan.stages.length; si++) {
    const st = plan.stages[si], m = st.m, half = m >> 1, c = st.c, s = st.s;
    for (let base = 0; base < n; base += m) {
      for (let j = 0; j < half; j++) {
        const wr = c[j], wi = sign * s[j];
        const lo = base + j, hi = lo + half;
        const xr = outRe[hi], xi = outIm[hi];
        const tr = wr * xr - wi * xi, ti = wr * xi + wi * xr;
        const ur = outRe[lo], ui = outIm[lo];
        outRe[lo] = ur + tr;
        outIm[lo] = ui + ti;
        outRe[hi] = ur - tr;
        outIm[hi] = ui - ti;
      }
    }
  }
  if (inverse) {
    const k = 1 / n;
    for (let i = 0; i < n; i++) {
      outRe[i] *= k;
      outIm[i] *= k;
    }
  }
}

function makeWindow(type, n) {
  if (!(n > 0)) throw new Error('Window size must be positive, got ' + n);
  const w = new Float64Array(n);
  if (n === 1) { w[0] = 1; return w; }
  const d = n - 1;
  for (let i = 0; i < n; i++) {
    const f = (2 * Math.PI * i) / d;
    if (type === 'rect') w[i] = 1;
    else if (type === 'hann') w[i] = 0.5 * (1 - Math.cos(f));
    else if (type === 'hamming') w[i] = 0.54 - 0.46 * Math.cos(f);
    else if (type === 'blackman') w[i] = 0.42 - 0.5 * Math.cos(f) + 0.08 * Math.cos(2 * f);
    else throw new Error('Unsupported window type: ' + type);
  }
  return w;
}

function nextPow2(n) {
  if (n <= 1) return 1;
  let p = 1;
  while (p < n) p *= 2;
  return p;
}

// One spectrum() call the way the reference makes it: nothing is cached between calls.
function spectrum(samples, options) {
  const o = options || {};
  const sampleRate = o.sampleRate === undefined ? 1 : o.sampleRate;
  const sides = o.sides === undefined ? 'one' : o.sides;
  const n = o.fftSize === undefined ? nextPow2(samples.length) : o.fftSize;
  const plan = makePlan(n);                                    // new FFT(targetSize)
  const win = makeWindow(o.window === undefined ? 'rect' : o.window, n);
  const frame = new Float64Array(n);                           // buildFrame: truncate / zero-pad
  const lim = Math.min(n, samples.length);
  for (let i = 0; i < lim; i++) frame[i] = samples[i] === undefined ? 0 : samples[i];
  const windowed = new Float64Array(n);
  for (let i = 0; i < n; i++) windowed[i] = frame[i] * win[i];
  const re = new Float64Array(n), im = new Float64Array(n);
  transform(plan, windowed, null, re, im, false);
  const mag = new Float64Array(n), ang = new Float64Array(n);
  for (let i = 0; i < n; i++) {
    mag[i] = Math.hypot(re[i], im[i]);
    ang[i] = Math.atan2(im[i], re[i]);
  }
  const bins = sides === 'one' ? Math.floor(n / 2) + 1 : n;
  const amp = new Float64Array(bins);
  const nyq = n % 2 === 0 ? n / 2 : -1;
  for (let k = 0; k < bins; k++) {
    if (sides !== 'one' || k === 0 || k === nyq) amp[k] = mag[k] / n;
    else amp[k] = (2 * mag[k]) / n;
  }
  const phase = sides === 'one' ? ang.slice(0, bins) : ang;
  if (!(n > 0)) throw new Error('FFT size must be positive, got ' + n);
  if (!(sampleRate > 0)) throw new Error('Sample rate must be positive, got ' + sampleRate);
  const freq = new Float64Array(bins);
  for (let i = 0; i < bins; i++) freq[i] = (i * sampleRate) / n;
  let maxIndex = 0, maxValue = amp[0], hasNonDc = false, nonDcIndex = 0, nonDcValue = 0;  // findPeak
  for (let i = 1; i < bins; i++) {
    const v = amp[i];
    if (v > nonDcValue) { nonDcValue = v; nonDcIndex = i; }
    if (v > 0) hasNonDc = true;
    if (v > maxValue) { maxValue = v; maxIndex = i; }
  }
  const idx = hasNonDc ? nonDcIndex : maxIndex;
  return { frequencies: freq, amplitude: amp, phase: phase,
           peak: { index: idx, frequency: freq[idx], amplitude: amp[idx], phase: phase[idx] } };
}

function readStdin() {
  return require('fs').readFileSync(0, 'utf8');
}

function main(argv) {
  const mode = argv[2];
  if (mode === 'fft') {
    const req = JSON.parse(readStdin());
    const res = req.cases.map(function (cs) {
      const plan = makePlan(cs.n);
      const oRe = new Float64Array(cs.n), oIm = new Float64Array(cs.n);
      transform(plan, Float64Array.from(cs.re), cs.im ? Float64Array.from(cs.im) : null, oRe, oIm, !!cs.inverse);
      return { re: Array.from(oRe), im: Array.from(oIm) };
    });
    process.stdout.write(JSON.stringify({ results: res }));
    return 0;
  }
  if (mode === 'spectrum') {
    const req = JSON.parse(readStdin());
    const res = req.cases.map(function (cs) {
      const r = spectrum(cs.samples, cs.options);
      return { frequencies: Array.from(r.frequencies), amplitude: Array.from(r.amplitude), phase: Array.from(r.phase), peak: r.peak };
    });
    process.stdout.write(JSON.stringify({ results: res }));
    return 0;
  }
  if (mode === 'time') {
    // Loop shape of bench/run.ts:13-26: plan and out reused, a checksum defeats dead-code elimination.
    const n = parseInt(argv[3], 10), rows = parseInt(argv[4], 10), seconds = parseFloat(argv[5]);
    const complexInput = argv[6] === 'complex';
    const plan = makePlan(n);
    let seed = 1337;
    const rnd = function () {  // xorshift32 -> uniform(-1, 1); timing does not depend on the values
      seed ^= seed << 13; seed ^= seed >>> 17; seed ^= seed << 5;
      return (seed >>> 0) / 2147483648 - 1;
    };
    const re = [], im = [];
    for (let r = 0; r < rows; r++) {
      const a = new Float64Array(n), b = new Float64Array(n);
      for (let i = 0; i < n; i++) { a[i] = rnd(); b[i] = rnd(); }
      re.push(a);
      im.push(complexInput ? b : null);
    }
    const oRe = new Float64Array(n), oIm = new Float64Array(n);
    let chk = 0;
    for (let w = 0; w < Math.min(rows, 64); w++) transform(plan, re[w], im[w], oRe, oIm, false);  // warm-up (JIT)
    const t0 = process.hrtime.bigint();
    let done = 0, el = 0;
    do {
      for (let r = 0; r < rows; r++) {
        transform(plan, re[r], im[r], oRe, oIm, false);
        chk += oRe[1] + oIm[n - 1];
      }
      done += rows;
      el = Number(process.hrtime.bigint() - t0) / 1e9;
    } while (el < seconds);
    process.stdout.write(JSON.stringify({ n: n, transforms: done, seconds: el, checksum: chk, node: process.version }));
    return 0;
  }
  process.stderr.write('usage: pdsp_oracle.js fft | time N ROWS SECONDS [complex]\n');
  return 2;
}

if (require.main === module) process.exitCode = main(process.argv);
module.exports = { makePlan: makePlan, transform: transform, spectrum: spectrum };

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
//   node oracle/pdsp_oracle.js fft            < {"cases":[{"n","re","im"|null,"inverse"}]}  > outputs
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
module.exports = { makePlan: makePlan, transform: transform };

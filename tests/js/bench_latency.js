'use strict';
// Single-call latency of the JS drop-in, in the shape of the reference's own benchmark
// (bench/run.ts:12-28: one plan, `out` reused, a checksum over the result so the work cannot be
// elided), with the Node CPU restatement of the same algorithm (oracle/pdsp_oracle.js) timed
// next to it when it is present.  Prints one JSON line.
// Run by `bench.py --workload single1024` (its cpu_baseline leg) or by hand:
//   node tests/js/bench_latency.js [iterations]
const path = require('path');
const p = require(path.join(__dirname, '..', '..', 'pragma-dsp_amd', 'js'));
const cpu = require(path.join(__dirname, '..', '..', 'oracle', 'pdsp_oracle.js'));  // CPU baseline + checker

const iters = parseInt(process.argv[2] || '2000', 10);
// optional: a comma-separated list of sizes (default: all four, plus the spectrumBatch rows)
const only = process.argv[3] ? process.argv[3].split(',').map((v) => parseInt(v, 10)) : null;
const now = () => Number(process.hrtime.bigint()) / 1e3;  // microseconds

function stats(ts) {
  ts.sort((a, b) => a - b);
  return { median_us: ts[ts.length >> 1], min_us: ts[0], p95_us: ts[Math.floor(ts.length * 0.95)] };
}

function timed(fn) {
  for (let i = 0; i < 50; i++) fn();
  const ts = new Array(iters);
  for (let i = 0; i < iters; i++) {
    const t0 = now();
    fn();
    ts[i] = now() - t0;
  }
  return stats(ts);
}

// batch forms: per-call times over `reps` calls -> median / min per element (a mean over 20 calls swung by 30 %)
function timedCalls(fn, reps, per) {
  for (let w = 0; w < 8; w++) fn();
  const ts = new Array(reps);
  for (let r = 0; r < reps; r++) {
    const t0 = now();
    fn();
    ts[r] = (now() - t0) / per;
  }
  return stats(ts);
}

const out = { node: process.version, iterations: iters, cases: [] };
let seed = 1337;
const rnd = () => { seed ^= seed << 13; seed ^= seed >>> 17; seed ^= seed << 5; return (seed >>> 0) / 2147483648 - 1; };
for (const n of [1024, 2048, 4096, 16384]) {
  if (only && only.indexOf(n) < 0) continue;
  const input = new Float64Array(n);
  for (let i = 0; i < n; i++) input[i] = rnd();
  const fft = new p.fourier.FFT(n);
  const res = fft.createComplexArray();
  let checksum = 0;
  const gpu = timed(() => {
    const r = fft.forward(input, res);
    checksum += r.real[1] * 0.001 + r.imag[n - 1] * 0.002;
  });
  const row = { n: n, op: 'FFT.forward', gpu_dropin: gpu };
  if (cpu) {
    const plan = cpu.makePlan(n), oRe = new Float64Array(n), oIm = new Float64Array(n);
    row.node_cpu = timed(() => {
      cpu.transform(plan, input, null, oRe, oIm, false);
      checksum += oRe[1] * 0.001 + oIm[n - 1] * 0.002;
    });
    let d = 0, m = 0;
    fft.forward(input, res);
    for (let i = 0; i < n; i++) {
      d = Math.max(d, Math.abs(res.real[i] - oRe[i]), Math.abs(res.imag[i] - oIm[i]));
      m = Math.max(m, Math.abs(oRe[i]), Math.abs(oIm[i]));
    }
    row.max_abs_diff_over_max = d / m;
  }
  out.cases.push(row);
  const sp = timed(() => {
    const r = p.spectrum(input, { sampleRate: 48000, fftSize: n, window: 'hann' });
    checksum += r.peak.amplitude;
  });
  const srow = { n: n, op: 'spectrum(hann, one-sided)', gpu_dropin: sp };
  if (cpu && cpu.spectrum) {
    // the reference's own one-shot cost: plan and window rebuilt per call (spectrum.ts:114-116)
    srow.node_cpu = timed(() => {
      const r = cpu.spectrum(input, { sampleRate: 48000, fftSize: n, window: 'hann' });
      checksum += r.peak.amplitude;
    });
  }
  out.cases.push(srow);
}
// spectrumBatch: 256 frames per call
for (const n of [1024, 4096]) {
  if (only) continue;
  const frames = [];
  for (let b = 0; b < 256; b++) {
    const f = new Float64Array(n);
    for (let i = 0; i < n; i++) f[i] = rnd();
    frames.push(f);
  }
  const opts = { sampleRate: 48000, fftSize: n, window: 'hann' };
  const reps = 80;
  let acc = 0;
  const st = timedCalls(() => { acc += p.spectrumBatch(frames, opts)[255].peak.amplitude; }, reps, 256);
  out.cases.push({ n: n, op: 'spectrumBatch(256 frames, hann) per frame', gpu_dropin: st, guard: acc });
  // the same frames as Float32Arrays (audio): read where they lie through pdsp_spectrum_rows_host_f32in
  const f32 = frames.map((f) => Float32Array.from(f));
  const st32 = timedCalls(() => { acc += p.spectrumBatch(f32, opts)[255].peak.amplitude; }, reps, 256);
  out.cases.push({ n: n, op: 'spectrumBatch(256 Float32Array frames, hann) per frame', gpu_dropin: st32, guard: acc });
}
// FFT.forwardBatch: the reference's batch idiom (bench/reallife/signals.ts:264-270, `for (...) fft.forward(input)`)
// as one call, 256 rows, beside the same rows through 256 forward() calls and through the Node CPU path
for (const n of [1024, 4096]) {
  if (only) continue;
  const rows = [];
  for (let b = 0; b < 256; b++) {
    const f = new Float64Array(n);
    for (let i = 0; i < n; i++) f[i] = rnd();
    rows.push(f);
  }
  const fft = new p.fourier.FFT(n);
  let acc = 0;
  const stb = timedCalls(() => { acc += fft.forwardBatch(rows)[255].real[1]; }, 80, 256);
  const outc = fft.createComplexArray();
  let t0 = now();
  for (let r = 0; r < 4; r++) for (let b = 0; b < 256; b++) acc += fft.forward(rows[b], outc).real[1];
  const usLoop = (now() - t0) / (4 * 256);
  const row = { n: n, op: 'FFT.forwardBatch(256 rows) per row', gpu_dropin: stb,
                gpu_dropin_loop_of_forward: { median_us: usLoop }, guard: acc };
  if (cpu) {
    const plan = cpu.makePlan(n), oRe = new Float64Array(n), oIm = new Float64Array(n);
    t0 = now();
    for (let r = 0; r < 4; r++) for (let b = 0; b < 256; b++) { cpu.transform(plan, rows[b], null, oRe, oIm, false); acc += oRe[1]; }
    row.node_cpu = { median_us: (now() - t0) / (4 * 256) };
  }
  out.cases.push(row);
}
out.checksum_guard = Number.isFinite(seed) ? 1 : 0;
process.stdout.write(JSON.stringify(out) + '\n');

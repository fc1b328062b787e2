'use strict';
// The reference's hot-path test suite run in the reference's OWN language against the drop-in: every it(...) of
// pragma-dsp's test/fft.test.ts, test/spectrum.test.ts, test/window.test.ts and
// test/reallife/{signals, phase, scaling, edge_cases}.test.ts, with its assertions and tolerances, against
// require('pragma-dsp_amd/js') -- the module a maintainer re-exports from src/core, src/xform/fourier and
// src/public/spectrum (INTEGRATION.md section 2).  No vitest in the image: `it` / `expect` below are a dozen
// lines.  Fixtures arrive as JSON written by tests/test_js_host.py from tests/golden/*.npz.
//   node tests/js/reference_suite.js fixtures.json results.json
const fs = require('fs');
const path = require('path');
const p = require(path.join(__dirname, '..', '..', 'pragma-dsp_amd', 'js'));
const { FFT, magnitude, phase, createWindow } = p.fourier;
const spectrum = p.spectrum;

const fx = JSON.parse(fs.readFileSync(process.argv[2], 'utf8'));
const results = [];
function it(name, fn) {
  try {
    fn();
    results.push({ name: name, ok: true });
  } catch (e) {
    results.push({ name: name, ok: false, error: String(e && e.message ? e.message : e).slice(0, 300) });
  }
}
function fail(msg) { throw new Error(msg); }
const expect = (a) => ({
  toBe: (b) => { if (!Object.is(a, b) && !(a === 0 && b === 0)) fail('expected ' + a + ' to be ' + b); },
  toBeLessThan: (b) => { if (!(a < b)) fail('expected ' + a + ' < ' + b); },
  toBeLessThanOrEqual: (b) => { if (!(a <= b)) fail('expected ' + a + ' <= ' + b); },
  toBeCloseTo: (b, digits) => { if (!(Math.abs(a - b) < 0.5 * Math.pow(10, -digits))) fail('expected ' + a + ' close to ' + b + ' (' + digits + ' digits)'); },
});
const maxAbsErr = (a, b) => { let m = 0; for (let i = 0; i < a.length; i++) m = Math.max(m, Math.abs(a[i] - (b ? b[i] : 0))); return m; };
const closeArray = (a, b, tol) => { expect(a.length).toBe(b.length); expect(maxAbsErr(a, b)).toBeLessThanOrEqual(tol); };
const wrap = (d) => { d = Math.abs(d); return Math.min(d, Math.abs(d - 2 * Math.PI)); };
const find = (pred) => { const c = fx.reallife.find(pred); if (!c) fail('fixture case missing'); return c; };
const opts = (c, sides) => ({ sampleRate: c.sampleRate, fftSize: c.n, window: 'rect', sides: sides || 'one' });

// ---- test/fft.test.ts, test/spectrum.test.ts, test/window.test.ts
for (const c of fx.v01.filter((c) => c.kind === 'random_normal' && [8, 16, 32].indexOf(c.n) >= 0)) {
  it('matches numpy fft for ' + c.name, () => {
    const r = new FFT(c.n).forward(c.input);
    closeArray(r.real, c.fftRe, 1e-6);
    closeArray(r.imag, c.fftIm, 1e-6);
  });
  it('round-trips ' + c.name, () => {
    const fft = new FFT(c.n);
    const back = fft.inverse(fft.forward(c.input));
    closeArray(back.real, c.input, 1e-6);
    closeArray(back.imag, new Array(c.n).fill(0), 1e-6);
  });
}
it('returns correct peak bin + frequency + amplitude', () => {
  const c = fx.v01.find((c) => c.kind === 'sine_bin_centered');
  const r = spectrum(c.input, opts(c));
  expect(r.peak.index).toBe(c.meta.binCenteredK);
  expect(Math.abs(r.peak.frequency - c.meta.expectedPeakHz)).toBeLessThanOrEqual(1e-6);
  expect(Math.abs(r.peak.amplitude - c.meta.amplitude)).toBeLessThanOrEqual(1e-3);
});
for (const w of fx.windows) {
  it('matches ' + w.type + ' window n=' + w.n, () => closeArray(createWindow(w.type, w.n), w.values, 1e-8));
}

// ---- test/reallife/signals.test.ts
for (const c of fx.reallife.filter((c) => c.family === 'pure_sine')) {
  it('matches NumPy FFT for ' + c.name, () => {
    const r = new FFT(c.n).forward(c.signal);
    closeArray(r.real, c.fftRe, 1e-10);
    closeArray(r.imag, c.fftIm, 1e-10);
  });
  it('magnitude matches NumPy for ' + c.name, () => closeArray(magnitude(new FFT(c.n).forward(c.signal)), c.magnitude, 1e-10));
  it('phase matches NumPy for ' + c.name, () => {
    const ph = phase(new FFT(c.n).forward(c.signal));
    for (let i = 0; i < c.n; i++) if (c.magnitude[i] > 1e-6) expect(wrap(ph[i] - c.phase[i])).toBeLessThan(1e-10);
  });
  it('round-trips correctly for ' + c.name, () => {
    const fft = new FFT(c.n);
    const back = fft.inverse(fft.forward(c.signal));
    closeArray(back.real, c.signal, 1e-10);
    expect(maxAbsErr(back.imag)).toBeLessThan(1e-10);
  });
}
for (const c of fx.reallife.filter((c) => c.family === 'multi_tone')) {
  it('matches NumPy FFT for ' + c.name, () => {
    const r = new FFT(c.n).forward(c.signal);
    closeArray(r.real, c.fftRe, 1e-10);
    closeArray(r.imag, c.fftIm, 1e-10);
  });
  it('detects correct peaks for ' + c.name, () => {
    const mag = magnitude(new FFT(c.n).forward(c.signal));
    c.params.bin_indices.forEach((b, i) => expect(mag[b]).toBeCloseTo(c.n * c.params.amplitudes[i] / 2, 5));
  });
}
for (const c of fx.reallife.filter((c) => c.family === 'chirp')) {
  it('matches NumPy FFT for ' + c.name, () => {
    const r = new FFT(c.n).forward(c.signal);
    closeArray(r.real, c.fftRe, 1e-10);
    closeArray(r.imag, c.fftIm, 1e-10);
  });
  it('round-trips correctly for ' + c.name, () => {
    const fft = new FFT(c.n);
    closeArray(fft.inverse(fft.forward(c.signal)).real, c.signal, 1e-10);
  });
}
const flat = (c) => { const mag = magnitude(new FFT(c.n).forward(c.signal)); for (let i = 0; i < c.n; i++) expect(mag[i]).toBeCloseTo(c.params.amplitude, 10); };
const onlyBin = (c, bin, level) => {
  const mag = magnitude(new FFT(c.n).forward(c.signal));
  expect(mag[bin]).toBeCloseTo(c.n * level, 10);
  for (let i = 0; i < c.n; i++) if (i !== bin) expect(mag[i]).toBeLessThan(1e-10);
};
const zerosOut = (c) => { const r = new FFT(c.n).forward(c.signal); for (let i = 0; i < c.n; i++) { expect(r.real[i]).toBe(0); expect(r.imag[i]).toBe(0); } };
it('impulse has flat magnitude spectrum', () => flat(find((c) => c.kind === 'impulse')));
it('DC signal has energy only in bin 0', () => { const c = find((c) => c.kind === 'dc'); onlyBin(c, 0, c.params.level); });
it('Nyquist signal has energy only at Nyquist bin', () => { const c = find((c) => c.kind === 'nyquist'); onlyBin(c, c.n / 2, c.params.amplitude); });
it('zero input gives zero output', () => zerosOut(find((c) => c.kind === 'zeros')));

// ---- test/reallife/phase.test.ts
it('cosine leads sine by 90 degrees at peak bin', () => {
  const s = find((c) => c.kind === 'pure_sine_bin_centered' && c.params.bin_index === 8);
  const k = find((c) => c.kind === 'cosine' && c.params.bin_index === 8);
  const fft = new FFT(s.n);
  let d = phase(fft.forward(k.signal))[8] - phase(fft.forward(s.signal))[8];
  while (d > Math.PI) d -= 2 * Math.PI;
  while (d < -Math.PI) d += 2 * Math.PI;
  expect(Math.abs(d - Math.PI / 2)).toBeLessThan(1e-6);
});
for (const c of fx.reallife.filter((c) => c.kind === 'pure_sine_phase')) {
  it('phase is correct for ' + c.name, () => {
    const b = c.params.bin_index;
    expect(wrap(phase(new FFT(c.n).forward(c.signal))[b] - c.phase[b])).toBeLessThan(1e-10);
  });
  it('spectrum() reports correct peak phase for ' + c.name, () => {
    const r = spectrum(c.signal, opts(c));
    expect(r.peak.index).toBe(c.params.bin_index);
    expect(wrap(r.peak.phase - c.phase[c.params.bin_index])).toBeLessThan(1e-10);
  });
}
it('phase array has correct length for one-sided spectrum', () => { const c = fx.reallife[0]; expect(spectrum(c.signal, opts(c)).phase.length).toBe(c.n / 2 + 1); });
it('phase array has correct length for two-sided spectrum', () => { const c = fx.reallife[0]; expect(spectrum(c.signal, opts(c, 'two')).phase.length).toBe(c.n); });
it('DC phase is 0 for positive DC signal', () => expect(phase(new FFT(64).forward(new Float64Array(64).fill(1.0)))[0]).toBeCloseTo(0, 10));
it('DC phase is pi for negative DC signal', () => expect(Math.abs(phase(new FFT(64).forward(new Float64Array(64).fill(-1.0)))[0])).toBeCloseTo(Math.PI, 10));

// ---- test/reallife/scaling.test.ts
const centered = fx.reallife.filter((c) => c.kind === 'pure_sine_bin_centered');
for (const c of centered) {
  it('returns correct amplitude for ' + c.name, () => {
    const r = spectrum(c.signal, opts(c));
    expect(r.peak.index).toBe(c.params.bin_index);
    expect(r.peak.amplitude).toBeCloseTo(c.params.amplitude, 2);
  });
  it('returns correct amplitude for ' + c.name + ' (two-sided)', () => {
    const amp = spectrum(c.signal, opts(c, 'two')).amplitude;
    expect(amp[c.params.bin_index]).toBeCloseTo(c.params.amplitude / 2, 2);
    expect(amp[c.n - c.params.bin_index]).toBeCloseTo(c.params.amplitude / 2, 2);
  });
  it('peak frequency matches expected for ' + c.name, () => expect(spectrum(c.signal, opts(c)).peak.frequency).toBeCloseTo(c.params.frequency_hz, 6));
}
it('DC bin is not doubled', () => { const c = find((c) => c.kind === 'dc'); expect(spectrum(c.signal, opts(c)).amplitude[0]).toBeCloseTo(c.params.level, 6); });
it('Nyquist bin is not doubled', () => { const c = find((c) => c.kind === 'nyquist'); expect(spectrum(c.signal, opts(c)).amplitude[c.n / 2]).toBeCloseTo(c.params.amplitude, 6); });
it('returns full N bins for two-sided', () => {
  const c = fx.reallife[0];
  const r = spectrum(c.signal, opts(c, 'two'));
  expect(r.amplitude.length).toBe(c.n);
  expect(r.frequencies.length).toBe(c.n);
  expect(r.phase.length).toBe(c.n);
});
it('frequency axis is correctly scaled', () => {
  const c = fx.reallife[0];
  const f = spectrum(c.signal, opts(c)).frequencies;
  expect(f[0]).toBe(0);
  for (let i = 0; i < f.length; i++) expect(f[i]).toBeCloseTo(i * c.sampleRate / c.n, 10);
  expect(f[f.length - 1]).toBeCloseTo(c.sampleRate / 2, 10);
});
it('ignores DC when there are non-DC components', () => { const c = find((c) => c.kind === 'dc_plus_sine'); expect(spectrum(c.signal, opts(c)).peak.index).toBe(c.params.sine_bin); });
it('returns DC as peak when it is the only component', () => { const c = find((c) => c.kind === 'dc'); expect(spectrum(c.signal, opts(c)).peak.index).toBe(0); });

// ---- test/reallife/edge_cases.test.ts
it('FFT of zeros gives zeros', () => zerosOut(find((c) => c.kind === 'zeros')));
it('spectrum() of zeros gives zeros', () => {
  const r = spectrum(new Float64Array(64), { sampleRate: 48000, fftSize: 64, window: 'rect', sides: 'one' });
  for (let i = 0; i < r.amplitude.length; i++) expect(r.amplitude[i]).toBe(0);
  expect(r.peak.amplitude).toBe(0);
});
it('DC signal has energy only in bin 0 (edge cases)', () => { const c = find((c) => c.kind === 'dc'); onlyBin(c, 0, c.params.level); });
it('alternating +1/-1 has energy only at Nyquist', () => { const c = find((c) => c.kind === 'nyquist'); onlyBin(c, c.n / 2, c.params.amplitude); });
it('impulse at position 0 gives flat magnitude spectrum', () => flat(find((c) => c.kind === 'impulse' && c.params.position === 0)));
it('impulse at middle position gives correct phase pattern', () => flat(find((c) => c.kind === 'impulse' && c.params.position > 0)));
it('handles tiny amplitude signals without underflow', () => {
  const c = find((c) => c.kind === 'tiny');
  const r = new FFT(c.n).forward(c.signal);
  for (let i = 0; i < c.n; i++) { expect(Number.isFinite(r.real[i])).toBe(true); expect(Number.isFinite(r.imag[i])).toBe(true); }
  expect(maxAbsErr(r.real, c.fftRe)).toBeLessThan(1e-20);
});
it('handles large amplitude signals without overflow', () => {
  const c = find((c) => c.kind === 'large');
  const r = new FFT(c.n).forward(c.signal);
  for (let i = 0; i < c.n; i++) {
    expect(Number.isFinite(r.real[i])).toBe(true);
    expect(Number.isFinite(r.imag[i])).toBe(true);
    const want = c.fftRe[i];
    if (Math.abs(want) > 1) expect(Math.abs(r.real[i] - want) / Math.abs(want)).toBeLessThan(1e-9);
    else expect(Math.abs(r.real[i] - want)).toBeLessThan(1e-6);
  }
});
it('handles input shorter than fftSize', () => {
  const r = spectrum(new Float64Array([1, 2, 3, 4]), { sampleRate: 48000, fftSize: 16, window: 'rect', sides: 'one' });
  expect(r.amplitude.length).toBe(16 / 2 + 1);
  expect(Number.isFinite(r.peak.amplitude)).toBe(true);
  expect(Number.isFinite(r.peak.frequency)).toBe(true);
});
it('zero-padding preserves signal content', () => {
  const r = spectrum(new Float64Array([1, 1, 1, 1]), { sampleRate: 48000, fftSize: 16, window: 'rect', sides: 'one' });
  expect(r.amplitude[0]).toBeCloseTo(4 / 16, 6);
});
it('IFFT(FFT(x)) = x for all special signals', () => {
  for (const c of fx.reallife.filter((c) => c.family === 'special')) {
    const fft = new FFT(c.n);
    const back = fft.inverse(fft.forward(c.signal));
    expect(maxAbsErr(back.real, c.signal)).toBeLessThan(1e-9);
    expect(maxAbsErr(back.imag)).toBeLessThan(1e-9);
  }
});

fs.writeFileSync(process.argv[3], JSON.stringify(results));

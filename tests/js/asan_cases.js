'use strict';
// Argument handling of the N-API addon under AddressSanitizer / UBSan (tests/test_js_host.py builds an instrumented
// copy; no GPU needed: every call below ends in a validation error or in host-side index math).
const native = require(process.argv[2]);
const log = [];
const attempt = (name, fn) => { try { const r = fn(); log.push([name, 'ok', r === undefined ? null : String(r).slice(0, 60)]); } catch (e) { log.push([name, 'throws', e.message.slice(0, 90)]); } };
const w64 = new Float64Array(64);
attempt('windowMake', () => { native.windowMake(3, 64, w64); return w64[1]; });
attempt('windowMake short out', () => native.windowMake(1, 64, new Float64Array(8)));
attempt('windowMake size 0', () => native.windowMake(1, 0, w64));
attempt('windowMake bad type', () => native.windowMake(9, 8, w64));
attempt('planCreate 12', () => native.planCreate(12));
attempt('planCreate 0', () => native.planCreate(0));
attempt('planCreate -8', () => native.planCreate(-8));
attempt('nextPow2', () => [0, 1, 5, 1025, 2 ** 31 + 5].map(native.nextPow2).join(','));
attempt('binFrequencies', () => { const f = new Float64Array(5); native.binFrequencies(8, 48000, 0, f); return f.join(','); });
attempt('binFrequencies short out', () => native.binFrequencies(8, 48000, 0, new Float64Array(2)));
attempt('binFrequencies rate 0', () => native.binFrequencies(8, 0, 0, new Float64Array(5)));
attempt('fftShift', () => { const o = new Float64Array(5); native.fftShift(Float64Array.from([0, 1, 2, 3, 4]), o); return o.join(','); });
attempt('fftShift short out', () => native.fftShift(Float64Array.from([0, 1, 2, 3, 4]), new Float64Array(2)));
attempt('applyWindow length mismatch', () => native.applyWindow(new Float64Array(3), new Float64Array(2), new Float64Array(3)));
attempt('magnitude short out', () => native.magnitude(new Float64Array(8), new Float64Array(8), new Float64Array(4)));
attempt('phase mismatched planes', () => native.phase(new Float64Array(8), new Float64Array(4), new Float64Array(8)));
attempt('spectrum short out', () => native.spectrum(new Float64Array(8), 48000, -1, 0, 0, new Float64Array(2), new Float64Array(5), new Float64Array(5)));
attempt('spectrum bad size', () => native.spectrum(new Float64Array(8), 48000, 12, 0, 0, new Float64Array(7), new Float64Array(7), new Float64Array(7)));
attempt('spectrumBatch short out', () => native.spectrumBatch(new Float64Array(16), 2, 8, 48000, -1, 0, 0, new Float64Array(5), new Float64Array(5), new Float64Array(10), new Float64Array(8)));
attempt('transform wrong types', () => native.transform({}, [1, 2, 3], null, new Float64Array(8), new Float64Array(8), false));
attempt('wrong argument count', () => native.windowMake(1));
console.log(JSON.stringify(log));

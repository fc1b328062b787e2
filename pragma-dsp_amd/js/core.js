'use strict';
// Drop-in for `pragma-dsp/core` (reference src/core/fft.ts): same exports, same
// signatures, same error texts; the numeric work runs on the GPU through the addon.
// Node-12 compatible (no `??`, no optional chaining).
const native = require('./native');

// fft.ts:6-14 -- `fill` goes to both planes
function createComplexArray(size, fill) {
  const real = new Float64Array(size);
  const imag = new Float64Array(size);
  if (fill !== undefined && fill !== 0) {
    real.fill(fill);
    imag.fill(fill);
  }
  return { real: real, imag: imag };
}

// fft.ts:16 -- plus an integer check (the reference's int32 coercion lets 2.5 through)
function isPowerOfTwo(n) {
  return typeof n === 'number' && Number.isInteger(n) && n > 0 && n <= 0x40000000 && (n & (n - 1)) === 0;
}

// fft.ts:18-23 (without the int32 overflow above 2^30)
function nextPowerOfTwo(n) {
  if (!(n > 1)) return 1;
  return native.nextPow2(Math.ceil(n));
}

// ArrayLike<number> -> Float64Array; holes / undefined / null read as 0 (`input[i] ?? 0`)
function toF64(input) {
  if (input instanceof Float64Array) return input;
  if (ArrayBuffer.isView(input)) return Float64Array.from(input);
  const n = input.length >>> 0;
  const out = new Float64Array(n);
  for (let i = 0; i < n; i += 1) {
    const v = input[i];
    out[i] = v === undefined || v === null ? 0 : v;
  }
  return out;
}

class Radix2Fft {
  constructor(size) {
    if (!isPowerOfTwo(size)) {
      throw new Error('FFT size must be power of two, got ' + size);
    }
    this._plan = native.planCreate(size); // freed by the addon's finalizer
    Object.defineProperty(this, 'size', { value: size, enumerable: true, writable: false });
  }

  forward(input, out) {
    return this._transform(input, null, out, false);
  }

  forwardComplex(input, out) {
    return this._transform(input.real, input.imag, out, false);
  }

  inverse(input, out) {
    return this._transform(input.real, input.imag, out, true);
  }

  // Extensions: the same three transforms on MANY rows in one call -- the loop of the reference's batch idiom
  // (bench/reallife/signals.ts:264-270: `for (...) fft.forward(input)`) as one device batch.  Element i of the
  // result equals forward(inputs[i]) / forwardComplex(inputs[i]) / inverse(inputs[i]); the ComplexArrays of a
  // batch are views into one buffer per plane.  Length checks and messages are those of fft.ts:95-104, per row.
  forwardBatch(inputs) {
    return this._transformBatch(inputs, false, false);
  }

  forwardComplexBatch(inputs) {
    return this._transformBatch(inputs, true, false);
  }

  inverseBatch(inputs) {
    return this._transformBatch(inputs, true, true);
  }

  _transformBatch(inputs, complex, inverse) {
    const batch = inputs.length >>> 0;
    const n = this.size;
    let typed = Array.isArray(inputs);  // every plane a Float64Array: the rows are read where they lie
    for (let b = 0; b < batch; b += 1) {
      const r = complex ? inputs[b].real : inputs[b];
      const i = complex ? inputs[b].imag : null;
      if (r.length !== n) throw new Error('FFT input length ' + r.length + ' != size ' + n);
      if (i && i.length !== n) throw new Error('FFT input length ' + i.length + ' != size ' + n);
      typed = typed && r instanceof Float64Array && (!complex || i instanceof Float64Array);
    }
    const ore = new Float64Array(batch * n);
    const oim = new Float64Array(batch * n);
    if (batch > 0 && typed) {
      native.transformRows(this._plan, inputs, complex, ore, oim, inverse);
    } else if (batch > 0) {  // plain arrays, other typed arrays, holes (`?? 0`), a missing imag plane: flattened to f64
      const re = new Float64Array(batch * n);
      const im = complex ? new Float64Array(batch * n) : null;
      for (let b = 0; b < batch; b += 1) {
        re.set(toF64(complex ? inputs[b].real : inputs[b]), b * n);
        if (im && inputs[b].imag) im.set(toF64(inputs[b].imag), b * n);
      }
      native.transformBatch(this._plan, batch, re, im, ore, oim, inverse);
    }
    const out = new Array(batch);
    for (let b = 0; b < batch; b += 1) {
      out[b] = { real: ore.subarray(b * n, (b + 1) * n), imag: oim.subarray(b * n, (b + 1) * n) };
    }
    return out;
  }

  _transform(inputReal, inputImag, out, inverse) {
    if (inputReal.length !== this.size) {
      throw new Error('FFT input length ' + inputReal.length + ' != size ' + this.size);
    }
    if (inputImag && inputImag.length !== this.size) {
      throw new Error('FFT input length ' + inputImag.length + ' != size ' + this.size);
    }
    const result = out === undefined || out === null ? createComplexArray(this.size) : out;
    const direct = result.real instanceof Float64Array && result.imag instanceof Float64Array &&
      result.real.length === this.size && result.imag.length === this.size;
    const ore = direct ? result.real : new Float64Array(this.size);
    const oim = direct ? result.imag : new Float64Array(this.size);
    native.transform(this._plan, toF64(inputReal), inputImag ? toF64(inputImag) : null, ore, oim, inverse);
    if (!direct) {
      for (let i = 0; i < this.size; i += 1) {
        result.real[i] = ore[i];
        result.imag[i] = oim[i];
      }
    }
    return result; // the same object when `out` was supplied
  }
}

module.exports = { createComplexArray, isPowerOfTwo, nextPowerOfTwo, Radix2Fft, _toF64: toF64 };

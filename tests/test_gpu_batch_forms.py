"""The batch forms of the drop-in surface (extensions, same names in the JS host and in this mirror): spectrumBatch --
the map of the reference's spectrumStream (src/effect/index.ts:190-194) -- and FFT.forwardBatch / forwardComplexBatch /
inverseBatch -- the loop of bench/reallife/signals.ts:264-270 as one device batch.  Element i must equal the one-call
form on element i exactly, and the oracle at the reference's own tolerance (f64 mode, 1e-10 absolute on unit-scale
signals, signals.test.ts:22-23)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_spectrum_batch_equals_spectrum_per_frame_and_oracle(oracle_mod):
    import pragma_dsp_amd as pd
    rng = np.random.default_rng(3)
    t = np.arange(1024)
    frames = [np.sin(2 * np.pi * (5 + i) * t / 1024) + 0.05 * rng.standard_normal(1024) for i in range(7)]
    frames += [rng.standard_normal(300) for _ in range(3)] + [[0, 1, 0, -1, 0, 1, 0, -1]]   # runs of other lengths
    frames += [rng.standard_normal(1024).astype(np.float32) for _ in range(4)]               # an f32 (audio) run
    for opts in ({"sampleRate": 48000, "window": "hann"}, {"sampleRate": 8000, "fftSize": 2048, "sides": "two", "window": "blackman"}):
        got = pd.spectrumBatch(frames, opts)
        assert len(got) == len(frames)
        for f, g in zip(frames, got):
            one = pd.spectrum(f, opts)
            assert np.array_equal(g.frequencies, one.frequencies) and np.array_equal(g.amplitude, one.amplitude)
            assert np.array_equal(g.phase, one.phase) and g.peak == one.peak
            w = oracle_mod.spectrum(np.asarray(f, dtype=np.float64), sample_rate=opts["sampleRate"], fft_size=opts.get("fftSize"),
                                    window=opts["window"], sides=opts.get("sides", "one"))
            assert np.abs(g.amplitude - w["amplitude"]).max() <= 1e-10
            if opts.get("sides", "one") == "one":
                assert g.peak.index == w["peak"]["index"]
    assert pd.spectrumBatch([], {"sampleRate": 48000}) == []
    with pytest.raises(pd.PdspError, match="FFT size must be power of two, got 12"):
        pd.spectrumBatch(frames[:2], {"fftSize": 12})
    with pytest.raises(pd.PdspError, match="Sample rate must be positive, got 0"):
        pd.spectrumBatch(frames[:2], {"sampleRate": 0})


@pytest.mark.parametrize("n,count", [(8, 3), (1024, 9), (4096, 300)])   # 300 x 4096: the chunked host path
def test_transform_batches_equal_the_one_row_calls_and_oracle(oracle_mod, n, count):
    import pragma_dsp_amd as pd
    rng = np.random.default_rng(n)
    rows = [rng.standard_normal(n) for _ in range(count)]
    fft = pd.FFT(n)
    fwd = fft.forwardBatch(rows)
    cplx = [pd.ComplexArray(rows[i], rows[(i + 1) % count]) for i in range(count)]
    fc = fft.forwardComplexBatch(cplx)
    back = fft.inverseBatch(fc)
    assert len(fwd) == len(fc) == len(back) == count
    oplan = oracle_mod.Plan(n)
    for i in sorted({0, 1, count // 2, count - 1}):
        one = fft.forward(rows[i])
        assert np.array_equal(fwd[i].real, one.real) and np.array_equal(fwd[i].imag, one.imag)
        onec = fft.forwardComplex(cplx[i])
        assert np.array_equal(fc[i].real, onec.real) and np.array_equal(fc[i].imag, onec.imag)
        oneb = fft.inverse(fc[i])
        assert np.array_equal(back[i].real, oneb.real) and np.array_equal(back[i].imag, oneb.imag)
        wre, wim = oplan.forward_complex(cplx[i].real, cplx[i].imag)
        scale = max(np.abs(wre).max(), np.abs(wim).max())
        assert max(np.abs(fc[i].real - wre).max(), np.abs(fc[i].imag - wim).max()) <= 1e-12 * scale
        assert max(np.abs(back[i].real - cplx[i].real).max(), np.abs(back[i].imag - cplx[i].imag).max()) <= 1e-12
    assert fft.forwardBatch([]) == []
    with pytest.raises(pd.PdspError, match=f"FFT input length 3 != size {n}"):
        fft.forwardBatch([rows[0], [1, 2, 3]])

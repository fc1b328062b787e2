"""Streaming front-end vs the one-shot drop-in: spectrumStream(frames, opts) must give, per
frame and in order, exactly what spectrum(frame, opts) gives (the reference asserts its
Effect path bit-identical to spectrum(), test/reallife/effect.test.ts:34-45, :111-146)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def same(a, b):
    return (np.array_equal(a.frequencies, b.frequencies) and np.array_equal(a.amplitude, b.amplitude)
            and np.array_equal(a.phase, b.phase) and a.peak == b.peak)


@pytest.mark.parametrize("opts", [{"sampleRate": 48000, "fftSize": 1024, "window": "hann"},
                                  {"sampleRate": 8000.0, "sides": "two", "window": "blackman"},
                                  {}])
@pytest.mark.parametrize("batch_frames", [1, 2, 4, 64])
def test_stream_equals_per_frame_spectrum(pdsp, opts, batch_frames):
    from pragma_dsp_amd.stream import spectrumStream
    rng = np.random.default_rng(42)
    frames = [rng.standard_normal(1000) for _ in range(7)]
    got = list(spectrumStream(frames, opts, batch_frames=batch_frames))
    assert len(got) == 7
    for f, g in zip(frames, got):
        assert same(g, pdsp.spectrum(f, opts))


def test_stream_edge_cases(pdsp):
    from pragma_dsp_amd.stream import SpectrumStream, spectrumStream
    assert list(spectrumStream([], {"sampleRate": 48000})) == []          # empty stream
    rng = np.random.default_rng(1)
    frames = [rng.standard_normal(n) for n in (100, 128, 300, 300, 64, 2000, 5)]  # sizes change mid-stream
    got = list(spectrumStream(frames, {"window": "hamming"}, batch_frames=3))
    assert [len(g.amplitude) for g in got] == [65, 65, 257, 257, 33, 1025, 5]
    for f, g in zip(frames, got):
        assert same(g, pdsp.spectrum(f, {"window": "hamming"}))
    st = SpectrumStream({"fftSize": 256, "sampleRate": 100}, batch_frames=2)
    out = []
    for i in range(5):
        out += st.push(np.sin(0.3 * i * np.arange(256)))
    assert len(out) in (2, 4)            # results trail the input by at most the in-flight batches
    out += st.flush()
    assert len(out) == 5 and st.flush() == []
    with pytest.raises(pdsp.PdspError, match="FFT size must be power of two, got 12"):
        SpectrumStream({"fftSize": 12})
    with pytest.raises(pdsp.PdspError, match="Sample rate must be positive, got -1"):
        SpectrumStream({"sampleRate": -1})
    with pytest.raises(pdsp.PdspError, match="Unsupported window type: kaiser"):
        SpectrumStream({"window": "kaiser"}).push([1, 2, 3])

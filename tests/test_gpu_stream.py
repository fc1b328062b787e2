"""Streaming front-end: spectrumStream(frames, opts) must give, per frame and in
order, what spectrum(frame, opts) gives -- checked (1) against the CPU ORACLE's spectrum() directly
(src/effect/index.ts:143-194 is a verbatim copy of spectrum.ts:36-142, so the oracle of one is the
oracle of the other), amplitude / phase at the f64 tolerance and the one-sided peak index exactly, and
(2) bit for bit against the one-shot drop-in (the reference asserts its Effect path bit-identical to
spectrum(), test/reallife/effect.test.ts:34-45, :111-146)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def wrap(d):
    return np.abs((np.asarray(d) + np.pi) % (2 * np.pi) - np.pi)


def assert_matches_oracle(oracle_mod, frame, opts, got, bits=64):
    """One streamed result against oracle.spectrum(frame, opts): f64 mode at 1e-12 of the row's maximum
    (phase where the bin is above the noise floor, as signals.test.ts:41-47 masks it), f32 mode at the
    stated 1e-5; one-sided peak index exact (two-sided: the oracle's bin or its mirror)."""
    w = oracle_mod.spectrum(frame, sample_rate=opts.get("sampleRate", 1), fft_size=opts.get("fftSize"),
                            window=opts.get("window", "rect"), sides=opts.get("sides", "one"))
    tol = 1e-12 if bits == 64 else 1e-5
    top = max(w["amplitude"].max(), 1e-300)
    assert np.array_equal(got.frequencies, w["frequencies"])
    assert np.abs(got.amplitude - w["amplitude"]).max() <= tol * top
    mask = w["amplitude"] > (1e-4 if bits == 64 else 1e-3) * top
    assert wrap(got.phase[mask] - w["phase"][mask]).max(initial=0) <= (1e-9 if bits == 64 else 1e-2)
    n = len(w["amplitude"])
    if opts.get("sides", "one") == "one":
        assert got.peak.index == w["peak"]["index"]
    else:
        assert got.peak.index in (w["peak"]["index"], (n - w["peak"]["index"]) % n)
    assert abs(got.peak.amplitude - w["peak"]["amplitude"]) <= tol * top
    assert got.peak.frequency == got.frequencies[got.peak.index]


def same(a, b):
    return (np.array_equal(a.frequencies, b.frequencies) and np.array_equal(a.amplitude, b.amplitude)
            and np.array_equal(a.phase, b.phase) and a.peak == b.peak)


@pytest.mark.parametrize("opts", [{"sampleRate": 48000, "fftSize": 1024, "window": "hann"},
                                  {"sampleRate": 8000.0, "sides": "two", "window": "blackman"},
                                  {}])
@pytest.mark.parametrize("batch_frames", [1, 2, 4, 64])
def test_stream_equals_per_frame_spectrum(pdsp, oracle_mod, opts, batch_frames):
    from pragma_dsp_amd.stream import spectrumStream
    rng = np.random.default_rng(42)
    t = np.arange(1000)
    frames = [rng.standard_normal(1000) + 3.0 * np.sin(2 * np.pi * (20 + 7 * i) * t / 1024) for i in range(7)]
    got = list(spectrumStream(frames, opts, batch_frames=batch_frames))
    assert len(got) == 7
    for f, g in zip(frames, got):
        assert_matches_oracle(oracle_mod, f, opts, g)   # the oracle, directly
        assert same(g, pdsp.spectrum(f, opts))           # and bit-identical to the one-shot drop-in


def test_stream_f32_mode_vs_oracle(pdsp, oracle_mod):
    from pragma_dsp_amd.stream import spectrumStream
    rng = np.random.default_rng(43)
    t = np.arange(4096)
    frames = [rng.standard_normal(4096) + 5.0 * np.sin(2 * np.pi * (100 + 31 * i) * t / 4096) for i in range(9)]
    opts = {"sampleRate": 44100, "window": "hamming"}
    prev = pdsp.lib.pdsp_set_host_precision(32)
    try:
        got = list(spectrumStream(frames, opts, batch_frames=4))
        assert len(got) == 9
        for f, g in zip(frames, got):
            assert_matches_oracle(oracle_mod, f, opts, g, bits=32)
            assert same(g, pdsp.spectrum(f, opts))
    finally:
        pdsp.lib.pdsp_set_host_precision(prev)


def test_stream_interleaved_sizes_keep_every_result(pdsp, oracle_mod):
    """Three frame lengths interleaved with batches far larger than the runs, so that a size is
    revisited while both of its staging slots are still in flight behind batches of OTHER sizes
    (_free_slot dropped the results harvested inside its recursion).  One result per
    input frame, in input order (src/effect/index.ts:190-194)."""
    from pragma_dsp_amd.stream import SpectrumStream, spectrumStream
    rng = np.random.default_rng(7)
    lens = {"A": 256, "B": 64, "C": 1024}
    order = "BACACACABCBCAACCBBA" * 3
    frames = [rng.standard_normal(lens[c]) + np.sin(0.3 * np.arange(lens[c])) for c in order]
    opts = {"sampleRate": 1000.0, "window": "hann"}
    for bf in (1, 2, 64):
        st = SpectrumStream(opts, batch_frames=bf)
        got = []
        for f in frames:
            got += st.push(f)
        got += st.flush()
        assert len(got) == len(frames), (bf, len(got))
        for f, g in zip(frames, got):
            assert len(g.amplitude) == len(f) // 2 + 1
            assert_matches_oracle(oracle_mod, f, opts, g)
            assert same(g, pdsp.spectrum(f, opts))
    assert len(list(spectrumStream(frames, opts, batch_frames=1024))) == len(frames)


def test_stream_slot_budget_and_dtype_rule(pdsp):
    """A slot never pins more than slotBytes of input staging (at least one frame), and the per-size
    precision follows spectrum()'s rule (f64 where the f64 tables reach, f32 beyond)."""
    from pragma_dsp_amd.stream import SpectrumStream
    import torch
    st = SpectrumStream({"window": "rect", "slotBytes": 1 << 20}, batch_frames=1024)
    assert st._frames_per_slot(1024) == 128 and st._frames_per_slot(1 << 22) == 1
    big = int(pdsp.lib.pdsp_max_size(8))
    assert st._dtype_for(big) == torch.float64 and st._dtype_for(2 * big) == torch.float32
    rng = np.random.default_rng(3)
    frames = [rng.standard_normal(1024) for _ in range(300)]  # > 2 slots of 128 frames
    got = []
    for f in frames:
        got += st.push(f)
    got += st.flush()
    assert len(got) == 300
    for i in (0, 127, 128, 299):
        assert same(got[i], pdsp.spectrum(frames[i], {"window": "rect"}))


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

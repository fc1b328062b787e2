"""Batched host-f64 calls large enough to be cut into chunks on several workers (pdsp_capi.hip, run_chunked):
pdsp_spectrum_batch_host_f64 -- what the JS drop-in's spectrumBatch() binds, the map of the reference's
spectrumStream (src/effect/index.ts:190-194) -- and pdsp_fft_transform_host_f64 on many rows
(Radix2Fft.transform, src/core/fft.ts:89-151).  The chunked result must equal the one-shot sequence
(PDSP_HOST_THREADS=1) bit for bit, and both must meet the oracle at the precision's tolerance."""
import ctypes as C
import os
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture()
def capi():
    import pragma_dsp_amd  # noqa: F401
    from pragma_dsp_amd import _capi
    prev = _capi.lib.pdsp_set_host_precision(0)
    env = os.environ.get("PDSP_HOST_THREADS")
    yield _capi
    _capi.lib.pdsp_set_host_precision(prev)
    if env is None:
        os.environ.pop("PDSP_HOST_THREADS", None)
    else:
        os.environ["PDSP_HOST_THREADS"] = env


def _spectrum_batch(capi, x, n, window, sides, threads=None):
    if threads is not None:  # (never from concurrent threads: setenv beside another thread's getenv is a race)
        os.environ["PDSP_HOST_THREADS"] = str(threads)
    lib = capi.lib
    batch, length = x.shape
    bins = n // 2 + 1 if sides == 0 else n
    freq = np.full(bins, np.nan)
    amp = np.full((batch, bins), np.nan)
    ph = np.full((batch, bins), np.nan)
    peaks = (capi.Peak * batch)()
    nb = C.c_longlong(0)
    capi.check(lib.pdsp_spectrum_batch_host_f64(capi.dptr(x), batch, length, 48000.0, n, window, sides, capi.dptr(freq),
                                                capi.dptr(amp), capi.dptr(ph), peaks, C.byref(nb)))
    assert nb.value == bins
    pk = np.array([(p.index, p.frequency, p.amplitude, p.phase) for p in peaks])
    return freq, amp, ph, pk


@pytest.mark.parametrize("precision,tol", [(64, 1e-12), (32, 1e-5)])
@pytest.mark.parametrize("n,length,batch,window,sides", [
    (1024, 1024, 1501, 1, 0),    # hann, one-sided; the last chunk is partial
    (4096, 3000, 700, 3, 0),     # blackman over zero-padded frames (buildFrame, spectrum.ts:36-43)
    (2048, 2500, 1100, 0, 1),    # rect, two-sided, frames longer than N are truncated
    (16384, 16384, 90, 2, 0),    # hamming at configs[3]'s frame size
])
def test_chunked_spectrum_batch_equals_one_shot_and_oracle(capi, oracle_mod, precision, tol, n, length, batch, window, sides):
    capi.lib.pdsp_set_host_precision(precision)
    rng = np.random.default_rng(1337 + n)
    t = np.arange(length)
    x = rng.standard_normal((batch, length)) + 2.0 * np.sin(2 * np.pi * rng.integers(3, n // 4, (batch, 1)) * t / n)
    one = _spectrum_batch(capi, x, n, window, sides, 1)
    for threads in (2, 5):
        got = _spectrum_batch(capi, x, n, window, sides, threads)
        for a, b, what in zip(one, got, ("frequencies", "amplitude", "phase", "peaks")):
            assert np.array_equal(a, b), f"{what}: chunked on {threads} workers differs from the one-shot sequence"
    # the oracle's spectrum() on rows drawn over the whole batch (first, last, chunk boundaries)
    kinds = ["rect", "hann", "hamming", "blackman"]
    rows = sorted({0, 1, batch // 2, batch - 2, batch - 1} | set(int(r) for r in rng.integers(0, batch, 6)))
    for r in rows:
        want = oracle_mod.spectrum(x[r], sample_rate=48000.0, fft_size=n, window=kinds[window],
                                   sides="one" if sides == 0 else "two")
        scale = np.abs(want["amplitude"]).max()
        assert np.abs(one[1][r] - want["amplitude"]).max() <= tol * scale
        assert np.array_equal(one[0], want["frequencies"])
        if sides == 0:  # the two-sided peak may be the oracle's mirror bin
            assert int(one[3][r][0]) == want["peak"]["index"]
            assert abs(one[3][r][2] - want["peak"]["amplitude"]) <= tol * scale


def _transform(capi, plan, re, im, inverse, threads, out=None):
    os.environ["PDSP_HOST_THREADS"] = str(threads)
    batch, n = re.shape
    ore, oim = out if out is not None else (np.full_like(re, np.nan), np.full_like(re, np.nan))
    capi.check(capi.lib.pdsp_fft_transform_host_f64(plan, batch, n, capi.dptr(re), capi.dptr(im) if im is not None else None,
                                                    capi.dptr(ore), capi.dptr(oim), inverse))
    return ore, oim


@pytest.mark.parametrize("precision,tol", [(64, 1e-12), (32, 1e-5)])
@pytest.mark.parametrize("n,batch", [(1024, 2100), (4096, 530), (8192, 300)])
def test_chunked_transform_equals_one_shot_and_oracle(capi, oracle_mod, precision, tol, n, batch):
    capi.lib.pdsp_set_host_precision(precision)
    rng = np.random.default_rng(7 + n)
    re = rng.standard_normal((batch, n))
    im = rng.standard_normal((batch, n))
    plan = C.c_void_p()
    capi.check(capi.lib.pdsp_plan_create(n, -1, C.byref(plan)))
    try:
        oplan = oracle_mod.Plan(n)
        rows = [0, 1, batch // 3, batch - 1]
        for imag, inverse in ((im, 0), (None, 0), (im, 1)):
            one = _transform(capi, plan, re, imag, inverse, 1)
            many = _transform(capi, plan, re, imag, inverse, 4)
            assert np.array_equal(one[0], many[0]) and np.array_equal(one[1], many[1])
            if inverse:
                wre, wim = oplan.inverse(re[rows], im[rows])
            elif imag is None:
                wre, wim = oplan.forward(re[rows])
            else:
                wre, wim = oplan.forward_complex(re[rows], im[rows])
            want = wre + 1j * wim
            got = many[0][rows] + 1j * many[1][rows]
            assert (np.abs(got - want).max(axis=1) / np.abs(want).max(axis=1)).max() <= tol
        # output planes that ARE the input planes: the call keeps the one-shot sequence (every input is read
        # before any output is written) and still returns the transform
        want = _transform(capi, plan, re, im, 0, 1)
        re2, im2 = re.copy(), im.copy()
        _transform(capi, plan, re2, im2, 0, 4, out=(re2, im2))
        assert np.array_equal(re2, want[0]) and np.array_equal(im2, want[1])
    finally:
        capi.lib.pdsp_plan_destroy(plan)


def test_chunked_calls_from_two_threads(capi, oracle_mod):
    """Two host threads, each in a chunked spectrumBatch of its own size (own cached plans), at the same time."""
    capi.lib.pdsp_set_host_precision(64)
    rng = np.random.default_rng(99)
    cases = [(1024, rng.standard_normal((1300, 1024))), (2048, rng.standard_normal((700, 2048)))]
    want = [_spectrum_batch(capi, x, n, 1, 0, 1) for n, x in cases]
    os.environ["PDSP_HOST_THREADS"] = "3"
    got = [None, None]
    errs = []

    def run(i):
        try:
            for _ in range(3):
                got[i] = _spectrum_batch(capi, cases[i][1], cases[i][0], 1, 0)
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    for w, g in zip(want, got):
        for a, b in zip(w, g):
            assert np.array_equal(a, b)


@pytest.mark.parametrize("precision", [64, 32])
@pytest.mark.parametrize("n,length,batch", [(1024, 1024, 1501), (4096, 3000, 40), (8, 8, 3)])
def test_rows_variant_equals_contiguous_frames(capi, precision, n, length, batch):
    """pdsp_spectrum_rows_host_f64 (one pointer per frame: an array of Float64Arrays taken where it lies) against
    pdsp_spectrum_batch_host_f64 on the same frames flattened: identical, chunked or not."""
    capi.lib.pdsp_set_host_precision(precision)
    rng = np.random.default_rng(5 + n)
    frames = [rng.standard_normal(length) for _ in range(batch)]     # separately allocated rows
    x = np.stack(frames)
    for threads in (1, 4):
        want = _spectrum_batch(capi, x, n, 1, 0, threads)
        bins = n // 2 + 1
        freq, amp, ph = np.full(bins, np.nan), np.full((batch, bins), np.nan), np.full((batch, bins), np.nan)
        peaks = (capi.Peak * batch)()
        dp = C.POINTER(C.c_double)
        rows = (dp * batch)(*[capi.dptr(f) for f in frames])
        capi.check(capi.lib.pdsp_spectrum_rows_host_f64(rows, batch, length, 48000.0, n, 1, 0, capi.dptr(freq), capi.dptr(amp),
                                                        capi.dptr(ph), peaks, None))
        pk = np.array([(p.index, p.frequency, p.amplitude, p.phase) for p in peaks])
        for a, b in zip(want, (freq, amp, ph, pk)):
            assert np.array_equal(a, b)
    # a null row is refused before any device work
    rows[batch // 2] = None
    rc = capi.lib.pdsp_spectrum_rows_host_f64(rows, batch, length, 48000.0, n, 1, 0, capi.dptr(freq), capi.dptr(amp),
                                              capi.dptr(ph), peaks, None)
    assert rc == capi.ERR_BAD_ARG


@pytest.mark.parametrize("precision", [64, 32])
@pytest.mark.parametrize("n,batch", [(1024, 2100), (8, 3)])
def test_transform_rows_variant_equals_contiguous_planes(capi, precision, n, batch):
    """pdsp_fft_transform_rows_host_f64 (one pointer per input row) against pdsp_fft_transform_host_f64 on the same rows
    as contiguous planes: identical, chunked or not, forward (real and complex) and inverse."""
    capi.lib.pdsp_set_host_precision(precision)
    rng = np.random.default_rng(11 + n)
    re_rows = [rng.standard_normal(n) for _ in range(batch)]
    im_rows = [rng.standard_normal(n) for _ in range(batch)]
    re, im = np.stack(re_rows), np.stack(im_rows)
    dp = C.POINTER(C.c_double)
    rp = (dp * batch)(*[capi.dptr(r) for r in re_rows])
    ip = (dp * batch)(*[capi.dptr(r) for r in im_rows])
    plan = C.c_void_p()
    capi.check(capi.lib.pdsp_plan_create(n, -1, C.byref(plan)))
    try:
        for threads in (1, 4):
            for use_im, inverse in ((True, 0), (False, 0), (True, 1)):
                want = _transform(capi, plan, re, im if use_im else None, inverse, threads)
                ore, oim = np.full_like(re, np.nan), np.full_like(re, np.nan)
                capi.check(capi.lib.pdsp_fft_transform_rows_host_f64(plan, batch, n, rp, ip if use_im else None, capi.dptr(ore),
                                                                     capi.dptr(oim), inverse))
                assert np.array_equal(ore, want[0]) and np.array_equal(oim, want[1])
        assert capi.lib.pdsp_fft_transform_rows_host_f64(plan, batch, n + 1, rp, ip, capi.dptr(ore), capi.dptr(oim), 0) == capi.ERR_INPUT_LENGTH
        assert capi.lib.pdsp_fft_transform_rows_host_f64(plan, batch, n, rp, None, capi.dptr(ore), capi.dptr(oim), 1) == capi.ERR_BAD_ARG
    finally:
        capi.lib.pdsp_plan_destroy(plan)


@pytest.mark.parametrize("precision", [64, 32])
@pytest.mark.parametrize("n,length,batch", [(1024, 1024, 1501), (4096, 3000, 40), (8, 8, 3)])
def test_f32_input_rows_equal_the_widened_f64_frames(capi, precision, n, length, batch):
    """pdsp_spectrum_rows_host_f32in (Float32Array audio frames, one pointer per frame) against
    pdsp_spectrum_batch_host_f64 on the same samples widened to f64 (exact): identical results, chunked or not."""
    capi.lib.pdsp_set_host_precision(precision)
    rng = np.random.default_rng(21 + n)
    frames = [rng.standard_normal(length).astype(np.float32) for _ in range(batch)]
    x = np.stack(frames).astype(np.float64)
    fp = C.POINTER(C.c_float)
    rows = (fp * batch)(*[f.ctypes.data_as(fp) for f in frames])
    bins = n // 2 + 1
    for threads in (1, 4):
        want = _spectrum_batch(capi, x, n, 2, 0, threads)
        freq, amp, ph = np.full(bins, np.nan), np.full((batch, bins), np.nan), np.full((batch, bins), np.nan)
        peaks = (capi.Peak * batch)()
        capi.check(capi.lib.pdsp_spectrum_rows_host_f32in(rows, batch, length, 48000.0, n, 2, 0, capi.dptr(freq), capi.dptr(amp),
                                                          capi.dptr(ph), peaks, None))
        pk = np.array([(p.index, p.frequency, p.amplitude, p.phase) for p in peaks])
        for a, b in zip(want, (freq, amp, ph, pk)):
            assert np.array_equal(a, b)


def test_two_threads_share_one_plan_on_the_chunked_path(capi):
    """Two host threads in pdsp_fft_transform_host_f64 on the SAME plan, both calls large enough to be chunked: the
    plan's mutex serialises them (its staging slots and streams belong to one call at a time); both get their rows."""
    capi.lib.pdsp_set_host_precision(64)
    n, batch = 2048, 1100
    rng = np.random.default_rng(5)
    data = [(rng.standard_normal((batch, n)), rng.standard_normal((batch, n))) for _ in range(2)]
    plan = C.c_void_p()
    capi.check(capi.lib.pdsp_plan_create(n, -1, C.byref(plan)))
    try:
        want = [_transform(capi, plan, re, im, 0, 1) for re, im in data]
        os.environ["PDSP_HOST_THREADS"] = "3"
        got, errs = [None, None], []

        def run(i):
            try:
                for _ in range(3):
                    ore, oim = np.full_like(data[i][0], np.nan), np.full_like(data[i][0], np.nan)
                    capi.check(capi.lib.pdsp_fft_transform_host_f64(plan, batch, n, capi.dptr(data[i][0]), capi.dptr(data[i][1]),
                                                                    capi.dptr(ore), capi.dptr(oim), 0))
                    got[i] = (ore, oim)
            except Exception as e:  # noqa: BLE001
                errs.append(e)

        ts = [threading.Thread(target=run, args=(i,)) for i in range(2)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        assert not errs, errs
        for w, g in zip(want, got):
            assert np.array_equal(w[0], g[0]) and np.array_equal(w[1], g[1])
    finally:
        capi.lib.pdsp_plan_destroy(plan)


def test_chunked_vs_one_shot_fuzz(capi):
    """Random sizes, frame lengths, batch sizes, windows, sides, precisions and worker counts: the chunked batched
    spectrum (contiguous, row-pointer and f32-row forms) always equals the one-shot sequence bit for bit."""
    rng = np.random.default_rng(20260105)
    dp, fp = C.POINTER(C.c_double), C.POINTER(C.c_float)
    for case in range(24):
        n = int(2 ** rng.integers(6, 14))
        length = int(rng.choice([n, n, n - rng.integers(1, n // 2), n + rng.integers(1, n)]))
        # enough frames for 1 ... 5 chunks of staging around the 4 MiB threshold
        batch = int(max(2, rng.integers(1, 6) * (3 << 20) // (16 * n)) + rng.integers(0, 7))
        window, sides, precision = int(rng.integers(0, 4)), int(rng.integers(0, 2)), int(rng.choice([64, 32]))
        threads = int(rng.integers(2, 7))
        capi.lib.pdsp_set_host_precision(precision)
        x32 = rng.standard_normal((batch, length)).astype(np.float32)
        x = x32.astype(np.float64)
        want = _spectrum_batch(capi, x, n, window, sides, 1)
        got = _spectrum_batch(capi, x, n, window, sides, threads)
        tag = f"case {case}: n={n} len={length} batch={batch} win={window} sides={sides} f{precision} threads={threads}"
        for a, b in zip(want, got):
            assert np.array_equal(a, b), tag
        bins = n // 2 + 1 if sides == 0 else n
        for rows, fn in (((dp * batch)(*[capi.dptr(r) for r in x]), capi.lib.pdsp_spectrum_rows_host_f64),
                         ((fp * batch)(*[r.ctypes.data_as(fp) for r in x32]), capi.lib.pdsp_spectrum_rows_host_f32in)):
            freq, amp, ph = np.full(bins, np.nan), np.full((batch, bins), np.nan), np.full((batch, bins), np.nan)
            peaks = (capi.Peak * batch)()
            capi.check(fn(rows, batch, length, 48000.0, n, window, sides, capi.dptr(freq), capi.dptr(amp), capi.dptr(ph), peaks, None))
            pk = np.array([(p.index, p.frequency, p.amplitude, p.phase) for p in peaks])
            for a, b in zip(want, (freq, amp, ph, pk)):
                assert np.array_equal(a, b), tag + " (rows form)"

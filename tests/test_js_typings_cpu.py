"""The TypeScript declarations beside the JS host (pragma-dsp_amd/js/*.d.ts) name every public export of the
module they describe -- the reference's hot-path surface (src/core/fft.ts:63-87,
src/xform/fourier.ts:11-165, src/public/spectrum.ts:15-34, 107-142) -- and nothing the module lacks.  There is no
tsc in the image, so this is a name-level check, not a type check."""
import os
import re

JS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pragma-dsp_amd", "js")


def _runtime_exports(name):
    src = open(os.path.join(JS, name + ".js")).read()
    body = re.search(r"module\.exports\s*=\s*\{(.*?)\};", src, re.S).group(1)
    body = re.sub(r"//[^\n]*", "", body)
    names = [re.split(r"[:\s]", part.strip())[0] for part in re.sub(r"\{[^}]*\}", "", body).split(",") if part.strip()]
    return {n for n in names if not n.startswith("_")}


def _declared(name):
    src = open(os.path.join(JS, name + ".d.ts")).read()
    return set(re.findall(r"^export (?:declare )?(?:function|class|const) (\w+)", src, re.M))


def test_declarations_match_runtime_exports():
    for mod in ("core", "fourier", "spectrum"):
        assert _declared(mod) == _runtime_exports(mod), mod
    idx = open(os.path.join(JS, "index.d.ts")).read()
    for name in ("spectrum", "spectrumBatch", "spectrumStream", "core", "fourier", "Radix2Fft", "FFT", "createWindow", "applyWindow",
                 "magnitude", "phase", "fftShift", "fftShiftComplex", "binFrequencies", "createComplexArray",
                 "isPowerOfTwo", "nextPowerOfTwo"):
        assert re.search(r"\b%s\b" % name, idx), name


def test_reference_type_names_are_declared():
    # the type names a TypeScript caller of the reference imports (spectrum.ts:15-34, fft.ts:1-4, fourier.ts:11)
    text = "".join(open(os.path.join(JS, f)).read() for f in ("core.d.ts", "fourier.d.ts", "spectrum.d.ts"))
    for t in ("ComplexArray", "WindowType", "SpectrumOptions", "SpectrumPeak", "SpectrumResult"):
        assert re.search(r"export type %s\b" % t, text), t

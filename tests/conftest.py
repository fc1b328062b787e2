import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    """The CPU oracle (test infrastructure).  Built on demand with gcc."""
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def manifest():
    import json
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def reallife():
    import numpy as np
    return np.load(os.path.join(GOLDEN, "reallife_n1024.npz"))


@pytest.fixture(scope="session")
def windows_dsp():
    import numpy as np
    return np.load(os.path.join(GOLDEN, "windows_dsp.npz"))


@pytest.fixture(scope="session")
def v01():
    import numpy as np
    return np.load(os.path.join(GOLDEN, "v01_fixture.npz"))


@pytest.fixture(scope="session")
def pdsp():
    """The product package (ctypes over libpdsp_hip.so).  Never falls back."""
    import pragma_dsp_amd
    return pragma_dsp_amd


def rel_err(got, want):
    """The stated fp32 tolerance metric: max|got-want| / max|want| per transform
    (never element-wise relative)."""
    import numpy as np
    got = np.asarray(got)
    want = np.asarray(want)
    scale = np.abs(want).max(axis=-1, keepdims=True)
    scale = np.where(scale == 0, 1.0, scale)
    return float((np.abs(got - want) / scale).max())

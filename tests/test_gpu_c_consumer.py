"""The C ABI from plain C (tests/c/abi_smoke.c, built with gcc -std=c11 -pedantic -Werror against include/pdsp_hip.h): what a
cgo / JNI / N-API binding sees.  BASELINE configs[0] -- the reference's README signal through spectrum() -- plus a
forward / inverse round trip and an argument error with the reference's text."""
import json
import subprocess

import numpy as np
import pytest

from test_capi_cpu import _build_c_consumer

pytestmark = pytest.mark.gpu


def test_plain_c_consumer_runs_configs0_through_the_abi(tmp_path):
    exe = _build_c_consumer(tmp_path)
    p = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, (p.returncode, p.stdout, p.stderr)
    d = json.loads(p.stdout)
    assert d["bins"] == 5 and d["peak"]["index"] == 2 and d["peak"]["frequency"] == 12000.0
    assert abs(d["peak"]["amplitude"] - 1) < 1e-12 and abs(d["peak"]["phase"] + np.pi / 2) < 1e-12
    assert np.abs(np.array(d["amplitude"]) - [0, 0, 1, 0, 0]).max() < 1e-12       # SURVEY 8(a) known answer, README.md:11
    assert abs(d["X2"][0]) < 1e-12 and abs(d["X2"][1] + 4) < 1e-12 and d["round_trip_err"] < 1e-12
    assert d["size12_status"] == 1 and d["size12_message"] == "FFT size must be power of two, got 12"
    assert d["next_pow2_1000"] == 1024 and d["version"] >= 100

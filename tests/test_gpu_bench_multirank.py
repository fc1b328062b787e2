"""bench.py's N > 1 path and its RCCL leg, as far as ONE card can run them: the exchange leg in a
world of one rank over a real RCCL group (`--rccl-selftest`), and two ranks sharing the card with real compute and
the gloo exchange (`--gpus 2 --share-gpu`), in the self-launching form.  Small batches: these are plumbing checks of
statuses, fields and teardown, not measurements."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
QUICK = ["--steps", "3", "--warmup", "1", "--batch", "4096", "--ramp-seconds", "0", "--no-measure-traffic", "--no-also"]


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def _line(p):
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (p.stdout[-2000:], p.stderr[-2000:])
    return json.loads(lines[0])


def test_rccl_selftest_one_rank_group_on_this_card():
    p = subprocess.run([sys.executable, BENCH, "--rccl-selftest", "--no-cpu-baseline"] + QUICK, capture_output=True,
                       text=True, timeout=600, env=_env(), cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    d = _line(p)
    g = d["gather"]
    assert "error" not in g, g
    assert g["backend"] == "nccl" and g["ranks"] == 1 and g["rccl_version"]
    for leg in ("peaks_16B_per_frame", "slabs"):
        assert g[leg]["own_rows_intact"] is True and g[leg]["rows_gathered"] == 4096
    assert g["slabs"]["bytes_per_rank"] == 2 * 4096 * 4096 * 4 and g["peaks_16B_per_frame"]["bytes_per_rank"] == 16 * 4096
    assert g["group_destroyed"] is True
    assert "destroy_process_group() was not called" not in p.stderr
    assert d["n_gpus"] == 1 and d["value"] > 0


def test_two_ranks_sharing_the_card_gloo_exchange_and_parity():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--share-gpu"] + QUICK, capture_output=True, text=True,
                       timeout=900, env=_env(), cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    d = _line(p)
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8192 and len(d["per_rank_kernel_ms"]) == 2
    assert d["parity"]["ok"] is True and d["parity"]["max_rel_err"] <= 1e-5   # 256 rows of rank 0's timed output vs the oracle
    g = d["gather"]
    assert "error" not in g and g["backend"] == "gloo" and g["ranks"] == 2
    assert g["peaks_16B_per_frame"]["rows_gathered"] == 8192 and g["slabs"]["rows_gathered"] == 8192
    assert "rehearsal" in d and "destroy_process_group() was not called" not in p.stderr

"""bench.py's N>1 plumbing on CPU (no GPU here): `--gpus N` without WORLD_SIZE makes bench.py the
launcher of N child ranks; with WORLD_SIZE set (torch.distributed.run, the
driver's form) it is a rank.  `--dry-run` runs everything but the compute -- rendezvous over gloo,
the contiguous row split of pragma-dsp_amd/shard.py, the gather of 16-byte peak records and the
max-over-ranks reduction -- and says so in its line (value null): it is never a measurement."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    return env


def _run(argv, timeout=240, env=None):
    return subprocess.run([sys.executable, BENCH] + argv, capture_output=True, text=True, timeout=timeout,
                          env=env or _env(), cwd=ROOT)


@pytest.mark.parametrize("gpus,batch", [(2, 65536), (3, 7)])
def test_gpus_flag_launches_child_ranks(gpus, batch):
    p = _run(["--gpus", str(gpus), "--dry-run", "--batch", str(batch)])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout  # ONE JSON line, rank 0's, relayed by the parent
    d = json.loads(lines[0])
    assert d["n_gpus"] == gpus and d["ranks_seen"] == gpus and d["dry_run"] is True and d["value"] is None
    assert d["global_batch"] == gpus * batch and d["rows"] == [0, batch] and d["gather_ok"] is True
    assert d["backend"] == "gloo"


def test_child_failure_makes_the_parent_exit_nonzero():
    p = _run(["--gpus", "2", "--dry-run", "--fail-rank", "1"])
    assert p.returncode != 0
    assert "rank 1 exited with 3" in p.stderr
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]  # no result line is relayed


def test_rank0_failure_is_reported_too():
    p = _run(["--gpus", "2", "--dry-run", "--fail-rank", "0"])
    assert p.returncode != 0 and "rank 0 exited with 3" in p.stderr


def test_driver_form_torch_distributed_run():
    """`python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2`: WORLD_SIZE is set, so
    bench.py must run as a rank and NOT spawn another layer of children."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "2", "--dry-run"],
                       capture_output=True, text=True, timeout=300, env=_env(), cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["gather_ok"] is True


def test_single_process_without_gpu_fails_loudly():
    """No CPU fallback: without a GPU the real workload refuses to run (and prints no result line)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    p = _run(["--steps", "1", "--warmup", "0"])
    assert p.returncode == 2 and "needs a GPU" in p.stderr
    p = _run(["--gpus", "2", "--steps", "1"])
    assert p.returncode == 2 and "GPU(s) visible" in p.stderr

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


@pytest.mark.parametrize("gpus,batch", [(2, 65536), (3, 7), (8, 4096)])
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


def test_abandoned_exchange_exits_5_on_every_rank_and_the_line_survives():
    """A gather that never returns (rank 1 never joins it) is abandoned by the watchdog after --gather-timeout:
    rank 0's line is printed with gather.error and relayed, and the status is 5 -- a hang is a finding, not a
    success (VERDICT r2 item 1b / ADVICE r2 medium)."""
    p = _run(["--gpus", "2", "--dry-run", "--batch", "64", "--simulate-hang", "--gather-timeout", "3"])
    assert p.returncode == 5, (p.returncode, p.stderr[-2000:])
    assert "rank 0 exited with 5" in p.stderr and "rank 1 exited with 5" in p.stderr
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and "abandoned" in d["gather"]["error"] and "gather_ok" not in d


def test_abandoned_exchange_in_the_driver_form_is_nonzero_too():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "2", "--dry-run",
                        "--batch", "64", "--simulate-hang", "--gather-timeout", "3"],
                       capture_output=True, text=True, timeout=300, env=_env(), cwd=ROOT)
    assert p.returncode != 0
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and "abandoned" in json.loads(lines[0])["gather"]["error"]


def test_world_of_one_watchdog_status():
    p = _run(["--dry-run", "--batch", "64", "--simulate-hang", "--gather-timeout", "2"])
    assert p.returncode == 5
    assert "abandoned" in json.loads(p.stdout.strip().splitlines()[-1])["gather"]["error"]


def test_rank0_stdout_larger_than_a_pipe_buffer_does_not_stall_the_launch():
    """rank 0's stdout used to be a pipe read only after exit; 64 KiB of stray output blocked the child
    until --launch-timeout.  It is a file now."""
    p = _run(["--gpus", "2", "--dry-run", "--batch", "64", "--stdout-noise", "300000", "--launch-timeout", "120"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["gather_ok"] is True


def test_a_line_printed_before_a_failure_is_still_relayed():
    """The measurement survives a later failure: rank 0's line is relayed even when the launch ends non-zero."""
    import tempfile
    child = "\n".join(["import os, sys",
                       "if os.environ['RANK'] == '0':",
                       "    print('{\"value\": 1}', flush=True)",
                       "sys.exit(5)", ""])
    driver = "\n".join(["import sys",
                        "sys.path.insert(0, sys.argv[1])",
                        "import bench",
                        "args = bench.parse_args(['--gpus', '2', '--launch-timeout', '60', '--dry-run'])",
                        "bench.__file__ = sys.argv[2]  # launch_ranks starts children of `__file__`",
                        "raise SystemExit(bench.launch_ranks(args, []))", ""])
    with tempfile.TemporaryDirectory() as td:
        cpath = os.path.join(td, "_child.py")
        with open(cpath, "w") as f:
            f.write(child)
        p = subprocess.run([sys.executable, "-c", driver, ROOT, cpath], capture_output=True, text=True, timeout=120,
                           env=_env(), cwd=ROOT)
    assert p.returncode == 5, (p.returncode, p.stderr[-1000:])
    assert p.stdout.strip() == '{"value": 1}'


def test_single_process_without_gpu_fails_loudly():
    """No CPU fallback: without a GPU the real workload refuses to run (and prints no result line)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    p = _run(["--steps", "1", "--warmup", "0"])
    assert p.returncode == 2 and "needs a GPU" in p.stderr
    p = _run(["--gpus", "2", "--steps", "1"])
    assert p.returncode == 2 and "GPU(s) visible" in p.stderr


def test_orphan_rccl_groups_are_found_in_the_exception_frames_and_aborted(monkeypatch):
    """A bring-up that fails inside dist.new_group leaves a half-made ProcessGroupNCCL no caller holds; bench.py finds it
    in the frames the exception passed through and aborts it (rehearsed on the GPU box: no "destroy_process_group() was
    not called" warning).  Here with a stand-in class, no GPU."""
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import bench

    class FakeGroup:
        aborted = 0

        def abort(self):
            FakeGroup.aborted += 1

    monkeypatch.setattr(dist, "ProcessGroupNCCL", FakeGroup, raising=False)

    def helper_that_fails():
        backend_class = FakeGroup()  # the local torch's _new_process_group_helper holds when eager connect throws
        other = FakeGroup()  # noqa: F841  (a second one in the same frame)
        assert backend_class is not None
        raise RuntimeError("NCCL error: invalid usage")

    def new_group():
        helper_that_fails()

    try:
        new_group()
    except RuntimeError as exc:
        assert bench.close_orphan_rccl_groups(exc) == 2
    assert FakeGroup.aborted == 2
    try:
        raise ValueError("nothing to close")
    except ValueError as exc:
        assert bench.close_orphan_rccl_groups(exc) == 0

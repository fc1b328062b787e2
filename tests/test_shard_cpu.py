"""The N>1 path on CPU: world_size 2 (and 3) over gloo.  The shard logic and the final
gather are the product's (pragma-dsp_amd/shard.py); the per-row compute is stood in
for by the oracle, since there is no GPU here -- what is under test is the partition,
the ragged-shard gather and the max-over-ranks reduction."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, batch, n, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle
        from pragma_dsp_amd.shard import ShardedBatch, gather_rows, max_over_ranks, my_rows, shard_bounds

        rng = np.random.default_rng(1337)  # same global batch on every rank
        re = rng.standard_normal((batch, n))
        im = rng.standard_normal((batch, n))
        plan = oracle.Plan(n)

        def compute(lre, lim):  # stands in for BatchedFft.forward on this rank's rows
            if lre.shape[0] == 0:
                return torch.empty((0, n), dtype=torch.float64), torch.empty((0, n), dtype=torch.float64)
            a, b = plan.forward_complex(lre.numpy(), lim.numpy())
            return torch.from_numpy(a), torch.from_numpy(b)

        sb = ShardedBatch(batch, compute)
        assert (sb.start, sb.stop) == shard_bounds(batch, world)[rank] == my_rows(batch)
        lre, lim = sb.local(torch.from_numpy(re)), sb.local(torch.from_numpy(im))
        assert lre.shape[0] == sb.rows
        ore, oim = sb.run(lre, lim, gather=True)
        wre, wim = plan.forward_complex(re, im) if batch else (np.empty((0, n)), np.empty((0, n)))
        ok = (ore.shape == (batch, n) and np.array_equal(ore.numpy(), wre) and np.array_equal(oim.numpy(), wim))
        # reduced outputs (peaks) gather the same way
        peaks = torch.arange(sb.start, sb.stop, dtype=torch.int32)
        ok = ok and torch.equal(gather_rows(peaks, batch), torch.arange(batch, dtype=torch.int32))
        ok = ok and max_over_ranks(float(rank + 1)) == float(world)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,batch", [(2, 8), (2, 7), (2, 1), (3, 10), (2, 0)])
def test_sharded_batch_over_gloo(world, batch):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, batch, 64, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0, f"rank exited with {p.exitcode}"
    results = dict(q.get(timeout=10) for _ in range(world))
    assert results == {r: True for r in range(world)}


def test_shard_bounds_properties():
    from pragma_dsp_amd.shard import gather_rows, shard_bounds
    for batch in (0, 1, 7, 8, 65536, 524288, 1000003):
        for world in (1, 2, 3, 4, 8):
            b = shard_bounds(batch, world)
            assert b[0][0] == 0 and b[-1][1] == batch
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [y - x for x, y in b]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
    assert shard_bounds(524288, 8) == [(r * 65536, (r + 1) * 65536) for r in range(8)]  # BASELINE configs[4]
    with pytest.raises(ValueError):
        shard_bounds(4, 0)
    t = torch.arange(6).reshape(3, 2)
    assert gather_rows(t, 3) is t  # single process: no collective

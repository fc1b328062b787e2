"""Batch sharding across the GPUs of one node.

Every transform / frame is independent (the reference processes frames one by one:
src/effect/index.ts:190-194, bench/reallife/signals.ts:264-270), so the path shards by
rows with NO collective during compute: rank r of R owns the contiguous rows
[start_r, stop_r), the plan (twiddles, window) is replicated per device.  The only
exchange step is the optional final gather of the per-rank output slabs, one RCCL
all-gather over xGMI (backend "nccl" on ROCm); on CPU the same code runs over gloo,
which is how tests/test_shard_cpu.py covers it with world_size 2.

One process per GPU, launched by torch.distributed.run; nothing here spawns processes.
"""
from __future__ import annotations

from typing import Callable, Sequence

import torch
import torch.distributed as dist


def shard_bounds(batch: int, world: int) -> list[tuple[int, int]]:
    """Contiguous split of `batch` rows over `world` ranks; the first batch % world
    ranks take one extra row, so sizes differ by at most one and stay ordered."""
    if batch < 0 or world <= 0:
        raise ValueError(f"bad shard request: batch={batch} world={world}")
    base, extra = divmod(batch, world)
    out, start = [], 0
    for r in range(world):
        n = base + (1 if r < extra else 0)
        out.append((start, start + n))
        start += n
    return out


def my_rows(batch: int, rank: int | None = None, world: int | None = None) -> tuple[int, int]:
    """[start, stop) of this rank (defaults: the initialised process group, else 0 of 1)."""
    if world is None:
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    return shard_bounds(batch, world)[rank]


def gather_rows(local: torch.Tensor, batch: int, group=None, force_collective: bool = False) -> torch.Tensor:
    """All-gather the per-rank slabs `local` ([rows_r, ...]) into the full [batch, ...]
    tensor on every rank, in rank order.  Ragged shards are padded to the largest one
    for the collective (one all_gather_into_tensor = one RCCL ring over xGMI) and
    trimmed afterwards.  A single process returns `local` itself -- unless
    `force_collective` asks for the collective call even in a group of one (bench.py's
    `--rccl-selftest`: the same all_gather_into_tensor on one card)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1 and not (force_collective and dist.is_initialized()):
        if local.shape[0] != batch:
            raise ValueError(f"local slab has {local.shape[0]} rows, batch is {batch}")
        return local
    bounds = shard_bounds(batch, world)
    rank = dist.get_rank(group)
    rows = bounds[rank][1] - bounds[rank][0]
    if local.shape[0] != rows:
        raise ValueError(f"rank {rank} owns {rows} rows but its slab has {local.shape[0]}")
    most = max(b - a for a, b in bounds)
    send = local.contiguous()
    if rows != most:  # pad the short shards
        pad = torch.zeros((most - rows,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        send = torch.cat([send, pad], dim=0)
    full = torch.empty((world * most,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(full, send, group=group)
    if all(b - a == most for a, b in bounds):
        return full
    return torch.cat([full[r * most:r * most + (b - a)] for r, (a, b) in enumerate(bounds)], dim=0)


class ShardedBatch:
    """Runs a per-row device function on this rank's rows of a global batch.

    `compute(local_inputs...) -> tensor or tuple of tensors` is e.g.
    `BatchedFft(N).forward` or `BatchedFft(N).spectrum`; it sees only local rows, so no
    collective sits on the data path.  `gather=True` appends the final all-gather.
    """

    def __init__(self, batch: int, compute: Callable, group=None):
        self.batch = int(batch)
        self.compute = compute
        self.group = group
        self.start, self.stop = my_rows(self.batch,
                                        dist.get_rank(group) if dist.is_initialized() else 0,
                                        dist.get_world_size(group) if dist.is_initialized() else 1)

    @property
    def rows(self) -> int:
        return self.stop - self.start

    def local(self, full: torch.Tensor) -> torch.Tensor:
        """This rank's rows of a tensor indexed by global row."""
        return full[self.start:self.stop]

    def run(self, *local_inputs, gather: bool = False):
        for t in local_inputs:
            if t is not None and t.shape[0] != self.rows:
                raise ValueError(f"expected {self.rows} local rows, got {t.shape[0]}")
        out = self.compute(*local_inputs)
        if not gather:
            return out
        if isinstance(out, (tuple, list)):
            return type(out)(None if o is None else gather_rows(o, self.batch, self.group) for o in out)
        return gather_rows(out, self.batch, self.group)


def max_over_ranks(value: float, device=None, group=None) -> float:
    """MAX all-reduce of a host scalar (bench timing contract)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())

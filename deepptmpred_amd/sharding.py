"""Data-parallel scoring across the GPUs of one node: one process per GPU.

Every site is scored independently (the reference's ``model.predict`` has no
cross-sample term at inference: BatchNorm uses its moving statistics,
model.py:135-205), so the path shards by contiguous site ranges with replicated
weights and needs no collective in the data path.  The only communication is the
gather of the per-rank probability slices onto rank 0 after the scoring call
(``torch.distributed`` -- RCCL on GPUs, gloo in the CPU tests).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Rank r scores sites [r*n/G, (r+1)*n/G): contiguous, sizes differ by at most one."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"bad rank/world: {rank}/{world}")
    return (rank * n) // world, ((rank + 1) * n) // world


def shard_inputs(x: Sequence[np.ndarray], rank: int, world: int) -> List[np.ndarray]:
    n = len(x[0])
    lo, hi = shard_range(n, rank, world)
    return [np.asarray(a)[lo:hi] for a in x]


def predict_sharded(score_fn: Callable[[List[np.ndarray]], np.ndarray], x: Sequence[np.ndarray],
                    group=None, dst: int = 0) -> Optional[np.ndarray]:
    """Scores ``x = [x21, x33, x51, struct, esm]`` (every rank holds, or can slice, the
    full arrays) with ``score_fn`` on this rank's shard and gathers float32 ``[N,2]`` on
    rank ``dst`` in input order.  Returns ``None`` on the other ranks.

    ``score_fn`` is ``Model.predict`` in production; with an uninitialised process group
    the call degenerates to ``score_fn(x)``."""
    import torch
    import torch.distributed as dist

    n = len(x[0])
    if not (dist.is_available() and dist.is_initialized()):
        return np.asarray(score_fn(list(x)), dtype=np.float32)
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_range(n, rank, world)
    mine = np.asarray(score_fn([np.asarray(a)[lo:hi] for a in x]), dtype=np.float32).reshape(hi - lo, 2)
    # gather on the host: pad every slice to the longest so one gather call serves all ranks
    longest = max(shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world))
    buf = torch.zeros((longest, 2), dtype=torch.float32)
    buf[: hi - lo] = torch.from_numpy(mine)
    backend = dist.get_backend(group)
    if backend == "nccl":  # RCCL moves device memory only
        buf = buf.cuda()
    parts = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, parts, dst=dst, group=group)
    if rank != dst:
        return None
    out = np.empty((n, 2), dtype=np.float32)
    for r, p in enumerate(parts):
        a, b = shard_range(n, r, world)
        out[a:b] = p[: b - a].cpu().numpy()
    return out

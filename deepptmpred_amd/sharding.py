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


def predict_all_types(models, x: Sequence[np.ndarray], pass_sites: int = 262144) -> dict:
    """Scores the same sites with several weight sets (the 17 PTM types of predict.py:119-137).

    ``models`` maps a name to a ``Model`` on this process's GPU.  The inputs cross PCIe ONCE
    per pass of ``pass_sites`` sites (pinned staging, device buffers from torch) and every
    model scores the resident pass through the device-pointer entry point, so the upload
    cost is shared by all types; results are float32 ``[N,2]`` per name.  Combine with
    ``shard_inputs`` to spread the sites over the GPUs of a node (shard by sites, not by
    type: every rank stays balanced and every input is uploaded once)."""
    import torch

    names = list(models)
    n = len(x[0])
    out = {k: np.empty((n, 2), dtype=np.float32) for k in names}
    if n == 0 or not names:
        return out
    first = models[names[0]]
    dev = torch.device(f"cuda:{first.device}")
    torch.cuda.set_device(dev)
    # an explicit stream: a NULL stream argument means "the handle's own stream" to the library, which
    # would not be ordered after the uploads torch issues on its default stream
    tstream = torch.cuda.Stream(device=dev)
    stream = tstream.cuda_stream
    ws = first.spec.window_sizes
    dtypes = [np.uint8, np.uint8, np.uint8, np.float32, np.float32]
    widths = [ws[0], ws[1], ws[2], first.spec.struct_feature_dim, first.spec.esm_dim]
    cap = min(pass_sites, n)
    stage = [torch.empty((cap, w), dtype=torch.uint8 if d is np.uint8 else torch.float32).pin_memory()
             for w, d in zip(widths, dtypes)]
    dbuf = [torch.empty((cap, w), dtype=s.dtype, device=dev) for w, s in zip(widths, stage)]
    dout = torch.empty((cap, 2), dtype=torch.float32, device=dev)
    with torch.cuda.stream(tstream):
        for lo in range(0, n, cap):
            m = min(cap, n - lo)
            for s, d, a, dt in zip(stage, dbuf, x, dtypes):
                s[:m].copy_(torch.from_numpy(np.ascontiguousarray(np.asarray(a)[lo:lo + m], dtype=dt)))
                d[:m].copy_(s[:m], non_blocking=True)
            for k in names:
                models[k].scorer.score_device(dbuf[0].data_ptr(), dbuf[1].data_ptr(), dbuf[2].data_ptr(),
                                              dbuf[3].data_ptr(), dbuf[4].data_ptr(), m, dout.data_ptr(), stream)
                out[k][lo:lo + m] = dout[:m].cpu().numpy()      # synchronises the stream
    return out

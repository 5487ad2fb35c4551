"""Multi-rank path on CPU: two gloo processes score disjoint site ranges and gather on rank 0.
The scoring function injected here is the CPU oracle (there is no GPU in this test); the
sharding, ordering and gather code is the product's."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from deepptmpred_amd import spec  # noqa: E402
from deepptmpred_amd.sharding import predict_sharded, shard_inputs, shard_range  # noqa: E402


def test_shard_ranges_cover_everything():
    for n in (0, 1, 2, 7, 53, 80, 1000, 1_000_003):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)
    x = spec.synthetic_inputs(11, 1)
    parts = [shard_inputs(x, r, 3) for r in range(3)]
    assert sum(len(p[0]) for p in parts) == 11
    assert np.array_equal(np.concatenate([p[4] for p in parts]), x[4])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from oracle import ptm_oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w = spec.synthetic_weights(7)
    x = spec.synthetic_inputs(n, 42)
    calls = []

    def score(batch):
        calls.append(len(batch[0]))
        with np.errstate(over="ignore"):
            return O.forward(w, batch, np.float32).astype(np.float32)

    out = predict_sharded(score, x)
    lo, hi = shard_range(n, rank, world)
    assert calls == [hi - lo]
    if rank == 0:
        q.put(out)
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [37, 4])
def test_two_ranks_equal_one_rank(n):
    from oracle import ptm_oracle as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0
    w = spec.synthetic_weights(7)
    x = spec.synthetic_inputs(n, 42)
    with np.errstate(over="ignore"):
        want = O.forward(w, x, np.float32).astype(np.float32)
    assert got.shape == (n, 2) and got.dtype == np.float32
    # a row's result does not depend on which rows share its batch: agreement to rounding
    assert np.abs(got - want).max() < 1e-6


def test_no_process_group_is_plain_call():
    x = spec.synthetic_inputs(5, 3)
    out = predict_sharded(lambda b: np.full((len(b[0]), 2), 0.5, dtype=np.float32), x)
    assert out.shape == (5, 2)

"""Host logic of the training path on CPU: the learning-rate schedule, and the
data-parallel orchestration with two gloo ranks.  The per-rank arithmetic is injected
(the CPU oracle stands in for the GPU kernels); sharding, the N/N_pos exchange, the l2
scaling and the flat-gradient all-reduce are the product's code."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from deepptmpred_amd import spec  # noqa: E402
from deepptmpred_amd.sharding import shard_range  # noqa: E402
from deepptmpred_amd.training import AdamW, DataParallelTrainer, DropoutRates, labels_from_targets, lr_schedule  # noqa: E402


def test_lr_schedule_matches_reference():
    # trainer.py:56-59 with config.py:57,60 (5e-4, 5 warm-up epochs)
    lr = 5e-4
    seen = []
    for ep in range(8):
        lr = lr_schedule(ep, lr)
        seen.append(lr)
    assert np.allclose(seen[:5], [1e-4, 2e-4, 3e-4, 4e-4, 5e-4])
    assert np.allclose(seen[5:], [5e-4 * 0.95, 5e-4 * 0.95 ** 2, 5e-4 * 0.95 ** 3])


def test_labels_from_targets():
    assert labels_from_targets(np.array([[1, 0], [0, 1], [0, 1]])).tolist() == [0, 1, 1]
    assert labels_from_targets([1, 0, 1]).dtype == np.int32


class OracleBackend:
    """Per-rank arithmetic from oracle/train_ref.py, with the backend interface of
    deepptmpred_amd.training.NativeBackend."""

    def __init__(self, weights):
        from oracle.train_ref import TrainRef
        self.ref = TrainRef(weights)
        self.g = None

    def forward_backward(self, x, labels, n_global, n_pos_global, seed, rates, l2, l2_scale):
        rd = dict(embed=rates.embed, conv=rates.conv, tb=rates.tb, dense=rates.dense)
        loss, self.g = self.ref.grads(x, labels, seed, rd, n_global, n_pos_global, l2_scale)
        return loss

    def allreduce_grads(self, group=None):
        import torch.distributed as dist
        # one flat bucket: gradients + the BatchNorm statistics of this replica's batch
        bucket = dict(self.g)
        bucket.update({"stat:" + k: v.numpy() for k, v in self.ref.new_stats.items()})
        keys = list(bucket)
        flat = torch.from_numpy(np.concatenate([np.asarray(bucket[k], dtype=np.float64).ravel() for k in keys]))
        dist.all_reduce(flat, group=group)
        off = 0
        for k in keys:
            n = bucket[k].size
            bucket[k] = flat[off:off + n].numpy().reshape(bucket[k].shape).copy()
            off += n
        self.g = {k: v for k, v in bucket.items() if not k.startswith("stat:")}
        self.stats_sum = {k[5:]: v for k, v in bucket.items() if k.startswith("stat:")}

    def apply(self, lr, wd, stat_scale=1.0):
        if getattr(self, "stats_sum", None) is not None:
            self.ref.new_stats = {k: torch.from_numpy(v * stat_scale) for k, v in self.stats_sum.items()}
        self.ref.apply(self.g, lr, wd)

    def weights(self):
        return self.ref.weights()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w = spec.synthetic_weights(7)
    x = spec.synthetic_inputs(12, 9)
    y = np.array([1, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0])
    dp = DataParallelTrainer(OracleBackend(w), AdamW(5e-4, 1e-4), DropoutRates.off())
    loss = dp.train_on_batch(x, np.eye(2)[y], seed=3)
    if rank == 0:
        q.put((loss, dp.backend.weights()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_step_equals_sum_of_shards():
    from oracle.train_ref import TrainRef
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    loss, got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    # the same step written out by hand: per-shard gradients with GLOBAL N / N_pos, summed
    w = spec.synthetic_weights(7)
    x = spec.synthetic_inputs(12, 9)
    y = np.array([1, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0])
    rates = dict(embed=0, conv=0, tb=0, dense=0)
    total, want_loss, stats = None, 0.0, None
    ref = TrainRef(w)
    for r in range(2):
        lo, hi = shard_range(12, r, 2)
        shard = TrainRef(w)
        l, g = shard.grads([a[lo:hi] for a in x], y[lo:hi], 3 * 1000003 + r, rates, 12, int(y.sum()), 0.5)
        want_loss += l
        total = g if total is None else {k: total[k] + g[k] for k in g}
        stats = dict(shard.new_stats) if stats is None else {k: stats[k] + shard.new_stats[k] for k in stats}
    ref.new_stats = {k: v / 2 for k, v in stats.items()}      # moving stats follow the replica average
    ref.apply(total, 5e-4, 1e-4)
    assert abs(loss - want_loss) < 1e-9
    for k, v in ref.weights().items():
        assert np.abs(got[k] - v).max() < 1e-9, k

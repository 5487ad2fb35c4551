"""Training step and its data-parallel orchestration.

Mirrors what the reference does around ``model.fit`` (pred/train_PTM/trainer.py:
304-339): ``compile(optimizer=AdamW(learning_rate, weight_decay), loss=focal_loss)``
then per batch forward (training mode) -> class-balanced BCE (trainer.py:36-54) ->
backward -> AdamW.  The arithmetic runs in libdptm.so (``dptm_trainer_*``); this module
is host logic only: batching, the learning-rate schedule of trainer.py:56-59, and the
data-parallel exchange (one process per GPU):

  1. every replica is handed the labels of the WHOLE batch, so the GLOBAL counts ``N`` and
     ``N_pos`` that the single-device loss definition weights its samples with
     (trainer.py:39-44) are computed locally: no collective, no host round trip;
  2. local forward/backward on the replica's contiguous shard (BatchNorm normalises with the
     replica's own batch statistics, or -- ``sync_batchnorm=True`` -- with the global batch's:
     8 more small all-reduces per step, one per level of dependent BatchNorms and direction);
  3. ONE all-reduce (sum) over the flat fp32 gradient buffer: every tensor's gradient in
     one bucket, with the batch mean/variance of every BatchNorm riding in the slots of
     its (untrained) moving statistics and the replica's loss share in the bucket's spare
     float, so the loss needs no collective of its own;
  4. identical AdamW update on every replica; moving statistics move towards the
     replica-averaged batch statistics, so replicas stay bit-identical.

``fit`` keeps the training set on the device of EVERY replica (5.3 KB per site; the reference's
largest set, 59,707 phosphorylation sites, is 0.3 GB), gathers each step's shard there and reads
the per-step losses once per epoch, so that with one replica or many nothing in a step waits
for the GPU or crosses PCIe.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np

from . import _native
from .sharding import shard_range
from .spec import DEFAULT_SPEC, ModelSpec, pack_blob, unpack_blob


@dataclass
class AdamW:
    """Hyper-parameters of ``tensorflow_addons.optimizers.AdamW`` as the reference
    constructs it (trainer.py:305-306); beta_1 0.9, beta_2 0.999, epsilon 1e-7 are fixed
    in the kernel."""
    learning_rate: float = 5e-4
    weight_decay: float = 1e-4


@dataclass
class DropoutRates:
    """Training-time dropout rates of the graph (model.py:94,136,159,189-202 with
    config.py:52-53)."""
    embed: float = 1e-4
    conv: float = 0.3
    tb: float = 0.1
    dense: float = 0.3

    @classmethod
    def off(cls) -> "DropoutRates":
        return cls(0.0, 0.0, 0.0, 0.0)


def lr_schedule(epoch: int, lr: float, initial_lr: float = 5e-4, warmup_epochs: int = 5) -> float:
    """trainer.py:56-59: linear warm-up over ``warmup_epochs``, then x0.95 per epoch."""
    if epoch < warmup_epochs:
        return initial_lr * (epoch + 1) / warmup_epochs
    return lr * 0.95


def labels_from_targets(y) -> np.ndarray:
    """One-hot ``[N,2]`` targets (trainer.py:38 takes their argmax) or plain 0/1 labels."""
    y = np.asarray(y)
    if y.ndim == 2:
        y = np.argmax(y, axis=-1)
    return np.ascontiguousarray(y, dtype=np.int32)


class NativeBackend:
    """The HIP trainer on one GPU.  Batches are staged through torch tensors (device
    memory + stream plumbing only)."""

    def __init__(self, weights: Dict[str, np.ndarray], spec: ModelSpec = DEFAULT_SPEC, device: int = 0,
                 max_batch: int = 64):
        import torch
        self.torch = torch
        self.spec = spec
        self.device = torch.device(f"cuda:{device}")
        torch.cuda.set_device(self.device)
        self.trainer = _native.Trainer(pack_blob(weights, spec), spec, device, max_batch)
        ptr, count = self.trainer.grad_buffer()
        self._grad_view = self._wrap(ptr, count)
        self._stage = None
        self._data = None
        self._data_key = None
        # ONE stream carries staging copies, forward/backward, the gradient all-reduce and the optimizer update, so
        # they are ordered by the stream and not by host waits: the trainer's own main stream, seen by torch as an
        # external stream.  The step then starts and ends without a hand-over between streams, and the process has
        # no further busy stream to share a hardware queue with one of the step's four.
        self.stream = torch.cuda.ExternalStream(self.trainer.stream(), device=self.device)

    def _wrap(self, ptr: int, count: int):
        """torch view of library-owned device memory (for the RCCL all-reduce)."""
        class _Ext:
            __cuda_array_interface__ = {"shape": (count,), "typestr": "<f4", "data": (ptr, False), "version": 2}
        return self.torch.as_tensor(_Ext(), device=self.device)

    device_loss = True       # the loss can stay on the device and ride in the gradient all-reduce (allreduce_grads)

    def forward_backward(self, x: Sequence[np.ndarray], labels: np.ndarray, n_global: int, n_pos_global: int,
                         seed: int, rates: DropoutRates, l2: float, l2_scale: float, read_loss: bool = True):
        """Stages one local batch and enqueues its step; returns this replica's loss share, or ``None``
        without waiting for the GPU when ``read_loss`` is false."""
        t = self.torch
        # fixed staging buffers: a batch always lands at the same device addresses
        n = len(labels)
        d = self._staging()
        o = _native.TrainOpts()
        o.seed = seed & 0xFFFFFFFF
        o.rate_embed, o.rate_conv, o.rate_tb, o.rate_dense = rates.embed, rates.conv, rates.tb, rates.dense
        o.n_global, o.n_pos_global = int(n_global), int(n_pos_global)
        o.l2, o.l2_scale = l2, l2_scale
        self.stream.wait_stream(t.cuda.current_stream(self.device))
        with t.cuda.stream(self.stream):
            for buf, a in zip(d[:5], x):
                buf[:n].copy_(t.from_numpy(np.ascontiguousarray(a, dtype=np.uint8 if buf.dtype == t.uint8 else np.float32)),
                              non_blocking=False)
            d[5][:n].copy_(t.from_numpy(np.ascontiguousarray(labels, dtype=np.int32)))
            loss = self.trainer.forward_backward(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(),
                                                 d[4].data_ptr(), d[5].data_ptr(), len(labels), o, bool(read_loss),
                                                 self.stream.cuda_stream)
        return loss

    # ---- a data set resident on the device: batches are gathered there and the per-step losses stay there, so a
    # ---- step needs neither a host-to-device copy nor a wait for the GPU (Keras' fit reports the epoch mean only)
    RESIDENT_LIMIT_BYTES = 32 << 30

    def set_dataset(self, x: Sequence[np.ndarray], labels: np.ndarray) -> bool:
        """Uploads the whole training set once (a second call with the same arrays uploads nothing).  False when the
        set is larger than RESIDENT_LIMIT_BYTES: nothing is uploaded then."""
        t = self.torch
        n = len(labels)
        key = tuple(id(a) for a in x) + (id(labels), n)
        if self._data is not None and self._data_key == key:
            return True
        if n * (sum(self.spec.window_sizes) + 4 * (self.spec.struct_feature_dim + self.spec.esm_dim) + 4) > self.RESIDENT_LIMIT_BYTES:
            return False
        self.stream.wait_stream(t.cuda.current_stream(self.device))
        with t.cuda.stream(self.stream):
            data = [t.from_numpy(np.ascontiguousarray(a, dtype=np.uint8)).to(self.device) for a in x[:3]]
            data += [t.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(self.device) for a in x[3:5]]
            data.append(t.from_numpy(np.ascontiguousarray(labels, dtype=np.int32)).to(self.device))
        self._data = data
        self._data_key = key
        return True

    def drop_dataset(self) -> None:
        self._data = None
        self._data_key = None

    def _staging(self):
        t = self.torch
        if self._stage is None:
            mb = self.trainer.max_batch
            ws = self.spec.window_sizes
            self._stage = [t.empty((mb, ws[0]), dtype=t.uint8, device=self.device),
                           t.empty((mb, ws[1]), dtype=t.uint8, device=self.device),
                           t.empty((mb, ws[2]), dtype=t.uint8, device=self.device),
                           t.empty((mb, self.spec.struct_feature_dim), dtype=t.float32, device=self.device),
                           t.empty((mb, self.spec.esm_dim), dtype=t.float32, device=self.device),
                           t.empty((mb,), dtype=t.int32, device=self.device)]
        return self._stage

    def forward_backward_resident(self, idx_dev, n_global: int, n_pos_global: int, seed: int, rates: DropoutRates,
                                  l2: float, l2_scale: float, loss_out, proba_out=None) -> None:
        """One step on the rows ``idx_dev`` (device int64 tensor) of the resident data set; the step's loss is copied
        to the one-element device tensor ``loss_out`` (``None``: the caller takes it from ``allreduce_grads``) and
        its class probabilities -- the training-mode outputs -- to ``proba_out`` (2 floats per row).
        Nothing here waits for the GPU."""
        t = self.torch
        n = int(idx_dev.numel())
        o = _native.TrainOpts()
        o.seed = seed & 0xFFFFFFFF
        o.rate_embed, o.rate_conv, o.rate_tb, o.rate_dense = rates.embed, rates.conv, rates.tb, rates.dense
        o.n_global, o.n_pos_global = int(n_global), int(n_pos_global)
        o.l2, o.l2_scale = l2, l2_scale
        a = self._data
        if not idx_dev.is_contiguous():
            idx_dev = idx_dev.contiguous()
        with t.cuda.stream(self.stream):
            # one gather launch, the step, one launch for the loss and the probabilities (dptm_trainer_forward_backward_rows)
            self.trainer.forward_backward_rows(a[0].data_ptr(), a[1].data_ptr(), a[2].data_ptr(), a[3].data_ptr(),
                                               a[4].data_ptr(), a[5].data_ptr(), int(a[5].numel()), idx_dev.data_ptr(), n, o,
                                               loss_out.data_ptr() if loss_out is not None else 0,
                                               proba_out.data_ptr() if proba_out is not None else 0, self.stream.cuda_stream)

    def allreduce_floats(self, dev_tensor, group=None) -> None:
        """Sum of a device tensor over the replicas, in the order of the backend's stream."""
        import torch.distributed as dist
        with self.torch.cuda.stream(self.stream):
            if dist.get_backend(group) == "nccl":
                dist.all_reduce(dev_tensor, op=dist.ReduceOp.SUM, group=group)
            else:
                h = dev_tensor.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
                dev_tensor.copy_(h)

    def enable_sync_batchnorm(self, group=None) -> None:
        """Every BatchNorm then normalises with the statistics of the GLOBAL batch: its column sums are
        all-reduced over ``group`` (RCCL on GPUs) inside the step, on the stream the layer runs on, so
        N replicas reproduce a one-device run on the whole batch.  8 small collectives per step (``sync_bn_calls`` counts them)."""
        import torch.distributed as dist
        t = self.torch
        views = {}

        on_device = dist.get_backend(group) == "nccl"     # RCCL reduces device memory in stream order

        self.sync_bn_calls = 0

        def allreduce(ptr: int, count: int, stream: int) -> None:
            self.sync_bn_calls += 1
            v = views.get((ptr, count))
            if v is None:
                v = views[(ptr, count)] = self._wrap(ptr, count)
            s = t.cuda.ExternalStream(stream, device=self.device) if stream else self.stream
            with t.cuda.stream(s):
                if on_device:
                    dist.all_reduce(v, op=dist.ReduceOp.SUM, group=group)
                else:                                      # host backends (gloo): through host memory
                    h = v.cpu()                            # waits for the sums on this stream
                    dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
                    v.copy_(h)

        self.trainer.set_allreduce(allreduce)

    def disable_sync_batchnorm(self) -> None:
        self.trainer.set_allreduce(None)

    def allreduce_grads(self, group=None, loss_out=None) -> None:
        """ONE all-reduce (sum) of the flat gradient buffer, on the backend's stream (RCCL); host backends
        (gloo: CPU tests, one-GPU rehearsals) go through host memory.  With ``loss_out`` (a one-element device
        tensor) the replica's loss share rides in the bucket's spare float and the summed loss is copied there."""
        import torch.distributed as dist
        with self.torch.cuda.stream(self.stream):
            slot = self.trainer.stage_loss(self.stream.cuda_stream) if loss_out is not None else -1
            if dist.get_backend(group) == "nccl":
                dist.all_reduce(self._grad_view, op=dist.ReduceOp.SUM, group=group)
            else:
                h = self._grad_view.cpu()                  # waits for the gradients on this stream
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
                self._grad_view.copy_(h)
            if loss_out is not None:
                loss_out.copy_(self._grad_view[slot:slot + 1], non_blocking=True)

    def new_loss_buffer(self, n: int = 1):
        """``n`` zeroed floats on the device, allocated in the order of the backend's stream."""
        self.stream.wait_stream(self.torch.cuda.current_stream(self.device))
        with self.torch.cuda.stream(self.stream):
            return self.torch.zeros(n, dtype=self.torch.float32, device=self.device)

    def read_floats(self, dev_tensor) -> np.ndarray:
        """Host copy of a device tensor written on the backend's stream (waits for that stream)."""
        with self.torch.cuda.stream(self.stream):
            return dev_tensor.cpu().numpy()

    def apply(self, lr: float, weight_decay: float, stat_scale: float = 1.0) -> None:
        self.trainer.apply(lr, weight_decay, 1.0, stat_scale, self.stream.cuda_stream)

    def weights(self) -> Dict[str, np.ndarray]:
        return unpack_blob(self.trainer.get_weights(), self.spec)

    def close(self) -> None:
        self.trainer.close()


class DataParallelTrainer:
    """One training step over the global batch ``(x, y)``; every rank calls it with the
    same arrays and scores its own contiguous shard (``shard_range``)."""

    def __init__(self, backend, optimizer: AdamW = AdamW(), rates: DropoutRates = DropoutRates(),
                 l2: float = 1e-4, group=None, sync_batchnorm: bool = False):
        self.backend = backend
        if sync_batchnorm:
            backend.enable_sync_batchnorm(group)
        self.optimizer = optimizer
        self.rates = rates
        self.l2 = l2
        self.group = group
        self.steps = 0

    def _dist(self):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist, dist.get_rank(self.group), dist.get_world_size(self.group)
        return None, 0, 1

    def train_on_batch(self, x: Sequence[np.ndarray], y, lr: Optional[float] = None, seed: Optional[int] = None) -> float:
        """One step on the global batch ``(x, y)``: every rank passes the same arrays, works on its shard and
        returns the loss of the whole batch.  The only collective is the gradient all-reduce; the only wait for
        the GPU is the read of the loss at the end."""
        dist, rank, world = self._dist()
        labels = labels_from_targets(y)
        n = len(labels)
        if any(len(a) != n for a in x) or len(x) != 5:
            raise ValueError("train_on_batch expects [x21, x33, x51, struct, esm] and matching targets")
        lo, hi = shard_range(n, rank, world)
        if hi - lo < 1:
            raise ValueError("every replica needs at least one sample of the batch")
        local_x = [np.asarray(a)[lo:hi] for a in x]
        local_y = labels[lo:hi]
        # N and N_pos of trainer.py:39-40 over the whole batch: every rank holds all labels
        n_global, n_pos = n, int(labels.sum())
        step_seed = (self.steps if seed is None else seed) * 1000003 + rank
        step_lr = self.optimizer.learning_rate if lr is None else lr
        if dist is not None and getattr(self.backend, "device_loss", False):
            self.backend.forward_backward(local_x, local_y, n_global, n_pos, step_seed, self.rates, self.l2, 1.0 / world,
                                          read_loss=False)
            if getattr(self, "_loss_dev", None) is None:
                self._loss_dev = self.backend.new_loss_buffer(1)
            self.backend.allreduce_grads(self.group, loss_out=self._loss_dev)
            self.backend.apply(step_lr, self.optimizer.weight_decay, 1.0 / world)
            self.steps += 1
            return float(self.backend.read_floats(self._loss_dev)[0])
        loss = self.backend.forward_backward(local_x, local_y, n_global, n_pos, step_seed, self.rates, self.l2, 1.0 / world)
        if dist is not None:       # a backend that hands its loss share to the host: one more small host all-reduce
            import torch
            self.backend.allreduce_grads(self.group)
            lt = torch.tensor([loss], dtype=torch.float64)
            ldev = lt.cuda() if dist.get_backend(self.group) == "nccl" else lt
            dist.all_reduce(ldev, op=dist.ReduceOp.SUM, group=self.group)
            loss = float(ldev.cpu().item())
        self.backend.apply(step_lr, self.optimizer.weight_decay, 1.0 / world)
        self.steps += 1
        return loss

    def fit(self, x: Sequence[np.ndarray], y, batch_size: int = 64, epochs: int = 1, shuffle: bool = True,
            lr_scheduler: Optional[Callable[[int, float], float]] = None, seed: int = 0, verbose: int = 0,
            keep_resident: bool = False) -> Dict[str, List[float]]:
        """``model.fit`` loop (trainer.py:332-339) without the reference's callbacks other
        than the learning-rate schedule.  ``batch_size`` is the GLOBAL batch; with several replicas every rank
        calls this with the same arrays and seed, and takes its contiguous shard of every batch."""
        labels = y if isinstance(y, np.ndarray) and y.ndim == 1 and y.dtype == np.int32 else labels_from_targets(y)
        n = len(labels)
        lr = self.optimizer.learning_rate
        hist: Dict[str, List[float]] = {"loss": [], "lr": []}
        rng = np.random.default_rng(seed)
        dist, rank, world = self._dist()
        # the data set goes to the device once (every replica holds all of it: a shuffled batch may draw any row), the
        # shards of a batch are gathered there, and the per-step losses are read once per epoch; same batches, seeds
        # and arithmetic as the train_on_batch loop below
        resident = hasattr(self.backend, "set_dataset") and self.backend.set_dataset(x, labels)
        try:
            for ep in range(epochs):
                if lr_scheduler is not None:
                    lr = float(lr_scheduler(ep, lr))
                order = rng.permutation(n) if shuffle else np.arange(n)
                if resident:
                    torch = self.backend.torch
                    starts = list(range(0, n, batch_size))
                    losses_dev = self.backend.new_loss_buffer(len(starts))
                    proba_dev = self.backend.new_loss_buffer(2 * n)     # row i: the outputs for sample order[i]
                    with torch.cuda.stream(self.backend.stream):
                        order_dev = torch.from_numpy(np.ascontiguousarray(order, dtype=np.int64)).to(self.backend.device)
                    for k, s in enumerate(starts):
                        idx = order[s:s + batch_size]
                        lo, hi = shard_range(len(idx), rank, world)
                        if hi - lo < 1:
                            raise ValueError("every replica needs at least one sample of every batch")
                        own = losses_dev[k:k + 1]
                        self.backend.forward_backward_resident(order_dev[s + lo:s + hi], len(idx), int(labels[idx].sum()),
                                                               self.steps * 1000003 + rank, self.rates, self.l2, 1.0 / world,
                                                               own if world == 1 else None,
                                                               proba_dev[2 * (s + lo):2 * (s + hi)])
                        if world > 1:
                            self.backend.allreduce_grads(self.group, loss_out=own)
                        self.backend.apply(lr, self.optimizer.weight_decay, 1.0 / world)
                        self.steps += 1
                    if world > 1:          # every replica filled the rows of its shards: one sum per epoch joins them
                        self.backend.allreduce_floats(proba_dev, self.group)
                    hist["loss"].append(float(self.backend.read_floats(losses_dev).astype(np.float64).mean()))
                    proba = self.backend.read_floats(proba_dev).reshape(n, 2)
                else:
                    losses = []
                    can_read = world == 1 and hasattr(getattr(self.backend, "trainer", None), "get_proba")
                    proba = np.zeros((n, 2), np.float32) if can_read else None
                    for s in range(0, n, batch_size):
                        idx = order[s:s + batch_size]
                        losses.append(self.train_on_batch([np.asarray(a)[idx] for a in x], labels[idx], lr))
                        if can_read:
                            proba[s:s + len(idx)] = self.backend.trainer.get_proba(len(idx))
                    hist["loss"].append(float(np.mean(losses)))
                if proba is not None:
                    # what Keras logs as `accuracy` and `auc` (trainer.py:308): accumulated over the epoch's batches
                    # from the training-mode outputs, weights changing from batch to batch
                    from .callbacks import roc_auc
                    yl = labels[order]
                    hist.setdefault("accuracy", []).append(float(np.mean(np.argmax(proba, axis=1) == yl)))
                    hist.setdefault("auc", []).append(roc_auc(yl, proba[:, 1]))
                hist["lr"].append(lr)
                if verbose:
                    extra = "".join(f" - {k} {hist[k][-1]:.4f}" for k in ("accuracy", "auc") if k in hist)
                    print(f"epoch {ep + 1}/{epochs} - loss {hist['loss'][-1]:.4f}{extra} - lr {lr:.2e}")
        except BaseException:
            if resident:                        # never leave a device copy behind that a later call could mistake for its data
                self.backend.drop_dataset()
            raise
        if resident and not keep_resident:      # (keep_resident: the caller's next fit call passes the same arrays)
            self.backend.drop_dataset()
        return hist

"""The native training step (C ABI dptm_trainer_*) against the torch-autograd restatement
(oracle/train_ref.py, fp64).  Tolerances: gradients within 2e-4 of the largest entry of
each tensor (fp32 kernels with atomics vs an fp64 reference), loss 1e-4 relative."""
import numpy as np
import pytest

from deepptmpred_amd import _native, spec
from oracle.train_ref import TrainRef

pytestmark = pytest.mark.gpu


def _device_batch(x, y):
    import torch
    dev = torch.device("cuda:0")
    d = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in x]
    lab = torch.from_numpy(np.asarray(y, dtype=np.int32)).to(dev)
    return d, lab


def _opts(seed, rates, y, l2=1e-4):
    o = _native.TrainOpts()
    o.seed = seed
    o.rate_embed, o.rate_conv, o.rate_tb, o.rate_dense = rates
    o.n_global = len(y)
    o.n_pos_global = int(np.sum(y))
    o.l2 = l2
    o.l2_scale = 1.0
    return o


def _compare_grads(got_blob, want, label):
    got = spec.unpack_blob(got_blob)
    worst, bad = 0.0, []
    for k, g in want.items():
        scale = max(np.abs(g).max(), 1e-3)  # biases in front of a BatchNorm and the key bias have zero true gradient
        err = np.abs(got[k] - g).max() / scale
        worst = max(worst, err)
        if err >= 2e-4:
            bad.append(f"{k}: {err:.3e} (max |g| {scale:.3e})")
    assert not bad, f"{label}: gradients off: " + "; ".join(bad)
    return worst


# 48 sites: the small-batch kernels; 230 sites (4,830 to 11,730 window rows): the LDS-staged matrix products that
# large row stacks take, with a ragged last row tile
# The TransformerBlock kernels tile a branch by whole sites in four or eight wavefronts: 100 and 230 sites take the
# four-wavefront tiles (100 also the four-row-tile instance of the small-batch per-site kernels, 65..128 sites), 600
# sites the eight-wavefront tiles with fewer tiles than workgroups, 2,411 sites a ragged last tile and more tiles than
# workgroups (partial gradients accumulated over several tiles), 4,100 sites the per-site layers' LDS-staged
# products (row stacks of 4,096 and more).
@pytest.mark.parametrize("rates,n", [((0.0, 0.0, 0.0, 0.0), 48), ((1e-4, 0.3, 0.1, 0.3), 48), ((1e-4, 0.3, 0.1, 0.3), 100),
                                     ((1e-4, 0.3, 0.1, 0.3), 230), ((1e-4, 0.3, 0.1, 0.3), 600), ((1e-4, 0.3, 0.1, 0.3), 2411),
                                     ((1e-4, 0.3, 0.1, 0.3), 4100)])
def test_gradients_match_autograd(rates, n):
    import torch
    w = spec.synthetic_weights(7)
    x = spec.synthetic_inputs(n, 321)
    y = (np.random.default_rng(5).random(n) < 0.25).astype(np.int64)
    ref = TrainRef(w)
    rd = dict(zip(("embed", "conv", "tb", "dense"), rates))
    loss_ref, g_ref = ref.step(x, y, lr=5e-4, weight_decay=1e-4, seed=99, rates=rd)

    tr = _native.Trainer(spec.pack_blob(w), max_batch=max(64, n))
    d, lab = _device_batch(x, y)
    loss = tr.forward_backward(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(),
                               lab.data_ptr(), n, _opts(99, rates, y))
    torch.cuda.synchronize()
    assert abs(loss - loss_ref) < 1e-4 * max(1.0, abs(loss_ref)), (loss, loss_ref)
    worst = _compare_grads(tr.get_grads(), g_ref, f"rates={rates}")
    print("worst relative gradient error", worst)
    # optimizer step: weights (and BatchNorm moving statistics) after AdamW
    tr.apply(5e-4, 1e-4)
    got = spec.unpack_blob(tr.get_weights())
    for k, v in ref.weights().items():
        assert np.abs(got[k] - v).max() < 2e-5 * max(1.0, np.abs(v).max()), k
    tr.close()


# batch sizes on both sides of every switch the step makes: 1 row tile / 2 / 4 of the per-site kernels (32|33, 64|65),
# one round of one-site tiles / more (85|86), one-site four-wavefront tiles / whole sites in four wavefronts (112|113), four / eight row tiles of the per-site kernels
# (129), per-site kernels / layer-wise products (192|193),
# four- / eight-wavefront tiles (231|232), the three branches' launches as one / on three streams (540|541); 7 sites: a batch far below one row tile (with 2 sites BatchNorm over two rows amplifies fp32 rounding beyond the bound)
@pytest.mark.parametrize("n", [7, 33, 65, 85, 86, 112, 113, 129, 192, 193, 231, 232, 540, 541])
def test_gradients_at_regime_boundaries(n):
    import torch
    rates = (1e-4, 0.3, 0.1, 0.3)
    w = spec.synthetic_weights(11)
    x = spec.synthetic_inputs(n, 77 + n)
    y = (np.arange(n) % 3 == 0).astype(np.int64)
    ref = TrainRef(w)
    loss_ref, g_ref = ref.step(x, y, lr=5e-4, weight_decay=1e-4, seed=7, rates=dict(zip(("embed", "conv", "tb", "dense"), rates)))
    tr = _native.Trainer(spec.pack_blob(w), max_batch=max(64, n))
    d, lab = _device_batch(x, y)
    loss = tr.forward_backward(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(),
                               lab.data_ptr(), n, _opts(7, rates, y))
    torch.cuda.synchronize()
    assert abs(loss - loss_ref) < 1e-4 * max(1.0, abs(loss_ref)), (loss, loss_ref)
    _compare_grads(tr.get_grads(), g_ref, f"n={n}")
    tr.close()


def test_two_steps_track_reference():
    import torch
    w = spec.synthetic_weights(101)
    ref = TrainRef(w)
    tr = _native.Trainer(spec.pack_blob(w), max_batch=32)
    rates = (0.0, 0.0, 0.0, 0.0)
    for step in range(2):
        n = 32
        x = spec.synthetic_inputs(n, 700 + step)
        y = (np.random.default_rng(step).random(n) < 0.3).astype(np.int64)
        loss_ref, _ = ref.step(x, y, seed=step, rates=dict(embed=0, conv=0, tb=0, dense=0))
        d, lab = _device_batch(x, y)
        loss = tr.forward_backward(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(),
                                   d[4].data_ptr(), lab.data_ptr(), n, _opts(step, rates, y))
        torch.cuda.synchronize()
        assert abs(loss - loss_ref) < 5e-5 * max(1.0, abs(loss_ref)), (step, loss, loss_ref)
        tr.apply(5e-4, 1e-4)
    got = spec.unpack_blob(tr.get_weights())
    for k, v in ref.weights().items():
        assert np.abs(got[k] - v).max() < 1e-4 * max(1.0, np.abs(v).max()), k
    tr.close()


def test_keras_like_fit_reduces_loss_and_updates_predictions():
    """compile / fit / predict round trip through the Model surface (trainer.py:304-339)."""
    from deepptmpred_amd.model import create_ptm_model
    from deepptmpred_amd.training import AdamW, lr_schedule
    m = create_ptm_model(seed=5)
    x = spec.synthetic_inputs(256, 77)
    y = (x[3][:, 0] > 125).astype(np.int64)           # a learnable rule: SASA above the median
    before = m.predict(x)
    m.compile(optimizer=AdamW(learning_rate=5e-4, weight_decay=1e-4), loss=None, metrics=["accuracy"])
    hist = m.fit(x, np.eye(2)[y], batch_size=64, epochs=6, verbose=0,
                 callbacks=[lambda ep, lr: lr_schedule(ep, lr)])
    assert len(hist["loss"]) == 6 and hist["loss"][-1] < hist["loss"][0]
    after = m.predict(x)
    assert np.isfinite(after).all() and np.abs(after - before).max() > 1e-3
    ws = m.get_weights()
    assert all(np.isfinite(w).all() for w in ws)
    m.close()


def test_growing_batch_keeps_the_optimizer_state():
    """A batch larger than the trainer was built for rebuilds the trainer; Adam's moments, its step count and the
    step counter behind the dropout seeds carry over (one optimizer for the whole of fit, trainer.py:304-309)."""
    from deepptmpred_amd.model import create_ptm_model
    from deepptmpred_amd.training import AdamW
    m = create_ptm_model(seed=5)
    m.compile(optimizer=AdamW(learning_rate=5e-4, weight_decay=1e-4), loss=None)
    x = spec.synthetic_inputs(96, 31)
    y = (x[3][:, 0] > 125).astype(np.int64)
    m.train_on_batch([a[:32] for a in x], np.eye(2)[y[:32]])
    m0, v0, s0 = m._dp.backend.trainer.get_optimizer_state()
    assert s0 == 1 and np.abs(m0).max() > 0 and np.abs(v0).max() > 0
    m.train_on_batch(x, np.eye(2)[y])                     # 96 > 32: a new, larger trainer
    m1, v1, s1 = m._dp.backend.trainer.get_optimizer_state()
    assert s1 == 2 and m._dp.steps == 2
    # second moments only decay by 0.999 per step and grow by g^2: they cannot fall below 0.999 * v0
    assert (v1 >= 0.999 * v0 - 1e-12).all() and np.abs(m1 - m0).max() > 0
    m.close()


def test_fit_with_validation_and_reference_callbacks(tmp_path):
    """The callback set of trainer.py:311-328 around the native step: validation metrics per epoch,
    early stopping on val_auc with the best weights put back, best-only checkpoints, a CSV log."""
    from deepptmpred_amd.callbacks import CSVLogger, EarlyStopping, LearningRateScheduler, ModelCheckpoint, TensorBoard
    from deepptmpred_amd.model import create_ptm_model, load_model
    from deepptmpred_amd.training import AdamW, lr_schedule
    from oracle.train_ref import TrainRef
    m = create_ptm_model(seed=5)
    x = spec.synthetic_inputs(320, 78)
    y = (x[3][:, 0] > 125).astype(np.int64)
    tr = [a[:256] for a in x]
    va = [a[256:] for a in x]
    m.compile(optimizer=AdamW(learning_rate=5e-4, weight_decay=1e-4), loss=None, metrics=["accuracy"])
    ck = ModelCheckpoint(str(tmp_path / "best_model.h5"), monitor="val_auc", save_best_only=True, mode="max")
    es = EarlyStopping(monitor="val_auc", patience=3, min_delta=0.01, mode="max", restore_best_weights=True)
    hist = m.fit(tr, np.eye(2)[y[:256]], batch_size=64, epochs=8, validation_data=(va, np.eye(2)[y[256:]]), verbose=0,
                 callbacks=[LearningRateScheduler(lr_schedule), es, ck, CSVLogger(str(tmp_path / "log.csv")),
                            TensorBoard(log_dir=str(tmp_path / "tb"))])
    n_ep = len(hist.history["loss"])
    assert 1 <= n_ep <= 8 and set(hist) >= {"loss", "lr", "val_loss", "val_accuracy", "val_auc"}
    assert hist["lr"][0] == pytest.approx(1e-4) and all(0.0 <= a <= 1.0 for a in hist["val_auc"])
    assert len(open(tmp_path / "log.csv").read().strip().splitlines()) == n_ep + 1
    # the checkpoint holds the best epoch's weights and loads back through the Keras-HDF5 reader
    assert ck.saved_epochs and ck.saved_epochs[-1] == int(np.argmax(hist["val_auc"]))
    best = load_model(str(tmp_path / "best_model.h5"))
    p_best = best.predict(va)
    # val_loss is the oracle's loss (data term + l2) of the weights at that epoch
    if ck.saved_epochs[-1] == n_ep - 1 or es.stop_training:
        ref = TrainRef(best.named_weights())
        import torch
        want = float(ref.loss(torch.as_tensor(p_best, dtype=torch.float64), y[256:]).detach())
        assert hist["val_loss"][ck.saved_epochs[-1]] == pytest.approx(want, rel=1e-4)
    if es.stop_training:        # restore_best_weights: the model scores like the checkpoint
        assert np.abs(m.predict(va) - p_best).max() < 1e-6
    best.close()
    m.close()


def _sync_bn_worker(rank, world, port, q, backend):
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import torch.distributed as dist
    from deepptmpred_amd.training import AdamW, DataParallelTrainer, DropoutRates, NativeBackend
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    if backend:
        kw = {"device_id": torch.device("cuda:0")} if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    w = spec.synthetic_weights(7)
    x = spec.synthetic_inputs(24, 9)
    y = (np.arange(24) % 5 == 0).astype(np.int64)
    dp = DataParallelTrainer(NativeBackend(w, max_batch=24), AdamW(5e-4, 1e-4), DropoutRates.off(),
                             sync_batchnorm=bool(backend))
    losses = [dp.train_on_batch(x, np.eye(2)[y], seed=3 + i) for i in range(2)]
    if backend:     # synchronised BatchNorm: one collective per level of dependent BatchNorms and direction
        assert dp.backend.sync_bn_calls == 2 * 8, dp.backend.sync_bn_calls
    # the fit loop: data set on every replica's device, shards of each 12-site batch gathered there, the loss
    # riding in the gradient all-reduce and read once per epoch
    hist = dp.fit(x, y.astype(np.int32), batch_size=12, epochs=2, shuffle=True, seed=5)
    losses += hist["loss"] + hist["accuracy"] + hist["auc"]     # the metrics of the epoch's training-mode outputs
    assert all(0.0 <= v <= 1.0 for v in hist["accuracy"] + hist["auc"])
    assert dp.backend._data is None                 # the device copy does not outlive fit
    torch.cuda.synchronize()
    if rank == 0:
        q.put((losses, {k: np.asarray(v) for k, v in dp.backend.weights().items()}))
    if backend:
        dist.barrier()
        dist.destroy_process_group()


def test_replicas_with_synchronised_batchnorm_equal_one_device():
    """With the BatchNorm column sums all-reduced inside the step, two replicas on half the batch
    each must reproduce the one-device run on the whole batch: same loss, same weights after two
    AdamW steps (dropout off; fp32 sums in a different order, hence the small tolerance).  The two
    replicas share GPU 0, which RCCL refuses, so they talk through gloo (sums staged through host
    memory); the RCCL form of the callback -- an all-reduce on the layer's own stream -- is
    exercised with a one-rank group, where it must change nothing."""
    import multiprocessing as mp
    import socket
    ctx = mp.get_context("spawn")
    results = []
    for world, backend in ((1, None), (2, "gloo"), (1, "nccl")):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        q = ctx.Queue()
        procs = [ctx.Process(target=_sync_bn_worker, args=(r, world, port, q, backend)) for r in range(world)]
        for p in procs:
            p.start()
        results.append(q.get(timeout=150))
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
    loss1, w1 = results[0]
    for (loss2, w2), label in zip(results[1:], ("two gloo replicas", "one-rank RCCL group")):
        assert np.allclose(loss1, loss2, rtol=2e-5, atol=1e-6), (label, loss1, loss2)
        worst = max(float(np.abs(w1[k] - w2[k]).max()) for k in w1)
        print("one device vs", label, ": max weight difference", worst)
        assert worst < 2e-4, label      # Adam divides by |g|: float-atomic noise on a near-zero gradient moves a weight by a fraction of lr = 5e-4


def test_trainer_profile_times_the_branch_launches():
    """dptm_trainer_profile_begin / _end: one duration per sequence-branch launch of the last step, and the step
    gives the same gradients whether its branches ran side by side or, under the profile, one after another."""
    import torch
    n = 40
    w = spec.synthetic_weights(3)
    x = spec.synthetic_inputs(n, 11)
    y = (np.random.default_rng(2).random(n) < 0.3).astype(np.int64)
    tr = _native.Trainer(spec.pack_blob(w), max_batch=64)
    d, lab = _device_batch(x, y)
    rates = (1e-4, 0.3, 0.1, 0.3)

    def step():
        tr.forward_backward(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(),
                            lab.data_ptr(), n, _opts(5, rates, y))
        torch.cuda.synchronize()
        return tr.get_grads()

    g0 = step()
    tr.profile_begin()
    g1 = step()
    ms = tr.profile_end()
    assert len(ms) == 18 and all(0.0 < v < 50.0 for v in ms.values()), ms
    assert ms["tb_bwd51"] > ms["conv_b_fwd51"]
    assert 1 <= tr.lanes_independent() <= 4      # how many of the step's four streams have a hardware queue of their own
    g2 = step()
    scale = np.abs(g0).max()
    assert np.abs(g1 - g0).max() < 1e-5 * scale and np.abs(g2 - g0).max() < 1e-5 * scale
    with pytest.raises(Exception):
        tr.profile_end()          # nothing armed any more
    tr.close()


def test_step_on_the_trainers_own_stream_equals_a_step_on_any_other():
    """dptm_trainer_stream: the stream of the step's main lane.  A step enqueued on it (no hand-over between streams)
    gives what a step enqueued on torch's current stream or on the NULL stream gives, inputs staged on that stream
    included."""
    import torch
    n = 70
    w = spec.synthetic_weights(4)
    x = spec.synthetic_inputs(n, 13)
    y = (np.random.default_rng(8).random(n) < 0.3).astype(np.int64)
    tr = _native.Trainer(spec.pack_blob(w), max_batch=128)
    handle = tr.stream()
    assert handle != 0
    own = torch.cuda.ExternalStream(handle, device=torch.device("cuda:0"))
    rates = (1e-4, 0.3, 0.1, 0.3)
    d, lab = _device_batch(x, y)
    torch.cuda.synchronize()
    grads = []
    for stream in (None, torch.cuda.current_stream(), own):
        if stream is None:
            args, sh = (d, lab), 0
        else:
            with torch.cuda.stream(stream):        # staging copies ordered by the same stream as the step
                args = ([a.clone() for a in d], lab.clone())
            sh = stream.cuda_stream
        a, l = args
        tr.forward_backward(a[0].data_ptr(), a[1].data_ptr(), a[2].data_ptr(), a[3].data_ptr(), a[4].data_ptr(),
                            l.data_ptr(), n, _opts(21, rates, y), False, sh)
        tr.apply(0.0, 0.0, 1.0, 0.0, sh)       # learning rate 0 and no decay: the weights stay
        torch.cuda.synchronize()
        grads.append(tr.get_grads())
    scale = np.abs(grads[0]).max()
    assert scale > 0
    for g in grads[1:]:
        assert np.abs(g - grads[0]).max() < 1e-5 * scale
    tr.close()


def test_fit_with_the_data_set_on_the_device_equals_the_host_loop():
    """Model.fit keeps the training set on the device (batches gathered there, losses read once per epoch); the
    batches, seeds and arithmetic are those of the train_on_batch loop over host arrays."""
    from deepptmpred_amd import create_ptm_model
    from deepptmpred_amd.training import AdamW, NativeBackend
    n, b = 200, 64
    x = spec.synthetic_inputs(n, 5)
    y = (np.random.default_rng(3).random(n) < 0.3).astype(np.int64)
    runs = []
    for resident in (True, False):
        m = create_ptm_model(None, seed=2)
        m.compile(optimizer=AdamW(learning_rate=5e-4, weight_decay=1e-4))
        if not resident:
            orig = NativeBackend.set_dataset
            NativeBackend.set_dataset = lambda self, x_, l_: False
        try:
            h = m.fit(x, y, batch_size=b, epochs=2, verbose=0)
        finally:
            if not resident:
                NativeBackend.set_dataset = orig
        # `accuracy` and `auc` as Keras logs them (trainer.py:308): over the epoch's training-mode outputs
        assert list(h.history) == ["loss", "accuracy", "auc", "lr"]
        runs.append((h.history["loss"] + h.history["accuracy"] + h.history["auc"], m.get_weights()))
        m.close()
    (la, wa), (lb, wb) = runs
    assert np.allclose(la, lb, rtol=2e-4), (la, lb)
    assert all(0.0 <= v <= 1.0 for v in la[2:])
    for a, c in zip(wa, wb):
        assert np.abs(a - c).max() < 2e-4 * max(1.0, np.abs(c).max())


def test_step_on_rows_of_a_resident_data_set_equals_the_step_on_the_gathered_batch():
    """dptm_trainer_forward_backward_rows: gather on the device + step + loss / probabilities left at device
    addresses, against dptm_trainer_forward_backward on the same rows gathered on the host.  Same kernels on the same
    values: gradients, loss and probabilities must be equal to rounding of the float atomics (1e-5 of the largest
    gradient, 2e-5 on a probability).  Second call with fewer rows; an index beyond the data set is clamped to its last row."""
    import torch
    n_data = 300
    w = spec.synthetic_weights(9)
    x = spec.synthetic_inputs(n_data, 17)
    y = (np.random.default_rng(4).random(n_data) < 0.3).astype(np.int64)
    tr = _native.Trainer(spec.pack_blob(w), max_batch=128)
    d, lab = _device_batch(x, y)
    rates = (1e-4, 0.3, 0.1, 0.3)
    rng = np.random.default_rng(6)
    for n, bad in ((100, False), (37, True)):
        rows = rng.permutation(n_data)[:n].astype(np.int64)
        sent = rows.copy()
        if bad:
            sent[5] = n_data + 1000          # clamped to the last row
            rows[5] = n_data - 1
        o = _opts(31, rates, y[rows])
        rows_dev = torch.from_numpy(sent).cuda()
        loss_dev = torch.full((1,), -1.0, device="cuda")
        proba_dev = torch.full((n, 2), -1.0, device="cuda")
        torch.cuda.synchronize()
        tr.forward_backward_rows(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(),
                                 lab.data_ptr(), n_data, rows_dev.data_ptr(), n, o, loss_dev.data_ptr(), proba_dev.data_ptr())
        torch.cuda.synchronize()
        g_rows, p_rows, l_rows = tr.get_grads(), proba_dev.cpu().numpy(), float(loss_dev.cpu()[0])
        assert np.allclose(tr.get_proba(n), p_rows, atol=0) and abs(p_rows.sum(axis=1) - 1).max() < 1e-5
        g, lg = _device_batch([a[rows] for a in x], y[rows])
        loss = tr.forward_backward(g[0].data_ptr(), g[1].data_ptr(), g[2].data_ptr(), g[3].data_ptr(), g[4].data_ptr(),
                                   lg.data_ptr(), n, o)
        torch.cuda.synchronize()
        g_ref = tr.get_grads()
        assert abs(loss - l_rows) < 1e-5 * max(1.0, abs(loss)), (loss, l_rows)
        assert np.abs(tr.get_proba(n) - p_rows).max() < 2e-5      # (column sums by float atomics: two runs differ in the last bits)
        assert np.abs(g_rows - g_ref).max() < 1e-5 * np.abs(g_ref).max()
    # the outputs are optional
    tr.forward_backward_rows(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(),
                             lab.data_ptr(), n_data, rows_dev.data_ptr(), 37, o)
    torch.cuda.synchronize()
    with pytest.raises(Exception):
        tr.forward_backward_rows(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(),
                                 lab.data_ptr(), n_data, rows_dev.data_ptr(), 129, o)       # beyond max_batch
    tr.close()


def test_gradients_of_a_repeated_batch_equal_those_of_one_period():
    """Size-independent property at the large-batch sizes of the bench (8,192 and 16,384 sites per step): with dropout
    off, a batch made of k copies of a period has the period's loss and gradients -- the class weights N / (2 N_pos),
    the mean over N * 2, every BatchNorm's batch statistics and the l2 terms are the same.  The period (4,096 sites) sits
    next to the 4,100 sites checked against the autograd oracle above; the copies run through more rounds of tiles per
    workgroup, the tile counters and the five-launch backward pass."""
    import torch
    period = 4096
    rates = (0.0, 0.0, 0.0, 0.0)
    w = spec.synthetic_weights(13)
    x = spec.synthetic_inputs(period, 41)
    y = (np.random.default_rng(12).random(period) < 0.1).astype(np.int64)
    tr = _native.Trainer(spec.pack_blob(w), max_batch=4 * period)
    results = []
    for k in (1, 2, 4):
        xs = [np.concatenate([a] * k) for a in x]
        ys = np.concatenate([y] * k)
        d, lab = _device_batch(xs, ys)
        loss = tr.forward_backward(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(),
                                   lab.data_ptr(), k * period, _opts(3, rates, ys))
        torch.cuda.synchronize()
        results.append((loss, tr.get_grads()))
        del d, lab
    (l1, g1) = results[0]
    scale = np.abs(g1).max()
    assert scale > 0 and np.isfinite(g1).all()
    for (lk, gk), k in zip(results[1:], (2, 4)):
        assert abs(lk - l1) < 2e-5 * max(1.0, abs(l1)), (k, lk, l1)
        err = np.abs(gk - g1).max() / scale
        print("copies", k, "max gradient difference / largest gradient %.2e" % err)
        assert err < 2e-4, (k, err)
    tr.close()


_TILE_FORM_SCRIPT = """
import sys
import numpy as np
import torch
sys.path.insert(0, {root!r})
from deepptmpred_amd import _native, spec
n = {n}
w = spec.synthetic_weights(3)
x = spec.synthetic_inputs(n, 11)
y = (np.random.default_rng(2).random(n) < 0.3).astype(np.int64)
dev = torch.device("cuda:0")
d = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in x]
lab = torch.from_numpy(y.astype(np.int32)).to(dev)
tr = _native.Trainer(spec.pack_blob(w), max_batch=max(64, n))
o = _native.TrainOpts()
o.seed = 5
o.rate_embed, o.rate_conv, o.rate_tb, o.rate_dense = 1e-4, 0.3, 0.1, 0.3
o.n_global = n; o.n_pos_global = int(y.sum()); o.l2 = 1e-4; o.l2_scale = 1.0
for _ in range({steps}):       # (the same inputs and seed every time: the same masks, the same gradients)
    loss = tr.forward_backward(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(), lab.data_ptr(), n, o)
    torch.cuda.synchronize()
np.save({out!r}, np.concatenate([tr.get_grads(), np.float32([loss])]))
"""


def _grads_in_a_process(tmp_path, tag, env_extra, n, steps=1):
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / f"grads_{tag}.npy")
    r = subprocess.run([sys.executable, "-c", _TILE_FORM_SCRIPT.format(root=root, n=n, out=out, steps=steps)],
                       env=dict(os.environ, **env_extra), capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    return np.load(out)


def test_one_site_tiles_with_and_without_shared_row_blocks_agree(tmp_path):
    """Batches of up to 85 sites run one site per workgroup with TWO wavefronts per 32-row block (row products cut by
    output tile or contraction half, a head per lane).  A second process with that form switched off
    (DPTM_TRAIN_PAIR=0: one wavefront per row block, the other two only share the weight-gradient units) must leave
    the same loss and gradients -- dropout on, so the masks of both forms are the same masks."""
    got = {form: _grads_in_a_process(tmp_path, "pair" + form, {"DPTM_TRAIN_PAIR": form}, 40) for form in ("1", "0")}
    a, b = got["1"], got["0"]
    assert a.shape == b.shape and np.isfinite(a).all()
    assert abs(a[-1] - b[-1]) < 1e-6 * max(1.0, abs(b[-1]))                      # loss
    assert np.abs(a[:-1] - b[:-1]).max() < 1e-5 * np.abs(b[:-1]).max()           # every gradient, flat


@pytest.mark.parametrize("n", [40, 300])
def test_graph_replay_of_the_step_equals_eager_issue(tmp_path, n):
    """DPTM_TRAIN_GRAPH=1 (opt-in): the second call of a configuration records the step's launches -- all streams, the
    event hand-overs between them -- and the third replays them; the step's scalars then come from a launch of their
    own in front of the graph.  Loss and gradients of the replayed step equal those of the step issued eagerly."""
    eager = _grads_in_a_process(tmp_path, f"eager{n}", {"DPTM_TRAIN_GRAPH": "0"}, n, steps=3)
    graph = _grads_in_a_process(tmp_path, f"graph{n}", {"DPTM_TRAIN_GRAPH": "1"}, n, steps=3)
    assert eager.shape == graph.shape and np.isfinite(graph).all()
    assert abs(eager[-1] - graph[-1]) < 1e-6 * max(1.0, abs(eager[-1]))
    assert np.abs(eager[:-1] - graph[:-1]).max() < 1e-5 * np.abs(eager[:-1]).max()

"""Drop-in for the Keras model object the reference's predictors hold.

The reference builds its model with ``create_ptm_model(config)``
(pred/train_PTM/model.py:143-210), restores it with
``load_model(path, custom_objects={...})`` (predict.py:569-577, case.py:468-476,
trainer.py:345-353) and scores with ``model.predict([x21, x33, x51, struct, esm])``
(predict.py:598-599, case.py:495, trainer.py:62).  ``Model`` keeps those call
signatures; the arithmetic runs in libdptm.so on an MI355X.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import _native
from .spec import (DEFAULT_SPEC, LETTER_CODES, PAD_CODE, ModelSpec, keras_initial_weights, load_flat, pack_blob,
                   save_flat, tensor_table, unpack_blob)


def _as_tokens(a, width: int, name: str) -> np.ndarray:
    """Keras feeds the window arrays to ``Embedding`` (model.py:157), which casts to
    int32; values must be valid codes 0..20."""
    a = np.asarray(a)
    if a.ndim != 2 or a.shape[1] != width:
        raise ValueError(f"{name}: expected shape [N,{width}], got {a.shape}")
    if a.dtype.kind == "f":
        a = a.astype(np.int32)  # same truncation as tf.cast(float -> int32)
    elif a.dtype.kind not in "iu":
        raise ValueError(f"{name}: expected an integer or float array, got dtype {a.dtype}")
    if a.size and (a.min() < 0 or a.max() > PAD_CODE):
        raise ValueError(f"{name}: residue codes must lie in 0..{PAD_CODE}")
    return np.ascontiguousarray(a, dtype=np.uint8)


def _as_features(a, width: int, name: str) -> np.ndarray:
    a = np.asarray(a)
    if a.ndim != 2 or a.shape[1] != width:
        raise ValueError(f"{name}: expected shape [N,{width}], got {a.shape}")
    if a.dtype.kind not in "fiu":
        raise ValueError(f"{name}: expected a numeric array, got dtype {a.dtype}")
    return np.ascontiguousarray(a, dtype=np.float32)  # keras casts inputs to float32


def encode_sequence(sequence: str) -> np.ndarray:
    """Residue letters -> codes with the table of predict.py:549-553 (unknown -> 20)."""
    return np.fromiter((LETTER_CODES.get(c, PAD_CODE) for c in sequence), dtype=np.uint8,
                       count=len(sequence))


class Model:
    """Holds one weight set; ``predict`` mirrors ``keras.Model.predict``."""

    def __init__(self, weights: Dict[str, np.ndarray], spec: ModelSpec = DEFAULT_SPEC,
                 device: int = 0):
        self.spec = spec
        self.device = device
        self._weights = {k: np.asarray(v, dtype=np.float32) for k, v in weights.items()}
        self._blob = pack_blob(self._weights, spec)
        self._scorer: Optional[_native.Scorer] = None

    # -- native handle
    @property
    def scorer(self) -> _native.Scorer:
        if self._scorer is None:
            self._scorer = _native.Scorer(self._blob, self.spec, self.device)
        return self._scorer

    def close(self) -> None:
        if self._scorer is not None:
            self._scorer.close()
            self._scorer = None

    # -- keras-like surface
    def predict(self, x, batch_size: Optional[int] = 32, verbose="auto", **_ignored) -> np.ndarray:
        """``model.predict([x21, x33, x51, struct, esm])`` -> float32 ``[N,2]``.

        Every site is scored independently, so ``batch_size`` does not change the
        result; it is accepted for signature compatibility (Keras default 32,
        trainer.py:62 passes 128) and the library picks its own pass size."""
        if not isinstance(x, (list, tuple)) or len(x) != 5:
            raise ValueError("predict expects [x21, x33, x51, struct, esm] (model.py:210 input order)")
        ws = self.spec.window_sizes
        toks = [_as_tokens(x[i], ws[i], f"seq_input_{ws[i]}") for i in range(3)]
        st = _as_features(x[3], self.spec.struct_feature_dim, "struct_input")
        esm = _as_features(x[4], self.spec.esm_dim, "esm_input")
        n = len(toks[0])
        if any(len(a) != n for a in toks[1:] + [st, esm]):
            raise ValueError("predict: all five inputs must have the same number of rows")
        if n == 0:
            return np.zeros((0, self.spec.n_classes), dtype=np.float32)
        if getattr(self, "_dirty", False):      # weights were trained since the last scorer was built
            self._pull_trained_weights()
            self._dirty = False
        return self.scorer.score(toks[0], toks[1], toks[2], st, esm)

    def __call__(self, x, training=False):
        return self.predict(x)

    def predict_protein(self, sequence: str, positions: Sequence[int], struct_features,
                        esm_table=None) -> np.ndarray:
        """Fused ``prepare_protein_data`` + ``predict`` (predict.py:462-485, 598-599):
        windows and ESM rows are assembled on the GPU from the raw sequence."""
        codes = encode_sequence(sequence)
        pos = np.ascontiguousarray(np.asarray(positions, dtype=np.int64))
        if pos.ndim != 1:
            raise ValueError("positions must be one-dimensional")
        if pos.size and (pos.min() < 1 or pos.max() > len(sequence)):
            raise IndexError("position outside the sequence (predict.py:535 raises IndexError)")
        st = _as_features(struct_features, self.spec.struct_feature_dim, "struct_features")
        if len(st) != len(pos):
            raise ValueError("struct_features and positions differ in length")
        table = None
        if esm_table is not None:
            table = _as_features(esm_table, self.spec.esm_dim, "esm_table")
        if pos.size == 0:
            return np.zeros((0, self.spec.n_classes), dtype=np.float32)
        return self.scorer.score_protein(codes, table, pos.astype(np.int32), st)

    def predict_proteins(self, proteins) -> List[np.ndarray]:
        """Many proteins in ONE native call (``dptm_score_proteins``), instead of the per-protein loop
        around ``predict_ptm_sites`` (predict.py:584-612).  ``proteins`` is a sequence of
        ``(sequence, positions, struct_features, esm_table_or_None)`` with the meaning of
        ``predict_protein``; returns one float32 ``[n_i, 2]`` array per protein, equal to what
        ``predict_protein`` gives for it."""
        f = self.spec.struct_feature_dim
        codes, pos_all, prot_all, st_all, tables, counts = [], [], [], [], [], []
        seq_off, esm_off = [0], [0]
        for k, (sequence, positions, struct_features, esm_table) in enumerate(proteins):
            pos = np.ascontiguousarray(np.asarray(positions, dtype=np.int64)).reshape(-1)
            if pos.size and (pos.min() < 1 or pos.max() > len(sequence)):
                raise IndexError(f"protein {k}: position outside the sequence (predict.py:535 raises IndexError)")
            st = _as_features(np.asarray(struct_features).reshape(-1, f), f, f"protein {k}: struct_features")
            if len(st) != len(pos):
                raise ValueError(f"protein {k}: struct_features and positions differ in length")
            codes.append(encode_sequence(sequence))
            seq_off.append(seq_off[-1] + len(sequence))
            rows = 0
            if esm_table is not None and len(esm_table):
                t = _as_features(esm_table, self.spec.esm_dim, f"protein {k}: esm_table")
                tables.append(t)
                rows = len(t)
            esm_off.append(esm_off[-1] + rows)
            pos_all.append(pos.astype(np.int32))
            prot_all.append(np.full(len(pos), k, dtype=np.int32))
            st_all.append(st)
            counts.append(len(pos))
        n = int(sum(counts))
        if n == 0:
            return [np.zeros((0, self.spec.n_classes), dtype=np.float32) for _ in counts]
        if getattr(self, "_dirty", False):
            self._pull_trained_weights()
            self._dirty = False
        cat = lambda parts, dt, shape: (np.ascontiguousarray(np.concatenate(parts), dtype=dt) if parts
                                        else np.zeros(shape, dtype=dt))
        out = self.scorer.score_proteins(cat(codes, np.uint8, (0,)), np.asarray(seq_off, dtype=np.int64),
                                         cat(tables, np.float32, (0, self.spec.esm_dim)) if tables else None,
                                         np.asarray(esm_off, dtype=np.int64), cat(pos_all, np.int32, (0,)),
                                         cat(prot_all, np.int32, (0,)), cat(st_all, np.float32, (0, f)))
        return [a for a in np.split(out, np.cumsum(counts)[:-1])]

    # -- training (trainer.py:304-339)
    def compile(self, optimizer=None, loss=None, metrics=None, **_ignored) -> None:
        """``model.compile(optimizer=AdamW(...), loss=..., metrics=...)`` (trainer.py:304-309).
        Only the optimizer's ``learning_rate`` and ``weight_decay`` are used; the loss is
        always the class-balanced cross-entropy of trainer.py:36-54, built into the kernels."""
        from .training import AdamW
        lr = getattr(optimizer, "learning_rate", 5e-4) if optimizer is not None else 5e-4
        wd = getattr(optimizer, "weight_decay", 1e-4) if optimizer is not None else 1e-4
        self._optimizer = AdamW(float(lr), float(wd))
        self._dp = None

    def _trainer(self, batch_size: int):
        from .training import DataParallelTrainer, DropoutRates, NativeBackend
        if getattr(self, "_optimizer", None) is None:
            raise RuntimeError("call compile() before training")
        if getattr(self, "_dp", None) is None or self._dp_batch < batch_size:
            # A larger batch needs a larger trainer.  The reference keeps one optimizer for all of fit()
            # (trainer.py:304-309): Adam's moments, the step count behind its bias correction and the step counter
            # the dropout seeds follow move over to the new trainer.
            carried = None
            if getattr(self, "_dp", None) is not None:
                carried = (self._dp.backend.trainer.get_optimizer_state(), self._dp.steps)
                self._pull_trained_weights()
                self._dp.backend.close()
            backend = NativeBackend(self._weights, self.spec, self.device, max_batch=batch_size)
            self._dp = DataParallelTrainer(backend, self._optimizer,
                                           DropoutRates(self.spec.weight_decay, self.spec.dropout_rate, 0.1,
                                                        self.spec.dropout_rate), l2=self.spec.weight_decay)
            if carried is not None:
                (m, v, step), steps = carried
                backend.trainer.set_optimizer_state(m, v, step)
                self._dp.steps = steps
            self._dp_batch = batch_size
        return self._dp

    def _pull_trained_weights(self) -> None:
        if getattr(self, "_dp", None) is not None:
            self._weights = {k: np.asarray(v, dtype=np.float32) for k, v in self._dp.backend.weights().items()}
            self._blob = pack_blob(self._weights, self.spec)
            self.close()

    def train_on_batch(self, x, y) -> float:
        n = len(np.asarray(y))
        loss = self._trainer(n).train_on_batch(x, y)
        self._dirty = True
        return loss

    def fit(self, x, y, batch_size: int = 64, epochs: int = 1, validation_data=None, callbacks=None,
            verbose=1, shuffle: bool = True, **_ignored):
        """``model.fit`` as trainer.py:330-339 calls it: one pass over the data per epoch through the
        native training step, then -- with ``validation_data`` -- ``val_loss`` (data term of the loss +
        the l2 regularisers), ``val_accuracy`` and ``val_auc`` from the scoring path, then the callbacks
        (``deepptmpred_amd.callbacks``: LearningRateScheduler, EarlyStopping, ModelCheckpoint, CSVLogger;
        a plain ``schedule(epoch, lr)`` callable is taken as a learning-rate schedule).  Returns a
        ``History`` (``.history['loss']`` ...)."""
        from .callbacks import Callback, History, LearningRateScheduler, balanced_bce, roc_auc
        from .training import labels_from_targets
        cbs = []
        for cb in callbacks or []:
            if isinstance(cb, Callback):
                cbs.append(cb)
            elif callable(getattr(cb, "schedule", None)):
                cbs.append(LearningRateScheduler(cb.schedule))
            elif callable(cb):
                cbs.append(LearningRateScheduler(cb))
        for cb in cbs:
            cb.set_model(self)
            cb.on_train_begin()
        trainer = self._trainer(batch_size)
        hist = History()
        lr = trainer.optimizer.learning_rate
        y_labels = labels_from_targets(y)         # one array for all epochs: the device copy of the data set is kept
        try:
            for epoch in range(int(epochs)):
                for cb in cbs:
                    lr = cb.on_epoch_begin(epoch, lr)
                trainer = self._trainer(batch_size)       # a callback may have replaced the weights
                h1 = trainer.fit(x, y_labels, batch_size, 1, shuffle, lambda _e, _lr, _v=lr: _v, seed=epoch, verbose=0,
                                 keep_resident=True)
                self._dirty = True
                logs = {"loss": h1["loss"][0]}
                for k in ("accuracy", "auc"):           # the compiled metrics of trainer.py:308, over the epoch's batches
                    if k in h1:
                        logs[k] = h1[k][0]
                if validation_data is not None:
                    vx, vy = validation_data
                    vl = labels_from_targets(vy)
                    p = self.predict(vx)
                    reg = sum(float(self.spec.weight_decay) * float(np.sum(np.square(self._weights[n].astype(np.float64))))
                              for n in self._l2_names())
                    logs["val_loss"] = balanced_bce(p, vl) + reg
                    logs["val_accuracy"] = float(np.mean(np.argmax(p, axis=1) == vl))
                    logs["val_auc"] = roc_auc(vl, p[:, 1])
                logs["lr"] = lr
                for k, v in logs.items():
                    hist.setdefault(k, []).append(v)
                if verbose:
                    print("Epoch %d/%d - " % (epoch + 1, epochs) + " - ".join("%s: %.4g" % kv for kv in logs.items()))
                for cb in cbs:
                    cb.on_epoch_end(epoch, logs)
                if any(cb.stop_training for cb in cbs):
                    break
        finally:
            # the device copy of the training set never outlives fit(), whatever ends it (a callback that raises,
            # an interrupt): a later fit() must not mistake it for its own data
            for holder in (trainer, getattr(self, "_dp", None)):
                if holder is not None and hasattr(holder.backend, "drop_dataset"):
                    holder.backend.drop_dataset()
        for cb in cbs:
            cb.on_train_end()
        return hist

    def _l2_names(self) -> List[str]:
        """Kernels that carry the l2(weight_decay) regulariser (model.py:130-138, 200, 204)."""
        return [n for n, _ in tensor_table(self.spec)
                if n.endswith((".sc_k", ".c3a_k", ".c3b_k")) or n in ("t1.k", "t2.k")]

    def get_weights(self) -> List[np.ndarray]:
        """``model.get_weights()``: every array of the model in Keras' ``model.weights`` order for the
        graph of model.py:210 (depth-sorted layers: the three ``[21, 64]`` embedding tables first).
        ``named_weights()`` gives the same arrays by canonical name."""
        if getattr(self, "_dirty", False):
            self._pull_trained_weights()
            self._dirty = False
        return [self._weights[n].copy() for n in self._keras_order()]

    def _keras_order(self) -> List[str]:
        from .keras_h5 import keras_weight_order
        return keras_weight_order(self.spec)

    def set_weights(self, arrays: Sequence[np.ndarray]) -> None:
        """``model.set_weights(list)`` with the list in the order ``get_weights`` returns."""
        shapes = dict(tensor_table(self.spec))
        table = [(n, shapes[n]) for n in self._keras_order()]
        if len(arrays) != len(table):
            raise ValueError(f"set_weights: expected {len(table)} arrays, got {len(arrays)}")
        new = {}
        for (name, shape), a in zip(table, arrays):
            a = np.asarray(a, dtype=np.float32)
            if a.shape != tuple(shape):
                raise ValueError(f"set_weights: {name} has shape {a.shape}, expected {shape}")
            new[name] = a
        self._weights = new
        self._blob = pack_blob(new, self.spec)
        self.close()
        # a live trainer still holds the old weights (and their Adam moments): drop it, the next training
        # call starts a fresh one from the weights just set
        if getattr(self, "_dp", None) is not None:
            self._dp.backend.close()
            self._dp = None
        self._dirty = False

    def named_weights(self) -> Dict[str, np.ndarray]:
        """name -> array in the canonical order of ``spec.tensor_table`` (the order of the C ABI's blob)."""
        if getattr(self, "_dirty", False):
            self._pull_trained_weights()
            self._dirty = False
        return {n: self._weights[n] for n, _ in tensor_table(self.spec)}

    def save(self, filepath: str) -> None:
        """``model.save(path)``: a Keras-2 HDF5 weight tree for ``*.h5`` / ``*.hdf5`` paths
        (trainer.py:321-326 checkpoints), the flat ``.dptm`` container otherwise."""
        if getattr(self, "_dirty", False):
            self._pull_trained_weights()
            self._dirty = False
        path = os.fspath(filepath)
        if path.endswith((".h5", ".hdf5", ".keras.h5")):
            from .keras_h5 import write_keras_h5
            write_keras_h5(path, self._weights, self.spec)
        else:
            save_flat(path, self._weights, self.spec)

    save_weights = save

    def count_params(self) -> int:
        return int(self._blob.size)


def load_model(filepath, custom_objects=None, compile=True, device: int = 0) -> Model:  # noqa: A002
    """Stand-in for ``tensorflow.keras.models.load_model`` (predict.py:569-577).

    ``custom_objects`` and ``compile`` are accepted and ignored: the custom layers
    (PositionEmbedding, MultiHeadSelfAttention, TransformerBlock) are built into the
    native kernels and the loss is not needed for scoring.  Reads the repo's flat
    ``.dptm`` container; Keras ``.h5`` files go through ``deepptmpred_amd.keras_h5``."""
    path = os.fspath(filepath)
    if not os.path.exists(path):
        raise OSError(f"No file or directory found at {path}")
    with open(path, "rb") as fh:
        head = fh.read(8)
    if head.startswith(b"\x89HDF\r\n\x1a\n"):
        from .keras_h5 import read_keras_h5  # deferred: pure-python HDF5 subset reader
        weights = read_keras_h5(path)
        return Model(weights, DEFAULT_SPEC, device)
    weights, spec, _ = load_flat(path)
    return Model(weights, spec, device)


def create_ptm_model(config=None, seed: int = 0, device: int = 0) -> Model:
    """Stand-in for ``create_ptm_model(config)`` (model.py:143): a freshly initialised model in the
    state Keras' default initialisers give it (``spec.keras_initial_weights``: Glorot kernels, zero
    biases, identity BatchNormalization, head bias [0.1, 1.0]).  ``config`` may carry the reference's
    attribute names; only the reference dimensions are supported."""
    spec = DEFAULT_SPEC
    if config is not None:
        for attr in ("embed_dim", "num_heads", "ff_dim", "esm_dim", "struct_feature_dim", "max_features"):
            if hasattr(config, attr) and getattr(config, attr) != getattr(spec, attr):
                raise ValueError(f"create_ptm_model: {attr}={getattr(config, attr)} is not supported")
        if hasattr(config, "window_sizes") and tuple(config.window_sizes) != spec.window_sizes:
            raise ValueError("create_ptm_model: window_sizes must be [21, 33, 51]")
    return Model(keras_initial_weights(seed, spec), spec, device)

"""The Keras callbacks the reference's training loop passes to ``model.fit``
(pred/train_PTM/trainer.py:311-328), same class names and constructor arguments, driven by
``deepptmpred_amd.Model.fit`` once per epoch.  Only what that loop uses is implemented."""
from __future__ import annotations

import csv
import os
from typing import Callable, Dict, List, Optional

import numpy as np


class Callback:
    model = None
    stop_training = False

    def set_model(self, model) -> None:
        self.model = model

    def on_train_begin(self) -> None:
        pass

    def on_epoch_begin(self, epoch: int, lr: float) -> float:
        return lr

    def on_epoch_end(self, epoch: int, logs: Dict[str, float]) -> None:
        pass

    def on_train_end(self) -> None:
        pass


class LearningRateScheduler(Callback):
    """``LearningRateScheduler(schedule)`` with ``schedule(epoch, lr) -> lr`` (trainer.py:56-59, 312)."""

    def __init__(self, schedule: Callable[[int, float], float], verbose: int = 0):
        self.schedule = schedule
        self.verbose = verbose

    def on_epoch_begin(self, epoch: int, lr: float) -> float:
        return float(self.schedule(epoch, lr))


def _improved(mode: str, monitor: str):
    if mode == "auto":
        mode = "max" if ("acc" in monitor or "auc" in monitor) else "min"
    if mode == "max":
        return lambda cur, best, delta: cur - delta > best, -np.inf
    return lambda cur, best, delta: cur + delta < best, np.inf


class EarlyStopping(Callback):
    """Keras semantics (trainer.py:313-319): stop after ``patience`` epochs without an improvement of
    more than ``min_delta`` in ``monitor``; optionally put the best epoch's weights back."""

    def __init__(self, monitor: str = "val_loss", min_delta: float = 0.0, patience: int = 0, verbose: int = 0,
                 mode: str = "auto", restore_best_weights: bool = False, **_ignored):
        self.monitor, self.min_delta, self.patience = monitor, abs(float(min_delta)), int(patience)
        self.restore_best_weights, self.verbose = restore_best_weights, verbose
        self._better, self._start = _improved(mode, monitor)
        self.on_train_begin()

    def on_train_begin(self) -> None:
        self.best, self.wait, self.stopped_epoch, self.best_epoch = self._start, 0, 0, 0
        self.best_weights = None
        self.stop_training = False

    def on_epoch_end(self, epoch: int, logs: Dict[str, float]) -> None:
        cur = logs.get(self.monitor)
        if cur is None:
            return
        if self._better(cur, self.best, self.min_delta):
            self.best, self.wait, self.best_epoch = cur, 0, epoch
            if self.restore_best_weights and self.model is not None:
                self.best_weights = self.model.get_weights()
            return
        self.wait += 1
        if self.wait >= self.patience and epoch > 0:
            self.stopped_epoch = epoch
            self.stop_training = True
            if self.restore_best_weights and self.best_weights is not None and self.model is not None:
                self.model.set_weights(self.best_weights)


class ModelCheckpoint(Callback):
    """``ModelCheckpoint(filepath, monitor, save_best_only, mode)`` (trainer.py:320-325): ``model.save``
    every epoch, or only when ``monitor`` improved."""

    def __init__(self, filepath: str, monitor: str = "val_loss", verbose: int = 0, save_best_only: bool = False,
                 mode: str = "auto", **_ignored):
        self.filepath, self.monitor, self.save_best_only, self.verbose = filepath, monitor, save_best_only, verbose
        self._better, self.best = _improved(mode, monitor)
        self.saved_epochs: List[int] = []

    def on_epoch_end(self, epoch: int, logs: Dict[str, float]) -> None:
        if self.model is None:
            return
        if self.save_best_only:
            cur = logs.get(self.monitor)
            if cur is None or not self._better(cur, self.best, 0.0):
                return
            self.best = cur
        path = self.filepath.format(epoch=epoch + 1, **logs)
        os.makedirs(os.path.dirname(os.path.abspath(path)) or ".", exist_ok=True)
        self.model.save(path)
        self.saved_epochs.append(epoch)


class CSVLogger(Callback):
    """``CSVLogger(filename)`` (trainer.py:326): one row per epoch, columns ``epoch`` + the sorted log keys."""

    def __init__(self, filename: str, separator: str = ",", append: bool = False):
        self.filename, self.separator, self.append = filename, separator, append
        self._keys: Optional[List[str]] = None

    def on_train_begin(self) -> None:
        self._keys = None
        os.makedirs(os.path.dirname(os.path.abspath(self.filename)) or ".", exist_ok=True)
        if not self.append and os.path.exists(self.filename):
            os.remove(self.filename)

    def on_epoch_end(self, epoch: int, logs: Dict[str, float]) -> None:
        new = self._keys is None
        if new:
            self._keys = sorted(logs)
        with open(self.filename, "a", newline="") as fh:
            w = csv.writer(fh, delimiter=self.separator)
            if new and fh.tell() == 0:
                w.writerow(["epoch"] + self._keys)
            w.writerow([epoch] + [logs.get(k, "") for k in self._keys])


class TensorBoard(Callback):
    """Accepted for signature compatibility (trainer.py:327); writes nothing."""

    def __init__(self, log_dir: str = "logs", **_ignored):
        self.log_dir = log_dir


class History(dict):
    """What ``fit`` returns: ``history.history['loss']`` as in Keras; also usable as the dict itself."""

    @property
    def history(self) -> "History":
        return self


def roc_auc(labels: np.ndarray, scores: np.ndarray) -> float:
    """Area under the ROC curve by ranks (ties averaged).  Keras' ``AUC`` metric (trainer.py:308)
    approximates the same integral with 200 thresholds."""
    labels = np.asarray(labels).astype(bool)
    scores = np.asarray(scores, dtype=np.float64)
    n_pos, n_neg = int(labels.sum()), int((~labels).sum())
    if n_pos == 0 or n_neg == 0:
        return float("nan")
    # average ranks (1-based) with ties sharing the mean of their positions
    _, inverse, counts = np.unique(scores, return_inverse=True, return_counts=True)
    last = np.cumsum(counts).astype(np.float64)             # rank of the last member of each group of equal scores
    r = (last - 0.5 * (counts - 1))[inverse]
    return float((r[labels].sum() - n_pos * (n_pos + 1) / 2.0) / (n_pos * n_neg))


def balanced_bce(proba: np.ndarray, labels: np.ndarray) -> float:
    """The data term of the reference's loss (trainer.py:36-54) on a whole set: class weights
    N/(2 N_pos) and N/(2 N_neg), Keras binary cross-entropy over both softmax columns, mean over N*2."""
    p = np.asarray(proba, dtype=np.float64)
    y = np.asarray(labels).astype(np.int64)
    n, n_pos = len(y), int(y.sum())
    eps = 1e-7
    w = np.where(y == 1, n / (2.0 * n_pos + eps), n / (2.0 * (n - n_pos) + eps))
    onehot = np.eye(2)[y]
    pc = np.clip(p, eps, 1.0 - eps)
    bce = -(onehot * np.log(pc + eps) + (1.0 - onehot) * np.log(1.0 - pc + eps))
    return float((w[:, None] * bce).mean())

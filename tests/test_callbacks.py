"""Keras-named callbacks and metrics of the fit loop (host logic, no GPU)."""
import csv

import numpy as np
import pytest

from deepptmpred_amd.callbacks import (CSVLogger, EarlyStopping, History, LearningRateScheduler, ModelCheckpoint,
                                       balanced_bce, roc_auc)
from deepptmpred_amd.training import lr_schedule


class FakeModel:
    def __init__(self):
        self.w = [np.zeros(2)]
        self.saved = []

    def get_weights(self):
        return [a.copy() for a in self.w]

    def set_weights(self, w):
        self.w = [a.copy() for a in w]

    def save(self, path):
        self.saved.append(path)


def test_early_stopping_like_trainer_py():
    """monitor val_auc, mode max, min_delta 0.01, patience 2, restore_best_weights (trainer.py:313-319)"""
    m = FakeModel()
    es = EarlyStopping(monitor="val_auc", patience=2, min_delta=0.01, mode="max", restore_best_weights=True)
    es.set_model(m)
    es.on_train_begin()
    aucs = [0.70, 0.80, 0.805, 0.79, 0.90]
    stopped = None
    for ep, a in enumerate(aucs):
        m.w = [np.full(2, float(ep))]
        es.on_epoch_end(ep, {"val_auc": a})
        if es.stop_training:
            stopped = ep
            break
    # 0.805 is not an improvement by more than 0.01 over 0.80; two such epochs stop the run at epoch 3
    assert stopped == 3 and es.best == 0.80 and es.best_epoch == 1
    assert np.array_equal(m.w[0], [1.0, 1.0])          # weights of the best epoch are back


def test_early_stopping_min_mode_and_missing_monitor():
    es = EarlyStopping(monitor="val_loss", patience=1)
    es.on_epoch_end(0, {"val_loss": 1.0})
    es.on_epoch_end(1, {"loss": 0.5})                   # monitor absent: ignored
    assert not es.stop_training
    es.on_epoch_end(2, {"val_loss": 1.5})
    assert es.stop_training and es.stopped_epoch == 2


def test_model_checkpoint_and_csv_logger(tmp_path):
    m = FakeModel()
    ck = ModelCheckpoint(str(tmp_path / "best.h5"), monitor="val_auc", save_best_only=True, mode="max")
    ck.set_model(m)
    log = CSVLogger(str(tmp_path / "sub" / "log.csv"))
    log.on_train_begin()
    for ep, a in enumerate([0.6, 0.5, 0.7]):
        logs = {"loss": 1.0 / (ep + 1), "val_auc": a}
        ck.on_epoch_end(ep, logs)
        log.on_epoch_end(ep, logs)
    assert ck.saved_epochs == [0, 2] and len(m.saved) == 2
    rows = list(csv.reader(open(tmp_path / "sub" / "log.csv")))
    assert rows[0] == ["epoch", "loss", "val_auc"] and len(rows) == 4 and rows[3][0] == "2"


def test_lr_scheduler_matches_reference_schedule():
    cb = LearningRateScheduler(lambda e, lr: lr_schedule(e, lr, 5e-4, 5))
    lr = 5e-4
    seen = []
    for e in range(8):
        lr = cb.on_epoch_begin(e, lr)
        seen.append(lr)
    assert seen[:5] == pytest.approx([1e-4, 2e-4, 3e-4, 4e-4, 5e-4])       # linear warm-up (trainer.py:56-59)
    assert seen[5] == pytest.approx(5e-4 * 0.95) and seen[7] == pytest.approx(5e-4 * 0.95 ** 3)


def test_roc_auc_and_loss_known_answers():
    y = np.array([0, 0, 1, 1])
    assert roc_auc(y, np.array([0.1, 0.4, 0.35, 0.8])) == pytest.approx(0.75)
    assert roc_auc(y, np.array([0.1, 0.2, 0.8, 0.9])) == 1.0
    assert roc_auc(y, np.array([0.5, 0.5, 0.5, 0.5])) == pytest.approx(0.5)     # ties count half
    assert np.isnan(roc_auc(np.zeros(3), np.arange(3)))
    # balanced set, confident and right: loss -> -2 log(p) / 2 per element pair with weights 1
    p = np.array([[0.9, 0.1], [0.1, 0.9]])
    assert balanced_bce(p, np.array([0, 1])) == pytest.approx(-np.log(0.9 + 1e-7), rel=1e-5)
    # a 1:3 imbalance weights the positive three times the negatives
    p = np.tile([[0.5, 0.5]], (4, 1))
    assert balanced_bce(p, np.array([1, 0, 0, 0])) == pytest.approx(np.mean([2.0, 2 / 3, 2 / 3, 2 / 3]) * -np.log(0.5 + 1e-7), rel=1e-5)


def test_history_behaves_like_keras_and_dict():
    h = History()
    h.setdefault("loss", []).append(0.3)
    assert h.history["loss"] == [0.3] and h["loss"] == [0.3]

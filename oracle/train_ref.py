"""CPU restatement of one training step (forward in training mode, loss, backward,
AdamW) on torch autograd.  TEST INFRASTRUCTURE ONLY (same import rule as ptm_oracle.py).

PARITY UNPINNED: the reference's training step is ``model.fit`` on a Keras graph
(pred/train_PTM/trainer.py:304-339) and cannot be run here.  Restated from the text:

* graph in training mode (model.py:143-210): BatchNormalization uses the batch mean and
  the biased batch variance and moves its statistics with momentum 0.99; Dropout(0.3) /
  TransformerBlock dropout 0.1 (model.py:94,108-109) / SpatialDropout1D(dropout_rate) in
  the conv block (model.py:136) / SpatialDropout1D(weight_decay=1e-4) after the position
  embedding (model.py:159).  Keras' dropout random stream cannot be reproduced, so masks
  come from a stateless 32-bit hash of (seed, stream id, element index) that the HIP
  kernels implement identically; rate 0 disables a site.
* loss (trainer.py:36-54) under this interpretation of the Keras call: labels =
  argmax(y_true); N, N_pos over the (global) batch; w_pos = N/(2 N_pos + eps), w_neg =
  N/(2 N_neg + eps), eps = 1e-7; per-sample weight broadcast over both softmax columns
  times keras.backend.binary_crossentropy(y_onehot, y_pred) (probabilities clipped to
  [eps, 1-eps], logs taken of value + eps); mean over N*2 elements; plus the l2(1e-4)
  kernel regularisers of the nine conv kernels and the two trunk Dense kernels
  (model.py:130-138, 200, 204).
* optimiser (trainer.py:305-306): tensorflow-addons AdamW = decoupled decay
  ``var -= weight_decay * var`` (not scaled by the learning rate) followed by Keras Adam
  ``var -= lr * sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + 1e-7)``, b1 0.9, b2 0.999, applied
  to every trainable variable.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

WINDOWS = (21, 33, 51)
HEADS = 4
BN_MOMENTUM = 0.99
K_EPS = 1e-7
L2 = 1e-4
L2_TENSORS = tuple(f"b{w}.{n}" for w in WINDOWS for n in ("sc_k", "c3a_k", "c3b_k")) + ("t1.k", "t2.k")

# dropout stream ids (shared with the HIP kernels, dptm_train.hip)
def stream_id(branch: int, site: int) -> int:
    """branch 0..2 -> sites 0 embed-spatial, 1 conv-spatial, 2 tb1.att, 3 tb1.ffn, 4 tb2.att,
    5 tb2.ffn; branch 3 -> 0 struct, 1 esm, 2 merged, 3 trunk1."""
    return branch * 8 + site


def hash_u32(seed: int, stream: int, idx: np.ndarray) -> np.ndarray:
    """lowbias32 of the element index mixed with seed and stream; uint32 arithmetic."""
    x = idx.astype(np.uint64)
    x = (x ^ ((np.uint64(seed) * np.uint64(0x9E3779B9)) & np.uint64(0xFFFFFFFF))) & np.uint64(0xFFFFFFFF)
    x = (x + ((np.uint64(stream) * np.uint64(0x85EBCA6B)) & np.uint64(0xFFFFFFFF))) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    return x.astype(np.uint32)


def keep_mask(seed: int, stream: int, n_elems: int, rate: float) -> np.ndarray:
    """1.0/(1-rate) where the element is kept, 0 where dropped (inverted dropout)."""
    if rate <= 0.0:
        return np.ones(n_elems, dtype=np.float64)
    thr = np.uint32(min(int(rate * 4294967296.0), 0xFFFFFFFF))
    keep = hash_u32(seed, stream, np.arange(n_elems, dtype=np.uint64)) >= thr
    return keep.astype(np.float64) / (1.0 - rate)


class TrainRef:
    def __init__(self, weights: Dict[str, np.ndarray], dtype=torch.float64):
        self.dtype = dtype
        self.p = {k: torch.tensor(np.asarray(v), dtype=dtype) for k, v in weights.items()}
        self.trainable = [k for k in self.p if not k.endswith((".mean", ".var"))]
        for k in self.trainable:
            self.p[k].requires_grad_(True)
        self.m = {k: torch.zeros_like(self.p[k]) for k in self.trainable}
        self.v = {k: torch.zeros_like(self.p[k]) for k in self.trainable}
        self.t = 0
        self.new_stats: Dict[str, torch.Tensor] = {}

    # ---- layers in training mode
    def _bn(self, x, prefix, dims):
        p = self.p
        mu = x.mean(dim=dims)
        var = x.var(dim=dims, unbiased=False)
        self.new_stats[prefix + ".mean"] = (BN_MOMENTUM * p[prefix + ".mean"] + (1 - BN_MOMENTUM) * mu).detach()
        self.new_stats[prefix + ".var"] = (BN_MOMENTUM * p[prefix + ".var"] + (1 - BN_MOMENTUM) * var).detach()
        shape = [1] * x.dim()
        shape[-1] = -1
        return (x - mu.view(shape)) / torch.sqrt(var.view(shape) + 1e-3) * p[prefix + ".gamma"].view(shape) \
            + p[prefix + ".beta"].view(shape)

    def _drop(self, x, stream, rate, spatial=False):
        if rate <= 0.0:
            return x
        if spatial:  # [B,W,C]: one mask value per (sample, channel)
            B, _, C = x.shape
            m = keep_mask(self.seed, stream, B * C, rate).reshape(B, 1, C)
        else:
            m = keep_mask(self.seed, stream, x.numel(), rate).reshape(x.shape)
        return x * torch.tensor(m, dtype=self.dtype)

    def _conv(self, x, k, b):  # x [B,W,C]; k [K,in,out]
        w = self.p[k].permute(2, 1, 0)
        return F.conv1d(x.transpose(1, 2), w, self.p[b], padding=(w.shape[-1] - 1) // 2).transpose(1, 2)

    def _tblock(self, x, pfx, bi, site0):
        p = self.p
        B, W, D = x.shape
        dh = D // HEADS
        q = (x @ p[pfx + ".wq"] + p[pfx + ".bq"]).view(B, W, HEADS, dh).permute(0, 2, 1, 3)
        k = (x @ p[pfx + ".wk"] + p[pfx + ".bk"]).view(B, W, HEADS, dh).permute(0, 2, 1, 3)
        v = (x @ p[pfx + ".wv"] + p[pfx + ".bv"]).view(B, W, HEADS, dh).permute(0, 2, 1, 3)
        a = torch.softmax(q @ k.transpose(-1, -2) / (dh ** 0.5), dim=-1)
        o = (a @ v).permute(0, 2, 1, 3).reshape(B, W, D) @ p[pfx + ".wo"] + p[pfx + ".bo"]
        o = self._drop(o, stream_id(bi, site0), self.rates["tb"])
        h = F.layer_norm(x + o, (D,), p[pfx + ".ln1_g"], p[pfx + ".ln1_b"], eps=1e-6)
        f = F.silu(h @ p[pfx + ".ffn1_k"] + p[pfx + ".ffn1_b"]) @ p[pfx + ".ffn2_k"] + p[pfx + ".ffn2_b"]
        f = self._drop(f, stream_id(bi, site0 + 1), self.rates["tb"])
        return F.layer_norm(h + f, (D,), p[pfx + ".ln2_g"], p[pfx + ".ln2_b"], eps=1e-6)

    def _branch(self, tok, bi, win):
        p = self.p
        pfx = f"b{win}"
        x = p[pfx + ".emb"][tok] + p[pfx + ".pos"].unsqueeze(0)
        x = self._drop(x, stream_id(bi, 0), self.rates["embed"], spatial=True)
        sc = self._conv(x, pfx + ".sc_k", pfx + ".sc_b")
        y = self._bn(F.silu(self._conv(x, pfx + ".c3a_k", pfx + ".c3a_b")), pfx + ".bn1", (0, 1))
        y = self._drop(y, stream_id(bi, 1), self.rates["conv"], spatial=True)
        y = self._bn(F.silu(self._conv(y, pfx + ".c3b_k", pfx + ".c3b_b")), pfx + ".bn2", (0, 1))
        x = y + sc
        x = self._tblock(x, pfx + ".tb1", bi, 2)
        x = self._tblock(x, pfx + ".tb2", bi, 4)
        a = torch.softmax(torch.tanh(x @ p[pfx + ".pool_k"] + p[pfx + ".pool_b"]).squeeze(-1), dim=1)
        return (x * a.unsqueeze(-1)).sum(dim=1)

    def forward(self, inputs: Sequence[np.ndarray], seed: int = 0, rates: Optional[Dict[str, float]] = None):
        """Training-mode forward; ``rates`` keys: embed, conv, tb, dense (0 disables)."""
        self.seed = seed
        self.rates = {"embed": 1e-4, "conv": 0.3, "tb": 0.1, "dense": 0.3}
        if rates is not None:
            self.rates.update(rates)
        p = self.p
        feats = [self._branch(torch.as_tensor(np.asarray(t).astype(np.int64)), bi, w)
                 for bi, (w, t) in enumerate(zip(WINDOWS, inputs[:3]))]
        st = torch.tensor(np.asarray(inputs[3], dtype=np.float32), dtype=self.dtype)
        esm = torch.tensor(np.asarray(inputs[4], dtype=np.float32), dtype=self.dtype)
        sb = self._bn(F.silu(st @ p["struct.k"] + p["struct.b"]), "struct.bn", (0,))
        sb = self._drop(sb, stream_id(3, 0), self.rates["dense"])
        eb = self._bn(F.silu(esm @ p["esm1.k"] + p["esm1.b"]), "esm1.bn", (0,))
        eb = self._drop(eb, stream_id(3, 1), self.rates["dense"])
        eb = F.silu(eb @ p["esm2.k"] + p["esm2.b"])
        m = torch.cat(feats + [sb, eb], dim=-1)
        m = self._drop(m, stream_id(3, 2), self.rates["dense"])
        x = self._bn(F.silu(m @ p["t1.k"] + p["t1.b"]), "t1.bn", (0,))
        x = self._drop(x, stream_id(3, 3), self.rates["dense"])
        x = self._bn(F.silu(x @ p["t2.k"] + p["t2.b"]), "t2.bn", (0,))
        return torch.softmax(x @ p["head.k"] + p["head.b"], dim=-1)

    def loss(self, proba, labels, n_global=None, n_pos_global=None, with_l2=True, l2_scale=1.0):
        """trainer.py:36-54 (+ the kernel regularisers Keras adds to the compiled loss)."""
        y = torch.as_tensor(np.asarray(labels).astype(np.int64))
        onehot = F.one_hot(y, 2).to(self.dtype)
        n = float(len(y)) if n_global is None else float(n_global)
        n_pos = float(y.sum()) if n_pos_global is None else float(n_pos_global)
        w_pos = n / (2.0 * n_pos + K_EPS)
        w_neg = n / (2.0 * (n - n_pos) + K_EPS)
        wts = (w_pos * y + w_neg * (1 - y)).to(self.dtype)
        o = proba.clamp(K_EPS, 1.0 - K_EPS)
        bce = -(onehot * torch.log(o + K_EPS) + (1 - onehot) * torch.log(1 - o + K_EPS))
        data = (wts.unsqueeze(1) * bce).sum() / (n * 2.0)
        reg = sum((self.p[k] ** 2).sum() for k in L2_TENSORS) * (L2 * l2_scale) if with_l2 else 0.0
        return data + reg

    def grads(self, inputs, labels, seed=0, rates=None, n_global=None, n_pos_global=None, l2_scale=1.0):
        """Forward + backward only.  Returns (loss share, grads dict of numpy arrays)."""
        for k in self.trainable:
            self.p[k].grad = None
        proba = self.forward(inputs, seed, rates)
        loss = self.loss(proba, labels, n_global, n_pos_global, True, l2_scale)
        loss.backward()
        return float(loss.detach()), {k: self.p[k].grad.detach().clone().numpy() for k in self.trainable}

    def step(self, inputs, labels, lr=5e-4, weight_decay=1e-4, seed=0, rates=None,
             n_global=None, n_pos_global=None):
        """One training step.  Returns (loss, grads dict).  Updates weights and BN stats."""
        loss, grads = self.grads(inputs, labels, seed, rates, n_global, n_pos_global)
        self.apply(grads, lr, weight_decay)
        return loss, grads

    def apply(self, grads, lr=5e-4, weight_decay=1e-4, b1=0.9, b2=0.999, eps=1e-7):
        self.t += 1
        lr_t = lr * np.sqrt(1.0 - b2 ** self.t) / (1.0 - b1 ** self.t)
        with torch.no_grad():
            for k in self.trainable:
                g = torch.as_tensor(grads[k], dtype=self.dtype)
                self.p[k].mul_(1.0 - weight_decay)
                self.m[k].mul_(b1).add_(g, alpha=1 - b1)
                self.v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
                self.p[k].sub_(lr_t * self.m[k] / (self.v[k].sqrt() + eps))
            for k, v in self.new_stats.items():
                self.p[k].copy_(v)

    def weights(self) -> Dict[str, np.ndarray]:
        return {k: v.detach().numpy().copy() for k, v in self.p.items()}

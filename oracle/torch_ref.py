"""Second, independent CPU restatement of the scoring path on torch's CPU kernels.
TEST INFRASTRUCTURE ONLY (same import rule as ``ptm_oracle.py``).

PARITY UNPINNED (see ``ptm_oracle.py``).  This file was written from the text of
``pred/train_PTM/model.py:16-210`` with torch's own layer functions
(``F.conv1d``, ``F.batch_norm(training=False)``, ``F.layer_norm``, ``F.silu``,
``F.embedding``) instead of hand-rolled arithmetic, so that a misreading in one
restatement shows up as a disagreement between the two.  It is also the
"Keras-equivalent CPU" baseline timed by ``bench.py`` (oneDNN/MKL-class kernels,
the same class TensorFlow-CPU uses).
"""
from __future__ import annotations

from typing import Dict, Sequence

import numpy as np
import torch
import torch.nn.functional as F

WINDOWS = (21, 33, 51)
HEADS = 4


class TorchRef:
    def __init__(self, weights: Dict[str, np.ndarray], dtype=torch.float32):
        self.dtype = dtype
        self.w = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dtype) for k, v in weights.items()}
        # Conv1D kernels [k,in,out] -> torch conv1d weight [out,in,k]
        for k in list(self.w):
            if k.endswith(("sc_k", "c3a_k", "c3b_k")):
                self.w[k + ":t"] = self.w[k].permute(2, 1, 0).contiguous()

    def _bn(self, x, p, channels_dim=1):
        w = self.w
        return F.batch_norm(x, w[p + ".mean"], w[p + ".var"], w[p + ".gamma"], w[p + ".beta"],
                            training=False, eps=1e-3)

    def _tblock(self, x, p):
        w = self.w
        B, W, D = x.shape
        dh = D // HEADS
        q = F.linear(x, w[p + ".wq"].t(), w[p + ".bq"]).view(B, W, HEADS, dh).permute(0, 2, 1, 3)
        k = F.linear(x, w[p + ".wk"].t(), w[p + ".bk"]).view(B, W, HEADS, dh).permute(0, 2, 1, 3)
        v = F.linear(x, w[p + ".wv"].t(), w[p + ".bv"]).view(B, W, HEADS, dh).permute(0, 2, 1, 3)
        att = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) / (dh ** 0.5), dim=-1)
        o = torch.matmul(att, v).permute(0, 2, 1, 3).reshape(B, W, D)
        o = F.linear(o, w[p + ".wo"].t(), w[p + ".bo"])
        h = F.layer_norm(x + o, (D,), w[p + ".ln1_g"], w[p + ".ln1_b"], eps=1e-6)
        f = F.silu(F.linear(h, w[p + ".ffn1_k"].t(), w[p + ".ffn1_b"]))
        f = F.linear(f, w[p + ".ffn2_k"].t(), w[p + ".ffn2_b"])
        return F.layer_norm(h + f, (D,), w[p + ".ln2_g"], w[p + ".ln2_b"], eps=1e-6)

    def _branch(self, tok, win):
        w = self.w
        p = f"b{win}"
        x = F.embedding(tok.long(), w[p + ".emb"]) + w[p + ".pos"].unsqueeze(0)   # [B,W,D]
        xc = x.transpose(1, 2)                                                      # [B,D,W]
        sc = F.conv1d(xc, w[p + ".sc_k:t"], w[p + ".sc_b"])
        y = F.silu(F.conv1d(xc, w[p + ".c3a_k:t"], w[p + ".c3a_b"], padding=1))
        y = self._bn(y, p + ".bn1")
        y = F.silu(F.conv1d(y, w[p + ".c3b_k:t"], w[p + ".c3b_b"], padding=1))
        y = self._bn(y, p + ".bn2")
        x = (y + sc).transpose(1, 2)
        x = self._tblock(x, p + ".tb1")
        x = self._tblock(x, p + ".tb2")
        a = torch.tanh(F.linear(x, w[p + ".pool_k"].t(), w[p + ".pool_b"])).squeeze(-1)
        a = torch.softmax(a, dim=1)
        return torch.einsum("bwd,bw->bd", x, a)

    @torch.no_grad()
    def forward(self, inputs: Sequence[np.ndarray]) -> torch.Tensor:
        w = self.w
        feats = []
        for win, tok in zip(WINDOWS, inputs[:3]):
            feats.append(self._branch(torch.as_tensor(np.asarray(tok).astype(np.int64)), win))
        st = torch.as_tensor(np.asarray(inputs[3]).astype(np.float32)).to(self.dtype)
        esm = torch.as_tensor(np.asarray(inputs[4]).astype(np.float32)).to(self.dtype)
        sb = self._bn(F.silu(F.linear(st, w["struct.k"].t(), w["struct.b"])), "struct.bn")
        eb = self._bn(F.silu(F.linear(esm, w["esm1.k"].t(), w["esm1.b"])), "esm1.bn")
        eb = F.silu(F.linear(eb, w["esm2.k"].t(), w["esm2.b"]))
        m = torch.cat(feats + [sb, eb], dim=-1)
        x = self._bn(F.silu(F.linear(m, w["t1.k"].t(), w["t1.b"])), "t1.bn")
        x = self._bn(F.silu(F.linear(x, w["t2.k"].t(), w["t2.b"])), "t2.bn")
        return torch.softmax(F.linear(x, w["head.k"].t(), w["head.b"]), dim=-1)

    def predict(self, inputs: Sequence[np.ndarray], batch_size: int = 32) -> np.ndarray:
        n = len(inputs[0])
        out = np.zeros((n, 2), dtype=np.float32)
        for s in range(0, n, batch_size):
            e = min(n, s + batch_size)
            out[s:e] = self.forward([np.asarray(a)[s:e] for a in inputs]).to(torch.float32).numpy()
        return out

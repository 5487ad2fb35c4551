"""CPU restatement (NumPy) of DeepPTMPred's scoring path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module; the product package never does.

PARITY UNPINNED: the reference ships no tests, golden vectors or weights for this
path, and its own implementation (TensorFlow/Keras 2.15 graph) cannot be imported
in the build container (tensorflow, keras, h5py absent).  This file restates
``pred/train_PTM/model.py`` line by line, plus the Keras-2.15 layer defaults the
reference relies on but does not spell out:

* ``BatchNormalization``: epsilon 1e-3, inference uses moving mean/variance,
  weight order gamma, beta, moving_mean, moving_variance.
* ``LayerNormalization(epsilon=1e-6)``: over the last axis, biased variance.
* ``Conv1D``: kernel ``[k, in, out]``, cross-correlation, ``padding='same'`` with
  k=3 pads one zero row on each side of EVERY window.
* ``Dense``: kernel ``[in, out]``; ``swish(x) = x * sigmoid(x)``.
* ``Embedding``: float inputs are cast to int32 before the lookup.

It is cross-checked against ``oracle/torch_ref.py`` (an independent restatement
on torch's CPU kernels); the two agreeing is what stands in for a real oracle.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np

BN_EPS = 1e-3
LN_EPS = 1e-6
WINDOWS = (21, 33, 51)
HEADS = 4


def swish(x):
    # keras 'swish' activation: x * sigmoid(x)  (model.py:103,134,138,187,191,194,200,204)
    return x / (1.0 + np.exp(-x))


def softmax(x, axis=-1):
    m = np.max(x, axis=axis, keepdims=True)
    e = np.exp(x - m)
    return e / np.sum(e, axis=axis, keepdims=True)


def batchnorm_infer(x, w, prefix):
    """keras BatchNormalization at inference (model.py:135,139,188,192,201,205)."""
    dt = x.dtype
    g, b = w[prefix + ".gamma"].astype(dt), w[prefix + ".beta"].astype(dt)
    mu, var = w[prefix + ".mean"].astype(dt), w[prefix + ".var"].astype(dt)
    return (x - mu) / np.sqrt(var + dt.type(BN_EPS)) * g + b


def layernorm(x, g, b):
    """keras LayerNormalization(epsilon=1e-6) (model.py:106-107,114,118)."""
    dt = x.dtype
    mu = x.mean(axis=-1, keepdims=True)
    var = ((x - mu) ** 2).mean(axis=-1, keepdims=True)
    return (x - mu) / np.sqrt(var + dt.type(LN_EPS)) * g.astype(dt) + b.astype(dt)


def conv1d_same(x, k, b):
    """Conv1D(padding='same', strides=1) on [B,W,C]; kernel [K,in,out] (model.py:130-138)."""
    dt = x.dtype
    K = k.shape[0]
    pad_l = (K - 1) // 2
    pad_r = K - 1 - pad_l
    xp = np.pad(x, ((0, 0), (pad_l, pad_r), (0, 0)))
    W = x.shape[1]
    out = np.zeros(x.shape[:2] + (k.shape[2],), dtype=dt)
    for t in range(K):
        out += xp[:, t:t + W, :] @ k[t].astype(dt)
    return out + b.astype(dt)


def residual_conv_block(x, w, p):
    """model.py:129-141."""
    shortcut = conv1d_same(x, w[p + ".sc_k"], w[p + ".sc_b"])
    y = swish(conv1d_same(x, w[p + ".c3a_k"], w[p + ".c3a_b"]))
    y = batchnorm_infer(y, w, p + ".bn1")
    # SpatialDropout1D: identity at inference
    y = swish(conv1d_same(y, w[p + ".c3b_k"], w[p + ".c3b_b"]))
    y = batchnorm_infer(y, w, p + ".bn2")
    return y + shortcut


def mhsa(x, w, p):
    """MultiHeadSelfAttention.call (model.py:60-83)."""
    dt = x.dtype
    B, W, D = x.shape
    dh = D // HEADS
    q = x @ w[p + ".wq"].astype(dt) + w[p + ".bq"].astype(dt)
    k = x @ w[p + ".wk"].astype(dt) + w[p + ".bk"].astype(dt)
    v = x @ w[p + ".wv"].astype(dt) + w[p + ".bv"].astype(dt)
    q = q.reshape(B, W, HEADS, dh).transpose(0, 2, 1, 3)
    k = k.reshape(B, W, HEADS, dh).transpose(0, 2, 1, 3)
    v = v.reshape(B, W, HEADS, dh).transpose(0, 2, 1, 3)
    s = (q @ k.transpose(0, 1, 3, 2)) / dt.type(np.sqrt(dh))
    a = softmax(s, axis=-1)
    o = (a @ v).transpose(0, 2, 1, 3).reshape(B, W, D)
    return o @ w[p + ".wo"].astype(dt) + w[p + ".bo"].astype(dt)


def transformer_block(x, w, p):
    """TransformerBlock.call with dropout inactive (model.py:111-118)."""
    dt = x.dtype
    h = layernorm(x + mhsa(x, w, p), w[p + ".ln1_g"], w[p + ".ln1_b"])
    u = swish(h @ w[p + ".ffn1_k"].astype(dt) + w[p + ".ffn1_b"].astype(dt))
    f = u @ w[p + ".ffn2_k"].astype(dt) + w[p + ".ffn2_b"].astype(dt)
    return layernorm(h + f, w[p + ".ln2_g"], w[p + ".ln2_b"])


def attention_pool(x, w, p):
    """model.py:176-182: Dense(1,tanh) -> softmax over positions -> weighted sum."""
    dt = x.dtype
    a = np.tanh(x @ w[p + ".pool_k"].astype(dt) + w[p + ".pool_b"].astype(dt))[..., 0]
    a = softmax(a, axis=-1)
    return (x * a[..., None]).sum(axis=1)


def sequence_branch(tok, w, win, dtype=np.float64, stages=None):
    """model.py:157-184 for one window size.  ``tok`` int [B,win]."""
    p = f"b{win}"
    tok = np.asarray(tok).astype(np.int32)
    x = w[p + ".emb"].astype(dtype)[tok] + w[p + ".pos"].astype(dtype)[None, :, :]
    if stages is not None:
        stages[p + ".embed"] = x
    x = residual_conv_block(x, w, p)
    if stages is not None:
        stages[p + ".conv"] = x
    x = transformer_block(x, w, p + ".tb1")
    if stages is not None:
        stages[p + ".tb1"] = x
    x = transformer_block(x, w, p + ".tb2")
    if stages is not None:
        stages[p + ".tb2"] = x
    x = attention_pool(x, w, p)
    if stages is not None:
        stages[p + ".pool"] = x
    return x


def forward(weights: Dict[str, np.ndarray], inputs: Sequence[np.ndarray], dtype=np.float64,
            stages: Dict[str, np.ndarray] | None = None) -> np.ndarray:
    """create_ptm_model's graph (model.py:143-210) at inference.

    ``inputs`` = ``[x21, x33, x51, struct, esm]`` (model.py:210 input order).
    Returns probabilities ``[N,2]`` in ``dtype``.  ``stages`` (optional dict)
    receives the intermediate activations."""
    dtype = np.dtype(dtype)
    w = weights
    branches: List[np.ndarray] = []
    for win, tok in zip(WINDOWS, inputs[:3]):
        branches.append(sequence_branch(tok, w, win, dtype, stages))
    st = np.asarray(inputs[3]).astype(np.float32).astype(dtype)   # keras casts inputs to fp32
    esm = np.asarray(inputs[4]).astype(np.float32).astype(dtype)
    sb = swish(st @ w["struct.k"].astype(dtype) + w["struct.b"].astype(dtype))
    sb = batchnorm_infer(sb, w, "struct.bn")                       # model.py:187-188
    eb = swish(esm @ w["esm1.k"].astype(dtype) + w["esm1.b"].astype(dtype))
    eb = batchnorm_infer(eb, w, "esm1.bn")                         # model.py:191-192
    eb = swish(eb @ w["esm2.k"].astype(dtype) + w["esm2.b"].astype(dtype))  # model.py:194
    merged = np.concatenate(branches + [sb, eb], axis=-1)          # model.py:197
    if stages is not None:
        stages["struct"] = sb
        stages["esm"] = eb
        stages["merged"] = merged
    x = swish(merged @ w["t1.k"].astype(dtype) + w["t1.b"].astype(dtype))
    x = batchnorm_infer(x, w, "t1.bn")                             # model.py:200-201
    x = swish(x @ w["t2.k"].astype(dtype) + w["t2.b"].astype(dtype))
    x = batchnorm_infer(x, w, "t2.bn")                             # model.py:204-205
    if stages is not None:
        stages["trunk"] = x
    logits = x @ w["head.k"].astype(dtype) + w["head.b"].astype(dtype)
    return softmax(logits, axis=-1)                                # model.py:207-208


def predict(weights, inputs, batch_size: int = 32, dtype=np.float32) -> np.ndarray:
    """``model.predict(inputs)`` (predict.py:599): batches of ``batch_size``, fp32 out."""
    n = len(inputs[0])
    out = np.zeros((n, 2), dtype=np.float32)
    for s in range(0, n, batch_size):
        e = min(n, s + batch_size)
        out[s:e] = forward(weights, [np.asarray(a)[s:e] for a in inputs], dtype).astype(np.float32)
    return out


# --------------------------------------------------------------------- window assembly

_LETTERS = {"A": 0, "C": 1, "D": 2, "E": 3, "F": 4, "G": 5, "H": 6, "I": 7, "K": 8, "L": 9,
            "M": 10, "N": 11, "P": 12, "Q": 13, "R": 14, "S": 15, "T": 16, "V": 17, "W": 18,
            "Y": 19, "*": 20}


def get_sequence_windows(sequence: str, positions: Sequence[int], window_size: int) -> List[str]:
    """predict.py:529-545 (= case.py:428-444, data_loader.py:201-222)."""
    half = (window_size - 1) // 2
    out = []
    for pos in positions:
        center = sequence[pos - 1]
        left = sequence[max(0, pos - 1 - half):pos - 1]
        right = sequence[pos:min(len(sequence), pos + half)]
        lp = max(0, half - len(left))
        rp = max(0, half - len(right))
        out.append("*" * lp + left + center + right + "*" * rp)
    return out


def encode_sequences(windows: Sequence[str]) -> np.ndarray:
    """predict.py:547-560: letter -> code, unknown letters -> 20; numpy default int."""
    if len(windows) == 0:
        return np.zeros((0, 0), dtype=int)
    enc = np.zeros((len(windows), len(windows[0])), dtype=int)
    for i, s in enumerate(windows):
        for j, aa in enumerate(s):
            enc[i, j] = _LETTERS.get(aa, 20)
    return enc


def prepare_esm_features(table, positions: Sequence[int], esm_dim: int = 1280) -> np.ndarray:
    """predict.py:487-501: row ``pos-1`` of the per-protein table, zeros when the
    table is missing or too short."""
    rows = []
    for pos in positions:
        if table is not None and pos - 1 < len(table):
            rows.append(np.array(table[pos - 1], dtype=np.float64, copy=True))
        else:
            rows.append(np.zeros(esm_dim))
    return np.array(rows).reshape(len(rows), esm_dim)


DEFAULT_STRUCT_ROW = np.array([0.5, 0.0, 0.0, 0, 1, 0, 85.0, 85.0])  # predict.py:524-526


def target_positions(sequence: str, target_aa: Sequence[str]) -> List[int]:
    """predict.py:465-467: 1-based positions whose residue is a target amino acid."""
    return [i + 1 for i, aa in enumerate(sequence) if aa in target_aa]

"""Keras-2 "legacy" HDF5 checkpoints <-> the canonical weight dict of ``spec.tensor_table``.

The reference saves whole models with ``ModelCheckpoint(... .h5)`` (pred/train_PTM/
trainer.py:321-326) and restores them with ``load_model`` (predict.py:569-577).  Such a
file holds ``/model_weights/<layer>/<weight_names...>`` with the attributes
``layer_names`` (on ``model_weights``) and ``weight_names`` (on every layer group);
``save_weights`` files hold the same tree at the root.

Layer names are the ones Keras auto-generates for ``create_ptm_model`` in a fresh
session (trainer.py:300 clears the session before every build), in creation order:
per branch one Embedding, one PositionEmbedding, three Conv1D, two BatchNormalization,
two TransformerBlock and one Dense; ``dense_*`` numbers also count the Dense layers that
live INSIDE the transformer blocks (two per block at construction, four more when the
block's attention layer is first built), which is why the branch pooling layers are
``dense_12``, ``dense_25`` and ``dense_38``.  Inside a layer Keras loads weights by
position, not by name; the order assumed for a TransformerBlock is attribute-tracking
order (model.py:54-57, 101-107): query, key, value, combine (kernel, bias each), the two
FFN Dense layers, layernorm1, layernorm2.  None of the reference's ``.h5`` files ship
with it, so this ordering is UNVERIFIED against a real checkpoint; shapes are checked.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np

from .hdf5_min import H5File, H5FormatError, H5Writer
from .spec import DEFAULT_SPEC, ModelSpec, tensor_table


def _sfx(base: str, i: int) -> str:
    return base if i == 0 else f"{base}_{i}"


def keras_layer_table(spec: ModelSpec = DEFAULT_SPEC) -> List[Tuple[str, List[Tuple[str, str]]]]:
    """[(keras layer name, [(keras weight name, canonical tensor name), ...]), ...] in creation order."""
    out: List[Tuple[str, List[Tuple[str, str]]]] = []
    bn_w = ("gamma", "beta", "moving_mean", "moving_variance")
    bn_c = ("gamma", "beta", "mean", "var")
    for bi, w in enumerate(spec.window_sizes):
        p = f"b{w}"
        emb, pos = _sfx("embedding", bi), _sfx("position_embedding", bi)
        out.append((emb, [(f"{emb}/embeddings:0", f"{p}.emb")]))
        out.append((pos, [(f"{pos}/position_embedding:0", f"{p}.pos")]))
        for ci, cname in enumerate(("sc", "c3a", "c3b")):
            ln = _sfx("conv1d", 3 * bi + ci)
            out.append((ln, [(f"{ln}/kernel:0", f"{p}.{cname}_k"), (f"{ln}/bias:0", f"{p}.{cname}_b")]))
            if ci >= 1:
                bn = _sfx("batch_normalization", 2 * bi + ci - 1)
                out.append((bn, [(f"{bn}/{a}:0", f"{p}.bn{ci}.{b}") for a, b in zip(bn_w, bn_c)]))
        for ti in range(2):
            tb = _sfx("transformer_block", 2 * bi + ti)
            c = f"{p}.tb{ti + 1}"
            d0 = 13 * bi + 6 * ti          # first Dense number created for this block (its FFN)
            att = _sfx("multi_head_self_attention", 2 * bi + ti)
            names = []
            for j, (kk, bb) in enumerate((("wq", "bq"), ("wk", "bk"), ("wv", "bv"), ("wo", "bo"))):
                dn = _sfx("dense", d0 + 2 + j)
                names += [(f"{tb}/{att}/{dn}/kernel:0", f"{c}.{kk}"), (f"{tb}/{att}/{dn}/bias:0", f"{c}.{bb}")]
            for j, (kk, bb) in enumerate((("ffn1_k", "ffn1_b"), ("ffn2_k", "ffn2_b"))):
                dn = _sfx("dense", d0 + j)
                names += [(f"{tb}/{dn}/kernel:0", f"{c}.{kk}"), (f"{tb}/{dn}/bias:0", f"{c}.{bb}")]
            for j in (1, 2):
                lnn = _sfx("layer_normalization", 4 * bi + 2 * ti + j - 1)
                names += [(f"{tb}/{lnn}/gamma:0", f"{c}.ln{j}_g"), (f"{tb}/{lnn}/beta:0", f"{c}.ln{j}_b")]
            out.append((tb, names))
        dn = _sfx("dense", 13 * bi + 12)
        out.append((dn, [(f"{dn}/kernel:0", f"{p}.pool_k"), (f"{dn}/bias:0", f"{p}.pool_b")]))
    nb = 13 * len(spec.window_sizes)
    nbn = 2 * len(spec.window_sizes)
    tail = [("struct", nb, nbn), ("esm1", nb + 1, nbn + 1), ("esm2", nb + 2, None), ("t1", nb + 3, nbn + 2),
            ("t2", nb + 4, nbn + 3), ("head", nb + 5, None)]
    for cname, di, bni in tail:
        dn = _sfx("dense", di)
        out.append((dn, [(f"{dn}/kernel:0", f"{cname}.k"), (f"{dn}/bias:0", f"{cname}.b")]))
        if bni is not None:
            bn = _sfx("batch_normalization", bni)
            out.append((bn, [(f"{bn}/{a}:0", f"{cname}.bn.{b}") for a, b in zip(bn_w, bn_c)]))
    return out


def read_keras_h5(path: str, spec: ModelSpec = DEFAULT_SPEC) -> Dict[str, np.ndarray]:
    """Canonical weight dict from a Keras-2 ``.h5`` model or weights file.  A damaged or truncated file
    raises ``H5FormatError`` (a ``ValueError``), whatever low-level error the parser ran into."""
    import struct
    try:
        return _read_keras_h5(path, spec)
    except H5FormatError:
        raise
    except (struct.error, IndexError, KeyError, UnicodeDecodeError, OverflowError, RecursionError, MemoryError,
            ValueError, TypeError, AttributeError, AssertionError) as e:
        raise H5FormatError(f"{path}: not a readable Keras-2 HDF5 checkpoint ({type(e).__name__}: {e})") from e


def _read_keras_h5(path: str, spec: ModelSpec) -> Dict[str, np.ndarray]:
    f = H5File(path)
    root = f["model_weights"] if "model_weights" in f.keys() and "layer_names" not in f.attrs else f
    shapes = dict(tensor_table(spec))
    have = set(root.keys())
    out: Dict[str, np.ndarray] = {}
    for layer, pairs in keras_layer_table(spec):
        if layer not in have:
            raise H5FormatError(f"{path}: layer group {layer!r} not found (file has {sorted(have)[:8]}...); "
                                "the checkpoint was not written by create_ptm_model in a fresh Keras session")
        g = root[layer]
        wn = g.attrs.get("weight_names")
        if wn is None:
            raise H5FormatError(f"{path}: layer {layer!r} has no weight_names attribute")
        wn = [x.decode("utf-8") if isinstance(x, bytes) else str(x) for x in np.atleast_1d(wn)]
        if len(wn) != len(pairs):
            raise H5FormatError(f"{path}: layer {layer!r} holds {len(wn)} weights, expected {len(pairs)}")
        for name, (_, canon) in zip(wn, pairs):     # by position, as Keras does
            a = np.asarray(g[name].read(), dtype=np.float32)
            if a.shape != tuple(shapes[canon]):
                raise H5FormatError(f"{path}: {layer}/{name} has shape {a.shape}, expected {shapes[canon]} for {canon}")
            out[canon] = a
    missing = [n for n, _ in tensor_table(spec) if n not in out]
    if missing:
        raise H5FormatError(f"{path}: tensors not found: {missing[:5]}")
    return out


def write_keras_h5(path: str, weights: Dict[str, np.ndarray], spec: ModelSpec = DEFAULT_SPEC,
                   keras_version: str = "2.15.0") -> None:
    """Writes the ``/model_weights`` tree of a Keras-2 checkpoint (no ``model_config``:
    Keras restores it with ``create_ptm_model(config)`` + ``model.load_weights(path)``)."""
    w = H5Writer()
    w.set_attr("/", "backend", b"tensorflow")
    w.set_attr("/", "keras_version", keras_version.encode())
    w.create_group("/model_weights")
    w.set_attr("/model_weights", "backend", b"tensorflow")
    w.set_attr("/model_weights", "keras_version", keras_version.encode())
    table = keras_layer_table(spec)
    w.set_attr("/model_weights", "layer_names", [ln.encode() for ln, _ in table])
    for layer, pairs in table:
        w.create_group(f"/model_weights/{layer}")
        w.set_attr(f"/model_weights/{layer}", "weight_names", [kn.encode() for kn, _ in pairs])
        for kn, canon in pairs:
            w.create_dataset(f"/model_weights/{layer}/{kn}", np.asarray(weights[canon], dtype=np.float32))
    w.save(path)

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


def keras_model_layers(spec: ModelSpec = DEFAULT_SPEC) -> List[str]:
    """Names of ``model.layers`` for ``create_ptm_model`` (model.py:143-210) in the order Keras 2 gives a
    functional model: by decreasing depth (longest path to the output), ties broken by the order of a
    depth-first walk from the output through each layer's inputs (keras/engine/functional.py,
    ``_map_graph_network``).  ``load_weights`` on an ``.h5`` file pairs the file's weighted layers with the
    model's weighted layers BY POSITION in this order, so a checkpoint has to list them this way.  Layer
    names are the auto-generated ones of a fresh session, numbered in creation order."""
    counts: Dict[str, int] = {}
    inputs: Dict[str, List[str]] = {}

    def name(base: str) -> str:
        i = counts.get(base, 0)
        counts[base] = i + 1
        return _sfx(base, i)

    def layer(base: str, *ins: str, given: str = "") -> str:
        n = given or name(base)
        inputs[n] = list(ins)
        return n

    def transformer_block(x: str) -> str:
        n = name("transformer_block")                 # __init__ (model.py:94-109): sub-layers take their names now
        name("multi_head_self_attention")
        name("dense"), name("dense"), name("sequential")
        name("layer_normalization"), name("layer_normalization")
        name("dropout"), name("dropout")
        inputs[n] = [x]
        for _ in range(4):                            # first call builds the attention layer (model.py:53-57)
            name("dense")
        return n

    seq_in = [layer("", given=f"seq_input_{w}") for w in spec.window_sizes]
    struct_in = layer("", given="struct_input")
    esm_in = layer("", given="esm_input")
    branches = []
    for inp in seq_in:
        x = layer("embedding", inp)
        x = layer("position_embedding", x)
        x = layer("spatial_dropout1d", x)
        shortcut = layer("conv1d", x)                 # residual_conv_block, model.py:129-141
        y = layer("conv1d", x)
        y = layer("batch_normalization", y)
        y = layer("spatial_dropout1d", y)
        y = layer("conv1d", y)
        y = layer("batch_normalization", y)
        x = layer("add", y, shortcut)
        x = transformer_block(x)
        x = transformer_block(x)
        a = layer("dense", x)                         # attention pooling, model.py:176-182
        a = layer("flatten", a)
        a = layer("activation", a)
        a = layer("repeat_vector", a)
        a = layer("permute", a)
        x = layer("multiply", x, a)
        branches.append(layer("lambda", x))
    st = layer("dense", struct_in)
    st = layer("batch_normalization", st)
    st = layer("dropout", st)
    e = layer("dense", esm_in)
    e = layer("batch_normalization", e)
    e = layer("dropout", e)
    e = layer("dense", e)
    x = layer("concatenate", *branches, st, e)
    x = layer("dropout", x)
    x = layer("dense", x)
    x = layer("batch_normalization", x)
    x = layer("dropout", x)
    x = layer("dense", x)
    x = layer("batch_normalization", x)
    out = layer("dense", x)

    index: Dict[str, int] = {}
    finished: List[str] = []                          # post-order: inputs before the layers that use them

    def walk(n: str) -> None:
        if n in index:
            return
        index[n] = len(index)
        for m in inputs[n]:
            walk(m)
        finished.append(n)

    import sys
    limit = sys.getrecursionlimit()
    sys.setrecursionlimit(max(limit, 10000))
    try:
        walk(out)
    finally:
        sys.setrecursionlimit(limit)
    depth: Dict[str, int] = {}
    for n in reversed(finished):
        d = depth.setdefault(n, 0)
        for m in inputs[n]:
            depth[m] = max(depth.get(m, 0), d + 1)
    return sorted(finished, key=lambda n: (-depth[n], index[n]))


def keras_weight_order(spec: ModelSpec = DEFAULT_SPEC) -> List[str]:
    """Canonical tensor names in the order of Keras' ``model.weights`` / ``model.get_weights()`` for the
    model of model.py:210: layer by layer in ``model.layers`` order (``keras_model_layers``: by depth, so
    the three Embedding tables come first), and inside a layer trainable before non-trainable variables
    in attribute-tracking order -- BatchNormalization gamma, beta, moving mean, moving variance; a
    TransformerBlock query, key, value, combine, the two FFN layers, the two LayerNormalizations."""
    by_name = dict(keras_layer_table(spec))
    order = [canon for ln in keras_model_layers(spec) if ln in by_name for _, canon in by_name[ln]]
    assert sorted(order) == sorted(n for n, _ in tensor_table(spec))
    return order


def _split_suffix(name: str) -> Tuple[str, int]:
    """'dense_12' -> ('dense', 12); 'dense' -> ('dense', 0)."""
    head, sep, tail = name.rpartition("_")
    if sep and tail.isdigit():
        return head, int(tail)
    return name, 0


def _layers_by_creation_order(names: List[str], spec: ModelSpec) -> Dict[str, str]:
    """Maps the layer names of ``keras_layer_table`` to the names a file actually uses when its layers were not
    numbered from zero (a model built in a session that already held layers: ``dense_137`` ...).  Keras numbers
    a class's layers in creation order, so within a class the k-th smallest suffix is the k-th layer
    ``create_ptm_model`` creates, whatever the numbering started at and whatever order the file lists them in."""
    want: Dict[str, List[Tuple[int, str]]] = {}
    for layer, _ in keras_layer_table(spec):
        cls, k = _split_suffix(layer)
        want.setdefault(cls, []).append((k, layer))
    have: Dict[str, List[Tuple[int, str]]] = {}
    for n in names:
        cls, k = _split_suffix(n)
        if cls in want:
            have.setdefault(cls, []).append((k, n))
    out: Dict[str, str] = {}
    for cls, ws in want.items():
        hs = sorted(have.get(cls, []))
        if len(hs) != len(ws):
            raise H5FormatError(f"expected {len(ws)} weighted '{cls}' layers, the file has {len(hs)}")
        for (_, canon), (_, actual) in zip(sorted(ws), hs):
            out[canon] = actual
    return out


def _block_by_inner_names(wn: List[str], arrays: List[np.ndarray], c: str) -> Dict[str, np.ndarray]:
    """Tensors of one TransformerBlock identified by what they are, not by where they stand: the four 64x64
    Dense layers of the attention are query, key, value, combine in creation order (model.py:54-57) = increasing
    ``dense_N``; the FFN layers are told apart by shape; the LayerNormalizations by increasing number.  Returns
    {} when the names do not have the ``.../<inner layer>/<kernel|bias|gamma|beta>:0`` form."""
    groups: Dict[str, Dict[str, np.ndarray]] = {}
    for n, a in zip(wn, arrays):
        parts = n.split("/")
        if len(parts) < 2:
            return {}
        groups.setdefault(parts[-2], {})[parts[-1].split(":")[0]] = a
    dense = sorted((_split_suffix(k)[1], v) for k, v in groups.items() if _split_suffix(k)[0] == "dense")
    norms = sorted((_split_suffix(k)[1], v) for k, v in groups.items() if _split_suffix(k)[0] == "layer_normalization")
    if len(dense) != 6 or len(norms) != 2 or any(set(v) != {"kernel", "bias"} for _, v in dense) or \
            any(set(v) != {"gamma", "beta"} for _, v in norms):
        return {}
    square = [v for _, v in dense if v["kernel"].shape[0] == v["kernel"].shape[1]]
    wide = [v for _, v in dense if v["kernel"].shape[0] < v["kernel"].shape[1]]
    tall = [v for _, v in dense if v["kernel"].shape[0] > v["kernel"].shape[1]]
    if len(square) != 4 or len(wide) != 1 or len(tall) != 1:
        return {}
    out = {}
    for (kk, bb), v in zip((("wq", "bq"), ("wk", "bk"), ("wv", "bv"), ("wo", "bo")), square):
        out[f"{c}.{kk}"], out[f"{c}.{bb}"] = v["kernel"], v["bias"]
    out[f"{c}.ffn1_k"], out[f"{c}.ffn1_b"] = wide[0]["kernel"], wide[0]["bias"]
    out[f"{c}.ffn2_k"], out[f"{c}.ffn2_b"] = tall[0]["kernel"], tall[0]["bias"]
    for j, (_, v) in enumerate(norms, start=1):
        out[f"{c}.ln{j}_g"], out[f"{c}.ln{j}_b"] = v["gamma"], v["beta"]
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
    table = keras_layer_table(spec)
    actual = {layer: layer for layer, _ in table}
    if any(layer not in have for layer, _ in table):
        # not the numbering of a fresh session: identify layers by class and creation order instead
        def weighted(n: str) -> bool:
            wn_ = root[n].attrs.get("weight_names")
            return wn_ is not None and len(np.atleast_1d(wn_)) > 0
        try:
            actual = _layers_by_creation_order([n for n in have if weighted(n)], spec)
        except H5FormatError as e:
            raise H5FormatError(f"{path}: layer groups match neither the names of a fresh Keras session nor the layer "
                                f"counts of create_ptm_model ({e}); file has {sorted(have)[:8]}...") from e
    out: Dict[str, np.ndarray] = {}
    for layer, pairs in table:
        g = root[actual[layer]]
        wn = g.attrs.get("weight_names")
        if wn is None:
            raise H5FormatError(f"{path}: layer {actual[layer]!r} has no weight_names attribute")
        wn = [x.decode("utf-8") if isinstance(x, bytes) else str(x) for x in np.atleast_1d(wn)]
        if len(wn) != len(pairs):
            raise H5FormatError(f"{path}: layer {actual[layer]!r} holds {len(wn)} weights, expected {len(pairs)}")
        arrays = [np.asarray(g[name].read(), dtype=np.float32) for name in wn]
        found = {canon: a for (_, canon), a in zip(pairs, arrays)}     # by position, as Keras pairs them
        if _split_suffix(layer)[0] == "transformer_block":
            named = _block_by_inner_names(wn, arrays, pairs[0][1].rsplit(".", 1)[0])
            if named:
                found = named
        for canon, a in found.items():
            if a.shape != tuple(shapes[canon]):
                raise H5FormatError(f"{path}: {actual[layer]} holds shape {a.shape} where {canon} expects {shapes[canon]}")
            out[canon] = a
    missing = [n for n, _ in tensor_table(spec) if n not in out]
    if missing:
        raise H5FormatError(f"{path}: tensors not found: {missing[:5]}")
    return out


def write_keras_h5(path: str, weights: Dict[str, np.ndarray], spec: ModelSpec = DEFAULT_SPEC,
                   keras_version: str = "2.15.0") -> None:
    """Writes the ``/model_weights`` tree of a Keras-2 checkpoint (no ``model_config``: Keras restores it with
    ``create_ptm_model(config)`` + ``model.load_weights(path)``).  ``layer_names`` lists the weighted layers in
    ``model.layers`` order (``keras_model_layers``), which is how Keras pairs them on load; with no TensorFlow at
    hand this direction is checked against the published ordering rule only, not against Keras itself."""
    w = H5Writer()
    w.set_attr("/", "backend", b"tensorflow")
    w.set_attr("/", "keras_version", keras_version.encode())
    w.create_group("/model_weights")
    w.set_attr("/model_weights", "backend", b"tensorflow")
    w.set_attr("/model_weights", "keras_version", keras_version.encode())
    by_name = dict(keras_layer_table(spec))
    # weighted layers in model.layers order: load_weights pairs them with the model's by position
    table = [(ln, by_name[ln]) for ln in keras_model_layers(spec) if ln in by_name]
    assert len(table) == len(by_name)
    w.set_attr("/model_weights", "layer_names", [ln.encode() for ln, _ in table])
    for layer, pairs in table:
        w.create_group(f"/model_weights/{layer}")
        w.set_attr(f"/model_weights/{layer}", "weight_names", [kn.encode() for kn, _ in pairs])
        for kn, canon in pairs:
            w.create_dataset(f"/model_weights/{layer}/{kn}", np.asarray(weights[canon], dtype=np.float32))
    w.save(path)

"""Model dimensions, the canonical weight order and the flat weight container.

The PTM-site classifier this package scores is defined by the reference in
``pred/train_PTM/model.py:143-210`` with the dimensions of
``pred/train_PTM/config.py:49-76`` (= ``predict.py:163-169``).  Everything the
native library and the CPU oracle need to agree on lives here:

* ``ModelSpec``       -- the dimensions (attribute names follow the reference's
                         ``Config`` / ``PredictConfig`` classes).
* ``tensor_table``    -- every weight tensor, in ONE fixed canonical order.  The
                         C ABI (``include/dptm.h``) takes a single little-endian
                         fp32 blob that is the concatenation of these tensors in
                         this order, each C-contiguous in its Keras shape.
* ``synthetic_weights`` / ``synthetic_inputs`` -- seeded generators used by the
                         tests and by ``bench.py`` (no trained weights ship with
                         the reference mount).
* ``save_flat`` / ``load_flat`` -- a repo-native weight file: ``<name>.dptm`` =
                         JSON header line + raw fp32 blob.
"""
from __future__ import annotations

import json
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import numpy as np

BN_EPS = 1e-3  # keras BatchNormalization default epsilon (model.py:135,139,188,192,201,205)
LN_EPS = 1e-6  # model.py:106-107

# predict.py:549-553 / data_loader.py:226-230: letter -> code; anything else -> 20
LETTER_CODES = {
    "A": 0, "C": 1, "D": 2, "E": 3, "F": 4, "G": 5, "H": 6, "I": 7, "K": 8, "L": 9,
    "M": 10, "N": 11, "P": 12, "Q": 13, "R": 14, "S": 15, "T": 16, "V": 17, "W": 18,
    "Y": 19, "*": 20,
}
PAD_CODE = 20

# predict.py:119-137
PTM_AA_MAP = {
    "phosphorylation": ["S", "T"],
    "ubiquitination": ["K"],
    "acetylation": ["K"],
    "hydroxylation": ["P"],
    "gamma_carboxyglutamic_acid": ["E"],
    "lys_methylation": ["K"],
    "malonylation": ["K"],
    "arg_methylation": ["R"],
    "crotonylation": ["K"],
    "succinylation": ["K"],
    "glutathionylation": ["C"],
    "sumoylation": ["K"],
    "s_nitrosylation": ["C"],
    "glutarylation": ["K"],
    "citrullination": ["R"],
    "o_linked_glycosylation": ["S", "T"],
    "n_linked_glycosylation": ["N"],
}


@dataclass(frozen=True)
class ModelSpec:
    """Dimensions of the classifier (config.py:49-76).  Only this one shape is
    built natively; the library rejects any other spec."""

    window_sizes: Tuple[int, ...] = (21, 33, 51)
    max_features: int = 21
    embed_dim: int = 64
    num_heads: int = 4
    ff_dim: int = 256
    esm_dim: int = 1280
    struct_feature_dim: int = 8
    dropout_rate: float = 0.3
    weight_decay: float = 1e-4
    struct_units: int = 128
    esm_units1: int = 512
    esm_units2: int = 256
    trunk_units1: int = 512
    trunk_units2: int = 256
    n_classes: int = 2

    @property
    def concat_dim(self) -> int:
        return len(self.window_sizes) * self.embed_dim + self.struct_units + self.esm_units2

    def as_c_array(self) -> List[int]:
        """The ints of ``dptm_spec`` in include/dptm.h, in field order."""
        return [
            len(self.window_sizes), *self.window_sizes, self.max_features, self.embed_dim,
            self.num_heads, self.ff_dim, self.esm_dim, self.struct_feature_dim,
            self.struct_units, self.esm_units1, self.esm_units2, self.trunk_units1,
            self.trunk_units2, self.n_classes,
        ]


DEFAULT_SPEC = ModelSpec()


def _bn(prefix: str, n: int) -> List[Tuple[str, Tuple[int, ...]]]:
    # keras BatchNormalization weight order: gamma, beta, moving_mean, moving_variance
    return [(f"{prefix}.gamma", (n,)), (f"{prefix}.beta", (n,)),
            (f"{prefix}.mean", (n,)), (f"{prefix}.var", (n,))]


def _tblock(prefix: str, d: int, f: int) -> List[Tuple[str, Tuple[int, ...]]]:
    # attribute-tracking order of model.py:54-57 and 101-107
    return [
        (f"{prefix}.wq", (d, d)), (f"{prefix}.bq", (d,)),
        (f"{prefix}.wk", (d, d)), (f"{prefix}.bk", (d,)),
        (f"{prefix}.wv", (d, d)), (f"{prefix}.bv", (d,)),
        (f"{prefix}.wo", (d, d)), (f"{prefix}.bo", (d,)),
        (f"{prefix}.ffn1_k", (d, f)), (f"{prefix}.ffn1_b", (f,)),
        (f"{prefix}.ffn2_k", (f, d)), (f"{prefix}.ffn2_b", (d,)),
        (f"{prefix}.ln1_g", (d,)), (f"{prefix}.ln1_b", (d,)),
        (f"{prefix}.ln2_g", (d,)), (f"{prefix}.ln2_b", (d,)),
    ]


def tensor_table(spec: ModelSpec = DEFAULT_SPEC) -> List[Tuple[str, Tuple[int, ...]]]:
    """(name, shape) of every weight tensor in the canonical blob order."""
    d, f = spec.embed_dim, spec.ff_dim
    t: List[Tuple[str, Tuple[int, ...]]] = []
    for w in spec.window_sizes:
        p = f"b{w}"
        t += [(f"{p}.emb", (spec.max_features, d)), (f"{p}.pos", (w, d))]
        t += [(f"{p}.sc_k", (1, d, d)), (f"{p}.sc_b", (d,))]
        t += [(f"{p}.c3a_k", (3, d, d)), (f"{p}.c3a_b", (d,))] + _bn(f"{p}.bn1", d)
        t += [(f"{p}.c3b_k", (3, d, d)), (f"{p}.c3b_b", (d,))] + _bn(f"{p}.bn2", d)
        t += _tblock(f"{p}.tb1", d, f) + _tblock(f"{p}.tb2", d, f)
        t += [(f"{p}.pool_k", (d, 1)), (f"{p}.pool_b", (1,))]
    t += [("struct.k", (spec.struct_feature_dim, spec.struct_units)),
          ("struct.b", (spec.struct_units,))] + _bn("struct.bn", spec.struct_units)
    t += [("esm1.k", (spec.esm_dim, spec.esm_units1)),
          ("esm1.b", (spec.esm_units1,))] + _bn("esm1.bn", spec.esm_units1)
    t += [("esm2.k", (spec.esm_units1, spec.esm_units2)), ("esm2.b", (spec.esm_units2,))]
    t += [("t1.k", (spec.concat_dim, spec.trunk_units1)),
          ("t1.b", (spec.trunk_units1,))] + _bn("t1.bn", spec.trunk_units1)
    t += [("t2.k", (spec.trunk_units1, spec.trunk_units2)),
          ("t2.b", (spec.trunk_units2,))] + _bn("t2.bn", spec.trunk_units2)
    t += [("head.k", (spec.trunk_units2, spec.n_classes)), ("head.b", (spec.n_classes,))]
    return t


def n_params(spec: ModelSpec = DEFAULT_SPEC) -> int:
    return int(sum(int(np.prod(s)) for _, s in tensor_table(spec)))


def pack_blob(weights: Dict[str, np.ndarray], spec: ModelSpec = DEFAULT_SPEC) -> np.ndarray:
    """Concatenate a name->array dict into the canonical fp32 blob."""
    parts = []
    for name, shape in tensor_table(spec):
        a = np.asarray(weights[name], dtype=np.float32)
        if a.shape != tuple(shape):
            raise ValueError(f"weight {name}: shape {a.shape}, expected {shape}")
        parts.append(np.ascontiguousarray(a).ravel())
    return np.concatenate(parts)


def unpack_blob(blob: np.ndarray, spec: ModelSpec = DEFAULT_SPEC) -> Dict[str, np.ndarray]:
    blob = np.asarray(blob, dtype=np.float32).ravel()
    if blob.size != n_params(spec):
        raise ValueError(f"blob has {blob.size} floats, expected {n_params(spec)}")
    out, off = {}, 0
    for name, shape in tensor_table(spec):
        n = int(np.prod(shape))
        out[name] = blob[off:off + n].reshape(shape)
        off += n
    return out


_MAGIC = "DPTM-FLAT-1"


def save_flat(path: str, weights: Dict[str, np.ndarray], spec: ModelSpec = DEFAULT_SPEC,
              meta: Dict | None = None) -> None:
    """Write ``path``: one JSON header line, then the raw little-endian fp32 blob."""
    blob = pack_blob(weights, spec).astype("<f4")
    header = {
        "magic": _MAGIC, "n_floats": int(blob.size),
        "spec": {k: (list(v) if isinstance(v, tuple) else v) for k, v in spec.__dict__.items()},
        "tensors": [[n, list(s)] for n, s in tensor_table(spec)],
        "meta": meta or {},
    }
    with open(path, "wb") as fh:
        fh.write((json.dumps(header) + "\n").encode("utf-8"))
        fh.write(blob.tobytes())


def load_flat(path: str) -> Tuple[Dict[str, np.ndarray], ModelSpec, Dict]:
    with open(path, "rb") as fh:
        header = json.loads(fh.readline().decode("utf-8"))
        if header.get("magic") != _MAGIC:
            raise ValueError(f"{path}: not a {_MAGIC} file")
        raw = fh.read()
    sd = dict(header["spec"])
    sd["window_sizes"] = tuple(sd["window_sizes"])
    spec = ModelSpec(**sd)
    blob = np.frombuffer(raw, dtype="<f4")
    if blob.size != header["n_floats"]:
        raise ValueError(f"{path}: truncated blob ({blob.size} of {header['n_floats']} floats)")
    return unpack_blob(blob.astype(np.float32), spec), spec, header.get("meta", {})


# --------------------------------------------------------------------------- synthetic data

def synthetic_weights(seed: int = 7, spec: ModelSpec = DEFAULT_SPEC) -> Dict[str, np.ndarray]:
    """Seeded stand-in for a trained model: Glorot-uniform kernels, small random
    biases, BatchNorm statistics with strictly positive variance, norm scales
    near one.  Deterministic in ``seed`` (numpy ``default_rng``)."""
    rng = np.random.default_rng(seed)
    out: Dict[str, np.ndarray] = {}
    for name, shape in tensor_table(spec):
        leaf = name.rsplit(".", 1)[1]
        if leaf in ("gamma", "ln1_g", "ln2_g"):
            a = rng.uniform(0.8, 1.2, shape)
        elif leaf in ("beta", "mean", "ln1_b", "ln2_b"):
            a = rng.normal(0.0, 0.1, shape)
        elif leaf == "var":
            a = rng.uniform(0.5, 1.5, shape)
        elif len(shape) == 1:  # bias vectors
            a = rng.normal(0.0, 0.05, shape)
        elif leaf in ("emb", "pos"):
            lim = np.sqrt(6.0 / (shape[0] + shape[1]))
            a = rng.uniform(-lim, lim, shape)
        else:  # dense [in,out] or conv [k,in,out] kernels: Glorot uniform
            rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
            fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
            lim = np.sqrt(6.0 / (fan_in + fan_out))
            a = rng.uniform(-lim, lim, shape)
        out[name] = a.astype(np.float32)
    # Structure rows are raw (A^2, degrees, 0-100 pLDDT; predict.py:308-318), so a
    # trained first layer must have small weights on them; the trunk/head scales
    # spread the synthetic probabilities over (0,1) instead of saturating them,
    # which keeps parity checks sensitive.
    raw_scale = np.array([0.01, 0.01, 0.01, 1, 1, 1, 0.02, 0.02], dtype=np.float32)
    out["struct.k"] = (out["struct.k"] * raw_scale[: spec.struct_feature_dim, None]).astype(np.float32)
    for name, s in (("t1.k", 2.0), ("t2.k", 2.0), ("head.k", 2.5)):
        out[name] = (out[name] * np.float32(s)).astype(np.float32)
    return out


def keras_initial_weights(seed: int = 0, spec: ModelSpec = DEFAULT_SPEC) -> Dict[str, np.ndarray]:
    """The state ``create_ptm_model(config)`` starts from (model.py:143-210 under the Keras-2 default
    initialisers): ``Embedding`` tables uniform(-0.05, 0.05) (model.py:157), position tables and every
    Dense / Conv1D kernel Glorot-uniform (model.py:22-27; limit sqrt(6 / (fan_in + fan_out)) with the
    kernel width counted into both fans), all biases zero, BatchNormalization gamma 1 / beta 0 /
    moving mean 0 / moving variance 1, LayerNormalization gamma 1 / beta 0, and the head bias
    ``Constant([0.1, 1.0])`` (model.py:207-208).  The values come from numpy's ``default_rng(seed)``,
    not TensorFlow's generator: the distribution is the reference's, the draws are not."""
    rng = np.random.default_rng(seed)
    out: Dict[str, np.ndarray] = {}
    for name, shape in tensor_table(spec):
        leaf = name.rsplit(".", 1)[1]
        if leaf in ("gamma", "var", "ln1_g", "ln2_g"):
            a = np.ones(shape)
        elif leaf == "emb":
            a = rng.uniform(-0.05, 0.05, shape)
        elif len(shape) == 1:      # biases, beta, moving mean
            a = np.zeros(shape)
        else:
            rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
            lim = np.sqrt(6.0 / (shape[-2] * rf + shape[-1] * rf))
            a = rng.uniform(-lim, lim, shape)
        out[name] = a.astype(np.float32)
    if spec.n_classes == 2:
        out["head.b"] = np.array([0.1, 1.0], dtype=np.float32)
    return out


def synthetic_inputs(n: int, seed: int = 1234, spec: ModelSpec = DEFAULT_SPEC,
                     pad_fraction: float = 0.05):
    """Synthetic sites shaped like real ones: residue codes 0..19 with edge
    padding (code 20) on ``pad_fraction`` of the sites, ESM rows ~N(0, 0.27^2)
    with 1 % of entries x10 (the shipped tables have std 0.27-0.29 and range
    -8.8..3.7), raw structure rows [sasa, phi, psi, H, E, L, plddt, plddt].

    Returns ``[tok_w0, tok_w1, tok_w2, struct, esm]`` in the model's input order:
    uint8 ``[n,W]`` x3, float32 ``[n,8]``, float32 ``[n,1280]``."""
    rng = np.random.default_rng(seed)
    toks = []
    # one residue string per site at the widest window; narrower windows are its centre
    wmax = max(spec.window_sizes)
    hmax = (wmax - 1) // 2
    wide = rng.integers(0, 20, size=(n, wmax), dtype=np.uint8)
    padded = rng.random(n) < pad_fraction
    side = rng.integers(0, 2, size=n)
    plen = rng.integers(1, hmax + 1, size=n)
    col = np.arange(wmax)[None, :]
    left_mask = padded[:, None] & (side[:, None] == 0) & (col < plen[:, None])
    right_mask = padded[:, None] & (side[:, None] == 1) & (col >= wmax - plen[:, None])
    wide[left_mask | right_mask] = PAD_CODE
    for w in spec.window_sizes:
        h = (w - 1) // 2
        toks.append(np.ascontiguousarray(wide[:, hmax - h:hmax + h + 1]))
    esm = rng.normal(0.0, 0.27, size=(n, spec.esm_dim)).astype(np.float32)
    heavy = rng.random((n, spec.esm_dim)) < 0.01
    esm[heavy] *= 10.0
    st = np.zeros((n, spec.struct_feature_dim), dtype=np.float32)
    st[:, 0] = rng.uniform(0, 250, n)
    st[:, 1] = rng.uniform(-180, 180, n)
    st[:, 2] = rng.uniform(-180, 180, n)
    ss = rng.integers(0, 3, n)
    st[np.arange(n), 3 + ss] = 1.0
    st[:, 6] = rng.uniform(20, 100, n)
    st[:, 7] = rng.uniform(20, 100, n)
    return toks + [st, esm]

"""The C-ABI library loads and exports every entry point include/dptm.h declares
(no compute: this runs without a GPU), and the host-side wrapper validates inputs."""
import os
import re

import numpy as np
import pytest

from deepptmpred_amd import _native, spec
from deepptmpred_amd.model import Model, create_ptm_model, encode_sequence, load_model

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "dptm.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dptm_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_exported():
    lib = _native.load_library()
    names = declared_symbols()
    assert "dptm_create" in names and "dptm_score" in names and "dptm_score_protein" in names
    for n in names:
        assert hasattr(lib, n), f"{n} declared in dptm.h but not exported"
    assert sorted(_native.EXPORTS) == names


def test_weight_count_matches_spec():
    lib = _native.load_library()
    cs = _native.DptmSpec.from_spec(spec.DEFAULT_SPEC)
    assert lib.dptm_weight_count(cs) == spec.n_params()
    bad = _native.DptmSpec.from_spec(spec.ModelSpec(embed_dim=32))
    assert lib.dptm_weight_count(bad) < 0
    assert b"unsupported" in lib.dptm_last_error()


def test_predict_validates_inputs_without_gpu():
    m = create_ptm_model(seed=1)
    x = spec.synthetic_inputs(4, 1)
    with pytest.raises(ValueError):
        m.predict(x[:4])
    bad = list(x)
    bad[0] = bad[0][:, :20]
    with pytest.raises(ValueError):
        m.predict(bad)
    bad = list(x)
    bad[1] = bad[1].astype(np.int64) + 30
    with pytest.raises(ValueError):
        m.predict(bad)
    empty = [a[:0] for a in x]
    out = m.predict(empty)
    assert out.shape == (0, 2) and out.dtype == np.float32
    with pytest.raises(IndexError):
        m.predict_protein("ACDK", [5], np.zeros((1, 8)))


def test_encode_sequence_table():
    assert encode_sequence("ACDEFGHIKLMNPQRSTVWY*XB").tolist() == list(range(21)) + [20, 20]


def test_model_weights_roundtrip(tmp_path):
    m = create_ptm_model(seed=3)
    ws = m.get_weights()
    assert len(ws) == len(spec.tensor_table()) and sum(w.size for w in ws) == spec.n_params()
    p = tmp_path / "w.dptm"
    m.save(str(p))
    m2 = load_model(str(p), custom_objects={"PositionEmbedding": object}, compile=False)
    for a, b in zip(ws, m2.get_weights()):
        assert np.array_equal(a, b)
    with pytest.raises(OSError):
        load_model(str(tmp_path / "missing.h5"))
    with pytest.raises(ValueError):
        m.set_weights(ws[:-1])


def test_get_weights_follow_keras_model_weights_order():
    """model.get_weights() of the graph of model.py:210 lists layers by depth: the three Embedding tables
    first, then the three position tables, ...; the head Dense last.  set_weights takes the same order."""
    from deepptmpred_amd.keras_h5 import keras_weight_order
    m = Model(spec.synthetic_weights(4))
    ws = m.get_weights()
    order = keras_weight_order()
    assert order[:6] == ["b21.emb", "b33.emb", "b51.emb", "b21.pos", "b33.pos", "b51.pos"]
    assert order[-2:] == ["head.k", "head.b"]
    assert [w.shape for w in ws[:3]] == [(21, 64)] * 3
    assert [w.shape for w in ws[3:6]] == [(21, 64), (33, 64), (51, 64)]
    assert ws[-1].shape == (2,) and ws[-2].shape == (256, 2)
    # a BatchNormalization layer lists trainable before non-trainable variables
    i = order.index("t1.bn.gamma")
    assert order[i:i + 4] == ["t1.bn.gamma", "t1.bn.beta", "t1.bn.mean", "t1.bn.var"]
    named = m.named_weights()
    assert list(named) == [n for n, _ in spec.tensor_table()]
    for a, n in zip(ws, order):
        assert np.array_equal(a, named[n])
    m2 = Model(spec.synthetic_weights(9))
    m2.set_weights(ws)
    assert all(np.array_equal(m2.named_weights()[k], v) for k, v in named.items())


def test_create_ptm_model_starts_from_the_keras_initial_state():
    """model.py:143-210 under Keras defaults: zero biases, identity BatchNormalization / LayerNormalization,
    Embedding uniform(+-0.05), Glorot-uniform kernels and position tables, head bias [0.1, 1.0]."""
    w = create_ptm_model(seed=0).named_weights()
    assert w["head.b"].tolist() == [np.float32(0.1), np.float32(1.0)]
    for n, a in w.items():
        leaf = n.rsplit(".", 1)[1]
        if leaf in ("gamma", "var", "ln1_g", "ln2_g"):
            assert np.all(a == 1.0), n
        elif leaf in ("beta", "mean", "ln1_b", "ln2_b") or (a.ndim == 1 and n != "head.b"):
            assert np.all(a == 0.0), n
    for p in ("b21", "b33", "b51"):
        assert np.abs(w[f"{p}.emb"]).max() <= 0.05 and np.abs(w[f"{p}.emb"]).max() > 0.045
    glorot = {"b51.pos": (51, 64), "b21.sc_k": (64, 64), "b33.c3a_k": (192, 192), "b51.tb2.ffn1_k": (64, 256),
              "esm1.k": (1280, 512), "t1.k": (576, 512), "head.k": (256, 2), "b21.pool_k": (64, 1)}
    for n, (fi, fo) in glorot.items():
        lim = np.sqrt(6.0 / (fi + fo))
        assert np.abs(w[n]).max() <= lim * (1 + 1e-6) and np.abs(w[n]).max() > 0.9 * lim, n
        assert abs(float(w[n].std()) - lim / np.sqrt(3)) < 0.1 * lim, n
    a = create_ptm_model(seed=1).named_weights()
    assert not np.array_equal(a["t1.k"], w["t1.k"])


def test_low_level_buffers_are_checked_before_the_c_call():
    """The C side reads raw memory, so the binding refuses wrong dtypes, shapes and strided views
    instead of passing their address on (no handle is needed to get that far)."""
    ok = np.zeros((4, 21), dtype=np.uint8)
    assert _native._buffer(ok, np.uint8, (4, 21), "t") is ok
    assert _native._buffer(np.zeros(7, np.uint8), np.uint8, (None,), "t").shape == (7,)
    with pytest.raises(ValueError, match="dtype"):
        _native._buffer(ok.astype(np.int64), np.uint8, (4, 21), "t")
    with pytest.raises(ValueError, match="shape"):
        _native._buffer(ok, np.uint8, (4, 33), "t")
    with pytest.raises(ValueError, match="contiguous"):
        _native._buffer(np.zeros((4, 42), np.uint8)[:, ::2], np.uint8, (4, 21), "t")
    with pytest.raises(ValueError):
        _native._buffer([[0] * 21] * 4, np.uint8, (4, 21), "t")


def test_null_handles_and_arguments_are_refused():
    """Entry points that can be reached without a GPU reject NULL handles / arguments with DPTM_EINVAL."""
    lib = _native.load_library()
    assert lib.dptm_sync(None) == _native.DPTM_EINVAL
    assert lib.dptm_set_precision(None, 0) == _native.DPTM_EINVAL
    assert lib.dptm_score(None, None, None, None, None, None, 4, None) == _native.DPTM_EINVAL
    assert lib.dptm_score_device(None, None, None, None, None, None, 4, None, None) == _native.DPTM_EINVAL
    assert lib.dptm_profile_begin(None) == _native.DPTM_EINVAL
    assert b"NULL handle" in lib.dptm_last_error()
    assert lib.dptm_chunk_sites(None) == 0
    lib.dptm_destroy(None)           # a no-op
    lib.dptm_trainer_destroy(None)
    assert lib.dptm_trainer_param_floats(None) == 0
    assert lib.dptm_create(None, 0, None, 0, None) == _native.DPTM_EINVAL


def test_host_conversion_rounds_like_the_device():
    """dptm_convert_rows: fp32 -> bfloat16 / float16 bit patterns, round to nearest even, against torch's CPU casts
    (the same IEEE conversions the kernels' v_cvt instructions perform): normal values, ties, subnormal halves,
    overflow to infinity, signed zeros; a length that leaves a scalar tail behind the vector loop."""
    import torch
    rng = np.random.default_rng(5)
    x = np.concatenate([
        rng.normal(0, 0.27, 100003), rng.normal(0, 1e-6, 5000), rng.normal(0, 3e4, 5000), rng.uniform(-7e-5, 7e-5, 5000),
        np.array([0.0, -0.0, 1.0, -1.0, 65504.0, 65519.9, 65520.0, 1e38, -1e38, 5.96e-8, 2.98e-8, 2.9802325e-08, 6.1e-5]),
        np.float32(1.0) + np.arange(64, dtype=np.float32) * np.float32(2.0 ** -9),       # bf16 ties and near-ties
        np.float32(1.0) + np.arange(64, dtype=np.float32) * np.float32(2.0 ** -12),      # f16 ties and near-ties
    ]).astype(np.float32)
    t = torch.from_numpy(x)
    want_bf = t.to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
    want_h = t.to(torch.float16).view(torch.int16).numpy().view(np.uint16)
    assert np.array_equal(_native.convert_rows(x, "bf16"), want_bf)
    assert np.array_equal(_native.convert_rows(x, "f16"), want_h)
    # pieces shorter than a vector register go through the scalar code alone
    sub = np.concatenate([x[:3000], x[100003:104000], x[-141:]])
    for mode, want in (("bf16", t.to(torch.bfloat16)), ("f16", t.to(torch.float16))):
        ref = np.concatenate([want.view(torch.int16).numpy().view(np.uint16)[:3000],
                              want.view(torch.int16).numpy().view(np.uint16)[100003:104000],
                              want.view(torch.int16).numpy().view(np.uint16)[-141:]])
        got = np.concatenate([_native.convert_rows(sub[i:i + 7], mode) for i in range(0, len(sub), 7)])
        assert np.array_equal(got, ref), mode
    assert _native.convert_rows(np.zeros((0, 4), np.float32), "bf16").shape == (0, 4)
    with pytest.raises(ValueError):
        _native.convert_rows(x, "f16x3")

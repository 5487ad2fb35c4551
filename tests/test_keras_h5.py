"""Keras-2 HDF5 checkpoints without h5py: the pure-Python HDF5 subset reader/writer
against the real libhdf5 (driven through ctypes when the image has one), and the Keras
layer/weight layout mapping."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import h5_ctypes  # noqa: E402

from deepptmpred_amd import spec  # noqa: E402
from deepptmpred_amd.hdf5_min import H5File, H5FormatError, H5Writer  # noqa: E402
from deepptmpred_amd.keras_h5 import keras_layer_table, read_keras_h5, write_keras_h5  # noqa: E402
from deepptmpred_amd.model import Model, load_model  # noqa: E402

LIB = h5_ctypes.load()
needs_lib = pytest.mark.skipif(LIB is None, reason="no libhdf5 on this machine")


def _tree(n_groups=70):
    rng = np.random.default_rng(0)
    ds = {f"/model_weights/layer_{i}/layer_{i}/kernel:0": rng.normal(size=(3 + i % 5, 5)).astype(np.float32)
          for i in range(n_groups)}
    ds["/model_weights/deep/a/b/c/bias:0"] = rng.normal(size=(7,)).astype(np.float32)
    attrs = {("/model_weights", "layer_names"): [f"layer_{i}".encode() for i in range(n_groups)] + [b"deep"]}
    for i in range(n_groups):
        attrs[(f"/model_weights/layer_{i}", "weight_names")] = [f"layer_{i}/kernel:0".encode()]
    return ds, attrs


def test_writer_reader_roundtrip(tmp_path):
    ds, attrs = _tree()
    w = H5Writer()
    for k, v in ds.items():
        w.create_dataset(k, v)
    for (o, n), v in attrs.items():
        w.set_attr(o, n, v)
    w.set_attr("/", "keras_version", "2.15.0")
    w.set_attr("/", "answer", np.array([1.5, 2.5], dtype=np.float32))
    p = str(tmp_path / "a.h5")
    w.save(p)
    f = H5File(p)
    assert f.attrs["keras_version"] == b"2.15.0" and f.attrs["answer"].tolist() == [1.5, 2.5]
    assert len(f["model_weights"].keys()) == 71
    for k, v in ds.items():
        assert np.array_equal(f[k].read(), v)
    assert [bytes(x) for x in f["model_weights"].attrs["layer_names"]][:2] == [b"layer_0", b"layer_1"]
    with pytest.raises(KeyError):
        f["model_weights/nope"]
    bad = tmp_path / "bad.h5"
    bad.write_bytes(b"not hdf5 at all")
    with pytest.raises(H5FormatError):
        H5File(str(bad))


@needs_lib
def test_reader_reads_what_libhdf5_writes(tmp_path):
    ds, attrs = _tree()
    p = str(tmp_path / "ref.h5")
    h5_ctypes.write_tree(LIB, p, ds, attrs)
    f = H5File(p)
    for k, v in ds.items():
        assert np.array_equal(f[k].read(), v)
    assert [bytes(x).rstrip(b"\x00") for x in f["model_weights"].attrs["layer_names"]] == attrs[("/model_weights", "layer_names")]


@needs_lib
def test_libhdf5_reads_what_writer_writes(tmp_path):
    ds, attrs = _tree()
    w = H5Writer()
    for k, v in ds.items():
        w.create_dataset(k, v)
    for (o, n), v in attrs.items():
        w.set_attr(o, n, v)
    p = str(tmp_path / "mine.h5")
    w.save(p)
    for k, v in ds.items():
        assert np.array_equal(h5_ctypes.read_dataset(LIB, p, k), v)
    got = [x.rstrip(b"\x00") for x in h5_ctypes.read_str_attr(LIB, p, "/model_weights", "layer_names")]
    assert got == attrs[("/model_weights", "layer_names")]


def test_keras_layer_table_covers_every_tensor():
    table = keras_layer_table()
    canon = [c for _, pairs in table for _, c in pairs]
    assert sorted(canon) == sorted(n for n, _ in spec.tensor_table())
    names = [ln for ln, _ in table]
    assert len(names) == len(set(names)) == 40
    # Dense numbering counts the Dense layers inside the transformer blocks
    assert {"dense_12", "dense_25", "dense_38", "dense_39", "dense_44", "conv1d_8", "batch_normalization_9",
            "transformer_block_5", "embedding_2", "position_embedding"} <= set(names)
    tb = dict(table)["transformer_block"]
    assert [c.split(".")[-1] for _, c in tb] == ["wq", "bq", "wk", "bk", "wv", "bv", "wo", "bo", "ffn1_k", "ffn1_b",
                                                  "ffn2_k", "ffn2_b", "ln1_g", "ln1_b", "ln2_g", "ln2_b"]


def test_keras_checkpoint_roundtrip_through_model(tmp_path):
    w = spec.synthetic_weights(11)
    p = str(tmp_path / "ptm_data_210_39_64_best_model.h5")
    Model(w).save(p)
    back = read_keras_h5(p)
    assert all(np.array_equal(back[k], w[k]) for k in w)
    m = load_model(p, custom_objects={"PositionEmbedding": None, "focal_loss": None})
    from deepptmpred_amd.keras_h5 import keras_weight_order
    for a, name in zip(m.get_weights(), keras_weight_order()):
        assert np.array_equal(a, w[name])
    # a save_weights-style file: the same tree at the root
    f = H5File(p)
    assert f.attrs["backend"] == b"tensorflow"
    assert len(f["model_weights"].attrs["layer_names"]) == 40


@needs_lib
def test_read_checkpoint_written_by_libhdf5(tmp_path):
    """The same Keras layout produced by the real library (as h5py would), read by us."""
    w = spec.synthetic_weights(12)
    table = keras_layer_table()
    ds, attrs = {}, {("/model_weights", "layer_names"): [ln.encode() for ln, _ in table]}
    for ln, pairs in table:
        attrs[(f"/model_weights/{ln}", "weight_names")] = [kn.encode() for kn, _ in pairs]
        for kn, canon in pairs:
            ds[f"/model_weights/{ln}/{kn}"] = w[canon]
    p = str(tmp_path / "keras_like.h5")
    h5_ctypes.write_tree(LIB, p, ds, attrs)
    back = read_keras_h5(p)
    assert all(np.array_equal(back[k], w[k]) for k in w)
    # and the real library can read ours
    p2 = str(tmp_path / "ours.h5")
    write_keras_h5(p2, w)
    assert np.array_equal(h5_ctypes.read_dataset(LIB, p2, "/model_weights/dense_40/dense_40/kernel:0"), w["esm1.k"])


def test_wrong_layout_is_reported(tmp_path):
    w = H5Writer()
    w.create_dataset("/model_weights/something/kernel:0", np.zeros((2, 2), np.float32))
    p = str(tmp_path / "other.h5")
    w.save(p)
    with pytest.raises(H5FormatError):
        read_keras_h5(p)


def test_damaged_files_raise_format_error(tmp_path):
    """Truncation, flipped bytes and zeroed blocks end in H5FormatError (or, where the damage missed
    everything the reader touches, in a clean read) -- never in a raw parser exception or a hang."""
    from deepptmpred_amd import keras_h5
    from deepptmpred_amd.hdf5_min import H5FormatError
    w = spec.synthetic_weights(3)
    good = tmp_path / "m.h5"
    keras_h5.write_keras_h5(str(good), w)
    raw = good.read_bytes()
    rng = np.random.default_rng(0)
    bad = tmp_path / "c.h5"
    seen_error = 0
    for trial in range(45):
        b = bytearray(raw)
        if trial % 3 == 0:
            b = b[:int(rng.integers(16, len(b)))]
        elif trial % 3 == 1:
            for _ in range(8):
                b[int(rng.integers(0, 20000))] = int(rng.integers(0, 256))
        else:
            i = int(rng.integers(0, len(b) - 64))
            b[i:i + 64] = bytes(64)
        bad.write_bytes(bytes(b))
        try:
            got = keras_h5.read_keras_h5(str(bad))
            assert set(got) == set(w)
        except H5FormatError:
            seen_error += 1
    assert seen_error >= 10
    with pytest.raises(H5FormatError):
        bad.write_bytes(b"not an hdf5 file at all")
        keras_h5.read_keras_h5(str(bad))


def test_written_layer_order_is_keras_depth_order(tmp_path):
    """Keras-2 load_weights pairs the file's weighted layers with model.layers by position, and a functional
    model's layers are sorted by depth, ties by the walk from the output (functional.py, _map_graph_network).
    Hand-derived for model.py:143-210: depth 26 inputs, 25 embeddings, 24 position embeddings, 23 spatial
    dropouts, 22 first 3-tap convs, 21 their BatchNorms, 20 spatial dropouts, 19 second 3-tap convs, 18 {their
    BatchNorm, the shortcut conv} per branch (Add lists x before shortcut), 17 adds, 16/15 transformer blocks,
    14 pooling Dense, ... 11 {struct_input, ESM Dense 512}, 10 {struct Dense, ESM BatchNorm}, 9 {struct
    BatchNorm}, 8 {ESM Dense 256}, 5 trunk Dense 512, 4 BN, 2 Dense 256, 1 BN, 0 head."""
    from deepptmpred_amd.keras_h5 import keras_model_layers
    want = ["embedding", "embedding_1", "embedding_2", "position_embedding", "position_embedding_1", "position_embedding_2",
            "conv1d_1", "conv1d_4", "conv1d_7", "batch_normalization", "batch_normalization_2", "batch_normalization_4",
            "conv1d_2", "conv1d_5", "conv1d_8",
            "batch_normalization_1", "conv1d", "batch_normalization_3", "conv1d_3", "batch_normalization_5", "conv1d_6",
            "transformer_block", "transformer_block_2", "transformer_block_4",
            "transformer_block_1", "transformer_block_3", "transformer_block_5",
            "dense_12", "dense_25", "dense_38", "dense_40", "dense_39", "batch_normalization_7", "batch_normalization_6",
            "dense_41", "dense_42", "batch_normalization_8", "dense_43", "batch_normalization_9", "dense_44"]
    layers = keras_model_layers()
    assert len(layers) == 77 and len(set(layers)) == 77
    assert layers[:3] == ["seq_input_21", "seq_input_33", "seq_input_51"] and layers[-1] == "dense_44"
    assert layers.index("esm_input") > layers.index("activation_2") and layers.index("struct_input") > layers.index("repeat_vector_2")
    weighted = set(n for n, _ in keras_layer_table())
    assert [n for n in layers if n in weighted] == want
    p = str(tmp_path / "o.h5")
    write_keras_h5(p, spec.synthetic_weights(2))
    got = [bytes(x).rstrip(b"\x00").decode() for x in H5File(p)["model_weights"].attrs["layer_names"]]
    assert got == want


def _renumbered_tree(w, offsets, inner_shuffle=False):
    """The Keras layout with every class's layer numbers shifted (a model built in a session that already held
    layers), layer_names in depth order; optionally the weights inside a transformer block listed in another order."""
    from deepptmpred_amd.keras_h5 import _split_suffix, keras_model_layers
    table = dict(keras_layer_table())

    def ren(name):
        cls, k = _split_suffix(name)
        return f"{cls}_{k + offsets[cls]}" if cls in offsets else name

    def ren_path(kn):
        return "/".join(ren(part) if ":" not in part else part for part in kn.split("/"))

    order = [n for n in keras_model_layers() if n in table]
    ds, attrs = {}, {("/model_weights", "layer_names"): [ren(n).encode() for n in order]}
    for ln in order:
        pairs = list(table[ln])
        if inner_shuffle and ln.startswith("transformer_block"):
            pairs = pairs[8:12] + pairs[:8] + pairs[12:]            # FFN first, then the attention, then the norms
        attrs[(f"/model_weights/{ren(ln)}", "weight_names")] = [ren_path(kn).encode() for kn, _ in pairs]
        for kn, canon in pairs:
            ds[f"/model_weights/{ren(ln)}/{ren_path(kn)}"] = w[canon]
    return ds, attrs


@pytest.mark.parametrize("inner_shuffle", [False, True])
def test_reader_falls_back_to_creation_order_and_shapes(tmp_path, inner_shuffle):
    """A checkpoint whose layers are dense_137.., transformer_block_23.. (not a fresh session) loads to the same
    dict: classes and creation order identify the layers, shapes are checked, and inside a transformer block the
    attention matrices are told apart by their Dense numbers rather than by their place in weight_names."""
    w = spec.synthetic_weights(13)
    offsets = {"dense": 137, "transformer_block": 23, "conv1d": 9, "batch_normalization": 10, "embedding": 3,
               "position_embedding": 3, "multi_head_self_attention": 23, "layer_normalization": 46}
    ds, attrs = _renumbered_tree(w, offsets, inner_shuffle)
    p = str(tmp_path / "renumbered.h5")
    wr = H5Writer()
    for k, v in ds.items():
        wr.create_dataset(k, v)
    for (o, n), v in attrs.items():
        wr.set_attr(o, n, v)
    wr.save(p)
    back = read_keras_h5(p)
    assert set(back) == set(w) and all(np.array_equal(back[k], w[k]) for k in w)
    if LIB is not None:                               # the same file produced by the real library
        p2 = str(tmp_path / "renumbered_lib.h5")
        h5_ctypes.write_tree(LIB, p2, ds, attrs)
        back = read_keras_h5(p2)
        assert all(np.array_equal(back[k], w[k]) for k in w)


def test_fallback_reports_a_wrong_layer_count(tmp_path):
    w = spec.synthetic_weights(13)
    ds, attrs = _renumbered_tree(w, {"dense": 50})
    victim = "/model_weights/dense_94"                 # the head: dense_44 + 50
    ds = {k: v for k, v in ds.items() if not k.startswith(victim + "/")}
    attrs = {k: v for k, v in attrs.items() if k[0] != victim}
    wr = H5Writer()
    for k, v in ds.items():
        wr.create_dataset(k, v)
    for (o, n), v in attrs.items():
        wr.set_attr(o, n, v)
    p = str(tmp_path / "short.h5")
    wr.save(p)
    with pytest.raises(H5FormatError, match="dense"):
        read_keras_h5(p)

"""Many proteins per call (dptm_score_proteins / _device, dptm_assemble_proteins_device) against the
per-protein calls they replace (predict.py:584-612) and against the CPU oracle.  Windows, codes and
row indices are integer work: bit-exact.  Probabilities: 1e-4 absolute (fp32 path)."""
import json
import os

import numpy as np
import pytest

from deepptmpred_amd import spec
from deepptmpred_amd.model import Model, encode_sequence
from oracle import ptm_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-4
LETTERS = "ACDEFGHIKLMNPQRSTVWY"


@pytest.fixture(scope="module")
def weights():
    return spec.synthetic_weights(7)


@pytest.fixture(scope="module")
def model(weights):
    m = Model(weights)
    yield m
    m.close()


def oracle_protein(weights, seq, pos, st, table):
    toks = [O.encode_sequences(O.get_sequence_windows(seq, list(pos), w)) for w in (21, 33, 51)]
    esm = np.zeros((len(pos), 1280))
    if table is not None:
        for i, p in enumerate(pos):
            if p - 1 < len(table):
                esm[i] = table[p - 1]
    with np.errstate(over="ignore"):
        return O.forward(weights, toks + [np.asarray(st, dtype=np.float64), esm], np.float64), toks, esm


def random_protein(rng, L, with_table=True, table_rows=None, site_letters="ST"):
    seq = "".join(rng.choice(list(LETTERS), L))
    pos = [i + 1 for i, c in enumerate(seq) if c in site_letters]
    table = None
    if with_table:
        table = (rng.standard_normal((L if table_rows is None else table_rows, 1280)) * 0.27).astype(np.float32)
    st = np.column_stack([rng.uniform(0, 250, len(pos)), rng.uniform(-180, 180, (len(pos), 2)),
                          np.eye(3)[rng.integers(0, 3, len(pos))], rng.uniform(20, 100, (len(pos), 2))]).astype(np.float32)
    return seq, pos, st.reshape(len(pos), 8), table


def test_shipped_and_edge_proteins_in_one_call(model, weights, golden_dir):
    """AKT1 + tau (shipped sequences and ESM rows) together with edge cases: a 5-residue protein whose every
    window is mostly padding, a protein without a table, one with a table shorter than the sequence, letters
    outside the 20 (code 20), a protein with no sites, sites listed in descending order."""
    g = np.load(os.path.join(golden_dir, "proteins.npz"))
    meta = json.load(open(os.path.join(golden_dir, "proteins.json")))
    rng = np.random.default_rng(11)
    jobs = []
    for pid in ("P31749", "P10636"):
        seq, pos, esm = meta[pid]["sequence"], g[f"{pid}_phospho_positions"], g[f"{pid}_phospho_esm"]
        table = np.zeros((meta[pid]["esm_rows"], 1280), dtype=np.float32)
        table[pos - 1] = esm
        jobs.append((seq, list(pos), np.tile(O.DEFAULT_STRUCT_ROW, (len(pos), 1)), table))
    jobs.append(("STSTS", [1, 2, 3, 4, 5], rng.uniform(0, 90, (5, 8)), (rng.standard_normal((5, 1280)) * 0.3).astype(np.float32)))
    jobs.append(random_protein(rng, 120, with_table=False))
    jobs.append(random_protein(rng, 200, table_rows=150))
    seq_x = "MKXUSBZTOSJ" * 6
    jobs.append((seq_x, [i + 1 for i, c in enumerate(seq_x) if c in "ST"], rng.uniform(0, 90, (18, 8)),
                 (rng.standard_normal((len(seq_x), 1280)) * 0.3).astype(np.float32)))
    jobs.append(("ACDEFG", [], np.zeros((0, 8)), None))
    s, p, st, t = random_protein(rng, 90)
    jobs.append((s, p[::-1], st[::-1].copy(), t))
    got = model.predict_proteins(jobs)
    assert len(got) == len(jobs) and sum(len(a) for a in got) <= 512      # the whole call is one small pass
    for (seq, pos, st, table), p_many in zip(jobs, got):
        assert p_many.shape == (len(pos), 2)
        if not len(pos):
            continue
        alone = model.predict_protein(seq, pos, st, table)
        assert np.array_equal(p_many, alone)                               # same call regime: bit-identical
        ref, _, _ = oracle_protein(weights, seq, pos, st, table)
        assert np.abs(p_many - ref).max() < TOL


def test_large_batch_matches_device_form_and_assembled_inputs(model, weights):
    """> one internal pass of sites: host form (pipelined, host-gathered ESM rows) == device form (row-index
    gather inside the ESM layer) == predict() on windows assembled by the oracle's host code, bit for bit;
    a sample of rows is checked against the oracle."""
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    n_prot = 700
    lens = np.clip(np.rint(rng.normal(500, 300, n_prot)), 30, 2000).astype(np.int64)
    seq_off = np.zeros(n_prot + 1, dtype=np.int64)
    np.cumsum(lens, out=seq_off[1:])
    codes = rng.integers(0, 20, seq_off[-1]).astype(np.uint8)
    codes[rng.integers(0, len(codes), 50)] = 20 + rng.integers(0, 5, 50).astype(np.uint8)   # unknown letters
    rows = lens.copy()
    rows[::7] = 0                       # proteins without a table
    rows[3::11] = rows[3::11] // 2      # tables shorter than the sequence
    esm_off = np.zeros(n_prot + 1, dtype=np.int64)
    np.cumsum(rows, out=esm_off[1:])
    tables = (rng.standard_normal((int(esm_off[-1]), 1280)) * 0.27).astype(np.float32)
    gidx = np.nonzero((codes == 15) | (codes == 16) | (codes == 19))[0]
    prot = (np.searchsorted(seq_off, gidx, side="right") - 1).astype(np.int32)
    pos = (gidx - seq_off[prot] + 1).astype(np.int32)
    perm = rng.permutation(len(gidx))                # sites need not be grouped by protein
    prot, pos = prot[perm], pos[perm]
    n = len(pos)
    assert n > model.scorer.chunk_sites
    st = spec.synthetic_inputs(n, 3)[3]
    host = model.scorer.score_proteins(codes, seq_off, tables, esm_off, pos, prot, st)
    d = {k: torch.from_numpy(v).to(dev) for k, v in dict(codes=codes, seq_off=seq_off, tables=tables, esm_off=esm_off,
                                                         pos=pos, prot=prot, st=st).items()}
    out = torch.empty((n, 2), dtype=torch.float32, device=dev)
    model.scorer.score_proteins_device(d["codes"].data_ptr(), d["seq_off"].data_ptr(), d["tables"].data_ptr(),
                                       d["esm_off"].data_ptr(), n_prot, d["pos"].data_ptr(), d["prot"].data_ptr(),
                                       d["st"].data_ptr(), n, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), host)
    # the same sites with host-side assembly (the reference's indexing restated in NumPy)
    toks = []
    c = np.minimum(codes, 20)
    for w in (21, 33, 51):
        half = (w - 1) // 2
        i = (pos.astype(np.int64) - 1 - half)[:, None] + np.arange(w)[None, :]
        L = (seq_off[prot + 1] - seq_off[prot])[:, None]
        ok = (i >= 0) & (i < L)
        toks.append(np.where(ok, c[np.clip(seq_off[prot][:, None] + i, 0, len(c) - 1)], 20).astype(np.uint8))
    r = pos.astype(np.int64) - 1
    has = r < (esm_off[prot + 1] - esm_off[prot])
    esm = np.zeros((n, 1280), dtype=np.float32)
    esm[has] = tables[(esm_off[prot] + r)[has]]
    assert np.array_equal(model.predict(toks + [st, esm]), host)
    k = np.r_[0:150, n - 150:n]
    with np.errstate(over="ignore"):
        ref = O.forward(weights, [a[k] for a in toks] + [st[k], esm[k]], np.float64)
    assert np.abs(host[k] - ref).max() < TOL


@pytest.mark.parametrize("byte_offset,codes_offset", [(0, 0), (1, 0), (4, 0), (0, 1), (0, 2)])
def test_assemble_proteins_bit_exact(model, byte_offset, codes_offset):
    """Every residue of 40 proteins (lengths 1..300) as a site: windows equal the reference's string code
    (predict.py:529-560) byte for byte and the row index follows predict.py:493; also with output arrays
    that start off a 16-byte boundary (byte stores instead of 16-byte stores), and with a residue array that
    starts off a 4-byte boundary (a fetch per byte instead of per 32-bit word)."""
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(9)
    lens = np.r_[1, 2, 10, 11, 25, 26, 51, rng.integers(1, 300, 33)]
    seqs = ["".join(rng.choice(list(LETTERS + "XB"), int(L))) for L in lens]
    rows = np.array([0 if k % 5 == 0 else (L // 2 if k % 5 == 1 else L) for k, L in enumerate(lens)], dtype=np.int64)
    seq_off = np.r_[0, np.cumsum(lens)].astype(np.int64)
    esm_off = np.r_[0, np.cumsum(rows)].astype(np.int64)
    codes = np.concatenate([encode_sequence(s) for s in seqs])
    prot = np.repeat(np.arange(len(seqs)), lens).astype(np.int32)
    pos = np.concatenate([np.arange(1, L + 1) for L in lens]).astype(np.int32)
    n = len(pos)
    want = [np.concatenate([O.encode_sequences(O.get_sequence_windows(s, list(range(1, len(s) + 1)), w)) for s in seqs])
            for w in (21, 33, 51)]
    want_row = np.where(pos - 1 < rows[prot], esm_off[prot] + pos - 1, -1)
    d = [torch.from_numpy(a).to(dev) for a in (codes, seq_off, esm_off, pos, prot)]
    shifted = torch.full((len(codes) + 8,), 77, dtype=torch.uint8, device=dev)     # 77: not a residue code
    shifted[codes_offset:codes_offset + len(codes)] = d[0]
    codes_ptr = shifted.data_ptr() + codes_offset
    assert shifted.data_ptr() % 4 == 0
    raw = [torch.full((n * w + 8,), 255, dtype=torch.uint8, device=dev) for w in (21, 33, 51)]
    erow = torch.empty(n, dtype=torch.int32, device=dev)
    model.scorer.assemble_proteins_device(codes_ptr, d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(),
                                          n, *[t.data_ptr() + byte_offset for t in raw], erow.data_ptr(),
                                          torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for t, w, exp in zip(raw, (21, 33, 51), want):
        flat = t.cpu().numpy()
        assert np.array_equal(flat[byte_offset:byte_offset + n * w].reshape(n, w), exp.astype(np.uint8))
        assert (flat[:byte_offset] == 255).all() and (flat[byte_offset + n * w:] == 255).all()   # nothing outside
    assert np.array_equal(erow.cpu().numpy(), want_row)


def test_bad_tables_are_rejected(model):
    codes = encode_sequence("ACDKSTACDKST")
    off = np.array([0, 6, 12], dtype=np.int64)
    zero = np.zeros(3, dtype=np.int64)
    st = np.zeros((1, 8), dtype=np.float32)
    one = lambda v: np.array([v], dtype=np.int32)
    sc = model.scorer
    with pytest.raises(ValueError, match="outside"):
        sc.score_proteins(codes, off, None, zero, one(7), one(0), st)
    with pytest.raises(ValueError, match="outside"):
        sc.score_proteins(codes, off, None, zero, one(0), one(1), st)
    with pytest.raises(ValueError, match="names protein"):
        sc.score_proteins(codes, off, None, zero, one(1), one(2), st)
    with pytest.raises(ValueError, match="decrease"):
        sc.score_proteins(codes, np.array([0, 14, 12], dtype=np.int64), None, zero, one(1), one(0), st)
    with pytest.raises(ValueError):
        sc.score_proteins(codes, np.array([0, 6, 11], dtype=np.int64), None, zero, one(1), one(0), st)
    ok = sc.score_proteins(codes, off, None, zero, one(6), one(1), st)         # still works afterwards
    assert ok.shape == (1, 2) and np.isfinite(ok).all()


def test_predictor_batch_table(model, golden_dir, tmp_path):
    """PTMPredictor.predict_ptm_sites_batch == the per-protein loop of predict.py:584-612."""
    import pandas as pd
    from deepptmpred_amd.predictor import PredictConfig, PTMPredictor
    meta = json.load(open(os.path.join(golden_dir, "proteins.json")))
    g = np.load(os.path.join(golden_dir, "proteins.npz"))
    cfg = PredictConfig("phosphorylation", str(tmp_path))
    os.makedirs(cfg.custom_esm_dir)
    pos = g["P10636_phospho_positions"]
    table = np.zeros((meta["P10636"]["esm_rows"], 1280))
    table[pos - 1] = g["P10636_phospho_esm"]
    np.savez(os.path.join(cfg.custom_esm_dir, "P10636_full_esm.npz"), features=table)
    pred = PTMPredictor(cfg, model=model)
    prots = [{"protein_id": "P10636", "protein_sequence": meta["P10636"]["sequence"]},
             {"protein_id": "P31749", "protein_sequence": meta["P31749"]["sequence"]},          # no ESM file: zero rows
             {"protein_id": "tiny", "protein_sequence": "MSTK", "positions": [2, 3]}]
    batch = pred.predict_ptm_sites_batch(prots)
    loop = pd.concat([pred.predict_ptm_sites(p["protein_id"], p["protein_sequence"], p.get("positions")) for p in prots],
                     ignore_index=True)
    assert batch["protein_id"].tolist() == loop["protein_id"].tolist()
    assert batch["position"].tolist() == loop["position"].tolist() and batch["residue"].tolist() == loop["residue"].tolist()
    assert np.array_equal(batch["probability"].to_numpy(), loop["probability"].to_numpy())
    assert batch["prediction"].tolist() == loop["prediction"].tolist()
    assert len(batch) == 80 + 53 + 2

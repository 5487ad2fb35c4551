"""Half-precision-MFMA modes of the HIP path against the CPU oracle (through the C ABI).

"f16x3" (operands split into two float16 values, three MFMAs per product, fp32 accumulation)
must meet the same 1e-4 bound on probabilities as the fp32 path.  "f16" and "bf16" round every
operand once; their bounds below are looser and stated per mode -- they are opt-in speed modes,
not parity modes."""
import numpy as np
import pytest

from deepptmpred_amd import spec
from deepptmpred_amd.model import Model
from oracle import ptm_oracle as O

pytestmark = pytest.mark.gpu

# absolute bound on probabilities, per mode
BOUND = {"f32": 1e-4, "f16x3": 1e-4, "f16": 1e-2, "bf16": 6e-2}


def oracle(weights, x):
    with np.errstate(over="ignore"):
        return O.forward(weights, x, np.float64)


@pytest.fixture(scope="module")
def weights():
    return spec.synthetic_weights(7)


@pytest.fixture()
def model(weights):
    m = Model(weights)
    yield m
    m.close()


@pytest.mark.parametrize("mode", ["f16x3", "f16", "bf16"])
@pytest.mark.parametrize("n", [1, 5, 13, 257, 2000])
def test_mode_vs_oracle(model, weights, mode, n):
    x = spec.synthetic_inputs(n, 4000 + n)
    ref = oracle(weights, x)
    model.scorer.set_precision(mode)
    p = model.predict(x)
    err = np.abs(p - ref).max()
    print(mode, n, "max abs err %.3e" % err)
    assert np.isfinite(p).all()
    assert err < BOUND[mode]
    assert np.allclose(p.sum(axis=1), 1.0, atol=1e-6)


def test_f16x3_meets_parity_bound_on_other_weights():
    for seed in (100, 101):
        w = spec.synthetic_weights(seed)
        m = Model(w)
        m.scorer.set_precision("f16x3")
        x = spec.synthetic_inputs(300, seed)
        assert np.abs(m.predict(x) - oracle(w, x)).max() < BOUND["f16x3"]
        m.close()


def test_f16x3_edges_and_padding(model, weights):
    x = spec.synthetic_inputs(16, 8)
    x[0][:] = 20; x[1][:] = 20; x[2][:] = 20           # every window entirely '*'
    x[3][:] = np.array([0.5, 0, 0, 0, 1, 0, 85, 85], dtype=np.float32)   # predict.py:526
    x[4][:8] = 0.0                                      # missing-ESM rows (predict.py:500)
    model.scorer.set_precision("f16x3")
    assert np.abs(model.predict(x) - oracle(weights, x)).max() < BOUND["f16x3"]


@pytest.mark.parametrize("mode", ["f16x3", "bf16"])
def test_mode_rows_independent(model, mode):
    """Per-site scoring in every mode.  A prefix of the batch keeps every site at the same place of
    its 256-row tile, so its rows are bit-identical; a permutation moves sites to other offsets of
    the aligned 32-key attention tiles, which changes the fp32 summation order only."""
    model.scorer.set_precision(mode)
    n = 3000
    x = spec.synthetic_inputs(n, 78)
    p = model.predict(x)
    assert np.array_equal(model.predict([a[:777] for a in x]), p[:777])
    perm = np.random.default_rng(1).permutation(n)
    q = model.predict([a[perm] for a in x])
    d = np.abs(q - p[perm]).max()
    print(mode, "max abs difference under permutation %.3e" % d)
    assert d < (2e-5 if mode == "f16x3" else BOUND[mode])


def test_mode_switch_round_trip(model, weights):
    """Switching modes on one handle leaves the fp32 results bit-identical."""
    x = spec.synthetic_inputs(100, 5)
    a = model.predict(x)
    model.scorer.set_precision("bf16")
    model.predict(x)
    model.scorer.set_precision("f32")
    assert np.array_equal(model.predict(x), a)
    with pytest.raises(ValueError):
        model.scorer.set_precision("fp8")


@pytest.mark.parametrize("mode", ["bf16", "f16"])
def test_host_pipeline_sends_half_rows_and_scores_alike(model, mode):
    """A host-array call larger than one pass sends the ESM rows over PCIe as the mode's 16-bit values (converted on
    the host, round to nearest even): every probability equals, bit for bit, what the device-resident call gives for
    the fp32 rows (which the kernel rounds itself).  Three passes with a ragged tail.  The many-proteins entry point takes the same path for its gathered rows."""
    import torch
    sc = model.scorer
    sc.set_precision(mode)
    c = sc.chunk_sites
    n = 2 * c + 1333      # (a last pass of at most 512 sites would take the small-call kernels, whose dense layers stay fp32)
    small = spec.synthetic_inputs(4096, 93)
    small[4][5, :6] = [0.0, -0.0, 65519.0, 1e-7, 3e-8, 1.0009765625]      # ties, subnormal halves, the largest half
    reps = -(-n // 4096)
    x = [np.tile(a, (reps, 1))[:n].copy() for a in small]
    p = model.predict(x)                                   # pipeline: half rows over PCIe
    # the device-resident call rounds the fp32 rows itself; it is made once per pass of the host pipeline, so that
    # every site sits at the same place of its tile in both (that place decides the order of the attention sums
    # in these modes)
    dev = torch.device("cuda:0")
    d = [torch.from_numpy(a).to(dev) for a in x]
    out = torch.empty((n, 2), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    for c0 in range(0, n, c):
        m_ = min(c, n - c0)
        sc.score_device(d[0][c0:].data_ptr(), d[1][c0:].data_ptr(), d[2][c0:].data_ptr(), d[3][c0:].data_ptr(),
                        d[4][c0:].data_ptr(), m_, out[c0:].data_ptr(), 0)
    sc.sync()
    assert np.isfinite(p).all() and np.array_equal(p, out.cpu().numpy())
    del d, out
    # a caller that keeps its rows as 16-bit values (dptm_convert_rows) hands them over as they are: same probabilities,
    # for the pipelined call and for a one-pass call of more than 512 sites
    from deepptmpred_amd import _native
    x16 = _native.convert_rows(x[4], mode)
    assert np.array_equal(sc.score_rows16(x[0], x[1], x[2], x[3], x16), p)
    k = 700
    assert np.array_equal(sc.score_rows16(x[0][:k], x[1][:k], x[2][:k], x[3][:k], np.ascontiguousarray(x16[:k])),
                          sc.score(x[0][:k], x[1][:k], x[2][:k], x[3][:k], np.ascontiguousarray(x[4][:k])))
    sc.set_precision("f32")
    with pytest.raises(ValueError):
        sc.score_rows16(x[0][:k], x[1][:k], x[2][:k], x[3][:k], np.ascontiguousarray(x16[:k]))      # fp32 mode has no use for them
    sc.set_precision(mode)
    # many proteins per call: 40,000 sites of 200 proteins, rows gathered (and converted) by the host, against the
    # device-resident form of the same call (tables resident as fp32, rows gathered inside the ESM layer)
    from deepptmpred_amd.model import encode_sequence
    L, P, per = 300, 200, 200
    rng = np.random.default_rng(7)
    codes = np.concatenate([encode_sequence("".join(rng.choice(list("ACDEFGHIKLMNPQRSTVWY"), L))) for _ in range(P)])
    tables = rng.normal(0, 0.27, (P * L, 1280)).astype(np.float32)
    offs = np.arange(P + 1, dtype=np.int64) * L
    pos = np.concatenate([np.sort(rng.choice(np.arange(1, L + 1), per, replace=False)) for _ in range(P)]).astype(np.int32)
    prot = np.repeat(np.arange(P, dtype=np.int32), per)
    st = np.ascontiguousarray(np.tile(O.DEFAULT_STRUCT_ROW, (P * per, 1)), dtype=np.float32)
    got = sc.score_proteins(codes, offs, tables, offs.copy(), pos, prot, st)
    dv = [torch.from_numpy(a).to(dev) for a in (codes, offs, tables, offs.copy(), pos, prot, st)]
    out = torch.empty((P * per, 2), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    for c0 in range(0, P * per, c):
        m_ = min(c, P * per - c0)
        sc.score_proteins_device(dv[0].data_ptr(), dv[1].data_ptr(), dv[2].data_ptr(), dv[3].data_ptr(), P,
                                 dv[4][c0:].data_ptr(), dv[5][c0:].data_ptr(), dv[6][c0:].data_ptr(), m_, out[c0:].data_ptr(), 0)
    sc.sync()
    assert np.isfinite(got).all() and np.array_equal(got, out.cpu().numpy())

"""dptm_score_device_noalloc (static device workspace, no allocation) against dptm_score_device through the
C ABI: bit-identical probabilities on two calls in a row, the second smaller than the first; sizes beyond
the workspace are refused."""
import ctypes as C

import numpy as np
import pytest

from deepptmpred_amd import _native, spec

pytestmark = pytest.mark.gpu


def test_noalloc_entry_matches_score_device():
    import torch
    lib = _native.load_library()
    dev = torch.device("cuda:0")
    sc = _native.Scorer(spec.pack_blob(spec.synthetic_weights(7)))
    cap = int(lib.dptm_noalloc_max_sites())
    assert cap == 4096
    # NULL = the handle's own (non-blocking) stream, which torch's fills on its default stream are not ordered with:
    # the inputs and the pre-filled outputs are complete (synchronize) before either call is enqueued
    stream = C.c_void_p(None)
    side = torch.cuda.Stream(device=dev)

    def both(n, seed, on_side_stream=False):
        st = C.c_void_p(side.cuda_stream) if on_side_stream else stream
        with torch.cuda.stream(side if on_side_stream else torch.cuda.current_stream()):
            x = [torch.from_numpy(a).to(dev) for a in spec.synthetic_inputs(n, seed)]
            a = torch.full((n, 2), -1.0, dtype=torch.float32, device=dev)
            b = torch.full((n, 2), -2.0, dtype=torch.float32, device=dev)
            p = [t.data_ptr() for t in x]
            if not on_side_stream:
                torch.cuda.synchronize()
            # on the caller's stream the fills, both calls and the read-back are ordered by that stream alone
            assert lib.dptm_score_device(sc._h, p[0], p[1], p[2], p[3], p[4], n, a.data_ptr(), st) == 0
            if not on_side_stream:
                torch.cuda.synchronize()
            rc = lib.dptm_score_device_noalloc(sc._h, p[0], p[1], p[2], p[3], p[4], n, b.data_ptr(), st)
            assert rc == 0, lib.dptm_last_error()
            if not on_side_stream:
                torch.cuda.synchronize()
            ah, bh = a.cpu().numpy(), b.cpu().numpy()
        return ah, bh

    a, b = both(cap, 31)                      # fills the workspace
    assert np.isfinite(a).all() and np.abs(a.sum(axis=1) - 1).max() < 1e-6
    assert np.array_equal(a, b)
    a, b = both(1000, 32)                     # smaller: the workspace still holds the first call's rows
    assert np.array_equal(a, b)
    a, b = both(37, 33)                       # a protein's worth of sites: the small-call kernels, here on one stream
    assert np.array_equal(a, b)
    a, b = both(2000, 35, on_side_stream=True)   # a caller-supplied stream orders everything
    assert np.isfinite(a).all() and np.array_equal(a, b)
    for mode in ("bf16", "f16x3"):               # the half-precision kernels through the same workspace
        sc.set_precision(mode)
        a, b = both(1500, 36)
        assert np.isfinite(a).all() and np.abs(a.sum(axis=1) - 1).max() < 1e-5
        assert np.array_equal(a, b), mode
    sc.set_precision("f32")
    x = [torch.from_numpy(t).to(dev) for t in spec.synthetic_inputs(cap + 1, 34)]
    out = torch.empty((cap + 1, 2), dtype=torch.float32, device=dev)
    p = [t.data_ptr() for t in x]
    rc = lib.dptm_score_device_noalloc(sc._h, p[0], p[1], p[2], p[3], p[4], cap + 1, out.data_ptr(), stream)
    assert rc == _native.DPTM_EINVAL and b"at most 4096" in lib.dptm_last_error()
    sc.close()

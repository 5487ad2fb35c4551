"""Host arrays in, host arrays out (Model.predict) per precision mode: sites per second, PCIe included.
usage: [DPTM_HOST_THREADS=k] python tools/streamed_rate.py [sites] [modes...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from deepptmpred_amd import spec
from deepptmpred_amd.model import Model
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
modes = sys.argv[2:] or ["f32", "f16x3", "f16", "bf16"]
m = Model(spec.synthetic_weights(7))
x = spec.synthetic_inputs(n, 78)
for mode in modes:
    m.scorer.set_precision(mode)
    m.predict([a[:4096] for a in x]); m.predict(x)
    best = 1e9
    for _ in range(3):
        t = time.perf_counter(); p = m.predict(x); best = min(best, time.perf_counter() - t)
    assert np.isfinite(p).all()
    print("threads", os.environ.get("DPTM_HOST_THREADS", "default"), mode, "%.3e sites/s" % (n / best), "%.1f GB/s of fp32 input" % (n * 5261 / best / 1e9))
    if mode in ("bf16", "f16"):      # the caller keeps its ESM rows as 16-bit values (dptm_score_rows16)
        from deepptmpred_amd import _native
        x16 = _native.convert_rows(x[4], mode)
        q = m.scorer.score_rows16(x[0], x[1], x[2], x[3], x16)
        assert np.array_equal(p, q)
        best = 1e9
        for _ in range(3):
            t = time.perf_counter(); q = m.scorer.score_rows16(x[0], x[1], x[2], x[3], x16); best = min(best, time.perf_counter() - t)
        print("threads", os.environ.get("DPTM_HOST_THREADS", "default"), mode, "rows16 %.3e sites/s" % (n / best), "%.1f GB/s of 16-bit input" % (n * (2560 + 141) / best / 1e9))

"""PCIe-inclusive rate of the host-buffer entry point (dptm_score through Model.predict)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from deepptmpred_amd import spec
from deepptmpred_amd.model import Model
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
m = Model(spec.synthetic_weights(7))
x = spec.synthetic_inputs(n, 3)
for mode in ("f32", "f16x3", "bf16"):
    m.scorer.set_precision(mode)
    m.predict([a[:40000] for a in x])
    t0 = time.perf_counter()
    p = m.predict(x)
    dt = time.perf_counter() - t0
    print(mode, "host-buffer path: %.0f sites/s (%.1f ms for %d sites, %.2f GB/s of inputs)" % (n / dt, dt * 1e3, n, n * 5261 / dt / 1e9))

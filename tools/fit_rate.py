"""Steps per second of Model.fit (host arrays in, the Keras-style loop) at the reference's batch size.
usage: python tools/fit_rate.py [sites] [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from deepptmpred_amd import spec, create_ptm_model
from deepptmpred_amd.training import AdamW
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12800
b = int(sys.argv[2]) if len(sys.argv) > 2 else 64
x = spec.synthetic_inputs(n, 3)
y = (np.random.default_rng(1).random(n) < 0.05).astype(np.int64)
m = create_ptm_model(None)
m.compile(optimizer=AdamW(learning_rate=5e-4, weight_decay=1e-4))
m.fit(x, y, batch_size=b, epochs=1, verbose=0)          # warm-up epoch (trainer creation, first launches)
t0 = time.perf_counter()
h = m.fit(x, y, batch_size=b, epochs=2, verbose=0)
dt = time.perf_counter() - t0
steps = 2 * ((n + b - 1) // b)
print("Model.fit: %d steps of %d sites in %.2f s = %.3f ms per step (%.0f sites/s); loss %s" % (steps, b, dt, dt / steps * 1e3, 2 * n / dt, h.history["loss"]))

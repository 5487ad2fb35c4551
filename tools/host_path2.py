import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from deepptmpred_amd import spec
from deepptmpred_amd.model import Model
m = Model(spec.synthetic_weights(7))
x = spec.synthetic_inputs(262144, 3)
print([a.dtype for a in x])
sc = m.scorer
sc.score(*[a[:70000] for a in x])
for k in (1, 2, 4, 8):
    n = 32768 * k
    xs = [np.ascontiguousarray(a[:n]) for a in x]
    for rep in range(2):
        t0 = time.perf_counter()
        sc.score(*xs)
        dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    m.predict(xs)
    dt2 = time.perf_counter() - t0
    print(k, "chunks: score %.1f ms, predict %.1f ms" % (dt * 1e3, dt2 * 1e3))

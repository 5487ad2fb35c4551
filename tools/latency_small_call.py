import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from deepptmpred_amd import spec
from deepptmpred_amd.model import Model
w = spec.synthetic_weights(7)
m = Model(w); m.predict(spec.synthetic_inputs(8, 1)); m.close()
x = spec.synthetic_inputs(53, 1)
for rep in range(3):
    m = Model(w)
    t0 = time.perf_counter(); sc = m.scorer; t1 = time.perf_counter()
    m.predict(x); t2 = time.perf_counter()
    m.predict(x); t3 = time.perf_counter()
    m.close(); t4 = time.perf_counter()
    print("create %.1f ms, first predict %.1f ms, second %.2f ms, close %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3))

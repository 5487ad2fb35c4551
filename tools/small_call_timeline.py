"""Latency of a 53-site host-array call (the reference's own case: AKT1's S/T sites) and of its device-resident form.
usage: python tools/small_call_timeline.py [sites]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from deepptmpred_amd import spec
from deepptmpred_amd.model import Model
n = int(sys.argv[1]) if len(sys.argv) > 1 else 53
m = Model(spec.synthetic_weights(7))
x = spec.synthetic_inputs(n, 1)
for _ in range(5):
    m.predict(x)
t = []
for _ in range(200):
    t0 = time.perf_counter(); m.predict(x); t.append(time.perf_counter() - t0)
t.sort()
print("host arrays, %d sites: median %.1f us, best %.1f us" % (n, t[100] * 1e6, t[0] * 1e6))
dev = torch.device("cuda:0")
d = [torch.from_numpy(a).to(dev) for a in x]
out = torch.empty((n, 2), dtype=torch.float32, device=dev)
sc = m.scorer
torch.cuda.synchronize()
for _ in range(5):
    sc.score_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(), n, out.data_ptr(), 0)
sc.sync()
t = []
for _ in range(200):
    t0 = time.perf_counter()
    sc.score_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(), n, out.data_ptr(), 0)
    sc.sync()
    t.append(time.perf_counter() - t0)
t.sort()
print("device-resident, %d sites: median %.1f us, best %.1f us" % (n, t[100] * 1e6, t[0] * 1e6))
# back to back without waiting in between: the device time per call
t0 = time.perf_counter()
for _ in range(500):
    sc.score_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(), n, out.data_ptr(), 0)
sc.sync()
print("device-resident, back to back: %.1f us per call" % ((time.perf_counter() - t0) / 500 * 1e6))

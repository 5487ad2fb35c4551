"""Host time to enqueue one training step versus the step's device time (is the step launch-bound on the host?).
usage: python tools/train_host_enqueue.py [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from deepptmpred_amd import spec, _native
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
x = spec.synthetic_inputs(n, 3)
y = (np.random.default_rng(1).random(n) < 0.05).astype(np.int32)
dev = torch.device("cuda:0")
d = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in x]
lab = torch.from_numpy(y).to(dev)
tr = _native.Trainer(spec.pack_blob(spec.synthetic_weights(7)), max_batch=n)
o = _native.TrainOpts()
o.seed = 1; o.rate_embed, o.rate_conv, o.rate_tb, o.rate_dense = 1e-4, 0.3, 0.1, 0.3
o.n_global = n; o.n_pos_global = int(y.sum()); o.l2 = 1e-4; o.l2_scale = 1.0
s = torch.cuda.Stream(device=dev)
st = s.cuda_stream
def step():
    tr.forward_backward(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(), lab.data_ptr(), n, o, False, st)
    tr.apply(5e-4, 1e-4, 1.0, 1.0, st)
for _ in range(10):
    step()
torch.cuda.synchronize()
K = 100
t0 = time.perf_counter()
for _ in range(K):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("batch %d: host enqueue %.1f us per step, device done %.1f us per step (queue drained %.1f us after the last enqueue)"
      % (n, (t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6, (t2 - t1) * 1e6))
# one step at a time: enqueue, then wait
lat = []
for _ in range(30):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    lat.append(((t1 - t0) * 1e6, (t2 - t0) * 1e6))
print("isolated steps: enqueue %.1f us, enqueue + device %.1f us (medians)" % tuple(np.median(np.array(lat), axis=0)))

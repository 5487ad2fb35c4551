"""Wall time of the training step's forward_backward call (no optimizer step, no checks) for A/B builds selected with
DPTM_LIBRARY; DPTM_TRAIN_SERIAL=1 keeps the branches on one stream.  usage: python tools/train_kernel_times.py [batch] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from deepptmpred_amd import spec, _native
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
w = spec.synthetic_weights(7)
x = spec.synthetic_inputs(n, 3)
y = (np.random.default_rng(1).random(n) < 0.05).astype(np.int32)
dev = torch.device("cuda:0")
d = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in x]
lab = torch.from_numpy(y).to(dev)
tr = _native.Trainer(spec.pack_blob(w), max_batch=n)
o = _native.TrainOpts()
o.seed = 1; o.rate_embed, o.rate_conv, o.rate_tb, o.rate_dense = 1e-4, 0.3, 0.1, 0.3
o.n_global = n; o.n_pos_global = int(y.sum()); o.l2 = 1e-4; o.l2_scale = 1.0
def run():
    tr.forward_backward(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(), lab.data_ptr(), n, o, False)
for _ in range(2):
    run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    run()
torch.cuda.synchronize()
print("batch %d: %.3f ms per forward_backward (%s)" % (n, (time.perf_counter() - t0) / reps * 1e3, os.environ.get("DPTM_LIBRARY", "in-tree library")))

"""Per-kernel clocks of one training step (dptm_trainer_profile_begin / _end) for A/B builds selected with DPTM_LIBRARY.
usage: [DPTM_LIBRARY=...] python tools/train_profile_ab.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from deepptmpred_amd import spec, _native
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
x = spec.synthetic_inputs(n, 3)
y = (np.random.default_rng(1).random(n) < 0.05).astype(np.int32)
dev = torch.device("cuda:0")
d = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in x]
lab = torch.from_numpy(y).to(dev)
tr = _native.Trainer(spec.pack_blob(spec.synthetic_weights(7)), max_batch=n)
o = _native.TrainOpts()
o.seed = 1; o.rate_embed, o.rate_conv, o.rate_tb, o.rate_dense = 1e-4, 0.3, 0.1, 0.3
o.n_global = n; o.n_pos_global = int(y.sum()); o.l2 = 1e-4; o.l2_scale = 1.0
def run():
    tr.forward_backward(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(), lab.data_ptr(), n, o, False)
for _ in range(3):
    run()
torch.cuda.synchronize()
acc = {}
for _ in range(3):
    tr.profile_begin(); run(); ms = tr.profile_end()
    for k, v in ms.items():
        acc[k] = acc.get(k, 0.0) + v / 3
print(os.environ.get("DPTM_LIBRARY", "in-tree library"), "batch", n)
for k in _native.Trainer.PROFILE_KERNELS:
    print("   %-12s %s" % (k, "  ".join("%7.4f" % acc[f"{k}{w}"] for w in (21, 33, 51))), " ms (W = 21, 33, 51)")
print("   sum %.3f ms" % sum(acc.values()))

import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from deepptmpred_amd import _native, spec
from oracle.train_ref import TrainRef
w = spec.synthetic_weights(7)
n = 48
x = spec.synthetic_inputs(n, 321)
y = (np.random.default_rng(5).random(n) < 0.25).astype(np.int64)
ref = TrainRef(w)
rates = dict(embed=0, conv=0, tb=0, dense=0)
proba = ref.forward(x, 99, rates)
loss = ref.loss(proba, y)
loss.backward()
g_ref = {k: ref.p[k].grad.numpy() for k in ref.trainable}
dev = torch.device("cuda:0")
d = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in x]
lab = torch.from_numpy(y.astype(np.int32)).to(dev)
o = _native.TrainOpts(); o.seed=99; o.n_global=n; o.n_pos_global=int(y.sum()); o.l2=1e-4; o.l2_scale=1.0
tr = _native.Trainer(spec.pack_blob(w), max_batch=64)
l = tr.forward_backward(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(), lab.data_ptr(), n, o)
torch.cuda.synchronize()
p = tr.get_proba(n)
print("loss", l, float(loss), "proba max err", np.abs(p - proba.detach().numpy()).max())
got = spec.unpack_blob(tr.get_grads())
for k, g in g_ref.items():
    if k.startswith(("b33", "b51")): continue
    sc = max(np.abs(g).max(), 1e-6)
    e = np.abs(got[k]-g)
    print(f"{k:18s} rel {e.max()/sc:.2e}  at {np.unravel_index(e.argmax(), e.shape)}  ratio-median {np.median(got[k].ravel()[np.abs(g.ravel())>0.1*sc]/g.ravel()[np.abs(g.ravel())>0.1*sc]):.5f}")

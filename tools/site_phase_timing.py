"""Per-phase clocks of the per-site layer kernels of a small batch (debug build: `make -C deepptmpred_amd/csrc phase`).
usage: DPTM_LIBRARY=$PWD/deepptmpred_amd/libdptm_phase.so python tools/site_phase_timing.py [batch]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from deepptmpred_amd import spec, _native
names = ["start-up", "operand loads", "wait for the other wavefronts", "bias / swish / statistics / store", "masks + products",
         "LDS reduction", "upstream gradient read", "dropout masks, gradient out", "BatchNorm backward sums", "d(a)",
         "bias gradient", "weight gradient"]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
lib = _native.load_library()
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
buf = (C.c_ulonglong * 32)()
for it in range(2):
    tr.forward_backward(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(), lab.data_ptr(), n, o)
    torch.cuda.synchronize()
    lib.dptm_debug_phase_cycles_site(buf, 1)
v = np.array(list(buf)[:12], dtype=np.float64)
print("batch", n, ": share of wavefront time per phase, all per-site layer kernels of one step")
for nm, c in zip(names, v):
    print("   %-36s %5.1f %%  %12.0f" % (nm, 100 * c / v.sum(), c))

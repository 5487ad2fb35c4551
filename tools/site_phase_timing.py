"""Per-phase clocks of the small-batch per-site layer kernels (debug build: `make -C deepptmpred_amd/csrc phase`).
usage: DPTM_LIBRARY=$PWD/deepptmpred_amd/libdptm_phase.so python tools/site_phase_timing.py [batch]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from deepptmpred_amd import spec, _native
names = ["fwd start-up", "product: operand loads", "barrier after product", "fwd epilogue", "product: masks + MFMA", "product: LDS reduction",
         "bwd upstream read", "bwd masks / gradient out", "bwd BatchNorm sums", "bwd d(a)", "bwd bias gradient", "bwd weight gradient"]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
lib = _native.load_library()
x = spec.synthetic_inputs(n, 3)
y = (np.random.default_rng(1).random(n) < 0.05).astype(np.int32)
dev = torch.device("cuda:0")
d = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in x]
lab = torch.from_numpy(y).to(dev)
tr = _native.Trainer(spec.pack_blob(spec.synthetic_weights(7)), max_batch=n)
o = _native.TrainOpts()
o.seed = 1; o.rate_embed, o.rate_conv, o.rate_tb, o.rate_dense = 1e-4, 0.3, 0.1, 0.3
o.n_global = n; o.n_pos_global = int(y.sum()); o.l2 = 1e-4; o.l2_scale = 1.0
buf = (C.c_ulonglong * 32)()
for it in range(3):
    tr.forward_backward(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(), lab.data_ptr(), n, o)
    torch.cuda.synchronize()
    lib.dptm_debug_phase_cycles_site(buf, 1)
v = np.array(list(buf)[:12], dtype=np.float64)
# wavefronts per step: 8 per workgroup; forward layers N = 128, 512, 256, 512, 256 -> 52 workgroups, backward 256, 512, 192, 256, 128, 512 -> 58
print("batch", n, ": s_memtime ticks per wavefront and launch (100 MHz clock: 100 ticks = 1 us), averaged over the layers")
for i, (nm, c) in enumerate(zip(names, v)):
    waves = 8 * (52 if i in (0, 3) else 58 if i >= 6 else 110)
    print("   %-26s %8.0f ticks  (%4.1f %% of all)" % (nm, c / waves, 100 * c / v.sum()))

"""What the Model.fit loop adds to the 64-site training step: the step alone on fixed device buffers ("step"), the step
on gathered rows of the resident data set ("rows": dptm_trainer_forward_backward_rows), and with the loss /
probability outputs as DataParallelTrainer.fit asks for them ("all").
usage: python tools/fit_overhead.py [batch] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from deepptmpred_amd import spec
from deepptmpred_amd.training import DropoutRates, NativeBackend

b = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
n = 12800
x = spec.synthetic_inputs(n, 3)
y = (np.random.default_rng(1).random(n) < 0.05).astype(np.int32)
be = NativeBackend(spec.synthetic_weights(1), max_batch=b)
be.set_dataset(x, y)
rates = DropoutRates(1e-4, 0.3, 0.1, 0.3)
order = torch.from_numpy(np.random.default_rng(2).permutation(n).astype(np.int64)).cuda()
losses = be.new_loss_buffer(steps)
proba = be.new_loss_buffer(2 * n)


def run(mode):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        s = (k * b) % (n - b)
        idx = order[s:s + b]
        if mode == "step":            # same staged rows every step: no gathers, no copies
            d = be._staging()
            o = be_opts(k)
            with torch.cuda.stream(be.stream):
                be.trainer.forward_backward(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(),
                                            d[5].data_ptr(), b, o, False, be.stream.cuda_stream)
        elif mode == "rows":
            be.forward_backward_resident(idx, b, 3, k, rates, 1e-4, 1.0, None)
        else:
            be.forward_backward_resident(idx, b, 3, k, rates, 1e-4, 1.0, losses[k:k + 1], proba[2 * s:2 * (s + b)])
        be.apply(5e-4, 1e-4)
    with torch.cuda.stream(be.stream):
        be.stream.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def be_opts(k):
    from deepptmpred_amd import _native
    o = _native.TrainOpts()
    o.seed = k
    o.rate_embed, o.rate_conv, o.rate_tb, o.rate_dense = 1e-4, 0.3, 0.1, 0.3
    o.n_global, o.n_pos_global, o.l2, o.l2_scale = b, 3, 1e-4, 1.0
    return o


be.forward_backward([a[:b] for a in x], y[:b], b, 3, 0, rates, 1e-4, 1.0)       # fills the host path's staging buffers
for mode in ("step", "rows", "all", "step", "rows", "all"):
    print("%-7s %.3f ms per step" % (mode, run(mode)))

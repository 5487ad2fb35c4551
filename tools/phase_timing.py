"""Per-phase cycles of the branch kernels (debug build with -DDPTM_PHASE_TIMING, selected
through DPTM_LIBRARY).  usage: DPTM_LIBRARY=$PWD/deepptmpred_amd/libdptm_phase.so python tools/phase_timing.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from deepptmpred_amd import spec, _native
from deepptmpred_amd.model import Model
names = ["conv2 (+barrier) [f32: whole conv block]", "x frags", "barrier A", "K,V,Q proj", "barrier B", "attention", "O proj+LN1", "FFN+LN2", "pooling", "setup+embedding", "barrier 0", "conv1"]
m = Model(spec.synthetic_weights(7))
lib = _native.load_library()
x = spec.synthetic_inputs(32768, 3)
buf = (C.c_ulonglong * 32)()
for mode in ("f32", "bf16", "f16x3"):
    m.scorer.set_precision(mode)
    m.predict(x)
    rd = lib.dptm_debug_phase_cycles_f32 if mode == "f32" else lib.dptm_debug_phase_cycles_half
    rd(buf, 1)
    m.predict(x)
    rd(buf, 1)
    v = np.array(list(buf)[:12], dtype=np.float64)
    print(mode, "share of wavefront time per phase, all three branches (s_memtime ticks):")
    for n, c in zip(names, v):
        print("   %-14s %5.1f %%" % (n, 100 * c / v.sum()))
    w = np.array(list(buf)[16:32], dtype=np.float64)
    print("   per wavefront index, %% of all time: barrier waits", np.round(100 * w[:8] / v.sum(), 2), "attention", np.round(100 * w[8:] / v.sum(), 2))
    tiles = -(-32768 // 12) + -(-32768 // 7) + -(-32768 // 5)
    print("   ticks per wavefront-tile: %.0f  (tiles %d; compare with the kernels' tile period)" % (v.sum() / (tiles * 8), tiles))
    m.scorer.profile_begin(); m.predict(x); pr = m.scorer.profile_end()
    ms = sum(pr[k][0] for k in ("branch21", "branch33", "branch51"))
    print("   branch kernels %.3f ms -> average tile period %.2f us" % (ms, ms * 1e3 / (tiles / 256)))

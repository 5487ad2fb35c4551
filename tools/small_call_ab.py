"""A/B of one small scoring call (default 53 sites) under the library named by DPTM_LIBRARY.
usage: DPTM_LIBRARY=$PWD/deepptmpred_amd/libdptm_new.so python tools/small_call_ab.py [sites ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from deepptmpred_amd import spec
from deepptmpred_amd.model import Model

dev = torch.device("cuda:0")
m = Model(spec.synthetic_weights(7))
for n in [int(a) for a in sys.argv[1:]] or [53]:
    r = bench.measure_small_call(m, m.scorer, dev, n)
    print(os.environ.get("DPTM_LIBRARY", "libdptm.so").split("/")[-1], n, "sites:",
          {k: round(v, 1) for k, v in r.items() if isinstance(v, float)})

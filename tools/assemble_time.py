"""Time of the window-assembly kernel alone over the S/T sites of 20,000 synthetic proteins in one launch (HIP events
behind the C ABI), for A/B builds selected with DPTM_LIBRARY.   usage: python tools/assemble_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from deepptmpred_amd import spec
from deepptmpred_amd.model import Model
dev = torch.device("cuda:0")
d = bench.synthetic_proteome(20000, 4321, dev)
n = d["n"]
sc = Model(spec.synthetic_weights(7), device=0).scorer
toks = [torch.empty((n, w), dtype=torch.uint8, device=dev) for w in (21, 33, 51)]
erow = torch.empty(n, dtype=torch.int32, device=dev)
stream = torch.cuda.current_stream().cuda_stream
def assemble():
    sc.assemble_proteins_device(d["codes"].data_ptr(), d["seq_off"].data_ptr(), d["esm_off"].data_ptr(), d["positions"].data_ptr(),
                                d["site_protein"].data_ptr(), n, toks[0].data_ptr(), toks[1].data_ptr(), toks[2].data_ptr(),
                                erow.data_ptr(), stream)
assemble()
torch.cuda.synchronize()
sc.profile_begin()
for _ in range(20):
    assemble()
ms, launches = sc.profile_end()["assemble"]
print("%s: %d sites, %.1f us per launch, %.2f TB/s" % (os.environ.get("DPTM_LIBRARY", "libdptm.so").split("/")[-1], n, ms / launches * 1e3,
                                                    n * bench.ASSEMBLE_BYTES_PER_SITE * launches / (ms * 1e-3) / 1e12))

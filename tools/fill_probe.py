"""Store floor for the window-assembly kernel: time of a plain fill / copy of its 110 MB of output on the GPU.
usage: python tools/fill_probe.py"""
import torch
dev=torch.device("cuda:0")
n=110_100_000
x=torch.empty(n,dtype=torch.uint8,device=dev)
y=torch.empty(n,dtype=torch.uint8,device=dev)
def t(f,k=20):
    f(); torch.cuda.synchronize()
    a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(k): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/k*1e3
for name,f,byt in (("fill 110MB", lambda: x.fill_(7), n), ("copy 110MB", lambda: y.copy_(x), 2*n), ("fill as int32", lambda: x.view(torch.int32).fill_(7), n)):
    us=t(f); print(name, "%.1f us  %.2f TB/s"%(us, byt/us/1e6))
big=torch.empty(2_000_000_000,dtype=torch.uint8,device=dev)
us=t(lambda: big.fill_(3),5); print("fill 2GB %.1f us %.2f TB/s"%(us,2e9/us/1e6))

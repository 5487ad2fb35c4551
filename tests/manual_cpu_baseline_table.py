"""CPU-restatement timings: batch 32 (the Keras predict default) at 1 thread and at all threads, batch 4096
at all threads; median of 3 repeats, one warm-up pass excluded."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepptmpred_amd import spec
from oracle.torch_ref import TorchRef

threads_all = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
w = spec.synthetic_weights(7)
ref = TorchRef(w, torch.float32)
for batch, threads, n in ((32, 1, 2048), (32, threads_all, 4096), (4096, threads_all, 32768)):
    torch.set_num_threads(threads)
    x = spec.synthetic_inputs(n, 6)
    ref.predict([a[:batch * 2] for a in x], batch_size=batch)
    rates = []
    for _ in range(3):
        t0 = time.perf_counter()
        ref.predict(x, batch_size=batch)
        rates.append(n / (time.perf_counter() - t0))
    print("batch %5d  threads %2d  %8.0f sites/s (median of 3 over %d sites)" % (batch, threads, statistics.median(rates), n), flush=True)

#!/usr/bin/env python
"""Throughput of the scoring path on MI355X: sites ("windows") scored per second.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--sites S]

Workload (BASELINE.json config 3): S = 1,000,000 synthetic sites per GPU -- three token
windows (21/33/51 residues), one 1280-float ESM-2 row and 8 structure features each --
resident in HBM before the timed region; one "step" scores all of them once in fp32
through the C ABI (dptm_score_device).  For N > 1 every rank scores its own S sites
(sites are independent: no data-path collective), so scaling is weak and `value` is the
whole-job rate.  Rank 0 prints ONE JSON line.

`--gpus N` with N > 1 outside a launcher starts the N ranks itself (fresh child processes
under torch.distributed.run, before this process touches a GPU) and relays rank 0's line.

Also reported on the same line (N = 1):
  roofline     the dominant kernel (fused W=51 sequence branch): algorithmic FLOPs per
               launch / its average launch time, measured with HIP events on the launch
               stream in an extra profiled step, against the dense fp32 MFMA peak.
  cpu_baseline the torch-CPU restatement of the reference graph (oracle/torch_ref.py)
               timed on a bounded sample of the same workload on the host cores.
  variants     the other arithmetic modes on the same resident inputs.
  streamed_from_host  the same model with NumPy arrays in and out (PCIe included).
  small_call   latency of one 53-site call (the reference's own use), host arrays and device-resident.
  proteome     many small proteins per call through dptm_score_proteins_device: window
               assembly + ESM row gather on the device (BASELINE north_star 1 and 4).
  sweep        BASELINE config 4 on one GPU: 17 weight sets over 262,144 resident sites (every site of every type
               checked: finite, rows sum to 1, types pairwise distinct); sweep_1m: the same over 1,000,000 sites.
  train        BASELINE config 5 on one GPU: training step at batch 64, 8,192 and 32,768, the Model.fit loop around
               it, and the torch-CPU restatement of the step on the host cores (train.cpu_baseline).
Other modes (`--types T`, `--train`, `--proteome`) print their own single line and shard over
ranks the same way.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

FLOPS_PER_SITE = 31_234_816          # 2*MAC of every contraction of the forward graph
FLOPS_BRANCH = {21: 5_564_160, 33: 8_946_432, 51: 14_296_320}
BYTES_PER_SITE = 5_261               # 1280*4 + 8*4 + 105 + 4
ASSEMBLE_BYTES_PER_SITE = 105 + 4 + 4 + 4   # three windows + esm row index written, position + protein id read
PEAK_F32_MFMA_TFLOPS = 157.3         # dense fp32 matrix peak of MI355X
PEAK_F16_MFMA_TFLOPS = 2500.0        # dense f16 / bf16 matrix peak (no sparsity)
PEAK_MFMA = {"f32": PEAK_F32_MFMA_TFLOPS, "f16x3": PEAK_F16_MFMA_TFLOPS, "f16": PEAK_F16_MFMA_TFLOPS,
             "bf16": PEAK_F16_MFMA_TFLOPS}
KERNEL_NAME = {"f32": "branch_kernel<51>", "f16x3": "branch_kernel_h<51,f16x3>", "f16": "branch_kernel_h<51,f16>",
               "bf16": "branch_kernel_h<51,bf16>"}
TRAFFIC_FILES = [os.path.join(ROOT, "profiles", f) for f in ("r3_pmc_traffic.json", "r2_pmc_traffic.json", "r1_pmc_traffic.json")]
ISSUE_FILES = [os.path.join(ROOT, "profiles", f) for f in ("r3_pmc_summary.json", "r2_pmc_summary.json")]
TRAIN_TRAFFIC_FILES = [os.path.join(ROOT, "profiles", "r3_train_pmc_traffic.json")]
ASSEMBLE_TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r2_pmc_assemble.json")
PEAK_HBM_GBS = 8000.0
N_SIMD = 1024                        # 256 CUs x 4 SIMDs
PEAK_CLOCK_GHZ = 2.4
# contractions of both TransformerBlocks and the attention pooling per site of a W-row window (forward):
# per row and block 3 x 64x64 (q, k, v) + 64x64 (out) + 2 x 64x256 (FFN) + 2 x W x 64 (scores, mix); the backward
# pass has two products per forward product (input and weight gradient; dP, dV, dQ, dK for the attention)
TB_FWD_FLOPS = {w: w * (2 * (98304 + 256 * w) + 256) for w in (21, 33, 51)}


def measured_traffic(precision: str, sites_per_launch: float):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes
    (profiles/collect_pmc.sh: 2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction for 16 B/lane
    reads).  PMC counters cannot be read from inside this process; null when no pass was recorded
    for this precision and launch size."""
    for path in TRAFFIC_FILES:
        try:
            with open(path) as f:
                rec = json.load(f).get(precision)
        except (OSError, ValueError):
            continue
        if rec and abs(rec.get("sites_per_launch", 0) - sites_per_launch) <= 0.5:
            return rec.get("traffic_bytes_per_launch"), os.path.relpath(path, ROOT)
    return None, None


def assemble_traffic(sites_per_launch: int):
    """HBM-side bytes per all-sites launch of the assembly kernel from the committed --pmc passes of
    `bench.py --proteome` (profiles/collect_pmc.sh assemble); null for another launch size."""
    try:
        with open(ASSEMBLE_TRAFFIC_FILE) as f:
            rec = json.load(f)
    except (OSError, ValueError):
        return None, None
    if rec.get("sites_per_launch") != sites_per_launch:
        return None, None
    return rec.get("traffic_bytes_per_launch"), os.path.relpath(ASSEMBLE_TRAFFIC_FILE, ROOT)


def issue_bound(precision: str, sites_per_launch: float, avg_launch_s: float):
    """The half-precision branch kernels are bound by the SIMDs' issue ports, not by the matrix cores: a matrix
    instruction holds the port for its passes and a vector instruction for 4 clocks (tools/valu_rate_probe.hip;
    tools/coexec_probe.hip shows the two do not overlap).  `peak` is the site rate of the W=51 kernel if no SIMD
    ever waited: the instruction counts per launch come from the committed rocprofv3 --pmc pass
    (SQ_INSTS_VALU, SQ_INSTS_MFMA, SQ_VALU_MFMA_BUSY_CYCLES), the launch time is the live HIP-event one."""
    for path in ISSUE_FILES:
        try:
            with open(path) as f:
                rec = json.load(f).get(precision)
        except (OSError, ValueError):
            continue
        if not rec or "issue_cycles_per_site" not in rec or avg_launch_s <= 0:
            continue
        peak = N_SIMD * PEAK_CLOCK_GHZ * 1e9 / rec["issue_cycles_per_site"]
        achieved = sites_per_launch / avg_launch_s
        return {"kind": "issue-slot budget of the kernel's own instruction mix (a utilisation diagnostic, not a roofline)",
                "kernel": KERNEL_NAME[precision], "achieved": achieved, "peak": peak,
                "unit": "sites/s", "frac": achieved / peak,
                "issue_cycles_per_site": rec["issue_cycles_per_site"],
                "valu_insts_per_site": rec["valu_insts_per_site"], "mfma_insts_per_site": rec["mfma_insts_per_site"],
                "mfma_cycles_per_inst": rec["mfma_cycles_per_inst"], "valu_cycles_per_inst": 4,
                "simds": N_SIMD, "clock_ghz": PEAK_CLOCK_GHZ, "source": os.path.relpath(path, ROOT)}
    return None


# ------------------------------------------------------------------------------------ ranks

def self_launch(args) -> None:
    """`--gpus N` (N > 1) outside a launcher: this process has imported torch but touched no device.
    Start N fresh ranks as children, relay rank 0's JSON line and the launcher's return code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env, cwd=ROOT)
    lines = [l for l in proc.stdout.splitlines() if l.strip()]
    found = [l for l in lines if l.lstrip().startswith("{")]
    for l in lines:
        if l not in found:
            print(l, file=sys.stderr)
    if found:
        print(found[-1])
    sys.stdout.flush()
    sys.exit(proc.returncode if proc.returncode else (0 if found else 1))


class Ranks:
    """One process per GPU.  RCCL carries the barrier and the max-over-ranks reduction; under
    DPTM_BENCH_REHEARSAL=1 (multi-rank rehearsal on a one-GPU box) every rank sits on GPU 0 and gloo
    does (RCCL refuses two ranks on one device)."""

    def __init__(self, args):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={self.world} ranks")
        self.rehearsal = os.environ.get("DPTM_BENCH_REHEARSAL") == "1"
        if self.rehearsal:
            self.local = 0
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.rehearsal:
                dist.init_process_group("gloo", rank=self.rank, world_size=self.world)
            else:
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world,
                                        device_id=torch.device(f"cuda:{self.local}"))
            self.dist = dist
        torch.cuda.set_device(self.local)
        self.dev = torch.device(f"cuda:{self.local}")
        self.rdev = torch.device("cpu") if self.rehearsal else self.dev      # where reduced scalars live

    def barrier(self) -> None:
        torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(self, v: float) -> float:
        if self.dist is None:
            return v
        t = torch.tensor([v], device=self.rdev, dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, v: float) -> float:
        if self.dist is None:
            return v
        t = torch.tensor([v], device=self.rdev, dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def timed(self, step, steps: int, warmup: int) -> float:
        """W untimed steps, then exactly K steps between barrier + synchronize; max over ranks (seconds)."""
        for i in range(warmup):
            step(i)
        self.barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            step(warmup + i)
        self.barrier()
        return self.max_over_ranks(time.perf_counter() - t0)

    def finish(self) -> None:
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()


# ------------------------------------------------------------------------------------ inputs

def device_inputs(n: int, seed: int, dev: torch.device):
    """Same distributions as deepptmpred_amd.spec.synthetic_inputs, generated on the GPU."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    wide = torch.randint(0, 20, (n, 51), generator=g, device=dev, dtype=torch.uint8)
    padded = torch.rand(n, generator=g, device=dev) < 0.05
    side = torch.randint(0, 2, (n,), generator=g, device=dev)
    plen = torch.randint(1, 26, (n,), generator=g, device=dev)
    col = torch.arange(51, device=dev)[None, :]
    left = padded[:, None] & (side[:, None] == 0) & (col < plen[:, None])
    right = padded[:, None] & (side[:, None] == 1) & (col >= 51 - plen[:, None])
    wide[left | right] = 20
    toks = [wide[:, 25 - h:25 + h + 1].contiguous() for h in (10, 16, 25)]
    del left, right
    return toks + [device_struct(n, g, dev), device_esm(n, g, dev)]


def device_esm(rows: int, g, dev):
    esm = torch.empty((rows, 1280), device=dev, dtype=torch.float32)
    step = 1 << 17
    for s in range(0, rows, step):   # chunked: keeps the temporaries small
        e = min(rows, s + step)
        blk = torch.randn((e - s, 1280), generator=g, device=dev) * 0.27
        heavy = torch.rand((e - s, 1280), generator=g, device=dev) < 0.01
        esm[s:e] = torch.where(heavy, blk * 10.0, blk)
    return esm


def device_struct(n: int, g, dev):
    st = torch.zeros((n, 8), device=dev, dtype=torch.float32)
    st[:, 0] = torch.rand(n, generator=g, device=dev) * 250
    st[:, 1] = torch.rand(n, generator=g, device=dev) * 360 - 180
    st[:, 2] = torch.rand(n, generator=g, device=dev) * 360 - 180
    ss = torch.randint(0, 3, (n,), generator=g, device=dev)
    st[torch.arange(n, device=dev), 3 + ss] = 1.0
    st[:, 6] = torch.rand(n, generator=g, device=dev) * 80 + 20
    st[:, 7] = torch.rand(n, generator=g, device=dev) * 80 + 20
    return st


# ------------------------------------------------------------------------------------ CPU leg

def cpu_baseline(weights, budget_s: float = 12.0):
    """Times oracle/torch_ref.py (fp32) on the host cores: `value` = batch 4096 on all threads for
    ~budget_s; `grid` = medians of 3 repeats for batch {32, 4096} x threads {1, all} on small samples."""
    from deepptmpred_amd import spec
    from oracle.torch_ref import TorchRef
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    torch.set_num_threads(threads)
    ref = TorchRef(weights, torch.float32)
    probe = spec.synthetic_inputs(1024, 5)
    ref.predict(probe, batch_size=1024)          # warm-up
    t0 = time.perf_counter()
    ref.predict(probe, batch_size=1024)
    per_site = (time.perf_counter() - t0) / 1024
    n = int(min(max(budget_s / max(per_site, 1e-9), 2048), 65536))
    n = (n // 1024) * 1024
    x = spec.synthetic_inputs(n, 6)
    t0 = time.perf_counter()
    ref.predict(x, batch_size=4096)
    dt = time.perf_counter() - t0
    grid = {}
    for th in sorted({1, threads}):
        torch.set_num_threads(th)
        for batch, m in ((32, 512), (4096, 4096)):
            xs = [a[:m] for a in x]
            ref.predict(xs, batch_size=batch)
            reps = []
            for _ in range(3):
                t1 = time.perf_counter()
                ref.predict(xs, batch_size=batch)
                reps.append(m / (time.perf_counter() - t1))
            grid[f"batch{batch}_threads{th}"] = {"sites_per_s": float(np.median(reps)), "sites": m}
    torch.set_num_threads(threads)
    return {"value": n / dt, "unit": "sites/s", "cores": threads, "kind": "port",
            "sample": f"{n} synthetic sites, torch-CPU fp32 restatement, batch 4096, {dt:.1f} s",
            "grid": grid, "host_cpu_count": os.cpu_count()}


def cpu_train_baseline(batch: int = 64, budget_s: float = 10.0):
    """The training step of oracle/train_ref.py (torch autograd + AdamW, fp32, dropout on) on the host cores:
    forward in training mode, loss, backward and the optimizer update per step, batch 64 (config.py:56)."""
    from deepptmpred_amd import spec
    from oracle.train_ref import TrainRef
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    torch.set_num_threads(threads)
    ref = TrainRef(spec.synthetic_weights(7), torch.float32)
    x = spec.synthetic_inputs(batch, 99)
    y = (np.random.default_rng(5).random(batch) < 0.042).astype(np.int64)
    y[0] = 1                                        # at least one positive, as the class weights assume
    ref.step(x, y, seed=0)                          # warm-up
    t0 = time.perf_counter()
    steps = 0
    while steps < 3 or (time.perf_counter() - t0 < budget_s and steps < 200):
        ref.step(x, y, seed=steps + 1)
        steps += 1
    dt = time.perf_counter() - t0
    return {"value": batch * steps / dt, "unit": "sites/s", "ms_per_step": dt / steps * 1e3, "cores": threads, "kind": "port",
            "sample": f"{steps} steps of batch {batch}, torch-CPU fp32 autograd restatement of the training step, {dt:.1f} s"}


# ------------------------------------------------------------------------------------ training step

def measure_train(R: Ranks, batch: int, steps: int, warmup: int) -> dict:
    """fwd + bwd + gradient all-reduce + AdamW per step on synthetic imbalanced sites (labels ~
    Bernoulli(0.042), the phosphorylation ratio), one replica per GPU, dropout on."""
    from deepptmpred_amd import _native, spec
    from deepptmpred_amd.training import NativeBackend
    dev, rank, world = R.dev, R.rank, R.world
    backend = NativeBackend(spec.synthetic_weights(7), device=R.local, max_batch=batch)
    x = device_inputs(batch, 99 + rank, dev)
    g = torch.Generator(device=dev)
    g.manual_seed(5 + rank)
    labels = (torch.rand(batch, generator=g, device=dev) < 0.042).to(torch.int32)
    n_pos = int(R.sum_over_ranks(float(labels.sum().item())))
    o = _native.TrainOpts()
    o.rate_embed, o.rate_conv, o.rate_tb, o.rate_dense = 1e-4, 0.3, 0.1, 0.3
    o.n_global, o.n_pos_global, o.l2, o.l2_scale = batch * world, n_pos, 1e-4, 1.0 / world
    # one explicit stream for the whole step: the library's kernels, RCCL's all-reduce (which orders itself
    # after torch's current stream) and the optimizer update are then stream-ordered without host syncs
    # -- the trainer's own main stream, so that the step starts and ends without a hand-over between streams
    tstream = backend.stream
    st = tstream.cuda_stream
    torch.cuda.synchronize()

    def step(i):
        o.seed = i * 1000003 + rank
        with torch.cuda.stream(tstream):
            backend.trainer.forward_backward(x[0].data_ptr(), x[1].data_ptr(), x[2].data_ptr(), x[3].data_ptr(),
                                             x[4].data_ptr(), labels.data_ptr(), batch, o, False, st)
            if R.dist is not None:
                backend.allreduce_grads()
            backend.trainer.apply(5e-4, 1e-4, 1.0, 1.0 / world, st)

    elapsed = R.timed(step, steps, warmup)
    # extra steps with the per-kernel clocks armed (branches one after another on one stream, one launch per branch: at
    # small batches the timed steps above start the three branches' launches as one, so the first armed step is the
    # first ever launch of the per-branch kernels and pays their set-up; the second is the one reported)
    for k in range(2):
        backend.trainer.profile_begin()
        step(steps + warmup + k)
        kernel_ms = backend.trainer.profile_end()
    R.barrier()
    w = backend.trainer.get_weights()
    assert np.isfinite(w).all()
    backend.close()
    value = world * batch * steps / elapsed
    tf = value / world * 3 * FLOPS_PER_SITE / 1e12
    # dominant piece: both TransformerBlocks + pooling backward of the W=51 branch -- one launch per step at small
    # batches, five at large ones (pooling, then the row-wise and the attention half of each block); the library's
    # own clock around them ("tb_bwd51") is the duration, the committed --pmc passes sum the same launches
    bwd_s = kernel_ms["tb_bwd51"] * 1e-3
    bwd_tf = batch * 2 * TB_FWD_FLOPS[51] / bwd_s / 1e12 if bwd_s > 0 else 0.0
    traffic, traffic_file, launches = None, None, None
    for path in TRAIN_TRAFFIC_FILES:
        try:
            with open(path) as f:
                rec = json.load(f).get(f"batch{batch}")
        except (OSError, ValueError):
            continue
        if rec:
            traffic, traffic_file = rec.get("traffic_bytes_per_launch"), os.path.relpath(path, ROOT)
            launches = rec.get("launches_per_pass")
            break
    kernel_name = ("tb_bwd_pair_kernel<51>" if batch <= 112 else "tb_bwd_kernel<51,256,0>" if batch <= 231
                   else "tb_bwd_kernel<51,512,*> (+ tb_row_bwd_kernel where the backward pass is five launches)")
    roofline = {"bound": "mfma", "kernel": kernel_name, "achieved": bwd_tf,
                "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": bwd_tf / PEAK_F32_MFMA_TFLOPS, "traffic": traffic,
                "traffic_source": f"{traffic_file} (offline rocprofv3 --pmc passes of bench.py --train)" if traffic_file else None,
                "avg_launch_ms": kernel_ms["tb_bwd51"], "launches_per_pass": launches, "sites_per_launch": batch,
                "flops_per_site": 2 * TB_FWD_FLOPS[51]}
    return {"batch_per_gpu": batch, "steps": steps, "ms_per_step": elapsed / steps * 1e3, "sites_per_s": value,
            "tflops_per_gpu": tf, "frac_of_f32_mfma_peak": tf / PEAK_F32_MFMA_TFLOPS, "roofline": roofline,
            "kernel_ms_serial_step": {k: round(v, 4) for k, v in kernel_ms.items()},
            "kernel_ms_serial_total": round(sum(kernel_ms.values()), 4)}


def measure_fit(R: Ranks, sites: int = 6400, batch: int = 64) -> dict:
    """Wall time per step of the Keras-style loop (Model.fit on host arrays, trainer.py:332-339): the training set is
    uploaded once, batches are gathered on the device, losses are read once per epoch."""
    from deepptmpred_amd import create_ptm_model, spec
    from deepptmpred_amd.training import AdamW
    x = spec.synthetic_inputs(sites, 3)
    y = (np.random.default_rng(1).random(sites) < 0.042).astype(np.int64)
    m = create_ptm_model(None, device=R.local)
    m.compile(optimizer=AdamW(learning_rate=5e-4, weight_decay=1e-4))
    m.fit(x, y, batch_size=batch, epochs=1, verbose=0)          # warm-up epoch
    t0 = time.perf_counter()
    h = m.fit(x, y, batch_size=batch, epochs=2, verbose=0)
    dt = time.perf_counter() - t0
    steps = 2 * ((sites + batch - 1) // batch)
    m.close()
    assert np.isfinite(h.history["loss"]).all()
    return {"batch": batch, "steps": steps, "ms_per_step_wall": dt / steps * 1e3, "sites_per_s": 2 * sites / dt,
            "note": "Model.fit, host arrays in: data set resident on the device, losses read once per epoch"}


def train_main(args, R: Ranks) -> None:
    m = measure_train(R, args.batch, args.steps, args.warmup)
    cpu = cpu_train_baseline(min(args.batch, 4096)) if R.rank == 0 and R.world == 1 and not args.no_cpu_baseline else None
    if R.rank == 0:
        print(json.dumps({
            **({"cpu_baseline": cpu} if cpu else {}),
            "metric": "training sites/sec (fwd+bwd+all-reduce+AdamW)", "value": m["sites_per_s"], "unit": "sites/s",
            "n_gpus": R.world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": m["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic imbalanced sites (4.2% positive), seeded synthetic weights",
            "config": {"workload": "config 5: main.py-style training step", "batch_per_gpu": args.batch,
                       "parallelism": f"dp{R.world} (one flat 6.6 MB fp32 gradient all-reduce per step)"},
            "roofline": m["roofline"],
            "whole_path": {"tflops": m["tflops_per_gpu"], "frac_of_f32_mfma_peak": m["frac_of_f32_mfma_peak"],
                           "flops_per_site": 3 * FLOPS_PER_SITE},
            "kernel_ms_serial_step": m["kernel_ms_serial_step"],
            "kernel_ms_serial_total": m["kernel_ms_serial_total"]}))
    R.finish()


# ------------------------------------------------------------------------------------ 17-type sweep

def measure_sweep(R: Ranks, T: int, n: int, steps: int, warmup: int, precision: str, x=None, streamed_sites: int = 262144,
                  warm_sites: int = 0) -> dict:
    """The same resident sites scored by T weight sets (every PTM type has its own model,
    predict.py:119-137).  Sites are sharded over ranks, every rank holds all T models (6.5 MB each),
    inputs are read T times from HBM and never re-uploaded.  Counts (site, type) scorings per second;
    rank 0 also times the host-array form (one upload per pass shared by all types)."""
    from deepptmpred_amd import spec
    from deepptmpred_amd.model import Model
    from deepptmpred_amd.sharding import predict_all_types
    dev = R.dev
    models = {f"type{k:02d}": Model(spec.synthetic_weights(100 + k), device=R.local) for k in range(T)}
    for m in models.values():
        m.scorer.set_precision(precision)
    if x is None:
        x = device_inputs(n, 1234 + R.rank, dev)
    outs = [torch.empty((n, 2), device=dev, dtype=torch.float32) for _ in range(T)]
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream().cuda_stream
    ptrs = [t.data_ptr() for t in x]

    def step(_i):
        for m, o in zip(models.values(), outs):
            m.scorer.score_device(ptrs[0], ptrs[1], ptrs[2], ptrs[3], ptrs[4], n, o.data_ptr(), stream)

    if warm_sites:      # workspaces and first launches of every model on a few sites instead of a whole untimed step
        for m, o in zip(models.values(), outs):
            m.scorer.score_device(ptrs[0], ptrs[1], ptrs[2], ptrs[3], ptrs[4], min(n, warm_sites), o.data_ptr(), stream)
    elapsed = R.timed(step, steps, warmup)
    # every (site, type) of the run: finite probabilities that sum to one
    worst_row = 0.0
    for o in outs:
        assert bool(torch.isfinite(o).all().item())
        worst_row = max(worst_row, float((o.sum(dim=1) - 1).abs().max().item()))
    assert worst_row < 1e-5, worst_row
    # different weight sets must give different answers (the models are really distinct): no two types agree
    distinct = True
    for a in range(T):
        for b in range(a + 1, T):
            distinct &= float((outs[a][:4096] - outs[b][:4096]).abs().max().item()) > 1e-4
    assert distinct

    streamed = None
    if R.rank == 0 and streamed_sites:
        m = min(n, streamed_sites)
        block = spec.synthetic_inputs(min(m, 262144), 77)
        # larger host sets are the 262,144-site block cycled (generating 66 GB of random rows on the host would take
        # longer than scoring them)
        hx = block if m <= 262144 else [np.tile(a, (-(-m // 262144), 1))[:m] for a in block]
        predict_all_types(models, [a[:4096] for a in hx])            # warm-up: pinned buffers, first launches
        t1 = time.perf_counter()
        res = predict_all_types(models, hx)
        dt = time.perf_counter() - t1
        assert len(res) == T
        streamed = {"site_types_per_s_one_gpu": m * T / dt, "sites": m,
                    "note": "host arrays in, host arrays out; one upload per pass shared by all types"}
    R.barrier()
    for m in models.values():
        m.close()
    value = R.world * n * T * steps / elapsed
    return {"types": T, "sites_per_gpu": n, "steps": steps, "ms_per_step": elapsed / steps * 1e3,
            "site_types_per_s": value, "sites_per_s_all_types": value / T,
            "tflops_per_gpu": value / R.world * FLOPS_PER_SITE / 1e12,
            "frac_of_mfma_peak": value / R.world * FLOPS_PER_SITE / 1e12 / PEAK_MFMA[precision],
            "checked": {"site_types": n * T, "finite": True, "max_row_sum_err": worst_row, "types_pairwise_distinct": distinct},
            "streamed_from_host": streamed}


def sweep_main(args, R: Ranks) -> None:
    T, n = args.types, args.sites
    m = measure_sweep(R, T, n, args.steps, args.warmup, args.precision, None, args.streamed_sites, 4096)
    if R.rank == 0:
        print(json.dumps({
            "metric": "windows x PTM types scored/sec", "value": m["site_types_per_s"], "unit": "site-types/s",
            "n_gpus": R.world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": m["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision,
            "data": "synthetic (device-generated sites, seeded synthetic weight sets 100..)",
            "config": {"workload": f"config 4: {n:,} resident sites per GPU x {T} weight sets, {args.precision}, "
                                   f"{R.world} x MI355X", "sites_per_gpu": n, "types": T,
                       "parallelism": f"dp{R.world} (sites sharded, every rank holds all {T} models, no collective)"},
            "whole_path": {"tflops": m["tflops_per_gpu"], "sites_per_s_all_types": m["sites_per_s_all_types"]},
            "streamed_from_host": m["streamed_from_host"]}))
    R.finish()


# ------------------------------------------------------------------------------------ proteome

def synthetic_proteome(n_proteins: int, seed: int, dev: torch.device, max_sites: int = 0):
    """Many small proteins: lengths ~ N(500, 300) clipped to 50..3000, residues uniform over the 20
    letters, every S / T residue a site (predict.py:631-636 picks sites by target residue), one ESM-2
    table row per residue.  Returns device tensors in the layout dptm_score_proteins_device takes."""
    rng = np.random.default_rng(seed)
    lens = np.clip(np.rint(rng.normal(500, 300, n_proteins)), 50, 3000).astype(np.int64)
    offs = np.zeros(n_proteins + 1, dtype=np.int64)
    np.cumsum(lens, out=offs[1:])
    total = int(offs[-1])
    codes = rng.integers(0, 20, total, dtype=np.uint8)
    is_site = (codes == 15) | (codes == 16)          # S, T
    gidx = np.nonzero(is_site)[0]
    if max_sites and len(gidx) > max_sites:
        gidx = gidx[:max_sites]
    prot = (np.searchsorted(offs, gidx, side="right") - 1).astype(np.int32)
    pos = (gidx - offs[prot] + 1).astype(np.int32)   # 1-based, protein-local
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    n = len(gidx)
    d = {"codes": torch.from_numpy(codes).to(dev), "seq_off": torch.from_numpy(offs).to(dev),
         "esm_off": torch.from_numpy(offs.copy()).to(dev), "positions": torch.from_numpy(pos).to(dev),
         "site_protein": torch.from_numpy(prot).to(dev), "esm": device_esm(total, g, dev),
         "strct": device_struct(n, g, dev), "n": n, "proteins": n_proteins, "residues": total}
    return d


def measure_proteome(R: Ranks, sc, n_proteins: int, steps: int, warmup: int, seed: int = 4321) -> dict:
    """All S/T sites of `n_proteins` synthetic proteins per step through ONE call of
    dptm_score_proteins_device (windows assembled and ESM rows gathered on the device)."""
    d = synthetic_proteome(n_proteins, seed + R.rank, R.dev)
    n = d["n"]
    out = torch.empty((n, 2), device=R.dev, dtype=torch.float32)
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream().cuda_stream

    def step(_i):
        sc.score_proteins_device(d["codes"].data_ptr(), d["seq_off"].data_ptr(), d["esm"].data_ptr(),
                                 d["esm_off"].data_ptr(), d["proteins"], d["positions"].data_ptr(),
                                 d["site_protein"].data_ptr(), d["strct"].data_ptr(), n, out.data_ptr(), stream)

    elapsed = R.timed(step, steps, warmup)
    total_sites = R.sum_over_ranks(float(n))
    chk = out[:: max(1, n // 4096)].cpu().numpy()
    assert np.isfinite(chk).all() and np.abs(chk.sum(axis=1) - 1).max() < 1e-5
    sc.profile_begin()
    step(0)
    prof = sc.profile_end()
    R.barrier()
    a_ms, a_launches = prof["assemble"]
    a_gbs = n * ASSEMBLE_BYTES_PER_SITE / (a_ms * 1e-3) / 1e9 if a_ms > 0 else 0.0
    # the assembly kernel alone over all sites in ONE launch (inside the scoring call it runs once per 32,768-site
    # pass, a few microseconds each): HIP events around 5 launches
    toks = [torch.empty((n, w), dtype=torch.uint8, device=R.dev) for w in (21, 33, 51)]
    erow = torch.empty(n, dtype=torch.int32, device=R.dev)

    def assemble():
        sc.assemble_proteins_device(d["codes"].data_ptr(), d["seq_off"].data_ptr(), d["esm_off"].data_ptr(),
                                    d["positions"].data_ptr(), d["site_protein"].data_ptr(), n, toks[0].data_ptr(),
                                    toks[1].data_ptr(), toks[2].data_ptr(), erow.data_ptr(), stream)

    assemble()
    torch.cuda.synchronize()
    sc.profile_begin()
    for _ in range(5):
        assemble()
    s_ms, s_launches = sc.profile_end()["assemble"]
    s_gbs = n * ASSEMBLE_BYTES_PER_SITE * s_launches / (s_ms * 1e-3) / 1e9 if s_ms > 0 else 0.0
    del toks, erow
    return {"proteins_per_gpu": n_proteins, "residues_per_gpu": d["residues"], "sites_per_gpu": n, "steps": steps,
            "ms_per_step": elapsed / steps * 1e3, "sites_per_s": total_sites * steps / elapsed,
            "proteins_per_s": R.world * n_proteins * steps / elapsed,
            "assemble": {"ms_per_step": a_ms, "launches": a_launches, "bytes_per_site": ASSEMBLE_BYTES_PER_SITE,
                         "gbs_algorithmic": a_gbs, "frac_of_hbm_peak": a_gbs / PEAK_HBM_GBS},
            "assemble_one_launch": {"sites": n, "avg_launch_ms": s_ms / max(s_launches, 1), "gbs_algorithmic": s_gbs,
                                    "frac_of_hbm_peak": s_gbs / PEAK_HBM_GBS},
            "kernel_ms_per_step": {k: round(v[0], 3) for k, v in prof.items()}}


def proteome_main(args, R: Ranks) -> None:
    from deepptmpred_amd import spec
    from deepptmpred_amd.model import Model
    model = Model(spec.synthetic_weights(7), device=R.local)
    model.scorer.set_precision(args.precision)
    m = measure_proteome(R, model.scorer, args.proteins, args.steps, args.warmup)
    if R.rank == 0:
        a_traffic, a_file = assemble_traffic(m["assemble_one_launch"]["sites"])
        print(json.dumps({
            "metric": "windows scored/sec (proteome: many proteins per call, device-side assembly)",
            "value": m["sites_per_s"], "unit": "sites/s", "n_gpus": R.world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": m["ms_per_step"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision,
            "data": "synthetic proteins (lengths ~N(500,300), all S/T sites), seeded synthetic weights",
            "config": {"workload": f"proteome: {args.proteins:,} synthetic proteins per GPU, every S/T site, "
                                   f"{args.precision}, {R.world} x MI355X",
                       "parallelism": f"dp{R.world} (proteins sharded, no collective)"},
            "roofline": {"bound": "hbm", "kernel": "assemble_kernel (all sites in one launch)",
                         "achieved": m["assemble_one_launch"]["gbs_algorithmic"], "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": m["assemble_one_launch"]["frac_of_hbm_peak"], "traffic": a_traffic,
                         "traffic_source": f"{a_file} (offline rocprofv3 --pmc passes of this command)" if a_file else None,
                         "avg_launch_ms": m["assemble_one_launch"]["avg_launch_ms"],
                         "bytes_per_site": ASSEMBLE_BYTES_PER_SITE},
            "proteome": m}))
    R.finish()


# ------------------------------------------------------------------------------------ headline

def measure_streamed(model, n: int) -> dict:
    """NumPy arrays in, NumPy array out through Model.predict (dptm_score): PCIe included."""
    from deepptmpred_amd import spec
    hx = spec.synthetic_inputs(n, 78)
    model.predict([a[:4096] for a in hx])       # staging buffers, first launches
    model.predict(hx)                            # pinned pipeline allocated
    t1 = time.perf_counter()
    p = model.predict(hx)
    dt = time.perf_counter() - t1
    assert p.shape == (n, 2) and np.isfinite(p).all()
    return {"sites_per_s_one_gpu": n / dt, "sites": n, "input_gbs": n * BYTES_PER_SITE / dt / 1e9,
            "note": "host arrays in, host arrays out (Model.predict): pinned double-buffered pipeline"}


def measure_small_call(model, sc, dev, n: int = 53) -> dict:
    """Latency of ONE small call -- the reference's own use: the 53 S/T sites of a 480-residue protein per
    model.predict (predict.py:598-599) -- with host arrays in and out, and device-resident (enqueue + wait, and
    back to back without waiting in between = device time per call)."""
    from deepptmpred_amd import spec
    hx = spec.synthetic_inputs(n, 79)
    for _ in range(5):
        p = model.predict(hx)
    assert p.shape == (n, 2) and np.isfinite(p).all()
    t = []
    for _ in range(100):
        t0 = time.perf_counter()
        model.predict(hx)
        t.append(time.perf_counter() - t0)
    host_us = sorted(t)[50] * 1e6
    d = [torch.from_numpy(a).to(dev) for a in hx]
    out = torch.empty((n, 2), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    call = lambda: sc.score_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(), n,
                                   out.data_ptr(), 0)
    for _ in range(5):
        call()
    sc.sync()
    t = []
    for _ in range(100):
        t0 = time.perf_counter()
        call()
        sc.sync()
        t.append(time.perf_counter() - t0)
    dev_us = sorted(t)[50] * 1e6
    t0 = time.perf_counter()
    for _ in range(300):
        call()
    sc.sync()
    b2b_us = (time.perf_counter() - t0) / 300 * 1e6
    assert np.abs(out.cpu().numpy() - p).max() < 1e-6
    return {"sites": n, "host_arrays_us_per_call": host_us, "device_resident_us_per_call": dev_us,
            "device_resident_back_to_back_us_per_call": b2b_us,
            "note": "median of 100 calls; back to back = 300 calls enqueued without waiting in between"}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--types", type=int, default=0,
                    help="score the resident sites with this many weight sets (BASELINE config 4) instead")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--sites", type=int, default=1_000_000, help="sites per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true",
                    help="skip the short extra measurement of the other precision modes")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the streamed / proteome / sweep / train sub-blocks of the default line")
    ap.add_argument("--precision", default="f32", choices=["f32", "f16x3", "f16", "bf16"],
                    help="arithmetic of the sequence-branch matrix products for the headline number")
    ap.add_argument("--train", action="store_true",
                    help="measure the training step (BASELINE config 5) instead of scoring")
    ap.add_argument("--batch", type=int, default=64, help="--train: sites per GPU per step")
    ap.add_argument("--streamed-sites", type=int, default=262144,
                    help="--types: sites of the host-array leg (NumPy in, NumPy out; 0 skips it)")
    ap.add_argument("--proteome", action="store_true",
                    help="score many synthetic proteins per call with device-side window assembly instead")
    ap.add_argument("--proteins", type=int, default=20000, help="--proteome: proteins per GPU per step")
    args = ap.parse_args()
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)          # before anything touches a GPU
    R = Ranks(args)
    if args.train:
        return train_main(args, R)
    if args.types > 0:
        return sweep_main(args, R)
    if args.proteome:
        return proteome_main(args, R)
    rank, world, dev = R.rank, R.world, R.dev

    from deepptmpred_amd import spec
    from deepptmpred_amd.model import Model

    weights = spec.synthetic_weights(7)
    model = Model(weights, device=R.local)
    sc = model.scorer
    sc.set_precision(args.precision)
    n = args.sites
    x = device_inputs(n, 1234 + rank, dev)
    out = torch.empty((n, 2), device=dev, dtype=torch.float32)
    torch.cuda.synchronize()     # inputs are complete before any library stream reads them
    stream = torch.cuda.current_stream().cuda_stream
    ptrs = [t.data_ptr() for t in x]

    def step(_i=0):
        sc.score_device(ptrs[0], ptrs[1], ptrs[2], ptrs[3], ptrs[4], n, out.data_ptr(), stream)

    elapsed = R.timed(step, args.steps, args.warmup)

    # sanity: finite probabilities that sum to one
    chk = out[:: max(1, n // 4096)].cpu().numpy()
    assert np.isfinite(chk).all() and np.abs(chk.sum(axis=1) - 1).max() < 1e-5

    # one extra, profiled step: per-kernel HIP-event timings on the launch stream
    sc.profile_begin()
    step()
    prof = sc.profile_end()
    R.barrier()

    # the other arithmetic modes on the same resident inputs (rank 0, two steps each): reported
    # beside the headline, never as `value`
    variants = {}
    if rank == 0 and world == 1 and not args.no_variants:
        ref_out = out.clone()                      # the headline mode's probabilities of the same sites
        bound_vs_f32 = {"f32": 1e-4, "f16x3": 1e-4, "f16": 1e-2, "bf16": 6e-2}   # the modes' test bounds
        for mode in ("f32", "f16x3", "f16", "bf16"):
            if mode == args.precision:
                continue
            sc.set_precision(mode)
            step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            step()
            step()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t1) / 2
            sc.profile_begin()
            step()
            vp = sc.profile_end()
            torch.cuda.synchronize()
            # every site of the run: finite, rows sum to one, within the mode's bound of the headline mode's answer
            finite = bool(torch.isfinite(out).all().item())
            row_err = float((out.sum(dim=1) - 1).abs().max().item())
            diff = float((out - ref_out).abs().max().item())
            bound = max(bound_vs_f32[mode], bound_vs_f32[args.precision])
            assert finite and row_err < 1e-5 and diff < bound, (mode, finite, row_err, diff)
            v_ms, v_launches = vp["branch51"]
            v_avg_s = v_ms * 1e-3 / max(v_launches, 1)
            v_tf = n * FLOPS_BRANCH[51] / (v_ms * 1e-3) / 1e12 if v_ms > 0 else 0.0
            variants[mode] = {"sites_per_s_one_gpu": n / dt, "branch51_avg_launch_ms": v_avg_s * 1e3,
                              "checked": {"sites": n, "finite": finite, "max_row_sum_err": row_err,
                                          "max_abs_diff_vs_" + args.precision: diff, "bound": bound},
                              "roofline": {"bound": "mfma", "kernel": KERNEL_NAME[mode], "achieved": v_tf,
                                           "peak": PEAK_MFMA[mode], "unit": "TFLOP/s", "frac": v_tf / PEAK_MFMA[mode],
                                           "traffic": measured_traffic(mode, n / max(v_launches, 1))[0],
                                           "avg_launch_ms": v_avg_s * 1e3, "sites_per_launch": n / max(v_launches, 1),
                                           "flops_per_site": FLOPS_BRANCH[51]}}
            ib = issue_bound(mode, n / max(v_launches, 1), v_avg_s)
            if ib:      # a utilisation diagnostic from the kernel's own instruction counts, not a roofline
                variants[mode]["issue_budget"] = ib
        sc.set_precision(args.precision)
        del ref_out
    R.barrier()

    # configs 4 and 5 and the host / proteome forms, on one GPU, beside the headline (never `value`)
    extras = {}
    if world == 1 and not args.no_extras:
        def guarded(fn, *a):
            """a side measurement that fails leaves its error in the line instead of taking the headline with it"""
            try:
                return fn(*a)
            except Exception as e:      # noqa: BLE001 -- reported, not hidden
                torch.cuda.synchronize()
                return {"error": f"{type(e).__name__}: {e}"}

        extras["streamed_from_host"] = guarded(measure_streamed, model, min(n, 262144))
        extras["small_call"] = guarded(measure_small_call, model, sc, R.dev)
        extras["proteome"] = guarded(measure_proteome, R, sc, 20000 if n >= 262144 else 200, 3, 1)
        ns = min(n, 262144)
        xs = [t[:ns] for t in x]
        extras["sweep"] = guarded(measure_sweep, R, 17 if n >= 262144 else 3, ns, 2, 1, args.precision, xs)
        # config 4 at one GPU's resident million: all 17 weight sets over every site of the headline run, one step
        extras["sweep_1m"] = guarded(measure_sweep, R, 17 if n >= 262144 else 3, n, 1, 0, args.precision, x, 0, 4096)
        del x, xs, out
        torch.cuda.empty_cache()
        extras["train"] = {"batch64": guarded(measure_train, R, 64, 50, 10),
                           "batch8192": guarded(measure_train, R, 8192, 5, 2) if n >= 262144 else None,
                           "batch32768": guarded(measure_train, R, 32768, 3, 1) if n >= 262144 else None,
                           "fit_batch64": guarded(measure_fit, R, 6400 if n >= 262144 else 640)}
        if not args.no_cpu_baseline:
            extras["train"]["cpu_baseline"] = guarded(cpu_train_baseline, 64, 8.0 if n >= 262144 else 2.0)

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        value = world * n * args.steps / elapsed
        b51_ms, b51_launches = prof["branch51"]
        sites_per_launch = n / max(b51_launches, 1)
        avg_s = b51_ms * 1e-3 / max(b51_launches, 1)
        achieved = sites_per_launch * FLOPS_BRANCH[51] / avg_s / 1e12 if avg_s > 0 else 0.0
        total_kernel_ms = sum(v[0] for v in prof.values())
        traffic, traffic_file = measured_traffic(args.precision, sites_per_launch)
        line = {
            "metric": "windows scored/sec", "value": value, "unit": "sites/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision,
            "data": "synthetic (device-generated sites, seeded synthetic weights)",
            "config": {"workload": f"config 3: {n:,} synthetic sites per GPU (windows 21/33/51 + ESM-2 1280 + 8 struct), "
                                   f"{args.precision}, {world} x MI355X",
                       "sites_per_gpu": n, "parallelism": f"dp{world} (sites sharded, no collective)"},
            "roofline": {"bound": "mfma", "kernel": KERNEL_NAME[args.precision], "achieved": achieved,
                         "peak": PEAK_MFMA[args.precision], "unit": "TFLOP/s",
                         "frac": achieved / PEAK_MFMA[args.precision],
                         "traffic": traffic,
                         "traffic_source": f"{traffic_file} (offline rocprofv3 --pmc passes of this command)" if traffic_file else None,
                         "avg_launch_ms": avg_s * 1e3, "sites_per_launch": sites_per_launch,
                         "flops_per_site": FLOPS_BRANCH[51]},
            "whole_path": {"tflops": value / world * FLOPS_PER_SITE / 1e12,
                           "frac_of_f32_mfma_peak": value / world * FLOPS_PER_SITE / 1e12 / PEAK_F32_MFMA_TFLOPS,
                           "hbm_gbs_algorithmic": value / world * BYTES_PER_SITE / 1e9,
                           "frac_of_hbm_peak": value / world * BYTES_PER_SITE / 1e9 / PEAK_HBM_GBS},
            "kernel_ms_per_step": {k: round(v[0], 3) for k, v in prof.items()},
            "kernel_ms_total": round(total_kernel_ms, 3),
        }
        ib = issue_bound(args.precision, sites_per_launch, avg_s)
        if ib:                                   # matrix + vector issue slots of the kernel's own instruction mix, beside the roofline
            line["issue_budget"] = ib
        if variants:
            line["variants"] = variants
        line.update(extras)
        if not args.no_cpu_baseline and world == 1:      # the CPU leg is timed at N = 1 only
            line["cpu_baseline"] = cpu_baseline(weights)
        print(json.dumps(line))
    R.finish()


if __name__ == "__main__":
    main()

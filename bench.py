#!/usr/bin/env python
"""Throughput of the scoring path on MI355X: sites ("windows") scored per second.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--sites S]

Workload (BASELINE.json config 3): S = 1,000,000 synthetic sites per GPU -- three token
windows (21/33/51 residues), one 1280-float ESM-2 row and 8 structure features each --
resident in HBM before the timed region; one "step" scores all of them once in fp32
through the C ABI (dptm_score_device).  For N > 1 every rank scores its own S sites
(sites are independent: no data-path collective), so scaling is weak and `value` is the
whole-job rate.  Rank 0 prints ONE JSON line.

Also reported on the same line:
  roofline     the dominant kernel (fused W=51 sequence branch): algorithmic FLOPs per
               launch / its average launch time, measured with HIP events on the launch
               stream in an extra profiled step, against the dense fp32 MFMA peak.
  cpu_baseline the torch-CPU restatement of the reference graph (oracle/torch_ref.py)
               timed on a bounded sample of the same workload on the host cores.
"""
from __future__ import annotations

import argparse
import json
import os
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
PEAK_F32_MFMA_TFLOPS = 157.3         # dense fp32 matrix peak of MI355X
PEAK_F16_MFMA_TFLOPS = 2500.0        # dense f16 / bf16 matrix peak (no sparsity)
PEAK_MFMA = {"f32": PEAK_F32_MFMA_TFLOPS, "f16x3": PEAK_F16_MFMA_TFLOPS, "f16": PEAK_F16_MFMA_TFLOPS,
             "bf16": PEAK_F16_MFMA_TFLOPS}
KERNEL_NAME = {"f32": "branch_kernel<51>", "f16x3": "branch_kernel_h<51,f16x3>", "f16": "branch_kernel_h<51,f16>",
               "bf16": "branch_kernel_h<51,bf16>"}
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r1_pmc_traffic.json")


def measured_traffic(precision: str, sites_per_launch: float):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes
    (profiles/collect_pmc.sh: 2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction for 16 B/lane
    reads).  PMC counters cannot be read from inside this process; null when no pass was recorded
    for this precision and launch size."""
    try:
        with open(TRAFFIC_FILE) as f:
            rec = json.load(f).get(precision)
    except (OSError, ValueError):
        return None
    if not rec or abs(rec.get("sites_per_launch", 0) - sites_per_launch) > 0.5:
        return None
    return rec.get("traffic_bytes_per_launch")
PEAK_HBM_GBS = 8000.0


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
    esm = torch.empty((n, 1280), device=dev, dtype=torch.float32)
    step = 1 << 17
    for s in range(0, n, step):   # chunked: keeps the temporaries small
        e = min(n, s + step)
        blk = torch.randn((e - s, 1280), generator=g, device=dev) * 0.27
        heavy = torch.rand((e - s, 1280), generator=g, device=dev) < 0.01
        esm[s:e] = torch.where(heavy, blk * 10.0, blk)
    st = torch.zeros((n, 8), device=dev, dtype=torch.float32)
    st[:, 0] = torch.rand(n, generator=g, device=dev) * 250
    st[:, 1] = torch.rand(n, generator=g, device=dev) * 360 - 180
    st[:, 2] = torch.rand(n, generator=g, device=dev) * 360 - 180
    ss = torch.randint(0, 3, (n,), generator=g, device=dev)
    st[torch.arange(n, device=dev), 3 + ss] = 1.0
    st[:, 6] = torch.rand(n, generator=g, device=dev) * 80 + 20
    st[:, 7] = torch.rand(n, generator=g, device=dev) * 80 + 20
    return toks + [st, esm]


def cpu_baseline(weights, budget_s: float = 15.0):
    """Times oracle/torch_ref.py (fp32, batch 4096) on the host cores for ~budget_s."""
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
    return {"value": n / dt, "unit": "sites/s", "cores": threads, "kind": "port",
            "sample": f"{n} synthetic sites, torch-CPU fp32 restatement, batch 4096, {dt:.1f} s"}


def train_main(args) -> None:
    """--train: fwd + bwd + gradient all-reduce + AdamW per step on synthetic imbalanced
    sites (labels ~ Bernoulli(0.042), the phosphorylation ratio), one replica per GPU."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local}"))
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")
    from deepptmpred_amd import _native, spec
    from deepptmpred_amd.training import NativeBackend
    b = args.batch
    backend = NativeBackend(spec.synthetic_weights(7), device=local, max_batch=b)
    x = device_inputs(b, 99 + rank, dev)
    g = torch.Generator(device=dev)
    g.manual_seed(5 + rank)
    labels = (torch.rand(b, generator=g, device=dev) < 0.042).to(torch.int32)
    n_pos = torch.tensor([float(labels.sum().item())], device=dev)
    if dist is not None:
        dist.all_reduce(n_pos)
    o = _native.TrainOpts()
    o.rate_embed, o.rate_conv, o.rate_tb, o.rate_dense = 1e-4, 0.3, 0.1, 0.3
    o.n_global, o.n_pos_global, o.l2, o.l2_scale = b * world, int(n_pos.item()), 1e-4, 1.0 / world
    # one explicit stream for the whole step: the library's kernels, RCCL's all-reduce (which orders itself
    # after torch's current stream) and the optimizer update are then stream-ordered without host syncs
    tstream = torch.cuda.Stream(device=dev)
    st = tstream.cuda_stream
    torch.cuda.synchronize()

    def step(i):
        o.seed = i * 1000003 + rank
        with torch.cuda.stream(tstream):
            backend.trainer.forward_backward(x[0].data_ptr(), x[1].data_ptr(), x[2].data_ptr(), x[3].data_ptr(),
                                             x[4].data_ptr(), labels.data_ptr(), b, o, False, st)
            if dist is not None:
                backend.allreduce_grads()
            backend.trainer.apply(5e-4, 1e-4, 1.0, 1.0 / world, st)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        value = world * b * args.steps / elapsed
        print(json.dumps({
            "metric": "training sites/sec (fwd+bwd+all-reduce+AdamW)", "value": value, "unit": "sites/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic imbalanced sites (4.2% positive), seeded synthetic weights",
            "config": {"workload": "config 5: main.py-style training step", "batch_per_gpu": b,
                       "parallelism": f"dp{world} (one flat 6.6 MB fp32 gradient all-reduce per step)"},
            "whole_path": {"tflops": value / world * 3 * FLOPS_PER_SITE / 1e12}}))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def sweep_main(args) -> None:
    """--types T: the same resident sites scored by T weight sets (BASELINE config 4: every PTM type has
    its own model, predict.py:119-137).  Sites are sharded over ranks, every rank holds all T models
    (6.5 MB each), inputs are read T times from HBM and never re-uploaded.  `value` counts
    (site, type) scorings per second; rank 0 also times the host-array form (one upload per pass
    shared by all types) on a bounded sample."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local}"))
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")
    from deepptmpred_amd import spec
    from deepptmpred_amd.model import Model
    from deepptmpred_amd.sharding import predict_all_types

    T = args.types
    models = {f"type{k:02d}": Model(spec.synthetic_weights(100 + k), device=local) for k in range(T)}
    for m in models.values():
        m.scorer.set_precision(args.precision)
    n = args.sites
    x = device_inputs(n, 1234 + rank, dev)
    outs = [torch.empty((n, 2), device=dev, dtype=torch.float32) for _ in range(T)]
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream().cuda_stream
    ptrs = [t.data_ptr() for t in x]

    def step():
        for m, o in zip(models.values(), outs):
            m.scorer.score_device(ptrs[0], ptrs[1], ptrs[2], ptrs[3], ptrs[4], n, o.data_ptr(), stream)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    for o in outs:
        chk = o[:: max(1, n // 1024)].cpu().numpy()
        assert np.isfinite(chk).all() and np.abs(chk.sum(axis=1) - 1).max() < 1e-5
    # different weight sets must give different answers (the models are really distinct)
    assert float((outs[0][:1024] - outs[-1][:1024]).abs().max().item()) > 0 or T == 1

    streamed = None
    if rank == 0:
        m = min(n, 262144)
        hx = spec.synthetic_inputs(m, 77)
        predict_all_types(models, [a[:4096] for a in hx])            # warm-up: pinned buffers, first launches
        t1 = time.perf_counter()
        res = predict_all_types(models, hx)
        dt = time.perf_counter() - t1
        assert len(res) == T
        streamed = {"site_types_per_s_one_gpu": m * T / dt, "sites": m,
                    "note": "host arrays in, host arrays out; one upload per pass shared by all types"}
    barrier()
    if rank == 0:
        value = world * n * T * args.steps / elapsed
        print(json.dumps({
            "metric": "windows x PTM types scored/sec", "value": value, "unit": "site-types/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision,
            "data": "synthetic (device-generated sites, seeded synthetic weight sets 100..)",
            "config": {"workload": f"config 4: {n:,} resident sites per GPU x {T} weight sets, {args.precision}, "
                                   f"{world} x MI355X", "sites_per_gpu": n, "types": T,
                       "parallelism": f"dp{world} (sites sharded, every rank holds all {T} models, no collective)"},
            "whole_path": {"tflops": value / world * FLOPS_PER_SITE / 1e12,
                           "sites_per_s_all_types": value / T},
            "streamed_from_host": streamed}))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


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
    ap.add_argument("--precision", default="f32", choices=["f32", "f16x3", "f16", "bf16"],
                    help="arithmetic of the sequence-branch matrix products for the headline number")
    ap.add_argument("--train", action="store_true",
                    help="measure the training step (BASELINE config 5) instead of scoring")
    ap.add_argument("--batch", type=int, default=64, help="--train: sites per GPU per step")
    args = ap.parse_args()
    if args.train:
        return train_main(args)
    if args.types > 0:
        return sweep_main(args)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    dist = None
    # rehearsal of the multi-rank path on a one-GPU box: DPTM_BENCH_REHEARSAL=1 puts every rank on GPU 0 and
    # uses gloo for the barrier and the max-over-ranks reduction (RCCL refuses two ranks on one device)
    rehearsal = os.environ.get("DPTM_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device(f"cuda:{local}"))
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")
    rdev = torch.device("cpu") if rehearsal else dev      # where the reduced scalar lives

    from deepptmpred_amd import spec
    from deepptmpred_amd.model import Model

    weights = spec.synthetic_weights(7)
    model = Model(weights, device=local)
    sc = model.scorer
    sc.set_precision(args.precision)
    n = args.sites
    x = device_inputs(n, 1234 + rank, dev)
    out = torch.empty((n, 2), device=dev, dtype=torch.float32)
    torch.cuda.synchronize()     # inputs are complete before any library stream reads them
    stream = torch.cuda.current_stream().cuda_stream
    ptrs = [t.data_ptr() for t in x]

    def step():
        sc.score_device(ptrs[0], ptrs[1], ptrs[2], ptrs[3], ptrs[4], n, out.data_ptr(), stream)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=rdev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # sanity: finite probabilities that sum to one
    chk = out[:: max(1, n // 4096)].cpu().numpy()
    assert np.isfinite(chk).all() and np.abs(chk.sum(axis=1) - 1).max() < 1e-5

    # one extra, profiled step: per-kernel HIP-event timings on the launch stream
    sc.profile_begin()
    step()
    prof = sc.profile_end()
    barrier()

    # the other arithmetic modes on the same resident inputs (rank 0, two steps each): reported
    # beside the headline, never as `value`
    variants = {}
    if rank == 0 and world == 1 and not args.no_variants:
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
            v_ms, v_launches = vp["branch51"]
            v_tf = n * FLOPS_BRANCH[51] / (v_ms * 1e-3) / 1e12 if v_ms > 0 else 0.0
            variants[mode] = {"sites_per_s_one_gpu": n / dt, "branch51_avg_launch_ms": v_ms / max(v_launches, 1),
                              "branch51_tflops_algorithmic": v_tf, "branch51_frac_of_mfma_peak": v_tf / PEAK_MFMA[mode]}
        sc.set_precision(args.precision)
    barrier()

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        value = world * n * args.steps / elapsed
        b51_ms, b51_launches = prof["branch51"]
        sites_per_launch = n / max(b51_launches, 1)
        avg_s = b51_ms * 1e-3 / max(b51_launches, 1)
        achieved = sites_per_launch * FLOPS_BRANCH[51] / avg_s / 1e12 if avg_s > 0 else 0.0
        total_kernel_ms = sum(v[0] for v in prof.values())
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
                         "traffic": measured_traffic(args.precision, sites_per_launch),
                         "traffic_source": "profiles/r1_pmc_traffic.json (offline rocprofv3 --pmc passes of this command)",
                         "avg_launch_ms": avg_s * 1e3, "sites_per_launch": sites_per_launch,
                         "flops_per_site": FLOPS_BRANCH[51]},
            "whole_path": {"tflops": value / world * FLOPS_PER_SITE / 1e12,
                           "frac_of_f32_mfma_peak": value / world * FLOPS_PER_SITE / 1e12 / PEAK_F32_MFMA_TFLOPS,
                           "hbm_gbs_algorithmic": value / world * BYTES_PER_SITE / 1e9,
                           "frac_of_hbm_peak": value / world * BYTES_PER_SITE / 1e9 / PEAK_HBM_GBS},
            "kernel_ms_per_step": {k: round(v[0], 3) for k, v in prof.items()},
            "kernel_ms_total": round(total_kernel_ms, 3),
        }
        if variants:
            line["variants"] = variants
        if not args.no_cpu_baseline and world == 1:      # the CPU leg is timed at N = 1 only
            line["cpu_baseline"] = cpu_baseline(weights)
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

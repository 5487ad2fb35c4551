"""bench.py as a command: every mode prints exactly one JSON line with the fields its readers rely on
(small sizes; the numbers themselves are not judged here)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags, env=None):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], cwd=ROOT, capture_output=True,
                         text=True, timeout=300, env=dict(os.environ, **(env or {})))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_scoring_line():
    d = _run("--sites", "40000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-variants", "--no-extras")
    assert d["metric"] == "windows scored/sec" and d["unit"] == "sites/s" and d["n_gpus"] == 1
    assert d["steps"] == 2 and d["warmup"] == 1 and d["dtype"] == "f32" and d["scaling"] == "weak"
    assert d["value"] > 1e5 and abs(d["value"] - 40000 / (d["ms_per_step"] * 1e-3)) < 1e-3 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and 0 < r["frac"] < 1
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["traffic"] is None          # the recorded PMC passes are for the default launch size only
    assert "workload" in d["config"] and "model" not in d["config"]


def test_precision_and_cpu_baseline_line():
    d = _run("--sites", "20000", "--steps", "1", "--warmup", "1", "--precision", "bf16", "--no-variants", "--no-extras")
    assert d["dtype"] == "bf16" and d["roofline"]["kernel"].startswith("branch_kernel_h<51")
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "sites/s" and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    assert set(c["grid"]) >= {"batch32_threads1", "batch4096_threads1"} and all(v["sites_per_s"] > 0 for v in c["grid"].values())


def test_default_line_carries_the_other_configs():
    """Small sizes of what the default command adds beside the headline: host-streamed rate, proteome call,
    type sweep and training step."""
    d = _run("--sites", "30000", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-variants")
    assert d["streamed_from_host"]["sites_per_s_one_gpu"] > 0
    p = d["proteome"]
    assert p["sites_per_s"] > 0 and p["assemble"]["launches"] >= 1 and p["assemble"]["gbs_algorithmic"] > 0
    assert d["sweep"]["types"] == 3 and d["sweep"]["site_types_per_s"] > 0
    assert d["sweep"]["streamed_from_host"]["site_types_per_s_one_gpu"] > 0
    assert d["train"]["batch64"]["ms_per_step"] > 0 and d["train"]["batch64"]["batch_per_gpu"] == 64


REHEARSAL = {"DPTM_BENCH_REHEARSAL": "1"}      # both ranks on GPU 0, gloo for the barrier and the reductions


def test_gpus_2_launches_its_own_ranks():
    """`bench.py --gpus 2` outside a launcher starts two ranks itself and prints rank 0's single line."""
    d = _run("--gpus", "2", "--sites", "20000", "--steps", "2", "--warmup", "1", env=REHEARSAL)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 2
    assert abs(d["value"] - 2 * 20000 / (d["ms_per_step"] * 1e-3)) < 1e-3 * d["value"]
    assert "cpu_baseline" not in d and "train" not in d          # N = 1 only


def test_gpus_2_training_and_sweep_and_proteome():
    d = _run("--gpus", "2", "--train", "--batch", "64", "--steps", "3", "--warmup", "1", env=REHEARSAL)
    assert d["n_gpus"] == 2 and d["config"]["batch_per_gpu"] == 64 and d["value"] > 0 and d["roofline"]["frac"] > 0
    d = _run("--gpus", "2", "--types", "2", "--sites", "10000", "--steps", "1", "--warmup", "1", env=REHEARSAL)
    assert d["n_gpus"] == 2 and d["config"]["types"] == 2 and d["value"] > 0
    d = _run("--gpus", "2", "--proteome", "--proteins", "100", "--steps", "1", "--warmup", "1", env=REHEARSAL)
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["roofline"]["bound"] == "hbm"


def test_world_size_mismatch_is_refused():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], cwd=ROOT, capture_output=True,
                         text=True, timeout=120, env=dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert out.returncode != 0 and "WORLD_SIZE" in out.stderr


def test_type_sweep_line():
    d = _run("--types", "3", "--sites", "20000", "--steps", "1", "--warmup", "1")
    assert d["unit"] == "site-types/s" and d["config"]["types"] == 3 and d["value"] > 1e5
    assert d["streamed_from_host"]["site_types_per_s_one_gpu"] > 0


def test_training_line():
    d = _run("--train", "--batch", "64", "--steps", "5", "--warmup", "2")
    assert d["metric"].startswith("training sites/sec") and d["config"]["batch_per_gpu"] == 64
    assert d["value"] > 1e3 and d["ms_per_step"] > 0

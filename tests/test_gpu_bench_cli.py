"""bench.py as a command: every mode prints exactly one JSON line with the fields its readers rely on
(small sizes; the numbers themselves are not judged here)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], cwd=ROOT, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_scoring_line():
    d = _run("--sites", "40000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-variants")
    assert d["metric"] == "windows scored/sec" and d["unit"] == "sites/s" and d["n_gpus"] == 1
    assert d["steps"] == 2 and d["warmup"] == 1 and d["dtype"] == "f32" and d["scaling"] == "weak"
    assert d["value"] > 1e5 and abs(d["value"] - 40000 / (d["ms_per_step"] * 1e-3)) < 1e-3 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and 0 < r["frac"] < 1
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["traffic"] is None          # the recorded PMC passes are for the default launch size only
    assert "workload" in d["config"] and "model" not in d["config"]


def test_precision_and_cpu_baseline_line():
    d = _run("--sites", "20000", "--steps", "1", "--warmup", "1", "--precision", "bf16", "--no-variants")
    assert d["dtype"] == "bf16" and d["roofline"]["kernel"].startswith("branch_kernel_h<51")
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "sites/s" and c["cores"] >= 1 and c["value"] > 0 and c["sample"]


def test_type_sweep_line():
    d = _run("--types", "3", "--sites", "20000", "--steps", "1", "--warmup", "1")
    assert d["unit"] == "site-types/s" and d["config"]["types"] == 3 and d["value"] > 1e5
    assert d["streamed_from_host"]["site_types_per_s_one_gpu"] > 0


def test_training_line():
    d = _run("--train", "--batch", "64", "--steps", "5", "--warmup", "2")
    assert d["metric"].startswith("training sites/sec") and d["config"]["batch_per_gpu"] == 64
    assert d["value"] > 1e3 and d["ms_per_step"] > 0

"""The driver's contract with bench.py: one JSON line on stdout with the agreed keys (run as a child process, as the
driver does; tiny step counts -- the numbers themselves are not asserted here)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_train_line_has_the_contract_fields():
    d = _run("--gpus", "1", "--steps", "3", "--warmup", "2", "--cpu-batches", "1")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 2
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and abs(d["value"] - 512 / (d["ms_per_step"] * 1e-3)) < 0.02 * d["value"]
    roof = d["roofline"]
    assert roof["bound"] == "mfma" and roof["unit"] == "TFLOP/s" and roof["peak"] == pytest.approx(157.3)
    assert 0.0 < roof["frac"] < 1.0 and roof["frac"] == pytest.approx(roof["achieved"] / roof["peak"], rel=1e-3)
    assert roof["traffic"] is None or roof["traffic"] > 0
    cpu = d["cpu_baseline"]
    assert cpu["kind"] in ("port", "reference") and cpu["cores"] >= 1 and cpu["value"] > 0 and cpu["sample"]


def test_other_workloads_print_one_line():
    f = _run("--workload", "fbank", "--steps", "3", "--warmup", "1")
    assert f["roofline"]["bound"] == "hbm" and f["value"] > 0
    i = _run("--workload", "infer", "--minutes", "0.5", "--precision", "fp16")
    assert i["higher_is_better"] is False and i["value"] > 0 and i["config"]["windows"] == 3000

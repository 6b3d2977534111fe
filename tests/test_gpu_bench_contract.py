"""bench.py prints one JSON line with the fields the driver reads (small spectrum)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_has_the_contract_fields():
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4",
                                   "--warmup", "1", "--log2n", "20", "--trials", "2"], timeout=280).decode()
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in r, k
    assert r["n_gpus"] == 1 and r["steps"] == 4 and r["warmup"] == 1 and r["scaling"] == "weak"
    assert r["value"] > 0 and abs(r["value"] * r["ms_per_step"] / 1e3 - 1.0) < 1e-6
    assert r["vs_baseline"] is None and r["dtype"] == "f32" and r["data"] == "synthetic"
    assert "workload" in r["config"] and "model" not in r["config"]
    rf = r["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and rf["achieved"] > 0
    cb = r["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == r["unit"]
    assert r["with_optimization_searches_per_s"] > 0

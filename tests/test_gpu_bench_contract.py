"""bench.py prints exactly one JSON line with the agreed fields (run on the smallest configuration)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_fields():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "1", "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "Msamples/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert d["value"] > 0 and d["ms_per_step"] > 0
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["achieved"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and "traffic" in r
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["unit"] == "Msamples/s" and c["value"] > 0 and c["cores"] >= 1 and c["sample"]
    assert d["parity_spot_check"]["pixels_within_2e-3_rel"] > 0.99

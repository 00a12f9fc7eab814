"""bench.py prints one JSON line with the contract's keys (short run on the GPU box)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.gpu
def test_bench_line(gpu):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "50", "--warmup", "10",
                        "--no-finest", "--cpu-samples", "1"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in out, k
    assert out["n_gpus"] == 1 and out["steps"] == 50 and out["dtype"] == "f64" and out["vs_baseline"] is None
    assert "workload" in out["config"] and "model" not in out["config"]
    rf = out["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert out["cpu_baseline"]["kind"] in ("port", "reference") and out["cpu_baseline"]["cores"] >= 1
    assert out["value"] > 0 and out["ms_per_step"] > 0

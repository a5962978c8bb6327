"""bench.py's one-line contract: a short run must print exactly one JSON object with the agreed keys, a throughput that
follows from its own step time, and the roofline object (the CPU-baseline legs are skipped here: they take ~20 s)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "2", "--no-other-configs",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["metric"] == "audio-sec/sec" and d["unit"] == "audio-sec/sec" and d["higher_is_better"] is True
    assert (d["n_gpus"], d["steps"], d["warmup"], d["scaling"], d["dtype"], d["data"]) == (1, 3, 2, "weak", "f32", "synthetic")
    assert d["vs_baseline"] is None and "workload" in d["config"] and "model" not in d["config"]
    # 64 clips of 10 s per step and GPU
    assert d["value"] == pytest.approx(640.0 / (d["ms_per_step"] * 1e-3), rel=1e-3)
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"], rel=1e-3) and 0.2 < rf["frac"] < 1.0
    assert rf["traffic"] is None or rf["traffic"] > 0.5 * rf["algorithmic_bytes_per_launch"]
    assert "source" in rf["measured_peaks"] and "traffic_source" in rf

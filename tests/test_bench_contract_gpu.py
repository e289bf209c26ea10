"""bench.py's output contract (one JSON line with the driver's fields + roofline + cpu_baseline)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_bench_json_contract_small_batch():
    d = run_bench("--boards", "65536", "--steps", "40", "--warmup", "5")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"] == "env_steps_per_sec" and d["unit"] == "steps/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["steps"] == 40 and d["warmup"] == 5 and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    assert d["config"]["replay_matches_recording"] is True
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["algorithmic_bytes_per_launch"] == 47 * 65536
    assert abs(d["value"] - 65536 * 40 / (d["ms_per_step"] * 1e-3 * 40)) / d["value"] < 1e-6
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "steps/s" and c["cores"] >= 1 and c["value"] > 1e6
    assert c["python_interpreter_steps_per_s"] > 1e4


def test_bench_modes_agree_on_the_episode_counters():
    a = run_bench("--boards", "16384", "--steps", "30", "--warmup", "5", "--no-cpu-baseline")
    b = run_bench("--boards", "16384", "--steps", "30", "--warmup", "5", "--no-cpu-baseline", "--mode", "random")
    assert a["config"]["episodes_finished"] == b["config"]["episodes_finished"] > 0
    assert b["config"]["replay_matches_recording"] is True

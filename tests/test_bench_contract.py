"""bench.py prints ONE JSON line with the driver's contract keys (+ roofline, cpu_baseline)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric": str, "value": float, "unit": str, "n_gpus": int, "steps": int, "warmup": int, "ms_per_step": float,
            "higher_is_better": bool, "scaling": str, "dtype": str, "data": str, "config": dict, "roofline": dict}


def run_bench(*extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "52", "--warmup", "26", *extra],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, "bench.py must print exactly one line on stdout, got %d" % len(lines)
    return json.loads(lines[0])


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [("--cpu-seconds", "1"), ("--env", "tennis", "--no-cpu-baseline"), ("--no-graph", "--no-pipeline", "--no-cpu-baseline")])
def test_bench_json_contract(extra):
    d = run_bench(*extra)
    for k, t in REQUIRED.items():
        assert k in d, k
        assert isinstance(d[k], t) or (t is float and isinstance(d[k], int)), (k, type(d[k]))
    assert d["n_gpus"] == 1 and d["steps"] == 52 and d["warmup"] == 26 and d["vs_baseline"] is None
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 4096 * 52 / (d["ms_per_step"] * 52e-3)) / d["value"] < 1e-6
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0 < r["frac"] < 1
    if "cpu_baseline" in d:
        c = d["cpu_baseline"]
        assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c and c["unit"] == "env steps/s"


@pytest.mark.gpu
@pytest.mark.parametrize("steps,expect_tuning", [("208", True), ("200", False)])
def test_bench_exchange_forms_with_one_rank(steps, expect_tuning):
    """the multi-rank exchange path rehearsed with a single rank (TB_BENCH_FORCE_COLLECTIVE=1: RCCL is initialised and every
    collective is issued): with K a whole number of episodes the three exchange forms are tried and one is timed; otherwise
    the pipelined graph can run only once, nothing is tuned, and the single all-gather is timed"""
    env = dict(os.environ, TB_BENCH_FORCE_COLLECTIVE="1", TB_BENCH_FAKE_GATHER_US="1000")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", steps, "--warmup", "26", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.strip()][-1])
    w = d["config"]["workload"]
    assert d["value"] > 0 and "all-gather" in w
    assert ("exchange form chosen on this node" in w) == expect_tuning
    if not expect_tuning:
        assert "1 RCCL all-gather of rollouts at the collect boundary" in w

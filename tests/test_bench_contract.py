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
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "52", "--warmup", "20", *extra],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, "bench.py must print exactly one line on stdout, got %d" % len(lines)
    return json.loads(lines[0])


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [("--cpu-seconds", "1"), ("--env", "tennis", "--no-cpu-baseline", "--no-sweep"),
                                   ("--no-graph", "--no-pipeline", "--no-cpu-baseline", "--no-sweep", "--rollout-steps", "104")])
def test_bench_json_contract(extra):
    d = run_bench(*extra)
    for k, t in REQUIRED.items():
        assert k in d, k
        assert isinstance(d[k], t) or (t is float and isinstance(d[k], int)), (k, type(d[k]))
    assert d["n_gpus"] == 1 and d["steps"] == 52 and d["warmup"] == 20 and d["vs_baseline"] is None
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"] and "invalid" not in d
    # K = 52 is rounded up to whole rollouts (whole episodes): what is timed is the headline workload whatever K says
    T = d["rollout_steps"]
    assert d["steps_timed"] == T * d["rollouts_timed"] >= 52 and T == (104 if "--rollout-steps" in extra else 1040)
    assert abs(d["value"] - 4096 * d["steps_timed"] / (d["ms_per_step"] * d["steps_timed"] * 1e-3)) / d["value"] < 1e-6
    swing = "tennis" not in extra
    if swing:
        # warm-up: the 20 asked for rounded up to an episode, then whole untimed rollouts (one from prepare(), the rest until
        # --settle-seconds have passed): the timed rollout starts at an episode start, in the process's steady state
        assert d["steps_timed"] % 26 == 0 and d["warmup_run"] >= 26 + T and (d["warmup_run"] - 26) % T == 0
        assert abs(d["substeps_per_agent_step"] / 5.09 - 1) < 0.05
        assert d["timed_region_ms"] >= 5.0 or T == 104
    else:
        assert d["substeps_per_agent_step"] == 1.0
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0 < r["frac"] < 1
    assert 0 < r["read_only"]["frac"] < r["frac"]
    if "cpu_baseline" in d:
        c = d["cpu_baseline"]
        assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c and c["unit"] == "env steps/s"
        assert abs(c["substeps_per_agent_step"] / d["substeps_per_agent_step"] - 1) < 0.05   # GPU and CPU legs run the same workload
        rr = c["reference_record"]
        assert 100 < rr["agent_steps_per_s_collect"] < 400 and rr["agent_steps_per_s_overall"] < rr["agent_steps_per_s_collect"]
    if "--no-sweep" not in extra:
        sw = {(e["env"], e["envs"]): e for e in d["sweep"] if "variant" not in e}
        assert set(sw) == {("swing", 4096), ("swing", 1048576), ("tennis", 4096), ("tennis", 1048576)}
        lit = [e for e in d["sweep"] if "variant" in e]  # BASELINE configs[1] as worded (racket<->ball contact off), next to the full-contact headline
        assert len(lit) == 1 and lit[0]["env"] == "swing" and lit[0]["envs"] == 4096 and "configs[1]" in lit[0]["variant"]
        assert all(0 < e["read_frac"] < e["frac"] < 1 for e in sw.values())
        assert abs(sw[("swing", 1048576)]["substeps_per_agent_step"] / 5.09 - 1) < 0.05


@pytest.mark.gpu
@pytest.mark.parametrize("chunks", ["8", "1"])
def test_bench_exchange_forms_with_one_rank(chunks):
    """the multi-rank exchange path rehearsed with a single rank (TB_BENCH_FORCE_COLLECTIVE=1: RCCL is initialised and every
    collective is issued): the chunked, overlapped form (default) and the single all-gather, with the diagnostics an N > 1 line carries"""
    env = dict(os.environ, TB_BENCH_FORCE_COLLECTIVE="1", TB_BENCH_FAKE_GATHER_US="1000")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "208", "--rollout-steps", "208", "--warmup", "26", "--no-cpu-baseline",
                          "--gather-chunks", chunks], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.strip()][-1])
    w, x = d["config"]["workload"], d["exchange"]
    assert d["value"] > 0 and "invalid" not in d and "all-gather" in w
    assert x["ranks_seen"] == 1 and x["bytes_per_rank"] >= 208 * 4096 * 53 and x["rollout_ms"] > 0 and x["exchange_ms"] > 0.9 and x["exposed_exchange_ms"] >= 0
    assert ("step-chunks" in x["form"]) == (chunks == "8") and not x["note"]


def test_bench_launches_its_own_ranks_dry():
    """`python bench.py --gpus 2` with no launcher around it: the process becomes the launcher (it never imports torch), starts two
    ranks, relays ONE line and exits with their status. Here, without a GPU, the ranks run the dry leg (TB_BENCH_DRY_RUN=1):
    rendezvous, head count and the rollout exchange over gloo in both forms; the value is null and says why."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["TB_BENCH_DRY_RUN"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["warmup"] == 5 and d["value"] is None and d["dry_run"] is True and "dry run" in d["invalid"]
    x = d["exchange"]
    assert x["ranks_seen"] == 2 and x["gathered_ok"] is True and set(x["exchange_ms_by_chunks"]) == {"8", "1"}


def test_bench_launcher_passes_a_failing_rank_on():
    """a rank that dies must not leave a line that looks like a result: non-zero exit, nothing on stdout"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(TB_BENCH_DRY_RUN="1", TB_BENCH_DRY_FAIL_RANK="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode != 0 and not out.stdout.strip() and "2-rank run ended with status" in out.stderr


@pytest.mark.gpu
def test_bench_launches_its_own_ranks_on_one_gpu():
    """the same entry point with the envs: `python bench.py --gpus 2`, the two ranks sharing cuda:0 over gloo (TB_BENCH_REHEARSAL=1:
    RCCL refuses two ranks on one device; the rate means nothing, the control flow is the multi-GPU run's): one line, two ranks
    seen, both exchange forms timed"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["TB_BENCH_REHEARSAL"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "208", "--rollout-steps", "208", "--warmup", "26",
                          "--settle-seconds", "0.2"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    x = d["exchange"]
    assert d["n_gpus"] == 2 and d["value"] > 0 and "invalid" not in d and "replicas_only" not in d
    assert d["config"]["global_envs"] == 8192 and abs(d["substeps_per_agent_step"] / 5.09 - 1) < 0.05
    assert x["ranks_seen"] == 2 and "step-chunks" in x["form"] and x["exchange_ms"] > 0
    one = x["single_all_gather"]
    assert one["value"] > 0 and one["exchange_ms"] > 0 and "ONE all-gather" in one["form"]
    assert "cpu_baseline" not in d and "sweep" not in d  # N = 1 only


def test_bench_reports_one_fixed_exchange_form():
    """with N > 1 bench.py times the chunked, overlapped exchange (the default) and ONE all-gather after the rollout; `value` is always
    the default form's -- not the faster of two noisy measurements -- unless its gathered shards failed their check and the other's
    did not; and the line carries what the exchange should cost over xGMI, computed before any node was measured"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    pick = bench.pick_exchange_form
    assert pick(7.0, True, 5.5, True) == "chunked" and pick(5.0, True, 5.5, True) == "chunked"
    assert pick(7.0, True, 5.5, False) == "chunked" and pick(5.0, False, 5.5, True) == "single_all_gather"
    assert pick(5.0, False, 5.5, False) == "chunked"  # (reported as invalid by the caller)
    # 226 MB per rank, a 3.8 ms rollout, 8 GPUs: the direct all-gather is 1.5 ms -> single 72 %, chunked (8) 95 %
    p = bench.predict_exchange(8, 226e6, 3.8, 8)
    assert abs(p["direct"]["all_gather_ms"] - 226e6 / 153e9 * 1e3) < 1e-9 and abs(p["ring"]["all_gather_ms"] / p["direct"]["all_gather_ms"] - 7) < 1e-9
    assert 0.70 < p["direct"]["single_all_gather"]["efficiency"] < 0.74 and 0.94 < p["direct"]["chunked"]["efficiency"] < 0.96
    assert p["ring"]["single_all_gather"]["efficiency"] < 0.3 and p["ring"]["chunked"]["efficiency"] < p["direct"]["chunked"]["efficiency"]
    # against the N = 1 rollout (pool form, 3.8 ms) a chunked form whose own marked rollout takes 5.8 ms starts at 65 %
    p2 = bench.predict_exchange(8, 226e6, 5.8, 8, 3.8)
    assert 0.62 < p2["direct"]["chunked"]["efficiency_vs_n1"] < 0.66 and p2["direct"]["chunked"]["efficiency"] > 0.95
    assert abs(p2["direct"]["single_all_gather"]["efficiency_vs_n1"] - p["direct"]["single_all_gather"]["efficiency"]) < 1e-12
    assert bench.predict_exchange(1, 226e6, 3.8, 8) is None
    c = bench.cadence_profile("swing", 4096)
    assert c is not None and 1.0 < c[0]["kernel_us"] < 3.0 and 0.8 < c[0]["gap_us"] < 2.5 and c[1].endswith("cadence.json")
    assert abs(c[0]["cadence_build_vs_product"] - 1.0) < 0.05  # the probe's build replays at the product's rate


def test_bench_workload_constants_match_the_oracle():
    """what bench.py refuses a value against: the random-action SwingRacket workload's substeps per agent step (whole episodes),
    re-measured here on the CPU oracle; and the reference's own wall-clock record it quotes next to the CPU baseline"""
    import importlib.util
    import numpy as np
    from oracle import OracleBatch
    from tennisbot_rl_amd.params import ENV_SWING, F_AUTO_RESET, F_DEFAULT, default_params
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    n, rng = 2048, np.random.Generator(np.random.PCG64(0))
    b = OracleBatch(default_params(flags=F_DEFAULT | F_AUTO_RESET), ENV_SWING, n, seed=0, precision="f32", threads=4)
    b.reset()
    for t in range(26 * 3):
        b.step(rng.uniform(-1.0, 1.0, (n, 6)).astype(np.float32))
    sps = b.counters()[6] / (n * 26 * 3)
    assert abs(sps / bench.EXPECTED_SUBSTEPS[("swing", False)] - 1.0) < 0.02, sps
    assert bench.ALGO_BYTES["swing"] == {"read": 145, "write": 122} and bench.ALGO_BYTES["tennis"] == {"read": 117, "write": 146}
    rr = bench.reference_record()
    assert rr["episodes"] == 100 and 150 < rr["agent_steps_per_s_overall"] < rr["agent_steps_per_s_collect"] < 200


def test_rccl_failure_falls_back_to_labelled_replicas(tmp_path):
    """SURVEY.md 8e's fallback: where RCCL cannot be initialised bench.py's ranks still join their clocks (gloo, on the launcher's
    rendezvous store) and report 'replicas only' instead of dying. Two ranks under the driver's launcher; there is no GPU in this
    container, so the nccl group fails exactly where it would on a broken node."""
    import subprocess
    import sys
    script = tmp_path / "probe.py"
    script.write_text(
        "import importlib.util, os, sys\n"
        "import torch\n"
        "spec = importlib.util.spec_from_file_location('bench', os.path.join(%r, 'bench.py'))\n"
        "bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)\n"
        "why = bench.init_distributed(torch, torch.device('cuda', 0), int(os.environ['WORLD_SIZE']))\n"
        "assert why, 'an nccl group without GPUs cannot have come up'\n"
        "t = torch.ones(1); torch.distributed.all_reduce(t); assert int(t.item()) == 2\n"
        "torch.distributed.barrier(); torch.distributed.destroy_process_group()\n"
        "print('rank', os.environ['RANK'], 'replicas only:', why[:60])\n" % ROOT)
    port = 29700 + os.getpid() % 200
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), str(script)], capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert out.stdout.count("replicas only:") == 2

#!/usr/bin/env python3
"""tb_rollout -- T steps per launch, the state in registers between them, actions known up front -- against the one-launch-per-step
graph the headline measures (GPU). SwingRacket-v0 (pipelined: one launch per 26-step episode + the pool at the join) and
Tennisbot-v0 (one launch per 104 steps), 4096 envs, whole rollouts of 1040 steps, captured in one hipGraph each.
Writes gpurun_out/r04_rollout_rate.json."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from tennisbot_rl_amd.stepper import BatchedEnv
from tennisbot_rl_amd.params import ENV_SWING, ENV_TENNIS

out = {}
n, T = 4096, 1040
rng = np.random.Generator(np.random.PCG64(0))
acts = torch.from_numpy(rng.uniform(-1, 1, (T, n, 6)).astype(np.float32)).cuda()
for kind, name, chunk in ((ENV_SWING, "swing", 26), (ENV_TENNIS, "tennis", 104), (ENV_TENNIS, "tennis_26", 26)):
    env = BatchedEnv(kind, n, device="cuda:0", seed=0, pipeline=(kind == ENV_SWING), track_terminal_obs=False, reuse_buffers=False)
    env.reset()
    A = acts if kind == ENV_SWING else acts[:, :, :env.act_dim].contiguous()
    def run():
        for t0 in range(0, T, chunk):
            env.rollout(A[t0:t0 + chunk])
        if kind == ENV_SWING:
            env.flush()
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    ts = []
    for rep in range(10):
        torch.cuda.synchronize(); t0 = time.perf_counter(); run(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    eager = n * T / float(np.median(ts)) / 1e6
    row = {"steps_per_launch": chunk, "eager_M_steps_per_s": eager, "eager_ms": float(np.median(ts)) * 1e3}
    try:
        g = env.capture(run)
        for _ in range(3): g.replay()
        torch.cuda.synchronize()
        ts = []
        for rep in range(10):
            torch.cuda.synchronize(); t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        row["graph_M_steps_per_s"] = n * T / float(np.median(ts)) / 1e6
        row["graph_ms"] = float(np.median(ts)) * 1e3
    except Exception as e:
        row["graph_error"] = str(e)[:200]
    c = env.counters()
    row["substeps_per_agent_step"] = c["substeps"] / max(1, env._steps_issued * n) if hasattr(env, "_steps_issued") else None
    out[name] = row
    print(name, json.dumps(row), flush=True)
    env.close()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r04_rollout_rate.json"), "w"), indent=1)

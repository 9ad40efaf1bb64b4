#!/usr/bin/env python3
"""PPO collect on Tennisbot-v0 (GPU): the fused policy rollout kernels (MlpPolicy 12 -> 64 -> 64 towers inside the launch), 4096 and
16384 envs, untrained policy, n_steps = 1000 (one Tennisbot episode, tennisbot_env.py:208). Writes gpurun_out/r04_tennis_collect.json."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from tennisbot_rl_amd.ppo import PPOTrainer

out = {}
for n, n_steps in ((4096, 1000), (16384, 500)):
    tr = PPOTrainer("Tennisbot-v0", num_envs=n, n_steps=n_steps, seed=0)
    for _ in range(3): tr.collect()
    ts = []
    for rep in range(8):
        torch.cuda.synchronize(); t0 = time.perf_counter(); tr.collect(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    c = tr.env.counters()
    row = {"collect_ms": float(np.median(ts)) * 1e3, "collect_M_steps_per_s": n * n_steps / float(np.median(ts)) / 1e6, "fused": bool(tr.fused), "rollout_launch": bool(tr.rollout_launch),
           "episodes_finished": c["episodes_finished"], "racket_ball_contact_substeps": c["racket_ball_contact_substeps"]}
    out[str(n)] = row
    print(n, n_steps, json.dumps(row), flush=True)
    del tr; torch.cuda.empty_cache()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r04_tennis_collect.json"), "w"), indent=1)

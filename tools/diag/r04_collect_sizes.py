#!/usr/bin/env python3
"""PPO collect under the reference's trained policy at larger batches (GPU): which pipeline form serves it -- the automatic one
(above 16384 envs: one fast-forward kernel per episode end, as long as its slowest flight), deferred stragglers (ff_defer = 1: the
pool, with its sealed-fate exit), or everything in the pool (ff_defer = 2, up to 131072 envs)? Writes gpurun_out/r04_collect_sizes.json."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from tennisbot_rl_amd.ppo import PPOTrainer

out = {}
for n, n_steps in ((16384, 520), (32768, 260), (65536, 260), (131072, 104)):
    for label, opts in (("auto", {}), ("stragglers", {"ff_defer": True}), ("pool", {"ff_defer": "all"}), ("stragglers_no_seal", {"ff_defer": True, "ff_seal": False})):
        try:
            tr = PPOTrainer("SwingRacket-v0", num_envs=n, n_steps=n_steps, seed=0, options=opts)
        except Exception as e:  # a form this size does not have
            print(n, label, "unavailable:", str(e)[:100], flush=True)
            continue
        tr.policy.load_sb3_arrays(dict(np.load(os.path.join(ROOT, "tests", "golden", "ppo_swing_policy.npz"))))
        for _ in range(3): tr.collect()
        ts = []
        for rep in range(6):
            torch.cuda.synchronize(); t0 = time.perf_counter(); tr.collect(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        row = {"form": tr.env.pipeline_form() if hasattr(tr.env, "pipeline_form") else None, "collect_ms": float(np.median(ts)) * 1e3,
               "collect_M_steps_per_s": n * n_steps / float(np.median(ts)) / 1e6}
        out.setdefault(str(n), {})[label] = row
        print(n, n_steps, label, json.dumps(row), flush=True)
        del tr
        torch.cuda.empty_cache()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r04_collect_sizes.json"), "w"), indent=1)

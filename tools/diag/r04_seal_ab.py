#!/usr/bin/env python3
"""The pool's sealed-fate exit (TbOptions.ff_seal) on and off, same process, same box: PPO collect of n_steps = 1100 x 4096 envs under the
reference's trained policy and under the untrained one, and the random-action headline rollout. Writes gpurun_out/r04_seal_ab.json."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from tennisbot_rl_amd.ppo import PPOTrainer

def sync(): torch.cuda.synchronize()
out = {}
for label in ("reference_policy", "untrained"):
    for seal in (True, False, True, False):
        tr = PPOTrainer("SwingRacket-v0", num_envs=4096, n_steps=1100, seed=0, options={"ff_seal": seal})
        if label == "reference_policy":
            tr.policy.load_sb3_arrays(dict(np.load(os.path.join(ROOT, "tests", "golden", "ppo_swing_policy.npz"))))
        for _ in range(3): tr.collect()
        sync(); tr.env.counters_reset()
        ts = []
        for rep in range(8):
            sync(); t0 = time.perf_counter(); tr.collect(); sync(); ts.append(time.perf_counter() - t0)
        c = tr.env.counters()
        row = {"collect_ms": float(np.median(ts)) * 1e3, "collect_M_steps_per_s": 4096 * 1100 / float(np.median(ts)) / 1e6,
               "substeps": c["substeps"], "timeouts": c["timeouts"], "episodes": c["episodes_finished"], "sealed_substeps": tr.env.sealed_substeps()}
        out.setdefault(label, {}).setdefault("seal_on" if seal else "seal_off", []).append(row)
        print(label, seal, json.dumps(row), flush=True)
        del tr
        torch.cuda.empty_cache()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r04_seal_ab.json"), "w"), indent=1)

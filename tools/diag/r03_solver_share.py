#!/usr/bin/env python3
"""What the contact SOLVER costs a PPO collect under the reference's trained policy (GPU box): the same collect with the sweep cap of
the sequential-impulse solver at 50 (the product), 8, 4, 2, 1. Timing only -- a capped solve ends contacts differently, so the
trajectories (and the share of contact substeps) drift with it; the rollout kernels' time per launch is the figure to read."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from tennisbot_rl_amd.params import default_params  # noqa: E402
from tennisbot_rl_amd.ppo import PPOTrainer  # noqa: E402

for iters in (50, 8, 4, 2, 1, 50):
    tr = PPOTrainer("SwingRacket-v0", num_envs=4096, n_steps=1092, seed=0, params=default_params(solver_iters=iters))
    tr.policy.load_sb3_arrays(dict(np.load(os.path.join(ROOT, "tests", "golden", "ppo_swing_policy.npz"))))
    for _ in range(10):
        tr.collect()
    ts = []
    for _ in range(9):
        torch.cuda.synchronize(); t0 = time.perf_counter(); tr.collect(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts.sort()
    c = tr.env.counters()
    print(json.dumps({"solver_iters": iters, "collect_ms": round(ts[4] * 1e3, 3), "M_steps_per_s": round(4096 * 1092 / ts[4] / 1e6, 1),
                      "racket_contact_substeps": c.get("racket_contacts"), "substeps": c["substeps"]}), flush=True)
    del tr
    torch.cuda.empty_cache()

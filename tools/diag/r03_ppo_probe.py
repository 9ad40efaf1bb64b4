#!/usr/bin/env python3
"""PPO collect rate on the batched SwingRacket envs (GPU): at the trainer's default rollout (104 steps) and at the
reference's n_steps = 1100 (train_swing.py:49-50); optional learning curve (`curve`)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from tennisbot_rl_amd.ppo import PPOTrainer

out = {}
for n_steps, rollouts in ((104, 12), (1100, 5)):
    tr = PPOTrainer("SwingRacket-v0", num_envs=4096, n_steps=n_steps, seed=0)
    hist = tr.learn(rollouts * n_steps * 4096, log=None)
    rates = sorted(h["collect_steps_per_s"] for h in hist[2:])
    out["n_steps_%d" % n_steps] = {"graph": bool(tr.use_graph), "collect_steps_per_s_median": rates[len(rates) // 2], "collect_steps_per_s_max": rates[-1],
                                   "update_s_median": sorted(h["update_s"] for h in hist[2:])[len(hist[2:]) // 2],
                                   "mean_episode_reward": [round(h["mean_episode_reward"], 2) for h in hist]}
    print(n_steps, json.dumps(out["n_steps_%d" % n_steps]), flush=True)
    del tr
    torch.cuda.empty_cache()
if "curve" in sys.argv[1:]:
    tr = PPOTrainer("SwingRacket-v0", num_envs=4096, n_steps=104, seed=0)
    t0 = time.perf_counter()
    hist = tr.learn(62 * 104 * 4096, log=None)
    out["curve_104"] = {"seconds": time.perf_counter() - t0, "timesteps": tr.num_timesteps, "mean_episode_reward": [round(h["mean_episode_reward"], 2) for h in hist],
                        "collect_steps_per_s_median": sorted(h["collect_steps_per_s"] for h in hist)[len(hist) // 2], "update_s_median": sorted(h["update_s"] for h in hist)[len(hist) // 2]}
    print("curve", json.dumps(out["curve_104"]), flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r03_ppo_probe.json"), "w"), indent=1)

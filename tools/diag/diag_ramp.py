#!/usr/bin/env python3
"""How long does a fresh process take to reach its steady replay rate? The 1040-step SwingRacket graph at 4096 envs replayed
200 times back to back (1.3 s); rates of replays 1-10, 11-20, 41-50, 91-100, 191-200 (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tennisbot_rl_amd.params import ENV_SWING
from tennisbot_rl_amd.rollout import RolloutBuffer
from tennisbot_rl_amd.stepper import BatchedEnv
dev = torch.device("cuda", 0)
env = BatchedEnv(ENV_SWING, 4096, device=dev, seed=0, track_terminal_obs=False, pipeline=True)
buf = RolloutBuffer(ENV_SWING, 1040, 4096, dev); buf.actions.uniform_(-1, 1); buf.bind(env); env.reset()
for t in range(26): buf.step_into(env, t)
env.flush()
g = env.capture(lambda: buf.step_range(env, 0, 1040))
torch.cuda.synchronize()
out = []
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for k in range(N):
    t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); out.append(time.perf_counter() - t0)
med = lambda v: sorted(v)[len(v) // 2]
print(" ".join("%.0f" % (4096 * 1040 / med(out[a:a + 50]) / 1e6) for a in range(0, N, 50)), "(M env steps/s, median of each 50 replays = 0.32 s)")

#!/usr/bin/env python3
"""Is the SwingRacket graph's rate mode a property of the graph INSTANCE? One env, the same 1040-step rollout captured eight
times; median rate of 12 replays of each instance, twice round."""
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
t0 = time.perf_counter()
graphs = [env.capture(lambda: buf.step_range(env, 0, 1040)) for _ in range(8)]
print("8 captures: %.0f ms each" % ((time.perf_counter() - t0) / 8 * 1e3))
for rnd in range(2):
    rates = []
    for g in graphs:
        out = []
        for k in range(12):
            t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); out.append(time.perf_counter() - t0)
        out.sort(); rates.append(4096 * 1040 / out[6] / 1e6)
    print("round %d:" % rnd, " ".join("%.0f" % r for r in rates))

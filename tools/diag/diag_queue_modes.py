#!/usr/bin/env python3
"""Does the SwingRacket graph's rate depend on which hardware queues its streams landed on? One process, several envs created
one after the other with 0..k dummy streams created in between (HIP maps streams onto its hardware queues round-robin); per env
the median rate of 20 replays of the 1040-step rollout graph at 4096 envs (GPU box)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tennisbot_rl_amd.params import ENV_SWING
from tennisbot_rl_amd.rollout import RolloutBuffer
from tennisbot_rl_amd.stepper import BatchedEnv
dev = torch.device("cuda", 0)
keep = []
def run(tag):
    env = BatchedEnv(ENV_SWING, 4096, device=dev, seed=0, track_terminal_obs=False, pipeline=True)
    buf = RolloutBuffer(ENV_SWING, 1040, 4096, dev); buf.actions.uniform_(-1, 1); buf.bind(env); env.reset()
    for t in range(26): buf.step_into(env, t)
    env.flush()
    g = env.capture(lambda: buf.step_range(env, 0, 1040))
    torch.cuda.synchronize()
    out = []
    for k in range(20):
        t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); out.append(time.perf_counter() - t0)
    out.sort()
    print("%-40s median %.0f M env steps/s" % (tag, 4096 * 1040 / out[10] / 1e6), flush=True)
    keep.append((env, buf, g))   # keep its streams alive: the next env's streams get the next queue slots
for k in range(8):
    run("env %d (after %d dummy streams)" % (k, k))
    keep.append(torch.cuda.Stream())

#!/usr/bin/env python3
"""Per-process rate mode vs workgroup shape: in ONE process (one mode) the SwingRacket graph at 4096 envs with the step kernel
as 64 one-wave workgroups (default), 32 of two waves, 16 of four; and the non-pipelined kernel (fast-forward inside the step)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tennisbot_rl_amd.params import ENV_SWING
from tennisbot_rl_amd.rollout import RolloutBuffer
from tennisbot_rl_amd.stepper import BatchedEnv
dev = torch.device("cuda", 0)
def run(tag, pipeline=True, **opts):
    env = BatchedEnv(ENV_SWING, 4096, device=dev, seed=0, track_terminal_obs=False, pipeline=pipeline, options=opts or None)
    buf = RolloutBuffer(ENV_SWING, 1040, 4096, dev); buf.actions.uniform_(-1, 1); buf.bind(env); env.reset()
    for t in range(26): buf.step_into(env, t)
    env.flush()
    g = env.capture(lambda: buf.step_range(env, 0, 1040))
    torch.cuda.synchronize()
    out = []
    for k in range(20):
        t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); out.append(time.perf_counter() - t0)
    out.sort()
    print("%-34s median %.0f M" % (tag, 4096 * 1040 / out[10] / 1e6), end=" | ", flush=True)
    env.close()
run("block 64"); run("block 128", block=128); run("block 256", block=256); run("block 64 again"); run("not pipelined", pipeline=False)
print()

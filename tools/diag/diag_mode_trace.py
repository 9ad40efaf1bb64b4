#!/usr/bin/env python3
"""Per-process rate mode of the SwingRacket graph: run under `rocprofv3 --kernel-trace`, replays the 1040-step graph 20 times and
prints the median rate (host clock); tools/diag/diag_mode_summarize.py turns the trace into step-kernel durations and gaps."""
import sys, time, os
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
for k in range(20):
    t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); out.append(time.perf_counter() - t0)
out.sort()
print("median %.0f M env steps/s" % (4096 * 1040 / out[10] / 1e6), flush=True)

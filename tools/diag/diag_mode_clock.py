#!/usr/bin/env python3
"""Is the per-process rate mode a clock state? Replays the SwingRacket graph for ~2 s while `rocm-smi` samples clocks and power."""
import os, subprocess, sys, time
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
p = subprocess.Popen("sleep 0.7; rocm-smi --showclocks --showpower 2>/dev/null | grep -i 'sclk\\|fclk\\|mclk\\|socclk\\|Power (W)'", shell=True, stdout=subprocess.PIPE, text=True)
out = []
t_end = time.perf_counter() + 2.0
while time.perf_counter() < t_end:
    t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); out.append(time.perf_counter() - t0)
out.sort()
print("median %.0f M env steps/s over %d replays" % (4096 * 1040 / out[len(out) // 2] / 1e6, len(out)))
print(p.communicate()[0].strip())

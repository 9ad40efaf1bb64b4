#!/usr/bin/env python3
"""kernel-trace helper: one SwingRacket env batch, a few eager episodes with a given TbOptions set (run under rocprofv3)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tennisbot_rl_amd.params import ENV_SWING, F_DEFAULT, F_RACKET_GROUND, default_params
from tennisbot_rl_amd.stepper import BatchedEnv
n = int(sys.argv[1]); opts = json.loads(sys.argv[2]); rg = len(sys.argv) > 3 and sys.argv[3] == "rg"
env = BatchedEnv(ENV_SWING, n, device="cuda:0", seed=0, pipeline=True, track_terminal_obs=False, options=opts,
                 params=default_params(flags=F_DEFAULT | (F_RACKET_GROUND if rg else 0)))
env.reset()
g = torch.Generator(device="cuda:0"); g.manual_seed(1)
for ep in range(6):
    for t in range(26):
        env.step(torch.rand((n, 6), device="cuda:0", generator=g) * 2 - 1)
    env.flush(); torch.cuda.synchronize()
print(env.counters())

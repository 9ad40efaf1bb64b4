#!/usr/bin/env python3
"""The step kernels ALONE at 1 M envs (GPU box): steps 1..25 of a SwingRacket episode -- no fast-forward in flight, the last one joined
before the clock starts -- timed with HIP events, per TbOptions variant (workgroup size, static rows in registers or LDS), and
Tennisbot's step kernel beside them. usage: r03_step1m_probe.py [lib=<other build>] [n=<envs>]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from tennisbot_rl_amd import stepper  # noqa: E402

for x in sys.argv[1:]:
    if x.startswith("lib="):
        stepper.use_library(x[4:])
n = next((int(x[2:]) for x in sys.argv[1:] if x.startswith("n=")), 1048576)
from tennisbot_rl_amd.params import ENV_SWING, ENV_TENNIS, F_DEFAULT, default_params  # noqa: E402
from tennisbot_rl_amd.rollout import RolloutBuffer  # noqa: E402
from tennisbot_rl_amd.stepper import BatchedEnv  # noqa: E402

dev = torch.device("cuda", 0)


def probe(kind, opts):
    env = BatchedEnv(kind, n, device=dev, seed=0, params=default_params(flags=F_DEFAULT), track_terminal_obs=False, pipeline=kind == ENV_SWING, options=opts)
    T = 26
    buf = RolloutBuffer(kind, T, n, dev)
    torch.manual_seed(0)
    buf.actions.uniform_(-1.0, 1.0)
    buf.bind(env)
    env.reset()
    for t in range(T):  # one whole episode first (SwingRacket: its 26th step parks, the join finishes it)
        buf.step_into(env, t)
    env.flush()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for t in range(25):
            buf.step_into(env, t)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 25 * 1e3)
        buf.step_into(env, 25)
        env.flush()
        torch.cuda.synchronize()
    byts = (267 if kind == ENV_SWING else 263) * n
    print(json.dumps({"env": "swing" if kind == ENV_SWING else "tennis", "n": n, "opts": opts, "us_per_step": round(best, 2), "TB_per_s": round(byts / best / 1e6, 2)}), flush=True)
    env.close()
    del buf
    torch.cuda.empty_cache()


for o in ({}, dict(block=64), dict(block=256), dict(swing_reg_rows=False), dict(swing_reg_rows=False, block=256), {}):
    probe(ENV_SWING, o)
for o in ({}, dict(block=256)):
    probe(ENV_TENNIS, o)

#!/usr/bin/env python3
"""Diagnostic: which configurations capture a marked rollout graph (RolloutBuffer.capture_marked)?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tennisbot_rl_amd.params import ENV_SWING, ENV_TENNIS
from tennisbot_rl_amd.rollout import RolloutBuffer
from tennisbot_rl_amd.stepper import BatchedEnv


def attempt(kind, piped, T, chunks, warm):
    env = BatchedEnv(kind, 4096, device="cuda:0", seed=8, track_terminal_obs=False, pipeline=piped)
    buf = RolloutBuffer(kind, T, 4096, "cuda:0").bind(env)
    buf.actions.uniform_(-1, 1)
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        env.reset()
        for t in range(warm):
            buf.step_into(env, t)
        try:
            g = buf.capture_marked(env, chunks)
            g.replay()
            s.synchronize()
            return "ok"
        except Exception as exc:  # noqa: BLE001
            return "FAILED: %s" % str(exc).splitlines()[0]


for args in [(ENV_TENNIS, False, 104, 4, 0), (ENV_SWING, False, 104, 4, 0), (ENV_SWING, True, 20, 2, 0), (ENV_SWING, True, 26, 1, 0),
             (ENV_SWING, True, 52, 2, 0), (ENV_SWING, True, 104, 4, 5)]:
    print(args, attempt(*args), flush=True)
os._exit(0)

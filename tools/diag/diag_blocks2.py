#!/usr/bin/env python3
"""Workgroup size of the step kernels, 64 vs 128 vs 256 threads, alternated in one process: both envs, 4096 / 32768 / 131072 envs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tennisbot_rl_amd.params import ENV_SWING, ENV_TENNIS
from tennisbot_rl_amd.rollout import RolloutBuffer
from tennisbot_rl_amd.stepper import BatchedEnv
dev = torch.device("cuda", 0)
def run(kind, n, block):
    T = 1040 if n <= 32768 else 104
    env = BatchedEnv(kind, n, device=dev, seed=0, track_terminal_obs=False, pipeline=kind == ENV_SWING, options=dict(block=block))
    buf = RolloutBuffer(kind, T, n, dev); buf.actions.uniform_(-1, 1); buf.bind(env); env.reset()
    for t in range(26 if kind == ENV_SWING else 1040): buf.step_into(env, t % T)
    env.flush()
    g = env.capture(lambda: buf.step_range(env, 0, T))
    torch.cuda.synchronize()
    out = []
    for k in range(12):
        t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); out.append(time.perf_counter() - t0)
    out.sort(); env.close()
    return n * T / out[6] / 1e6
sizes = [int(x) for x in sys.argv[1:]] or [4096, 32768, 131072]
for kind, name in ((ENV_SWING, "swing"), (ENV_TENNIS, "tennis")):
    for n in sizes:
        r = {b: [] for b in (64, 128, 256)}
        for rep in range(3):
            for b in (64, 128, 256):
                r[b].append(run(kind, n, b))
        print("%-6s %7d envs: " % (name, n) + "   ".join("block %3d: %s" % (b, " ".join("%.0f" % x for x in r[b])) for b in r), flush=True)

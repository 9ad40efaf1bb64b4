#!/usr/bin/env python3
"""kernel-trace helper: one SwingRacket env batch, a few eager episodes with a given TbOptions set (run under rocprofv3)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from tennisbot_rl_amd.params import ENV_SWING, F_DEFAULT, F_RACKET_GROUND, default_params
from tennisbot_rl_amd.stepper import BatchedEnv
n = int(sys.argv[1]); opts = json.loads(sys.argv[2]); rg = "rg" in sys.argv[3:]
extra = [x for x in sys.argv[3:] if x.startswith("-D")]
if extra:  # a variant of the library, built into /tmp
    import subprocess
    from tennisbot_rl_amd import stepper
    from tennisbot_rl_amd.build import HIPCC_FLAGS, SOURCES, hipcc
    subprocess.check_call([hipcc()] + HIPCC_FLAGS + extra + ["-o", "/tmp/libtb_variant.so"] + SOURCES)
    stepper.use_library("/tmp/libtb_variant.so")
env = BatchedEnv(ENV_SWING, n, device="cuda:0", seed=0, pipeline=True, track_terminal_obs=False, options=opts,
                 params=default_params(flags=F_DEFAULT | (F_RACKET_GROUND if rg else 0)))
env.reset()
if "graph" in sys.argv[3:]:  # what bench.py times: one hipGraph of 104 steps, replayed
    from tennisbot_rl_amd.rollout import RolloutBuffer
    buf = RolloutBuffer(ENV_SWING, 104, n, torch.device("cuda", 0))
    buf.actions.uniform_(-1.0, 1.0)
    buf.bind(env)
    for t in range(26):
        buf.step_into(env, t)
    env.flush()
    gr = env.capture(lambda: buf.step_range(env, 0, 104))
    for _ in range(3):
        gr.replay(); torch.cuda.synchronize()
    print(env.counters())
    sys.exit(0)
g = torch.Generator(device="cuda:0"); g.manual_seed(1)
for ep in range(6):
    for t in range(26):
        env.step(torch.rand((n, 6), device="cuda:0", generator=g) * 2 - 1)
    env.flush(); torch.cuda.synchronize()
print(env.counters())

#!/usr/bin/env python3
"""What makes the slowest workgroup of a small-batch Tennisbot step launch 1.5 us slower than the first (r03_cadence_probe.py)?
Timing-only ablation builds (RESULTS ARE WRONG in them): the narrowphase switched off (-DTB_DIAG_NO_NARROW: no contacts, no sweep,
no solve), and the product build for reference; each as a replayed 1040-step graph at 4096 envs. Run on the GPU box."""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from tennisbot_rl_amd import stepper
from tennisbot_rl_amd.build import HIPCC_FLAGS, SOURCES, hipcc

variant = sys.argv[1] if len(sys.argv) > 1 else ""
if variant:
    lib = "/tmp/libtb_variant.so"
    subprocess.check_call([hipcc()] + HIPCC_FLAGS + variant.split() + ["-o", lib] + SOURCES)
    stepper.use_library(lib)
from tennisbot_rl_amd.params import ENV_TENNIS, default_params
from tennisbot_rl_amd.rollout import RolloutBuffer
from tennisbot_rl_amd.stepper import BatchedEnv
dev = torch.device("cuda", 0)
T, n = 1040, 4096
env = BatchedEnv(ENV_TENNIS, n, device=dev, seed=0, params=default_params(), track_terminal_obs=False)
buf = RolloutBuffer(ENV_TENNIS, T, n, dev); torch.manual_seed(0); buf.actions.uniform_(-1, 1); buf.bind(env); env.reset()
for t in range(T): buf.step_into(env, t)
g = env.capture(lambda: buf.step_range(env, 0, T))
for _ in range(100): g.replay()
torch.cuda.synchronize()
ts = []
for k in range(15):
    torch.cuda.synchronize(); t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
ts.sort()
c = env.counters()
print(json.dumps({"variant": variant or "product", "rate_M": n * T / ts[len(ts) // 2] / 1e6, "us_per_step": ts[len(ts) // 2] / T * 1e6,
                  "episodes_finished": c["episodes_finished"], "racket_ball_contact_substeps": c["racket_ball_contact_substeps"]}), flush=True)

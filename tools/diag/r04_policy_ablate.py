#!/usr/bin/env python3
"""What a step of the policy rollout kernel is made of (GPU; timing-only ablation builds, RESULTS ARE WRONG in them): the
rollout kernels of 1040 steps x 4096 envs, untrained policy (no contacts: the plain chain towers -> sample -> step -> observations),
as built, with deterministic actions (no noise draw), without the MLP towers (-DTB_DIAG_NO_TOWERS), without the env step
(-DTB_DIAG_NO_ENVSTEP), without either.   python3 tools/diag/r04_policy_ablate.py            (runs itself once per build)"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
VARIANTS = {"product": [], "no_towers": ["-DTB_DIAG_NO_TOWERS"], "no_envstep": ["-DTB_DIAG_NO_ENVSTEP"], "neither": ["-DTB_DIAG_NO_TOWERS", "-DTB_DIAG_NO_ENVSTEP"]}
if len(sys.argv) < 2:
    for v in VARIANTS:
        subprocess.check_call([sys.executable, os.path.abspath(__file__), v])
    sys.exit(0)
variant = sys.argv[1]
from tennisbot_rl_amd import stepper
from tennisbot_rl_amd.build import HIPCC_FLAGS, SOURCES, hipcc
if VARIANTS[variant]:
    out = "/tmp/libtb_%s.so" % variant
    subprocess.check_call([hipcc()] + HIPCC_FLAGS + VARIANTS[variant] + ["-o", out] + SOURCES)
    stepper.use_library(out)
import numpy as np, torch
from tennisbot_rl_amd.stepper import BatchedEnv
from tennisbot_rl_amd.params import ENV_SWING, OBS_DIM, ACT_DIM
from tennisbot_rl_amd.ppo import SWING_DEFAULTS, build_actor_critic, pack_policy
policy = build_actor_critic(OBS_DIM[ENV_SWING], ACT_DIM[ENV_SWING], tuple(SWING_DEFAULTS["net_arch"])).to("cuda:0")
blob = pack_policy(policy)
for det in (False, True):
    env = BatchedEnv(ENV_SWING, 4096, device="cuda:0", seed=8, pipeline=True, track_terminal_obs=False, options=dict(ff_defer="all"))
    o = env.reset()
    T = 1040
    for _ in range(2):
        (obs, rew, done), _p = env.policy_rollout(blob, o, T, seed=5, deterministic=det); env.flush(); o = obs[-1].contiguous()
    torch.cuda.synchronize()
    ts = []
    for rep in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        (obs, rew, done), _p = env.policy_rollout(blob, o, T, seed=5 + rep, deterministic=det)
        torch.cuda.current_stream().synchronize(); t1 = time.perf_counter()
        env.flush(); torch.cuda.synchronize()
        o = obs[-1].contiguous(); ts.append(t1 - t0)
    a = np.median(ts) * 1e3
    print("%-11s %-13s rollout kernels %.3f ms = %.2f us per step" % (variant, "deterministic" if det else "stochastic", a, a * 1e3 / T), flush=True)
    env.close()

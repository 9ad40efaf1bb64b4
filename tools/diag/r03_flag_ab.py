#!/usr/bin/env python3
"""Same-box A/B of compiler flags: two single-translation-unit builds of the library into /tmp (HIPCC_FLAGS + <A> / + <B>), each
workload in a process of its own per build. usage: r03_flag_ab.py "<flags A>" "<flags B>" [workload ...]   (run on the GPU box)
e.g. r03_flag_ab.py "" "-fno-slp-vectorize" swing4096 swing1m   -- how the two kernel builds of the product were chosen"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
WORK = {"swing4096": (0, 4096, 1040, 0), "swing32k": (0, 32768, 1040, 0), "swing128k": (0, 131072, 104, 0), "swing1m": (0, 1048576, 104, 0),
        "tennis4096": (1, 4096, 1040, 0), "tennis1m": (1, 1048576, 104, 0), "rg4096": (0, 4096, 1040, 1),
        "swing4m": (0, 4194304, 52, 0), "tennis4m": (1, 4194304, 52, 0), "tennis256k": (1, 262144, 104, 0), "swing256k": (0, 262144, 104, 0)}
if len(sys.argv) > 2 and sys.argv[1] == "--child":
    lib, what = sys.argv[2], sys.argv[3]
    import torch
    from tennisbot_rl_amd import stepper
    stepper.use_library(lib)
    from tennisbot_rl_amd.params import ENV_SWING, ENV_TENNIS, F_DEFAULT, F_RACKET_GROUND, default_params
    from tennisbot_rl_amd.rollout import RolloutBuffer
    from tennisbot_rl_amd.stepper import BatchedEnv
    dev = torch.device("cuda", 0)
    if what.startswith("collect_"):  # a PPO collect of 1100 x 4096 (policy inside the rollout kernel): untrained, or the reference's trained policy
        import numpy as np
        from tennisbot_rl_amd.ppo import PPOTrainer
        tr = PPOTrainer("SwingRacket-v0", num_envs=4096, n_steps=1100, seed=0, options=dict(policy_slices=3) if what.endswith("_s3") else None)
        if what.startswith("collect_ref"):
            tr.policy.load_sb3_arrays(dict(np.load(os.path.join(ROOT, "tests", "golden", "ppo_swing_policy.npz"))))
        for _ in range(20): tr.collect()
        ts = []
        for _ in range(15):
            torch.cuda.synchronize(); t0 = time.perf_counter(); tr.collect(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        ts.sort()
        print(json.dumps({"rate_M": round(4096 * 1100 / ts[len(ts) // 2] / 1e6, 1)})); sys.exit(0)
    opts = {}
    if what.endswith("_ldsrows"):  # (the step kernel with its static rows in LDS: 103 instead of 154 VGPRs)
        what, opts = what[:-len("_ldsrows")], dict(swing_reg_rows=False)
    k, n, T, rg = WORK[what]
    kind = ENV_TENNIS if k else ENV_SWING
    env = BatchedEnv(kind, n, device=dev, seed=0, params=default_params(flags=F_DEFAULT | (F_RACKET_GROUND if rg else 0)), track_terminal_obs=False, pipeline=kind == ENV_SWING, options=opts)
    buf = RolloutBuffer(kind, T, n, dev); torch.manual_seed(0); buf.actions.uniform_(-1, 1); buf.bind(env); env.reset()
    for t in range(T): buf.step_into(env, t)
    env.flush()
    g = env.capture(lambda: buf.step_range(env, 0, T))
    t_end = time.perf_counter() + (1.0 if n <= 32768 else 0.2)
    while time.perf_counter() < t_end: g.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(9 if n <= 32768 else 5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts.sort()
    print(json.dumps({"rate_M": round(n * T / ts[len(ts) // 2] / 1e6, 1)})); sys.exit(0)
from tennisbot_rl_amd.build import HIPCC_FLAGS, SOURCES, hipcc
A, B = sys.argv[1].split(), sys.argv[2].split()
libs = {"A: " + sys.argv[1]: "/tmp/libtb_ab_a.so", "B: " + sys.argv[2]: "/tmp/libtb_ab_b.so"}


def flags(extra):  # "~flag" takes `flag` out of the product's HIPCC_FLAGS (e.g. "~-fno-slp-vectorize ~-mllvm ~-amdgpu-sched-strategy=max-ilp": the build of rounds 1-2)
    drop = {x[1:] for x in extra if x.startswith("~")}
    return [f for f in HIPCC_FLAGS if f not in drop] + [x for x in extra if not x.startswith("~")]


procs = [subprocess.Popen([hipcc()] + flags(f) + ["-o", lib] + SOURCES) for f, lib in zip((A, B), libs.values())]
assert all(p.wait() == 0 for p in procs)
for what in sys.argv[3:] or list(WORK):
    row = {}
    for name, lib in libs.items():
        r = subprocess.run([sys.executable, __file__, "--child", lib, what], capture_output=True, text=True)
        try:
            row[name] = json.loads(r.stdout.strip().splitlines()[-1])["rate_M"]
        except Exception:
            row[name] = "failed: " + r.stderr[-300:]
    print(what, json.dumps(row), flush=True)

#!/usr/bin/env python3
"""Does an HBM-bound step kernel keep its rate when it only gets one or two wave slots per SIMD (the room a co-resident fast-forward
leaves)? Diagnostic build (-DTB_DIAG_LDS_PAD): tb_diag_set_lds_pad(bytes) pads every step launch's dynamic LDS so that fewer workgroups fit a CU. Tennisbot, 1 M envs,
64-thread workgroups, replayed 104-step graphs. usage: r03_occupancy_probe.py  (run on the GPU box; spawns itself per setting)"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) == 1:
    for pad, label in ((0, "unrestricted"), (19000, "8 waves per CU = 2 per SIMD"), (39000, "4 waves per CU = 1 per SIMD"), (0, "unrestricted")):
        env = dict(os.environ, TB_DIAG_LDS_PAD=str(pad))
        r = subprocess.run([sys.executable, __file__, label], env=env, capture_output=True, text=True)
        print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-500:], flush=True)
    sys.exit(0)
import torch
from tennisbot_rl_amd import stepper
from tennisbot_rl_amd.build import HIPCC_FLAGS, SOURCES, hipcc
lib = "/tmp/libtb_ldspad.so"
if not os.path.exists(lib):
    subprocess.check_call([hipcc()] + HIPCC_FLAGS + ["-DTB_DIAG_LDS_PAD", "-o", lib] + SOURCES)
stepper.use_library(lib)
stepper.load_library().tb_diag_set_lds_pad(int(os.environ.get("TB_DIAG_LDS_PAD", "0")))
from tennisbot_rl_amd.params import ENV_TENNIS, default_params
from tennisbot_rl_amd.rollout import RolloutBuffer
from tennisbot_rl_amd.stepper import BatchedEnv
dev = torch.device("cuda", 0)
T, n = 104, 1048576
env = BatchedEnv(ENV_TENNIS, n, device=dev, seed=0, params=default_params(), track_terminal_obs=False, options=dict(block=64))
buf = RolloutBuffer(ENV_TENNIS, T, n, dev); buf.actions.uniform_(-1, 1); buf.bind(env); env.reset()
for t in range(T): buf.step_into(env, t)
g = env.capture(lambda: buf.step_range(env, 0, T))
for _ in range(3): g.replay()
torch.cuda.synchronize()
ts = []
for k in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
ts.sort()
print(json.dumps({"setting": sys.argv[1], "lds_pad": os.environ.get("TB_DIAG_LDS_PAD"), "rate_G": n * T / ts[2] / 1e9, "us_per_step": ts[2] / T * 1e6, "TB_per_s": 263 * n / (ts[2] / T) / 1e12}))

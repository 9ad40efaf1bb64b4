#!/usr/bin/env python3
"""What does a wave pay when one of its lanes touches something? Tennisbot, 4096 envs, steady state, the -DTB_DIAG_STAMPS build: cycles
between the stamps around the static narrowphase and the contact solve, summed over all waves, against the number of wave-substeps
that entered the solver. Run on the GPU box."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from tennisbot_rl_amd import stepper
from tennisbot_rl_amd.build import HIPCC_FLAGS, SOURCES, hipcc
out = "/tmp/libtb_stamps.so"
subprocess.check_call([hipcc()] + HIPCC_FLAGS + ["-DTB_DIAG_STAMPS", "-o", out] + SOURCES)
stepper.use_library(out)
L = stepper.load_library()
L.tb_diag_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
from tennisbot_rl_amd.params import ENV_TENNIS
n = 4096
rng = np.random.Generator(np.random.PCG64(0))
acts = torch.from_numpy(rng.uniform(-1, 1, (104, n, 2)).astype(np.float32)).cuda()
env = stepper.BatchedEnv(ENV_TENNIS, n, seed=0, reuse_buffers=True)
env.reset()
for t in range(1040): env.step(acts[t % 104])
buf = (ctypes.c_ulonglong * 16)()
L.tb_diag_read_stamps(buf, 1)
T = 2080
for t in range(T): env.step(acts[t % 104])
L.tb_diag_read_stamps(buf, 1)
v = list(buf)
names = ["between", "racketNP", "staticNP", "velocity", "solve", "pose"]
waves = v[13]
print("wave-substeps %d (%d launches x %d waves), of which %d entered the solver (%.2f per launch), %d had a lane in the racket's reach" % (waves, T, n // 64, v[6], v[6] / T, v[12]))
for k, nm in enumerate(names):
    print("  %-9s %12d cycles = %7.1f per wave-substep" % (nm, v[k], v[k] / max(waves, 1)))
quiet = {"staticNP": 30.0, "solve": 30.0}  # (a wave that skips both pays about this much for the votes and the stamps themselves)
print("  per wave-substep that entered the solver: solve bucket %.0f cycles, static narrowphase bucket at most %.0f (clock 2.35 GHz: %.2f us + %.2f us)" % (
    (v[4] - quiet["solve"] * (waves - v[6])) / max(v[6], 1), (v[2] - quiet["staticNP"] * (waves - v[6])) / max(v[6], 1),
    (v[4] - quiet["solve"] * (waves - v[6])) / max(v[6], 1) / 2350.0, (v[2] - quiet["staticNP"] * (waves - v[6])) / max(v[6], 1) / 2350.0))
print(env.counters())

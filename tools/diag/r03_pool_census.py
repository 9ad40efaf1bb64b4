#!/usr/bin/env python3
"""Lane census of the pool run at the end of a PPO collect (diagnostic build, -DTB_DIAG_LANES; run on the GPU box): how full are the
waves of the ONE fast-forward launch that finishes a rollout's episode ends, with random actions and under the reference's trained
policy, whose struck balls fly 300-775 substeps?"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from tennisbot_rl_amd import stepper  # noqa: E402
from tennisbot_rl_amd.build import HIPCC_FLAGS, SOURCES, hipcc  # noqa: E402

out = "/tmp/libtb_lanes.so"
subprocess.check_call([hipcc()] + HIPCC_FLAGS + ["-DTB_DIAG_LANES", "-o", out] + SOURCES)
stepper.use_library(out)
L = stepper.load_library()
L.tb_diag_read_lanes.argtypes = [ctypes.c_void_p, ctypes.c_int]
from tennisbot_rl_amd.ppo import PPOTrainer  # noqa: E402

buf = (ctypes.c_ulonglong * 16)()
for label in ("reference_policy", "untrained"):
    tr = PPOTrainer("SwingRacket-v0", num_envs=4096, n_steps=1092, seed=0)
    if label.startswith("reference_policy"):
        tr.policy.load_sb3_arrays(dict(np.load(os.path.join(ROOT, "tests", "golden", "ppo_swing_policy.npz"))))
    for _ in range(2):
        tr.collect()
    torch.cuda.synchronize()
    L.tb_diag_read_lanes(buf, 1)
    tr.collect()
    torch.cuda.synchronize()
    L.tb_diag_read_lanes(buf, 1)
    v = list(buf)
    # the rollout kernels run one substep per step (64 lanes each, counted too): 1092 steps x 64 waves
    steps_ws = 1092 * 64
    ff_ws, ff_lanes = v[1] - steps_ws, v[0] - steps_ws * 64
    print("%-24s fast-forward: %d wave-substeps, %.1f active lanes each (%d lane-substeps); with a lane in the racket's sphere %.1f %%, with a contact %.1f %%"
          % (label, ff_ws, ff_lanes / max(ff_ws, 1), ff_lanes, 100.0 * v[3] / max(v[1], 1), 100.0 * v[9] / max(v[1], 1)), flush=True)
    del tr

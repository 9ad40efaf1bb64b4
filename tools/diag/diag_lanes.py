#!/usr/bin/env python3
"""Lane census of the fast-forward's wave votes (diagnostic build, -DTB_DIAG_LANES; run on the GPU box): per wave-substep, how
many lanes are active, how many ask for each branch a whole wave then runs. usage: diag_lanes.py [n_envs]"""
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
from tennisbot_rl_amd.params import ENV_SWING  # noqa: E402

out = "/tmp/libtb_lanes.so"
subprocess.check_call([hipcc()] + HIPCC_FLAGS + ["-DTB_DIAG_LANES", "-o", out] + SOURCES)
stepper.use_library(out)
L = stepper.load_library()
L.tb_diag_read_lanes.argtypes = [ctypes.c_void_p, ctypes.c_int]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rng = np.random.Generator(np.random.PCG64(0))
acts = torch.from_numpy(rng.uniform(-1, 1, (26, n, 6)).astype(np.float32)).cuda()
env = stepper.BatchedEnv(ENV_SWING, n, seed=0, pipeline=True, track_terminal_obs=False)
env.reset()
buf = (ctypes.c_ulonglong * 16)()
names = ["active", "inside the racket's bounding sphere", "needing the outline sweep", "near a static shape", "with a contact", "with a racket contact"]
for ep in range(2):
    for t in range(26):
        if t == 25:
            L.tb_diag_read_lanes(buf, 1)
            short = list(buf)
        env.step(acts[t])
    env.flush()
    L.tb_diag_read_lanes(buf, 1)
    ff = list(buf)
    for label, v in (("25 short steps", short), ("26th step + fast-forward", ff)):
        ws = max(v[1], 1)
        print("episode %d, %s: %d wave-substeps, %.1f active lanes each" % (ep, label, v[1], v[0] / ws))
        for k in range(1, 6):
            lanes, waves = v[2 * k], v[2 * k + 1]
            print("    lanes %-38s %10d = %5.2f %% of active lanes; wave-substeps with one %9d = %5.1f %%; %.2f lanes per such wave-substep"
                  % (names[k], lanes, 100.0 * lanes / max(v[0], 1), waves, 100.0 * waves / ws, lanes / max(waves, 1)))
        print("    lanes past the slab test (into the 12 cull planes) %d = %.2f %% of active lanes" % (v[12], 100.0 * v[12] / max(v[0], 1)))

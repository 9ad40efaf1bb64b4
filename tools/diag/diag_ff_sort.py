#!/usr/bin/env python3
"""Does the sort of the parked records do what it is for? Diagnostic build (-DTB_DIAG_STAMPS counts wave-substeps):
wave-substeps, wave-substeps with a lane in the racket's bounding sphere / in the outline sweep, with and without
tb_ff_sort_kernel, for one batch of SwingRacket episodes; and the ideal (sum over lanes / 64)."""
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

out = "/tmp/libtb_stamps.so"
subprocess.check_call([hipcc()] + HIPCC_FLAGS + ["-DTB_DIAG_STAMPS", "-o", out] + SOURCES)
stepper.use_library(out)
L = stepper.load_library()
L.tb_diag_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
buf = (ctypes.c_ulonglong * 16)()
for sort in (False, True):
    env = stepper.BatchedEnv(ENV_SWING, n, seed=0, pipeline=True, track_terminal_obs=False, options=dict(ff_lanes_per_wave=64, ff_sort=sort))
    env.reset()
    g = torch.Generator(device="cuda:0"); g.manual_seed(1)
    for ep in range(2):
        for t in range(26):
            if t == 25:
                env.flush(); L.tb_diag_read_stamps(buf, 1)
            env.step(torch.rand((n, 6), device="cuda:0", generator=g) * 2 - 1)
        env.flush()
        L.tb_diag_read_stamps(buf, 1)
        v = list(buf)
        c = env.counters()
        print("sort=%s episode %d: wave-substeps %d (ideal %d = substeps / 64), in reach %d, sweeping %d (lane-sweeps %d); stamped cycles by segment %s"
              % (sort, ep, v[13], c["substeps"] // 64, v[12], v[11], v[10], [int(x / max(v[13], 1)) for x in v[:6]]))
        env.counters_reset()
    env.close()

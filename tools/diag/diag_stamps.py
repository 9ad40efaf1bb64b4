#!/usr/bin/env python3
"""Where a fast-forward substep spends its cycles (diagnostic build with in-kernel s_memtime
stamps, -DTB_DIAG_STAMPS; run on the GPU box). Reads SHARES, not absolute time: the stamps
themselves cost ~40 cycles each and fence the scheduler."""
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
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rng = np.random.Generator(np.random.PCG64(0))
acts = torch.from_numpy(rng.uniform(-1, 1, (104, n, 6)).astype(np.float32)).cuda()
from tennisbot_rl_amd.params import F_DEFAULT, F_RACKET_GROUND, default_params  # noqa: E402
flags = F_DEFAULT | (F_RACKET_GROUND if os.environ.get("TB_DIAG_RACKET_GROUND") == "1" else 0)
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 0   # > 0: pipelined, tb_ff_kernel with that many parked envs per wave
if lanes:
    env = stepper.BatchedEnv(ENV_SWING, n, seed=0, params=default_params(flags=flags), pipeline=True, track_terminal_obs=False, options=dict(ff_lanes_per_wave=lanes, ff_sort=False))
else:
    env = stepper.BatchedEnv(ENV_SWING, n, seed=0, reuse_buffers=True, params=default_params(flags=flags))
env.reset()
names = ["between substeps", "racket narrowphase", "static narrowphase", "velocity update", "contact solve", "pose update"]
buf = (ctypes.c_ulonglong * 16)()
for ep in range(2):
    for t in range(26):
        if t == 25:
            L.tb_diag_read_stamps(buf, 1)
            short = list(buf)
        env.step(acts[(ep * 26 + t) % 104])
    env.flush() if lanes else None
    L.tb_diag_read_stamps(buf, 1)
    ff = list(buf)
    for label, v, launches in (("25 short steps", short, 25), ("fast-forward step", ff, 1)):
        waves = v[9] / launches
        tot = sum(v[:6])
        print("episode %d, %s: %.0f waves/launch, kernel span %.0f cycles/wave/launch; stamped %.0f cycles/wave/launch" % (ep, label, waves, v[8] / v[9], tot / v[9]))
        for k in range(6):
            print("    %-20s %6.1f %%  (%.0f cycles/wave/launch)" % (names[k], 100.0 * v[k] / tot, v[k] / v[9]))
        print("    entry -> state/outline loads landed: %.0f cycles/wave/launch; loads landed -> step computed, outputs issued: %.0f"
              % (v[7] / v[9], v[15] / v[9]))
        if v[14]:
            print("    shader clock while this kernel ran: %.2f GHz (s_memtime / s_memrealtime)" % (v[8] / v[14] * 0.1))
        print("    wave-substeps %d, with a lane inside the racket's bounding sphere %d, with a lane running the outline sweep %d (lane-sweeps %d)"
              % (v[13], v[12], v[11], v[10]))
if lanes:
    print("(pipelined: 'fast-forward step' = the 26th step kernel + tb_ff_kernel with %d envs per wave; per wave-substep: %s)" % (
        lanes, ", ".join("%s %.0f" % (names[k], ff[k] / max(ff[13], 1)) for k in range(6))))
else:
    print("max substeps of last fast-forward:", int(env.last_substeps().max()))

#!/usr/bin/env python3
"""Per-launch timing of real SwingRacket episodes on the GPU box (diagnostic):
event-timed duration of every step launch, with the max / mean substeps of the batch."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from tennisbot_rl_amd.params import ENV_SWING  # noqa: E402
from tennisbot_rl_amd.stepper import BatchedEnv  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
mode = sys.argv[2] if len(sys.argv) > 2 else "full"
from tennisbot_rl_amd.params import F_DEFAULT, F_NET, default_params  # noqa: E402
from tennisbot_rl_amd import stepper  # noqa: E402
if mode.startswith("lib="):
    stepper._LIB_PATH = mode[4:]
params = default_params(flags=F_NET if mode == "contact_off" else F_DEFAULT)
rng = np.random.Generator(np.random.PCG64(0))
acts = torch.from_numpy(rng.uniform(-1, 1, (104, n, 6)).astype(np.float32)).cuda()
env = BatchedEnv(ENV_SWING, n, seed=0, reuse_buffers=True, params=params)
env.reset()
for ep in range(4):
    short = []
    for t in range(26):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); env.step(acts[(ep * 26 + t) % 104]); e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3
        if t < 25:
            short.append(us)
        else:
            s = env.last_substeps()
            print("episode %d: short steps mean %.1f us (min %.1f max %.1f); fast-forward %.1f us, substeps max %d mean %.1f -> %.3f us per substep of the longest lane"
                  % (ep, np.mean(short), np.min(short), np.max(short), us, int(s.max()), float(s.float().mean()), us / int(s.max())))
print(env.counters())

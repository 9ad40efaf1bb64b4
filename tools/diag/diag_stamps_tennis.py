#!/usr/bin/env python3
"""Per-launch cycle shares of the Tennisbot step kernel (diagnostic -DTB_DIAG_STAMPS build), to see
what separates its fast launches from its slow ones. Run on the GPU box."""
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
from tennisbot_rl_amd.params import ENV_TENNIS  # noqa: E402

out = "/tmp/libtb_stamps.so"
subprocess.check_call([hipcc()] + HIPCC_FLAGS + ["-DTB_DIAG_STAMPS", "-o", out] + SOURCES)
stepper._LIB_PATH = out
L = stepper.load_library()
L.tb_diag_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
n = 4096
rng = np.random.Generator(np.random.PCG64(0))
acts = torch.from_numpy(rng.uniform(-1, 1, (104, n, 2)).astype(np.float32)).cuda()
env = stepper.BatchedEnv(ENV_TENNIS, n, seed=0, reuse_buffers=True)
env.reset()
names = ["between", "racketNP", "staticNP", "velocity", "solve", "pose"]
buf = (ctypes.c_ulonglong * 16)()
L.tb_diag_read_stamps(buf, 1)
print("step  span   " + "  ".join("%8s" % x for x in names) + "   in-reach sweeps(wave/lane)")
rows = []
for t in range(1100):
    env.step(acts[t % 104])
    L.tb_diag_read_stamps(buf, 1)
    v = list(buf)
    w = max(v[9], 1)
    rows.append([t, v[8] / w] + [v[k] / w for k in range(6)] + [v[12], v[11], v[10]])
rows = np.array(rows)
span = rows[:, 1]
print("span cycles/wave: p10 %.0f p50 %.0f p90 %.0f max %.0f" % tuple(np.percentile(span, [10, 50, 90, 100])))
slow = rows[span > np.percentile(span, 50) * 1.25]
print("%d of %d launches are >25%% above the median; their mean shares vs the others':" % (len(slow), len(rows)))
fast = rows[span <= np.percentile(span, 50) * 1.25]
for k, nm in enumerate(names):
    print("  %-10s slow %7.0f   fast %7.0f" % (nm, slow[:, 2 + k].mean() if len(slow) else 0, fast[:, 2 + k].mean()))
print("  wave-substeps with a lane in reach: slow %.1f fast %.1f; wave-sweeps slow %.1f fast %.1f" % (
    slow[:, 8].mean() if len(slow) else 0, fast[:, 8].mean(), slow[:, 9].mean() if len(slow) else 0, fast[:, 9].mean()))
print("slow launches at steps:", [int(x) for x in slow[:40, 0]])
c = env.counters()
print(c)

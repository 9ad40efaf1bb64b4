#!/usr/bin/env python3
"""Timing-only ablation of the fast-forward substep (run on the GPU box).

Builds variants of libtb_stepper.so with TB_DIAG_* macros into /tmp, puts every env in a
state that runs the full 776-substep fast-forward without contacts (ball far off the court)
and reports microseconds per substep of a full wave. Variant results are WRONG by
construction; only the times matter. Usage: python tools/diag/diag_substep.py [n_envs]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from helpers import make_words  # noqa: E402
from tennisbot_rl_amd import stepper  # noqa: E402
from tennisbot_rl_amd.build import HIPCC_FLAGS, SOURCES, hipcc  # noqa: E402
from tennisbot_rl_amd.params import ENV_SWING  # noqa: E402

VARIANTS = {
    "baseline": [],
    "no_angular": ["-DTB_DIAG_NO_ANGULAR"],
    "no_orient": ["-DTB_DIAG_NO_ORIENT"],
    "no_narrow": ["-DTB_DIAG_NO_NARROW"],
    "linear_only": ["-DTB_DIAG_NO_ANGULAR", "-DTB_DIAG_NO_ORIENT", "-DTB_DIAG_NO_NARROW"],
}
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rng = np.random.default_rng(0)
w, d = make_words(ENV_SWING, n, racket_pos=(8, 0, 1.0), racket_angvel=rng.uniform(-3, 3, (n, 3)), racket_vel=rng.uniform(-1, 1, (n, 3)),
                  ball_pos=(-20.0, 0, 1e4), goal=(-6, 0), spawn_pos=(8, 0, 0.5), init_dist=10.0, step_count=25)
for name, flags in VARIANTS.items():
    out = "/tmp/libtb_%s.so" % name
    subprocess.check_call([hipcc()] + HIPCC_FLAGS + flags + ["-o", out] + SOURCES)
    stepper._LIB = None
    stepper.use_library(out)
    env = stepper.BatchedEnv(ENV_SWING, n, auto_reset=False)
    a = torch.zeros((n, 6), device="cuda")
    times = []
    for rep in range(5):
        env.set_state_words(w.view(np.int32), d)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); env.step(a); e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) * 1e3)
    sub = int(env.last_substeps()[0])
    print("%-12s substeps %d  kernel %.1f us  per substep %.3f us" % (name, sub, min(times), min(times) / sub), flush=True)
    env.close()

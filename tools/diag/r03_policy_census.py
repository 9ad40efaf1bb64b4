#!/usr/bin/env python3
"""Lane census of the policy rollout kernels of ONE PPO collect (diagnostic build, -DTB_DIAG_LANES; run on the GPU box), read before
the pool's fast-forward launch runs: in how many of an env wave's substeps does some lane reach the racket's bounding sphere, need
the 38-edge outline sweep, or enter the contact solver -- with the untrained policy and under the reference's trained policy, whose
racket goes for the ball?"""
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
from tennisbot_rl_amd.ppo import PPOTrainer, pack_policy  # noqa: E402

buf = (ctypes.c_ulonglong * 16)()
for label in ("reference_policy", "untrained"):
    tr = PPOTrainer("SwingRacket-v0", num_envs=4096, n_steps=1092, seed=0)
    if label.startswith("reference_policy"):
        tr.policy.load_sb3_arrays(dict(np.load(os.path.join(ROOT, "tests", "golden", "ppo_swing_policy.npz"))))
    for _ in range(2):
        tr.collect()
    torch.cuda.synchronize()
    L.tb_diag_read_lanes(buf, 1)
    b, env = tr.buf, tr.env
    pack_policy(tr.policy, out=tr.packed)
    rec = b.record
    tr.obs_seq[0].copy_(tr.obs_in)
    env.policy_rollout_ptrs(tr.n_steps, tr.packed.data_ptr(), tr.obs_in.data_ptr(), b.actions[0].data_ptr(), tr._raw_actions.data_ptr(), tr.logps.data_ptr(),
                            tr.values.data_ptr(), b.obs[0].data_ptr(), b.rewards[0].data_ptr(), b.dones[0].data_ptr(), (rec, 0, 0, 0, rec, rec, rec), tr.noise_seed)
    torch.cuda.current_stream().synchronize()  # the rollout kernels are done; the pool's launch waits for flush()
    L.tb_diag_read_lanes(buf, 1)
    v = list(buf)
    env.flush()
    torch.cuda.synchronize()
    ws = max(v[1], 1)
    print("%-18s policy rollout kernels: %d wave-substeps, %.1f active lanes each; some lane in the racket's sphere %.1f %%, needs the outline sweep %.1f %% "
          "(%.2f lanes where one does), in the contact solver %.1f %% (racket contact %.1f %%)"
          % (label, v[1], v[0] / ws, 100.0 * v[3] / ws, 100.0 * v[5] / ws, v[4] / max(v[5], 1), 100.0 * v[9] / ws, 100.0 * v[11] / ws), flush=True)
    del tr

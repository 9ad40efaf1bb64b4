#!/usr/bin/env python3
"""Where the env wave of the policy rollout kernel spends its cycles (diagnostic build with in-kernel s_memtime stamps,
-DTB_DIAG_STAMPS; run on the GPU box): one PPO collect with the untrained policy and one under the reference's trained policy,
whose racket goes for the ball. SHARES, not absolute time: the stamps cost ~40 cycles each and fence the scheduler."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from tennisbot_rl_amd import stepper
from tennisbot_rl_amd.build import HIPCC_FLAGS, SOURCES, hipcc

out = "/tmp/libtb_stamps.so"
subprocess.check_call([hipcc()] + HIPCC_FLAGS + ["-DTB_DIAG_STAMPS", "-o", out] + SOURCES)
stepper.use_library(out)
L = stepper.load_library()
L.tb_diag_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
from tennisbot_rl_amd.ppo import PPOTrainer, pack_policy

names = ["between substeps (sampling, env logic, stores, barriers)", "racket narrowphase", "static narrowphase", "velocity update", "contact solve", "pose update"]
buf = (ctypes.c_ulonglong * 16)()
for label in ("untrained", "reference_policy"):
    tr = PPOTrainer("SwingRacket-v0", num_envs=4096, n_steps=1092, seed=0)
    if label == "reference_policy":
        tr.policy.load_sb3_arrays(dict(np.load(os.path.join(ROOT, "tests", "golden", "ppo_swing_policy.npz"))))
    for _ in range(2):
        tr.collect()
    torch.cuda.synchronize()
    L.tb_diag_read_stamps(buf, 1)
    b, env = tr.buf, tr.env
    pack_policy(tr.policy, out=tr.packed)
    rec = b.record
    tr.obs_seq[0].copy_(tr.obs_in)
    env.policy_rollout_ptrs(tr.n_steps, tr.packed.data_ptr(), tr.obs_in.data_ptr(), b.actions[0].data_ptr(), tr._raw_actions.data_ptr(), tr.logps.data_ptr(),
                            tr.values.data_ptr(), b.obs[0].data_ptr(), b.rewards[0].data_ptr(), b.dones[0].data_ptr(), (rec, 0, 0, 0, rec, rec, rec), tr.noise_seed)
    torch.cuda.current_stream().synchronize()  # the rollout kernels are done; the pool's launch waits for flush()
    L.tb_diag_read_stamps(buf, 1)
    v = list(buf)
    env.flush(); torch.cuda.synchronize()
    tot = sum(v[:6]); waves = max(v[9], 1)
    print("%s: %d env waves, stamped %.0f cycles per wave and step" % (label, waves, tot / waves / 26.0))
    for k in range(6):
        print("    %-60s %5.1f %%  (%.0f cycles per wave and step)" % (names[k], 100.0 * v[k] / tot, v[k] / waves / 26.0))
    del tr

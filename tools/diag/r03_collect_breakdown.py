#!/usr/bin/env python3
"""Where does a PPO collect of n_steps = 1100 x 4096 envs spend its time? (GPU) Phases of PPOTrainer._collect_fused timed apart,
with the untrained policy (short flights) and the reference's trained policy (struck balls fly longer)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from tennisbot_rl_amd.ppo import PPOTrainer, pack_policy

def sync(): torch.cuda.synchronize()
out = {}
for label in ("untrained", "reference_policy"):
    tr = PPOTrainer("SwingRacket-v0", num_envs=4096, n_steps=1100, seed=0)
    if label == "reference_policy":
        tr.policy.load_sb3_arrays(dict(np.load(os.path.join(ROOT, "tests", "golden", "ppo_swing_policy.npz"))))
    for _ in range(3): tr.collect()
    sync()
    rows = []
    for rep in range(6):
        buf, env = tr.buf, tr.env
        sync(); t0 = time.perf_counter()
        pack_policy(tr.policy, out=tr.packed); sync(); t1 = time.perf_counter()
        rec = buf.record
        tr.obs_seq[0].copy_(tr.obs_in)
        env.policy_rollout_ptrs(tr.n_steps, tr.packed.data_ptr(), tr.obs_in.data_ptr(), buf.actions[0].data_ptr(), tr._raw_actions.data_ptr(), tr.logps.data_ptr(),
                                tr.values.data_ptr(), buf.obs[0].data_ptr(), buf.rewards[0].data_ptr(), buf.dones[0].data_ptr(), (rec, 0, 0, 0, rec, rec, rec), tr.noise_seed)
        t2h = time.perf_counter()
        torch.cuda.current_stream().synchronize(); t2 = time.perf_counter()   # rollout kernels done (fast-forwards may still run)
        env.flush(); sync(); t3 = time.perf_counter()                            # + the fast-forwards' tail
        with torch.no_grad():
            tr.obs_seq[1:].copy_(buf.obs[:-1]); last = buf.obs[tr.n_steps - 1]; tr.last_value.copy_(tr.policy(last)[1]); tr.obs_in.copy_(last)
        sync(); t4 = time.perf_counter()
        rows.append(dict(pack_ms=(t1 - t0) * 1e3, host_issue_ms=(t2h - t1) * 1e3, rollout_kernels_ms=(t2 - t1) * 1e3, ff_tail_ms=(t3 - t2) * 1e3, bookkeeping_ms=(t4 - t3) * 1e3,
                         total_ms=(t4 - t0) * 1e3, steps_per_s_M=4096 * 1100 / (t4 - t0) / 1e6))
    med = {k: float(np.median([r[k] for r in rows])) for k in rows[0]}
    # the trainer's own collect(), as learn() times it
    ts = []
    for rep in range(6):
        sync(); t0 = time.perf_counter(); tr.collect(); sync(); ts.append(time.perf_counter() - t0)
    med["collect_call_ms"] = float(np.median(ts)) * 1e3
    med["collect_call_steps_per_s_M"] = 4096 * 1100 / float(np.median(ts)) / 1e6
    # ... and inside the training loop: the collect that follows an update() (1.4 s of library GEMMs and elementwise kernels),
    # and the one after that
    adv, ret = tr.advantages(tr.last_value)
    after, second = [], []
    for rep in range(3):
        tr.update(adv, ret); sync()
        t0 = time.perf_counter(); tr.collect(); sync(); after.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); tr.collect(); sync(); second.append(time.perf_counter() - t0)
    med["collect_after_update_ms"] = float(np.median(after)) * 1e3
    med["collect_after_that_ms"] = float(np.median(second)) * 1e3
    c = tr.env.counters()
    med["substeps_per_step"] = c["substeps"] / max(1, c["episodes_finished"] * 26)
    out[label] = med
    print(label, json.dumps({k: round(v, 3) for k, v in med.items()}), flush=True)
    del tr
    torch.cuda.empty_cache()
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r03_collect_breakdown.json"), "w"), indent=1)

#!/usr/bin/env python3
"""Round-2 probe (GPU): numbers that decide designs, not product code.
  1. KS statistic of the reference policy's return distribution vs the PyBullet record (default / URDF inertia / no drag)
  2. what racket<->court contact does to the SwingRacket workload: substeps per step, timeouts, fast-forward length
     distribution, with random actions and with the reference policy"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch


def main():
    import compare_reference_policy as crp
    from tennisbot_rl_amd.params import ENV_SWING, F_DEFAULT, F_RACKET_GROUND, default_params, urdf_file_inertia
    from tennisbot_rl_amd.stepper import BatchedEnv
    out = {}
    ref, _ = crp.reference_record()
    for name, over in (("default", {}), ("urdf_inertia", urdf_file_inertia()), ("no_drag", dict(lin_damp=0.0, ang_damp=0.0)), ("erp_0.2", dict(erp=0.2))):
        r = crp.rollout_rewards(num_envs=4096, episodes=4, **over)
        D, p = crp.ks_two_sample(ref, r)
        out["ks_" + name] = {"D": D, "p": p, **crp.summarize(r)}
        print(name, out["ks_" + name], flush=True)
    for rg in (False, True):
        flags = F_DEFAULT | (F_RACKET_GROUND if rg else 0)
        env = BatchedEnv(ENV_SWING, 4096, device="cuda:0", seed=3, params=default_params(flags=flags))
        env.reset()
        g = torch.Generator(device="cuda:0"); g.manual_seed(5)
        subs = []
        t0 = time.perf_counter()
        for ep in range(6):
            for t in range(26):
                a = torch.rand((4096, 6), device="cuda:0", generator=g) * 2 - 1
                env.step(a)
            subs.append(env.last_substeps().cpu().numpy().copy())
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        s = np.concatenate(subs)
        c = env.counters()
        out["random_rg%d" % rg] = {"seconds": el, "substeps_per_step": c["substeps"] / (4096 * 26 * 6), "timeouts": c["timeouts"], "episodes": c["episodes_finished"],
                                   "ff_mean": float(s.mean()), "ff_p50": float(np.percentile(s, 50)), "ff_p90": float(np.percentile(s, 90)), "ff_p99": float(np.percentile(s, 99)),
                                   "ff_max": int(s.max()), "racket_ball_contact_substeps": c["racket_ball_contact_substeps"],
                                   "wave_max_mean": float(s.reshape(-1, 64).max(1).mean())}
        print("random rg=%d" % rg, out["random_rg%d" % rg], flush=True)
        env.close()
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r02_probe.json"), "w"), indent=1)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Counterpart of the reference's `validate_swing.py` / `validate.py` (play a trained policy and
report how it does) on the batched MI355X envs -- headless only, many episodes at once.

  python validate_swing.py -m model/ppo_SwingRacket-v0.pt            # a checkpoint of train_swing.py
  python validate_swing.py --reference-policy                        # the reference's shipped ppo_swing weights
  python validate_swing.py --env Tennisbot-v0 -m model/ppo_Tennisbot-v0.pt --steps 1000

Like `model.predict(ob)` in validate_swing.py:35 the policy acts stochastically unless --deterministic.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", default="SwingRacket-v0", choices=["SwingRacket-v0", "Tennisbot-v0"])
    ap.add_argument("-m", "--model_file", type=str, default="", help="checkpoint written by train_swing.py --save")
    ap.add_argument("--reference-policy", action="store_true", help="tests/golden/ppo_swing_policy.npz (exported from backup_models/ppo_swing.zip)")
    ap.add_argument("--num-envs", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=104, help="agent steps per env (4 SwingRacket episodes)")
    ap.add_argument("--deterministic", action="store_true")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--headless", action="store_true", help="accepted for CLI compatibility; there is no GUI")
    args = ap.parse_args()

    import numpy as np
    import torch
    from tennisbot_rl_amd.ppo import PPOTrainer, pack_policy

    tr = PPOTrainer(args.env, num_envs=args.num_envs, n_steps=26, device="cuda:0", seed=args.seed, graph=False)
    if args.reference_policy:
        tr.policy.load_sb3_arrays(dict(np.load(os.path.join(ROOT, "tests", "golden", "ppo_swing_policy.npz"))))
    elif args.model_file:
        tr.load(args.model_file)
    else:
        sys.exit("give -m <checkpoint> or --reference-policy")
    print("------------- start running -------------")
    env, blob, obs = tr.env, pack_policy(tr.policy), tr.obs_in
    ret = torch.zeros(args.num_envs, device=env.device)
    finished, total = [], 0
    for t in range(args.steps):
        (obs, rew, done), _ = env.policy_step(blob, obs, seed=args.seed + 1, deterministic=args.deterministic)
        env.flush()
        ret += rew
        d = done.bool()
        if bool(d.any()):
            finished.append(ret[d].cpu().numpy())
            total += int(d.sum())
            ret[d] = 0.0
            print("-------------------- Done %d (mean episode reward so far %.3f) --------------------" % (total, float(np.concatenate(finished).mean())))
    if finished:
        r = np.concatenate(finished)
        print("%d episodes: mean reward %.3f, median %.3f, goal hits (reward >= 50) %.1f %%" % (r.size, r.mean(), np.median(r), 100.0 * (r >= 50).mean()))
    else:
        print("no episode finished in %d steps" % args.steps)


if __name__ == "__main__":
    main()

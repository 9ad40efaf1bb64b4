#!/usr/bin/env python3
"""Counterpart of the reference's `train_swing.py` / `train.py` on the batched MI355X envs.

  python train_swing.py                      # PPO on SwingRacket-v0, 4096 envs on cuda:0
  python train_swing.py --env Tennisbot-v0 --curri
  torchrun --nproc-per-node 8 train_swing.py # one process per GPU, sharded envs

Hyper-parameters follow train_swing.py:46-50,80-91 (net_arch pi=vf=[32,64,32], ent_coef 0.002,
total_timesteps 2e6) and train.py:76-80,104-110 (Tennisbot: 64-64, ent_coef 0.01, 1e6); the
rollout is n_steps per env x num_envs instead of 1100 x 1. `--curri` is the racket-size
curriculum of train.py:155-176 (progress thresholds -> racket scale, applied at each env's
next reset). Checkpoints hold the learner AND the env batch state.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CURRICULUM = ((3, 3.0), (5, 2.6), (10, 2.3), (15, 2.1), (25, 1.9), (45, 1.7), (70, 1.3), (101, 1.0))  # train.py:164-176 (% progress, scale)


def racket_scale_for(progress_percent):
    for limit, scale in CURRICULUM:
        if progress_percent < limit:
            return scale
    return 1.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", default="SwingRacket-v0", choices=["SwingRacket-v0", "Tennisbot-v0"])
    ap.add_argument("--num-envs", type=int, default=4096)
    ap.add_argument("--n-steps", type=int, default=104, help="agent steps per env per rollout (4 SwingRacket episodes)")
    ap.add_argument("--total-timesteps", type=float, default=None, help="default: 2e6 Swing / 1e6 Tennisbot, as the reference")
    ap.add_argument("--load", type=str, default=None, help="checkpoint written by --save")
    ap.add_argument("--load-reference", action="store_true", help="warm start from the reference's shipped ppo_swing policy (tests/golden/ppo_swing_policy.npz)")
    ap.add_argument("--save", type=str, default="./model/ppo_%s.pt")
    ap.add_argument("--curri", action="store_true", help="curriculum learning: size change of racket (Tennisbot-v0)")
    ap.add_argument("--gui", action="store_true", help="accepted for CLI compatibility; there is no GUI")
    ap.add_argument("-s", "--select", default="ppo", help="only ppo is provided (sac / tqc / trpo are third-party learners)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--racket-ground", action="store_true", help="also simulate racket<->court contact (court.urdf:19-24; TB_F_RACKET_GROUND, opt-in: DESIGN.md section 3)")
    ap.add_argument("--rolling-friction", action="store_true", help="also solve the rolling-friction rows of every ball contact (rollingFriction=.001 in racket.py:43-45, objects.py:29-31,48-50)")
    ap.add_argument("--no-fused", action="store_true", help="run the policy as torch modules between env steps instead of inside the step kernel")
    ap.add_argument("--log-json", type=str, default=None)
    args = ap.parse_args()
    if args.select != "ppo":
        sys.exit("only -s ppo is implemented on the batched envs")

    import torch
    from tennisbot_rl_amd.ppo import PPOTrainer

    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.distributed.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    total = args.total_timesteps or (2e6 if args.env == "SwingRacket-v0" else 1e6)
    params = None
    if args.racket_ground or args.rolling_friction:
        from tennisbot_rl_amd.params import F_DEFAULT, F_RACKET_GROUND, default_params, reference_rolling_friction
        params = default_params(flags=F_DEFAULT | (F_RACKET_GROUND if args.racket_ground else 0), **(reference_rolling_friction() if args.rolling_friction else {}))
    tr = PPOTrainer(args.env, num_envs=args.num_envs, n_steps=args.n_steps, device=torch.device("cuda", local_rank), seed=args.seed,
                    fused=not args.no_fused, params=params)
    if args.load_reference:
        import numpy as np
        tr.policy.load_sb3_arrays(dict(np.load(os.path.join(ROOT, "tests", "golden", "ppo_swing_policy.npz"))))
    if args.load:
        tr.load(args.load)
    history = []
    while tr.num_timesteps < total:
        if args.curri and args.env == "Tennisbot-v0":
            tr.env.set_racket_scale(racket_scale_for(100.0 * tr.num_timesteps / total))
        history += tr.learn(min(total, tr.num_timesteps + tr.n_steps * tr.num_envs * world))
    if tr.rank == 0:
        path = args.save % args.env if "%s" in args.save else args.save
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        tr.save(path)
        print("saved", path, "eval (stochastic policy, as EvalCallback in the reference):", tr.evaluate())
        if args.log_json:
            json.dump(history, open(args.log_json, "w"))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

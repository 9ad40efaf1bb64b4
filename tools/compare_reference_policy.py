#!/usr/bin/env python3
"""A statistical cross-check against PyBullet through the reference's own artifacts: the shipped PPO
policy (tests/golden/ppo_swing_policy.npz, exported from backup_models/ppo_swing.zip) is rolled out,
with the exploration noise it was trained with, on the batched HIP envs, and its episode rewards are
compared with the 100 PyBullet episodes stable-baselines3 recorded when that file was saved
(tests/golden/ppo_swing_reference_episodes.json). Not a trajectory-level parity proof -- the policy
is stochastic and the start states are random -- but an engine whose contact, restitution or timing
were off would move the goal-hit rate and the reward clusters."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def bullet_recomputed_inertia():
    from tennisbot_rl_amd.params import bullet_shape_inertia
    return bullet_shape_inertia()


def rollout_rewards(num_envs=4096, episodes=4, seed=0, racket_ground=False, gamma=None, net=True, **overrides):
    """episode returns of the reference's shipped policy (stochastic, its own log_std) on the HIP envs. gamma: also return, per
    episode, the DISCOUNTED return sum_t gamma^t r_t and the reference critic's V(s0) (its value head, evaluated by the same fused
    kernel on the episode's reset observation): (returns, discounted, v0)."""
    import torch
    from tennisbot_rl_amd.params import ENV_SWING, F_DEFAULT, F_RACKET_GROUND, default_params
    from tennisbot_rl_amd.ppo import build_actor_critic, pack_policy
    from tennisbot_rl_amd.stepper import BatchedEnv
    policy = build_actor_critic(6, 6, (32, 64, 32)).to("cuda:0")
    policy.load_sb3_arrays(dict(np.load(os.path.join(ROOT, "tests", "golden", "ppo_swing_policy.npz"))))
    blob = pack_policy(policy)
    flags = F_DEFAULT | (F_RACKET_GROUND if racket_ground else 0)
    if not net:
        from tennisbot_rl_amd.params import F_NET
        flags &= ~F_NET
    env = BatchedEnv(ENV_SWING, num_envs, device="cuda:0", seed=seed, pipeline=True, track_terminal_obs=False, params=default_params(flags=flags, **overrides))
    obs = env.reset()
    fused = True  # (since round 3 the fused policy kernels are instantiated with the extended contact set too)
    torch.manual_seed(seed + 17)
    out, disc, v0s = [], [], []
    for ep in range(episodes):
        steps, v0 = [], None
        for t in range(26):
            if fused:
                (obs, rew, done), (_, _, _, value) = env.policy_step(blob, obs, seed=seed + 17)
            else:
                with torch.no_grad():
                    act, value, _ = policy.act(obs)
                obs, rew, done = env.step(act.clamp(-1.0, 1.0))
            if t == 0:
                v0 = value.clone()
            steps.append((rew, done))
        env.flush()  # terminal rewards arrive from the side streams
        r = torch.stack([r for r, _ in steps])
        assert bool(steps[-1][1].all()) and not bool(torch.stack([d for _, d in steps[:-1]]).any())
        out.append(r.sum(0).cpu().numpy())
        if gamma is not None:
            w = torch.tensor([gamma ** t for t in range(26)], device=r.device, dtype=torch.float64)
            disc.append((r.double() * w[:, None]).sum(0).cpu().numpy())
            v0s.append(v0.double().cpu().numpy())
    env.close()
    if gamma is not None:
        return np.concatenate(out), np.concatenate(disc), np.concatenate(v0s)
    return np.concatenate(out)


def critic_calibration(v0, disc, bins=10):
    """The reference's critic (value_net of backup_models/ppo_swing.zip, trained under PyBullet on discounted returns, gamma = 0.99)
    against the returns realised HERE: episodes binned into `bins` quantile bins of V(s0); per bin the mean prediction and the mean
    realised discounted return; least-squares line realised = slope * predicted + intercept through the bin means, and the
    correlation of the per-episode values. A state-conditional check with thousands of samples: an engine that treats some
    region of start states differently from PyBullet bends the line there."""
    v0, disc = np.asarray(v0, np.float64), np.asarray(disc, np.float64)
    order = np.argsort(v0)
    chunks = np.array_split(order, bins)
    pred = np.array([v0[c].mean() for c in chunks])
    real = np.array([disc[c].mean() for c in chunks])
    sem = np.array([disc[c].std(ddof=1) / np.sqrt(c.size) for c in chunks])
    slope, intercept = np.polyfit(pred, real, 1)
    return {"bins": [{"n": int(c.size), "predicted": float(p), "realised": float(r), "sem": float(e)} for c, p, r, e in zip(chunks, pred, real, sem)],
            "slope": float(slope), "intercept": float(intercept), "corr": float(np.corrcoef(v0, disc)[0, 1]),
            "mean_predicted": float(v0.mean()), "mean_realised": float(disc.mean()), "max_bin_gap": float(np.abs(pred - real).max())}


def reference_record(exclude_interrupted=True):
    """(rewards, note) of the reference's PyBullet record. Two of its 100 episodes are 38 steps long although every
    SwingRacket episode is exactly 26 (swingracket_env.py:105-129): train_swing.py:111 hands EvalCallback the TRAINING env
    object, so every 1000 timesteps the evaluation resets that env in the middle of a training
    episode -- 1000 = 38 * 26 + 12 -- runs its own episodes, and leaves a fresh world behind; the training Monitor's
    episode then lasts its 12 steps before the interruption + a full 26 after it, acts once on a stale observation, and sums
    the rewards of two different worlds (the record's wall-clock shows the evaluation's 0.87 s gap inside exactly those two).
    They are not samples of the episode-return distribution and are left out of the distribution tests."""
    rec = json.load(open(os.path.join(ROOT, "tests", "golden", "ppo_swing_reference_episodes.json")))
    r, n = np.asarray(rec["episode_rewards"], np.float64), np.asarray(rec["episode_lengths"])
    keep = (n == 26) if exclude_interrupted else np.ones(r.size, bool)
    return r[keep], int((~keep).sum())


def ks_two_sample(a, b):
    """two-sample Kolmogorov-Smirnov statistic D and its asymptotic p-value (scipy.stats.ks_2samp)"""
    from scipy.stats import ks_2samp
    res = ks_2samp(np.asarray(a, np.float64), np.asarray(b, np.float64), alternative="two-sided", method="asymp")
    return float(res.statistic), float(res.pvalue)


def summarize(r):
    r = np.asarray(r, dtype=np.float64)
    goal = r >= 50.0  # only the goal bonus (+50) lifts an episode that high
    rest = r[~goal]
    # one contact bonus (+2) per agent step with racket contact before step 25; a good shot scores at most
    # 18.6 (+2), a goal 50 + ~19 (+2): anything in (21, 24) or (72, 76) carries TWO bonuses on a good shot
    two = ((r > 21.0) & (r < 24.0)) | ((r > 72.0) & (r < 76.0))
    return {"episodes": int(r.size), "mean": float(r.mean()), "goal_rate": float(goal.mean()), "two_bonus_good_shots": float(two.mean()),
            "goal_cluster_mean": float(r[goal].mean()) if goal.any() else None,
            "other_median": float(np.median(rest)), "other_p10": float(np.percentile(rest, 10)), "other_p90": float(np.percentile(rest, 90))}


def main():
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "ppo_swing_reference_episodes.json")))
    a = summarize(ref["episode_rewards"])
    print("PyBullet (reference's record, 100 episodes):", json.dumps(a))
    se = (a["goal_rate"] * (1 - a["goal_rate"]) / a["episodes"]) ** 0.5
    variants = {"default parameters": {}, "inertia as in the URDF files": dict(racket_inertia=(0.04, 0.08, 0.12), ball_inertia=1.0),
                "inertia recomputed from the collision shapes": bullet_recomputed_inertia(),
                "racket recomputed, ball as in the URDF": dict(racket_inertia=bullet_recomputed_inertia()["racket_inertia"], ball_inertia=1.0),
                "ball recomputed, racket as in the URDF": dict(racket_inertia=(0.04, 0.08, 0.12), ball_inertia=bullet_recomputed_inertia()["ball_inertia"])}
    if len(sys.argv) > 1 and sys.argv[1] == "--default-only":
        variants = {"default parameters": {}}
    if len(sys.argv) > 1 and sys.argv[1] == "--racket-ground":
        variants = {"default parameters (racket falls through the court)": {}, "racket<->court contact on": dict(racket_ground=True)}
    if len(sys.argv) > 1 and sys.argv[1] == "--rolling":
        from tennisbot_rl_amd.params import reference_rolling_friction
        variants = {"default parameters": {}, "rolling-friction rows on": reference_rolling_friction(),
                    "rolling-friction rows on, 10 x the coefficient": {k: 10 * v for k, v in reference_rolling_friction().items()}}
    if len(sys.argv) > 1 and sys.argv[1] == "--sensitivity":
        # not a fit (100 episodes cannot carry one): which recalled constants does the record constrain at all?
        variants = {"default parameters": {}, "no damping": dict(lin_damp=0.0, ang_damp=0.0), "damping 0.02": dict(lin_damp=0.02, ang_damp=0.02),
                    "damping 0.08": dict(lin_damp=0.08, ang_damp=0.08),
                    "racket restitution 0.9 (not the product 0.81)": dict(rest_racket=0.9), "racket restitution 0.5": dict(rest_racket=0.5),
                    "racket friction 0.2 (not the product 0.04)": dict(fric_racket=0.2), "restitution threshold 1.0": dict(rest_vel_threshold=1.0),
                    "racket mass 2": dict(racket_mass=2.0), "ball mass 0.058": dict(ball_mass=0.058), "10 solver iterations": dict(solver_iters=10)}
    ref98, dropped = reference_record()
    print("distribution tests use the %d uninterrupted episodes (%d were cut by an evaluation on the training env)" % (ref98.size, dropped))
    for name, over in variants.items():
        rew = rollout_rewards(**over)
        b = summarize(rew)
        b["ks_D"], b["ks_p"] = ks_two_sample(ref98, rew)
        print("HIP envs, %s %s:\n    %s" % (name, {k: (tuple(round(x, 5) for x in v) if isinstance(v, tuple) else v) for k, v in over.items()}, json.dumps(b)))
        rest = np.array([x for x in ref["episode_rewards"] if x < 50.0])
        se_med = 1.2533 * rest.std(ddof=1) / len(rest) ** 0.5
        print("    goal rate %+.3f = %+.1f s.e.;  median of the other episodes %+.2f = %+.1f s.e. (of the 100-episode sample)" % (
            b["goal_rate"] - a["goal_rate"], (b["goal_rate"] - a["goal_rate"]) / se, b["other_median"] - a["other_median"], (b["other_median"] - a["other_median"]) / se_med))


if __name__ == "__main__":
    main()

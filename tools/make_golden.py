#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the float64 CPU oracle (the numerical "truth" build).

The reference ships no golden vectors and PyBullet cannot be run here, so
these fixtures pin the build's OWN semantics: any later change to the oracle, the kernels
or the parameter defaults that moves a trajectory shows up as a diff against committed
data. They are NOT PyBullet outputs ("parity unpinned" at that boundary).

  swing_trajectories.npz   8 SwingRacket-v0 episodes (26 agent steps, auto-reset off)
  tennis_trajectories.npz  8 Tennisbot-v0 runs of 800 agent steps (auto-reset off)
Each holds: seed, actions [T,n,A] f32, obs0 [n,O], obs [T,n,O] f32, reward [T,n] f32,
done [T,n] u8, substeps [T,n] i32, final state rows [W,n] f64 and final done byte.
Action sequences are chosen so that several episodes hit the ball (racket swung at it).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import OracleBatch  # noqa: E402
from tennisbot_rl_amd.params import ENV_SWING, ENV_TENNIS, default_params  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def run(kind, n, T, seed, actions, precision="f64"):
    b = OracleBatch(default_params(), kind, n, seed=seed, precision=precision)
    obs0 = b.reset()
    obs, rew, done, sub = [], [], [], []
    for t in range(T):
        o, r, d, s = b.step(actions[t])
        obs.append(o); rew.append(r); done.append(d); sub.append(s)
    vals, dbyte = b.get_state_f64()
    c = b.counters()
    return dict(obs0=obs0, obs=np.stack(obs), reward=np.stack(rew), done=np.stack(done), substeps=np.stack(sub),
                final_state=vals, final_done=dbyte, counters=c)


def swing_actions(n, T, rng):
    a = rng.uniform(-1, 1, (T, n, 6)).astype(np.float32)
    # half of the envs: a committed swing toward -x (the ball sits ~0.47 m in front of the face),
    # with a little lift and spin, so that real racket<->ball impacts are part of the fixture
    k = n // 2
    a[:, :k, 0] = -rng.uniform(0.55, 1.0, (1, k))
    a[:, :k, 1] = rng.uniform(-0.08, 0.08, (1, k))
    a[:, :k, 2] = rng.uniform(0.0, 0.35, (1, k))
    a[:, :k, 3:] *= 0.15
    return a


def tennis_closed_loop(n, T, seed, rng):
    """actions from a y-tracking P-controller on the float64 oracle's own observations (plus
    noise), recorded as float32 so that any implementation can replay them open-loop: the
    racket intercepts the incoming ball in several envs (contact rewards, tennisbot_env.py:170-174)"""
    b = OracleBatch(default_params(), ENV_TENNIS, n, seed=seed, precision="f64")
    o = b.reset()
    acts = np.zeros((T, n, 2), np.float32)
    for t in range(T):
        ay = np.clip(4.0 * (o[:, 7] - o[:, 1]) - 1.5 * o[:, 4], -1, 1)
        ax = np.clip(0.5 * (9.5 - o[:, 0]) - 0.8 * o[:, 3], -1, 1)
        a = np.stack([ax, ay], 1) + rng.normal(0, 0.05, (n, 2))
        a[n // 2:] += rng.uniform(-0.5, 0.5, (n - n // 2, 2))  # second half: sloppier tracking
        acts[t] = np.clip(a, -1, 1).astype(np.float32)
        o, _, _, _ = b.step(acts[t])
    return acts


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(20261004)
    n = 8
    acts = swing_actions(n, 26, rng)
    g = run(ENV_SWING, n, 26, 101, acts)
    np.savez_compressed(os.path.join(OUT, "swing_trajectories.npz"), seed=101, actions=acts, **g)
    print("swing: rewards", g["reward"][-1], "substeps", g["substeps"][-1], "racket contacts", g["counters"][0])
    tseed = 207  # (a seed whose 8 runs contain racket<->ball interceptions with the tracking controller below)
    acts = tennis_closed_loop(n, 800, tseed, np.random.default_rng(20261004 + tseed))
    g = run(ENV_TENNIS, n, 800, tseed, acts)
    np.savez_compressed(os.path.join(OUT, "tennis_trajectories.npz"), seed=tseed, actions=acts, **g)
    print("tennis: done", g["done"][-1], "total reward", g["reward"].sum(0), "racket contacts", g["counters"][0])


if __name__ == "__main__":
    main()

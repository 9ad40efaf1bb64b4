#!/usr/bin/env python3
"""Soak of the fused policy rollout (tb_policy_rollout: policy, sampling and env steps in one kernel) against the f32 CPU oracle, under
the reference's trained policy -- where a fifth of an env wave's substeps run the outline sweep (one edge per lane of the env wave in
the 16-env form) and the contact solver. The oracle is stepped with the actions the kernel reports; every observation, reward, done flag
and counter bit for bit. Run on the GPU box:  tests/soak_policy.py <n_envs> <steps> [slices] [rg] [seed=K]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import OracleBatch  # noqa: E402  (a checker script kept with the tests: not collected by pytest)
from tennisbot_rl_amd.params import ACT_DIM, ENV_SWING, F_AUTO_RESET, F_DEFAULT, F_RACKET_GROUND, OBS_DIM, default_params, reference_rolling_friction  # noqa: E402
from tennisbot_rl_amd.ppo import SWING_DEFAULTS, build_actor_critic, pack_policy  # noqa: E402
from tennisbot_rl_amd.stepper import BatchedEnv  # noqa: E402


def main():
    n, T = int(sys.argv[1]), int(sys.argv[2])
    slices = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3].isdigit() else 0
    rg = "rg" in sys.argv[3:]
    seed = ([int(x[5:]) for x in sys.argv[3:] if x.startswith("seed=")] or [21])[0]
    policy = build_actor_critic(OBS_DIM[ENV_SWING], ACT_DIM[ENV_SWING], tuple(SWING_DEFAULTS["net_arch"])).to("cuda:0")
    policy.load_sb3_arrays(dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ppo_swing_policy.npz"))))
    blob = pack_policy(policy)
    p = default_params(flags=F_DEFAULT | (F_RACKET_GROUND if rg else 0), **(reference_rolling_friction() if rg else {}))
    env = BatchedEnv(ENV_SWING, n, device="cuda:0", seed=seed, pipeline=True, track_terminal_obs=False, params=p, options=dict(policy_slices=slices))
    pf = p.copy(); pf.flags |= F_AUTO_RESET
    ref = OracleBatch(pf, ENV_SWING, n, seed=seed, precision="f32")
    ref.L.tbo_set_threads(ref.h, 16)
    o = env.reset()
    assert np.array_equal(o.cpu().numpy(), ref.reset())
    done_steps = 0
    while done_steps < T:
        chunk = min(260, T - done_steps)  # whole episodes per call (<= 64 of them between two joins)
        (obs, rew, done), (act, raw, logp, value) = env.policy_rollout(blob, o, chunk, seed=5 * seed + done_steps)
        env.flush()
        torch.cuda.synchronize()
        act_h, obs_h, rew_h, done_h = act.cpu().numpy(), obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
        for t in range(chunk):
            o2, r2, d2, _ = ref.step(np.ascontiguousarray(act_h[t]))
            for name, a, b in (("obs", obs_h[t], o2), ("reward", rew_h[t], r2)):
                if not np.array_equal(a.view(np.uint32), b.view(np.uint32)):
                    raise SystemExit("MISMATCH %s at step %d" % (name, done_steps + t))
            if not np.array_equal(done_h[t] != 0, d2 != 0):
                raise SystemExit("MISMATCH done at step %d" % (done_steps + t))
        o = obs[-1].contiguous()
        done_steps += chunk
    got, want = env.counters(), [int(x) for x in ref.counters()]
    if list(got.values()) != want:
        raise SystemExit("MISMATCH counters: %r vs %r" % (got, want))
    print("fused policy rollout under the reference's trained policy, %d envs x %d steps (%s envs per env wave%s): %d racket-ball contact substeps, %d goal hits, "
          "%d episode ends, %d timeouts (%d of %d substeps booked by the pool's sealed-fate exit, not run), every observation / reward / done and the "
          "counters bit-identical to the f32 oracle"
          % (n, T, {0: "auto", 1: "16", 3: "48"}[slices], ", full contact set" if rg else "", got["racket_ball_contact_substeps"], got["goal_hits"], got["episodes_finished"],
             got["timeouts"], env.sealed_substeps(), got["substeps"]))


if __name__ == "__main__":
    main()

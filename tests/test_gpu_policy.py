"""The fused policy + step kernel (tb_policy_step) on a real MI355X.

Two claims, tested separately:
  * the env half is the ordinary step: fed the actions the kernel reports, a twin env (same seed)
    driven through tb_step -- itself bit-exact against the oracle, test_gpu_parity.py -- produces
    bit-identical obs / reward / done and state;
  * the policy half is SB3's MlpPolicy: mean, value and log-probability agree with the float64 numpy
    restatement of the torch module within 2e-5 absolute (fp32 tanh/exp through hip's libm vs
    numpy's; the tolerance is written here because this is the floating-point leg of the path), and
    the exploration noise is standard normal and fresh on every step and every graph replay.
"""
import numpy as np
import pytest

from tennisbot_rl_amd.params import ACT_DIM, ENV_SWING, ENV_TENNIS, OBS_DIM

pytestmark = pytest.mark.gpu

POLICY_TOL = 2e-5


@pytest.fixture(scope="module")
def torch():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch


def numpy_policy(policy, obs):
    """float64 restatement of ActorCritic.forward"""
    sd = {k: v.detach().cpu().double().numpy() for k, v in policy.state_dict().items()}

    def tower(prefix, x):
        k = 0
        while "%s.%d.weight" % (prefix, k) in sd:
            x = np.tanh(x @ sd["%s.%d.weight" % (prefix, k)].T + sd["%s.%d.bias" % (prefix, k)])
            k += 2
        return x
    x = obs.astype(np.float64)
    mean = tower("policy_net", x) @ sd["action_net.weight"].T + sd["action_net.bias"]
    value = tower("value_net_body", x) @ sd["value_net.weight"].T + sd["value_net.bias"]
    return mean, value[:, 0], sd["log_std"]


def make(torch, kind, n, seed=5, scale=1.0):
    from tennisbot_rl_amd.ppo import SWING_DEFAULTS, TENNIS_DEFAULTS, build_actor_critic, pack_policy
    from tennisbot_rl_amd.stepper import BatchedEnv
    torch.manual_seed(seed)
    arch = (SWING_DEFAULTS if kind == ENV_SWING else TENNIS_DEFAULTS)["net_arch"]
    policy = build_actor_critic(OBS_DIM[kind], ACT_DIM[kind], tuple(arch)).to("cuda:0")
    with torch.no_grad():  # SB3's init has a near-zero action head; make the test see real signal
        policy.action_net.weight.mul_(30.0 * scale)
        policy.log_std.copy_(torch.linspace(-1.0, 0.2, ACT_DIM[kind]))
    env = BatchedEnv(kind, n, device="cuda:0", seed=seed)
    twin = BatchedEnv(kind, n, device="cuda:0", seed=seed)
    return policy, pack_policy(policy), env, twin


@pytest.mark.parametrize("kind,n", [(ENV_SWING, 4096), (ENV_SWING, 257), (ENV_TENNIS, 4096), (ENV_TENNIS, 65)])
def test_deterministic_policy_step_matches_module_and_plain_step(torch, kind, n):
    policy, packed, env, twin = make(torch, kind, n)
    assert packed.numel() == env.policy_floats()
    obs_a, obs_b = env.reset(), twin.reset()
    assert torch.equal(obs_a, obs_b)
    for k in range(40):
        (o, r, d), (act, raw, logp, value) = env.policy_step(packed, obs_a, deterministic=True)
        mean, v, log_std = numpy_policy(policy, obs_a.cpu().numpy())
        np.testing.assert_allclose(raw.cpu().numpy(), mean, atol=POLICY_TOL, rtol=0, err_msg="mean, step %d" % k)
        np.testing.assert_allclose(value.cpu().numpy(), v, atol=POLICY_TOL, rtol=0, err_msg="value, step %d" % k)
        np.testing.assert_allclose(logp.cpu().numpy(), np.full(n, -(log_std + 0.5 * np.log(2 * np.pi)).sum()), atol=POLICY_TOL, rtol=0)
        assert torch.equal(act, raw.clamp(-1.0, 1.0))
        o2, r2, d2 = twin.step(act)
        assert torch.equal(o, o2) and torch.equal(r, r2) and torch.equal(d, d2), "env half differs from tb_step at step %d" % k
        obs_a = o
    wa, da = env.get_state_words()
    wb, db = twin.get_state_words()
    assert torch.equal(wa, wb) and torch.equal(da, db)


@pytest.mark.parametrize("kind", [ENV_SWING, ENV_TENNIS])
def test_stochastic_policy_step_samples_a_diagonal_gaussian(torch, kind):
    n = 8192
    policy, packed, env, twin = make(torch, kind, n, seed=9, scale=0.1)
    obs_a = env.reset()
    twin.reset()
    all_eps, prev = [], None
    for k in range(12):
        (o, r, d), (act, raw, logp, value) = env.policy_step(packed, obs_a, seed=1234)
        mean, v, log_std = numpy_policy(policy, obs_a.cpu().numpy())
        eps = (raw.cpu().numpy().astype(np.float64) - mean) / np.exp(log_std)
        want_logp = (-0.5 * eps ** 2 - log_std - 0.5 * np.log(2 * np.pi)).sum(-1)
        # eps is recovered through a division by std >= e^-1: tolerance scaled accordingly
        np.testing.assert_allclose(logp.cpu().numpy(), want_logp, atol=2e-3, rtol=0)
        np.testing.assert_allclose(value.cpu().numpy(), v, atol=POLICY_TOL, rtol=0)
        assert torch.equal(act, raw.clamp(-1.0, 1.0))
        o2, r2, d2 = twin.step(act)
        assert torch.equal(o, o2) and torch.equal(r, r2) and torch.equal(d, d2)
        if prev is not None:
            assert np.abs(eps - prev).mean() > 0.5, "noise repeated between steps"
        prev = eps
        all_eps.append(eps)
        obs_a = o
    e = np.concatenate(all_eps)  # ~100k x A samples
    assert abs(e.mean()) < 0.01 and abs(e.std() - 1.0) < 0.01
    assert abs((e ** 3).mean()) < 0.05 and abs((e ** 4).mean() - 3.0) < 0.1
    c = np.corrcoef(e.T)
    assert np.abs(c - np.eye(c.shape[0])).max() < 0.02, "action dimensions are correlated"
    # a different seed draws different noise; the same seed on a fresh twin batch reproduces it
    envb = type(env)(kind, n, device="cuda:0", seed=9)
    envc = type(env)(kind, n, device="cuda:0", seed=9)
    ob = envb.reset(); oc = envc.reset()
    _, (_, raw_b, _, _) = envb.policy_step(packed, ob, seed=1234)
    _, (_, raw_c, _, _) = envc.policy_step(packed, oc, seed=99)
    env0 = type(env)(kind, n, device="cuda:0", seed=9)
    _, (_, raw_0, _, _) = env0.policy_step(packed, env0.reset(), seed=1234)
    assert torch.equal(raw_b, raw_0) and not torch.equal(raw_b, raw_c)


def test_fused_rollout_pipelined_graph_equals_eager_unpipelined(torch):
    """the whole collect of the PPO trainer -- fused kernel, pipelined fast-forward, one hipGraph --
    against the same trainer stepping eagerly without the pipeline: identical buffers"""
    from tennisbot_rl_amd.ppo import PPOTrainer
    a = PPOTrainer("SwingRacket-v0", num_envs=1024, n_steps=26, device="cuda:0", seed=3, pipeline=True, graph=True, fused=True)
    b = PPOTrainer("SwingRacket-v0", num_envs=1024, n_steps=26, device="cuda:0", seed=3, pipeline=False, graph=False, fused=True)
    assert a.fused and b.fused
    for it in range(3):  # 1st: eager warm-up inside collect(), 2nd: capture, 3rd: replay
        a.collect(); b.collect()
        torch.cuda.synchronize()
        for name in ("obs", "actions", "rewards", "dones"):
            assert torch.equal(getattr(a.buf, name), getattr(b.buf, name)), "%s differ in rollout %d" % (name, it)
        assert torch.equal(a.values, b.values) and torch.equal(a.logps, b.logps) and torch.equal(a.obs_seq, b.obs_seq)
        assert torch.equal(a.last_value, b.last_value)
    assert a._graph is not None


def test_fused_and_torch_collect_learn_alike(torch):
    from tennisbot_rl_amd.ppo import PPOTrainer
    rewards = {}
    for fused in (True, False):
        tr = PPOTrainer("SwingRacket-v0", num_envs=2048, n_steps=52, device="cuda:0", seed=1, fused=fused, batch_size=26624)
        hist = tr.learn(2048 * 52 * 8, log=None)
        rewards[fused] = (hist[0]["mean_episode_reward"], hist[-1]["mean_episode_reward"])
    for fused, (first, last) in rewards.items():
        assert last > first + 1.0, "fused=%s did not improve: %r" % (fused, rewards)


def test_policy_step_rejects_bad_arguments(torch):
    from tennisbot_rl_amd.params import F_DEFAULT, F_RACKET_GROUND, default_params
    from tennisbot_rl_amd.stepper import BatchedEnv, StepperError
    policy, packed, env, _ = make(torch, ENV_SWING, 64)
    with pytest.raises(ValueError):
        env.policy_step(packed[:-4], env.reset())
    with pytest.raises(ValueError):
        env.policy_step(packed, torch.zeros((64, 5), device="cuda:0"))
    rg = BatchedEnv(ENV_SWING, 64, device="cuda:0", params=default_params(flags=F_DEFAULT | F_RACKET_GROUND))
    with pytest.raises(StepperError):
        rg.policy_step(packed, rg.reset())


def test_reference_policy_reward_distribution_matches_the_pybullet_record(torch):
    """The one PyBullet-derived pin there is: the reference's shipped policy, rolled out with its training noise on
    the HIP envs, against the PyBullet episodes recorded inside backup_models/ppo_swing.zip
    (tests/golden/ppo_swing_reference_episodes.json). Statistical, not trajectory-level (stochastic policy, random
    starts): a two-sample Kolmogorov-Smirnov test on the whole episode-return distribution, 98 recorded episodes
    (the two that an evaluation on the training env cut in two are left out, see compare_reference_policy.
    reference_record) against 16 384 simulated ones. Thresholds are fixed here: D below the 1 % critical value
    1.628 / sqrt(98) = 0.164, i.e. the record does not reject this engine at the 1 % level.
    NO HOLD-OUT: two engine constants -- inertia derived from the collision shapes (params.bullet_shape_inertia)
    instead of the URDF files' values, and contact ERP 0.08 instead of 0.2 -- were SELECTED on this very record in
    round 1; for them this test is a regression pin of that choice, not independent evidence.
    What it does show independently of that choice is discrimination: the URDF-file inertia, or no drag, are
    rejected by the same test."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import compare_reference_policy as crp
    from tennisbot_rl_amd.params import urdf_file_inertia
    ref, dropped = crp.reference_record()
    assert ref.size == 98 and dropped == 2
    crit_1pct = 1.628 / ref.size ** 0.5
    got = crp.rollout_rewards(num_envs=4096, episodes=4)
    D, p = crp.ks_two_sample(ref, got)
    assert D < crit_1pct and p > 0.01, (D, p)
    # the same test rejects the alternatives that were open: the URDF files' inertia, and an engine without Bullet's drag
    D_urdf, _ = crp.ks_two_sample(ref, crp.rollout_rewards(num_envs=4096, episodes=4, **urdf_file_inertia()))
    D_nodrag, _ = crp.ks_two_sample(ref, crp.rollout_rewards(num_envs=4096, episodes=4, lin_damp=0.0, ang_damp=0.0))
    assert D_urdf > D + 0.03 and D_nodrag > crit_1pct, (D, D_urdf, D_nodrag)
    # a finer feature ERP 0.08 was chosen on: 5 of the 100 PyBullet episodes keep the racket contact for a second
    # agent step after a good strike; with Bullet's library ERP (0.2) none would here
    s_ref, s_got = crp.summarize(ref), crp.summarize(got)
    assert 0.01 < s_got["two_bonus_good_shots"] < 0.10 and 0.03 < s_ref["two_bonus_good_shots"] < 0.07, (s_got, s_ref)


@pytest.mark.parametrize("kind,n,T,lead", [(ENV_SWING, 1000, 70, 9), (ENV_SWING, 4096, 52, 0), (ENV_TENNIS, 777, 150, 3)])
def test_policy_rollout_equals_repeated_policy_steps(torch, kind, n, T, lead):
    """tb_policy_rollout (whole episodes per launch, weights and env state resident in registers)
    against tb_policy_step called T times: every output of every step bit-identical, from a start in
    the middle of an episode, with the SwingRacket fast-forwards on the side streams"""
    from tennisbot_rl_amd.ppo import SWING_DEFAULTS, TENNIS_DEFAULTS, build_actor_critic, pack_policy
    from tennisbot_rl_amd.stepper import BatchedEnv
    torch.manual_seed(11)
    arch = (SWING_DEFAULTS if kind == ENV_SWING else TENNIS_DEFAULTS)["net_arch"]
    policy = build_actor_critic(OBS_DIM[kind], ACT_DIM[kind], tuple(arch)).to("cuda:0")
    with torch.no_grad():
        policy.action_net.weight.mul_(20.0)
        policy.log_std.fill_(-0.5)
    blob = pack_policy(policy)
    pipe = kind == ENV_SWING
    a = BatchedEnv(kind, n, device="cuda:0", seed=8, pipeline=pipe, track_terminal_obs=False)
    b = BatchedEnv(kind, n, device="cuda:0", seed=8, pipeline=pipe, track_terminal_obs=False)
    oa, ob = a.reset(), b.reset()
    for t in range(lead):
        (oa, _, _), _ = a.policy_step(blob, oa, seed=5)
        (ob, _, _), _ = b.policy_step(blob, ob, seed=5)
    (obs, rew, done), (act, raw, logp, value) = a.policy_rollout(blob, oa, T, seed=5)
    a.flush()
    steps = []
    for t in range(T):
        (ob_next, r, d), (ac, rw, lp, v) = b.policy_step(blob, ob, seed=5)
        steps.append((ob_next, r, d, ac, rw, lp, v))
        ob = ob_next
    b.flush()
    torch.cuda.synchronize()
    for t, (o, r, d, ac, rw, lp, v) in enumerate(steps):
        for name, x, y in (("obs", obs[t], o), ("done", done[t], d), ("actions", act[t], ac), ("raw", raw[t], rw), ("logp", logp[t], lp),
                           ("value", value[t], v), ("reward", rew[t], r)):
            assert torch.equal(x, y), "%s differs at step %d" % (name, t)
    if kind == ENV_SWING:
        assert int(done.sum()) == n * ((lead + T) // 26)
    wa, da = a.get_state_words(); wb, db = b.get_state_words()
    assert torch.equal(wa, wb) and torch.equal(da, db) and a.counters() == b.counters()
    # without the pipeline a SwingRacket rollout launch is refused, loudly
    if kind == ENV_SWING:
        from tennisbot_rl_amd.stepper import StepperError
        c = BatchedEnv(kind, 64, device="cuda:0", seed=8)
        with pytest.raises(StepperError):
            c.policy_rollout(blob, c.reset(), 4)


def test_trainer_collect_by_rollout_launch_equals_per_step_launches(torch):
    from tennisbot_rl_amd.ppo import PPOTrainer
    a = PPOTrainer("SwingRacket-v0", num_envs=1024, n_steps=52, device="cuda:0", seed=3, rollout_launch=True)
    b = PPOTrainer("SwingRacket-v0", num_envs=1024, n_steps=52, device="cuda:0", seed=3, rollout_launch=False)
    assert a.rollout_launch and not b.rollout_launch
    for it in range(3):  # eager, capture, replay
        a.collect(); b.collect()
        torch.cuda.synchronize()
        assert torch.equal(a.buf.raw, b.buf.raw), "rollout buffers differ in round %d" % it
        assert torch.equal(a.values, b.values) and torch.equal(a.logps, b.logps) and torch.equal(a._raw_actions, b._raw_actions)
        assert torch.equal(a.obs_seq, b.obs_seq) and torch.equal(a.last_value, b.last_value)

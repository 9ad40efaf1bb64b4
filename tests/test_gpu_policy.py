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


def extended_params():
    """the reference's full contact set: racket<->court contact + the rolling-friction rows"""
    from tennisbot_rl_amd.params import F_DEFAULT, F_RACKET_GROUND, default_params, reference_rolling_friction
    return default_params(flags=F_DEFAULT | F_RACKET_GROUND, **reference_rolling_friction())


def make(torch, kind, n, seed=5, scale=1.0, params=None):
    from tennisbot_rl_amd.ppo import SWING_DEFAULTS, TENNIS_DEFAULTS, build_actor_critic, pack_policy
    from tennisbot_rl_amd.stepper import BatchedEnv
    torch.manual_seed(seed)
    arch = (SWING_DEFAULTS if kind == ENV_SWING else TENNIS_DEFAULTS)["net_arch"]
    policy = build_actor_critic(OBS_DIM[kind], ACT_DIM[kind], tuple(arch)).to("cuda:0")
    with torch.no_grad():  # SB3's init has a near-zero action head; make the test see real signal
        policy.action_net.weight.mul_(30.0 * scale)
        policy.log_std.copy_(torch.linspace(-1.0, 0.2, ACT_DIM[kind]))
    env = BatchedEnv(kind, n, device="cuda:0", seed=seed, params=params)
    twin = BatchedEnv(kind, n, device="cuda:0", seed=seed, params=params)
    return policy, pack_policy(policy), env, twin


@pytest.mark.parametrize("kind,n,full", [(ENV_SWING, 4096, False), (ENV_SWING, 257, False), (ENV_TENNIS, 4096, False), (ENV_TENNIS, 65, False),
                                         (ENV_SWING, 1000, True), (ENV_TENNIS, 1000, True)])
def test_deterministic_policy_step_matches_module_and_plain_step(torch, kind, n, full):
    # full: the reference's full contact set (racket<->court contact, rolling friction) -- f3 composes with f1
    policy, packed, env, twin = make(torch, kind, n, params=extended_params() if full else None)
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
    (o, r, d), _ = rg.policy_step(packed, rg.reset())  # (refused until round 3: the policy kernels now exist for the extended contact set)
    assert o.shape == (64, 6) and bool(torch.isfinite(o).all())
    with pytest.raises(StepperError):
        BatchedEnv(ENV_SWING, 64, device="cuda:0", options=dict(policy_slices=2))


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


def test_leave_half_out_selection_of_the_two_constants_chosen_on_the_record(torch):
    """HOLD-OUT for the two engine constants that were selected on the PyBullet record in round 1 (inertia source, contact ERP).
    The rule is fixed here, before the second half of the record is looked at: of the four candidates {inertia from the collision
    shapes, from the URDF files} x {ERP 0.08, 0.2}, drop those under which the FIRST half's count of double-bonus episodes (racket
    contact reported on two consecutive agent steps of a good shot: 3 of 49) has binomial probability < 0.01, and of the rest take the
    smallest two-sample KS D against the first half's returns (episodes 0-48). The winner is then TESTED on episodes 49-97 with
    thresholds fixed here: D below the 1 % critical value 1.628 / sqrt(49) = 0.233, and the second half's double-bonus count (2 of
    49) not improbable under it (p >= 0.01). Measured (profiles/r03_pin_sensitivity.md): shape/0.08 wins (ERP 0.2 never produces a
    double bonus: p = 0; URDF inertia: D 0.196 vs 0.149) and passes with D = 0.144, p = 0.47; both URDF candidates fail the second half."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import compare_reference_policy as crp
    from pin_sensitivity import binom_tail, two_bonus_count
    from tennisbot_rl_amd.params import urdf_file_inertia
    ref, _ = crp.reference_record()
    first, second = ref[:49], ref[49:]
    assert first.size == second.size == 49 and two_bonus_count(first) == 3 and two_bonus_count(second) == 2
    cands = {}
    for iname, iover in (("shape", {}), ("urdf", urdf_file_inertia())):
        for erp in (0.08, 0.2):
            r = crp.rollout_rewards(num_envs=4096, episodes=4, erp=erp, **iover)
            rate = crp.summarize(r)["two_bonus_good_shots"]
            cands[(iname, erp)] = dict(D1=crp.ks_two_sample(first, r)[0], D2=crp.ks_two_sample(second, r)[0],
                                       p1=binom_tail(3, 49, rate), p2=binom_tail(2, 49, rate))
    alive = [k for k, v in cands.items() if v["p1"] >= 0.01]
    chosen = min(alive, key=lambda k: cands[k]["D1"])
    assert chosen == ("shape", 0.08), (chosen, cands)          # what round 1 had selected on the whole record
    crit = 1.628 / 49 ** 0.5
    won = cands[chosen]
    assert won["D2"] < crit and won["p2"] >= 0.01, won          # ... holds on the half it was not selected on
    assert cands[("urdf", 0.08)]["D2"] > crit and cands[("urdf", 0.2)]["D2"] > crit, cands   # and the alternatives do not
    assert cands[("shape", 0.2)]["p2"] < 0.01, cands


def test_reference_critic_predicts_the_returns_realised_here(torch):
    """A state-conditional check with thousands of samples instead of a 98-sample marginal: the reference's CRITIC (value head of
    backup_models/ppo_swing.zip, weights in tests/golden/ppo_swing_policy.npz) was fitted under PyBullet to the discounted return
    (gamma = 0.99) from the reset observation. 16 384 episodes of the shipped stochastic policy on the HIP envs, binned into 10
    quantile bins of V(s0): the realised mean return per bin against the predicted one. Bounds fixed here (measured: slope 0.987,
    intercept -1.20, correlation 0.53, means 24.9 predicted / 23.3 realised, largest bin gap 3.6; profiles/r03_pin_sensitivity.md):
    0.85 <= slope <= 1.15, |intercept| <= 3, per-episode correlation >= 0.45, means within 12 %, no bin off by more than 5.5.
    It discriminates: with the URDF files' inertia or half / double the damping the slope is below 0.2 (measured -0.03, 0.14, -0.61)
    -- those engines send the balls somewhere else than PyBullet did, in a way that depends on the start state. Not a trajectory pin:
    the parity text stays "unpinned"."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import compare_reference_policy as crp
    from tennisbot_rl_amd.params import urdf_file_inertia
    _, disc, v0 = crp.rollout_rewards(num_envs=4096, episodes=4, gamma=0.99)
    c = crp.critic_calibration(v0, disc)
    assert disc.size == 16384 and len(c["bins"]) == 10
    assert 0.85 <= c["slope"] <= 1.15 and abs(c["intercept"]) <= 3.0 and c["corr"] >= 0.45, c
    assert abs(c["mean_realised"] / c["mean_predicted"] - 1.0) <= 0.12 and c["max_bin_gap"] <= 5.5, c
    for over in (urdf_file_inertia(), dict(lin_damp=0.02, ang_damp=0.02), dict(lin_damp=0.08, ang_damp=0.08)):
        _, d2, v2 = crp.rollout_rewards(num_envs=4096, episodes=4, gamma=0.99, **over)
        c2 = crp.critic_calibration(v2, d2)
        assert c2["slope"] < 0.5 and c2["max_bin_gap"] > 15.0, (over, c2)


@pytest.mark.parametrize("kind,n,T,lead,full,slices", [(ENV_SWING, 1000, 70, 9, False, 0), (ENV_SWING, 4096, 52, 0, False, 0), (ENV_TENNIS, 777, 150, 3, False, 0),
                                                       (ENV_SWING, 1000, 70, 9, False, 3), (ENV_TENNIS, 5000, 40, 3, False, 0),
                                                       (ENV_SWING, 1000, 104, 5, True, 0), (ENV_SWING, 1000, 52, 0, True, 3), (ENV_TENNIS, 777, 150, 3, True, 1)])
def test_policy_rollout_equals_repeated_policy_steps(torch, kind, n, T, lead, full, slices):
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
    # slices: TbOptions.policy_slices (0 = auto: one 16-env slice per workgroup up to 4096 envs, three above); full: the extended contact set
    prm = extended_params() if full else None
    a = BatchedEnv(kind, n, device="cuda:0", seed=8, pipeline=pipe, track_terminal_obs=False, params=prm, options=dict(policy_slices=slices))
    b = BatchedEnv(kind, n, device="cuda:0", seed=8, pipeline=pipe, track_terminal_obs=False, params=prm)
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


@pytest.mark.parametrize("n,slices,n_edges", [(1000, 0, 0), (4096, 0, 0), (5000, 0, 0), (1000, 3, 0), (1000, 1, 64), (777, 1, 63), (900, 1, 5), (900, 1, 3),
                                              (600, 3, 11), (800, 1, 16), (800, 1, 17), (800, 1, 33), (800, 1, 48), (800, 1, 49), (700, 3, 64), (640, 3, 3)])
def test_fused_rollout_under_the_trained_policy_matches_the_oracle(torch, n, slices, n_edges):
    """the env wave of tb_policy_rollout against the ORACLE, directly, where it works hardest: under the reference's trained policy the
    racket goes for the ball, and a fifth of an env wave's substeps run the outline sweep and the contact solver (random weights
    hardly ever get there). Three episodes in one call; the oracle is stepped with the actions the kernel reports; every
    observation, reward, done flag and counter bit for bit, with batch sizes that end in partial waves and slices.
    n_edges: other racket outlines (tests/outlines.py) -- the env wave shares its outline sweeps, four queries at a time, 16 lanes each
    (outline_sweep_rows; 16 / 17, 48 / 49 = a lane's second / fourth edge begins; 64 edges = four per lane, 63 = one lane with three, 3 and 5 = most lanes with none, 11 = one edge for some), with 16 and with 48 envs per env wave"""
    import os
    from tennisbot_rl_amd.params import F_AUTO_RESET, default_params
    from tennisbot_rl_amd.ppo import SWING_DEFAULTS, build_actor_critic, pack_policy
    from tennisbot_rl_amd.stepper import BatchedEnv
    from oracle import OracleBatch
    policy = build_actor_critic(OBS_DIM[ENV_SWING], ACT_DIM[ENV_SWING], tuple(SWING_DEFAULTS["net_arch"])).to("cuda:0")
    policy.load_sb3_arrays(dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "ppo_swing_policy.npz"))))
    blob = pack_policy(policy)
    p = default_params()
    if n_edges:
        from outlines import with_outline
        p = with_outline(p, n_edges)
    env = BatchedEnv(ENV_SWING, n, device="cuda:0", seed=8, pipeline=True, track_terminal_obs=False, params=p, options=dict(policy_slices=slices))
    pf = p.copy(); pf.flags |= F_AUTO_RESET
    ref = OracleBatch(pf, ENV_SWING, n, seed=8, precision="f32")
    ref.L.tbo_set_threads(ref.h, 16)
    o = env.reset()
    assert np.array_equal(o.cpu().numpy(), ref.reset())
    T = 78
    (obs, rew, done), (act, raw, logp, value) = env.policy_rollout(blob, o, T, seed=5)
    env.flush()
    torch.cuda.synchronize()
    act_h, obs_h, rew_h, done_h = act.cpu().numpy(), obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
    for t in range(T):
        o2, r2, d2, s2 = ref.step(np.ascontiguousarray(act_h[t]))
        assert np.array_equal(obs_h[t].view(np.uint32), o2.view(np.uint32)), "obs differs at step %d" % t
        assert np.array_equal(rew_h[t].view(np.uint32), r2.view(np.uint32)), "reward differs at step %d" % t
        assert np.array_equal(done_h[t] != 0, d2 != 0), "done differs at step %d" % t
    got, want = env.counters(), ref.counters()
    assert list(got.values()) == [int(x) for x in want], (got, want)
    assert got["racket_ball_contact_substeps"] > (n // 4 if n_edges else n) and got["nonfinite_states"] == 0 and got["lockstep_violations"] == 0
    env.close()
    # ... and the same actions through the ordinary pipelined step kernel (tb_step: one launch per step, the outline table copied
    # lazily by the first wave whose ball gets past the racket's slab -- here most waves' do, the partial last one included)
    twin = BatchedEnv(ENV_SWING, n, device="cuda:0", seed=8, pipeline=True, track_terminal_obs=False, params=p)
    twin.reset()
    outs = []
    for t in range(T):
        o3, r3, d3 = twin.step(act[t])
        assert torch.equal(o3, obs[t]) and torch.equal(d3 != 0, done[t] != 0), "tb_step differs at step %d" % t
        outs.append(r3)  # (a parking step's reward is written by its fast-forward: compared after the join)
    twin.flush()
    torch.cuda.synchronize()
    for t in range(T):
        assert torch.equal(outs[t], rew[t]), "tb_step reward differs at step %d" % t
    assert twin.counters() == got
    twin.close()


@pytest.mark.parametrize("n,slices,scale", [(2000, 1, 3.0), (1500, 3, 2.3), (900, 1, 1.0)])
def test_tennis_fused_rollout_with_scaled_rackets_matches_the_oracle(torch, n, slices, scale):
    """Tennisbot's fused rollout against the oracle, directly: curriculum-sized rackets (train.py:164-176: scale 3 at the start) are hit
    by most balls, so the env wave's racket narrowphase -- the scaled query, the one-edge-per-lane sweep of the 16-env form, the lane-by-
    lane sweep of the 48-env form -- the racket row of the solver and the contact reward all run, with a random policy. 900 steps
    from a common reset (the balls need 2-3 s to reach the rackets), every observation / reward / done and the counters bit for bit."""
    from tennisbot_rl_amd.params import F_AUTO_RESET, default_params
    from tennisbot_rl_amd.ppo import TENNIS_DEFAULTS, build_actor_critic, pack_policy
    from tennisbot_rl_amd.stepper import BatchedEnv
    from oracle import OracleBatch
    torch.manual_seed(3)
    policy = build_actor_critic(OBS_DIM[ENV_TENNIS], ACT_DIM[ENV_TENNIS], tuple(TENNIS_DEFAULTS["net_arch"])).to("cuda:0")
    blob = pack_policy(policy)
    p = default_params(racket_scale=scale)
    env = BatchedEnv(ENV_TENNIS, n, device="cuda:0", seed=9, track_terminal_obs=False, params=p, options=dict(policy_slices=slices))
    pf = p.copy(); pf.flags |= F_AUTO_RESET
    ref = OracleBatch(pf, ENV_TENNIS, n, seed=9, precision="f32")
    ref.L.tbo_set_threads(ref.h, 16)
    o = env.reset()
    assert np.array_equal(o.cpu().numpy(), ref.reset())
    T = 900
    (obs, rew, done), (act, raw, logp, value) = env.policy_rollout(blob, o, T, seed=11)
    torch.cuda.synchronize()
    act_h, obs_h, rew_h, done_h = act.cpu().numpy(), obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
    for t in range(T):
        o2, r2, d2, s2 = ref.step(np.ascontiguousarray(act_h[t]))
        assert np.array_equal(obs_h[t].view(np.uint32), o2.view(np.uint32)), "obs differs at step %d" % t
        assert np.array_equal(rew_h[t].view(np.uint32), r2.view(np.uint32)), "reward differs at step %d" % t
        assert np.array_equal(done_h[t] != 0, d2 != 0), "done differs at step %d" % t
    got, want = env.counters(), ref.counters()
    assert list(got.values()) == [int(x) for x in want], (got, want)
    assert got["racket_ball_contact_substeps"] > (50 if scale > 1.5 else 0) and got["episodes_finished"] > n // 2 and got["nonfinite_states"] == 0, got
    env.close()

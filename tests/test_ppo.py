"""The on-device learner around the hot path: architecture / weight-layout
compatibility with the reference's shipped SB3 policy (CPU), a short training run (GPU)."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ppo_swing_policy.npz")


def numpy_forward(w, obs):
    h = obs
    for i in (0, 2, 4):
        h = np.tanh(h @ w["mlp_extractor__policy_net__%d__weight" % i].T + w["mlp_extractor__policy_net__%d__bias" % i])
    mean = h @ w["action_net__weight"].T + w["action_net__bias"]
    v = obs
    for i in (0, 2, 4):
        v = np.tanh(v @ w["mlp_extractor__value_net__%d__weight" % i].T + w["mlp_extractor__value_net__%d__bias" % i])
    return mean, (v @ w["value_net__weight"].T + w["value_net__bias"])[:, 0]


def test_reference_policy_fixture_and_architecture():
    """backup_models/ppo_swing.zip: 6 -> 32 -> 64 -> 32 -> 6 (tanh), same for the value tower
    (train_swing.py:80-82)"""
    import torch
    from tennisbot_rl_amd.ppo import build_actor_critic
    w = dict(np.load(GOLD))
    assert w["mlp_extractor__policy_net__0__weight"].shape == (32, 6) and w["mlp_extractor__policy_net__2__weight"].shape == (64, 32)
    assert w["action_net__weight"].shape == (6, 32) and w["value_net__weight"].shape == (1, 32)
    assert np.allclose(w["log_std"], [-.4295, -.1938, -.2215, -.3629, -.9987, -.6483], atol=1e-4)
    net = build_actor_critic(6, 6, (32, 64, 32)).load_sb3_arrays(w)
    obs = np.random.default_rng(0).uniform(-5, 5, (64, 6)).astype(np.float32)
    mean, value = net(torch.from_numpy(obs))
    m2, v2 = numpy_forward(w, obs.astype(np.float64))
    assert np.allclose(mean.detach().numpy(), m2, atol=1e-5) and np.allclose(value.detach().numpy(), v2, atol=1e-5)
    a, v, lp = net.act(torch.from_numpy(obs), deterministic=True)
    assert torch.equal(a, mean) and lp.shape == (64,)


def test_curriculum_schedule_matches_reference():
    import importlib.util
    spec = importlib.util.spec_from_file_location("train_swing", os.path.join(os.path.dirname(GOLD), "..", "..", "train_swing.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    # train.py:164-176: <3% -> 3, <5 -> 2.6, <10 -> 2.3, <15 -> 2.1, <25 -> 1.9, <45 -> 1.7, <70 -> 1.3, else 1
    assert [m.racket_scale_for(p) for p in (0, 2.9, 3, 4.9, 9, 14, 24, 44, 69, 70, 100)] == [3.0, 3.0, 2.6, 2.6, 2.3, 2.1, 1.9, 1.7, 1.3, 1.0, 1.0]


@pytest.mark.gpu
def test_reference_policy_drives_gpu_and_oracle_identically():
    """realistic (non-uniform) actions: the shipped policy's deterministic actions, computed once
    from the oracle's observations, drive both implementations; results stay bit-identical"""
    import torch
    from oracle import OracleBatch
    from tennisbot_rl_amd.params import ENV_SWING, F_AUTO_RESET, F_DEFAULT, default_params
    from tennisbot_rl_amd.ppo import build_actor_critic
    from tennisbot_rl_amd.stepper import BatchedEnv
    n = 1024
    net = build_actor_critic(6, 6, (32, 64, 32)).load_sb3_arrays(dict(np.load(GOLD)))
    env = BatchedEnv(ENV_SWING, n, seed=77)
    ref = OracleBatch(default_params(flags=F_DEFAULT | F_AUTO_RESET), ENV_SWING, n, seed=77, precision="f32")
    o_gpu, o_cpu = env.reset().cpu().numpy(), ref.reset()
    assert np.array_equal(o_gpu, o_cpu)
    hits = 0
    for t in range(52):
        with torch.no_grad():
            a = net.act(torch.from_numpy(o_cpu), deterministic=True)[0].clamp(-1, 1).numpy().astype(np.float32)
        obs, rew, done = env.step(torch.from_numpy(a).cuda())
        o_cpu, r2, d2, s2 = ref.step(a)
        assert np.array_equal(obs.cpu().numpy(), o_cpu) and np.array_equal(rew.cpu().numpy(), r2) and np.array_equal(done.cpu().numpy(), d2)
        hits += int((r2[:, ] == 2).sum())
    assert env.counters()["racket_ball_contact_substeps"] > 100  # the trained policy does hit the ball
    env.close()


@pytest.mark.gpu
def test_ppo_short_run_and_checkpoint(tmp_path):
    import torch
    from tennisbot_rl_amd.ppo import PPOTrainer
    tr = PPOTrainer("SwingRacket-v0", num_envs=512, n_steps=52, seed=1, n_epochs=2)
    hist = tr.learn(3 * 52 * 512, log=None)
    assert len(hist) == 3 and all(np.isfinite(h["policy_loss"]) and np.isfinite(h["value_loss"]) for h in hist)
    assert hist[0]["episodes"] == 2 * 512  # 52 steps = two 26-step episodes per env
    path = str(tmp_path / "ck.pt")
    tr.save(path)
    w0, _ = tr.env.get_state_words()
    tr2 = PPOTrainer("SwingRacket-v0", num_envs=512, n_steps=52, seed=2, n_epochs=2).load(path)
    w1, _ = tr2.env.get_state_words()
    assert torch.equal(w0, w1) and tr2.num_timesteps == tr.num_timesteps
    for a, b in zip(tr.policy.parameters(), tr2.policy.parameters()):
        assert torch.equal(a, b)
    assert np.isfinite(tr2.evaluate())


@pytest.mark.gpu
def test_resumed_checkpoint_collects_and_learns_like_an_unpipelined_trainer(tmp_path):
    """`train_swing.py --load ck.pt`: tb_set_state re-derives the episode phase of the restored envs, so the default
    collect (fused, whole episodes per launch, pipelined, graph) works on a resumed trainer -- its first rollouts equal
    those of a trainer that uses none of that, byte for byte, and learn() goes on improving nothing silently: a rollout
    without data or with lost terminal rewards raises"""
    import torch
    from tennisbot_rl_amd.ppo import PPOTrainer
    tr = PPOTrainer("SwingRacket-v0", num_envs=512, n_steps=52, seed=1, n_epochs=2)
    tr.learn(2 * 52 * 512, log=None)
    for _ in range(7):  # leave the envs in the middle of an episode: the checkpoint's phase is 7
        tr.env.step(torch.zeros((512, 6), device=tr.device))
    tr.env.flush()
    path = str(tmp_path / "ck.pt")
    tr.save(path)
    a = PPOTrainer("SwingRacket-v0", num_envs=512, n_steps=52, seed=5, n_epochs=2).load(path)
    b = PPOTrainer("SwingRacket-v0", num_envs=512, n_steps=52, seed=5, n_epochs=2, pipeline=False, graph=False, rollout_launch=False).load(path)
    assert a.env.phase() == 7 and a.rollout_launch and a.use_graph and not b.env.pipeline
    for rnd in range(3):  # eager + capture, replay, replay
        a.collect(); b.collect()
        torch.cuda.synchronize()
        assert torch.equal(a.buf.raw, b.buf.raw), "rollout %d after the resume differs" % rnd
        assert torch.equal(a.values, b.values) and torch.equal(a.logps, b.logps) and torch.equal(a.last_value, b.last_value)
        assert float(a.buf.dones.sum()) == 2 * 512 and float(a.buf.rewards.abs().sum()) > 0
    assert a._graph is not None and a.env.counters() == b.env.counters()
    hist = a.learn(a.num_timesteps + 2 * 52 * 512, log=None)
    assert len(hist) == 2 and all(h["episodes"] == 2 * 512 and np.isfinite(h["mean_episode_reward"]) for h in hist)


def _emulate_blob(blob, obs, kind):
    """What the fused kernel computes from the packed blob, restated with numpy from the MFMA
    semantics alone (v_mfma_f32_16x16x4_f32: D[i][j] += sum_g A[i][g] * B[g][j]; operand lane
    l = 16 * g + index; accumulator register r on lane group g holds row 4 g + r; a wave works on 16 envs):
    returns (mean [n, A], value [n], log_std [A]). Independent of the packer's index tables."""
    from tennisbot_rl_amd.params import ACT_DIM, ENV_SWING, OBS_DIM
    hidden = (32, 64, 32) if kind == ENV_SWING else (64, 64)
    n_obs, n_act = OBS_DIM[kind], ACT_DIM[kind]
    blob = np.asarray(blob, dtype=np.float64)
    n_env = obs.shape[0]
    pos, heads = 0, []
    for tower in range(2):
        # B operands of the first layer: chunk c, lane group g -> obs[4c + g] (0 beyond the observation)
        padded = np.concatenate([obs, np.zeros((n_env, -n_obs % 4))], 1)
        x = np.stack([np.stack([padded[:, 4 * c + g] for g in range(4)], 0) for c in range(padded.shape[1] // 4)], 0)  # [chunks][g][env]
        widths = list(hidden) + [16]
        for li, n_out in enumerate(widths):
            n_tiles, n_chunks = n_out // 16, x.shape[0]
            bias = blob[pos:pos + n_tiles * 16].reshape(n_tiles, 4, 4); pos += n_tiles * 16
            frag = blob[pos:pos + n_tiles * n_chunks * 64].reshape(n_tiles, n_chunks, 4, 16); pos += n_tiles * n_chunks * 64
            y = []
            for t in range(n_tiles):
                d = np.einsum("cgi,cge->ie", frag[t], x)  # D[i][env]
                for r in range(4):
                    y.append(np.stack([d[4 * g + r] + bias[t, g, r] for g in range(4)], 0))  # register r: [g][env]
            x = np.stack(y, 0)
            if li < len(hidden):
                x = np.tanh(x)
        heads.append(x)  # [4 regs][g][env] of the head tile
    out = lambda head, i: head[i & 3, i >> 2]  # head row i: register i & 3 of lane group i >> 2
    mean = np.stack([out(heads[0], i) for i in range(n_act)], -1)
    value = out(heads[1], 0)
    log_std = blob[pos:pos + n_act]
    assert pos + (n_act + 3) // 4 * 4 == blob.size
    return mean, value, log_std


@pytest.mark.parametrize("env_id", ["SwingRacket-v0", "Tennisbot-v0"])
def test_pack_policy_matches_the_module_under_mfma_semantics(env_id):
    """host logic of the fused policy step: the fragment-ordered blob, pushed through a numpy
    restatement of the MFMA data flow, reproduces the torch module (float64, 1e-12)"""
    import torch
    from tennisbot_rl_amd.params import ACT_DIM, ENV_SWING, ENV_TENNIS, OBS_DIM
    from tennisbot_rl_amd.ppo import SWING_DEFAULTS, TENNIS_DEFAULTS, build_actor_critic, pack_policy
    kind = ENV_SWING if env_id == "SwingRacket-v0" else ENV_TENNIS
    torch.manual_seed(4)
    arch = (SWING_DEFAULTS if kind == ENV_SWING else TENNIS_DEFAULTS)["net_arch"]
    policy = build_actor_critic(OBS_DIM[kind], ACT_DIM[kind], tuple(arch))
    with torch.no_grad():
        for p in policy.parameters():
            p.copy_(torch.randn_like(p) * 0.3)
    blob = pack_policy(policy).numpy()
    obs = np.random.default_rng(0).normal(size=(32, OBS_DIM[kind])).astype(np.float32)
    mean, value, log_std = _emulate_blob(blob, obs.astype(np.float64), kind)
    with torch.no_grad():
        want_mean, want_value = policy.double()(torch.from_numpy(obs).double())
    np.testing.assert_allclose(mean, want_mean.numpy(), atol=1e-6)   # blob is float32: weights rounded once
    np.testing.assert_allclose(value, want_value.numpy(), atol=1e-6)
    np.testing.assert_allclose(log_std, policy.log_std.detach().numpy(), atol=0)
    # refreshing in place keeps the address (captured graphs bake it in)
    buf = torch.zeros(blob.size)
    assert pack_policy(policy.float(), out=buf).data_ptr() == buf.data_ptr()


def test_reference_episode_record_and_shape_inertia():
    """the fixture extracted from backup_models/ppo_swing.zip (tools/export_reference_episode_stats.py)
    and the Bullet shape-inertia rule that it pins (params.bullet_shape_inertia)"""
    import json
    from tennisbot_rl_amd.params import bullet_shape_inertia, default_params, load_scene, urdf_file_inertia
    rec = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ppo_swing_reference_episodes.json")))
    r = np.array(rec["episode_rewards"])
    assert r.size == 100 and rec["num_timesteps"] == 236000
    # every SwingRacket episode is 26 steps; the record's two 38-step entries are training episodes that an EvalCallback on
    # the TRAINING env (train_swing.py:111, quirk D.11) reset after 12 steps: 1000 = 38 * 26 + 12, eval_freq = 1000, and they
    # are 38 episodes apart (38 * 26 + 12 = 1000 again); the evaluation's 0.86 s show up in exactly their wall-clock gaps
    n, t = np.array(rec["episode_lengths"]), np.array(rec["episode_wallclock_s"])
    cut = np.flatnonzero(n == 38)
    assert set(n) == {26, 38} and list(cut) == [24, 62] and np.all(np.diff(t)[cut - 1] > 5 * np.median(np.diff(t)))
    assert abs(r.mean() - 31.5256) < 1e-3 and int((r >= 50).sum()) == 27  # 27 goal hits in the last 100 PyBullet episodes
    sc = load_scene()
    got = bullet_shape_inertia()
    ext = [hi - lo + 0.002 for lo, hi in zip(sc["racket"]["bbox_min"], sc["racket"]["bbox_max"])]
    want = [4.0 / 12 * (ext[1] ** 2 + ext[2] ** 2), 4.0 / 12 * (ext[0] ** 2 + ext[2] ** 2), 4.0 / 12 * (ext[0] ** 2 + ext[1] ** 2)]
    assert np.allclose(got["racket_inertia"], want, rtol=1e-12) and np.allclose(want, (0.19366, 0.16326, 0.03104), atol=1e-5)
    assert got["ball_inertia"] == pytest.approx(0.4 * 0.05 * 0.0335 ** 2)
    p = default_params()
    assert np.allclose(list(p.racket_inertia), want, rtol=1e-6) and p.ball_inv_inertia == pytest.approx(1.0 / got["ball_inertia"], rel=1e-6)
    pu = default_params(**urdf_file_inertia())
    assert np.allclose(list(pu.racket_inertia), (0.04, 0.08, 0.12), rtol=1e-6) and pu.ball_inv_inertia == pytest.approx(1.0)

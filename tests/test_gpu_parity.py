"""Parity of the HIP path (through the C ABI) with the CPU oracle on a real MI355X.

Bar (BASELINE.json north_star): done flags and step/substep counters bit-exact; floats
within a stated float32 tolerance. The float32 oracle build follows the same operation
order as the kernels, so these tests assert BIT equality for floats too and would fall
back to the stated tolerance only through TOL below (kept at 0 while it holds).
The float64 build measures how far float32 drifts from the "true" trajectory.
"""
import numpy as np
import pytest

from helpers import make_words
from oracle import OracleBatch
from outlines import with_outline
from tennisbot_rl_amd.params import (ENV_SWING, ENV_TENNIS, F_AUTO_RESET, F_DEFAULT, F_NET, F_RACKET_GROUND, STATE_WORDS, default_params,
                                     reference_rolling_friction)

pytestmark = pytest.mark.gpu

TOL = 0.0  # float tolerance HIP vs float32 oracle: bit-exact


@pytest.fixture(scope="module")
def torch():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch


def make_pair(torch, kind, n, seed=11, auto_reset=True, env_id_base=0, options=None, **over):
    from tennisbot_rl_amd.stepper import BatchedEnv
    flags = over.pop("flags", F_DEFAULT)
    p = default_params(flags=flags, **over)
    env = BatchedEnv(kind, n, device="cuda:0", seed=seed, env_id_base=env_id_base, params=p, auto_reset=auto_reset, options=options)
    pf = p.copy()
    pf.flags = (pf.flags | F_AUTO_RESET) if auto_reset else (pf.flags & ~F_AUTO_RESET)
    ref = OracleBatch(pf, kind, n, seed=seed, env_id_base=env_id_base, precision="f32")
    return env, ref


def same(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    if TOL == 0.0:
        if a.dtype.kind == "f":
            ok = np.array_equal(a.view(np.uint32), b.view(np.uint32)) or np.array_equal(a, b)
        else:
            ok = np.array_equal(a, b)
        if not ok:
            bad = np.argwhere(a != b)
            raise AssertionError("%s: %d mismatches, first at %s: %r vs %r" % (what, len(bad), bad[0], a[tuple(bad[0])], b[tuple(bad[0])]))
    else:
        np.testing.assert_allclose(a, b, rtol=TOL, atol=TOL, err_msg=what)


def compare_state(env, ref, what):
    w_gpu, d_gpu = env.get_state_words()
    w_cpu, d_cpu = ref.get_state_words()
    w_gpu = w_gpu.cpu().numpy().view(np.uint32)
    nw = STATE_WORDS[env.kind]
    same(w_gpu[nw - 2:], w_cpu[nw - 2:], what + " step_count/episode")  # integers: always exact
    same(d_gpu.cpu().numpy(), d_cpu, what + " done byte")
    same(w_gpu[: nw - 2].view(np.float32), w_cpu[: nw - 2].view(np.float32), what + " float state")


def run_lockstep(torch, env, ref, steps, rng, what, check_state_every=1):
    n = env.num_envs
    same(env.reset().cpu().numpy(), ref.reset(), what + " reset obs")
    compare_state(env, ref, what + " after reset")
    for t in range(steps):
        a = rng.uniform(-1, 1, (n, env.act_dim)).astype(np.float32)
        obs, rew, done = env.step(torch.from_numpy(a).cuda())
        o2, r2, d2, s2, t2 = ref.step(a, want_terminal=True)
        tag = "%s step %d" % (what, t)
        same(done.cpu().numpy(), d2, tag + " done")
        same(env.last_substeps().cpu().numpy(), s2, tag + " substeps")
        same(obs.cpu().numpy(), o2, tag + " obs")
        same(rew.cpu().numpy(), r2, tag + " reward")
        if env.auto_reset and d2.any():
            m = d2.astype(bool)
            same(env.terminal_obs().cpu().numpy()[m], t2[m], tag + " terminal obs")
        if t % check_state_every == 0:
            compare_state(env, ref, tag)
    got, want = env.counters(), ref.counters()
    assert list(got.values()) == [int(x) for x in want], (got, want)


def test_native_library_is_the_one_running(torch):
    from tennisbot_rl_amd import stepper
    L = stepper.load_library()
    assert L.tb_abi_version() == 4
    with open("/proc/self/maps") as f:
        assert "libtb_stepper.so" in f.read()


@pytest.mark.parametrize("n", [1, 63, 64, 65, 257, 4096])
def test_swing_lockstep_ragged_sizes(torch, n):
    env, ref = make_pair(torch, ENV_SWING, n)
    run_lockstep(torch, env, ref, 54, np.random.default_rng(n), "swing n=%d" % n)  # two full episodes + resets
    env.close()


@pytest.mark.parametrize("n,reg_rows", [(1, True), (65, True), (4096, True), (65, False), (4096, False)])
def test_tennis_lockstep(torch, n, reg_rows):
    # both builds of the Tennisbot step kernel (TbOptions.tennis_reg_rows): static contact rows in registers / in scratch
    env, ref = make_pair(torch, ENV_TENNIS, n, options=dict(tennis_reg_rows=reg_rows))
    # 1010 steps: past the 1000-step timeout (tennisbot_env.py:201-203), so every env finishes at least once
    run_lockstep(torch, env, ref, 1010, np.random.default_rng(100 + n), "tennis n=%d" % n, check_state_every=50)
    c = env.counters()
    assert c["episodes_finished"] > 0 and c["nonfinite_states"] == 0
    env.close()


def test_baseline_full_sizes_in_lockstep_with_the_oracle(torch):
    """BASELINE.json configs[3] and [4] at their full global sizes on one GPU (the env ids of all 8
    shards): SwingRacket-v0 32 768 envs through a whole episode with its fast-forward, Tennisbot-v0
    65 536 envs with the Magnus / random-spin extension -- every output and the full state bit-exact
    against the oracle (OpenMP over envs so that the CPU side stays at seconds)."""
    import os
    threads = max(1, min(16, len(os.sched_getaffinity(0))))
    from tennisbot_rl_amd.stepper import BatchedEnv
    p = default_params()
    env = BatchedEnv(ENV_SWING, 32768, device="cuda:0", seed=3, params=p)
    pf = p.copy(); pf.flags |= F_AUTO_RESET
    ref = OracleBatch(pf, ENV_SWING, 32768, seed=3, precision="f32", threads=threads)
    run_lockstep(torch, env, ref, 28, np.random.default_rng(77), "swing 32768", check_state_every=9)
    assert env.counters()["episodes_finished"] == 32768
    env.close()
    p = default_params(magnus_k=2e-4, ball_spin_max=150.0)
    env = BatchedEnv(ENV_TENNIS, 65536, device="cuda:0", seed=4, params=p)
    pf = p.copy(); pf.flags |= F_AUTO_RESET
    ref = OracleBatch(pf, ENV_TENNIS, 65536, seed=4, precision="f32", threads=threads)
    run_lockstep(torch, env, ref, 120, np.random.default_rng(78), "tennis 65536", check_state_every=40)
    env.close()


def test_swing_without_auto_reset_sticky_done(torch):
    env, ref = make_pair(torch, ENV_SWING, 300, auto_reset=False)
    run_lockstep(torch, env, ref, 30, np.random.default_rng(5), "swing sticky")  # 4 steps past done
    assert env.get_state()["done"].min() == 2  # pending force consumed once, then plain done
    env.close()


def test_tennis_without_auto_reset(torch):
    env, ref = make_pair(torch, ENV_TENNIS, 300, auto_reset=False)
    run_lockstep(torch, env, ref, 400, np.random.default_rng(6), "tennis sticky", check_state_every=25)
    env.close()


@pytest.mark.parametrize("n", [1024, 4096])
def test_contact_off_bench_mode(torch, n):
    """BASELINE configs[1]: racket-only dynamics, racket<->ball pair disabled -- at the config's own 4096 envs too"""
    env, ref = make_pair(torch, ENV_SWING, n, flags=F_NET)
    run_lockstep(torch, env, ref, 27, np.random.default_rng(7), "swing contact-off n=%d" % n)
    assert env.counters()["racket_ball_contact_substeps"] == 0
    env.close()


def test_forced_contacts_exercise_the_solver(torch):
    """random policies rarely hit the ball; inject states where every lane is in contact
    (racket face, rim, ground, net, goal) so that the narrowphase sweep + impulse solver
    are compared lane by lane"""
    n = 512
    rng = np.random.default_rng(21)
    p = default_params()
    env, ref = make_pair(torch, ENV_TENNIS, n, auto_reset=False)
    rp = np.stack([rng.uniform(8, 12, n), rng.uniform(-4, 4, n), rng.uniform(0.7, 1.5, n)], 1)
    # random racket orientation, ball placed just outside the +-x face at a random spot of the outline
    q = rng.normal(size=(n, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    side = np.where(rng.random(n) < 0.5, -1.0, 1.0)
    loc = np.stack([side * (p.racket_half_thick + p.hull_margin + p.ball_radius + rng.uniform(-0.004, 0.0006, n)),
                    rng.uniform(-0.16, 0.16, n), rng.uniform(-0.1, 0.22, n)], 1)
    def rot(q, v):
        u, w = q[:, :3], q[:, 3:4]
        t = 2 * np.cross(u, v)
        return v + w * t + np.cross(u, t)
    bp = rp + rot(q, loc)
    bv = rot(q, np.stack([-side * rng.uniform(1, 25, n), rng.uniform(-5, 5, n), rng.uniform(-5, 5, n)], 1))
    w, d = make_words(ENV_TENNIS, n, racket_pos=rp, racket_quat=q, racket_vel=rng.uniform(-3, 3, (n, 3)),
                      racket_angvel=rng.uniform(-4, 4, (n, 3)), ball_pos=bp, ball_vel=bv, ball_angvel=rng.uniform(-20, 20, (n, 3)),
                      shoot_force=(30, 0, 20), step_count=50)
    env.set_state_words(w.view(np.int32), d); ref.set_state_words(w, d)
    for t in range(6):
        a = rng.uniform(-1, 1, (n, 2)).astype(np.float32)
        obs, rew, done = env.step(torch.from_numpy(a).cuda())
        o2, r2, d2, s2 = ref.step(a)
        same(obs.cpu().numpy(), o2, "forced racket contact obs %d" % t)
        same(rew.cpu().numpy(), r2, "forced racket contact reward %d" % t)
        compare_state(env, ref, "forced racket contact %d" % t)
    assert env.counters()["racket_ball_contact_substeps"] > n // 2
    env.close()

    # statics: ground, net, goal (Swing has all three)
    env, ref = make_pair(torch, ENV_SWING, n, auto_reset=False)
    which = rng.integers(0, 3, n)
    goal = np.stack([rng.uniform(-11, -4, n), rng.uniform(-4, 4, n)], 1)
    r = p.ball_radius
    ground = np.stack([rng.uniform(-13.9, 14.2, n), rng.uniform(-6.9, 7.1, n), 0.005 + r + rng.uniform(-0.003, 0.0006, n)], 1)
    net = np.stack([np.where(rng.random(n) < 0.5, -1, 1) * (p.net_half[0] + r + rng.uniform(-0.003, 0.0006, n)), rng.uniform(-6.4, 6.4, n), rng.uniform(0.05, 0.56, n)], 1)
    ang = rng.uniform(0, 2 * np.pi, n); rad = rng.uniform(0, 1.55, n)
    gl = np.stack([goal[:, 0] + rad * np.cos(ang), goal[:, 1] + rad * np.sin(ang), 0.125 + r + rng.uniform(-0.003, 0.0006, n)], 1)
    bp = np.where(which[:, None] == 0, ground, np.where(which[:, None] == 1, net, gl))
    bv = np.stack([rng.uniform(-12, 12, n), rng.uniform(-6, 6, n), rng.uniform(-9, 1, n)], 1)
    w, d = make_words(ENV_SWING, n, racket_pos=(9, 0, 1.0), ball_pos=bp, ball_vel=bv, ball_angvel=rng.uniform(-30, 30, (n, 3)),
                      goal=goal, spawn_pos=(9, 0, 0.6), init_dist=rng.uniform(8, 20, n), step_count=rng.integers(0, 30, n))
    env.set_state_words(w.view(np.int32), d); ref.set_state_words(w, d)
    for t in range(4):
        a = rng.uniform(-1, 1, (n, 6)).astype(np.float32)
        obs, rew, done = env.step(torch.from_numpy(a).cuda())
        o2, r2, d2, s2 = ref.step(a)
        same(done.cpu().numpy(), d2, "forced statics done %d" % t)
        same(env.last_substeps().cpu().numpy(), s2, "forced statics substeps %d" % t)
        same(obs.cpu().numpy(), o2, "forced statics obs %d" % t)
        same(rew.cpu().numpy(), r2, "forced statics reward %d" % t)
        compare_state(env, ref, "forced statics %d" % t)
    c = env.counters()
    assert c["ball_court_terminations"] > 0 and c["goal_hits"] > 0
    env.close()


@pytest.mark.parametrize("kind,rg,roll", [(ENV_TENNIS, False, False), (ENV_SWING, False, False), (ENV_TENNIS, True, False), (ENV_SWING, True, False),
                                          (ENV_TENNIS, False, True), (ENV_SWING, False, True), (ENV_TENNIS, True, True)])
def test_fuzzed_states_around_the_racket_stay_bit_exact(torch, kind, rg, roll):
    """16 384 random states with the ball anywhere in a thin shell around a randomly oriented (Tennisbot:
    randomly scaled) racket that itself hovers close to the court -- faces, rim, handle, corners, grazing
    and deep overlaps, ball on racket AND ground at once -- stepped 8 times: every output and the whole
    state bit-exact against the oracle. The outline sweep's inside / edge / corner cases and the
    multi-row solver only occur by accident in the policy-driven tests. rg: with the opt-in
    racket<->court contact as well (racket, ball and court in one solve); roll: with the opt-in
    rolling-friction rows (TbParams.roll_*) in every ball contact."""
    n = 2048 if rg else 16384
    rng = np.random.default_rng(97 + kind + 10 * rg + 100 * roll)
    p = default_params()
    over = reference_rolling_friction() if roll else {}
    env, ref = make_pair(torch, kind, n, auto_reset=False, flags=F_DEFAULT | (F_RACKET_GROUND if rg else 0), **over)
    ref.L.tbo_set_threads(ref.h, 8)
    scale = rng.uniform(1.0, 3.0, n) if kind == ENV_TENNIS else np.ones(n)
    q = rng.normal(size=(n, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    rp = np.stack([rng.uniform(8, 12, n), rng.uniform(-4, 4, n), rng.uniform(0.05, 1.2, n) * scale], 1)
    if rg:  # keep every hull vertex above the court (the stateless manifold ignores vertices below it): COM at least its reach up
        rp[:, 2] = rng.uniform(0.52, 0.75, n) * scale
    ht, r = float(p.racket_half_thick), float(p.ball_radius)
    pad = r + 0.004
    loc = np.stack([rng.uniform(-(ht * scale + pad), ht * scale + pad), rng.uniform(-0.16 * scale - pad, 0.16 * scale + pad),
                    rng.uniform(-0.51 * scale - pad, 0.21 * scale + pad)], 1)
    # push two thirds of the samples onto the shell (the rest stay inside: deep overlaps)
    k = rng.integers(0, 3, n); sgn = np.where(rng.random(n) < 0.5, -1.0, 1.0); shell = rng.random(n) < 0.67
    ext = np.stack([ht * scale, 0.15 * scale, np.where(sgn > 0, 0.197, 0.5) * scale], 1)
    idx = np.arange(n)
    loc[idx[shell], k[shell]] = (sgn * (ext[idx, k] + r + rng.uniform(-0.004, 0.001, n)))[shell]

    def rot(q, v):
        u, w = q[:, :3], q[:, 3:4]
        t = 2 * np.cross(u, v)
        return v + w * t + np.cross(u, t)
    bp = rp + rot(q, loc)
    bp[:, 2] = np.maximum(bp[:, 2], 0.005 + r - 0.003)  # never (much) under the court
    fields = dict(racket_pos=rp, racket_quat=q, racket_vel=rng.uniform(-4, 4, (n, 3)), racket_angvel=rng.uniform(-6, 6, (n, 3)),
                  ball_pos=bp, ball_vel=rng.uniform(-15, 15, (n, 3)), ball_angvel=rng.uniform(-60, 60, (n, 3)))
    if kind == ENV_TENNIS:
        fields.update(shoot_force=(30, 0, 20), step_count=50, racket_scale=scale)
        A = 2
    else:
        fields.update(goal=np.stack([rng.uniform(-11, -4, n), rng.uniform(-4, 4, n)], 1), spawn_pos=(9, 0, 0.6), init_dist=rng.uniform(8, 20, n),
                      step_count=rng.integers(0, 24, n))
        A = 6
    w, d = make_words(kind, n, **fields)
    env.set_state_words(w.view(np.int32), d); ref.set_state_words(w, d)
    plain = None
    if roll:  # the same states without the rolling rows: they must end up elsewhere
        plain, _ = make_pair(torch, kind, n, auto_reset=False, flags=F_DEFAULT | (F_RACKET_GROUND if rg else 0))
        plain.set_state_words(w.view(np.int32), d)
    for t in range(4 if rg else 8):
        a = rng.uniform(-1, 1, (n, A)).astype(np.float32)
        if plain is not None:
            plain.step(torch.from_numpy(a).cuda())
        obs, rew, done = env.step(torch.from_numpy(a).cuda())
        o2, r2, d2, s2 = ref.step(a)
        same(done.cpu().numpy(), d2, "fuzz done %d" % t)
        same(env.last_substeps().cpu().numpy(), s2, "fuzz substeps %d" % t)
        same(obs.cpu().numpy(), o2, "fuzz obs %d" % t)
        same(rew.cpu().numpy(), r2, "fuzz reward %d" % t)
        compare_state(env, ref, "fuzz %d" % t)
    c = env.counters()
    assert c["racket_ball_contact_substeps"] > n // 8 and c["nonfinite_states"] == 0
    if plain is not None:
        differ = (env.get_state_words()[0] != plain.get_state_words()[0]).any(0)
        assert int(differ.sum()) > n // 8
        plain.close()
    env.close()


@pytest.mark.parametrize("kind,n,steps", [(ENV_SWING, 4096, 46), (ENV_TENNIS, 1000, 700)])
def test_rolling_friction_rows_in_lockstep_and_not_a_no_op(torch, kind, n, steps):
    """whole episodes with the opt-in rolling-friction rows: bit-exact against the oracle, the
    same through the pipelined fast-forward kernel on SwingRacket, and the rows do change what balls do"""
    from tennisbot_rl_amd.stepper import BatchedEnv, StepperError
    roll = reference_rolling_friction()
    env, ref = make_pair(torch, kind, n, **roll)
    ref.L.tbo_set_threads(ref.h, 8)
    run_lockstep(torch, env, ref, steps, np.random.default_rng(5), "rolling kind %d" % kind, check_state_every=13)
    rng = np.random.default_rng(5)
    acts = torch.from_numpy(rng.uniform(-1, 1, (steps, n, env.act_dim)).astype(np.float32)).cuda()
    # a second pass next to an env without the rows; on SwingRacket through tb_ff_kernel<extended> on the side streams
    plain, _ = make_pair(torch, kind, n)
    again = BatchedEnv(kind, n, device="cuda:0", seed=11, params=default_params(**roll), pipeline=kind == ENV_SWING)
    plain.reset(); again.reset()
    differ = torch.zeros(n, dtype=torch.bool, device="cuda")
    for t in range(steps):
        differ |= (again.step(acts[t])[0] != plain.step(acts[t])[0]).any(1)
    again.flush()
    assert torch.equal(env.get_state_words()[0], again.get_state_words()[0]) and env.counters() == again.counters()
    if kind == ENV_TENNIS:  # balls that bounce on the court before they reach the racket; SwingRacket's contacts are
        assert int(differ.sum()) > 0  # rare under random actions (the fuzz test forces them)
    others = [plain, again]
    # (until round 3 the fused policy kernels refused the extended contact set; now they are instantiated for it:
    #  tests/test_gpu_policy.py holds their parity tests)
    (o, _, _), _ = env.policy_step(torch.zeros(env.policy_floats(), device="cuda"), env.reset())
    assert bool(torch.isfinite(o).all())
    env.close()
    for e in others:
        e.close()


def test_rollout_equals_repeated_steps(torch):
    from tennisbot_rl_amd.stepper import BatchedEnv
    n, T = 1000, 60
    rng = np.random.default_rng(3)
    for kind, A in ((ENV_SWING, 6), (ENV_TENNIS, 2)):
        acts = torch.from_numpy(rng.uniform(-1, 1, (T, n, A)).astype(np.float32)).cuda()
        e1 = BatchedEnv(kind, n, seed=4)
        e2 = BatchedEnv(kind, n, seed=4)
        e1.reset(); e2.reset()
        obs, rew, done = e1.rollout(acts)
        sub_total = e1.last_substeps().clone()
        acc = torch.zeros_like(sub_total)
        for t in range(T):
            o, r, d = e2.step(acts[t])
            acc += e2.last_substeps()
            assert torch.equal(o, obs[t]) and torch.equal(r, rew[t]) and torch.equal(d, done[t])
        assert torch.equal(acc, sub_total)
        w1, d1 = e1.get_state_words(); w2, d2 = e2.get_state_words()
        assert torch.equal(w1, w2) and torch.equal(d1, d2)
        e1.close(); e2.close()


# The pipeline's two forms: ff_defer=False -- one fast-forward kernel per episode end, on a side stream ("slots"); None -- the automatic
# choice, which up to 16384 envs parks every episode end into the pool and runs ONE fast-forward at the join (TbOptions.ff_defer = 2)
BOTH_PIPELINES = pytest.mark.parametrize("defer", [False, None], ids=["slots", "auto"])


@BOTH_PIPELINES
def test_pipelined_rollout_equals_repeated_steps(torch, defer):
    """tb_rollout with the pipeline on: launches that end where the episodes end (<= 26 steps each),
    fast-forwards on the side streams or at the join -- from any starting phase, also captured in a hipGraph"""
    from tennisbot_rl_amd.stepper import BatchedEnv
    n, T = 1000, 75
    rng = np.random.default_rng(13)
    acts = torch.from_numpy(rng.uniform(-1, 1, (T + 9, n, 6)).astype(np.float32)).cuda()
    e1 = BatchedEnv(ENV_SWING, n, seed=4, pipeline=True, track_terminal_obs=False, options=dict(ff_defer=defer))
    e2 = BatchedEnv(ENV_SWING, n, seed=4)
    e1.reset(); e2.reset()
    for t in range(9):  # start the rollout in the middle of an episode
        e1.step(acts[T + t]); e2.step(acts[T + t])
    obs, rew, done = e1.rollout(acts[:T])
    e1.flush()
    for t in range(T):
        o, r, d = e2.step(acts[t])
        assert torch.equal(o, obs[t]) and torch.equal(d, done[t]), "step %d" % t
        assert torch.equal(r, rew[t]), "reward, step %d" % t
    assert int(done.sum()) == 3 * n  # the episode ends at rollout steps 16, 42, 68
    # and as a graph: same buffers, replayed twice
    a_static = acts[:52].clone()
    out = {}

    def body():
        out["r"] = e1.rollout(a_static)
    g = e1.capture(body)
    for rnd in range(2):
        g.replay()
        torch.cuda.synchronize()
        for t in range(52):
            o, r, d = e2.step(a_static[t])
            assert torch.equal(o, out["r"][0][t]) and torch.equal(r, out["r"][1][t]) and torch.equal(d, out["r"][2][t]), "graph round %d step %d" % (rnd, t)
    w1, d1 = e1.get_state_words(); w2, d2 = e2.get_state_words()
    assert torch.equal(w1, w2) and torch.equal(d1, d2)
    assert e1.counters() == e2.counters()
    e1.close(); e2.close()


def test_sharding_independence(torch):
    """RNG keyed by the GLOBAL env id: two half batches == one whole batch"""
    from tennisbot_rl_amd.stepper import BatchedEnv
    n = 512
    rng = np.random.default_rng(8)
    acts = torch.from_numpy(rng.uniform(-1, 1, (30, n, 6)).astype(np.float32)).cuda()
    whole = BatchedEnv(ENV_SWING, n, seed=5)
    lo = BatchedEnv(ENV_SWING, n // 2, seed=5, env_id_base=0)
    hi = BatchedEnv(ENV_SWING, n // 2, seed=5, env_id_base=n // 2)
    a = whole.reset(); b = torch.cat([lo.reset(), hi.reset()])
    assert torch.equal(a, b)
    for t in range(30):
        o, r, d = whole.step(acts[t])
        o1, r1, d1 = lo.step(acts[t, : n // 2].contiguous())
        o2, r2, d2 = hi.step(acts[t, n // 2:].contiguous())
        assert torch.equal(o, torch.cat([o1, o2])) and torch.equal(r, torch.cat([r1, r2])) and torch.equal(d, torch.cat([d1, d2]))
    for e in (whole, lo, hi):
        e.close()


def test_masked_reset_and_state_roundtrip(torch):
    from tennisbot_rl_amd.stepper import BatchedEnv
    n = 200
    env, ref = make_pair(torch, ENV_TENNIS, n, auto_reset=False)
    env.reset(); ref.reset()
    mask = (np.arange(n) % 3 == 0).astype(np.uint8)
    o1 = env.reset(torch.from_numpy(mask).cuda())
    o2 = ref.reset(mask)
    same(o1.cpu().numpy()[mask.astype(bool)], o2[mask.astype(bool)], "masked reset obs")
    compare_state(env, ref, "masked reset")
    st = env.get_state()
    assert set(np.unique(st["episode"])) == {0, 1}
    w, d = env.get_state_words()
    other = BatchedEnv(ENV_TENNIS, n, seed=999, auto_reset=False)
    other.set_state_words(w, d)
    a = torch.zeros((n, 2), device="cuda")
    x = env.step(a); y = other.step(a)
    assert all(torch.equal(p, q) for p, q in zip(x, y))
    env.close(); other.close()


def test_scaled_racket_and_spin_extensions(torch):
    """curriculum racket scale (tennisbot_env.py:213-215) and the configs[4] extensions
    (Magnus term, random spin) stay in lockstep with the oracle"""
    env, ref = make_pair(torch, ENV_TENNIS, 512, magnus_k=2e-4, ball_spin_max=150.0)
    run_lockstep(torch, env, ref, 300, np.random.default_rng(31), "tennis magnus", check_state_every=30)
    env.close()
    from tennisbot_rl_amd.stepper import BatchedEnv
    p = default_params(racket_scale=2.3)
    env = BatchedEnv(ENV_TENNIS, 512, seed=2, params=p)
    pf = p.copy(); pf.flags |= F_AUTO_RESET
    ref = OracleBatch(pf, ENV_TENNIS, 512, seed=2, precision="f32")
    run_lockstep(torch, env, ref, 300, np.random.default_rng(32), "tennis scale 2.3", check_state_every=30)
    env.close()


@pytest.mark.parametrize("kind", [ENV_SWING, ENV_TENNIS])
def test_randomised_engine_parameters_stay_bit_exact(torch, kind):
    """every engine constant recalled from Bullet is a TbParams field: six random draws of all
    of them at once -- gravity, damping, restitution / friction / rolling coefficients, ERP, thresholds, masses,
    inertia, solver cap and tolerance, Magnus and spin extensions, racket scale -- and the HIP path still follows the
    oracle bit for bit through whole episodes, forced contacts included (the ball is dropped onto the racket)"""
    rng = np.random.default_rng(1234 + kind)
    n = 1024
    for trial in range(6):
        over = dict(
            gravity=rng.uniform(3.0, 15.0), lin_damp=rng.uniform(0.0, 0.1), ang_damp=rng.uniform(0.0, 0.1),
            max_ang_step=rng.uniform(0.3, 1.2), rest_vel_threshold=rng.uniform(0.0, 1.0), erp=rng.uniform(0.02, 0.4),
            contact_threshold=rng.uniform(2e-4, 3e-3), solver_iters=int(rng.integers(4, 80)), solver_tol=10.0 ** rng.uniform(-7, -4),
            racket_mass=rng.uniform(1.0, 8.0), racket_inertia=tuple(rng.uniform(0.02, 0.3, 3)), ball_mass=rng.uniform(0.03, 0.2),
            ball_inertia=10.0 ** rng.uniform(-5, -3), rest_racket=rng.uniform(0.0, 1.0), rest_court=rng.uniform(0.0, 1.0),
            rest_goal=rng.uniform(0.0, 0.9), fric_racket=rng.uniform(0.0, 0.8), fric_court=rng.uniform(0.0, 0.8), fric_goal=rng.uniform(0.0, 0.8),
            magnus_k=rng.choice([0.0, 1e-4, 5e-4]), ball_spin_max=rng.choice([0.0, 50.0, 200.0]),
            lin_damp_quad=rng.uniform(0.0, 0.1), ang_damp_quad=rng.uniform(0.0, 0.1))
        if trial % 2:
            over.update(roll_racket=rng.uniform(0, 2e-3), roll_court=rng.uniform(0, 2e-3), roll_goal=rng.uniform(0, 2e-3))
        scale = float(rng.uniform(1.0, 3.0)) if kind == ENV_TENNIS else 1.0
        env, ref = make_pair(torch, kind, n, seed=100 + trial, racket_scale=scale, **over)
        ref.L.tbo_set_threads(ref.h, 8)
        what = "params trial %d kind %d" % (trial, kind)
        run_lockstep(torch, env, ref, 60 if kind == ENV_SWING else 200, rng, what, check_state_every=20)
        # ... and from states in which the ball sits on the racket face (a solve on the first step)
        w, d = env.get_state_words()
        w = w.cpu().numpy().copy().view(np.uint32)
        f = w.view(np.float32)
        rp = f[0:3].copy()  # racket COM rows
        f[13:16] = rp + np.array([[-(float(env.params.racket_half_thick) * scale + float(env.params.ball_radius))], [0.0], [0.0]], np.float32)
        f[16:19] = np.array([[4.0], [0.5], [-0.5]], np.float32)
        dn = d.cpu().numpy()
        env.set_state_words(w.view(np.int32), dn)
        ref.set_state_words(w, dn)
        for t in range(12):
            a = rng.uniform(-1, 1, (n, env.act_dim)).astype(np.float32)
            obs, rew, done = env.step(torch.from_numpy(a).cuda())
            o2, r2, d2, s2 = ref.step(a)
            same(obs.cpu().numpy(), o2, what + " contact obs %d" % t)
            same(rew.cpu().numpy(), r2, what + " contact reward %d" % t)
            same(done.cpu().numpy(), d2, what + " contact done %d" % t)
        compare_state(env, ref, what + " after contacts")
        assert env.counters()["racket_ball_contact_substeps"] > 0
        env.close()


@pytest.mark.parametrize("kind,steps,piped,options,flags", [
    (ENV_SWING, 30, False, None, F_DEFAULT), (ENV_SWING, 30, True, None, F_DEFAULT), (ENV_TENNIS, 120, False, None, F_DEFAULT),
    (ENV_SWING, 30, True, dict(ff_phases=3), F_DEFAULT), (ENV_SWING, 30, True, dict(ff_phases=2), F_DEFAULT | F_RACKET_GROUND),
    (ENV_SWING, 30, True, dict(ff_phases=3, ff_sort=True, ff_lanes_per_wave=16), F_DEFAULT),
    (ENV_SWING, 30, True, dict(swing_reg_rows=False), F_DEFAULT)])  # (the step kernel with its static rows in LDS: the automatic choice until round 3's build flags)
def test_large_batch_instantiations_in_lockstep_with_the_oracle(torch, kind, steps, piped, options, flags):
    """above 131 072 envs tb_create picks other launch shapes and kernel variants (128-thread workgroups, the fast-forward
    instantiation that re-reads its cull planes and shares the outline sweep): 200 003 envs -- ragged against every workgroup
    size -- against the oracle, through a whole SwingRacket episode end (in-kernel and side-stream fast-forward) and
    Tennisbot's first arrivals at the racket. With ff_phases > 1 (the default from 256 K envs on) the first phase kernel also
    hands every env whose ball reaches the racket to the next one BEFORE that substep (substep<ESC>), with and without
    racket<->court contact."""
    from tennisbot_rl_amd.rollout import RolloutBuffer
    from tennisbot_rl_amd.stepper import BatchedEnv
    n = 200003
    p = default_params(flags=flags)
    env = BatchedEnv(kind, n, device="cuda:0", seed=21, params=p, pipeline=piped, track_terminal_obs=False, options=options)
    pf = p.copy(); pf.flags |= F_AUTO_RESET
    ref = OracleBatch(pf, kind, n, seed=21, precision="f32")
    ref.L.tbo_set_threads(ref.h, 16)
    rng = np.random.default_rng(9)
    same(env.reset().cpu().numpy(), ref.reset(), "large reset obs")
    outs = []
    for t in range(steps):
        a = rng.uniform(-1, 1, (n, env.act_dim)).astype(np.float32)
        obs, rew, done = env.step(torch.from_numpy(a).cuda())
        o2, r2, d2, s2 = ref.step(a)
        same(obs.cpu().numpy(), o2, "large obs %d" % t)
        same(done.cpu().numpy(), d2, "large done %d" % t)
        if piped:
            outs.append((rew, r2))  # terminal rewards arrive late from the side streams
        else:
            same(rew.cpu().numpy(), r2, "large reward %d" % t)
    env.flush()
    for t, (rew, r2) in enumerate(outs):
        same(rew.cpu().numpy(), r2, "large reward %d" % t)
    compare_state(env, ref, "large final state")
    got, want = env.counters(), ref.counters()
    assert list(got.values()) == [int(x) for x in want], (got, want)
    assert got["nonfinite_states"] == 0 and got["substeps"] >= n * steps
    env.close()


def test_first_phase_hands_over_what_it_cannot_hold(torch):
    """the large-batch fast-forward's first phase kernel (substep<ESC>) keeps only two static contact rows (ground; net OR goal) and
    no racket row at all: an env whose ball is near both the net and a goal, or reaches the racket, must leave for the next phase
    kernel BEFORE that substep. 131 075 envs one agent step before their episode end, the balls low beside the net with the goal
    disc right there, on the ground, on goals elsewhere, and next to the falling racket: rewards, obs and the states after the
    restart bit-exact against the oracle."""
    from tennisbot_rl_amd.stepper import BatchedEnv
    n = 131075
    rng = np.random.default_rng(31)
    p = default_params()
    env = BatchedEnv(ENV_SWING, n, device="cuda:0", seed=3, params=p, pipeline=True, track_terminal_obs=False, options=dict(ff_phases=2))
    pf = p.copy(); pf.flags |= F_AUTO_RESET
    ref = OracleBatch(pf, ENV_SWING, n, seed=3, precision="f32")
    ref.L.tbo_set_threads(ref.h, 16)
    env.reset(); ref.reset()
    r = float(p.ball_radius)
    which = rng.integers(0, 4, n)
    side = np.where(rng.random(n) < 0.5, -1.0, 1.0)
    at_net = np.stack([side * (p.net_half[0] + r + rng.uniform(0.0, 0.08, n)), rng.uniform(-6, 6, n), r + rng.uniform(0.0, 0.2, n)], 1)
    goal_at_net = np.stack([at_net[:, 0] + rng.uniform(-0.6, 0.6, n), at_net[:, 1] + rng.uniform(-0.6, 0.6, n)], 1)
    goal_far = np.stack([rng.uniform(-11, -4, n), rng.uniform(-4, 4, n)], 1)
    ground = np.stack([rng.uniform(-13, 13), rng.uniform(-6, 6), 0.0])[None, :] + np.stack([rng.uniform(-1, 1, n), rng.uniform(-1, 1, n), r + rng.uniform(0.0, 0.3, n)], 1)
    ang = rng.uniform(0, 2 * np.pi, n); rad = rng.uniform(0, 1.4, n)
    on_goal = np.stack([goal_far[:, 0] + rad * np.cos(ang), goal_far[:, 1] + rad * np.sin(ang), 0.125 + r + rng.uniform(0.0, 0.3, n)], 1)
    rp = np.stack([rng.uniform(7, 12, n), rng.uniform(-4, 4, n), rng.uniform(0.8, 1.6, n)], 1)
    by_racket = rp + np.stack([rng.uniform(-0.08, 0.08, n), rng.uniform(-0.15, 0.15, n), rng.uniform(-0.1, 0.6, n)], 1)
    bp = np.where(which[:, None] == 0, at_net, np.where(which[:, None] == 1, ground, np.where(which[:, None] == 2, on_goal, by_racket)))
    goal = np.where(which[:, None] == 0, goal_at_net, goal_far)
    bv = np.stack([rng.uniform(-3, 3, n) - 2.0 * side * (which == 0), rng.uniform(-2, 2, n), rng.uniform(-3, 0.5, n)], 1)
    q = rng.normal(size=(n, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, d = make_words(ENV_SWING, n, racket_pos=rp, racket_quat=q, racket_vel=rng.uniform(-2, 2, (n, 3)), racket_angvel=rng.uniform(-4, 4, (n, 3)),
                      ball_pos=bp, ball_vel=bv, ball_angvel=rng.uniform(-20, 20, (n, 3)), goal=goal, spawn_pos=rp, init_dist=rng.uniform(8, 20, n), step_count=25)
    env.set_state_words(w.view(np.int32), d); ref.set_state_words(w, d)
    assert env.phase() == 25  # the host still knows the episode phase: the phased fast-forward is the one that runs
    outs = []
    for t in range(3):
        a = rng.uniform(-1, 1, (n, 6)).astype(np.float32)
        obs, rew, done = env.step(torch.from_numpy(a).cuda())
        o2, r2, d2, s2 = ref.step(a)
        same(obs.cpu().numpy(), o2, "hand-over obs %d" % t)
        same(done.cpu().numpy(), d2, "hand-over done %d" % t)
        outs.append((rew, r2))
    env.flush()
    for t, (rew, r2) in enumerate(outs):
        same(rew.cpu().numpy(), r2, "hand-over reward %d" % t)
    compare_state(env, ref, "hand-over final state")
    got, want = env.counters(), ref.counters()
    assert list(got.values()) == [int(x) for x in want], (got, want)
    assert got["goal_hits"] > 1000 and got["ball_court_terminations"] > 1000 and got["racket_ball_contact_substeps"] > 1000
    env.close()


def test_maximum_size_outline_stays_bit_exact(torch):
    """an outline with TB_MAX_HULL = 64 edges fills the table the kernels stage into LDS to the last row (the racket cull
    planes travel right behind it): an egg-shaped 64-gon, balls all around it and on its rim, Tennisbot with random scales"""
    import copy
    from tennisbot_rl_amd.params import load_scene
    sc = copy.deepcopy(load_scene())
    ang = np.linspace(0.0, 2.0 * np.pi, 64, endpoint=False)
    y = 0.16 * np.cos(ang) * (1.0 + 0.15 * np.cos(2 * ang))
    z = 0.5 + 0.3 * np.sin(ang) - 0.12 * np.sin(ang) ** 2   # link frame; the COM (inertial origin) is at z = 0.5
    sc["racket"]["hull_yz_ccw"] = [[float(a), float(b)] for a, b in zip(y, z)]
    n = 4096
    rng = np.random.default_rng(404)
    p = default_params(scene=sc)
    assert p.n_hull == 64
    from tennisbot_rl_amd.stepper import BatchedEnv
    env = BatchedEnv(ENV_TENNIS, n, device="cuda:0", seed=5, params=p, auto_reset=False)
    pf = p.copy(); pf.flags &= ~F_AUTO_RESET
    ref = OracleBatch(pf, ENV_TENNIS, n, seed=5, precision="f32")
    ref.L.tbo_set_threads(ref.h, 8)
    scale = rng.uniform(1.0, 3.0, n)
    q = rng.normal(size=(n, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    rp = np.stack([rng.uniform(8, 12, n), rng.uniform(-4, 4, n), rng.uniform(0.6, 1.5, n) * scale], 1)
    r = float(p.ball_radius)
    th = rng.uniform(0, 2 * np.pi, n)
    rad = rng.uniform(0.0, 1.15, n)  # inside, on the rim, just outside
    loc = np.stack([rng.uniform(-1, 1, n) * (float(p.racket_half_thick) * scale + r + 0.003),
                    0.17 * np.cos(th) * rad * scale, (0.3 * np.sin(th) * rad) * scale], 1)

    def rot(q, v):
        u, w = q[:, :3], q[:, 3:4]
        t = 2 * np.cross(u, v)
        return v + w * t + np.cross(u, t)
    fields = dict(racket_pos=rp, racket_quat=q, racket_vel=rng.uniform(-3, 3, (n, 3)), racket_angvel=rng.uniform(-5, 5, (n, 3)),
                  ball_pos=rp + rot(q, loc), ball_vel=rng.uniform(-12, 12, (n, 3)), ball_angvel=rng.uniform(-40, 40, (n, 3)),
                  shoot_force=(30, 0, 20), step_count=50, racket_scale=scale)
    w, d = make_words(ENV_TENNIS, n, **fields)
    env.set_state_words(w.view(np.int32), d); ref.set_state_words(w, d)
    for t in range(8):
        a = rng.uniform(-1, 1, (n, 2)).astype(np.float32)
        obs, rew, done = env.step(torch.from_numpy(a).cuda())
        o2, r2, d2, s2 = ref.step(a)
        same(obs.cpu().numpy(), o2, "64-gon obs %d" % t)
        same(rew.cpu().numpy(), r2, "64-gon reward %d" % t)
        same(done.cpu().numpy(), d2, "64-gon done %d" % t)
        compare_state(env, ref, "64-gon %d" % t)
    assert env.counters()["racket_ball_contact_substeps"] > n // 8
    env.close()


def test_float32_drift_vs_float64_truth(torch):
    """stated float32 tolerance: over one Swing episode without contacts the float32 state stays
    within 2e-4 (abs, metres / m/s) of the float64 oracle; done and step counters agree wherever
    the float64 trajectory is not within 1e-4 m of a contact threshold"""
    from tennisbot_rl_amd.stepper import BatchedEnv
    n = 2048
    rng = np.random.default_rng(17)
    p = default_params()
    env = BatchedEnv(ENV_SWING, n, seed=13, params=p, auto_reset=False)
    ref = OracleBatch(p, ENV_SWING, n, seed=13, precision="f64")
    env.reset(); ref.reset()
    for t in range(25):
        a = rng.uniform(-1, 1, (n, 6)).astype(np.float32)
        env.step(torch.from_numpy(a).cuda()); ref.step(a)
    g, c = env.get_state(), ref.get_state()
    quiet = ref.counters()[0] == 0 and env.counters()["racket_ball_contact_substeps"] == 0
    for k in ("racket_pos", "racket_vel", "ball_pos", "ball_vel"):
        err = np.abs(g[k] - c[k]).max()
        if quiet:
            assert err < 2e-4, (k, err)
    assert np.array_equal(g["step_count"], c["step_count"])


@pytest.mark.parametrize("n,reg_rows,ff", [(4096, True, {}), (1000, True, {}), (1000, False, {}),
                                           (1000, None, dict(ff_lanes_per_wave=1)), (1000, None, dict(ff_lanes_per_wave=7)), (4096, None, dict(ff_lanes_per_wave=64, ff_sort=False)),
                                           (1000, None, dict(ff_sort=True)), (4096, None, dict(ff_sort=True, ff_phases=2)), (40000, None, {}),
                                           (4096, None, dict(ff_phases=1)), (4096, None, dict(ff_phases=2)), (70000, None, dict(ff_phases=3, ff_sort=True))])
def test_pipelined_fast_forward_is_bit_identical(torch, n, reg_rows, ff):
    """tb_set_pipeline: the fast-forward runs on a side stream and writes the terminal step's
    reward late; after flush() every output equals the unpipelined path bit for bit (both builds
    of the pipelined step kernel: static contact rows in registers, as small batches run it, and in
    scratch; every way tb_ff_kernel hands parked envs to lanes, TbOptions.ff_lanes_per_wave / ff_sort:
    1, 7 or 64 per wave, sorted by predicted flight length in 1024-env groups; the loop in one kernel or cut into
    budgeted phases whose survivors are compacted for the next kernel, TbOptions.ff_phases)"""
    from tennisbot_rl_amd.rollout import RolloutBuffer
    from tennisbot_rl_amd.stepper import BatchedEnv
    T = 26 * 4 + 7
    rng = np.random.default_rng(41)
    acts = torch.from_numpy(rng.uniform(-1, 1, (T, n, 6)).astype(np.float32)).cuda()
    a = BatchedEnv(ENV_SWING, n, seed=6, pipeline=True, options=dict(swing_reg_rows=reg_rows, **ff))
    b = BatchedEnv(ENV_SWING, n, seed=6)
    ba, bb = RolloutBuffer(ENV_SWING, T, n, "cuda:0"), RolloutBuffer(ENV_SWING, T, n, "cuda:0")
    ba.actions.copy_(acts); bb.actions.copy_(acts)
    assert torch.equal(a.reset(), b.reset())
    for t in range(T):
        ba.step_into(a, t); bb.step_into(b, t)
    a.flush()
    torch.cuda.synchronize()
    assert torch.equal(ba.dones, bb.dones) and int(ba.dones.sum()) == 4 * n
    assert torch.equal(ba.obs, bb.obs)
    assert torch.equal(ba.rewards, bb.rewards) and float(ba.rewards.abs().sum()) > 0
    assert torch.equal(a.terminal_obs(), b.terminal_obs())
    wa, da = a.get_state_words(); wb, db = b.get_state_words()
    assert torch.equal(wa, wb) and torch.equal(da, db)
    assert a.counters() == b.counters()
    # a restored lockstep state keeps the pipeline armed (tb_set_state re-derives the phase) ...
    a.set_state_words(wa, da)
    assert a.phase() == T % 26
    # ... a masked reset breaks the lockstep: from here on every step may end some env's episode, every launch
    # gets a slot and a (mostly idle) fast-forward kernel behind it
    mask = (torch.arange(n, device="cuda:0") % 3 == 0)
    assert torch.equal(a.reset(mask), b.reset(mask)) and a.phase() == -1
    for t in range(30):
        oa, ra, dna = a.step(acts[t]); ob, rb, dnb = b.step(acts[t])
        a.flush()
        assert torch.equal(oa, ob) and torch.equal(ra, rb) and torch.equal(dna, dnb)
    # ... and a full reset re-arms it
    assert torch.equal(a.reset(), b.reset())
    outs = []
    for t in range(26):
        outs.append((a.step(acts[t]), b.step(acts[t])))
    a.flush()
    for (oa, ra, dna), (ob, rb, dnb) in outs:
        assert torch.equal(oa, ob) and torch.equal(ra, rb) and torch.equal(dna, dnb)
    a.close(); b.close()


def test_hipgraph_replay_equals_eager(torch):
    """BatchedEnv.capture: K steps (+ pipelined fast-forward branches) as one hipGraph launch"""
    from tennisbot_rl_amd.rollout import RolloutBuffer
    from tennisbot_rl_amd.stepper import BatchedEnv
    n, T = 2048, 78
    rng = np.random.default_rng(43)
    acts = torch.from_numpy(rng.uniform(-1, 1, (T, n, 6)).astype(np.float32)).cuda()
    for pipeline in (True, False):
        a = BatchedEnv(ENV_SWING, n, seed=8, pipeline=pipeline)
        b = BatchedEnv(ENV_SWING, n, seed=8)
        ba, bb = RolloutBuffer(ENV_SWING, T, n, "cuda:0").bind(a), RolloutBuffer(ENV_SWING, T, n, "cuda:0")
        ba.actions.copy_(acts); bb.actions.copy_(acts)
        a.reset(); b.reset()

        def body():
            for t in range(T):
                ba.step_into(a, t)
        g = a.capture(body)
        for rep in range(2):  # 78 = 3 whole episodes: the graph can be replayed back to back
            g.replay()
            for t in range(T):
                bb.step_into(b, t)
            torch.cuda.synchronize()
            assert torch.equal(ba.obs, bb.obs) and torch.equal(ba.rewards, bb.rewards) and torch.equal(ba.dones, bb.dones)
        wa, da = a.get_state_words(); wb, db = b.get_state_words()
        assert torch.equal(wa, wb) and torch.equal(da, db) and a.counters() == b.counters()
        a.close(); b.close()


@BOTH_PIPELINES
def test_graph_is_refused_at_another_phase_or_after_set_params(torch, defer):
    """a pipelined SwingRacket graph bakes in which of its steps end an episode; K % 26 != 0 moves the phase, and
    a second replay would let episode ends fall into launches without a fast-forward slot (terminal rewards lost).
    StepGraph refuses that instead (and a graph captured before set_params, whose launches carry the old block);
    the first replay of a 30-step graph equals eager stepping, and nothing is counted as a lockstep violation"""
    from tennisbot_rl_amd.rollout import RolloutBuffer
    from tennisbot_rl_amd.stepper import BatchedEnv, StepperError
    n, T = 1000, 30
    rng = np.random.default_rng(44)
    acts = torch.from_numpy(rng.uniform(-1, 1, (T, n, 6)).astype(np.float32)).cuda()
    a = BatchedEnv(ENV_SWING, n, seed=9, pipeline=True, track_terminal_obs=False, options=dict(ff_defer=defer))
    b = BatchedEnv(ENV_SWING, n, seed=9)
    ba, bb = RolloutBuffer(ENV_SWING, T, n, "cuda:0").bind(a), RolloutBuffer(ENV_SWING, T, n, "cuda:0")
    ba.actions.copy_(acts); bb.actions.copy_(acts)
    a.reset(); b.reset()
    g = a.capture(lambda: ba.step_range(a, 0, T))
    assert g.n_steps == T and g.phase == 0 and not g.repeatable and a.phase() == 0  # the capture ran nothing
    g.replay()
    for t in range(T):
        bb.step_into(b, t)
    torch.cuda.synchronize()
    assert torch.equal(ba.obs, bb.obs) and torch.equal(ba.rewards, bb.rewards) and torch.equal(ba.dones, bb.dones)
    assert a.phase() == 4
    with pytest.raises(StepperError, match="phase"):
        g.replay()
    # 22 eager steps bring the envs back to phase 0: the same graph is valid again
    more = torch.from_numpy(rng.uniform(-1, 1, (22, n, 6)).astype(np.float32)).cuda()
    for t in range(22):
        a.step(more[t]); b.step(more[t])
    a.flush()
    g.replay()
    for t in range(T):
        bb.step_into(b, t)
    torch.cuda.synchronize()
    assert torch.equal(ba.obs, bb.obs) and torch.equal(ba.rewards, bb.rewards) and torch.equal(ba.dones, bb.dones)
    ca, cb = a.counters(), b.counters()
    assert ca == cb and ca["lockstep_violations"] == 0
    a.set_params(default_params(lin_damp=0.05))
    with pytest.raises(StepperError, match="set_params"):
        g.replay()
    a.close(); b.close()


def test_curriculum_scale_reaches_replayed_graphs(torch):
    """train.py:164-176 calls set_racket_scale at every rollout start, and the trainer REPLAYS one captured
    rollout: the scale is read by the reset code from device memory (tb_set_racket_scale), not from the
    captured kernel arguments, so a replay rebuilds rackets with the new scale -- in lockstep with the oracle"""
    n, K = 512, 300
    env, ref = make_pair(torch, ENV_TENNIS, n)
    rng = np.random.default_rng(52)
    same(env.reset().cpu().numpy(), ref.reset(), "reset")
    acts = rng.uniform(-1, 1, (K, n, 2)).astype(np.float32)
    a_dev = torch.from_numpy(acts).cuda()
    out = {}

    def body():
        out["r"] = [env.step(a_dev[t]) for t in range(K)]
    g = env.capture(body)
    for rnd, scale in enumerate((3.0, 2.3, 1.3, 1.0)):
        env.set_racket_scale(scale)
        p = ref.params.copy(); p.racket_scale = scale; ref.set_params(p)
        g.replay()
        torch.cuda.synchronize()
        for t in range(K):
            o2, r2, d2, _ = ref.step(acts[t])
            o, r, d = out["r"][t]
            same(d.cpu().numpy(), d2, "replayed curriculum done %d/%d" % (rnd, t))
            same(o.cpu().numpy(), o2, "replayed curriculum obs %d/%d" % (rnd, t))
            same(r.cpu().numpy(), r2, "replayed curriculum reward %d/%d" % (rnd, t))
        compare_state(env, ref, "replayed curriculum round %d" % rnd)
    sc = env.get_state()["racket_scale"][:, 0]
    assert len(np.unique(sc)) >= 2 and np.float32(1.0) in sc   # the last scales arrived, env by env
    env.close()


@BOTH_PIPELINES
def test_restored_lockstep_state_rearms_the_pipeline(torch, defer):
    """tb_set_state re-derives the episode phase when every injected env is running at the same step count:
    a checkpoint restored into a fresh handle can go on with whole-episode launches (tb_policy_rollout needs
    the phase) -- and ragged step counts leave the phase unknown"""
    from tennisbot_rl_amd.stepper import BatchedEnv
    n = 700
    rng = np.random.default_rng(45)
    acts = torch.from_numpy(rng.uniform(-1, 1, (60, n, 6)).astype(np.float32)).cuda()
    a = BatchedEnv(ENV_SWING, n, seed=10, pipeline=True, track_terminal_obs=False, options=dict(ff_defer=defer))
    b = BatchedEnv(ENV_SWING, n, seed=10, pipeline=True, track_terminal_obs=False, options=dict(ff_defer=defer))
    twin = BatchedEnv(ENV_SWING, n, seed=10)
    a.reset(); twin.reset()
    for t in range(33):
        a.step(acts[t]); twin.step(acts[t])
    a.flush()
    w, d = a.get_state_words()
    b.set_state_words(w, d)
    assert a.phase() == 7 and b.phase() == 7
    ob, rb, db = b.rollout(acts[33:60])   # pipelined whole-episode launches from the restored phase
    b.flush()
    for t in range(33, 60):
        o, r, dn = twin.step(acts[t])
        assert torch.equal(o, ob[t - 33]) and torch.equal(r, rb[t - 33]) and torch.equal(dn, db[t - 33]), "step %d" % t
    assert b.counters()["lockstep_violations"] == 0
    w2 = w.clone(); w2[28, 0] += 1   # one env a step ahead: no common phase
    b.set_state_words(w2, d)
    assert b.phase() == -1
    a.close(); b.close(); twin.close()


def test_curriculum_scale_applies_at_each_envs_own_reset(torch):
    """train.py:164-176 calls env.set_racket_scale(s) between rollouts; tennisbot_env.py:230-234
    uses it at the next reset. In a batch every env must switch at ITS next reset."""
    n = 512
    env, ref = make_pair(torch, ENV_TENNIS, n)
    rng = np.random.default_rng(51)
    same(env.reset().cpu().numpy(), ref.reset(), "reset")
    scales = {0: 3.0, 300: 2.3, 700: 1.3}
    for t in range(1100):
        if t in scales:
            env.set_racket_scale(scales[t])
            p = ref.params.copy(); p.racket_scale = scales[t]; ref.set_params(p)
        a = rng.uniform(-1, 1, (n, 2)).astype(np.float32)
        obs, rew, done = env.step(torch.from_numpy(a).cuda())
        o2, r2, d2, s2 = ref.step(a)
        same(done.cpu().numpy(), d2, "scale curriculum done %d" % t)
        same(obs.cpu().numpy(), o2, "scale curriculum obs %d" % t)
        same(rew.cpu().numpy(), r2, "scale curriculum reward %d" % t)
        if t % 100 == 50:
            compare_state(env, ref, "scale curriculum %d" % t)
            sc = env.get_state()["racket_scale"][:, 0]
            assert set(np.unique(sc)) <= {1.0, np.float32(3.0), np.float32(2.3), np.float32(1.3)}
    st = env.get_state()
    assert len(np.unique(st["racket_scale"])) >= 2   # envs are at different curriculum stages at the same time
    env.close()


def test_racket_ground_contact_opt_in(torch):
    """row f3: racket<->court contact (TB_F_RACKET_GROUND). Whole SwingRacket episodes (the racket
    lands during the fast-forward) and injected states with the racket on / in the ground at random
    orientations, in lockstep with the oracle"""
    env, ref = make_pair(torch, ENV_SWING, 1024, flags=F_DEFAULT | F_RACKET_GROUND)
    run_lockstep(torch, env, ref, 54, np.random.default_rng(61), "swing racket-ground", check_state_every=1)
    st = env.get_state()
    env.close()
    n = 512
    rng = np.random.default_rng(62)
    env, ref = make_pair(torch, ENV_SWING, n, auto_reset=False, flags=F_DEFAULT | F_RACKET_GROUND)
    q = rng.normal(size=(n, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    q[: n // 4] = (0, 0, 0, 1)                                   # upright
    q[n // 4: n // 2] = (0, np.sqrt(0.5), 0, np.sqrt(0.5))       # lying on a face
    # COM height that puts the LOWEST hull vertex within [-4 mm, +12 mm] of the ground's top face
    p = default_params()
    hv = p.hull_vertices()
    verts = np.concatenate([np.c_[np.full(len(hv), sx * p.racket_half_thick), hv] for sx in (-1, 1)])   # [76, 3] racket frame

    def rot(qq, v):
        u, w0 = qq[:, None, :3], qq[:, None, 3:4]
        tt = 2 * np.cross(u, v[None])
        return v[None] + w0 * tt + np.cross(u, tt)
    lowest = rot(q, verts)[:, :, 2].min(1)
    z = -lowest + 0.005 + 0.001 + rng.uniform(-0.004, 0.012, n)
    w, d = make_words(ENV_SWING, n, racket_pos=np.stack([rng.uniform(6, 13.9, n), rng.uniform(-6.9, 7.2, n), z], 1), racket_quat=q,
                      racket_vel=rng.uniform(-2, 2, (n, 3)), racket_angvel=rng.uniform(-3, 3, (n, 3)), ball_pos=(0.0, 3.0, 50.0),
                      goal=(-6, 0), spawn_pos=(9, 0, 0.6), init_dist=10.0, step_count=rng.integers(0, 20, n))
    env.set_state_words(w.view(np.int32), d); ref.set_state_words(w, d)
    for t in range(12):
        a = rng.uniform(-1, 1, (n, 6)).astype(np.float32)
        obs, rew, done = env.step(torch.from_numpy(a).cuda())
        o2, r2, d2, s2 = ref.step(a)
        same(obs.cpu().numpy(), o2, "racket-ground obs %d" % t)
        compare_state(env, ref, "racket-ground forced %d" % t)
    assert env.counters()["nonfinite_states"] == 0
    env.close()


@BOTH_PIPELINES
def test_step_sequence_equals_per_step_calls_and_chunked_gather_views(torch, defer):
    """tb_step_sequence (one host call for a run of rollout slots) against step_into per slot, with the
    pipelined fast-forward on; and the chunked-gather bookkeeping on a single rank (no collective)"""
    from tennisbot_rl_amd.rollout import RolloutBuffer
    from tennisbot_rl_amd.stepper import BatchedEnv
    n, T = 1000, 78
    rng = np.random.default_rng(3)
    acts = torch.from_numpy(rng.uniform(-1, 1, (T, n, 6)).astype(np.float32)).cuda()
    bufs = []
    for mode in ("per_step", "sequence"):
        env = BatchedEnv(ENV_SWING, n, device="cuda:0", seed=21, track_terminal_obs=False, pipeline=True, options=dict(ff_defer=defer))
        buf = RolloutBuffer(ENV_SWING, T, n, "cuda:0").bind(env)
        buf.actions.copy_(acts)
        env.reset()
        if mode == "per_step":
            for t in range(T):
                buf.step_into(env, t)
        else:
            buf.step_range(env, 0, 30)
            buf.step_range(env, 30, T)
        env.flush()
        torch.cuda.synchronize()
        bufs.append(buf)
        with pytest.raises(ValueError):
            buf.step_range(env, 10, T + 1)
    assert torch.equal(bufs[0].raw, bufs[1].raw)
    assert float(bufs[0].dones.sum()) == 3 * n  # three whole episodes went through the side streams / the pool
    buf = bufs[1]
    buf.begin_gather(3)
    for c in range(3):
        buf.gather_chunk(c)  # single rank: nothing to exchange
    shards = buf.finish_gather()
    assert len(shards) == 1 and torch.equal(shards[0][0][0], buf.obs)


@pytest.mark.parametrize("kind,n_chunks,defer", [(ENV_SWING, 4, False), (ENV_SWING, 13, False), (ENV_SWING, 4, "all"), (ENV_SWING, 13, "all"), (ENV_TENNIS, 8, None)],
                         ids=["swing-4-slots", "swing-13-slots", "swing-4-pool", "swing-13-pool", "tennis-8"])
def test_progress_marks_release_each_chunk_of_one_graph(torch, kind, n_chunks, defer):
    """RolloutBuffer.capture_marked: the whole rollout is ONE hipGraph; mark c (tb_mark_record: a counter in
    pinned host memory, bumped by a kernel node behind chunk c's steps; the fast-forwards the chunk is owed are
    counted the same way) tells the host that chunk c's records are final while the graph is still stepping. A copy issued on a
    side stream right after the host saw the mark must hold exactly this replay's chunk; issued any earlier it
    would pick up the previous replay's bytes. Also bit-identical to call-by-call stepping."""
    from tennisbot_rl_amd.rollout import RolloutBuffer
    from tennisbot_rl_amd.stepper import BatchedEnv, StepperError
    n, T = 4096, 104
    A = 6 if kind == ENV_SWING else 2
    rng = np.random.default_rng(21)
    acts = torch.from_numpy(rng.uniform(-1, 1, (T, n, A)).astype(np.float32)).cuda()
    piped = kind == ENV_SWING
    ref_env = BatchedEnv(kind, n, device="cuda:0", seed=8, track_terminal_obs=False, pipeline=piped)
    ref = RolloutBuffer(kind, T, n, "cuda:0").bind(ref_env)
    env = BatchedEnv(kind, n, device="cuda:0", seed=8, track_terminal_obs=False, pipeline=piped, options=dict(ff_defer=defer))
    assert env.pipeline_form() == ("none" if not piped else "slots" if defer is False else "pool")
    buf = RolloutBuffer(kind, T, n, "cuda:0").bind(env)
    ref.actions.copy_(acts); buf.actions.copy_(acts)
    main, side = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(main):
        ref_env.reset(); env.reset()
        for t in range(5):  # chunk boundaries off the episode grid
            ref.step_into(ref_env, t); buf.step_into(env, t)
        graph = buf.capture_marked(env, n_chunks)
        assert [env.mark_count(c) for c in range(n_chunks)] == [0] * n_chunks  # a capture runs nothing
        shadow = torch.zeros_like(buf.raw)
        cb = buf.record * (T // n_chunks)
        early = 0
        for rnd in range(3):
            for t in range(T):
                ref.step_into(ref_env, t)
            ref_env.flush()
            main.synchronize()
            env.mark_begin()
            graph.replay()
            early += env.mark_count(n_chunks - 1) == rnd  # the last mark has not fired yet: the host really runs ahead of the graph
            for c in range(n_chunks):
                env.mark_host_wait(c)
                with torch.cuda.stream(side):
                    shadow[c * cb:(c + 1) * cb].copy_(buf.raw[c * cb:(c + 1) * cb], non_blocking=True)
            side.synchronize(); main.synchronize()
            assert [env.mark_count(c) for c in range(n_chunks)] == [rnd + 1] * n_chunks
            assert torch.equal(ref.raw, buf.raw), "round %d" % rnd
            bad = [c for c in range(n_chunks) if not torch.equal(shadow[c * cb:(c + 1) * cb], buf.raw[c * cb:(c + 1) * cb])]
            assert not bad, "round %d: chunks %s were copied before they were final" % (rnd, bad)
        # (`early` = rounds in which the host got here before the last mark fired: normally all 3, but a host thread descheduled for a
        #  millisecond misses it -- timing, not semantics, so it is not asserted)
        assert env.counters() == ref_env.counters()
        env.mark_begin()
        with pytest.raises(StepperError):  # a mark that nobody fires: the wait gives up
            env.mark_host_wait(0, timeout_ms=20)
    env.close(); ref_env.close()


_ABANDONED_CAPTURE_CASE = r"""
import sys
import numpy as np, torch
from tennisbot_rl_amd.params import ENV_SWING
from tennisbot_rl_amd.stepper import BatchedEnv
n = 512
rng = np.random.default_rng(8)
acts = torch.from_numpy(rng.uniform(-1, 1, (60, n, 6)).astype(np.float32)).cuda()
# argv[1] = "all" (what the automatic choice is at this size): the episode end inside the abandoned capture was "parked" straight into
# the pool (TbOptions.ff_defer = 2) by a launch that never ran -- nothing may be pending on its account afterwards; "slots": one
# fast-forward kernel per episode end, whose side stream the capture had forked
env = BatchedEnv(ENV_SWING, n, device="cuda:0", seed=4, track_terminal_obs=False, pipeline=True, options=dict(ff_defer=False if sys.argv[1] == "slots" else sys.argv[1]))
twin = BatchedEnv(ENV_SWING, n, device="cuda:0", seed=4, track_terminal_obs=False, pipeline=True)
env.reset(); twin.reset()
for t in range(20):
    env.step(acts[t]); twin.step(acts[t])

def bad():
    for t in range(20, 50):  # crosses an episode end: side streams get forked into the capture
        env.step(acts[t])
    torch.cuda.synchronize()  # not capturable

try:
    env.capture(bad)
    sys.exit("the capture was expected to fail")
except Exception:
    pass
# HIP (ROCm 7.2) keeps refusing work on the legacy default stream of a process whose capture was
# invalidated -- torch's own ops included -- so the survivor continues on a stream of its own
with torch.cuda.stream(torch.cuda.Stream()):
    for t in range(20, 60):
        a, b = env.step(acts[t]), twin.step(acts[t])
        env.flush(); twin.flush()
        if not all(torch.equal(x, y) for x, y in zip(a, b)):
            sys.exit("step %d after the abandoned capture differs" % t)
    wa, da = env.get_state_words()
    wb, db = twin.get_state_words()
    if not (torch.equal(wa, wb) and torch.equal(da, db)):
        sys.exit("state differs")
print("survived")
"""


@pytest.mark.parametrize("defer", ["slots", "all"])
def test_abandoned_capture_leaves_the_env_usable(torch, defer):
    """a capture that fails half-way (here: a host synchronisation inside it) must not poison the
    handle: tb_pipeline_recover replaces the forked side streams and restores the phase hint, and
    the env then steps exactly like a twin that never saw the capture. Runs in a process of its own:
    the HIP runtime stays unusable on the default stream of a process this has happened to."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _ABANDONED_CAPTURE_CASE, defer], cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "survived" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_replay_orders_itself_behind_unflushed_eager_fast_forwards(torch):
    """StepGraph.replay() after eager pipelined steps that were NOT flushed: the captured launches bake in slot indices and carry
    no wait on the handle's side streams, so replay() itself must join the eager fast-forwards still reading those slots. Racket<->court
    contact on: its fast-forwards run for milliseconds, eight episodes leave all eight slots busy when the replay starts."""
    from tennisbot_rl_amd.rollout import RolloutBuffer
    from tennisbot_rl_amd.stepper import BatchedEnv
    n, T, E = 2048, 52, 8 * 26
    p = default_params(flags=F_DEFAULT | F_RACKET_GROUND)
    rng = np.random.default_rng(77)
    acts = torch.from_numpy(rng.uniform(-1, 1, (T, n, 6)).astype(np.float32)).cuda()
    eager = torch.from_numpy(rng.uniform(-1, 1, (E, n, 6)).astype(np.float32)).cuda()
    a = BatchedEnv(ENV_SWING, n, seed=3, params=p, pipeline=True, track_terminal_obs=False, options=dict(ff_defer=False))  # slots: kernels in flight
    b = BatchedEnv(ENV_SWING, n, seed=3, params=p)
    ba, bb = RolloutBuffer(ENV_SWING, T, n, "cuda:0").bind(a), RolloutBuffer(ENV_SWING, T, n, "cuda:0")
    ea = RolloutBuffer(ENV_SWING, E, n, "cuda:0").bind(a)
    ba.actions.copy_(acts); bb.actions.copy_(acts); ea.actions.copy_(eager)
    a.reset(); b.reset()
    g = a.capture(lambda: ba.step_range(a, 0, T))
    ea.step_range(a, 0, E)  # eight episodes, eight fast-forwards in flight, no flush
    g.replay()
    eb_rew = []
    for t in range(E):
        eb_rew.append(b.step(eager[t])[1])
    for t in range(T):
        bb.step_into(b, t)
    a.flush()
    torch.cuda.synchronize()
    assert torch.equal(ea.rewards, torch.stack(eb_rew)), "terminal rewards of the eager episodes were lost or garbled"
    assert torch.equal(ba.obs, bb.obs) and torch.equal(ba.rewards, bb.rewards) and torch.equal(ba.dones, bb.dones)
    ca, cb = a.counters(), b.counters()
    assert ca == cb and ca["lockstep_violations"] == 0
    a.close(); b.close()


def test_set_pipeline_failure_is_all_or_nothing(torch):
    """a device allocation that fails in the middle of tb_set_pipeline leaves NOTHING behind: the handle steps on unpipelined
    (no kernel parks into a half-built slot), and a second tb_set_pipeline builds the pipeline from scratch -- both in lockstep
    with the oracle"""
    env, ref = make_pair(torch, ENV_SWING, 1000, seed=21)
    rng = np.random.default_rng(5)
    same(env.reset().cpu().numpy(), ref.reset(), "reset obs")
    L, h = env.L, env._h
    for nth in (1, 2, 5, 17):  # first slot's records, its flags, second slot's records, somewhere in the fourth slot
        assert L.tb_diag_fail_alloc(nth) == 0
        rc = L.tb_set_pipeline(h, 1)
        assert rc == 2, rc  # hipErrorOutOfMemory, passed through
        assert b"pipeline_malloc" in L.tb_last_error() or b"hipErrorOutOfMemory" in L.tb_last_error() or b"out of memory" in L.tb_last_error().lower()
        for t in range(27):  # through an episode end: the fast-forward runs inside the step kernel, as without a pipeline
            a = rng.uniform(-1, 1, (1000, 6)).astype(np.float32)
            obs, rew, done = env.step(torch.from_numpy(a).cuda())
            o2, r2, d2, s2 = ref.step(a)
            same(done.cpu().numpy(), d2, "done"); same(rew.cpu().numpy(), r2, "reward"); same(obs.cpu().numpy(), o2, "obs")
            same(env.last_substeps().cpu().numpy(), s2, "substeps")
    L.tb_diag_fail_alloc(0)
    same(env.reset().cpu().numpy(), ref.reset(), "reset obs")  # lockstep again: the pipelined kernels know the phase
    assert L.tb_set_pipeline(h, 1) == 0
    env.pipeline, env._term = True, None
    rews, want = [], []
    for t in range(54):
        a = rng.uniform(-1, 1, (1000, 6)).astype(np.float32)
        obs, rew, done = env.step(torch.from_numpy(a).cuda())
        o2, r2, d2, s2 = ref.step(a)
        same(done.cpu().numpy(), d2, "piped done"); same(obs.cpu().numpy(), o2, "piped obs")
        rews.append(rew); want.append(r2)
    env.flush()
    same(torch.stack(rews).cpu().numpy(), np.stack(want), "piped rewards")
    c = env.counters()
    assert c["lockstep_violations"] == 0 and list(c.values()) == [int(x) for x in ref.counters()]
    env.close()


@pytest.mark.parametrize("n,options", [(4096, dict(ff_defer=False)), (4096, dict(ff_defer="all")), (1000, dict(ff_defer=True, ff_defer_margin=3)),
                                       (140000, dict(ff_phases=3)), (140000, dict(ff_phases=2, ff_lanes_per_wave=16))],
                         ids=["slots", "pool", "stragglers", "big-3-phases", "big-2-phases"])
def test_fast_forward_with_balls_thrown_at_a_spinning_racket(torch, n, options):
    """random-action episodes enter the fast-forward with the ball next to the racket and strike it, if at all, in its first substeps.
    Here every env enters it (step 26) with the ball 0.6-1.3 m OUTSIDE the bounding sphere of a spinning, falling racket, thrown at
    where the racket will be: most come into reach 5-150 substeps into the loop, a good part strike the racket -- late racket
    contacts, resting balls that run to the 800-substep limit, lanes that the first phase of the large-batch form hands over
    long after its start. Rewards, substep counts and the counters bit-exact against the oracle through every form of the
    fast-forward (its own kernel per episode, the pool, stragglers parked again, the large-batch phases)."""
    from tennisbot_rl_amd.stepper import BatchedEnv
    rng = np.random.default_rng(1234 + n)
    p = default_params()
    env = BatchedEnv(ENV_SWING, n, device="cuda:0", seed=5, params=p, pipeline=True, track_terminal_obs=False, options=options)
    pf = p.copy(); pf.flags |= F_AUTO_RESET
    ref = OracleBatch(pf, ENV_SWING, n, seed=5, precision="f32")
    ref.L.tbo_set_threads(ref.h, 16)
    q = rng.normal(size=(n, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    rp = np.stack([rng.uniform(8, 10, n), rng.uniform(-2, 2, n), rng.uniform(1.5, 4.0, n)], 1)
    rv = rng.uniform(-1.5, 1.5, (n, 3))
    u = rng.normal(size=(n, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
    dist = rng.uniform(1.2, 1.9, n)  # the sphere's radius is ~0.6
    bp = rp + u * dist[:, None]
    bp[:, 2] = np.maximum(bp[:, 2], 0.3)
    tof = rng.uniform(0.05, 0.6, n)  # aim at where a racket in free fall will be after `tof` seconds
    target = rp + rv * tof[:, None] + np.array([0.0, 0.0, -0.5 * 9.81])[None, :] * (tof ** 2)[:, None]
    bv = (target - bp) / tof[:, None] + np.array([0.0, 0.0, 0.5 * 9.81])[None, :] * tof[:, None] + rng.normal(scale=0.3, size=(n, 3))
    fields = dict(racket_pos=rp, racket_quat=q, racket_vel=rv, racket_angvel=rng.uniform(-9, 9, (n, 3)), ball_pos=bp, ball_vel=bv,
                  ball_angvel=rng.uniform(-30, 30, (n, 3)), goal=np.stack([rng.uniform(-11, -4, n), rng.uniform(-4, 4, n)], 1), spawn_pos=(9, 0, 0.6),
                  init_dist=rng.uniform(8, 20, n), step_count=25)
    w, d = make_words(ENV_SWING, n, **fields)
    env.set_state_words(torch.from_numpy(w.view(np.int32)).cuda(), torch.from_numpy(d).cuda()); ref.set_state_words(w, d)
    assert env.phase() == 25
    outs = []
    for t in range(27):  # the step that parks everything, then a whole ordinary episode behind it
        a = rng.uniform(-1, 1, (n, 6)).astype(np.float32)
        obs, rew, done = env.step(torch.from_numpy(a).cuda())
        o2, r2, d2, s2 = ref.step(a)
        same(done.cpu().numpy(), d2, "done %d" % t); same(obs.cpu().numpy(), o2, "obs %d" % t)
        outs.append((rew, r2))
        if t == 0:
            assert d2.all() and 0.2 * n < (s2 > 30).sum()
    env.flush()
    for t, (rew, r2) in enumerate(outs):
        same(rew.cpu().numpy(), r2, "reward %d" % t)
    got, want = env.counters(), ref.counters()
    assert list(got.values()) == [int(x) for x in want], (got, want)
    assert got["racket_ball_contact_substeps"] > n // 20 and got["nonfinite_states"] == 0 and got["lockstep_violations"] == 0
    env.close()


@pytest.mark.parametrize("options", [dict(ff_defer="all"), dict(ff_defer=True, ff_defer_margin=40)], ids=["pool", "stragglers"])
def test_sealed_fate_exit_books_exactly_what_the_full_flight_gives(torch, options, seed=4321, n=8192, over=None, off=False):
    """TbOptions.ff_seal: the pool leaves a flight whose ball has fallen below the court, out of the racket's reach for good, and books
    the substeps up to the 800-substep timeout instead of running them. The oracle has no such exit: rewards, done flags,
    observations and every counter (substeps and timeouts among them) must be those of the flights run to their end. The states
    are chosen against the exit's argument: rackets that the court does not hold (default contact set) fall under it beside the
    balls, swinging about their anchors at up to 8 m/s, balls pass within centimetres of them below the court, and a racket that
    does strike there can send the ball back up into the court's underside -- an exit taken one check too early shows as a
    missing racket contact, a wrong reward or a wrong substep count."""
    from tennisbot_rl_amd.stepper import BatchedEnv
    rng = np.random.default_rng(seed)
    p = default_params(**(over or {}))  # over: other engine parameters; off: they are OUTSIDE the exit's argument -- it must then stay off (and everything still match)
    pf = p.copy(); pf.flags |= F_AUTO_RESET
    ref = OracleBatch(pf, ENV_SWING, n, seed=5, precision="f32")
    ref.L.tbo_set_threads(ref.h, 16)
    q = rng.normal(size=(n, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    spawn = np.array([9.0, 0.0, 0.6])
    rp = spawn[None, :] + np.stack([rng.uniform(-1.5, 1.5, n), rng.uniform(-4, 4, n), rng.uniform(-8, 3, n)], 1)
    rv = np.stack([rng.uniform(-8, 8, n), rng.uniform(-4, 4, n), rng.uniform(-6, 2, n)], 1)
    u = rng.normal(size=(n, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
    bp = rp + u * rng.uniform(0.7, 6.0, n)[:, None]
    kind = rng.integers(0, 3, n)
    # a third drift beside the racket, a third are thrown at where it will be, a third fly anywhere (most of those out of the court)
    tof = rng.uniform(0.1, 1.5, n)
    aimed = (rp + rv * tof[:, None] - bp) / tof[:, None] + rng.normal(scale=0.5, size=(n, 3))
    bv = np.where((kind == 0)[:, None], rv + rng.normal(scale=0.7, size=(n, 3)), np.where((kind == 1)[:, None], aimed, rng.normal(scale=12.0, size=(n, 3))))
    fields = dict(racket_pos=rp, racket_quat=q, racket_vel=rv, racket_angvel=rng.uniform(-9, 9, (n, 3)), ball_pos=bp, ball_vel=bv,
                  ball_angvel=rng.uniform(-30, 30, (n, 3)), goal=np.stack([rng.uniform(-11, -4, n), rng.uniform(-4, 4, n)], 1), spawn_pos=tuple(spawn),
                  init_dist=rng.uniform(8, 20, n), step_count=25)
    w, d = make_words(ENV_SWING, n, **fields)
    ref.set_state_words(w, d)
    acts = [rng.uniform(-1, 1, (n, 6)).astype(np.float32) for _ in range(27)]
    want = [ref.step(a) for a in acts[:1]]
    ref_first = want[0]
    assert ref_first[2].all()
    late = ref.counters()
    assert late[3] > n // (4 if not over else 40) and (over or late[0] > n // 50), late  # timeouts; racket contacts inside the fast-forward
    want += [ref.step(a) for a in acts[1:]]
    results = {}
    for seal in (True, False):
        env = BatchedEnv(ENV_SWING, n, device="cuda:0", seed=5, params=p, pipeline=True, track_terminal_obs=False, options=dict(options, ff_seal=seal))
        env.set_state_words(torch.from_numpy(w.view(np.int32)).cuda(), torch.from_numpy(d).cuda())
        outs = []
        for t, a in enumerate(acts):
            obs, rew, done = env.step(torch.from_numpy(a).cuda())
            same(done.cpu().numpy(), want[t][2], "done %d" % t); same(obs.cpu().numpy(), want[t][0], "obs %d" % t)
            outs.append(rew)
        env.flush()
        for t, rew in enumerate(outs):
            same(rew.cpu().numpy(), want[t][1], "reward %d (ff_seal %s)" % (t, seal))
        got = env.counters()
        assert list(got.values()) == [int(x) for x in ref.counters()], (seal, got, ref.counters())
        results[seal] = (env.sealed_substeps(), got["substeps"])
        env.close()
    assert results[False][0] == 0
    booked, total = results[True]
    if off:
        assert booked == 0, results
    else:
        assert booked > 0.3 * total, results  # the exit did fire: a good part of these flights' substeps were never run


@pytest.mark.parametrize("over", [dict(lin_damp=0.0, lin_damp_quad=0.0), dict(racket_mass=0.6, lin_damp_quad=0.0), dict(racket_mass=12.0, gravity=3.0), dict(lin_damp=0.5, lin_damp_quad=0.04)],
                         ids=["no-drag", "light-racket", "heavy-racket-low-gravity", "thick-air"])
def test_sealed_fate_exit_with_other_engine_parameters_inside_its_argument(torch, over):
    """the oscillator bound of the exit's claim (2) against the engine itself, not against the recurrence it was derived on: an undamped racket
    (its swing never decays), a light one (w dt = 0.04, near the limit's order), a heavy slow one under weak gravity (long flights, slow
    falls), thick air -- the exit fires and every output is still that of the full flights"""
    test_sealed_fate_exit_books_exactly_what_the_full_flight_gives(torch, dict(ff_defer="all"), seed=99, n=8192, over=over)


@pytest.mark.parametrize("over", [dict(magnus_k=0.002), dict(racket_mass=0.005), dict(lin_damp_quad=0.1)], ids=["magnus", "stiff-racket-spring", "heavy-drag"])
def test_sealed_fate_exit_stays_off_outside_its_argument(torch, over):
    """the exit's argument needs no Magnus force (a spinning ball's lift can bring it back), (w dt)^2 <= 0.04 for the racket's
    restoring spring and dt kd <= 0.2 at 1000 m/s for the drag (fate_sealed, seal_params_ok): with parameters beyond any of them
    the pool runs every flight to its end -- nothing booked -- and every output still matches the oracle"""
    test_sealed_fate_exit_books_exactly_what_the_full_flight_gives(torch, dict(ff_defer="all"), seed=77, n=4096, over=over, off=True)


def thrown_at_racket_through_the_short_steps(torch, n, n_edges, options, step0=12, threads=16):
    """balls thrown at spinning rackets from inside the bounding sphere to well outside it, starting at agent step `step0`: half of
    them arrive during the short steps step0+1 .. 25 (the step kernels' racket narrowphase), the others in the fast-forward; the rest
    of that episode and a whole ordinary one behind it, bit for bit against the oracle"""
    from tennisbot_rl_amd.stepper import BatchedEnv
    rng = np.random.default_rng(77 + n_edges)
    p = with_outline(default_params(), n_edges) if n_edges else default_params()
    env = BatchedEnv(ENV_SWING, n, device="cuda:0", seed=5, params=p, pipeline=True, track_terminal_obs=False, options=options)
    pf = p.copy(); pf.flags |= F_AUTO_RESET
    ref = OracleBatch(pf, ENV_SWING, n, seed=5, precision="f32")
    ref.L.tbo_set_threads(ref.h, threads)
    q = rng.normal(size=(n, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    rp = np.stack([rng.uniform(8, 10, n), rng.uniform(-2, 2, n), rng.uniform(1.5, 4.0, n)], 1)
    rv = rng.uniform(-1.5, 1.5, (n, 3))
    u = rng.normal(size=(n, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
    dist = rng.uniform(0.1, 0.9, n)  # from inside the bounding sphere to well outside it
    bp = rp + u * dist[:, None]
    bp[:, 2] = np.maximum(bp[:, 2], 0.3)
    tof = rng.uniform(0.02, 0.3, n)
    target = rp + rv * tof[:, None] + np.array([0.0, 0.0, -0.5 * 9.81])[None, :] * (tof ** 2)[:, None]
    bv = (target - bp) / tof[:, None] + np.array([0.0, 0.0, 0.5 * 9.81])[None, :] * tof[:, None] + rng.normal(scale=0.3, size=(n, 3))
    fields = dict(racket_pos=rp, racket_quat=q, racket_vel=rv, racket_angvel=rng.uniform(-9, 9, (n, 3)), ball_pos=bp, ball_vel=bv,
                  ball_angvel=rng.uniform(-30, 30, (n, 3)), goal=np.stack([rng.uniform(-11, -4, n), rng.uniform(-4, 4, n)], 1), spawn_pos=(9, 0, 0.6),
                  init_dist=rng.uniform(8, 20, n), step_count=step0)
    w, d = make_words(ENV_SWING, n, **fields)
    env.set_state_words(torch.from_numpy(w.view(np.int32)).cuda(), torch.from_numpy(d).cuda()); ref.set_state_words(w, d)
    assert env.phase() == step0
    outs = []
    short_step_contacts = 0
    for t in range(26 - step0 + 26):
        a = rng.uniform(-1, 1, (n, 6)).astype(np.float32)
        obs, rew, done = env.step(torch.from_numpy(a).cuda())
        o2, r2, d2, s2 = ref.step(a)
        same(done.cpu().numpy(), d2, "done %d" % t); same(obs.cpu().numpy(), o2, "obs %d" % t)
        outs.append((rew, r2))
        if t < 24 - step0:
            short_step_contacts += int((r2 == 2.0).sum())  # the contact bonus of swingracket_env.py:98-101: only a short step's racket row pays it
    env.flush()
    for t, (rew, r2) in enumerate(outs):
        same(rew.cpu().numpy(), r2, "reward %d" % t)
    got, want = env.counters(), ref.counters()
    assert list(got.values()) == [int(x) for x in want], (got, want)
    assert got["racket_ball_contact_substeps"] > n // 20 and got["nonfinite_states"] == 0 and got["lockstep_violations"] == 0
    assert short_step_contacts > n // 50, short_step_contacts  # the step kernels' own outline sweeps were exercised, not only the fast-forward's
    env.close()


@pytest.mark.parametrize("n_edges", [3, 4, 5, 8, 11, 38, 40, 48, 56, 60, 62, 63, 64])
def test_outline_sweep_with_other_outlines(torch, n_edges):
    """the outline sweep takes four edges per trip and the rest one by one: outlines of 3 (no whole trip), 4 and 8 (no rest), 5 and
    11 edges and the largest the table holds, 64 -- balls thrown at spinning rackets as above, through the step kernels (a ball
    next to the racket during the 25 short steps) and both forms of the fast-forward, bit for bit against the oracle"""
    for options in (dict(ff_defer="all"), dict(ff_defer=False)):
        thrown_at_racket_through_the_short_steps(torch, 3000, n_edges, options)


def test_pool_run_never_reruns_a_consumed_record(torch):
    """with every episode end parked straight into the pool (ff_defer = 2) a region's records are rewritten by
    the next launch that parks into it -- unless an env does not park there. The library keeps episodes in lockstep, so only a
    caller that replays a captured graph at another episode phase (raw hipGraphLaunch through the C ABI; stepper.StepGraph refuses)
    gets there: the captured parking launch then parks nobody, and the pool run of that replay finds the LAST replay's records in
    its region, each with a destination pointer. They must not run again: a consumed record says so itself (its tag is cleared by
    the pool run). Before round 4 the stale episodes were re-run and their rewards written through the stale pointers."""
    from tennisbot_rl_amd.stepper import BatchedEnv
    n, K = 1000, 26
    rng = np.random.default_rng(3)
    env = BatchedEnv(ENV_SWING, n, device="cuda:0", seed=2, pipeline=True, track_terminal_obs=False, options=dict(ff_defer="all"))
    assert env.pipeline_form() == "pool"
    env.reset()
    acts = torch.from_numpy(rng.uniform(-1, 1, (K, n, 6)).astype(np.float32)).cuda()
    obs = torch.zeros((K, n, 6), device="cuda"); rew = torch.zeros((K, n), device="cuda"); done = torch.zeros((K, n), dtype=torch.uint8, device="cuda")
    g = env.capture(lambda: [env.step(acts[t], out=(obs[t], rew[t], done[t])) for t in range(K)])
    g.replay()
    torch.cuda.synchronize()
    first = rew[K - 1].clone()
    assert (first != 0).float().mean() > 0.05 and bool(done[K - 1].all())  # the pool run paid the terminal rewards (a ball that drops straight down earns exactly 0)
    for t in range(10):  # move the envs to phase 10 outside the graph
        env.step(acts[t])
    env.flush(); torch.cuda.synchronize()
    assert env.phase() == 10 and not g.valid()
    with pytest.raises(Exception):
        g.replay()  # the Python surface refuses ...
    rew.fill_(-777.0)
    g.graph.replay()  # ... a C-ABI caller's own hipGraphLaunch does not ask
    torch.cuda.synchronize()
    c = env.counters()
    assert c["lockstep_violations"] == n, c  # every env ended its episode in a launch captured without a parking slot: reported
    # the captured parking launch (graph step 25, envs at step 9 of their episode) parked nobody and wrote plain step rewards;
    # the pool run behind it found only consumed records in its region: nothing of the first replay's episodes ran again
    last = rew[K - 1].cpu().numpy()
    assert np.array_equal(last, np.zeros(n, np.float32)) or set(np.unique(last)) <= {0.0, 2.0}, np.unique(last)[:8]
    nz = first.cpu().numpy() != 0
    assert not (last[nz] == first.cpu().numpy()[nz]).any()  # none of the first replay's terminal rewards came back
    env.close()


@pytest.mark.parametrize("n,options,n_edges", [(3000, dict(block=128), 38), (3000, dict(block=256), 38), (3000, dict(block=256, ff_defer=False), 11),
                                               (1061, dict(block=128), 63), (3000, dict(block=128, swing_reg_rows=False), 38), (20011, None, 38),
                                               (20011, dict(block=256), 5)],
                         ids=["block128", "block256", "block256-slots-11gon", "block128-63gon", "block128-lds-rows", "auto-block-20011", "block256-20011-5gon"])
def test_lazily_copied_outline_table_in_multi_wave_workgroups(torch, n, options, n_edges):
    """The pipelined SwingRacket step kernel copies the outline table into LDS lazily: the first WAVE whose ball gets past the racket's
    slab test copies it for itself, without a workgroup barrier (substep<LAZYTAB>, tb_device.hpp). Up to 16384 envs workgroups are one
    wave; here they are two and four waves (TbOptions.block = 128 / 256, and the automatic 128 above 16384 envs) that copy into and
    read from the SAME LDS table, with balls past the slab during the short steps in most workgroups and batch sizes that end in a
    partial wave and a partial workgroup -- the variant no test reached before round 4."""
    thrown_at_racket_through_the_short_steps(torch, n, n_edges, options)


@pytest.mark.parametrize("n_edges", [3, 4, 6, 38, 63])
def test_tennisbot_rackets_with_other_outlines(torch, n_edges):
    """the same for Tennisbot, whose step kernel reads the outline table from memory instead of an LDS copy: rackets of random scale and
    attitude with another outline, balls thrown at them, a batch that ends in a partial wave -- 40 steps bit for bit"""
    n = 3000
    rng = np.random.default_rng(177 + n_edges)
    p = with_outline(default_params(), n_edges)
    from tennisbot_rl_amd.stepper import BatchedEnv
    env = BatchedEnv(ENV_TENNIS, n, device="cuda:0", seed=11, params=p, auto_reset=True)
    pf = p.copy(); pf.flags |= F_AUTO_RESET
    ref = OracleBatch(pf, ENV_TENNIS, n, seed=11, precision="f32")
    ref.L.tbo_set_threads(ref.h, 16)
    scale = rng.uniform(1.0, 3.0, n)
    q = rng.normal(size=(n, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    rp = np.stack([rng.uniform(8, 12, n), rng.uniform(-4, 4, n), rng.uniform(1.0, 2.5, n)], 1)
    rv = rng.uniform(-1.0, 1.0, (n, 3))
    u = rng.normal(size=(n, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
    bp = rp + u * (rng.uniform(0.1, 0.9, n) * scale)[:, None]
    bp[:, 2] = np.maximum(bp[:, 2], 0.3)
    tof = rng.uniform(0.02, 0.15, n)
    bv = (rp + rv * tof[:, None] - bp) / tof[:, None] + np.array([0.0, 0.0, 0.5 * 9.81])[None, :] * tof[:, None] + rng.normal(scale=0.3, size=(n, 3))
    w, d = make_words(ENV_TENNIS, n, racket_pos=rp, racket_quat=q, racket_vel=rv, racket_angvel=rng.uniform(-6, 6, (n, 3)), ball_pos=bp, ball_vel=bv,
                      ball_angvel=rng.uniform(-30, 30, (n, 3)), shoot_force=(30, 0, 20), step_count=50, racket_scale=scale)
    env.set_state_words(w.view(np.int32), d); ref.set_state_words(w, d)
    for t in range(40):
        a = rng.uniform(-1, 1, (n, 2)).astype(np.float32)
        obs, rew, done = env.step(torch.from_numpy(a).cuda())
        o2, r2, d2, s2 = ref.step(a)
        same(done.cpu().numpy(), d2, "done %d" % t); same(obs.cpu().numpy(), o2, "obs %d" % t); same(rew.cpu().numpy(), r2, "reward %d" % t)
        if t % 8 == 7:
            compare_state(env, ref, "state %d" % t)
    got, want = env.counters(), ref.counters()
    assert list(got.values()) == [int(x) for x in want], (got, want)
    assert got["racket_ball_contact_substeps"] > n // 20 and got["nonfinite_states"] == 0
    env.close()


@pytest.mark.parametrize("kind", [ENV_SWING, ENV_TENNIS])
def test_nonfinite_states_are_counted_like_the_oracle_counts_them(torch, kind):
    """the nonfinite_states counter with something to count: an infinity or a NaN injected into one of the 22 state values of every
    tenth env (each value in turn, both signs), then stepped -- the kernels' one-comparison test (x * 0 summed: NaN for any
    non-finite x) must give the verdict of the oracle's 22 isfinite() calls, step by step, without auto-reset (the state stays)"""
    n = 660
    env, ref = make_pair(torch, kind, n, auto_reset=False)
    rng = np.random.default_rng(5)
    same(env.reset().cpu().numpy(), ref.reset(), "reset obs")
    A = env.act_dim
    for t in range(3):
        a = rng.uniform(-1, 1, (n, A)).astype(np.float32)
        env.step(torch.from_numpy(a).cuda()); ref.step(a)
    w, d = env.get_state_words()
    w = w.cpu().numpy().view(np.uint32).copy(); d = d.cpu().numpy().copy()
    bad = [np.float32(np.inf), np.float32(-np.inf), np.float32(np.nan)]
    poisoned = 0
    for i in range(0, n, 10):
        row, val = (i // 10) % 22, bad[(i // 10) % 3]
        w[row, i] = val.view(np.uint32)
        poisoned += 1
    env.set_state_words(w.view(np.int32), d); ref.set_state_words(w, d)
    for t in range(2):
        a = rng.uniform(-1, 1, (n, A)).astype(np.float32)
        env.step(torch.from_numpy(a).cuda()); ref.step(a)
        got, want = env.counters(), [int(x) for x in ref.counters()]
        assert list(got.values()) == want, (t, got, want)
    assert got["nonfinite_states"] >= poisoned  # (every poisoned env is still non-finite one step later: counted in both steps)
    env.close()


def test_pipeline_form_follows_size_flags_and_marks(torch):
    """tb_pipeline_form: what TbOptions.ff_defer = 0 (auto) resolves to -- every episode end into the pool up to 16384 envs, the
    stragglers only above that with racket<->court contact, plain slots otherwise"""
    from tennisbot_rl_amd.stepper import BatchedEnv
    rgp = default_params(flags=F_DEFAULT | F_RACKET_GROUND)
    cases = [(ENV_SWING, 1000, None, {}, True, "pool"), (ENV_SWING, 16384, rgp, {}, True, "pool"), (ENV_SWING, 16385, None, {}, True, "slots"),
             (ENV_SWING, 16385, rgp, {}, True, "slots+pool"), (ENV_SWING, 1000, None, dict(ff_defer=False), True, "slots"),
             (ENV_SWING, 1000, None, dict(ff_defer=True), True, "slots+pool"), (ENV_SWING, 40000, None, dict(ff_defer="all"), True, "pool"),
             (ENV_SWING, 1000, None, {}, False, "none"), (ENV_TENNIS, 1000, None, {}, False, "none")]
    for kind, n, p, opts, piped, want in cases:
        env = BatchedEnv(kind, n, seed=1, params=p, pipeline=piped, track_terminal_obs=False, options=opts)
        assert env.pipeline_form() == want, (kind, n, opts, piped, env.pipeline_form(), want)
        if want in ("pool", "slots+pool"):  # a mark promises final steps: nothing is deferred past it unless the pool form was asked for (it runs at each mark)
            assert env.L.tb_mark_enable(env._h, 1) == 0
            assert env.pipeline_form() == ("pool" if opts.get("ff_defer") == "all" else "slots")
            assert env.L.tb_mark_enable(env._h, 0) == 0
            assert env.pipeline_form() == want
        env.close()


def test_pool_is_allocated_when_set_params_turns_racket_ground_on(torch):
    """above 16384 envs a handle gets no pool (14 KB per env) unless its parameter block asks for racket<->court contact -- and a
    handle that starts without the flag gets the pool from the tb_set_params call that turns it on (round 4: lazy allocation,
    ADVICE r03). 20000 envs: slots form, then set_params -> slots + pool, and two episodes in lockstep with the oracle through it."""
    from tennisbot_rl_amd.stepper import BatchedEnv
    n = 20000
    env = BatchedEnv(ENV_SWING, n, device="cuda:0", seed=4, pipeline=True, track_terminal_obs=False)
    assert env.pipeline_form() == "slots"
    p = default_params(flags=F_DEFAULT | F_RACKET_GROUND)
    env.set_params(p)
    assert env.pipeline_form() == "slots+pool"
    pf = p.copy(); pf.flags |= F_AUTO_RESET
    ref = OracleBatch(pf, ENV_SWING, n, seed=4, precision="f32")
    ref.L.tbo_set_threads(ref.h, 16)
    rng = np.random.default_rng(8)
    same(env.reset().cpu().numpy(), ref.reset(), "reset obs")
    outs = []
    for t in range(52):
        a = rng.uniform(-1, 1, (n, 6)).astype(np.float32)
        obs, rew, done = env.step(torch.from_numpy(a).cuda())
        o2, r2, d2, s2 = ref.step(a)
        same(done.cpu().numpy(), d2, "done %d" % t); same(obs.cpu().numpy(), o2, "obs %d" % t)
        outs.append((rew, r2))
    env.flush()
    for t, (rew, r2) in enumerate(outs):
        same(rew.cpu().numpy(), r2, "reward %d" % t)
    got, want = env.counters(), ref.counters()
    assert list(got.values()) == [int(x) for x in want], (got, want)
    env.close()


@pytest.mark.parametrize("rg,n,episodes,margin", [(False, 4096, 3, 0), (True, 2048, 3, 0), (False, 64, 700, 1), (True, 1000, 2, 200),
                                                  (False, 4096, 3, "all"), (True, 1000, 3, "all"), (False, 256, 70, "all"), (False, 8192, 17, "all")])
def test_deferred_stragglers_are_bit_identical(torch, rg, n, episodes, margin):
    """TbOptions.ff_defer: envs still running after their ballistic estimate + margin substeps leave their episode's fast-forward
    kernel for a pool that ONE launch finishes at the join. Against an unpipelined twin: every reward, done flag, observation,
    counter and the state bit for bit -- eagerly and through a replayed graph; with racket<->court contact (where it is the default);
    and over 700 episodes without a join, so that the pool (64 n records) fills up and later stragglers finish in place."""
    from tennisbot_rl_amd.rollout import RolloutBuffer
    from tennisbot_rl_amd.stepper import BatchedEnv
    T = 26 * episodes
    p = default_params(flags=F_DEFAULT | (F_RACKET_GROUND if rg else 0))
    rng = np.random.default_rng(31 + n)
    # margin "all": TbOptions.ff_defer = 2, every parked env straight into the pool (70 episodes without a join: more than its 64 regions)
    opts = dict(ff_defer="all") if margin == "all" else dict(ff_defer=True, ff_defer_margin=margin)
    a = BatchedEnv(ENV_SWING, n, seed=4, params=p, pipeline=True, track_terminal_obs=False, options=opts)
    b = BatchedEnv(ENV_SWING, n, seed=4, params=p)
    ba, bb = RolloutBuffer(ENV_SWING, T, n, "cuda:0").bind(a), RolloutBuffer(ENV_SWING, T, n, "cuda:0")
    acts = torch.from_numpy(rng.uniform(-1, 1, (min(T, 104), n, 6)).astype(np.float32)).cuda()
    for t in range(T):
        ba.actions[t].copy_(acts[t % acts.shape[0]]); bb.actions[t].copy_(acts[t % acts.shape[0]])
    a.reset(); b.reset()
    ba.step_range(a, 0, T)  # no join in between: the pool collects the stragglers of every episode
    a.flush()
    for t in range(T):
        bb.step_into(b, t)
    torch.cuda.synchronize()
    assert torch.equal(ba.rewards, bb.rewards) and torch.equal(ba.dones, bb.dones) and torch.equal(ba.obs, bb.obs)
    ca, cb = a.counters(), b.counters()
    assert ca == cb and ca["lockstep_violations"] == 0 and ca["episodes_finished"] == n * episodes
    # (8192 envs x 17 episodes: 139 264 records in the pool -- its run then takes the large-batch instantiation)
    if episodes <= 3:  # the same rollout once more as a replayed graph (the pool run is its last kernel node)
        g = a.capture(lambda: ba.step_range(a, 0, T))
        for rep in range(2):
            g.replay()
            for t in range(T):
                bb.step_into(b, t)
            torch.cuda.synchronize()
            assert torch.equal(ba.rewards, bb.rewards) and torch.equal(ba.obs, bb.obs)
        assert a.counters() == b.counters()
    wa, da = a.get_state_words(); wb, db = b.get_state_words()
    assert torch.equal(wa, wb) and torch.equal(da, db)
    a.close(); b.close()

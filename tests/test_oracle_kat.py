"""Known-answer tests that pin the CPU restatement (oracle/tb_oracle.c).

The reference ships no tests or golden vectors and PyBullet is absent,
so the oracle is pinned by (a) closed forms that do not depend on recalled Bullet
constants, (b) the published Philox4x32-10 vectors, (c) the
geometry constants derived from the reference's asset files (Appendix C) and (d) the
env logic read off the reference sources (Appendix A). PyBullet parity itself stays
"parity unpinned".
"""
import math

import numpy as np
import pytest

from helpers import make_words
from oracle import OracleBatch, philox4x32, query_box, query_goal, query_racket, query_racket_ground
from tennisbot_rl_amd.params import ENV_SWING, ENV_TENNIS, F_AUTO_RESET, F_DEFAULT, F_RACKET_GROUND, bullet_shape_inertia, default_params, load_scene, reference_rolling_friction, urdf_file_inertia

DT = 1.0 / 240.0
G = 9.81
FAR = (0.0, 3.0, 50.0)  # ball parked where it touches nothing


def batch(kind, n=1, prec="f64", flags=F_DEFAULT, **over):
    return OracleBatch(default_params(flags=flags, **over), kind, n, seed=3, precision=prec)


# ---------------------------------------------------------------- RNG
def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32 10 rounds
    kats = [
        ([0, 0, 0, 0], [0, 0], [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]),
        ([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2, [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]),
        ([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0],
         [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]),
    ]
    for prec in ("f32", "f64"):
        for ctr, key, want in kats:
            assert list(philox4x32(ctr, key, prec)) == want


# ---------------------------------------------------------------- geometry (Appendix C)
def test_asset_constants():
    sc = load_scene()
    rk = sc["racket"]
    assert rk["stl_triangles"] == 240 and rk["stl_unique_vertices"] == 122 and rk["outline_points"] == 61
    assert len(rk["hull_yz_ccw"]) == 38
    assert rk["hull_area"] == pytest.approx(0.13571, abs=5e-6)
    assert rk["half_thickness"] == pytest.approx(0.0145, abs=1e-7)
    assert rk["bbox_max"][2] == pytest.approx(0.69715, abs=1e-5)
    assert rk["bbox_max"][1] == pytest.approx(0.15079, abs=1e-5)
    assert rk["mass"] == 4.0 and rk["inertia_diag"] == [0.04, 0.08, 0.12] and rk["inertial_origin"] == [0, 0, 0.5]
    assert sc["ball"] == {"mass": 0.05, "inertia_diag": [1.0, 1.0, 1.0], "radius": 0.0335}
    assert sc["court"]["ground_box_size"] == [28.0, 14.0, 0.01]
    assert sc["court"]["net_box_size"] == [0.194382, 12.6506, 1.0]
    assert sc["goal"] == {"radius": 1.5, "length": 0.25}
    h = np.array(rk["hull_yz_ccw"])
    per = np.linalg.norm(np.roll(h, -1, 0) - h, axis=1).sum()
    assert per == pytest.approx(1.6369, abs=1e-4)


def test_hull_table_is_ccw_convex_and_scale_is_per_env():
    p = default_params()
    v = p.hull_vertices()
    e = np.roll(v, -1, 0) - v
    assert np.all(e[:, 0] * np.roll(e, -1, 0)[:, 1] - e[:, 1] * np.roll(e, -1, 0)[:, 0] > 0)  # convex, CCW
    # COM frame: link z in [0, .697] -> [-0.5, 0.197]
    assert v[:, 1].min() == pytest.approx(-0.5, abs=1e-6) and v[:, 1].max() == pytest.approx(0.6971475 - 0.5, abs=1e-6)
    # geometry is stored at scale 1; racket_scale only says what the next reset builds (tennisbot_env.py:230-234)
    p3 = default_params(racket_scale=2.3)
    assert np.array_equal(p3.hull_vertices(), v) and p3.racket_half_thick == p.racket_half_thick and p3.racket_scale == np.float32(2.3)
    # a scaled racket is the same shape 2.3x larger: face distance and rim distance scale, margin and radius do not
    hx, m, r = p.racket_half_thick, 0.001, 0.0335
    hit, d, n, rr = query_racket(p3, (0, 0, 0), (0, 0, 0, 1), (0.2, 0.0, 0.1))
    assert d == pytest.approx(0.2 - 2.3 * hx - m - r, abs=1e-6) and np.allclose(n, (1, 0, 0))
    top = (0.6971475 - 0.5) * 2.3
    hit, d, n, rr = query_racket(p3, (0, 0, 0), (0, 0, 0, 1), (0.0, 0.0, top + 0.05))
    assert d == pytest.approx(0.05 - m - r, abs=2e-6) and np.allclose(n, (0, 0, 1), atol=1e-6)
    # a ball that clears the unit racket's rim by 3 cm is inside the scaled one's outline: face contact
    hit1, d1, _, _ = query_racket(p, (0, 0, 0), (0, 0, 0, 1), (hx + m + r - 1e-4, 0.18, 0.1))
    hit3, d3, n3, _ = query_racket(p3, (0, 0, 0), (0, 0, 0, 1), (2.3 * hx + m + r - 1e-4, 0.18, 0.1))
    assert not hit1 and hit3 and d3 == pytest.approx(-1e-4, abs=2e-6) and np.allclose(n3, (1, 0, 0))


def test_sphere_vs_racket_face_edge_and_deep():
    p = default_params()
    r, m, hx = 0.0335, 0.001, p.racket_half_thick
    q = (0, 0, 0, 1)
    # in front of the +x face, inside the outline: distance is along x only
    hit, d, n, rr = query_racket(p, (0, 0, 0), q, (0.1, 0.0, 0.1))
    assert not hit and d == pytest.approx(0.1 - hx - m - r, abs=1e-7) and np.allclose(n, (1, 0, 0))
    hit, d, n, rr = query_racket(p, (0, 0, 0), q, (-(hx + m + r) + 1e-4, 0.02, 0.05))
    assert hit and d == pytest.approx(-1e-4, abs=1e-6) and np.allclose(n, (-1, 0, 0))
    assert np.allclose(rr, (-(hx + m), 0.02, 0.05), atol=2e-6)  # point on the inflated racket surface
    # above the top of the head (link z = .6971 -> COM frame .1971), centred: rim contact along +z
    top = 0.6971475 - 0.5
    hit, d, n, rr = query_racket(p, (0, 0, 0), q, (0.0, 0.0, top + 0.05))
    assert d == pytest.approx(0.05 - m - r, abs=1e-6) and np.allclose(n, (0, 0, 1), atol=1e-6)
    # diagonal: outside in x and above the rim -> Euclidean distance to the edge
    hit, d, n, rr = query_racket(p, (0, 0, 0), q, (hx + 0.03, 0.0, top + 0.04))
    assert d == pytest.approx(0.05 - m - r, abs=1e-6) and np.allclose(n, (0.6, 0, 0.8), atol=1e-5)
    # centre inside the prism: least-penetration axis is x (thickness .029 << outline)
    hit, d, n, rr = query_racket(p, (0, 0, 0), q, (0.004, 0.0, 0.0))
    assert hit and d == pytest.approx((0.004 - hx) - m - r, abs=1e-7) and np.allclose(n, (1, 0, 0))
    # rotated + translated racket gives the same answer in its own frame
    qy = (0, math.sin(0.25), 0, math.cos(0.25))  # pitch 0.5 rad as in swingracket_env.py:167
    R = np.array([[math.cos(.5), 0, math.sin(.5)], [0, 1, 0], [-math.sin(.5), 0, math.cos(.5)]])
    c = np.array((5.0, 1.0, 2.0)) + R @ np.array((0.1, 0.0, 0.1))
    hit, d, n, rr = query_racket(p, (5.0, 1.0, 2.0), qy, c)
    assert d == pytest.approx(0.1 - hx - m - r, abs=2e-6) and np.allclose(n, R @ np.array((1, 0, 0)), atol=1e-6)
    # far away: culled
    assert query_racket(p, (0, 0, 0), q, (3.0, 0, 0))[0] is False


@pytest.mark.parametrize("n_edges", [None, 3, 5, 8, 64])
def test_sphere_vs_racket_matches_bruteforce(n_edges):
    """distance to the prism == brute-force minimum over a dense sampling of its surface; the racket's own outline (38 edges from
    racket.stl) and the synthetic convex ones the GPU parity tests use (tests/outlines.py)"""
    p = default_params()
    if n_edges is not None:
        from outlines import with_outline
        p = with_outline(p, n_edges)
    v = p.hull_vertices()
    hx = p.racket_half_thick
    t = np.linspace(0, 1, 400)[:, None]
    rim2 = np.concatenate([v[i] + t * (np.roll(v, -1, 0)[i] - v[i]) for i in range(len(v))])

    def inside(pt):
        e = np.roll(v, -1, 0) - v
        w = pt - v
        return np.all(e[:, 0] * w[:, 1] - e[:, 1] * w[:, 0] >= 0)
    rng = np.random.default_rng(1)
    checked = 0
    for _ in range(400):
        c = (rng.uniform(-1, 1, 3) * (0.15, 0.35, 0.6) + (0, 0, -0.12)).astype(np.float32).astype(np.float64)
        if np.linalg.norm(c) > p.hull_bound_radius:
            continue  # beyond the bounding sphere the query is culled (no distance reported)
        checked += 1
        hit, d, n, rr = query_racket(p, (0, 0, 0), (0, 0, 0, 1), c)
        d2 = np.min(np.linalg.norm(rim2 - c[1:], axis=1))
        ins = inside(c[1:])
        ax = abs(c[0]) - hx
        if ins and ax > 0:
            want = ax
        elif ins:
            continue  # deep case covered above
        else:
            want = math.hypot(max(ax, 0.0), d2)
        assert d + 0.001 + 0.0335 == pytest.approx(want, abs=2e-4)  # sampling resolution
    assert checked > 100


def test_sphere_vs_box_and_goal():
    p = default_params()
    r = 0.0335
    g = tuple(p.ground_half)
    hit, d, n = query_box(p, g, (1.0, 2.0, 0.005 + r + 0.0005))
    assert hit and d == pytest.approx(0.0005, abs=1e-7) and np.allclose(n, (0, 0, 1))
    hit, d, n = query_box(p, g, (1.0, 2.0, 0.005 + r + 0.001))
    assert not hit  # beyond the 0.67 mm manifold threshold
    hit, d, n = query_box(p, g, (1.0, 2.0, 0.005 + r - 0.01))
    assert hit and d == pytest.approx(-0.01, abs=1e-7)
    e = r + 0.0002  # edge region: Euclidean distance to the box edge, normal along the offset
    hit, d, n = query_box(p, g, (14.0 + 0.6 * e, 0.0, 0.005 + 0.8 * e))
    assert hit and d == pytest.approx(0.0002, abs=2e-6) and np.allclose(n, (0.6, 0, 0.8), atol=1e-4)
    hit, d, n = query_box(p, g, (14.0 + 0.6 * (r + 0.001), 0.0, 0.005 + 0.8 * (r + 0.001)))
    assert not hit  # per-axis separations are inside the threshold, the Euclidean one is not
    assert not query_box(p, g, (20.0, 0.0, 0.02))[0]  # off the court: nothing to land on
    nh = tuple(p.net_half)
    hit, d, n = query_box(p, nh, (-(nh[0] + r) + 0.002, 1.0, 0.3))
    assert hit and d == pytest.approx(-0.002, abs=1e-6) and np.allclose(n, (-1, 0, 0))
    hit, d, n = query_box(p, nh, (0.0, 0.0, 0.5 + r + 0.0003))
    assert hit and np.allclose(n, (0, 0, 1))
    # goal: top face at z = 0.125 (simplegoal.urdf length .25 centred on the link origin)
    hit, d, n = query_goal(p, -6.0, 1.0, (-6.5, 1.2, 0.125 + r + 0.0002))
    assert hit and d == pytest.approx(0.0002, abs=1e-6) and np.allclose(n, (0, 0, 1))
    hit, d, n = query_goal(p, -6.0, 1.0, (-6.0 + 1.5 + r + 0.0001, 1.0, 0.05))
    assert hit and d == pytest.approx(0.0001, abs=2e-6) and np.allclose(n, (1, 0, 0), atol=1e-6)
    assert not query_goal(p, -6.0, 1.0, (-6.0 + 1.6, 1.0, 0.05))[0]
    hit, d, n = query_goal(p, 0.0, 0.0, (1.5 + 0.6 * e, 0.0, 0.125 + 0.8 * e))  # rim
    assert hit and d == pytest.approx(0.0002, abs=2e-6) and np.allclose(n, (0.6, 0, 0.8), atol=1e-4)


# ---------------------------------------------------------------- closed forms (Appendix B.3)
@pytest.mark.parametrize("prec,tol", [("f64", 1e-9), ("f32", 2e-5)])
def test_free_fall_and_constant_force(prec, tol):
    b = batch(ENV_SWING, prec=prec, lin_damp=0.0, ang_damp=0.0)
    w, d = make_words(ENV_SWING, 1, racket_pos=(8, 0, 1), ball_pos=FAR, goal=(-6, 0), spawn_pos=(8, 0, 0.6), init_dist=10.0)
    b.set_state_words(w, d)
    k = 25
    a = np.array([[1, 0, 0, 0, 0, 0]], np.float32)
    for _ in range(k):
        b.step(a)
    s = b.get_state()
    # ball: v_k = -g k dt, p_k = p0 - g dt^2 k(k+1)/2
    assert s["ball_vel"][0, 2] == pytest.approx(-G * k * DT, rel=tol)
    assert s["ball_pos"][0, 2] - 50.0 == pytest.approx(-G * DT * DT * k * (k + 1) / 2, rel=50 * tol)
    # racket: a_x = 400/4 = 100 -> v = 10.4167, dx = 0.564236 ; z exactly hovering
    assert s["racket_vel"][0, 0] == pytest.approx(100 * k * DT, rel=tol)
    assert s["racket_pos"][0, 0] - 8.0 == pytest.approx(0.564236, abs=2e-6)
    assert s["racket_vel"][0, 2] == 0.0 and s["racket_pos"][0, 2] == 1.0
    assert s["step_count"][0] == k


@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_hover_is_exact(prec):
    """39.24 - 4*9.81 cancels exactly (swingracket_env.py:77, tennisbot_env.py:112): a zero
    action leaves the racket bit-for-bit at rest, also in float32."""
    for kind, A in ((ENV_SWING, 6), (ENV_TENNIS, 2)):
        b = batch(kind, prec=prec)
        w, d = make_words(kind, 1, racket_pos=(9.5, 0.25, 0.7), ball_pos=FAR)
        b.set_state_words(w, d)
        for _ in range(20):
            b.step(np.zeros((1, A), np.float32))
        s = b.get_state()
        assert np.all(s["racket_vel"] == 0) and np.all(s["racket_angvel"] == 0)
        assert np.array_equal(s["racket_pos"][0], np.float32((9.5, 0.25, 0.7)).astype(np.float64))


def test_pure_torque_and_quaternion():
    b = batch(ENV_SWING, lin_damp=0.0, ang_damp=0.0, **urdf_file_inertia())  # the closed forms below use round numbers
    w, d = make_words(ENV_SWING, 1, racket_pos=(8, 0, 1), ball_pos=FAR)
    b.set_state_words(w, d)
    a = np.array([[0, 0, 0, 0, 1, 0]], np.float32)  # T_y = 5 on I_yy = 0.08 -> 62.5 rad/s^2
    k = 12
    for _ in range(k):
        b.step(a)
    s = b.get_state()
    assert s["racket_angvel"][0, 1] == pytest.approx(62.5 * k * DT, rel=1e-9)
    assert s["racket_angvel"][0, 0] == 0 and s["racket_angvel"][0, 2] == 0
    theta = 62.5 * DT * DT * k * (k + 1) / 2  # exponential map about a fixed axis adds angles
    q = s["racket_quat"][0]
    assert np.allclose(q, (0, math.sin(theta / 2), 0, math.cos(theta / 2)), atol=1e-7)
    assert np.linalg.norm(q) == pytest.approx(1.0, abs=1e-7)


def test_rotation_clamp_and_small_angle_branch():
    b = batch(ENV_SWING, lin_damp=0.0, ang_damp=0.0)
    # |w| dt > pi/4 -> the step rotates by exactly pi/4 about the axis, scaled by |w|/clamp
    w0 = 400.0
    w, d = make_words(ENV_SWING, 1, racket_pos=(8, 0, 5), racket_angvel=(0, 0, w0), ball_pos=FAR)
    b.set_state_words(w, d)
    b.step(np.zeros((1, 6), np.float32))
    q = b.get_state()["racket_quat"][0]
    ang = (math.pi / 4) / DT
    s = math.sin(0.5 * ang * DT) / ang
    raw = np.array((0, 0, w0 * s, math.cos(math.pi / 8)))
    assert np.allclose(q, raw / np.linalg.norm(raw), atol=1e-7)
    # tiny |w| uses the Taylor branch: rotation angle = |w| dt to first order
    w, d = make_words(ENV_SWING, 1, racket_pos=(8, 0, 5), racket_angvel=(5e-4, 0, 0), ball_pos=FAR)
    b.set_state_words(w, d)
    b.step(np.zeros((1, 6), np.float32))
    q = b.get_state()["racket_quat"][0]
    assert q[0] == pytest.approx(0.5 * np.float32(5e-4) * DT, rel=1e-6) and q[3] == pytest.approx(1.0, abs=1e-12)


def test_gyroscopic_term_conserves_nothing_but_matches_formula():
    """one explicit step of w' = w + dt R I^-1 (-(w_b x I w_b)) at identity orientation"""
    b = batch(ENV_SWING, lin_damp=0.0, ang_damp=0.0, **urdf_file_inertia())
    w0 = np.float32((3.0, -2.0, 1.5)).astype(np.float64)
    w, d = make_words(ENV_SWING, 1, racket_pos=(8, 0, 5), racket_angvel=w0, ball_pos=FAR)
    b.set_state_words(w, d)
    b.step(np.zeros((1, 6), np.float32))
    I = np.array((0.04, 0.08, 0.12))
    want = w0 + DT * (-(np.cross(w0, I * w0)) / I)
    got = b.get_state()["racket_angvel"][0]
    assert np.allclose(got, want, rtol=1e-12)


def test_damping_one_step():
    b = batch(ENV_TENNIS)
    v0 = np.float32((15.0, -2.0, 3.0)).astype(np.float64)
    w, d = make_words(ENV_TENNIS, 1, racket_pos=(10, 0, 0.7), ball_pos=FAR, ball_vel=v0, step_count=10)
    b.set_state_words(w, d)
    b.step(np.zeros((1, 2), np.float32))
    kd = 0.04 + 0.04 * np.linalg.norm(v0)
    want = v0 + DT * (-v0 * kd + np.array((0, 0, -9.81)))
    assert np.allclose(b.get_state()["ball_vel"][0], want, rtol=1e-12, atol=1e-12)
    # ~9.6 m/s^2 of drag at 15 m/s (SURVEY.md B.2 "material for a 15 m/s ball")
    assert (v0[0] - b.get_state()["ball_vel"][0, 0]) / DT == pytest.approx(0.04 * (1 + np.linalg.norm(v0)) * 15, rel=1e-6)


def test_damping_terms_are_separate_parameters():
    """k1 and k2 of -m v (k1 + k2 |v|) are one value in Bullet (0.04) and two TbParams fields here: k2 = 0 leaves pure linear
    drag -- v_k = v0 (1 - k1 dt)^k without gravity, a closed form -- and k1 = 0 the speed-proportional term alone"""
    v0 = np.float32((15.0, -2.0, 3.0)).astype(np.float64)
    w, d = make_words(ENV_TENNIS, 1, racket_pos=(10, 0, 0.7), ball_pos=FAR, ball_vel=v0, step_count=10)
    b = batch(ENV_TENNIS, gravity=0.0, lin_damp_quad=0.0, ang_damp_quad=0.0)
    b.set_state_words(w, d)
    for _ in range(10):
        b.step(np.zeros((1, 2), np.float32))
    assert np.allclose(b.get_state()["ball_vel"][0], v0 * (1.0 - 0.04 * DT) ** 10, rtol=1e-12)
    b = batch(ENV_TENNIS, gravity=0.0, lin_damp=0.0, lin_damp_quad=0.04)
    b.set_state_words(w, d)
    b.step(np.zeros((1, 2), np.float32))
    assert np.allclose(b.get_state()["ball_vel"][0], v0 * (1.0 - DT * 0.04 * np.linalg.norm(v0)), rtol=1e-12)
    p = default_params(lin_damp=0.07)
    assert p.lin_damp_quad == p.lin_damp == np.float32(0.07) and p.ang_damp_quad == p.ang_damp  # unset: k2 follows k1, as in Bullet


def test_tennis_shoot_pulse():
    """5 substeps of F=(30,0,20) on 0.05 kg (tennisbot_env.py:118-119): v = (12.5, 0, 8.1290)"""
    b = batch(ENV_TENNIS, lin_damp=0.0, ang_damp=0.0)
    w, d = make_words(ENV_TENNIS, 1, racket_pos=(10, 0, 0.7), ball_pos=(-9, 0, 1.2), shoot_force=(30, 0, 20))
    b.set_state_words(w, d)
    for i in range(7):
        obs, rew, done, sub = b.step(np.zeros((1, 2), np.float32))
        if i == 4:
            v = b.get_state()["ball_vel"][0]
            assert v[0] == pytest.approx(12.5, rel=1e-7) and v[1] == 0
            assert v[2] == pytest.approx((400 - 9.81) * 5 / 240, rel=1e-6)
    v = b.get_state()["ball_vel"][0]
    assert v[0] == pytest.approx(12.5, rel=1e-7)  # pulse is over after 5 substeps
    assert v[2] == pytest.approx((400 - 9.81) * 5 / 240 - 2 * 9.81 / 240, rel=1e-6)


def _impact_expected(vn, dist, e, mb=0.05, mr=4.0, thr=0.2, erp=float(np.float32(0.08))):
    """normal-row closed form for a contact through both COMs (no angular coupling);
    vn = relative velocity along the normal that points from the racket to the ball"""
    rest = 0.0 if abs(vn) < thr else max(e * -vn, 0.0)
    pos = -dist * 240.0 if dist > 0 else -dist * erp * 240.0
    j = max((rest + pos - vn) / (1 / mb + 1 / mr), 0.0)
    return j


def test_head_on_impact_through_com():
    """ball hits the racket face on the COM axis: j = (1+e) v_n / (1/m_b + 1/m_r) (Appendix B.3)"""
    p = default_params(lin_damp=0.0, ang_damp=0.0, gravity=0.0)
    b = OracleBatch(p, ENV_TENNIS, 1, precision="f64")
    hx, m, r = p.racket_half_thick, p.hull_margin, p.ball_radius
    gap = 0.0003  # inside the manifold threshold, still separated
    x_ball = 10.0 - (hx + m + r + gap)
    w, d = make_words(ENV_TENNIS, 1, racket_pos=(10, 0, 1.0), ball_pos=(x_ball, 0, 1.0), ball_vel=(12.0, 0, 0), step_count=50)
    b.set_state_words(w, d)
    obs, rew, done, sub = b.step(np.zeros((1, 2), np.float32))
    s = b.get_state()
    dist = 10.0 - np.float32(x_ball).astype(np.float64) - (float(hx) + 0.001 + 0.0335)
    # hover force: racket F_z = 39.24 but gravity is 0 in this test -> a_z = 9.81; x untouched.
    # normal = -x (from the racket toward the ball), approach speed 12 -> vn = -12
    j = _impact_expected(-12.0, dist, 0.81)
    vb, vr = 12.0 - j / 0.05, 0.0 + j / 4.0
    assert s["ball_vel"][0, 0] == pytest.approx(vb, rel=1e-6)
    assert s["racket_vel"][0, 0] == pytest.approx(vr, rel=1e-6)
    assert vb < 0 and vr > 0
    # momentum is conserved by the impulse pair
    assert 0.05 * s["ball_vel"][0, 0] + 4.0 * s["racket_vel"][0, 0] == pytest.approx(0.05 * 12.0, rel=1e-6)
    # textbook value when the speculative gap is negligible
    assert j == pytest.approx((1 + 0.81) * 12.0 / (1 / 0.05 + 1 / 4.0), rel=0.02)
    # through the COM: the normal impulse gives no spin; the racket rises at 9.81 dt (hover force,
    # g = 0 here) so friction at the face, 0.0155 m off the COM, leaves a tiny w_y only
    wr = s["racket_angvel"][0]
    assert wr[0] == 0 and wr[2] == 0 and abs(wr[1]) < 1e-3
    assert rew[0] == 25 + 20  # contact reward + closest tier (tennisbot_env.py:170-174,90-92)


def test_ground_bounce_restitution_and_threshold():
    p = default_params(lin_damp=0.0, ang_damp=0.0)
    b = OracleBatch(p, ENV_TENNIS, 1, precision="f64")
    z = 0.005 + 0.0335 + 0.0002
    w, d = make_words(ENV_TENNIS, 1, racket_pos=(10, 0, 1.0), ball_pos=(-5, 0, z), ball_vel=(0, 0, -3.0), step_count=50)
    b.set_state_words(w, d)
    b.step(np.zeros((1, 2), np.float32))
    vz = b.get_state()["ball_vel"][0, 2]
    dist = np.float32(z).astype(np.float64) - (0.005 + 0.0335)
    vn = -3.0 - 9.81 * DT
    want = 0.81 * -vn - dist * 240.0
    assert vz == pytest.approx(want, rel=1e-6)
    # slow approach (< 0.2 m/s): no restitution, the ball is just stopped at the surface
    w, d = make_words(ENV_TENNIS, 1, racket_pos=(10, 0, 1.0), ball_pos=(-5, 0, 0.005 + 0.0335 - 0.001), ball_vel=(0, 0, -0.05), step_count=50)
    b.set_state_words(w, d)
    b.step(np.zeros((1, 2), np.float32))
    vz = b.get_state()["ball_vel"][0, 2]
    pen = np.float32(0.005 + 0.0335 - 0.001).astype(np.float64) - (0.005 + 0.0335)
    assert vz == pytest.approx(-pen * 0.08 * 240.0, rel=1e-5)  # Baumgarte push-out only (contact ERP 0.08)


def test_friction_is_bounded_by_mu_times_normal_impulse():
    p = default_params(lin_damp=0.0, ang_damp=0.0, **urdf_file_inertia())
    b = OracleBatch(p, ENV_TENNIS, 1, precision="f64")
    w, d = make_words(ENV_TENNIS, 1, racket_pos=(10, 0, 1.0), ball_pos=(-5, 0, 0.005 + 0.0335 + 0.0001), ball_vel=(6.0, 0, -3.0), step_count=50)
    b.set_state_words(w, d)
    b.step(np.zeros((1, 2), np.float32))
    s = b.get_state()
    jn = 0.05 * (s["ball_vel"][0, 2] - (-3.0 - 9.81 * DT))
    jt = 0.05 * (6.0 - s["ball_vel"][0, 0])
    assert jt == pytest.approx(0.2 * 0.2 * jn, rel=1e-6)  # sliding: saturated at mu j_n
    # ball inertia 1.0 (ball.urdf:15) -> the friction impulse barely spins it: dw = r j_t / I
    assert s["ball_angvel"][0, 1] == pytest.approx(0.0335 * jt / 1.0, rel=1e-4)


def _ground_hit(p, spin, prec="f64"):
    b = OracleBatch(p, ENV_TENNIS, 1, precision=prec)
    w, d = make_words(ENV_TENNIS, 1, racket_pos=(10, 0, 1.0), ball_pos=(-5, 0, 0.005 + 0.0335 + 0.0001), ball_vel=(0, 0, -3.0),
                      ball_angvel=spin, step_count=50)
    b.set_state_words(w, d)
    b.step(np.zeros((1, 2), np.float32))
    s = b.get_state()
    jn = float(p.ball_mass) * (s["ball_vel"][0, 2] - (-3.0 - 9.81 * DT))
    return s, jn


def test_rolling_friction_rows_are_opt_in_and_boxed_by_roll_times_normal_impulse():
    """TbParams.roll_* (SURVEY.md 8f.3): two angular rows along the contact's tangent axes, each boxed by
    roll * j_n; the spin about the normal is not touched (no spinning friction in the reference scene).
    Sliding friction is switched off here so that the closed forms hold exactly."""
    assert reference_rolling_friction() == dict(roll_racket=pytest.approx(4e-4), roll_court=pytest.approx(4e-4), roll_goal=pytest.approx(5e-4))
    base = dict(lin_damp=0.0, ang_damp=0.0, fric_court=0.0)
    inv_i = 1.0 / bullet_shape_inertia()["ball_inertia"]
    # off (the default): the bounce leaves the spin alone
    s, jn = _ground_hit(default_params(**base), (40.0, -25.0, 7.0))
    assert tuple(s["ball_angvel"][0]) == (40.0, -25.0, 7.0) and default_params().roll_court == 0.0
    # saturated: fast spin loses exactly roll * j_n / I about each tangent axis
    p = default_params(roll_court=4e-4, **base)
    s, jn = _ground_hit(p, (40.0, -25.0, 7.0))
    assert jn > 0.25
    dw = 4e-4 * jn * inv_i
    assert 1.0 < dw < 25.0
    assert s["ball_angvel"][0, 0] == pytest.approx(40.0 - dw, rel=1e-6)
    assert s["ball_angvel"][0, 1] == pytest.approx(-25.0 + dw, rel=1e-6)
    assert s["ball_angvel"][0, 2] == 7.0
    assert s["ball_vel"][0, 0] == 0.0 and s["ball_vel"][0, 1] == 0.0  # angular-only rows
    # unsaturated: a slow tangent spin is stopped, not reversed
    s, jn = _ground_hit(p, (0.5 * dw, -0.25 * dw, 7.0))
    assert abs(s["ball_angvel"][0, 0]) < 1e-9 and abs(s["ball_angvel"][0, 1]) < 1e-9 and s["ball_angvel"][0, 2] == 7.0
    # the normal impulse itself does not depend on the rolling rows
    assert jn == pytest.approx(_ground_hit(default_params(**base), (0, 0, 0))[1], rel=1e-12)
    # f32 restatement agrees with f64 to rounding
    s32, _ = _ground_hit(p, (40.0, -25.0, 7.0), prec="f32")
    assert np.allclose(s32["ball_angvel"], _ground_hit(p, (40.0, -25.0, 7.0))[0]["ball_angvel"], rtol=2e-5)


def test_rolling_friction_on_the_racket_exchanges_angular_momentum():
    """ball <-> racket rolling rows apply equal and opposite angular impulses (friction off, hit through the COM)"""
    p = default_params(lin_damp=0.0, ang_damp=0.0, gravity=0.0, fric_racket=0.0, roll_racket=4e-4)
    b = OracleBatch(p, ENV_TENNIS, 1, precision="f64")
    hx, m, r = p.racket_half_thick, p.hull_margin, p.ball_radius
    x_ball = 10.0 - (hx + m + r + 0.0003)
    spin = (3.0, 200.0, -150.0)
    w, d = make_words(ENV_TENNIS, 1, racket_pos=(10, 0, 1.0), ball_pos=(x_ball, 0, 1.0), ball_vel=(12.0, 0, 0), ball_angvel=spin, step_count=50)
    b.set_state_words(w, d)
    b.step(np.zeros((1, 2), np.float32))
    s = b.get_state()
    jn = 0.05 * (12.0 - s["ball_vel"][0, 0])
    I_b = bullet_shape_inertia()["ball_inertia"]
    I_r = np.asarray(bullet_shape_inertia()["racket_inertia"])
    dLb = I_b * (s["ball_angvel"][0] - np.asarray(spin))
    dLr = I_r * s["racket_angvel"][0]  # identity orientation: body frame = world frame
    assert np.allclose(dLb + dLr, 0.0, atol=1e-9)  # inertia tables are stored in float32
    assert dLb[0] == 0.0  # about the normal (x): untouched
    # both tangent rows saturate: |dL| = roll * j_n each, opposing the ball's spin
    assert dLb[1] == pytest.approx(-4e-4 * jn, rel=1e-6) and dLb[2] == pytest.approx(4e-4 * jn, rel=1e-6)


# ---------------------------------------------------------------- reset distributions (Appendix A)
@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_swing_reset_distribution_and_obs(prec):
    n = 4096
    b = batch(ENV_SWING, n=n, prec=prec)
    obs = b.reset()
    s = b.get_state()
    sp = s["spawn_pos"]
    assert sp[:, 0].min() >= 5.5 and sp[:, 0].max() < 11 and abs(sp[:, 0].mean() - 8.25) < 0.1
    assert sp[:, 1].min() >= -4 and sp[:, 1].max() < 4 and abs(sp[:, 1].mean()) < 0.15
    assert np.all(sp[:, 2] == np.float32(0.6).astype(np.float64) if prec == "f32" else sp[:, 2] == 0.6)
    g = s["goal"]
    assert g[:, 0].max() <= -3 and g[:, 0].min() > -12 and g[:, 1].min() >= -5 and g[:, 1].max() < 5
    # COM = link + R(pitch .5)(0,0,.5): (x + .2397, y, 1.0388)
    assert np.allclose(s["racket_pos"][:, 0] - sp[:, 0], 0.5 * math.sin(0.5), atol=1e-6)
    assert np.allclose(s["racket_pos"][:, 2], 0.6 + 0.5 * math.cos(0.5), atol=1e-6)
    assert np.allclose(s["racket_quat"], (0, math.sin(0.25), 0, math.cos(0.25)), atol=1e-7)
    assert np.allclose(s["ball_pos"], np.stack([sp[:, 0] - 0.1, sp[:, 1], np.full(n, 1.4)], 1), atol=1e-6)
    assert np.allclose(s["init_dist"], np.linalg.norm(s["ball_pos"][:, :2] - g, axis=1), rtol=1e-6)
    assert np.all(s["step_count"] == 0) and np.all(s["done"] == 0) and np.all(s["episode"] == 0)
    want = np.concatenate([s["racket_pos"][:, :2], s["ball_pos"][:, :2], g], 1)
    assert np.allclose(obs, want, atol=1e-6)
    # declared observation_space (swingracket_env.py:34-39) contains every reset obs
    assert np.all(obs >= np.array([-20, -10, -20, -10, -15, -5])) and np.all(obs <= np.array([20, 10, 20, 10, 0, 5]))
    # a second reset starts a different episode (episode index keys the stream)
    obs2 = b.reset()
    assert not np.allclose(obs, obs2) and np.all(b.get_state()["episode"] == 1)


def test_tennis_reset_distribution():
    n = 4096
    b = batch(ENV_TENNIS, n=n)
    obs = b.reset()
    s = b.get_state()
    rp, bp, f = s["racket_pos"], s["ball_pos"], s["shoot_force"]
    assert rp[:, 0].min() >= 7.5 and rp[:, 0].max() < 12.5 and rp[:, 1].min() >= -5 and rp[:, 1].max() < 5
    assert rp[:, 2].min() >= 0.2 + 0.5 - 1e-6 and rp[:, 2].max() <= 0.21 + 0.5 + 1e-6  # COM z = link z + .5
    assert f[:, 0].min() >= 25 and f[:, 0].max() < 37.5 and f[:, 1].min() >= -10 and f[:, 1].max() < 10 and np.all(f[:, 2] == 20)
    assert bp[:, 0].min() >= -12 and bp[:, 0].max() < -6 and bp[:, 1].min() >= -1 and bp[:, 1].max() < 1
    assert bp[:, 2].min() >= 1 and bp[:, 2].max() < 1.5
    assert np.allclose(obs, np.concatenate([rp, s["racket_vel"], bp, s["ball_vel"]], 1), atol=1e-6)
    assert np.all(obs[:, 3:6] == 0) and np.all(obs[:, 9:] == 0)
    # racket scale moves the COM (tennisbot_env.py:234 globalScaling)
    b3 = OracleBatch(default_params(racket_scale=3.0), ENV_TENNIS, n, seed=3, precision="f64")
    b3.reset()
    assert np.allclose(b3.get_state()["racket_pos"][:, 2] - rp[:, 2], 0.5 * 2.0, atol=1e-6)


def test_reset_is_keyed_by_global_env_id():
    """sharding independence: env_id_base + i, not the local index, keys the stream"""
    p = default_params()
    whole = OracleBatch(p, ENV_SWING, 8, seed=9, precision="f32")
    a = whole.reset()
    lo = OracleBatch(p, ENV_SWING, 4, seed=9, env_id_base=0, precision="f32").reset()
    hi = OracleBatch(p, ENV_SWING, 4, seed=9, env_id_base=4, precision="f32").reset()
    assert np.array_equal(a, np.concatenate([lo, hi]))
    other = OracleBatch(p, ENV_SWING, 8, seed=10, precision="f32").reset()
    assert not np.array_equal(a, other)


# ---------------------------------------------------------------- env logic (Appendix A.1 / A.2 / D)
def test_swing_episode_is_always_26_agent_steps():
    n = 64
    b = batch(ENV_SWING, n=n, prec="f32")
    b.reset()
    rng = np.random.default_rng(0)
    for t in range(1, 27):
        obs, rew, done, sub = b.step(rng.uniform(-1, 1, (n, 6)).astype(np.float32))
        if t < 26:
            assert not done.any() and np.all(sub == 1)
        else:
            assert done.all() and np.all(sub >= 2) and np.all(sub <= 776)
    s = b.get_state()
    assert np.all(s["step_count"] == 25 + sub) and np.all(s["step_count"] <= 801)
    assert np.all(s["done"] == 1)  # restoring force pending (swingracket_env.py:135-141)


def test_swing_contact_bonus_window_and_quirks():
    """+2 only while step_count < 25 (swingracket_env.py:98); step 25 gets nothing (Appendix D.3)"""
    p = default_params()
    hx, m, r = p.racket_half_thick, p.hull_margin, p.ball_radius
    for sc, want in ((0, 2.0), (23, 2.0), (24, 0.0)):
        b = OracleBatch(p, ENV_SWING, 1, precision="f64")
        w, d = make_words(ENV_SWING, 1, racket_pos=(8, 0, 1.0), ball_pos=(8 - (hx + m + r) + 1e-4, 0, 1.0),
                          goal=(-6, 0), spawn_pos=(8, 0, 0.5), init_dist=14.0, step_count=sc)
        b.set_state_words(w, d)
        obs, rew, done, sub = b.step(np.zeros((1, 6), np.float32))
        assert rew[0] == want and done[0] == 0 and sub[0] == 1
        assert b.counters()[0] == 1


def test_swing_terminal_rewards():
    p = default_params()
    # ball resting just above the ground far from the goal: first fast-forward substep ends it
    b = OracleBatch(p, ENV_SWING, 1, precision="f64")
    w, d = make_words(ENV_SWING, 1, racket_pos=(8, 0, 1.0), ball_pos=(2.0, 0, 0.005 + 0.0335 + 0.0001),
                      goal=(-6, 0), spawn_pos=(8, 0, 0.5), init_dist=10.0, step_count=25)
    b.set_state_words(w, d)
    obs, rew, done, sub = b.step(np.zeros((1, 6), np.float32))
    assert done[0] == 1 and sub[0] == 2
    bx = b.get_state()["ball_pos"][0, 0]
    assert rew[0] == pytest.approx((10.0 - abs(bx + 6)) / 10.0 * 20, rel=1e-6)  # moved_dist_to_goal :63-73
    # on the goal disc: goal top (z=.125) is hit -> moved + 50 (:119-123); ground not touched
    b = OracleBatch(p, ENV_SWING, 1, precision="f64")
    w, d = make_words(ENV_SWING, 1, racket_pos=(8, 0, 1.0), ball_pos=(-5.5, 0.2, 0.125 + 0.0335 + 0.0001),
                      goal=(-6, 0), spawn_pos=(8, 0, 0.5), init_dist=10.0, step_count=25)
    b.set_state_words(w, d)
    obs, rew, done, sub = b.step(np.zeros((1, 6), np.float32))
    dist = np.linalg.norm(b.get_state()["ball_pos"][0, :2] - (-6, 0))
    assert done[0] == 1 and rew[0] == pytest.approx((10 - dist) / 10 * 20 + 50, rel=1e-6)
    c = b.counters()
    assert c[2] == 1 and c[1] == 0
    # timeout: ball off the court never lands (:127-128); 801 substeps in total
    b = OracleBatch(p, ENV_SWING, 1, precision="f64")
    w, d = make_words(ENV_SWING, 1, racket_pos=(8, 0, 1.0), ball_pos=(-20.0, 0, 1e4), goal=(-6, 0), spawn_pos=(8, 0, 0.5), init_dist=10.0, step_count=25)
    b.set_state_words(w, d)
    obs, rew, done, sub = b.step(np.zeros((1, 6), np.float32))
    assert done[0] == 1 and rew[0] == 0 and sub[0] == 776 and b.get_state()["step_count"][0] == 801 and b.counters()[3] == 1


def test_swing_fast_forward_racket_force_sequence():
    """first inner substep: no force at all (gravity uncompensated); afterwards the restoring
    force -50dx, -2dy, -2(dz-4) without gravity compensation (Appendix A.1 4a/4c, D.4)"""
    p = default_params(lin_damp=0.0, ang_damp=0.0)
    b = OracleBatch(p, ENV_SWING, 1, precision="f64")
    # ball lands on the 3rd inner substep? simpler: park it on the ground two substeps away: use timeout-free setup
    w, d = make_words(ENV_SWING, 1, racket_pos=(8.25, 0.5, 1.1), ball_pos=(2.0, 0, 0.005 + 0.0335 + 0.00067 + 0.0003),
                      goal=(-6, 0), spawn_pos=(8, 0, 0.6), init_dist=10.0, step_count=25)
    b.set_state_words(w, d)
    obs, rew, done, sub = b.step(np.zeros((1, 6), np.float32))
    assert sub[0] == 3  # outer substep + 2 inner ones
    s = b.get_state()
    g, dt = 9.81, DT
    # substep 1 (outer): hover. substep 2 (inner #1): free fall. substep 3: restoring force from the pose after #2
    vz1 = -g * dt
    z1 = float(np.float32(1.1)) + dt * vz1
    x0, y0, sz = 8.25, 0.5, float(np.float32(0.6))
    F = np.array((-50 * (x0 - 8.0), -2 * (y0 - 0.0), -2 * (z1 - sz - 4)))
    v = np.array((0, 0, vz1)) + dt * (F / 4.0 + (0, 0, -g))
    assert np.allclose(s["racket_vel"][0], v, rtol=1e-6)
    assert s["done"][0] == 1
    # stepping a finished env without reset: the pending force acts once, done stays True, reward 0
    pos = s["racket_pos"][0]
    obs, rew, done, sub = b.step(np.zeros((1, 6), np.float32))
    s2 = b.get_state()
    Fp = np.array((-50 * (pos[0] - 8.0), -2 * (pos[1] - 0.0), -2 * (pos[2] - sz - 4)))
    assert np.allclose(s2["racket_vel"][0], v + dt * (Fp / 4.0), rtol=1e-6)  # 39.24 hover + pending force
    assert done[0] == 1 and rew[0] == 0 and sub[0] == 1 and s2["done"][0] == 2
    v2 = s2["racket_vel"][0].copy()
    b.step(np.zeros((1, 6), np.float32))
    assert np.allclose(b.get_state()["racket_vel"][0], v2, rtol=1e-9)  # nothing pending any more


def test_tennis_rewards_and_termination():
    p = default_params()
    # tiers (tennisbot_env.py:90-102) through the pass-racket branch (:182-194)
    for dy, tier in ((0.3, 20), (0.7, 15), (1.5, 10), (2.5, 5), (3.5, 1), (4.5, 0)):
        b = OracleBatch(p, ENV_TENNIS, 1, precision="f64")
        w, d = make_words(ENV_TENNIS, 1, racket_pos=(10, 0, 0.7), ball_pos=(10.6, dy, 0.7), ball_vel=(0, 0, 0), step_count=100)
        b.set_state_words(w, d)
        obs, rew, done, sub = b.step(np.zeros((1, 2), np.float32))
        assert done[0] == 1 and rew[0] == tier and b.counters()[4] == 1
        # done is sticky and the reward keeps being paid while stepping a finished env (:193-194)
        obs, rew, done, sub = b.step(np.zeros((1, 2), np.float32))
        assert done[0] == 1 and rew[0] == tier and b.counters()[4] == 1
    # first 4 steps return (ob, 0, False) whatever happens (:138-139)
    b = OracleBatch(p, ENV_TENNIS, 1, precision="f64")
    w, d = make_words(ENV_TENNIS, 1, racket_pos=(10, 0, 0.7), ball_pos=(10.6, 0, 0.7), step_count=0)
    b.set_state_words(w, d)
    for t in range(1, 7):
        obs, rew, done, sub = b.step(np.zeros((1, 2), np.float32))
        assert (done[0], rew[0]) == ((0, 0.0) if t < 5 else (1, 20.0))
    # timeout (:201-203)
    b = OracleBatch(p, ENV_TENNIS, 1, precision="f64")
    w, d = make_words(ENV_TENNIS, 1, racket_pos=(10, 0, 0.7), ball_pos=(-5, 0, 0.0385), step_count=1000)
    b.set_state_words(w, d)
    obs, rew, done, sub = b.step(np.zeros((1, 2), np.float32))
    assert done[0] == 1 and rew[0] == 0 and b.counters()[3] == 1
    # racket force: 10 a at the COM, hover in z (:112-115)
    b = OracleBatch(default_params(lin_damp=0.0), ENV_TENNIS, 1, precision="f64")
    w, d = make_words(ENV_TENNIS, 1, racket_pos=(10, 0, 0.7), ball_pos=FAR, step_count=10)
    b.set_state_words(w, d)
    obs, rew, done, sub = b.step(np.array([[1.0, -0.5]], np.float32))
    assert np.allclose(b.get_state()["racket_vel"][0], (10 / 4 * DT, -5 / 4 * DT, 0), rtol=1e-12)
    assert np.allclose(obs[0, 3:6], (10 / 4 / 240, -5 / 4 / 240, 0), rtol=1e-6)


def test_auto_reset_semantics():
    n = 16
    b = batch(ENV_SWING, n=n, prec="f32", flags=F_DEFAULT | F_AUTO_RESET)
    first = b.reset()
    z = np.zeros((n, 6), np.float32)
    for _ in range(25):
        b.step(z)
    obs, rew, done, sub, term = b.step(z, want_terminal=True)
    s = b.get_state()
    assert done.all() and np.all(s["done"] == 0) and np.all(s["episode"] == 1) and np.all(s["step_count"] == 0)
    assert not np.isnan(term).any() and not np.allclose(term, obs)
    assert np.allclose(obs[:, 4:6], s["goal"], atol=1e-6)  # obs is the NEW episode's first observation
    assert np.allclose(term[:, 4:6], first[:, 4:6])        # terminal obs belongs to the old one
    assert b.counters()[5] == n
    # next episode runs again for exactly 26 steps
    for t in range(26):
        obs, rew, done, sub = b.step(z)
    assert done.all()


def test_contact_off_bench_mode_keeps_racket_dynamics():
    """BASELINE.json configs[1] 'racket-only dynamics, no ball contact': the racket<->ball pair
    is skipped, everything else is unchanged"""
    from tennisbot_rl_amd.params import F_NET
    n = 32
    rng = np.random.default_rng(5)
    acts = rng.uniform(-1, 1, (26, n, 6)).astype(np.float32)
    a = batch(ENV_SWING, n=n, prec="f32")
    c = batch(ENV_SWING, n=n, prec="f32", flags=F_NET)
    a.reset(), c.reset()
    for t in range(25):
        a.step(acts[t]), c.step(acts[t])
    sa, sc = a.get_state(), c.get_state()
    assert c.counters()[0] == 0
    untouched = a.counters()[0] == 0
    if untouched:
        assert np.array_equal(sa["racket_pos"], sc["racket_pos"])


# ---------------------------------------------------------------- racket <-> court (row f3, opt-in)
RG_FLAGS = F_DEFAULT | F_RACKET_GROUND


def test_racket_ground_support_query():
    """one narrowphase query = ONE point, the deepest hull vertex (Bullet's convex-convex pair finds one point per frame and
    lets the persistent manifold collect them): where it is, how far, and when there is none"""
    p = default_params(flags=RG_FLAGS)
    assert not default_params().flags & F_RACKET_GROUND
    hx, m, top = p.racket_half_thick, 0.001, 0.005
    # upright racket (handle down), 2 mm above the court: a corner of the handle end
    z = 0.5 + top + m + 0.002                       # COM 0.5 above the handle end
    pts = query_racket_ground(p, (8, 0, z), (0, 0, 0, 1))
    assert len(pts) == 1
    d, rr = pts[0]
    assert d == pytest.approx(0.002, abs=1e-6) and rr[2] == pytest.approx(-0.5 - m, abs=1e-6)
    assert abs(rr[0]) == pytest.approx(hx, abs=1e-7) and abs(rr[1]) == pytest.approx(0.019010, abs=1e-5)
    # tilted by 0.3 rad about x: the lower of the two handle corners, found by walking the outline downhill from vertex 0
    q = (math.sin(0.15), 0, 0, math.cos(0.15))
    (d, rr), = query_racket_ground(p, (8, 0, 0.49), q)
    verts = p.hull_vertices()
    heights = [math.cos(0.3) * vz + math.sin(0.3) * vy for vy, vz in verts]
    vy, vz = verts[int(np.argmin(heights))]
    assert rr[1] == pytest.approx(math.cos(0.3) * vy - math.sin(0.3) * vz, abs=1e-6) and d == pytest.approx(0.49 + min(heights) - m - top, abs=1e-6)
    # too high: culled
    assert query_racket_ground(p, (8, 0, 0.5 + top + m + 0.02), (0, 0, 0, 1)) == []
    # lying flat on its -x face (rotate +90 deg about y: local -x -> world -z), 1 mm into the ground: a vertex of that face
    q = (0, math.sin(math.pi / 4), 0, math.cos(math.pi / 4))
    (d, rr), = query_racket_ground(p, (8, 0, hx + top + m - 0.001), q)
    assert d == pytest.approx(-0.001, abs=2e-6) and rr[2] == pytest.approx(-hx - m, abs=1e-6)
    # deep in the ground (a tumbling racket's tip moves 3 cm per substep, the ground box is 1 cm thick): still a contact ...
    (d, rr), = query_racket_ground(p, (8, 0, 0.5 + top + m - 0.04), (0, 0, 0, 1))
    assert d == pytest.approx(-0.04, abs=2e-6)
    # ... unless the racket's COM is below the court, or it is beyond the court's edge
    assert query_racket_ground(p, (8, 0, -0.2), q) == []
    assert query_racket_ground(p, (14.5, 0, z), (0, 0, 0, 1)) == []
    # scaled racket: the handle end is 2.3 x 0.5 below the COM
    p3 = default_params(racket_scale=2.3, flags=RG_FLAGS)
    (d, rr), = query_racket_ground(p3, (8, 0, 2.3 * 0.5 + top + m + 0.001), (0, 0, 0, 1))
    assert d == pytest.approx(0.001, abs=2e-6)


@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_racket_ground_contact_dynamics(prec):
    """dropped upright racket: bounces (restitution .9*.9) on the first impact, tumbles, never tunnels -- its tip moves
    faster than the ground box is thick -- and comes to rest lying on a face, on a persistent manifold of four points that
    spans the face, whose warm-started impulses carry exactly its weight; with the flag off it falls through the court"""
    FARB = (0.0, 3.0, 50.0)
    for flag, lands in ((RG_FLAGS, True), (F_DEFAULT, False)):
        p = default_params(flags=flag, lin_damp=0.0, ang_damp=0.0)
        b = OracleBatch(p, ENV_SWING, 1, precision=prec)
        w, d = make_words(ENV_SWING, 1, racket_pos=(8, 0, 0.9), ball_pos=FARB, goal=(-6, 0), spawn_pos=(8, 0, 0.4), init_dist=10.0, step_count=30, done=2)
        b.set_state_words(w, d)
        zmin, vz_prev, bounce, most = 1e9, 0.0, None, 0
        a = np.array([[0, 0, -0.0981, 0, 0, 0]], np.float32)   # cancels the hover force: free fall
        for t in range(900 if lands else 300):
            b.step(a)
            s = b.get_state()
            vz = s["racket_vel"][0, 2]
            if bounce is None and vz > 0 and vz_prev < 0:
                bounce = (vz_prev, vz)
            vz_prev = vz
            zmin = min(zmin, s["racket_pos"][0, 2])
            most = max(most, len(b.manifold()[0]))
        if lands:
            # rebound = e |v| minus the speculative allowance d/dt of a contact caught up to 1 cm early
            assert bounce is not None and 0.0 < bounce[1] <= -0.81 * bounce[0] + 1e-9
            assert bounce[1] >= -0.81 * bounce[0] - 0.0101 * 240
            # at no time does the COM sink below what a racket lying flat allows (half thickness + ground top): nothing tunnels
            assert zmin > 0.0145 + 0.005 - 0.01
            assert np.isfinite(s["racket_quat"]).all() and abs(np.linalg.norm(s["racket_quat"][0]) - 1) < 1e-6
            # at rest, flat: COM at half thickness + margin + ground top, on four vertices of the lower face
            assert s["racket_pos"][0, 2] == pytest.approx(0.0145 + 0.001 + 0.005, abs=2e-4) and abs(s["racket_vel"][0, 2]) < 1e-3
            ids, imp = b.manifold()
            assert len(ids) == 4 and len(set(ids % 2)) == 1 and most == 4
            assert imp[:, 0].sum() == pytest.approx(4.0 * 9.81 * DT, rel=2e-3) and np.all(imp[:, 0] >= 0)
            verts = p.hull_vertices()[ids // 2]
            assert np.ptp(verts[:, 1]) > 0.4 and np.ptp(verts[:, 0]) > 0.15   # spans handle .. head and the head's width
            assert np.all(np.abs(imp[:, 1:]) <= 0.04 * imp[:, :1] + 1e-12)     # friction boxed by mu * normal impulse
        else:
            assert zmin < -1.0


def test_racket_ground_cache_is_warm_and_not_part_of_the_state_words():
    """the contact cache makes a resting racket cheap (the solve starts from the last impulses: the counter of solver sweeps
    is not exposed, so the check is on what warm starting guarantees -- impulses that do not change from substep to substep)
    and it is deliberately NOT in the state words: a restored state starts with an empty cache and rebuilds it"""
    FARB = (0.0, 3.0, 50.0)
    p = default_params(flags=RG_FLAGS, lin_damp=0.0, ang_damp=0.0)
    b = OracleBatch(p, ENV_SWING, 1, precision="f32")
    q = (0, math.sin(math.pi / 4), 0, math.cos(math.pi / 4))
    w, d = make_words(ENV_SWING, 1, racket_pos=(8, 0, 0.0145 + 0.006 + 0.01), racket_quat=q, ball_pos=FARB, goal=(-6, 0), spawn_pos=(8, 0, 0.4), init_dist=10.0, step_count=30, done=2)
    b.set_state_words(w, d)
    a = np.array([[0, 0, -0.0981, 0, 0, 0]], np.float32)
    for t in range(400):
        b.step(a)
    ids0, imp0 = b.manifold()
    b.step(a)
    ids1, imp1 = b.manifold()
    # (a redundant four-point manifold keeps shifting a little load between its points; the total is the racket's weight)
    assert len(ids0) >= 3 and np.array_equal(ids0, ids1) and np.allclose(imp0, imp1, rtol=5e-3, atol=2e-4)
    assert imp1[:, 0].sum() == pytest.approx(4.0 * 9.81 * DT, rel=1e-3)
    w2, d2 = b.get_state_words()
    b2 = OracleBatch(p, ENV_SWING, 1, precision="f32")
    b2.set_state_words(w2, d2)
    assert len(b2.manifold()[0]) == 0
    for t in range(60):
        b.step(a); b2.step(a)
    s, s2 = b.get_state(), b2.get_state()
    assert len(b2.manifold()[0]) >= 3
    assert np.allclose(s["racket_pos"], s2["racket_pos"], atol=2e-3) and abs(s2["racket_vel"][0, 2]) < 5e-3


@pytest.mark.parametrize("scale", [1.0, 2.0, 3.0])
def test_scaled_racket_has_s2_inertia_in_contact_response(scale):
    """globalScaling s (curriculum, tennisbot_env.py:234): the shape-derived inertia grows with s^2,
    the mass does not. An off-centre hit: the racket's angular momentum about its COM, (s^2 I) dw,
    equals the arm crossed with the impulse it received, and its linear momentum the impulse itself."""
    p = default_params(racket_scale=scale, lin_damp=0.0, ang_damp=0.0, gravity=0.0)
    b = OracleBatch(p, ENV_TENNIS, 1, precision="f64")
    r, ht = 0.0335, float(p.racket_half_thick) * scale
    rpos = np.array((10.0, 0.0, 3.0))
    ball = rpos + np.array((-(ht + 0.001 + r) - 0.0002, 0.04 * scale, 0.08 * scale))  # in front of the face, off the COM
    v0 = np.array((6.0, 0.5, -0.3))
    w, d = make_words(ENV_TENNIS, 1, racket_pos=rpos, ball_pos=ball, ball_vel=v0, step_count=50, racket_scale=scale)
    b.set_state_words(w, d)
    s0 = b.get_state()
    hit, dist, n, rr = query_racket(p, s0["racket_pos"][0], s0["racket_quat"][0], s0["ball_pos"][0])
    assert hit and n[0] == pytest.approx(-1.0)
    b.step(np.zeros((1, 2), np.float32))  # Tennisbot's own force on the racket: 4 * 9.81 upward, gravity is off here
    s1 = b.get_state()
    J = 0.05 * (s1["ball_vel"][0] - np.float32(v0).astype(np.float64))       # impulse on the ball
    dv = s1["racket_vel"][0] - np.array((0.0, 0.0, 4 * 9.81 / 4.0 * DT))
    assert np.allclose(4.0 * dv, -J, rtol=1e-9, atol=1e-12)                    # mass is NOT scaled
    inertia = np.array([float(np.format_float_positional(np.float32(x), unique=True)) for x in p.racket_inertia]) * scale * scale
    assert np.allclose(inertia * s1["racket_angvel"][0], np.cross(rr, -J), rtol=1e-6, atol=1e-12)
    assert np.abs(s1["racket_angvel"][0]).max() > 1e-3                          # the hit really was off-centre


def test_oblique_impacts_on_a_tumbling_racket_exchange_equal_and_opposite_impulses():
    """256 random hits -- racket at a random attitude with random velocity and spin, ball with spin arriving anywhere on the face at
    a random angle, friction and restitution acting -- checked against Newton's third law instead of against the solver's own
    formulas: what the ball gains in linear momentum the racket loses; the racket's angular momentum about its COM (world inertia
    R I R^T at the pre-step attitude) changes by arm x impulse with the arm the narrowphase reported; the ball's spin changes by
    (-r n) x impulse. The racket's own free motion (hover force, gyroscopic term) is taken from a twin batch whose balls are far away."""
    n = 256
    rng = np.random.default_rng(2024)
    p = default_params(lin_damp=0.0, ang_damp=0.0, gravity=0.0)
    r, mb, mr = float(p.ball_radius), 1.0 / float(p.ball_inv_mass), 1.0 / float(p.racket_inv_mass)
    ib = 1.0 / float(p.ball_inv_inertia)
    inertia = np.array([float(np.format_float_positional(np.float32(x), unique=True)) for x in p.racket_inertia])  # as the f64 build reads them
    q = rng.normal(size=(n, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)

    def rot(q, v):  # (x, y, z, w)
        u, w = q[:, :3], q[:, 3:4]
        t = 2 * np.cross(u, v)
        return v + w * t + np.cross(u, t)
    rp = np.tile(np.array((10.0, 0.0, 3.0)), (n, 1))
    side = np.where(rng.random(n) < 0.5, -1.0, 1.0)
    loc = np.stack([side * (float(p.racket_half_thick) + float(p.hull_margin) + r + 0.0002), rng.uniform(-0.08, 0.08, n), rng.uniform(-0.2, 0.2, n)], 1)
    bp = rp + rot(q, loc)
    vin = np.stack([-side * rng.uniform(3, 12, n), rng.uniform(-4, 4, n), rng.uniform(-4, 4, n)], 1)  # racket frame: toward the face, obliquely
    rv, rw = rng.uniform(-2, 2, (n, 3)), rng.uniform(-6, 6, (n, 3))
    fields = dict(racket_pos=rp, racket_quat=q, racket_vel=rv, racket_angvel=rw, ball_vel=rv + rot(q, vin), ball_angvel=rng.uniform(-30, 30, (n, 3)), step_count=5)
    hit, free = OracleBatch(p, ENV_SWING, n, precision="f64"), OracleBatch(p, ENV_SWING, n, precision="f64")
    w, d = make_words(ENV_SWING, n, ball_pos=bp, **fields); hit.set_state_words(w, d)
    w, d = make_words(ENV_SWING, n, ball_pos=bp + np.array((0.0, 0.0, 40.0)), **fields); free.set_state_words(w, d)
    s0 = hit.get_state()
    arms, normals, on = np.zeros((n, 3)), np.zeros((n, 3)), np.zeros(n, bool)
    for i in range(n):
        h, dist, nrm, rr = query_racket(p, s0["racket_pos"][i], s0["racket_quat"][i], s0["ball_pos"][i])
        on[i] = h  # (points beside the handle miss the outline: those envs are the control group below)
        if h:
            arms[i], normals[i] = rr, nrm
    assert on.sum() > n // 2
    a = np.zeros((n, 6), np.float32)
    hit.step(a); free.step(a)
    s1, f1 = hit.get_state(), free.get_state()
    J = mb * (s1["ball_vel"] - s0["ball_vel"])                                   # impulse the ball received (its free motion: none)
    assert np.abs(J[~on]).max(initial=0.0) == 0.0                                 # no contact, no impulse
    assert np.abs(J[on]).min(axis=0).max() >= 0.0 and (np.einsum("ij,ij->i", J, normals)[on] > 0).all()  # pushed away from the racket, every one
    assert np.allclose(mr * (s1["racket_vel"] - f1["racket_vel"]), -J, rtol=1e-9, atol=1e-11)
    R = np.stack([rot(s0["racket_quat"], np.tile(e, (n, 1))) for e in np.eye(3)], 2)  # columns = body axes in the world
    dw = s1["racket_angvel"] - f1["racket_angvel"]
    dL = np.einsum("nij,nj->ni", R, inertia * np.einsum("nji,nj->ni", R, dw))   # R I R^T dw
    assert np.allclose(dL, np.cross(arms, -J), rtol=1e-5, atol=3e-7)  # (typical size 0.04; the attitude is stored as float32: 2e-6 relative)
    assert np.allclose(ib * (s1["ball_angvel"] - s0["ball_angvel"]), np.cross(-r * normals, J), rtol=1e-5, atol=3e-7)
    tang = J - np.einsum("ij,ij->i", J, normals)[:, None] * normals
    assert (np.linalg.norm(tang, axis=1) > 1e-4).sum() > on.sum() // 2            # friction took part


def test_elastic_frictionless_impacts_conserve_kinetic_energy():
    """An invariant that knows nothing of the solver's formulas: with restitution 1 and no friction, a ball striking the face of a free
    racket anywhere, at any angle, leaves the total kinetic energy -- ball translation + ball spin + racket translation + racket
    rotation (world inertia R I R^T) -- where it was. 256 random hits on a racket at a random attitude, moving but not yet spinning
    (a tumbling racket's own semi-implicit Euler step does not conserve its rotational energy exactly); a twin batch whose balls are
    far away takes the same velocity update without the impulse: the two energies after the substep must agree. The ball starts
    1e-7 m off the surface, so neither the speculative margin nor the ERP push contributes."""
    n = 256
    rng = np.random.default_rng(77)
    p = default_params(lin_damp=0.0, ang_damp=0.0, rest_racket=1.0, fric_racket=0.0)
    r, mb, mr = float(p.ball_radius), 1.0 / float(p.ball_inv_mass), 1.0 / float(p.racket_inv_mass)
    ib = 1.0 / float(p.ball_inv_inertia)
    inertia = np.array([float(np.format_float_positional(np.float32(x), unique=True)) for x in p.racket_inertia])
    q = rng.normal(size=(n, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)

    def rot(q, v):
        u, w = q[:, :3], q[:, 3:4]
        t = 2 * np.cross(u, v)
        return v + w * t + np.cross(u, t)
    rp = np.tile(np.array((10.0, 0.0, 3.0)), (n, 1))
    side = np.where(rng.random(n) < 0.5, -1.0, 1.0)
    loc = np.stack([side * (float(p.racket_half_thick) + float(p.hull_margin) + r + 1e-7), rng.uniform(-0.08, 0.08, n), rng.uniform(-0.2, 0.2, n)], 1)
    vin = np.stack([-side * rng.uniform(3, 12, n), rng.uniform(-4, 4, n), rng.uniform(-4, 4, n)], 1)
    rv = rng.uniform(-2, 2, (n, 3))
    fields = dict(racket_pos=rp, racket_quat=q, racket_vel=rv, racket_angvel=np.zeros((n, 3)), ball_vel=rv + rot(q, vin),
                  ball_angvel=rng.uniform(-30, 30, (n, 3)), step_count=5)
    hit, free = OracleBatch(p, ENV_SWING, n, precision="f64"), OracleBatch(p, ENV_SWING, n, precision="f64")
    w, d = make_words(ENV_SWING, n, ball_pos=rp + rot(q, loc), **fields); hit.set_state_words(w, d)
    w, d = make_words(ENV_SWING, n, ball_pos=rp + rot(q, loc) + np.array((0.0, 0.0, 40.0)), **fields); free.set_state_words(w, d)
    q0 = hit.get_state()["racket_quat"]
    a = np.zeros((n, 6), np.float32)
    hit.step(a); free.step(a)

    def energy(s):
        wb = rot(q0 * np.array((-1.0, -1.0, -1.0, 1.0)), s["racket_angvel"])  # spin in the body frame of the pre-step attitude
        return (0.5 * mb * (s["ball_vel"] ** 2).sum(1) + 0.5 * ib * (s["ball_angvel"] ** 2).sum(1)
                + 0.5 * mr * (s["racket_vel"] ** 2).sum(1) + 0.5 * (inertia * wb ** 2).sum(1))
    s1, f1 = hit.get_state(), free.get_state()
    struck = np.abs(s1["ball_vel"] - f1["ball_vel"]).max(1) > 1e-3
    assert struck.sum() > n // 2                                           # (points beside the handle miss the outline)
    e1, e0 = energy(s1), energy(f1)
    # (1e-4 of the energy: what is left of the speculative d/dt term at hits near an edge, where the surface distance is not the 1e-7 m
    #  of a face hit, and of the solver's 4e-6 exit tolerance; typical 1e-6. A 1 % error in an effective mass would show as 1e-2.)
    assert np.abs(e1 - e0)[struck].max() < 1e-4 * e0[struck].max(), np.abs(e1 - e0)[struck].max()
    assert np.median(np.abs(e1 - e0)[struck] / e0[struck]) < 2e-6
    assert (np.abs(s1["racket_angvel"]).max(1)[struck] > 1e-3).mean() > 0.9  # ... and the racket did take up spin from the off-centre hits
    # the same hits at restitution 0.81: energy is lost in every one of them, never gained
    p2 = default_params(lin_damp=0.0, ang_damp=0.0, fric_racket=0.0)
    hit2 = OracleBatch(p2, ENV_SWING, n, precision="f64")
    w, d = make_words(ENV_SWING, n, ball_pos=rp + rot(q, loc), **fields); hit2.set_state_words(w, d)
    hit2.step(a)
    assert (energy(hit2.get_state()) < e0)[struck].all()

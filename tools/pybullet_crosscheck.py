#!/usr/bin/env python3
"""Cross-check / calibration against real PyBullet -- runs ONLY where `pybullet` is importable
and the reference's asset files are available (neither is true in the build image or on the
GPU box). Nothing from the reference is imported or executed: this is a
build-owned harness that issues the same PyBullet call sequence as SURVEY.md Appendix A
(swingracket_env.py:75-186, tennisbot_env.py:104-261, racket.py:35-45,92-143, objects.py:22-104)
for ONE world, injects the same initial state and action sequence into the CPU oracle
(float64), and reports per-step deviations. That is the measurement that would turn
"parity unpinned" into a pinned tolerance, and the place to calibrate the [3P-recalled]
TbParams (damping, ERP, contact threshold, margin, inertia source, friction model).

  python tools/pybullet_crosscheck.py --reference-root /path/to/tennisbot-rl --env swing --episodes 20
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference-root", required=True)
    ap.add_argument("--env", choices=["swing", "tennis"], default="swing")
    ap.add_argument("--episodes", type=int, default=10)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    try:
        import pybullet as p
    except ImportError:
        print("pybullet is not importable on this host: nothing to cross-check (parity stays unpinned)")
        return 2
    res = os.path.join(args.reference_root, "tennisbot", "resources")
    if not os.path.exists(os.path.join(res, "racket.urdf")):
        print("reference assets not found under", res)
        return 2
    from helpers import make_words
    from oracle import OracleBatch
    from tennisbot_rl_amd.params import ENV_SWING, ENV_TENNIS, default_params

    kind = ENV_SWING if args.env == "swing" else ENV_TENNIS
    rng = np.random.default_rng(args.seed)
    client = p.connect(p.DIRECT)
    worst = {}

    def load(name, pos, orn=(0, 0, 0, 1), **kw):
        return p.loadURDF(os.path.join(res, name), basePosition=pos, baseOrientation=orn, physicsClientId=client, **kw)

    def dyn(body):  # racket.py:43-45, objects.py:29-31,48-50
        p.changeDynamics(body, -1, restitution=0.9); p.changeDynamics(body, -1, lateralFriction=0.2); p.changeDynamics(body, -1, rollingFriction=0.001)

    for ep in range(args.episodes):
        p.resetSimulation(client); p.setGravity(0, 0, -9.81)
        court = load("court.urdf", (0, 0, 0)); dyn(court)
        if kind == ENV_SWING:  # swingracket_env.py:161-179
            x, y, z = rng.uniform(5.5, 11), rng.uniform(-4, 4), 0.6
            racket = load("racket.urdf", (x, y, z), p.getQuaternionFromEuler((0, 0.5, 0))); dyn(racket)
            ball = load("ball.urdf", (x - 0.1, y, z + 0.8)); dyn(ball)
            goal = (-3 - 9 * rng.random(), rng.uniform(-5, 5))
            goal_obj = load("simplegoal.urdf", (goal[0], goal[1], 0))
            steps, A = 26, 6
        else:  # tennisbot_env.py:227-246
            x, y, z = rng.uniform(7.5, 12.5), rng.uniform(-5, 5), rng.uniform(0.2, 0.21)
            racket = load("racket.urdf", (x, y, z)); dyn(racket)
            shoot = (rng.uniform(25, 37.5), rng.uniform(-10, 10), 20.0)
            ball = load("ball.urdf", (rng.uniform(-12, -6), rng.uniform(-1, 1), rng.uniform(1, 1.5))); dyn(ball)
            steps, A = 600, 2
        rp, rq = p.getBasePositionAndOrientation(racket); bp, _ = p.getBasePositionAndOrientation(ball)
        fields = dict(racket_pos=rp, racket_quat=rq, ball_pos=bp)
        if kind == ENV_SWING:
            d0 = float(np.linalg.norm(np.array(bp[:2]) - np.array(goal)))
            fields.update(goal=goal, spawn_pos=(x, y, z), init_dist=d0)
        else:
            fields.update(shoot_force=shoot)
        orc = OracleBatch(default_params(), kind, 1, precision="f64")
        w, d = make_words(kind, 1, **fields)
        orc.set_state_words(w, d)
        step_count, done = 0, False
        for t in range(steps):
            a = rng.uniform(-1, 1, A).astype(np.float32)
            pos = p.getBasePositionAndOrientation(racket)[0]
            if kind == ENV_SWING:  # swingracket_env.py:76-83
                p.applyExternalForce(racket, -1, [a[0] * 400, a[1] * 400, a[2] * 400 + 4 * 9.81], pos, p.WORLD_FRAME)
                p.applyExternalTorque(racket, -1, [a[3] * 5, a[4] * 5, a[5] * 5], p.WORLD_FRAME)
            else:  # tennisbot_env.py:112-121
                p.applyExternalForce(racket, -1, [a[0] * 10, a[1] * 10, 4 * 9.81], pos, p.WORLD_FRAME)
                if step_count < 5:
                    p.applyExternalForce(ball, -1, shoot, p.getBasePositionAndOrientation(ball)[0], p.WORLD_FRAME)
            p.stepSimulation(); step_count += 1
            if kind == ENV_SWING and step_count > 25:  # swingracket_env.py:105-141
                while not done:
                    p.stepSimulation(); step_count += 1
                    if len(p.getContactPoints(court, ball)) > 0 or len(p.getContactPoints(goal_obj, ball)) > 0 or step_count > 800:
                        done = True
                    c = p.getBasePositionAndOrientation(racket)[0]
                    p.applyExternalForce(racket, -1, [-50 * (c[0] - x), -2 * (c[1] - y), -2 * (c[2] - z - 4)], c, p.WORLD_FRAME)
            orc.step(a[None])
            st = orc.get_state()
            rp, rq = p.getBasePositionAndOrientation(racket); bp, _ = p.getBasePositionAndOrientation(ball)
            rv, rw = p.getBaseVelocity(racket); bv, bw = p.getBaseVelocity(ball)
            for name, ours, theirs in (("racket_pos", st["racket_pos"][0], rp), ("racket_quat", st["racket_quat"][0], rq), ("racket_vel", st["racket_vel"][0], rv),
                                       ("racket_angvel", st["racket_angvel"][0], rw), ("ball_pos", st["ball_pos"][0], bp), ("ball_vel", st["ball_vel"][0], bv)):
                err = float(np.abs(np.asarray(ours) - np.asarray(theirs)).max())
                worst[name] = max(worst.get(name, 0.0), err)
            if int(st["step_count"][0]) != step_count:
                worst["step_count_mismatch"] = worst.get("step_count_mismatch", 0) + 1
    print("max |oracle - pybullet| over %d episodes:" % args.episodes)
    for k, v in worst.items():
        print("  %-22s %.3e" % (k, v))
    return 0


if __name__ == "__main__":
    sys.exit(main())

"""Scene / engine parameters: the ctypes mirror of `TbParams` (include/tb_stepper.h).

The reference spreads these numbers over URDF files and `changeDynamics` calls that are
re-executed on every reset (racket.py:35-45, objects.py:16-18,29-31,48-50,102-104);
here they are parsed once (tools/extract_assets.py -> assets/scene.json) and handed to
the kernels as one POD block. Derived values are computed in float64 and rounded once.
"""
import ctypes
import json
import math
import os

import numpy as np

TB_MAX_HULL = 64
TB_HULL_REC = 8

ENV_SWING = 0   # SwingRacket-v0, tennisbot/__init__.py:8-11
ENV_TENNIS = 1  # Tennisbot-v0,   tennisbot/__init__.py:3-6

F_AUTO_RESET = 0x1
F_NET = 0x2
F_RACKET_BALL = 0x4
F_RACKET_GROUND = 0x8  # racket<->court contact (row f3; court.urdf:19-24 collides with everything): opt-in, DESIGN.md section 3
F_DEFAULT = F_NET | F_RACKET_BALL

DONE_NO, DONE_PENDING_FORCE, DONE_YES = 0, 1, 2

N_COUNTERS = 9
COUNTER_NAMES = ("racket_ball_contact_substeps", "ball_court_terminations", "goal_hits", "timeouts",
                 "pass_racket_terminations", "episodes_finished", "substeps", "nonfinite_states", "lockstep_violations")

OBS_DIM = {ENV_SWING: 6, ENV_TENNIS: 12}
ACT_DIM = {ENV_SWING: 6, ENV_TENNIS: 2}
STATE_WORDS = {ENV_SWING: 30, ENV_TENNIS: 28}

# SoA row names (include/tb_stepper.h TB_W_*); the last two rows are integers
_COMMON_ROWS = (["racket_pos"] * 3 + ["racket_quat"] * 4 + ["racket_vel"] * 3 + ["racket_angvel"] * 3
                + ["ball_pos"] * 3 + ["ball_vel"] * 3 + ["ball_angvel"] * 3)
STATE_ROWS = {
    ENV_SWING: _COMMON_ROWS + ["goal"] * 2 + ["spawn_pos"] * 3 + ["init_dist", "step_count", "episode"],
    ENV_TENNIS: _COMMON_ROWS + ["shoot_force"] * 3 + ["racket_scale", "step_count", "episode"],
}

_ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets", "scene.json")


class TbParams(ctypes.Structure):
    _fields_ = [
        ("dt", ctypes.c_float), ("inv_dt", ctypes.c_float), ("gravity", ctypes.c_float),
        ("lin_damp", ctypes.c_float), ("ang_damp", ctypes.c_float), ("lin_damp_quad", ctypes.c_float), ("ang_damp_quad", ctypes.c_float),
        ("max_ang_step", ctypes.c_float),
        ("rest_vel_threshold", ctypes.c_float), ("erp", ctypes.c_float), ("contact_threshold", ctypes.c_float),
        ("solver_iters", ctypes.c_int32), ("solver_tol", ctypes.c_float), ("flags", ctypes.c_uint32),
        ("racket_mass", ctypes.c_float), ("racket_inv_mass", ctypes.c_float),
        ("racket_inertia", ctypes.c_float * 3), ("racket_inv_inertia", ctypes.c_float * 3),
        ("racket_com", ctypes.c_float * 3), ("racket_half_thick", ctypes.c_float),
        ("hull_margin", ctypes.c_float), ("hull_bound_radius", ctypes.c_float), ("racket_scale", ctypes.c_float),
        ("ball_mass", ctypes.c_float), ("ball_inv_mass", ctypes.c_float), ("ball_inv_inertia", ctypes.c_float),
        ("ball_radius", ctypes.c_float), ("magnus_k", ctypes.c_float), ("ball_spin_max", ctypes.c_float),
        ("rest_racket", ctypes.c_float), ("rest_court", ctypes.c_float), ("rest_goal", ctypes.c_float),
        ("fric_racket", ctypes.c_float), ("fric_court", ctypes.c_float), ("fric_goal", ctypes.c_float),
        ("roll_racket", ctypes.c_float), ("roll_court", ctypes.c_float), ("roll_goal", ctypes.c_float),
        ("rest_racket_court", ctypes.c_float), ("fric_racket_court", ctypes.c_float), ("racket_ground_threshold", ctypes.c_float),
        ("ground_half", ctypes.c_float * 3), ("net_half", ctypes.c_float * 3),
        ("goal_radius", ctypes.c_float), ("goal_half_len", ctypes.c_float),
        ("n_hull", ctypes.c_int32), ("hull_edges", (ctypes.c_float * TB_HULL_REC) * TB_MAX_HULL),
    ]

    def copy(self):
        out = TbParams()
        ctypes.memmove(ctypes.byref(out), ctypes.byref(self), ctypes.sizeof(TbParams))
        return out

    def hull_vertices(self):
        """CCW (y, z) hull vertices in the COM frame at scale 1, as the kernels see them."""
        e = np.ctypeslib.as_array(self.hull_edges)[: self.n_hull]
        return e[:, :2].astype(np.float64)


class TbOptions(ctypes.Structure):
    """kernel-selection options of tb_create (include/tb_stepper.h, ABI v4): 0 = the library chooses.
    They never change a result, only which bit-identical instantiation runs."""
    _fields_ = [("struct_size", ctypes.c_uint32), ("block", ctypes.c_int32), ("tennis_reg_rows", ctypes.c_int32),
                ("swing_reg_rows", ctypes.c_int32), ("ff_lanes_per_wave", ctypes.c_int32), ("ff_sort", ctypes.c_int32),
                ("ff_phases", ctypes.c_int32), ("ff_defer", ctypes.c_int32), ("ff_defer_margin", ctypes.c_int32),
                ("policy_slices", ctypes.c_int32), ("ff_seal", ctypes.c_int32)]


def make_options(block=0, tennis_reg_rows=None, swing_reg_rows=None, ff_lanes_per_wave=0, ff_sort=None, ff_phases=0, policy_slices=0, ff_defer=None, ff_defer_margin=0,
                 ff_seal=None):
    """None = auto; True / False force a variant on / off"""
    def tri(x):
        return 0 if x is None else (1 if x else -1)
    o = TbOptions()
    o.struct_size = ctypes.sizeof(TbOptions)
    o.block, o.tennis_reg_rows, o.swing_reg_rows = int(block), tri(tennis_reg_rows), tri(swing_reg_rows)
    o.ff_lanes_per_wave, o.ff_sort, o.ff_phases = int(ff_lanes_per_wave), tri(ff_sort), int(ff_phases)
    o.policy_slices, o.ff_defer_margin = int(policy_slices), int(ff_defer_margin)
    o.ff_seal = tri(ff_seal)
    o.ff_defer = 2 if ff_defer == "all" else tri(ff_defer)  # "all": every parked env straight into the pool
    return o


def load_scene(path=_ASSETS):
    with open(path) as f:
        return json.load(f)


def hull_edge_table(hull_yz_link, com_yz, scale):
    """Edge records {a.y, a.z, e.y, e.z, 1/|e|^2, 1/|e|, 0, 0} for the CCW polygon, moved to
    the COM frame (hull vertices in the COM frame = STL vertices - inertial
    origin) and scaled by globalScaling (tennisbot_env.py:234)."""
    v = (np.asarray(hull_yz_link, dtype=np.float64) - np.asarray(com_yz, dtype=np.float64)) * float(scale)
    v = v.astype(np.float32).astype(np.float64)  # the vertices the kernels will see
    e = np.roll(v, -1, axis=0) - v
    l2 = (e ** 2).sum(1)
    rec = np.zeros((len(v), TB_HULL_REC), dtype=np.float64)
    rec[:, 0:2] = v
    rec[:, 2:4] = e
    rec[:, 4] = 1.0 / l2
    rec[:, 5] = 1.0 / np.sqrt(l2)
    return rec.astype(np.float32)


def bullet_shape_inertia(scene=None, hull_margin=0.001):
    """Inertia as Bullet derives it from the collision shapes -- what loadURDF uses unless
    URDF_USE_INERTIA_FROM_FILE is passed, and the reference passes no flags (racket.py:35-36,
    objects.py:25-27): a sphere gets 0.4 m r^2; the racket's hull sits behind a non-identity inertial
    frame, i.e. in a btCompoundShape, whose calculateLocalInertia is the box formula
    m/12 (ly^2+lz^2, lx^2+lz^2, lx^2+ly^2) over its AABB (the hull's, margin included).
    [3P-recalled] like every engine constant -- but this one is PINNED by the reference's own record:
    with these values the shipped ppo_swing policy reaches the goal in 23 % of the episodes against
    27 +- 4.4 % in the 100 PyBullet episodes stored in backup_models/ppo_swing.zip; with the URDF's
    diag(.04, .08, .12) / 1.0 it is 12 % (tools/compare_reference_policy.py)."""
    sc = scene or load_scene()
    rk, bl = sc["racket"], sc["ball"]
    ext = [(hi - lo) + 2.0 * hull_margin for lo, hi in zip(rk["bbox_min"], rk["bbox_max"])]
    m = rk["mass"] / 12.0
    return dict(racket_inertia=(m * (ext[1] ** 2 + ext[2] ** 2), m * (ext[0] ** 2 + ext[2] ** 2), m * (ext[0] ** 2 + ext[1] ** 2)),
                ball_inertia=0.4 * bl["mass"] * bl["radius"] ** 2)


def urdf_file_inertia(scene=None):
    """the <inertia> values of racket.urdf / ball.urdf (what URDF_USE_INERTIA_FROM_FILE would select)"""
    sc = scene or load_scene()
    return dict(racket_inertia=tuple(sc["racket"]["inertia_diag"]), ball_inertia=sc["ball"]["inertia_diag"][0])


def reference_rolling_friction():
    """The pairs' combined rolling-friction coefficients of the reference scene: rollingFriction .001 and
    lateralFriction .2 on racket, ball and court (racket.py:43-45, objects.py:29-31,48-50), Bullet's
    defaults 0 / 0.5 on the goal; [3P-recalled] pair rule rolling_a * friction_b + rolling_b * friction_a.
    `default_params(**reference_rolling_friction())` turns the rolling rows on (TbParams.roll_*)."""
    return dict(roll_racket=0.001 * 0.2 + 0.001 * 0.2, roll_court=0.001 * 0.2 + 0.001 * 0.2, roll_goal=0.001 * 0.5 + 0.0 * 0.2)


def default_params(racket_scale=1.0, flags=F_DEFAULT, scene=None, **overrides):
    """Reference scene + the Bullet defaults of SURVEY.md Appendix B.2.

    Any TbParams field can be overridden by keyword (calibration on a host with pybullet).
    Derived fields (inverses, hull table, bound radius) are recomputed from the primary ones.
    Inertia defaults to bullet_shape_inertia(); `**urdf_file_inertia()` selects the URDF values.
    """
    sc = scene or load_scene()
    rk, bl = sc["racket"], sc["ball"]
    s = float(racket_scale)
    shape_inertia = bullet_shape_inertia(sc)
    prim = dict(
        dt=1.0 / 240.0,                 # racket.py:24; PyBullet default fixed time step
        gravity=9.81,                   # swingracket_env.py:154
        lin_damp=0.04, ang_damp=0.04,   # [3P-recalled] PyBullet default damping: force -m v (k1 + k2 |v|), k1 = k2 = this value
        lin_damp_quad=None, ang_damp_quad=None,  # k2 on its own (None = the same as k1, as in Bullet)
        max_ang_step=0.25 * math.pi,
        rest_vel_threshold=0.2,
        # contact ERP: Bullet's library default is 0.2, but PyBullet's server creates its world with
        # solverInfo.m_erp2 = 0.08 ([3P-recalled] PhysicsServerCommandProcessor::createEmptyDynamicsWorld,
        # next to numIterations 50 and a least-squares residual exit). The reference's own PyBullet
        # record supports the smaller push: 5 of its 100 episodes keep the racket contact for a second
        # agent step after a good strike; with 0.2 that never happens here (restitution .81 + .2 > 1:
        # the ball always clears within the substep), with 0.08 in ~3 %
        erp=0.08,
        contact_threshold=0.02 * bl["radius"],
        solver_iters=50, solver_tol=4e-6,
        racket_mass=rk["mass"], racket_inertia=shape_inertia["racket_inertia"],
        hull_margin=0.001,
        ball_mass=bl["mass"], ball_inertia=shape_inertia["ball_inertia"], ball_radius=bl["radius"],
        magnus_k=0.0, ball_spin_max=0.0,
        # restitution .9 / lateralFriction .2 on racket, ball, court (racket.py:43-45,
        # objects.py:29-31,48-50); the goal keeps Bullet's defaults (0 / 0.5); pair = product
        rest_racket=0.9 * 0.9, rest_court=0.9 * 0.9, rest_goal=0.9 * 0.0,
        fric_racket=0.2 * 0.2, fric_court=0.2 * 0.2, fric_goal=0.2 * 0.5,
        roll_racket=0.0, roll_court=0.0, roll_goal=0.0,  # rolling-friction rows: opt-in, reference_rolling_friction()
        rest_racket_court=0.9 * 0.9, fric_racket_court=0.2 * 0.2,
        ground_half=tuple(0.5 * x for x in sc["court"]["ground_box_size"]),
        net_half=tuple(0.5 * x for x in sc["court"]["net_box_size"]),
        goal_radius=sc["goal"]["radius"], goal_half_len=0.5 * sc["goal"]["length"],
    )
    unknown = set(overrides) - set(prim)
    if unknown:
        raise TypeError("unknown parameter(s): %s" % sorted(unknown))
    prim.update(overrides)
    for k in ("lin_damp", "ang_damp"):
        if prim[k + "_quad"] is None:
            prim[k + "_quad"] = prim[k]

    p = TbParams()
    for k in ("dt", "gravity", "lin_damp", "ang_damp", "lin_damp_quad", "ang_damp_quad", "max_ang_step", "rest_vel_threshold", "erp",
              "contact_threshold", "solver_tol", "racket_mass", "hull_margin", "ball_mass", "ball_radius", "magnus_k",
              "ball_spin_max", "rest_racket", "rest_court", "rest_goal", "fric_racket", "fric_court",
              "fric_goal", "roll_racket", "roll_court", "roll_goal", "rest_racket_court", "fric_racket_court", "goal_radius",
              "goal_half_len"):
        setattr(p, k, float(prim[k]))
    p.inv_dt = 1.0 / float(prim["dt"])
    p.solver_iters = int(prim["solver_iters"])
    p.flags = int(flags)
    p.racket_inv_mass = 1.0 / float(prim["racket_mass"])
    p.ball_inv_mass = 1.0 / float(prim["ball_mass"])
    p.ball_inv_inertia = 1.0 / float(prim["ball_inertia"])
    for i in range(3):
        p.racket_inertia[i] = float(prim["racket_inertia"][i])
        p.racket_inv_inertia[i] = 1.0 / float(prim["racket_inertia"][i])
        p.racket_com[i] = float(rk["inertial_origin"][i])
        p.ground_half[i] = float(prim["ground_half"][i])
        p.net_half[i] = float(prim["net_half"][i])
    # geometry is stored at scale 1; `racket_scale` is what the NEXT reset of an env builds its
    # racket with (tennisbot_env.py:213-215,230-234) and the kernels scale per env on the fly
    p.racket_half_thick = float(rk["half_thickness"])
    p.racket_scale = s
    com = rk["inertial_origin"]
    rec = hull_edge_table(rk["hull_yz_ccw"], (com[1], com[2]), 1.0)
    if len(rec) > TB_MAX_HULL:
        raise ValueError("hull has %d vertices, the kernels take at most %d" % (len(rec), TB_MAX_HULL))
    p.n_hull = len(rec)
    np.ctypeslib.as_array(p.hull_edges)[: len(rec)] = rec
    # bound radius about the COM at scale 1, x extent included (slightly rounded up)
    vmax = float(np.sqrt((rec[:, :2].astype(np.float64) ** 2).sum(1).max() + float(p.racket_half_thick) ** 2))
    p.hull_bound_radius = vmax * 1.0001
    p.racket_ground_threshold = 0.02 * vmax
    return p

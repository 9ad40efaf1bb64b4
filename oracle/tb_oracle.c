/*
 * tb_oracle.c -- CPU restatement of the SwingRacket-v0 / Tennisbot-v0 hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE. Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; the shipped path (tennisbot_rl_amd) never
 * links, imports or calls anything in oracle/.
 *
 * PARITY STATUS: "parity unpinned" at the PyBullet boundary. The reference ships no
 * tests, golden vectors or fixtures, and its arithmetic lives in
 * the third-party `pybullet` wheel (Bullet3; un-pinned in tennisbot/setup.py:5, most
 * likely 3.2.5), which is absent here and cannot be fetched. What this file restates:
 *   (1) the env logic, line by line, from the reference's own Python sources (cited at
 *       each function) -- this part is pinned by those sources;
 *   (2) Bullet's published single-step pipeline for free-floating multibodies
 *       (semi-implicit Euler, k1+k2|v| damping, sequential-impulse
 *       contacts with product-rule restitution/friction, Baumgarte ERP, exponential-map
 *       orientation update) with every recalled constant a TbParams field -- this part
 *       is pinned by closed-form known-answer tests (tests/test_oracle_kat.py) and,
 *       statistically, by the one PyBullet record the reference holds: the rewards of the
 *       last 100 PyBullet training episodes stored inside backup_models/ppo_swing.zip
 *       (tests/golden/ppo_swing_reference_episodes.json). No
 *       trajectory-level PyBullet vector exists, hence "unpinned" stays in this header.
 *   (3) Since round 4 a third implementation that shares no code and no hand-derived formula with this file or with the HIP
 *       kernels (tests/test_oracle_independent.py: 12 x 12 inverse mass matrix, Jacobian rows, plain projected Gauss-Seidel;
 *       tests/test_independent_episode.py: numpy narrowphase, Python Philox, the env logic re-read from the reference) agrees
 *       with the float64 build through whole episodes of both envs. That pins the derivations, not Bullet's constants.
 *
 * One source, two builds:  -DTBO_F64 -> libtb_oracle_f64.so  (the numerical "truth")
 *                          (default) -> libtb_oracle_f32.so  (same operation order as the
 *                          HIP kernels, -ffp-contract=off + explicit fma, so integer
 *                          outputs and -- in practice -- float outputs match bit for bit)
 *
 * Arithmetic contract shared with the HIP kernels (DESIGN.md "Arithmetic contract"):
 * no implicit contraction; dot3 = fma(az,bz, fma(ay,by, ax*bx)); cross component
 * = fma(a1,b2, -(a2*b1)); IEEE sqrt and divide; sinc/cos of the half rotation angle by
 * the fixed polynomials below (f32 build) or libm (f64 build).
 */
#include "tb_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef TBO_F64
typedef double real;
#define FMA(a, b, c) fma((a), (b), (c))
#define SQRT(a) sqrt(a)
#define FABS(a) fabs(a)
#else
typedef float real;
#define FMA(a, b, c) fmaf((a), (b), (c))
#define SQRT(a) sqrtf(a)
#define FABS(a) fabsf(a)
#endif
#define R(x) ((real)(x))

typedef struct { real x, y, z; } v3;
typedef struct { real x, y, z, w; } q4;

/* ------------------------------------------------------------------ vector algebra */
static inline v3 V3(real x, real y, real z) { v3 r = {x, y, z}; return r; }
static inline v3 add3(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub3(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mul3(real s, v3 a) { return V3(s * a.x, s * a.y, s * a.z); }
static inline v3 neg3(v3 a) { return V3(-a.x, -a.y, -a.z); }
/* s*x + y */
static inline v3 axpy3(real s, v3 x, v3 y) { return V3(FMA(s, x.x, y.x), FMA(s, x.y, y.y), FMA(s, x.z, y.z)); }
static inline real dot3(v3 a, v3 b) { return FMA(a.z, b.z, FMA(a.y, b.y, a.x * b.x)); }
static inline v3 cross3(v3 a, v3 b) {
  return V3(FMA(a.y, b.z, -(a.z * b.y)), FMA(a.z, b.x, -(a.x * b.z)), FMA(a.x, b.y, -(a.y * b.x)));
}
/* rotate v by unit quaternion q: v + w t + u x t, t = 2 (u x v) */
static inline v3 qrot(q4 q, v3 v) {
  v3 u = V3(q.x, q.y, q.z);
  v3 t = mul3(R(2), cross3(u, v));
  return add3(axpy3(q.w, t, v), cross3(u, t));
}
static inline v3 qrot_inv(q4 q, v3 v) {
  q4 c = {-q.x, -q.y, -q.z, q.w};
  return qrot(c, v);
}
/* Hamilton product a (x) b */
static inline q4 qmul(q4 a, q4 b) {
  q4 r;
  r.w = FMA(-a.z, b.z, FMA(-a.y, b.y, FMA(-a.x, b.x, a.w * b.w)));
  r.x = FMA(-a.z, b.y, FMA(a.y, b.z, FMA(a.x, b.w, a.w * b.x)));
  r.y = FMA(a.z, b.x, FMA(a.y, b.w, FMA(-a.x, b.z, a.w * b.y)));
  r.z = FMA(a.z, b.w, FMA(-a.y, b.x, FMA(a.x, b.y, a.w * b.z)));
  return r;
}

/* Orientation step in terms of z = x^2, x = half the rotation angle of the substep (clamped
 * to pi/8 by the pi/4 per-substep limit): sinc_half(z) = sin(x)/x, cos_half(z) = cos(x).
 * Writing both as functions of z removes the square root and the division of the textbook
 * form sin(x)/|w| (and with them Bullet's separate small-angle Taylor branch).
 * f32 build: fixed Taylor polynomials, truncation < 1e-11 on [0, (pi/8)^2]. f64: libm. */
static inline real sinc_half(real z) {
#ifdef TBO_F64
  real x = sqrt(z);
  return x > 1e-4 ? sin(x) / x : 1.0 - z / 6.0;
#else
  real p = FMA(z, R(1.0 / 362880.0), R(-1.0 / 5040.0));
  p = FMA(z, p, R(1.0 / 120.0));
  p = FMA(z, p, R(-1.0 / 6.0));
  return FMA(z, p, R(1.0));
#endif
}
static inline real cos_half(real z) {
#ifdef TBO_F64
  return cos(sqrt(z));
#else
  real p = FMA(z, R(-1.0 / 3628800.0), R(1.0 / 40320.0));
  p = FMA(z, p, R(-1.0 / 720.0));
  p = FMA(z, p, R(1.0 / 24.0));
  p = FMA(z, p, R(-0.5));
  return FMA(z, p, R(1.0));
#endif
}

/* ------------------------------------------------------------------ counter RNG
 * Philox4x32-10 (Salmon et al., SC'11). Replaces the reference's unseedable global
 * Python `random` / numpy streams (swingracket_env.py:161-162,173; tennisbot_env.py:
 * 227-229,238-239; objects.py:91-93): only the DISTRIBUTIONS are contract there. */
void tbo_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
/* 24 random bits -> [0, 1) exactly representable in float32 */
static inline real u01(uint32_t u) { return (real)(u >> 8) * R(5.9604644775390625e-08); }
/* random.uniform(a, b) = a + (b - a) * random()   (CPython Lib/random.py) */
static inline real uniform(real lo, real span, uint32_t u) { return lo + span * u01(u); }

/* ------------------------------------------------------------------ parameters in `real`
 * f32 build: the TbParams floats as given (exactly what the HIP kernels consume).
 * f64 build: the "intended" values -- each float is widened to the shortest decimal that
 * round-trips to it (9.81f -> 9.81, not 9.81000042), dt = 1/inv_dt, and every derived
 * value (inverses, hull edge records) is recomputed in double -- so that closed forms
 * such as 4*9.81 - m g = 0 (swingracket_env.py:77) hold exactly, as they do in the
 * reference's float64 PyBullet build. */
typedef struct {
  real dt, inv_dt, gravity, lin_damp, ang_damp, lin_damp_quad, ang_damp_quad, max_ang_step, rest_vel_threshold, erp, contact_threshold;
  int solver_iters; uint32_t flags; real solver_tol;
  real racket_inv_mass, racket_inertia[3], racket_inv_inertia[3], racket_com[3], racket_half_thick, hull_margin, hull_bound_radius;
  real ball_inv_mass, ball_inv_inertia, ball_radius, magnus_k, ball_spin_max;
  real rest_racket, rest_court, rest_goal, fric_racket, fric_court, fric_goal;
  real rest_racket_court, fric_racket_court, racket_ground_threshold;
  real roll_racket, roll_court, roll_goal;
  real ground_half[3], net_half[3], goal_radius, goal_half_len;
  real racket_scale;
  int n_hull; real hull_edges[TB_MAX_HULL][6];
} Prm;

#ifdef TBO_F64
#include <stdio.h>
static double widen(float x) {
  char buf[32];
  for (int digits = 6; digits <= 9; ++digits) {
    snprintf(buf, sizeof buf, "%.*g", digits, (double)x);
    double d = strtod(buf, NULL);
    if ((float)d == x) return d;
  }
  return (double)x;
}
#define W(x) widen(x)
#else
#define W(x) (x)
#endif

static void prm_from(Prm *Q, const TbParams *P) {
  memset(Q, 0, sizeof *Q);
  Q->inv_dt = W(P->inv_dt);
  Q->dt = W(P->dt);
  Q->gravity = W(P->gravity); Q->lin_damp = W(P->lin_damp); Q->ang_damp = W(P->ang_damp); Q->lin_damp_quad = W(P->lin_damp_quad); Q->ang_damp_quad = W(P->ang_damp_quad);
  Q->max_ang_step = W(P->max_ang_step); Q->rest_vel_threshold = W(P->rest_vel_threshold); Q->erp = W(P->erp);
  Q->contact_threshold = W(P->contact_threshold); Q->solver_iters = P->solver_iters; Q->flags = P->flags; Q->solver_tol = W(P->solver_tol);
  Q->racket_inv_mass = W(P->racket_inv_mass);
  for (int i = 0; i < 3; ++i) {
    Q->racket_inertia[i] = W(P->racket_inertia[i]); Q->racket_inv_inertia[i] = W(P->racket_inv_inertia[i]);
    Q->racket_com[i] = W(P->racket_com[i]); Q->ground_half[i] = W(P->ground_half[i]); Q->net_half[i] = W(P->net_half[i]);
  }
  Q->racket_half_thick = P->racket_half_thick; /* mesh data: exact as stored */
  Q->hull_margin = W(P->hull_margin); Q->hull_bound_radius = W(P->hull_bound_radius); Q->racket_scale = W(P->racket_scale);
  Q->ball_inv_mass = W(P->ball_inv_mass); Q->ball_inv_inertia = W(P->ball_inv_inertia); Q->ball_radius = W(P->ball_radius);
  Q->magnus_k = W(P->magnus_k); Q->ball_spin_max = W(P->ball_spin_max);
  Q->rest_racket = W(P->rest_racket); Q->rest_court = W(P->rest_court); Q->rest_goal = W(P->rest_goal);
  Q->fric_racket = W(P->fric_racket); Q->fric_court = W(P->fric_court); Q->fric_goal = W(P->fric_goal);
  Q->rest_racket_court = W(P->rest_racket_court); Q->fric_racket_court = W(P->fric_racket_court);
  Q->roll_racket = W(P->roll_racket); Q->roll_court = W(P->roll_court); Q->roll_goal = W(P->roll_goal);
  Q->racket_ground_threshold = P->racket_ground_threshold; /* derived from mesh data: exact as stored */
  Q->goal_radius = W(P->goal_radius); Q->goal_half_len = W(P->goal_half_len);
  Q->n_hull = P->n_hull;
  for (int i = 0; i < P->n_hull; ++i)
    for (int k = 0; k < 6; ++k) Q->hull_edges[i][k] = P->hull_edges[i][k];
#ifdef TBO_F64
  if ((float)(1.0 / Q->inv_dt) == P->dt) Q->dt = 1.0 / Q->inv_dt;
  Q->racket_inv_mass = 1.0 / W(P->racket_mass);
  Q->ball_inv_mass = 1.0 / W(P->ball_mass);
  for (int i = 0; i < 3; ++i) Q->racket_inv_inertia[i] = 1.0 / Q->racket_inertia[i];
  for (int i = 0; i < P->n_hull; ++i) { /* edge records from the (exact) float vertices */
    int j = (i + 1) % P->n_hull;
    double ey = (double)P->hull_edges[j][0] - (double)P->hull_edges[i][0], ez = (double)P->hull_edges[j][1] - (double)P->hull_edges[i][1];
    double l2 = ey * ey + ez * ez;
    Q->hull_edges[i][2] = ey; Q->hull_edges[i][3] = ez; Q->hull_edges[i][4] = 1.0 / l2; Q->hull_edges[i][5] = 1.0 / sqrt(l2);
  }
#endif
}

/* ------------------------------------------------------------------ bodies */
typedef struct { v3 p; q4 q; v3 v; v3 w; } Racket;
typedef struct { v3 p; v3 v; v3 w; } Ball;

/* racket <-> court contact cache (TB_F_RACKET_GROUND): up to 4 hull vertices in contact with the ground's top face,
 * each with the impulses the last solve gave it (the next solve starts from them), and the outline vertex the last
 * support query ended at (the next one starts its walk there). See racket_vs_ground. */
#define MAX_RG 4
typedef struct {
  int n;
  int id[MAX_RG];                             /* hull vertex k = 2 i + side (side 0: x = -half thickness) */
  real jn[MAX_RG], jt1[MAX_RG], jt2[MAX_RG];
  int deep;                                   /* outline vertex index of the last support point */
} Manifold;

typedef struct {
  Racket r;
  Ball b;
  real aux[6]; /* swing: goal.x goal.y spawn.x spawn.y spawn.z d0 ; tennis: shoot xyz, racket scale */
  int32_t step_count;
  uint32_t episode;
  uint8_t done;
  Manifold m;
#ifdef TBO_TRACE_STATIONARY
  struct { int timed_out, first[5], bits_at_end, n_rg_at_end, substeps, first_racket, ball_below_court, solves, sweeps, rows; } trace;
#endif
} Env;

struct TboBatch {
  TbParams P0; /* as handed in */
  Prm P;
  int kind, n, threads;
  uint64_t seed, env_id_base;
  Env *e;
  uint64_t counters[TB_N_COUNTERS];
};

/* ------------------------------------------------------------------ narrowphase
 * getContactPoints only feeds `len(...) > 0` (swingracket_env.py:99-100,111-112,
 * 119-120; tennisbot_env.py:170-171) but the same manifold drives the impulse, so each
 * query returns distance, normal (toward the ball) and the arm to the racket point. */
typedef struct {
  int hit;
  real dist; /* surface-to-surface distance, negative = penetration */
  v3 n;      /* unit, from the other body toward the ball centre */
  v3 rr;     /* racket only: contact point on the racket relative to the racket COM */
} Hit;

/* ball vs racket: the racket's dynamic mesh collides as the convex hull of racket.stl,
 * i.e. a prism over a convex polygon in the racket's (y, z) plane,
 * inflated by the URDF hull margin. racket.urdf:12-16. */
static Hit sphere_vs_racket(const Prm *P, const Racket *rk, v3 c, real s) {
  Hit h; memset(&h, 0, sizeof h);
  const real r = P->ball_radius, thr = P->contact_threshold;
  v3 d = sub3(c, rk->p);
  real reach = ((P->hull_bound_radius * s + P->hull_margin) + r) + thr;
  if (dot3(d, d) > reach * reach) return h;
  /* globalScaling s (tennisbot_env.py:234): dist(p, s*Hull) = s * dist(p/s, Hull), so the query
   * point is taken to the unscaled outline; margin and ball radius are not scaled */
  real inv_s = R(1) / s;
  v3 l = mul3(inv_s, qrot_inv(rk->q, d));
  real ax = FABS(l.x) - P->racket_half_thick;
  real sx = l.x < R(0) ? R(-1) : R(1);
  int inside = 1, deep_edge = 0;
  real best_d2 = R(3.0e38), best_ry = R(0), best_rz = R(0), max_sd = R(-3.0e38);
  for (int i = 0; i < P->n_hull; ++i) {
    const real *E = P->hull_edges[i];
    real ay = E[0], az = E[1], ey = E[2], ez = E[3], il2 = E[4], il = E[5];
    real wy = l.y - ay, wz = l.z - az;
    real cr = FMA(ey, wz, -(ez * wy));
    real sd = -(cr * il);
    if (sd > max_sd) { max_sd = sd; deep_edge = i; }
    /* the closest boundary point of a convex outline lies on an edge that faces the point
     * (cr < 0): edges seen from behind cannot hold it and are skipped */
    if (cr < R(0)) {
      inside = 0;
      real t = FMA(wy, ey, wz * ez) * il2;
      t = t < R(0) ? R(0) : (t > R(1) ? R(1) : t);
      real ry = FMA(-t, ey, wy), rz = FMA(-t, ez, wz);
      real d2 = FMA(ry, ry, rz * rz);
      if (d2 < best_d2) { best_d2 = d2; best_ry = ry; best_rz = rz; }
    }
  }
  real dist_hull; v3 nl;
  if (inside) {
    if (ax > R(0) || ax >= max_sd) { dist_hull = ax; nl = V3(sx, R(0), R(0)); }
    else {
      const real *E = P->hull_edges[deep_edge];
      dist_hull = max_sd;
      nl = V3(R(0), E[3] * E[5], -(E[2] * E[5]));
    }
  } else {
    real dx = ax > R(0) ? sx * ax : R(0);
    real dd = FMA(dx, dx, best_d2);
    dist_hull = SQRT(dd);
    real inv = R(1) / dist_hull;
    nl = V3(dx * inv, best_ry * inv, best_rz * inv);
  }
  h.dist = (dist_hull * s - P->hull_margin) - r;
  h.hit = h.dist < thr;
  h.n = qrot(rk->q, nl);
  h.rr = axpy3(-(r + h.dist), h.n, d);
  return h;
}

/* ball vs an axis-aligned static box centred at the origin (court.urdf:19-24 ground,
 * :43-47 "net"; both origins sit inside <geometry> and are ignored => centred at 0) */
static Hit sphere_vs_box(const Prm *P, const real half[3], v3 c) {
  Hit h; memset(&h, 0, sizeof h);
  const real r = P->ball_radius, thr = P->contact_threshold;
  real hx = half[0], hy = half[1], hz = half[2];
  real sx = FABS(c.x) - hx, sy = FABS(c.y) - hy, sz = FABS(c.z) - hz; /* per-axis separation */
  if (sx - r >= thr || sy - r >= thr || sz - r >= thr) return h;
  real gx = c.x < R(0) ? R(-1) : R(1), gy = c.y < R(0) ? R(-1) : R(1), gz = c.z < R(0) ? R(-1) : R(1);
  int ox = sx > R(0), oy = sy > R(0), oz = sz > R(0);
  int nout = ox + oy + oz;
  real ds;
  if (nout == 0) { /* centre inside the box: least-penetration face */
    if (sz >= sx && sz >= sy) { ds = sz; h.n = V3(R(0), R(0), gz); }
    else if (sx >= sy) { ds = sx; h.n = V3(gx, R(0), R(0)); }
    else { ds = sy; h.n = V3(R(0), gy, R(0)); }
  } else if (nout == 1) { /* face region */
    if (oz) { ds = sz; h.n = V3(R(0), R(0), gz); }
    else if (ox) { ds = sx; h.n = V3(gx, R(0), R(0)); }
    else { ds = sy; h.n = V3(R(0), gy, R(0)); }
  } else { /* edge / corner region */
    v3 dl = V3(ox ? gx * sx : R(0), oy ? gy * sy : R(0), oz ? gz * sz : R(0));
    ds = SQRT(dot3(dl, dl));
    h.n = mul3(R(1) / ds, dl);
  }
  h.dist = ds - r;
  h.hit = h.dist < thr;
  return h;
}

/* ball vs the goal: static z-axis cylinder centred at (gx, gy, 0)
 * (simplegoal.urdf:17-22, objects.py:99-104) */
static Hit sphere_vs_goal(const Prm *P, real gx, real gy, v3 c) {
  Hit h; memset(&h, 0, sizeof h);
  const real r = P->ball_radius, thr = P->contact_threshold;
  const real RG = P->goal_radius, hl = P->goal_half_len;
  real rx = c.x - gx, ry = c.y - gy, rz = c.z;
  real sz = FABS(rz) - hl;
  if (sz - r >= thr) return h;
  real rad2 = FMA(rx, rx, ry * ry);
  real reach = (RG + r) + thr;
  if (rad2 > reach * reach) return h;
  real rad = SQRT(rad2);
  real sr = rad - RG;
  real gz = rz < R(0) ? R(-1) : R(1);
  v3 radial = rad > R(0) ? V3(rx / rad, ry / rad, R(0)) : V3(R(1), R(0), R(0));
  real ds;
  if (sr <= R(0) && sz <= R(0)) {
    if (sz >= sr) { ds = sz; h.n = V3(R(0), R(0), gz); }
    else { ds = sr; h.n = radial; }
  } else if (sr <= R(0)) { ds = sz; h.n = V3(R(0), R(0), gz); }
  else if (sz <= R(0)) { ds = sr; h.n = radial; }
  else {
    ds = SQRT(FMA(sr, sr, sz * sz));
    real inv = R(1) / ds;
    h.n = V3(radial.x * (sr * inv), radial.y * (sr * inv), gz * (sz * inv));
  }
  h.dist = ds - r;
  h.hit = h.dist < thr;
  return h;
}

/* ------------------------------------------------------------------ contact solver
 * Sequential impulses as in Bullet's multibody solver:
 * normal row with restitution (product rule, velocity threshold), Baumgarte ERP on
 * penetration, speculative margin on positive distance; two friction rows along
 * btPlaneSpace1(n), each boxed by mu * normal impulse. No warm start. Bullet always runs its
 * iteration cap; here a sweep whose every update is <= solver_tol * (the largest normal impulse
 * of the solve) ends it: converged to ~1e-6 of the contact force. In float32 the updates
 * otherwise oscillate by ulps forever, and rows of a redundant manifold (a racket resting on four
 * points) keep trading tiny impulses that are large only relative to themselves. */
#define ROW_BALL_STATIC 0 /* ball pushed off a static shape */
#define ROW_BALL_RACKET 1 /* ball pushed off the racket, racket pushed back */
typedef struct {
  int kind;
  v3 n, rr, t1, t2; /* n: toward the pushed body; rr: contact point relative to the racket COM */
  real mu, target, kn, kt1, kt2, jn, jt1, jt2;
  real inv_s2;      /* 1 / scale^2: the racket's inverse inertia at this env's globalScaling (see racket_invI) */
  /* rolling friction (TbParams.roll_*; ball rows only): two angular rows along t1 / t2, boxed by roll * jn */
  real roll, kr1, kr2, jr1, jr2;
} Row;

/* I_w^-1 x for a racket built with globalScaling s (tennisbot_env.py:234). Bullet derives the inertia
 * from the collision shape (params.bullet_shape_inertia), so scaling the shape by s scales the box
 * formula's extents and the inertia by s^2 (the unscaled 1 mm margin changes that by < 0.3 %); the
 * mass is not scaled. P holds the scale-1 inverse inertia; inv_s2 = 1 / (s * s). */
static inline v3 racket_invI(const Prm *P, q4 q, v3 x, real inv_s2) {
  v3 b = qrot_inv(q, x);
  b = V3((b.x * P->racket_inv_inertia[0]) * inv_s2, (b.y * P->racket_inv_inertia[1]) * inv_s2, (b.z * P->racket_inv_inertia[2]) * inv_s2);
  return qrot(q, b);
}
static inline v3 rel_vel(const Row *c, const Racket *rk, const Ball *b, v3 rb) {
  v3 pv = add3(b->v, cross3(b->w, rb));
  if (c->kind == ROW_BALL_RACKET) pv = sub3(pv, add3(rk->v, cross3(rk->w, c->rr)));
  return pv;
}
static inline void plane_space(v3 n, v3 *p, v3 *q) {
  if (FABS(n.z) > R(0.7071067811865475244)) {
    real a = FMA(n.y, n.y, n.z * n.z);
    real k = R(1) / SQRT(a);
    *p = V3(R(0), -(n.z * k), n.y * k);
    *q = V3(a * k, -(n.x * p->z), n.x * p->y);
  } else {
    real a = FMA(n.x, n.x, n.y * n.y);
    real k = R(1) / SQRT(a);
    *p = V3(-(n.y * k), n.x * k, R(0));
    *q = V3(-(n.z * p->y), n.z * p->x, a * k);
  }
}
static inline void apply_impulse(const Prm *P, const Row *c, Racket *rk, Ball *b, v3 rb, v3 dir, real j, int angular_ball) {
  b->v = axpy3(j * P->ball_inv_mass, dir, b->v);
  if (angular_ball) b->w = axpy3(j * P->ball_inv_inertia, cross3(rb, dir), b->w);
  if (c->kind == ROW_BALL_RACKET) {
    rk->v = axpy3(-(j * P->racket_inv_mass), dir, rk->v);
    rk->w = axpy3(-j, racket_invI(P, rk->q, cross3(c->rr, dir), c->inv_s2), rk->w);
  }
}
static void setup_row(const Prm *P, Row *c, const Hit *h, int kind, real e, real mu, real roll, const Racket *rk, const Ball *b, real scale) {
  const real r = P->ball_radius;
  memset(c, 0, sizeof *c);
  c->kind = kind; c->n = h->n; c->rr = h->rr; c->mu = mu;
  c->inv_s2 = R(1) / (scale * scale);
  plane_space(c->n, &c->t1, &c->t2);
  v3 rb = mul3(-r, c->n);
  real kn, kt1, kt2;
  kn = P->ball_inv_mass; kt1 = FMA(P->ball_inv_inertia, r * r, P->ball_inv_mass); kt2 = kt1;
  if (kind == ROW_BALL_RACKET) { kn = kn + P->racket_inv_mass; kt1 = kt1 + P->racket_inv_mass; kt2 = kt2 + P->racket_inv_mass; }
  if (kind != ROW_BALL_STATIC) {
    v3 a;
    a = cross3(c->rr, c->n);  kn = kn + dot3(a, racket_invI(P, rk->q, a, c->inv_s2));
    a = cross3(c->rr, c->t1); kt1 = kt1 + dot3(a, racket_invI(P, rk->q, a, c->inv_s2));
    a = cross3(c->rr, c->t2); kt2 = kt2 + dot3(a, racket_invI(P, rk->q, a, c->inv_s2));
  }
  c->kn = R(1) / kn; c->kt1 = R(1) / kt1; c->kt2 = R(1) / kt2;
  real vn = dot3(c->n, rel_vel(c, rk, b, rb));
  real rest = FABS(vn) < P->rest_vel_threshold ? R(0) : e * (-vn);
  if (rest < R(0)) rest = R(0);
  real pos = h->dist > R(0) ? -(h->dist * P->inv_dt) : -(h->dist * P->erp) * P->inv_dt;
  c->target = rest + pos; /* the normal row drives vn toward this value */
  /* [3P-recalled] Bullet's torsional rows for rolling friction: angular-only Jacobians along the two
   * friction directions, target relative spin 0, effective mass 1 / (t.I_b^-1 t + t.I_r^-1 t) */
  c->roll = roll;
  if (c->roll > R(0)) {
    real k1 = P->ball_inv_inertia, k2 = P->ball_inv_inertia;
    if (kind == ROW_BALL_RACKET) {
      k1 = k1 + dot3(c->t1, racket_invI(P, rk->q, c->t1, c->inv_s2));
      k2 = k2 + dot3(c->t2, racket_invI(P, rk->q, c->t2, c->inv_s2));
    }
    c->kr1 = R(1) / k1; c->kr2 = R(1) / k2;
  }
}

/* racket vs the court's ground box (court.urdf:19-24) with a PERSISTENT manifold, the way Bullet's
 * convex-convex pair works [3P-recalled]: per substep the narrowphase finds ONE point -- the deepest -- and adds it to the
 * pair's cached manifold of at most 4 points; cached points are refreshed with the new pose and dropped once they are farther
 * from the ground than the manifold threshold; a fifth point replaces the cached one whose removal leaves the largest
 * contact area (the deepest is never dropped); every point keeps the impulses of the last solve, and the next solve starts
 * from them (warm start). A racket that has come to rest is therefore solved in one or two sweeps, and one that flies
 * costs a support query. Restated for a prism over a convex outline above a plane:
 *   - contact points are hull VERTICES (k = 2 i + side); the deepest vertex is found by walking the outline downhill from
 *     where the last query ended (the height of outline vertex i, zr . v_i, is unimodal around a convex polygon);
 *   - a point is in contact while its height above the ground's top face is < racket_ground_threshold * scale and it is
 *     over the court (|x|, |y| inside the box) -- however deep: a tumbling racket's tip moves 3 cm per substep, three times
 *     the thickness of the 1 cm ground box, and Bullet's penetration solver (EPA) separates a hull from a thin plate it has
 *     pierced along the plate's normal all the same. Only a racket whose COM is below the plate is under the court;
 *   - warm-start factor 1 (Bullet: 0.85): the converged impulses do not depend on it, the number of sweeps does.
 * The cache lives in the env (Env.m), across substeps AND across env.step() calls; it is not part of the state words
 * (a restored or injected state starts with an empty cache, like a fresh PyBullet world). */
static inline v3 hull_vertex(const Prm *P, int k, real s) {
  real hx = P->racket_half_thick * s;
  return V3((k & 1) ? hx : -hx, P->hull_edges[k >> 1][0] * s, P->hull_edges[k >> 1][1] * s);
}
/* height of hull vertex v above the ground's top face (margin included); zr = world z in the racket frame */
static inline real vertex_height(const Prm *P, const Racket *rk, v3 zr, v3 v) {
  real dz = FMA(zr.x, v.x, FMA(zr.z, v.z, zr.y * v.y));
  return ((rk->p.z + dz) - P->hull_margin) - P->ground_half[2];
}
/* the (y, z) part of that height for outline vertex i */
static inline real outline_height(const Prm *P, v3 zr, int i, real s) {
  return FMA(zr.z, P->hull_edges[i][1] * s, zr.y * (P->hull_edges[i][0] * s));
}
static inline int vertex_supported(const Prm *P, const Racket *rk, v3 xr, v3 yr, v3 v, real h, real thr) {
  if (!(h < thr)) return 0; /* above the manifold threshold */
  real wx = rk->p.x + FMA(xr.x, v.x, FMA(xr.z, v.z, xr.y * v.y)), wy = rk->p.y + FMA(yr.x, v.x, FMA(yr.z, v.z, yr.y * v.y));
  return !(FABS(wx) > P->ground_half[0] || FABS(wy) > P->ground_half[1]);
}
/* Bullet's calcArea4Points on racket-frame positions: the largest of the three ways to cross two diagonals */
static inline real area4(v3 p0, v3 p1, v3 p2, v3 p3) {
  v3 c0 = cross3(sub3(p0, p1), sub3(p2, p3)), c1 = cross3(sub3(p0, p2), sub3(p1, p3)), c2 = cross3(sub3(p0, p3), sub3(p1, p2));
  real a0 = dot3(c0, c0), a1 = dot3(c1, c1), a2 = dot3(c2, c2);
  real m = a0 > a1 ? a0 : a1;
  return m > a2 ? m : a2;
}
/* updates the cache M for the racket's present pose; fills out[] (one Hit per cached point, in cache order) */
static int racket_vs_ground(const Prm *P, const Racket *rk, real s, Manifold *M, Hit out[MAX_RG]) {
  const real top = P->ground_half[2], thr = P->racket_ground_threshold * s;
  if ((rk->p.z - (P->hull_bound_radius * s + P->hull_margin)) - top >= thr || !(rk->p.z > top)) { M->n = 0; return 0; }
  const int nh = P->n_hull;
  v3 zr = qrot_inv(rk->q, V3(R(0), R(0), R(1)));
  /* support query: walk the outline downhill from the last support point */
  int i = M->deep;
  real fi = outline_height(P, zr, i, s);
  for (int it = 0; it < nh; ++it) {
    int j = i + 1 == nh ? 0 : i + 1;
    real fj = outline_height(P, zr, j, s);
    if (fj < fi) { i = j; fi = fj; continue; }
    j = i == 0 ? nh - 1 : i - 1;
    fj = outline_height(P, zr, j, s);
    if (fj < fi) { i = j; fi = fj; continue; }
    break;
  }
  M->deep = i;
  const int side = zr.x > R(0) ? 0 : 1; /* the face that looks down */
  const real hx = P->racket_half_thick * s;
  const real h_deep = ((rk->p.z + FMA(zr.x, side ? hx : -hx, fi)) - P->hull_margin) - top;
  if (M->n == 0 && !(h_deep < thr)) return 0; /* nothing cached, nothing near: the common case of a racket in flight */
  v3 xr = qrot_inv(rk->q, V3(R(1), R(0), R(0))), yr = qrot_inv(rk->q, V3(R(0), R(1), R(0)));
  /* refresh the cached points, drop the ones that have left */
  real h[MAX_RG + 1];
  int n = 0;
  for (int j = 0; j < M->n; ++j) {
    v3 v = hull_vertex(P, M->id[j], s);
    real hj = vertex_height(P, rk, zr, v);
    if (!vertex_supported(P, rk, xr, yr, v, hj, thr)) continue;
    M->id[n] = M->id[j]; M->jn[n] = M->jn[j]; M->jt1[n] = M->jt1[j]; M->jt2[n] = M->jt2[j]; h[n] = hj; ++n;
  }
  M->n = n;
  /* add the support point */
  const int kd = 2 * i + side;
  int known = 0;
  for (int j = 0; j < n; ++j) known |= M->id[j] == kd;
  v3 vd = hull_vertex(P, kd, s);
  if (!known && vertex_supported(P, rk, xr, yr, vd, h_deep, thr)) {
    int slot = n;
    if (n == MAX_RG) { /* full: the new point replaces the cached one (never the deepest) whose loss keeps the largest area */
      int deepest = 0;
      for (int j = 1; j < MAX_RG; ++j) if (h[j] < h[deepest]) deepest = j;
      if (h_deep < h[deepest]) deepest = -1; /* the new point is the deepest of the five: any cached one may go */
      v3 c[MAX_RG];
      for (int j = 0; j < MAX_RG; ++j) c[j] = hull_vertex(P, M->id[j], s);
      real best = R(-1);
      slot = -1;
      for (int j = 0; j < MAX_RG; ++j) {
        if (j == deepest) continue;
        real a = area4(j == 0 ? vd : c[0], j == 1 ? vd : c[1], j == 2 ? vd : c[2], j == 3 ? vd : c[3]);
        if (a > best) { best = a; slot = j; }
      }
    } else M->n = n + 1;
    M->id[slot] = kd; M->jn[slot] = R(0); M->jt1[slot] = R(0); M->jt2[slot] = R(0); h[slot] = h_deep;
  }
  for (int j = 0; j < M->n; ++j) {
    v3 v = hull_vertex(P, M->id[j], s);
    Hit *o = &out[j];
    o->hit = 1;
    o->dist = h[j];
    o->n = V3(R(0), R(0), R(1));
    o->rr = qrot(rk->q, v);
    o->rr.z = o->rr.z - P->hull_margin; /* the point on the inflated hull */
  }
  return M->n;
}

/* World-frame inverse inertia of the racket, W = R diag(I^-1 / s^2) R^T (6 unique entries), for the racket<->court rows:
 * with up to 4 points x 3 directions per solve, one matrix per substep is cheaper than 12 rotate-scale-rotate round trips. */
typedef struct { real xx, xy, xz, yy, yz, zz; } Sym3;
static Sym3 world_inv_inertia(const Prm *P, q4 q, real inv_s2) {
  real x2 = q.x + q.x, y2 = q.y + q.y, z2 = q.z + q.z;
  real xx = q.x * x2, xy = q.x * y2, xz = q.x * z2, yy = q.y * y2, yz = q.y * z2, zz = q.z * z2, wx = q.w * x2, wy = q.w * y2, wz = q.w * z2;
  real r00 = R(1) - (yy + zz), r01 = xy - wz, r02 = xz + wy, r10 = xy + wz, r11 = R(1) - (xx + zz), r12 = yz - wx, r20 = xz - wy, r21 = yz + wx, r22 = R(1) - (xx + yy);
  real d0 = P->racket_inv_inertia[0] * inv_s2, d1 = P->racket_inv_inertia[1] * inv_s2, d2 = P->racket_inv_inertia[2] * inv_s2;
  Sym3 W;
  W.xx = FMA(r02 * d2, r02, FMA(r01 * d1, r01, (r00 * d0) * r00));
  W.xy = FMA(r02 * d2, r12, FMA(r01 * d1, r11, (r00 * d0) * r10));
  W.xz = FMA(r02 * d2, r22, FMA(r01 * d1, r21, (r00 * d0) * r20));
  W.yy = FMA(r12 * d2, r12, FMA(r11 * d1, r11, (r10 * d0) * r10));
  W.yz = FMA(r12 * d2, r22, FMA(r11 * d1, r21, (r10 * d0) * r20));
  W.zz = FMA(r22 * d2, r22, FMA(r21 * d1, r21, (r20 * d0) * r20));
  return W;
}
static inline v3 sym3_mul(const Sym3 *W, v3 a) {
  return V3(FMA(W->xz, a.z, FMA(W->xy, a.y, W->xx * a.x)), FMA(W->yz, a.z, FMA(W->yy, a.y, W->xy * a.x)), FMA(W->zz, a.z, FMA(W->yz, a.y, W->xz * a.x)));
}
/* one racket<->court row: contact normal +z, friction directions btPlaneSpace1(+z) = (0,-1,0), (1,0,0); an = W (rr x n) etc. */
typedef struct { v3 rr, an, at1, at2; real target, kn, kt1, kt2, jn, jt1, jt2; } RowG;
static void setup_ground_row(const Prm *P, RowG *c, const Hit *h, const Sym3 *W, const Racket *rk) {
  v3 rr = h->rr;
  c->rr = rr;
  v3 a = V3(rr.y, -rr.x, R(0));
  c->an = sym3_mul(W, a);  c->kn = R(1) / (P->racket_inv_mass + dot3(a, c->an));
  a = V3(rr.z, R(0), -rr.x);
  c->at1 = sym3_mul(W, a); c->kt1 = R(1) / (P->racket_inv_mass + dot3(a, c->at1));
  a = V3(R(0), rr.z, -rr.y);
  c->at2 = sym3_mul(W, a); c->kt2 = R(1) / (P->racket_inv_mass + dot3(a, c->at2));
  v3 pv = add3(rk->v, cross3(rk->w, rr));
  real vn = pv.z;
  real rest = FABS(vn) < P->rest_vel_threshold ? R(0) : P->rest_racket_court * (-vn);
  if (rest < R(0)) rest = R(0);
  real pos = h->dist > R(0) ? -(h->dist * P->inv_dt) : -(h->dist * P->erp) * P->inv_dt;
  c->target = rest + pos;
}

#ifdef TBO_TRACE_STATIONARY
/* Stationarity tracer (-DTBO_TRACE_STATIONARY builds only; tools/stationary_trace.py -> profiles/r04_stationary.md):
 * for every fast-forward that ends by the 800-substep limit, the first loop substep k from which the WHOLE simulation
 * state -- racket (13 reals), ball (9), the racket<->court cache (count, vertex ids, three impulses per point, support
 * vertex) -- repeats with period p = 1 .. 4 through to the limit (the pending restoring force is a function of the
 * racket position, so it repeats with the state). Kept per env (Env.trace), so the OpenMP env loop may run. */
#define TRACE_MAXP 4
typedef struct { Racket r; Ball b; Manifold m; } Snap;
static inline void snap_take(Snap *s, const Env *e) {
  memset(s, 0, sizeof *s); s->r = e->r; s->b = e->b; s->m.n = e->m.n; s->m.deep = e->m.deep;
  for (int j = 0; j < e->m.n; ++j) { s->m.id[j] = e->m.id[j]; s->m.jn[j] = e->m.jn[j]; s->m.jt1[j] = e->m.jt1[j]; s->m.jt2[j] = e->m.jt2[j]; }
}
#endif
/* diagnostics (-DTBO_DIAG builds only): solver sweeps and solves so far. Plain globals: single-threaded use -- they are
 * not in the default builds, whose env loop runs under `#pragma omp parallel` (a data race, and a cache line every thread's
 * every solve would fight over inside the cpu_baseline leg). */
#ifdef TBO_TRACE_STATIONARY
static __thread int t_sweeps, t_solves, t_rows;  /* of the calling thread's current env: summed into Env.trace by swing_step */
#define TBO_COUNT(x) ((void)0)
#define TBO_TRACE_SOLVE(nrg) (t_solves++, t_rows += (nrg))
#define TBO_TRACE_SWEEP() (t_sweeps++)
#elif defined(TBO_DIAG)
static uint64_t g_sweeps, g_solves;
void tbo_debug_solver(uint64_t out[2], int reset) { out[0] = g_sweeps; out[1] = g_solves; if (reset) { g_sweeps = 0; g_solves = 0; } }
#define TBO_COUNT(x) ((x)++)
#define TBO_TRACE_SOLVE(nrg) ((void)0)
#define TBO_TRACE_SWEEP() ((void)0)
#else
#define TBO_COUNT(x) ((void)0)
#define TBO_TRACE_SOLVE(nrg) ((void)0)
#define TBO_TRACE_SWEEP() ((void)0)
#endif
static void solve_contacts(const Prm *P, Row *rows, int nrows, RowG *rg, int nrg, Racket *rk, Ball *b) {
  const real r = P->ball_radius;
  TBO_COUNT(g_solves);
  TBO_TRACE_SOLVE(nrg);
  real jref = R(0); /* largest normal impulse seen in this solve: the scale updates are judged against */
  /* warm start of the racket<->court rows: the cached impulses of the last solve are applied before the first sweep */
  for (int i = 0; i < nrg; ++i) {
    RowG *c = &rg[i];
    if (c->jn > jref) jref = c->jn;
    rk->v.z = FMA(c->jn, P->racket_inv_mass, rk->v.z);  rk->w = axpy3(c->jn, c->an, rk->w);
    rk->v.y = FMA(-c->jt1, P->racket_inv_mass, rk->v.y); rk->w = axpy3(c->jt1, c->at1, rk->w);
    rk->v.x = FMA(c->jt2, P->racket_inv_mass, rk->v.x);  rk->w = axpy3(c->jt2, c->at2, rk->w);
  }
  for (int it = 0; it < P->solver_iters; ++it) {
    int moved = 0;
    TBO_COUNT(g_sweeps);
    TBO_TRACE_SWEEP();
    for (int i = 0; i < nrows; ++i) {
      Row *c = &rows[i];
      v3 rb = mul3(-r, c->n);
      real vn = dot3(c->n, rel_vel(c, rk, b, rb));
      real jn = FMA(c->target - vn, c->kn, c->jn);
      if (jn < R(0)) jn = R(0);
      real d = jn - c->jn;
      c->jn = jn;
      if (jn > jref) jref = jn;
      if (d != R(0)) { apply_impulse(P, c, rk, b, rb, c->n, d, 0); if (FABS(d) > P->solver_tol * jref) moved = 1; }
    }
    for (int i = 0; i < nrg; ++i) { /* racket<->court normals */
      RowG *c = &rg[i];
      real vn = add3(rk->v, cross3(rk->w, c->rr)).z;
      real jn = FMA(c->target - vn, c->kn, c->jn);
      if (jn < R(0)) jn = R(0);
      real d = jn - c->jn;
      c->jn = jn;
      if (jn > jref) jref = jn;
      if (d != R(0)) {
        rk->v.z = FMA(d, P->racket_inv_mass, rk->v.z); rk->w = axpy3(d, c->an, rk->w);
        if (FABS(d) > P->solver_tol * jref) moved = 1;
      }
    }
    for (int i = 0; i < nrows; ++i) { /* rolling rows: after the normals, before sliding friction */
      Row *c = &rows[i];
      real lim = c->roll * c->jn;
      if (!(lim > R(0))) continue;
      for (int k = 0; k < 2; ++k) {
        v3 t = k ? c->t2 : c->t1;
        real *acc = k ? &c->jr2 : &c->jr1;
        real kr = k ? c->kr2 : c->kr1;
        real wt = dot3(t, c->kind == ROW_BALL_RACKET ? sub3(b->w, rk->w) : b->w);
        real jr = FMA(-wt, kr, *acc);
        jr = jr < -lim ? -lim : (jr > lim ? lim : jr);
        real d = jr - *acc;
        *acc = jr;
        if (d != R(0)) {
          b->w = axpy3(d * P->ball_inv_inertia, t, b->w);
          if (c->kind == ROW_BALL_RACKET) rk->w = axpy3(-d, racket_invI(P, rk->q, t, c->inv_s2), rk->w);
          /* an angular impulse: judged against jref through the ball radius (a length) */
          if (FABS(d) > P->solver_tol * (FABS(jr) > jref * r ? FABS(jr) : jref * r)) moved = 1;
        }
      }
    }
    for (int i = 0; i < nrows; ++i) {
      Row *c = &rows[i];
      real lim = c->mu * c->jn;
      if (!(lim > R(0))) continue;
      v3 rb = mul3(-r, c->n);
      for (int k = 0; k < 2; ++k) {
        v3 t = k ? c->t2 : c->t1;
        real *acc = k ? &c->jt2 : &c->jt1;
        real kt = k ? c->kt2 : c->kt1;
        real vt = dot3(t, rel_vel(c, rk, b, rb));
        real jt = FMA(-vt, kt, *acc);
        jt = jt < -lim ? -lim : (jt > lim ? lim : jt);
        real d = jt - *acc;
        *acc = jt;
        if (d != R(0)) { apply_impulse(P, c, rk, b, rb, t, d, 1); if (FABS(d) > P->solver_tol * (FABS(jt) > jref ? FABS(jt) : jref)) moved = 1; }
      }
    }
    for (int i = 0; i < nrg; ++i) { /* racket<->court friction: t1 = (0,-1,0), t2 = (1,0,0) */
      RowG *c = &rg[i];
      real lim = P->fric_racket_court * c->jn;
      if (!(lim > R(0))) continue;
      {
        real vt = -(add3(rk->v, cross3(rk->w, c->rr)).y);
        real jt = FMA(-vt, c->kt1, c->jt1);
        jt = jt < -lim ? -lim : (jt > lim ? lim : jt);
        real d = jt - c->jt1;
        c->jt1 = jt;
        if (d != R(0)) {
          rk->v.y = FMA(-d, P->racket_inv_mass, rk->v.y); rk->w = axpy3(d, c->at1, rk->w);
          if (FABS(d) > P->solver_tol * (FABS(jt) > jref ? FABS(jt) : jref)) moved = 1;
        }
      }
      {
        real vt = add3(rk->v, cross3(rk->w, c->rr)).x;
        real jt = FMA(-vt, c->kt2, c->jt2);
        jt = jt < -lim ? -lim : (jt > lim ? lim : jt);
        real d = jt - c->jt2;
        c->jt2 = jt;
        if (d != R(0)) {
          rk->v.x = FMA(d, P->racket_inv_mass, rk->v.x); rk->w = axpy3(d, c->at2, rk->w);
          if (FABS(d) > P->solver_tol * (FABS(jt) > jref ? FABS(jt) : jref)) moved = 1;
        }
      }
    }
    if (!moved) break;
  }
}

/* ------------------------------------------------------------------ one p.stepSimulation()
 * swingracket_env.py:82,107; tennisbot_env.py:121. Appendix B.1: (1) narrowphase at the
 * pre-step poses, (2) velocity update, (3) contact solve, (4) pose update, (5) forces
 * cleared (they are arguments here, so "cleared" = not carried over). */
#define CT_RACKET 1
#define CT_GROUND 2
#define CT_NET 4
#define CT_GOAL 8
#define CT_RACKET_COURT 16

static void integrate_velocities(const Prm *P, Racket *rk, Ball *b, v3 Fr, v3 Tr, v3 Fb) {
  const real dt = P->dt, g = P->gravity;
  { /* racket, linear: v += dt (F/m + g - v (k1 + k2 |v|)) */
    real kd = FMA(P->lin_damp_quad, SQRT(dot3(rk->v, rk->v)), P->lin_damp);
    v3 a = V3(FMA(Fr.x, P->racket_inv_mass, -(rk->v.x * kd)), FMA(Fr.y, P->racket_inv_mass, -(rk->v.y * kd)),
              FMA(Fr.z, P->racket_inv_mass, -(rk->v.z * kd)) - g);
    rk->v = axpy3(dt, a, rk->v);
    /* angular, body frame: w' = I^-1 (T - w x I w - I w (k1 + k2 |w|)); a racket that neither
     * spins nor is torqued has zero angular acceleration and is left untouched */
    int active = (rk->w.x != R(0)) | (rk->w.y != R(0)) | (rk->w.z != R(0)) | (Tr.x != R(0)) | (Tr.y != R(0)) | (Tr.z != R(0));
    if (active) {
      int torqued = (Tr.x != R(0)) | (Tr.y != R(0)) | (Tr.z != R(0));
      v3 wb = qrot_inv(rk->q, rk->w), Tb = torqued ? qrot_inv(rk->q, Tr) : V3(R(0), R(0), R(0));
      v3 L = V3(P->racket_inertia[0] * wb.x, P->racket_inertia[1] * wb.y, P->racket_inertia[2] * wb.z);
      v3 gy = cross3(wb, L);
      real ka = FMA(P->ang_damp_quad, SQRT(dot3(wb, wb)), P->ang_damp);
      v3 ab = V3(P->racket_inv_inertia[0] * ((Tb.x - gy.x) - L.x * ka), P->racket_inv_inertia[1] * ((Tb.y - gy.y) - L.y * ka),
                 P->racket_inv_inertia[2] * ((Tb.z - gy.z) - L.z * ka));
      rk->w = axpy3(dt, qrot(rk->q, ab), rk->w);
    }
  }
  { /* ball: isotropic inertia => no gyroscopic term */
    if (P->magnus_k != R(0)) Fb = axpy3(P->magnus_k, cross3(b->w, b->v), Fb);
    real kd = FMA(P->lin_damp_quad, SQRT(dot3(b->v, b->v)), P->lin_damp);
    v3 a = V3(FMA(Fb.x, P->ball_inv_mass, -(b->v.x * kd)), FMA(Fb.y, P->ball_inv_mass, -(b->v.y * kd)),
              FMA(Fb.z, P->ball_inv_mass, -(b->v.z * kd)) - g);
    b->v = axpy3(dt, a, b->v);
    if ((b->w.x != R(0)) | (b->w.y != R(0)) | (b->w.z != R(0))) {
      real ka = FMA(P->ang_damp_quad, SQRT(dot3(b->w, b->w)), P->ang_damp);
      v3 aw = V3(-(b->w.x * ka), -(b->w.y * ka), -(b->w.z * ka));
      b->w = axpy3(dt, aw, b->w);
    }
  }
}

/* Appendix B.1 step 4. Bullet's exponential map: q' = normalize((w s, c) (x) q) with
 * s = sin(x)/|w|, c = cos(x), x = |w| dt / 2, and |w| dt clamped to max_ang_step WITHOUT
 * rescaling w (the quirk is kept: s = (dt/2) sinc(x) in both cases, only z = x^2 is clamped).
 * Unclamped, |q'|^2 = 1 + O(eps), so one Newton step of 1/sqrt at 1 normalises to O(eps^2). */
static void integrate_pose(const Prm *P, Racket *rk, Ball *b) {
  const real dt = P->dt;
  rk->p = axpy3(dt, rk->v, rk->p);
  b->p = axpy3(dt, b->v, b->p);
  real w2 = dot3(rk->w, rk->w);
  if (w2 > R(0)) {
    real h = R(0.5) * dt, hm = R(0.5) * P->max_ang_step;
    real z = (h * h) * w2, zc = hm * hm;
    int clamped = z > zc;
    if (clamped) z = zc;
    real s = h * sinc_half(z);
    q4 dq = {rk->w.x * s, rk->w.y * s, rk->w.z * s, cos_half(z)};
    q4 q = qmul(dq, rk->q);
    real n2 = FMA(q.w, q.w, FMA(q.z, q.z, FMA(q.y, q.y, q.x * q.x)));
    real inv = clamped ? R(1) / SQRT(n2) : FMA(R(-0.5), n2, R(1.5));
    rk->q.x = q.x * inv; rk->q.y = q.y * inv; rk->q.z = q.z * inv; rk->q.w = q.w * inv;
  }
}

static int substep(const Prm *P, int kind, Racket *rk, Ball *b, Manifold *M, v3 Fr, v3 Tr, v3 Fb, real gx, real gy, real scale) {
  Row rows[4];
  RowG rg[MAX_RG];
  int nrows = 0, bits = 0;
  Hit h;
  if (P->flags & TB_F_RACKET_BALL) {
    h = sphere_vs_racket(P, rk, b->p, scale);
    if (h.hit) { bits |= CT_RACKET; }
  } else h.hit = 0;
  Hit hg = sphere_vs_box(P, P->ground_half, b->p);
  Hit hn; hn.hit = 0;
  if (P->flags & TB_F_NET) hn = sphere_vs_box(P, P->net_half, b->p);
  Hit hc; hc.hit = 0;
  if (kind == TB_ENV_SWING) hc = sphere_vs_goal(P, gx, gy, b->p);
  if (hg.hit) bits |= CT_GROUND;
  if (hn.hit) bits |= CT_NET;
  if (hc.hit) bits |= CT_GOAL;
  Hit hrg[MAX_RG];
  int nrg = 0;
  if (P->flags & TB_F_RACKET_GROUND) nrg = racket_vs_ground(P, rk, scale, M, hrg);
  if (nrg) bits |= CT_RACKET_COURT;

  integrate_velocities(P, rk, b, Fr, Tr, Fb);

  if (bits) {
    if (bits & CT_RACKET) setup_row(P, &rows[nrows++], &h, ROW_BALL_RACKET, P->rest_racket, P->fric_racket, P->roll_racket, rk, b, scale);
    if (bits & CT_GROUND) setup_row(P, &rows[nrows++], &hg, ROW_BALL_STATIC, P->rest_court, P->fric_court, P->roll_court, rk, b, scale);
    if (bits & CT_NET) setup_row(P, &rows[nrows++], &hn, ROW_BALL_STATIC, P->rest_court, P->fric_court, P->roll_court, rk, b, scale);
    if (bits & CT_GOAL) setup_row(P, &rows[nrows++], &hc, ROW_BALL_STATIC, P->rest_goal, P->fric_goal, P->roll_goal, rk, b, scale);
    if (nrg) {
      Sym3 W = world_inv_inertia(P, rk->q, R(1) / (scale * scale));
      for (int j = 0; j < nrg; ++j) {
        setup_ground_row(P, &rg[j], &hrg[j], &W, rk);
        rg[j].jn = M->jn[j]; rg[j].jt1 = M->jt1[j]; rg[j].jt2 = M->jt2[j];
      }
    }
    solve_contacts(P, rows, nrows, rg, nrg, rk, b);
    for (int j = 0; j < nrg; ++j) { M->jn[j] = rg[j].jn; M->jt1[j] = rg[j].jt1; M->jt2[j] = rg[j].jt2; }
  }
  integrate_pose(P, rk, b);
  return bits;
}

/* ------------------------------------------------------------------ reset() */
static void fill_obs(int kind, const Env *e, float *obs) {
  if (kind == TB_ENV_SWING) { /* swingracket_env.py:143-144,184-185 */
    obs[0] = (float)e->r.p.x; obs[1] = (float)e->r.p.y; obs[2] = (float)e->b.p.x; obs[3] = (float)e->b.p.y;
    obs[4] = (float)e->aux[0]; obs[5] = (float)e->aux[1];
  } else { /* tennisbot_env.py:134-136,259-261 */
    obs[0] = (float)e->r.p.x; obs[1] = (float)e->r.p.y; obs[2] = (float)e->r.p.z;
    obs[3] = (float)e->r.v.x; obs[4] = (float)e->r.v.y; obs[5] = (float)e->r.v.z;
    obs[6] = (float)e->b.p.x; obs[7] = (float)e->b.p.y; obs[8] = (float)e->b.p.z;
    obs[9] = (float)e->b.v.x; obs[10] = (float)e->b.v.y; obs[11] = (float)e->b.v.z;
  }
}

static void reset_env(const struct TboBatch *B, Env *e, uint64_t env_id) {
  const Prm *P = &B->P;
  uint32_t key[2] = {(uint32_t)B->seed, (uint32_t)(B->seed >> 32)};
  uint32_t ctr[4] = {(uint32_t)env_id, (uint32_t)(env_id >> 32), e->episode, 0u};
  uint32_t u[4], w[4];
  tbo_philox4x32(ctr, key, u);
  v3 zero = V3(R(0), R(0), R(0));
  e->r.v = zero; e->r.w = zero; e->b.v = zero; e->b.w = zero;
  v3 com = V3(P->racket_com[0], P->racket_com[1], P->racket_com[2]);
  if (B->kind == TB_ENV_SWING) {
    /* swingracket_env.py:161-170: link pos x~U(5.5,11), y~U(-4,4), z=0.6, rpy=(0,0.5,0);
     * ball at (x-0.1, y, z+0.8); racket.py:131 reports the COM = link + R com */
    real x = uniform(R(5.5), R(5.5), u[0]), y = uniform(R(-4), R(8), u[1]), z = R(0.6);
    q4 q0 = {R(0), R(0.24740395925452292), R(0), R(0.96891242171064473)}; /* sin, cos of 0.25 */
    e->r.q = q0;
    e->r.p = add3(V3(x, y, z), qrot(q0, com));
    e->b.p = V3(x - R(0.1), y, z + R(0.8));
    /* swingracket_env.py:173: np.random.uniform(-3,-12) -> (-12,-3]; uniform(-5,5) */
    real gx = uniform(R(-3), R(-9), u[2]), gy = uniform(R(-5), R(10), u[3]);
    e->aux[0] = gx; e->aux[1] = gy; e->aux[2] = x; e->aux[3] = y; e->aux[4] = z;
    real dx = e->b.p.x - gx, dy = e->b.p.y - gy;
    e->aux[5] = SQRT(FMA(dx, dx, dy * dy)); /* swingracket_env.py:174-175 */
    ctr[3] = 1u;
  } else {
    /* tennisbot_env.py:227-246; objects.py:82-96 (ball born at (-9,0,1)) */
    ctr[3] = 1u;
    tbo_philox4x32(ctr, key, w);
    real x = uniform(R(7.5), R(5), u[0]), y = uniform(R(-5), R(10), u[1]), z = uniform(R(0.2), R(0.21) - R(0.2), u[2]);
    q4 q0 = {R(0), R(0), R(0), R(1)};
    e->r.q = q0;
    e->aux[3] = P->racket_scale; /* Racket(..., scale=self.racket_scale), tennisbot_env.py:230-234 */
    e->r.p = add3(V3(x, y, z), mul3(e->aux[3], com));
    e->aux[0] = uniform(R(25), R(12.5), u[3]);
    e->aux[1] = uniform(R(-10), R(20), w[0]);
    e->aux[2] = R(20);
    e->aux[4] = e->aux[5] = R(0);
    e->b.p = V3(uniform(R(-12), R(6), w[1]), uniform(R(-1), R(2), w[2]), uniform(R(1), R(0.5), w[3]));
    ctr[3] = 2u;
  }
  if (P->ball_spin_max != R(0)) { /* extension (BASELINE configs[4]); 0 = reference */
    tbo_philox4x32(ctr, key, w);
    real m = P->ball_spin_max;
    e->b.w = V3(uniform(-m, R(2) * m, w[0]), uniform(-m, R(2) * m, w[1]), uniform(-m, R(2) * m, w[2]));
  }
  e->step_count = 0;
  e->done = TB_DONE_NO;
  memset(&e->m, 0, sizeof e->m); /* a rebuilt world has no contacts yet */
}

/* ------------------------------------------------------------------ step() */
/* swingracket_env.py:63-73 */
static inline real moved_dist_to_goal(const Env *e) {
  real dx = e->b.p.x - e->aux[0], dy = e->b.p.y - e->aux[1];
  real d = SQRT(FMA(dx, dx, dy * dy));
  return ((e->aux[5] - d) / e->aux[5]) * R(20);
}

/* swingracket_env.py:75-145 */
static real swing_step(const Prm *P, Env *e, const float *a, int *substeps, uint64_t *cnt) {
  v3 F = V3(R(a[0]) * R(400), R(a[1]) * R(400), FMA(R(a[2]), R(400), R(4) * R(9.81))); /* :76-77 */
  v3 T = V3(R(a[3]) * R(5), R(a[4]) * R(5), R(a[5]) * R(5));                             /* :78 */
  v3 zero = V3(R(0), R(0), R(0));
  if (e->done == TB_DONE_PENDING_FORCE) { /* force issued at :135-141 is still in the accumulator */
    F = add3(F, V3(R(-50) * (e->r.p.x - e->aux[2]), R(-2) * (e->r.p.y - e->aux[3]), R(-2) * ((e->r.p.z - e->aux[4]) - R(4))));
    e->done = TB_DONE_YES;
  }
  int bits = substep(P, TB_ENV_SWING, &e->r, &e->b, &e->m, F, T, zero, e->aux[0], e->aux[1], R(1)); /* :82 */
  e->step_count += 1;                                                                   /* :83 */
  int ns = 1;
  real reward = R(0);
  if (bits & CT_RACKET) cnt[0]++;
  if (e->step_count < 25 && (bits & CT_RACKET)) reward += R(2); /* :98-101 */
  if (e->step_count > 25) {                                     /* :105 */
    v3 Fp = zero; /* forces were cleared by the substep above */
#ifdef TBO_TRACE_STATIONARY
    Snap ring[TRACE_MAXP + 1]; int first[TRACE_MAXP + 1] = {-1, -1, -1, -1, -1}, nloop = 0, first_racket = -1;
    t_sweeps = 0; t_solves = 0; t_rows = 0;
    snap_take(&ring[0], e); nloop = 1; /* ring[0]: the state the loop starts from (no force pending: not comparable, kept for indexing) */
#endif
    while (!e->done) { /* :106 */
      bits = substep(P, TB_ENV_SWING, &e->r, &e->b, &e->m, Fp, zero, zero, e->aux[0], e->aux[1], R(1)); /* :107 */
#ifdef TBO_TRACE_STATIONARY
      { /* ring[k % 5] = state after loop substep k; first[p] = the k from which state_k == state_(k-p) has held without a break */
        Snap *cur = &ring[nloop % (TRACE_MAXP + 1)];
        snap_take(cur, e);
        for (int p = 1; p <= TRACE_MAXP; ++p) {
          int same = nloop >= p && !memcmp(cur, &ring[(nloop - p) % (TRACE_MAXP + 1)], sizeof *cur);
          if (!same) first[p] = -1; else if (first[p] < 0) first[p] = nloop;
        }
        { /* the racket and its contact cache alone (the ball left out) */
          const Snap *prev = &ring[(nloop - 1) % (TRACE_MAXP + 1)];
          int same = !memcmp(&cur->r, &prev->r, sizeof cur->r) && !memcmp(&cur->m, &prev->m, sizeof cur->m);
          if (!same) first_racket = -1; else if (first_racket < 0) first_racket = nloop;
        }
        ++nloop;
      }
#endif
      e->step_count += 1; ns++;
      if (bits & CT_RACKET) cnt[0]++;
      if (bits & (CT_GROUND | CT_NET)) { e->done = TB_DONE_PENDING_FORCE; reward += moved_dist_to_goal(e); cnt[1]++; } /* :111-114 */
      if (bits & CT_GOAL) { reward += moved_dist_to_goal(e); reward += R(50); e->done = TB_DONE_PENDING_FORCE; cnt[2]++; } /* :119-123 */
#ifdef TBO_TRACE_STATIONARY
      if (e->step_count > 800) {
        e->trace.timed_out = !e->done; e->trace.bits_at_end = bits; e->trace.n_rg_at_end = e->m.n; e->trace.substeps = nloop - 1;
        for (int p = 0; p <= TRACE_MAXP; ++p) e->trace.first[p] = first[p];
        e->trace.first_racket = first_racket; e->trace.solves = t_solves; e->trace.sweeps = t_sweeps; e->trace.rows = t_rows;
        e->trace.ball_below_court = e->b.p.z < -(P->ground_half[2] + P->ball_radius);
      }
#endif
      if (e->step_count > 800) { if (!e->done) cnt[3]++; e->done = TB_DONE_PENDING_FORCE; } /* :127-128 */
      Fp = V3(R(-50) * (e->r.p.x - e->aux[2]), R(-2) * (e->r.p.y - e->aux[3]), R(-2) * ((e->r.p.z - e->aux[4]) - R(4))); /* :135-141 */
    }
  }
  *substeps = ns;
  if (e->m.n == 0) e->m.deep = 0; /* an empty cache is not kept between env.step() calls: the next support walk starts at vertex 0 */
  return reward;
}

/* tennisbot_env.py:90-102 */
static inline real dist_to_reward(real d) {
  return d < R(0.5) ? R(20) : d < R(1) ? R(15) : d < R(2) ? R(10) : d < R(3) ? R(5) : d < R(4) ? R(1) : R(0);
}

/* tennisbot_env.py:104-207 (the DELAY_MODE sleep at :124-126 is intentionally dropped) */
static real tennis_step(const Prm *P, Env *e, const float *a, float *obs, int *ret_done, uint64_t *cnt) {
  v3 zero = V3(R(0), R(0), R(0));
  v3 F = V3(R(a[0]) * R(10), R(a[1]) * R(10), R(4) * R(9.81)); /* :112-115 */
  v3 Fb = zero;
  if (e->step_count < 5) Fb = V3(e->aux[0], e->aux[1], e->aux[2]); /* :118-119 */
  int bits = substep(P, TB_ENV_TENNIS, &e->r, &e->b, &e->m, F, zero, Fb, R(0), R(0), e->aux[3]); /* :121 */
  e->step_count += 1;                                                        /* :122 */
  if (bits & CT_RACKET) cnt[0]++;
  fill_obs(TB_ENV_TENNIS, e, obs); /* :134-136 */
  real reward = R(0);
  *ret_done = 0;
  if (e->step_count < 5) { if (e->m.n == 0) e->m.deep = 0; return reward; } /* :138-139: returns the literal False, not self.done */
  real dz = e->b.p.z - e->r.p.z, dy = e->b.p.y - e->r.p.y;
  real delta = SQRT(FMA(dz, dz, dy * dy)); /* :142-143 */
  if (bits & CT_RACKET) { reward += R(25); reward += dist_to_reward(delta); } /* :170-174 */
  if (!(e->b.p.x - e->r.p.x < R(0.5))) { /* :182-194 */
    if (!e->done) cnt[4]++;
    e->done = TB_DONE_YES;
    reward += dist_to_reward(delta);
  }
  /* :197-198 `3 > x > 15` can never hold: no penalty */
  if (e->step_count > 1000) { if (!e->done) cnt[3]++; e->done = TB_DONE_YES; } /* :201-203 */
  *ret_done = e->done != TB_DONE_NO; /* :207 */
  if (e->m.n == 0) e->m.deep = 0;
  return reward;
}

/* ------------------------------------------------------------------ batch API */
TboBatch *tbo_create(const TbParams *params, int env_kind, int n_envs, uint64_t seed, uint64_t env_id_base) {
  if (!params || n_envs <= 0 || (env_kind != TB_ENV_SWING && env_kind != TB_ENV_TENNIS)) return NULL;
  if (params->n_hull < 3 || params->n_hull > TB_MAX_HULL) return NULL;
  TboBatch *B = (TboBatch *)calloc(1, sizeof *B);
  B->P0 = *params; prm_from(&B->P, params); B->kind = env_kind; B->n = n_envs; B->seed = seed; B->env_id_base = env_id_base; B->threads = 1;
  B->e = (Env *)calloc((size_t)n_envs, sizeof(Env));
  for (int i = 0; i < n_envs; ++i) { B->e[i].r.q.w = R(1); if (env_kind == TB_ENV_TENNIS) B->e[i].aux[3] = R(1); B->e[i].episode = 0xFFFFFFFFu; } /* first reset -> episode 0 */
  return B;
}
void tbo_destroy(TboBatch *B) { if (B) { free(B->e); free(B); } }
void tbo_set_params(TboBatch *B, const TbParams *p) { B->P0 = *p; prm_from(&B->P, p); }
void tbo_set_threads(TboBatch *B, int t) { B->threads = t < 1 ? 1 : t; }
int tbo_real_bytes(void) { return (int)sizeof(real); }

void tbo_reset(TboBatch *B, const uint8_t *mask, float *obs) {
  const int od = B->kind == TB_ENV_SWING ? TB_SWING_OBS_DIM : TB_TENNIS_OBS_DIM;
  for (int i = 0; i < B->n; ++i) {
    if (mask && !mask[i]) continue;
    B->e[i].episode += 1u;
    reset_env(B, &B->e[i], B->env_id_base + (uint64_t)i);
    if (obs) fill_obs(B->kind, &B->e[i], obs + (size_t)i * od);
  }
}

static int state_finite(const Env *e) {
  const real *f = (const real *)&e->r;
  for (size_t k = 0; k < sizeof(Racket) / sizeof(real); ++k) if (!isfinite(f[k])) return 0;
  f = (const real *)&e->b;
  for (size_t k = 0; k < sizeof(Ball) / sizeof(real); ++k) if (!isfinite(f[k])) return 0;
  return 1;
}

void tbo_step(TboBatch *B, const float *actions, float *obs, float *reward, uint8_t *done, float *terminal_obs, int32_t *substeps) {
  const int swing = B->kind == TB_ENV_SWING;
  const int od = swing ? TB_SWING_OBS_DIM : TB_TENNIS_OBS_DIM, ad = swing ? TB_SWING_ACT_DIM : TB_TENNIS_ACT_DIM;
  uint64_t total[TB_N_COUNTERS] = {0};
#pragma omp parallel num_threads(B->threads)
  {
    uint64_t cnt[TB_N_COUNTERS] = {0};
#pragma omp for schedule(static)
    for (int i = 0; i < B->n; ++i) {
      Env *e = &B->e[i];
      float o[TB_TENNIS_OBS_DIM];
      int ns = 1, d;
      real rew;
      if (swing) {
        rew = swing_step(&B->P, e, actions + (size_t)i * ad, &ns, cnt);
        fill_obs(TB_ENV_SWING, e, o);
        d = e->done != TB_DONE_NO; /* swingracket_env.py:145 returns self.done */
      } else rew = tennis_step(&B->P, e, actions + (size_t)i * ad, o, &d, cnt);
      cnt[6] += (uint64_t)ns;
      if (!state_finite(e)) cnt[7]++;
      if (d && (B->P.flags & TB_F_AUTO_RESET)) {
        cnt[5]++;
        if (terminal_obs) memcpy(terminal_obs + (size_t)i * od, o, sizeof(float) * od);
        e->episode += 1u;
        reset_env(B, e, B->env_id_base + (uint64_t)i);
        fill_obs(B->kind, e, o);
      }
      memcpy(obs + (size_t)i * od, o, sizeof(float) * od);
      reward[i] = (float)rew;
      done[i] = (uint8_t)d;
      if (substeps) substeps[i] = ns;
    }
#pragma omp critical
    for (int k = 0; k < TB_N_COUNTERS; ++k) total[k] += cnt[k];
  }
  for (int k = 0; k < TB_N_COUNTERS; ++k) B->counters[k] += total[k];
}

void tbo_counters(TboBatch *B, uint64_t *out) { memcpy(out, B->counters, sizeof B->counters); }
void tbo_counters_reset(TboBatch *B) { memset(B->counters, 0, sizeof B->counters); }

/* state exchange in the library's SoA word layout (include/tb_stepper.h TB_W_*) */
static int n_words(int kind) { return kind == TB_ENV_SWING ? TB_SWING_WORDS : TB_TENNIS_WORDS; }
static void env_to_vals(int kind, const Env *e, double *v) {
  const real *f = (const real *)&e->r;
  for (int k = 0; k < 13; ++k) v[k] = (double)f[k];
  f = (const real *)&e->b;
  for (int k = 0; k < 9; ++k) v[13 + k] = (double)f[k];
  int na = kind == TB_ENV_SWING ? 6 : 4;
  for (int k = 0; k < na; ++k) v[22 + k] = (double)e->aux[k];
  v[22 + na] = (double)e->step_count;
  v[23 + na] = (double)e->episode;
}
void tbo_get_state(TboBatch *B, uint32_t *words, uint8_t *done) {
  const int nw = n_words(B->kind), n = B->n;
  double v[TB_SWING_WORDS];
  for (int i = 0; i < n; ++i) {
    env_to_vals(B->kind, &B->e[i], v);
    for (int k = 0; k < nw - 2; ++k) { float f = (float)v[k]; memcpy(&words[(size_t)k * n + i], &f, 4); }
    int32_t sc = B->e[i].step_count; memcpy(&words[(size_t)(nw - 2) * n + i], &sc, 4);
    words[(size_t)(nw - 1) * n + i] = B->e[i].episode;
    if (done) done[i] = B->e[i].done;
  }
}
void tbo_get_state_f64(TboBatch *B, double *vals, uint8_t *done) {
  const int nw = n_words(B->kind), n = B->n;
  double v[TB_SWING_WORDS];
  for (int i = 0; i < n; ++i) {
    env_to_vals(B->kind, &B->e[i], v);
    for (int k = 0; k < nw; ++k) vals[(size_t)k * n + i] = v[k];
    if (done) done[i] = B->e[i].done;
  }
}
void tbo_set_state(TboBatch *B, const uint32_t *words, const uint8_t *done) {
  const int nw = n_words(B->kind), n = B->n;
  for (int i = 0; i < n; ++i) {
    Env *e = &B->e[i];
    float f[TB_SWING_WORDS];
    for (int k = 0; k < nw - 2; ++k) memcpy(&f[k], &words[(size_t)k * n + i], 4);
    real *d = (real *)&e->r;
    for (int k = 0; k < 13; ++k) d[k] = (real)f[k];
    d = (real *)&e->b;
    for (int k = 0; k < 9; ++k) d[k] = (real)f[13 + k];
    int na = B->kind == TB_ENV_SWING ? 6 : 4;
    for (int k = 0; k < na; ++k) e->aux[k] = (real)f[22 + k];
    memcpy(&e->step_count, &words[(size_t)(nw - 2) * n + i], 4);
    e->episode = words[(size_t)(nw - 1) * n + i];
    e->done = done ? done[i] : TB_DONE_NO;
    memset(&e->m, 0, sizeof e->m); /* the contact cache is not part of the state words */
  }
}

#ifdef TBO_TRACE_STATIONARY
/* out[i] = {timed_out, first[1..4], contact bits of the last substep, cached racket<->court points, loop substeps,
 * first substep of the racket-and-cache-only fixed point, ball below the court at the end, contact solves / solver sweeps /
 * racket<->court rows summed over the solves of that fast-forward}; clears the records */
void tbo_trace_stationary(TboBatch *B, int32_t *out) {
  for (int i = 0; i < B->n; ++i) {
    Env *e = &B->e[i];
    int32_t *o = out + 13 * (size_t)i;
    o[8] = e->trace.first_racket; o[9] = e->trace.ball_below_court; o[10] = e->trace.solves; o[11] = e->trace.sweeps; o[12] = e->trace.rows;
    o[0] = e->trace.timed_out; for (int p = 1; p <= TRACE_MAXP; ++p) o[p] = e->trace.first[p];
    o[5] = e->trace.bits_at_end; o[6] = e->trace.n_rg_at_end; o[7] = e->trace.substeps;
    memset(&e->trace, 0, sizeof e->trace);
  }
}
#endif

/* unit-level hooks for the known-answer tests */
int tbo_get_manifold(TboBatch *B, int env, int32_t ids[MAX_RG], double imp[3 * MAX_RG]) {
  const Manifold *M = &B->e[env].m;
  for (int j = 0; j < MAX_RG; ++j) { ids[j] = j < M->n ? M->id[j] : -1; imp[3 * j] = j < M->n ? M->jn[j] : 0; imp[3 * j + 1] = j < M->n ? M->jt1[j] : 0; imp[3 * j + 2] = j < M->n ? M->jt2[j] : 0; }
  return M->n;
}
int tbo_query_racket(const TbParams *P0, const float rp[3], const float rq[4], const float c[3], double out[8]) {
  Prm Pq, *P = &Pq; prm_from(P, P0);
  Racket rk; memset(&rk, 0, sizeof rk);
  rk.p = V3(R(rp[0]), R(rp[1]), R(rp[2]));
  rk.q.x = R(rq[0]); rk.q.y = R(rq[1]); rk.q.z = R(rq[2]); rk.q.w = R(rq[3]);
  Hit h = sphere_vs_racket(P, &rk, V3(R(c[0]), R(c[1]), R(c[2])), P->racket_scale);
  out[0] = h.dist; out[1] = h.n.x; out[2] = h.n.y; out[3] = h.n.z; out[4] = h.rr.x; out[5] = h.rr.y; out[6] = h.rr.z; out[7] = 0;
  return h.hit;
}
int tbo_query_racket_ground(const TbParams *P0, const float rp[3], const float rq[4], double out[MAX_RG * 8]) {
  Prm Pq, *P = &Pq; prm_from(P, P0);
  Racket rk; memset(&rk, 0, sizeof rk);
  rk.p = V3(R(rp[0]), R(rp[1]), R(rp[2]));
  rk.q.x = R(rq[0]); rk.q.y = R(rq[1]); rk.q.z = R(rq[2]); rk.q.w = R(rq[3]);
  Hit h[MAX_RG];
  Manifold M; memset(&M, 0, sizeof M);
  int n = 0;
  for (int pass = 0; pass < MAX_RG; ++pass) n = racket_vs_ground(P, &rk, P->racket_scale, &M, h); /* one point per query: let the cache fill */
  for (int j = 0; j < n; ++j) { out[8 * j] = h[j].dist; out[8 * j + 1] = h[j].rr.x; out[8 * j + 2] = h[j].rr.y; out[8 * j + 3] = h[j].rr.z; }
  return n;
}
int tbo_query_box(const TbParams *P0, const float half[3], const float c[3], double out[4]) {
  Prm Pq, *P = &Pq; prm_from(P, P0);
  real hh[3] = {W(half[0]), W(half[1]), W(half[2])};
  Hit h = sphere_vs_box(P, hh, V3(R(c[0]), R(c[1]), R(c[2])));
  out[0] = h.dist; out[1] = h.n.x; out[2] = h.n.y; out[3] = h.n.z;
  return h.hit;
}
int tbo_query_goal(const TbParams *P0, float gx, float gy, const float c[3], double out[4]) {
  Prm Pq, *P = &Pq; prm_from(P, P0);
  Hit h = sphere_vs_goal(P, R(gx), R(gy), V3(R(c[0]), R(c[1]), R(c[2])));
  out[0] = h.dist; out[1] = h.n.x; out[2] = h.n.y; out[3] = h.n.z;
  return h.hit;
}

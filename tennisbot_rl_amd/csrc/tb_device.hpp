// tb_device.hpp -- device-side rigid-body physics of the batched stepper (gfx950).
//
// One lane = one world: a 6-DoF floating racket, a ball and the static court / net /
// goal. Everything here runs in registers; the only memory the physics touches is the
// racket outline table, staged once per workgroup into LDS and read at a wave-uniform
// index (LDS broadcast, no bank conflicts).
//
// What this replaces (per world, per call) in the reference:
//   p.stepSimulation()          swingracket_env.py:82,107; tennisbot_env.py:121
//   p.getContactPoints(a, b)    swingracket_env.py:99,111,119; tennisbot_env.py:170
//   applyExternalForce/Torque   racket.py:97-100; objects.py:72
// The engine semantics follow Bullet's multibody pipeline as recalled in SURVEY.md
// Appendix B; each recalled constant is a TbParams field.
//
// Arithmetic contract: compiled with -ffp-contract=off; every fused
// multiply-add below is an explicit __builtin_fmaf; sqrt and divide are IEEE-rounded
// (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt); sin/cos of the half angle are
// fixed polynomials. Integer outputs (done, step counters, contact bits) therefore do not
// depend on host/device libm differences.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tb_stepper.h"

#define TB_DEV __device__ __forceinline__
#define TB_N_CULL 12
// the outline table in LDS / device memory: 2 float4 per edge, then the cull planes (TB_N_CULL x 3 floats) in 9 float4.
// The planes used to be kernel arguments: 36 SGPRs the loop-free step kernels had to spill at their very start.
#define TB_HULL_PLANES (2 * TB_MAX_HULL)
#define TB_HULL_KP (2 * TB_MAX_HULL + 9)
// ... and behind the planes a copy of the whole KParams block (TB_KP_ROWS float4) for the SF_COLD forms of substep
#define TB_KP_ROWS 18
#define TB_HULL_LDS (TB_HULL_KP + TB_KP_ROWS)
#define FMA(a, b, c) __builtin_fmaf((a), (b), (c))

namespace tb {

struct vec3 { float x, y, z; };
struct quat { float x, y, z, w; };

TB_DEV vec3 mk(float x, float y, float z) { vec3 r; r.x = x; r.y = y; r.z = z; return r; }
TB_DEV vec3 operator+(vec3 a, vec3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
TB_DEV vec3 operator-(vec3 a, vec3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
TB_DEV vec3 operator*(float s, vec3 a) { return mk(s * a.x, s * a.y, s * a.z); }
TB_DEV vec3 fma3(float s, vec3 x, vec3 y) { return mk(FMA(s, x.x, y.x), FMA(s, x.y, y.y), FMA(s, x.z, y.z)); }
TB_DEV float dot(vec3 a, vec3 b) { return FMA(a.z, b.z, FMA(a.y, b.y, a.x * b.x)); }
// any component non-zero (or NaN)? |x| + |y| + |z| is zero only if all three are (nothing cancels between magnitudes; a NaN or an
// overflow to infinity compares unequal to zero too): two additions with free |.| modifiers and ONE comparison where three
// comparisons and the scalar ors between them cost a lone wave three more issue slots
TB_DEV bool nonzero3(vec3 v) { return (fabsf(v.x) + fabsf(v.y)) + fabsf(v.z) != 0.0f; }
TB_DEV vec3 cross(vec3 a, vec3 b) {
  return mk(FMA(a.y, b.z, -(a.z * b.y)), FMA(a.z, b.x, -(a.x * b.z)), FMA(a.x, b.y, -(a.y * b.x)));
}
TB_DEV vec3 rotate(quat q, vec3 v) {
  vec3 u = mk(q.x, q.y, q.z);
  vec3 t = 2.0f * cross(u, v);
  return fma3(q.w, t, v) + cross(u, t);
}
TB_DEV vec3 rotate_inv(quat q, vec3 v) {
  quat c; c.x = -q.x; c.y = -q.y; c.z = -q.z; c.w = q.w;
  return rotate(c, v);
}
// Two vectors rotated by the same (inverse) quaternion side by side, as 2-wide packed fp32 (v_pk_mul_f32 / v_pk_fma_f32: two
// IEEE operations per instruction): component for component the very operations of rotate_inv, in the same order -- same bits.
// 21 packed instructions for what were 42; +1.1 % at 1 M envs. (The compiler packs much of the rest on its own -- 120 v_pk_* in the
// large-batch fast-forward. Hand-packing more lost: racket and ball linear updates side by side -4 %, the quaternion product as
// 8 packed instructions -20 % -- pair moves, and 40 bytes of scratch per lane that throttle the launch. With the unpacked build of
// round 3 -- no SLP vectoriser, this function the only packed code -- it is still worth +1.5 %; the linear updates side by side were
// tried once more: the same instruction count, 36-40 bytes of scratch again.)
typedef float f2 __attribute__((ext_vector_type(2)));
TB_DEV f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
TB_DEV void rotate_inv2(quat q, vec3 a, vec3 b, vec3& ra, vec3& rb) {
  const float ux = -q.x, uy = -q.y, uz = -q.z;
  const f2 vx = {a.x, b.x}, vy = {a.y, b.y}, vz = {a.z, b.z};
  const f2 UX = {ux, ux}, UY = {uy, uy}, UZ = {uz, uz}, W = {q.w, q.w}, two = {2.0f, 2.0f};
  // t = 2 cross(u, v)
  const f2 tx = two * fma2(UY, vz, -(UZ * vy)), ty = two * fma2(UZ, vx, -(UX * vz)), tz = two * fma2(UX, vy, -(UY * vx));
  // fma3(w, t, v) + cross(u, t)
  const f2 rx = fma2(W, tx, vx) + fma2(UY, tz, -(UZ * ty));
  const f2 ry = fma2(W, ty, vy) + fma2(UZ, tx, -(UX * tz));
  const f2 rz = fma2(W, tz, vz) + fma2(UX, ty, -(UY * tx));
  ra = mk(rx.x, ry.x, rz.x); rb = mk(rx.y, ry.y, rz.y);
}
TB_DEV quat qmul(quat a, quat b) {
  quat r;
  r.w = FMA(-a.z, b.z, FMA(-a.y, b.y, FMA(-a.x, b.x, a.w * b.w)));
  r.x = FMA(-a.z, b.y, FMA(a.y, b.z, FMA(a.x, b.w, a.w * b.x)));
  r.y = FMA(a.z, b.x, FMA(a.y, b.w, FMA(-a.x, b.z, a.w * b.y)));
  r.z = FMA(a.z, b.w, FMA(-a.y, b.x, FMA(a.x, b.y, a.w * b.z)));
  return r;
}
// orientation step as functions of z = x^2 (x = half the substep rotation angle, <= pi/8):
// sinc_half(z) = sin(x)/x, cos_half(z) = cos(x); no sqrt, no divide, no small-angle branch
TB_DEV float sinc_half(float z) {
  float p = FMA(z, (float)(1.0 / 362880.0), (float)(-1.0 / 5040.0));
  p = FMA(z, p, (float)(1.0 / 120.0));
  p = FMA(z, p, (float)(-1.0 / 6.0));
  return FMA(z, p, 1.0f);
}
TB_DEV float cos_half(float z) {
  float p = FMA(z, (float)(-1.0 / 3628800.0), (float)(1.0 / 40320.0));
  p = FMA(z, p, (float)(-1.0 / 720.0));
  p = FMA(z, p, (float)(1.0 / 24.0));
  p = FMA(z, p, -0.5f);
  return FMA(z, p, 1.0f);
}

// ---------------------------------------------------------------- counter RNG (Philox4x32-10)
// stands in for the reference's unseedable global `random` / numpy draws
// (swingracket_env.py:161-162,173; tennisbot_env.py:227-229,238-239; objects.py:91-93)
TB_DEV void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
    uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
    uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
    c0 = n0; c1 = l1; c2 = n2; c3 = l0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
TB_DEV float uniform(float lo, float span, uint32_t u) { return lo + span * ((float)(u >> 8) * 5.9604644775390625e-08f); }

// ---------------------------------------------------------------- kernel-argument parameter block
// the scalar part of TbParams travels in the kernarg segment (SGPRs, wave-uniform);
// the outline table goes through LDS
struct KParams {
  float dt, inv_dt, gravity, lin_damp, ang_damp, lin_damp_quad, ang_damp_quad, max_ang_step, rest_vel_threshold, erp, contact_threshold;
  int solver_iters; uint32_t flags; float solver_tol;
  float racket_inv_mass, racket_inertia[3], racket_inv_inertia[3], racket_com[3], racket_half_thick, hull_margin, hull_bound_radius, racket_scale;
  float ball_inv_mass, ball_inv_inertia, ball_radius, magnus_k, ball_spin_max;
  float rest_racket, rest_court, rest_goal, fric_racket, fric_court, fric_goal;
  float rest_racket_court, fric_racket_court, racket_ground_threshold;
  float roll_racket, roll_court, roll_goal;
  float ground_half[3], net_half[3], goal_radius, goal_half_len;
  float static_top;  // highest point of any enabled static shape (host-derived)
  // conservative convex superset of the outline (host-derived): half-planes n.p <= h in the COM
  // (y, z) frame -- 8 fixed directions + the longest edges. dist(p, outline) >= max(n.p - h).
  float ball_kn, ball_kt;  // 1/inv_mass and 1/(inv_mass + inv_inertia r^2): ball-only effective masses (host-derived)
  int n_hull;
};

struct Racket { vec3 p; quat q; vec3 v; vec3 w; };
struct Ball { vec3 p; vec3 v; vec3 w; };

struct Hit {
  bool hit;
  float dist;  // surface distance, negative = penetration
  vec3 n;      // unit, toward the ball centre
  vec3 rr;     // racket pair: contact point on the racket relative to its COM
};

constexpr int CT_RACKET = 1, CT_GROUND = 2, CT_NET = 4, CT_GOAL = 8, CT_RACKET_COURT = 16;
constexpr int CT_ESCAPE = 32;  // not a contact: substep's SF_ESC form left the env untouched, see there

// TB_STAMP / TB_LANES / TB_DIAG_*: instrumentation of the diagnostic builds only (cycle stamps, lane census, timing ablations).
// All of it lives in tb_diag.hpp; in the product build every one of these macros expands to nothing.
#include "tb_diag.hpp"

// ---------------------------------------------------------------- narrowphase
// sphere vs racket: prism over the convex (y, z) outline of racket.stl in the COM frame,
// inflated by the URDF hull margin (racket.urdf:12-16).
// `hull` points at the LDS copy of the edge records {a.y a.z e.y e.z | 1/|e|^2 1/|e| - -}.
// The bounding-sphere cull is done by the caller (substep) so that a whole wave can skip
// the sweep with one ballot; `d` = ball centre - racket COM.
// `s` is the env's racket scale (globalScaling, tennisbot_env.py:234): dist(p, s*Hull) =
// s * dist(p/s, Hull), so the query point goes to the unscaled outline; the margin and the ball
// radius are not scaled. SCALED = false (SwingRacket, s == 1) drops the multiplications by 1.
TB_DEV bool racket_in_reach(const KParams& P, vec3 d, float s) {
  float reach = ((P.hull_bound_radius * s + P.hull_margin) + P.ball_radius) + P.contact_threshold;
  return !(dot(d, d) > reach * reach);
}
// RELOAD: read the cull planes from LDS at every call instead of letting the compiler hoist the loop-invariant reads
// out of the fast-forward loop, where they occupy 36 VGPRs for the whole kernel (183 -> 166: a third wave per SIMD;
// +15 % SwingRacket at 1 M envs, same box; at 4096 envs, one wave per SIMD, the reads only lengthen the loop)
// The test is in three parts so that the expensive middle one can be shared by the wave:
//   racket_cull       per lane: local-frame culls; true = this lane's query needs the exact outline sweep
//   outline_sweep     WAVE-COOPERATIVE: must be reached by all active lanes together (the caller branches on __any)
//   racket_finish     per lane: distance, normal and arm from the sweep's result
template <bool SCALED>  // `dl` = rotate_inv(rk.q, ball - racket): substep rotates it together with the racket's spin (rotate_inv2)
TB_DEV bool racket_slab(const KParams& P, vec3 dl, float s, vec3& l, float& ax) {
  const float r = P.ball_radius, thr = P.contact_threshold;
  l = dl;
  if (SCALED) l = (1.0f / s) * l;
  ax = fabsf(l.x) - P.racket_half_thick;
  // Local-frame culls before the 38-edge sweep. In a SwingRacket episode the ball starts
  // 0.47 m in FRONT of the face, inside the bounding sphere, and falls alongside the racket:
  // without these every substep of every lane would sweep the outline.
  //  (1) slab: the distance to the prism is >= ax (its x separation), and the sweep's own
  //      result is monotone in it, so (ax - margin) - r >= thr implies "no hit" exactly;
  //  (2) a 12-plane convex superset of the outline: the 2-D distance is >= the largest plane
  //      separation; 0.1 mm of slack covers all rounding, lanes inside the slack just run the
  //      exact sweep. (A plain bounding box is too loose next to the handle, where the outline
  //      is a narrow wedge: a tumbling racket's ball spends many substeps there.)
  return !(((SCALED ? ax * s : ax) - P.hull_margin) - r >= thr);
}
// (2), each lane for itself; true = still not separated
template <bool SCALED, bool RELOAD>
TB_DEV bool racket_planes(const KParams& P, const float4* hull, vec3 l, float s) {
  const float r = P.ball_radius, thr = P.contact_threshold;
  float sep = -3.0e38f;
  int first = TB_HULL_PLANES;
#ifndef TB_HINT_RELOAD_PLANES
#define TB_HINT_RELOAD_PLANES 1  // (a scheduling hint like those listed in tb_kernels.hpp; tools/diag/r04_hint_recheck.py)
#endif
  if (RELOAD && TB_HINT_RELOAD_PLANES) asm volatile("" : "+v"(first));  // a row index the compiler cannot see through: the reads stay here (the index, not the
                                              // pointer: laundering the pointer loses its address space and the reads become flat loads)
  const float4* cpl = hull + first;
  float cp[3 * TB_N_CULL];
#pragma unroll
  for (int k = 0; k < 9; ++k) { float4 t = cpl[k]; cp[4 * k] = t.x; cp[4 * k + 1] = t.y; cp[4 * k + 2] = t.z; cp[4 * k + 3] = t.w; }
#pragma unroll
  for (int k = 0; k < TB_N_CULL; ++k) sep = fmaxf(sep, FMA(cp[3 * k + 1], l.z, cp[3 * k] * l.y) - cp[3 * k + 2]);
  return !(((SCALED ? sep * s : sep) - P.hull_margin) - r >= thr + 1.0e-4f);
}
// (Measured and dropped: testing the outline's bounding box -- four of the planes -- first and fetching the other eight only for lanes
//  it lets through: no difference at 1 M envs. And sharing (2) by the wave -- each past-the-slab lane's point broadcast, 12 lanes testing one plane each, the same
//  verdict bit for bit. 3 % of the lanes get past the slab, so four wave-substeps in five walk the 12 planes for one or two lanes;
//  still the shared form was 7 % slower at 1 M envs, same box: its per-lane trips serialise on LDS latency.)
template <bool SCALED, bool RELOAD = false>
TB_DEV bool racket_cull(const KParams& P, const float4* hull, vec3 dl, float s, vec3& l, float& ax) {
  if (!racket_slab<SCALED>(P, dl, s, l, ax)) return false;
  TB_LANES_ADD1(12);  // lanes past the slab test
  if (!racket_planes<SCALED, RELOAD>(P, hull, l, s)) return false;
  TB_DIAG_ADD_EACH(10, 1);    // lane-sweeps
  TB_DIAG_ADD_LEADER(11, 1);  // wave-sweeps
  return true;
}
// The outline sweep of one query point = n_hull dependent trips (two 16-byte LDS reads, ~25 branchy VALU each): ~12 k cycles
// by in-kernel stamps, and ONE lane in 64 doing it charges the whole wave (0.15 racket contacts per episode = 8 % of a
// wave's fast-forward substeps, a quarter of its time). So the wave does it together: for each lane that needs a sweep (one at
// a time, wave-uniform loop) the query point is broadcast, up to TB_SWEEP_HELPERS active lanes evaluate every
// TB_SWEEP_HELPERS-th edge each -- the very same arithmetic per edge -- and the asking lane combines their partial results by
// the sequential loop's own rule (smallest distance / largest signed distance, lowest edge index on ties), read from the
// helpers by cross-lane shuffles. Bit-identical to the one-lane loop; ~6 x shorter.
constexpr int TB_SWEEP_HELPERS = TB_DIAG_SWEEP_HELPERS;  // 8 (tb_diag.hpp; 4 / 16 measured: 8.81 / 8.57 G vs 8.80 G env steps/s at 1 M envs)
struct SweepOut { float best_d2, best_ry, best_rz, max_sd; int deep_edge; bool inside; };
TB_DEV SweepOut outline_sweep(const float4* hull, int n_hull, bool need, float qy, float qz) {
  const int lane = (int)(threadIdx.x & 63);
  const unsigned long long act = __ballot(1);
  const int rank = __popcll(act & ((1ull << lane) - 1ull));
  const int na = __popcll(act), G = na < TB_SWEEP_HELPERS ? na : TB_SWEEP_HELPERS;
  SweepOut mine;
  mine.best_d2 = 3.0e38f; mine.best_ry = 0.0f; mine.best_rz = 0.0f; mine.max_sd = -3.0e38f; mine.deep_edge = 0; mine.inside = true;
  for (unsigned long long todo = __ballot(need); todo; todo &= todo - 1ull) {
    const int src = __ffsll((long long)todo) - 1;
    const float py = __shfl(qy, src, 64), pz = __shfl(qz, src, 64);
    // my share of the edges, in increasing order (helpers: the first G active lanes)
    float bd2 = 3.0e38f, bry = 0.0f, brz = 0.0f, msd = -3.0e38f;
    int bi = 0x7fffffff, di = 0x7fffffff, ins = 1;
    if (rank < G) {
      for (int i = rank; i < n_hull; i += G) {
        const float4 e0 = hull[2 * i], e1 = hull[2 * i + 1];
        float wy = py - e0.x, wz = pz - e0.y;
        float cr = FMA(e0.z, wz, -(e0.w * wy));
        float sd = -(cr * e1.y);
        if (sd > msd) { msd = sd; di = i; }
        // the closest boundary point of a convex outline lies on an edge that faces the point
        // (cr < 0); edges seen from behind cannot hold it and are skipped
        if (cr < 0.0f) {
          ins = 0;
          float t = FMA(wy, e0.z, wz * e0.w) * e1.x;
          t = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);
          float ry = FMA(-t, e0.z, wy), rz = FMA(-t, e0.w, wz);
          float d2 = FMA(ry, ry, rz * rz);
          if (d2 < bd2) { bd2 = d2; bry = ry; brz = rz; bi = i; }
        }
      }
    }
    // combine the helpers' partials (every lane runs the shuffles; only lane `src` keeps the result)
    float cd2 = 3.0e38f, csd = -3.0e38f;
    int cbi = 0x7fffffff, cdi = 0x7fffffff, cins = 1, win = src;
    unsigned long long m = act;
    for (int g = 0; g < G; ++g) {
      const int pl = __ffsll((long long)m) - 1;  // physical lane of helper g (wave-uniform)
      m &= m - 1ull;
      const float od2 = __shfl(bd2, pl, 64), osd = __shfl(msd, pl, 64);
      const int obi = __shfl(bi, pl, 64), odi = __shfl(di, pl, 64), oins = __shfl(ins, pl, 64);
      if (od2 < cd2 || (od2 == cd2 && obi < cbi)) { cd2 = od2; cbi = obi; win = pl; }
      if (osd > csd || (osd == csd && odi < cdi)) { csd = osd; cdi = odi; }
      cins &= oins;
    }
    const float wry = __shfl(bry, win, 64), wrz = __shfl(brz, win, 64);
    if (lane == src) {
      mine.best_d2 = cd2; mine.best_ry = wry; mine.best_rz = wrz; mine.max_sd = csd; mine.deep_edge = cdi == 0x7fffffff ? 0 : cdi; mine.inside = cins != 0;
    }
  }
  return mine;
}

// the same sweep by one lane for itself (every instantiation but the large-batch fast-forward, see substep).
// One edge, WITHOUT branches: the closest-point arithmetic of an edge seen from behind (cr >= 0) is done and thrown away by the
// selects -- the values kept, and every operation that produced them, are those of the branching form (the oracle still has
// it): bit-identical. (The helpers' loop of the cooperative sweep above keeps its branch:
// without it the survivor kernels of a 1 M-env fast-forward issue 7 % more vector instructions and nothing gets faster.)
constexpr int TB_SWEEP_CHUNK = 4;  // edges requested together per trip (2: no gain; the branching one-edge-per-trip loop: EXPERIMENTS.md)
struct EdgeRec { float4 e0; float2 e1; };  // {a.y a.z e.y e.z}, {1/|e|^2 1/|e|}
TB_DEV EdgeRec outline_edge(const float4* hull, int i) {
  EdgeRec r;
  r.e0 = hull[2 * i];
  r.e1 = *reinterpret_cast<const float2*>(hull + 2 * i + 1);
  return r;
}
TB_DEV void sweep_edge(SweepOut& o, int i, const EdgeRec& r, float py, float pz) {
  const float4 e0 = r.e0;
  float wy = py - e0.x, wz = pz - e0.y;
  float cr = FMA(e0.z, wz, -(e0.w * wy));
  float sd = -(cr * r.e1.y);
  const bool deeper = sd > o.max_sd;
  o.max_sd = deeper ? sd : o.max_sd; o.deep_edge = deeper ? i : o.deep_edge;
  const bool faces = cr < 0.0f;
  float t = FMA(wy, e0.z, wz * e0.w) * r.e1.x;
  t = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);
  float ry = FMA(-t, e0.z, wy), rz = FMA(-t, e0.w, wz);
  float d2 = FMA(ry, ry, rz * rz);
  const bool closer = faces & (d2 < o.best_d2);
  o.inside = o.inside & !faces;
  o.best_d2 = closer ? d2 : o.best_d2; o.best_ry = closer ? ry : o.best_ry; o.best_rz = closer ? rz : o.best_rz;
}
// The branching loop was 38 DEPENDENT trips: read an edge, wait, test, branch, read the rest of it, wait (~12 k cycles for a lone
// wave by in-kernel stamps, two thirds of it waiting for the table). Here the records of four edges are requested together and
// evaluated side by side: a quarter of the waits, and four independent chains for the scheduler.
// (Also tried: the NEXT four records requested before the present four are evaluated. Written plainly the compiler folds "this
//  trip's records = the last trip's reads" back into reads at the top of the trip; pinned by a compiler barrier it measures +2 % at
//  4096 envs and -15 % at 32768, where the barrier also keeps the fast-forward loop from hoisting its cull planes.)
TB_DEV SweepOut outline_sweep_serial(const float4* hull, int n_hull, float py, float pz) {
  SweepOut o;
  o.best_d2 = 3.0e38f; o.best_ry = 0.0f; o.best_rz = 0.0f; o.max_sd = -3.0e38f; o.deep_edge = 0; o.inside = true;
  constexpr int C = TB_SWEEP_CHUNK;
  const int whole = n_hull - n_hull % C;
#pragma unroll 1
  for (int i = 0; i < whole; i += C) {
    EdgeRec r[C];
#pragma unroll
    for (int k = 0; k < C; ++k) r[k] = outline_edge(hull, i + k);
#pragma unroll
    for (int k = 0; k < C; ++k) sweep_edge(o, i + k, r[k], py, pz);
  }
#pragma unroll 1
  for (int i = whole; i < n_hull; ++i) sweep_edge(o, i, outline_edge(hull, i), py, pz);
  return o;
}

// The sweep SHARED BY THE WHOLE WAVE, FOUR QUERIES AT A TIME (the env wave of the policy rollout kernels, which brings all 64 lanes into
// substep: 16 or 48 of them hold envs, the others a far-away dummy and nothing to do). The wave's four rows of 16 lanes each take one
// asking lane's query; a lane evaluates the edges col, col + 16, col + 32, col + 48 of the outline for it -- the arithmetic of
// sweep_edge -- and the 16 partial results of a row are combined by four rotate-and-combine steps on DPP (row_ror 8, 4, 2, 1:
// register moves, not the LDS crossbar a shuffle goes through), by the sequential loop's own rule: the lexicographic minimum of
// (squared distance, edge index) over the facing edges, the lexicographic (largest signed distance, smallest edge index), the
// conjunction of "seen from inside". Both orders are total on what takes part (a NaN never replaced the running value in the loop
// and is left out here), so the grouping does not matter: bit-identical to outline_sweep_serial.
// For the lone env wave every LDS round trip is ~150 exposed cycles. A lane sweeping for itself pays ten of them; round 4's first
// shared form -- one edge per lane over all 64, one query at a time, six shuffle steps -- paid eight per asking lane (stamps, PPO
// collect under the trained policy: 3.5 lanes ask where one does, 2400 cycles each); a PASS here pays two -- the queries out, the
// results back -- and serves four lanes (racket narrowphase 2190 -> 1580 cycles per wave and step).
// MUST be reached by all 64 lanes of the wave (n_hull <= 64 = TB_MAX_HULL edges).
template <int N> TB_DEV int ror16(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x120 + N, 0xf, 0xf, false); }
template <int N> TB_DEV float ror16(float v) { return __int_as_float(ror16<N>(__float_as_int(v))); }
struct SweepPart { float d2, ry, rz, sd; int bi, di, ins; };
TB_DEV void sweep_merge(SweepPart& a, float od2, float ory, float orz, int obi, float osd, int odi, int oins) {
  const bool closer = od2 < a.d2 || (od2 == a.d2 && obi < a.bi);
  a.d2 = closer ? od2 : a.d2; a.ry = closer ? ory : a.ry; a.rz = closer ? orz : a.rz; a.bi = closer ? obi : a.bi;
  const bool deeper = osd > a.sd || (osd == a.sd && odi < a.di);
  a.sd = deeper ? osd : a.sd; a.di = deeper ? odi : a.di;
  a.ins &= oins;
}
template <int N> TB_DEV void sweep_merge_ror(SweepPart& a) {
  sweep_merge(a, ror16<N>(a.d2), ror16<N>(a.ry), ror16<N>(a.rz), ror16<N>(a.bi), ror16<N>(a.sd), ror16<N>(a.di), ror16<N>(a.ins));
}
TB_DEV SweepOut outline_sweep_rows(const float4* hull, int n_hull, bool need, float qy, float qz) {
  const int lane = (int)(threadIdx.x & 63), row = lane >> 4, col = lane & 15;
  SweepOut mine;
  mine.best_d2 = 3.0e38f; mine.best_ry = 0.0f; mine.best_rz = 0.0f; mine.max_sd = -3.0e38f; mine.deep_edge = 0; mine.inside = true;
  EdgeRec r[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    r[k].e0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f); r[k].e1 = make_float2(0.0f, 0.0f);
    if (col + 16 * k < n_hull) r[k] = outline_edge(hull, col + 16 * k);
  }
  const unsigned long long ask = __ballot(need);
  const int rank = __popcll(ask & ((1ull << lane) - 1ull));  // this lane's place among the asking lanes
  int pass = 0;
  for (unsigned long long todo = ask; todo; ++pass) {
    // the next four asking lanes, one per row (fewer left: the spare rows repeat the first and nobody reads them)
    const int s0 = __ffsll((long long)todo) - 1;
    int src = s0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (todo) { const int sk = __ffsll((long long)todo) - 1; src = row == k ? sk : src; todo &= todo - 1ull; }
    }
    const float py = __shfl(qy, src, 64), pz = __shfl(qz, src, 64);
    SweepPart a;
    a.d2 = 3.0e38f; a.ry = 0.0f; a.rz = 0.0f; a.sd = -3.0e38f; a.bi = 0x7fffffff; a.di = 0x7fffffff; a.ins = 1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int idx = col + 16 * k;
      if (16 * k < n_hull) {  // (wave-uniform)
        const float4 e0 = r[k].e0;
        float wy = py - e0.x, wz = pz - e0.y;
        float cr = FMA(e0.z, wz, -(e0.w * wy));
        float s1 = -(cr * r[k].e1.y);
        const bool faces = cr < 0.0f;
        float t = FMA(wy, e0.z, wz * e0.w) * r[k].e1.x;
        t = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);
        float y1 = FMA(-t, e0.z, wy), z1 = FMA(-t, e0.w, wz);
        float q1 = FMA(y1, y1, z1 * z1);
        const bool has = idx < n_hull;
        const bool takes_sd = has & (s1 > -3.0e38f), takes_d2 = has & faces & (q1 < 3.0e38f);
        sweep_merge(a, takes_d2 ? q1 : 3.0e38f, y1, z1, takes_d2 ? idx : 0x7fffffff, takes_sd ? s1 : -3.0e38f, takes_sd ? idx : 0x7fffffff, (has & faces) ? 0 : 1);
      }
    }
    sweep_merge_ror<8>(a); sweep_merge_ror<4>(a); sweep_merge_ror<2>(a); sweep_merge_ror<1>(a);
    // every lane of a row holds its query's result now: the asking lane reads it from the row that served it
    const int from = ((rank - 4 * pass) & 3) << 4;
    const float fd2 = __shfl(a.d2, from, 64), fry = __shfl(a.ry, from, 64), frz = __shfl(a.rz, from, 64), fsd = __shfl(a.sd, from, 64);
    const int fdi = __shfl(a.di, from, 64), fins = __shfl(a.ins, from, 64);
    if (need && (rank >> 2) == pass) {
      mine.best_d2 = fd2; mine.best_ry = fry; mine.best_rz = frz; mine.max_sd = fsd; mine.deep_edge = fdi == 0x7fffffff ? 0 : fdi; mine.inside = fins != 0;
    }
  }
  return mine;
}

template <bool SCALED>
TB_DEV Hit racket_finish(const KParams& P, const float4* hull, const Racket& rk, vec3 d, float s, vec3 l, float ax, const SweepOut& so) {
  Hit h;
  const float r = P.ball_radius, thr = P.contact_threshold;
  const float sx = l.x < 0.0f ? -1.0f : 1.0f;
  float dist_hull; vec3 nl;
  if (so.inside) {
    if (ax > 0.0f || ax >= so.max_sd) { dist_hull = ax; nl = mk(sx, 0.0f, 0.0f); }
    else {
      float4 e0 = hull[2 * so.deep_edge];
      float4 e1 = hull[2 * so.deep_edge + 1];
      dist_hull = so.max_sd;
      nl = mk(0.0f, e0.w * e1.y, -(e0.z * e1.y));
    }
  } else {
    float dx = ax > 0.0f ? sx * ax : 0.0f;
    float dd = FMA(dx, dx, so.best_d2);
    dist_hull = sqrtf(dd);
    float inv = 1.0f / dist_hull;
    nl = mk(dx * inv, so.best_ry * inv, so.best_rz * inv);
  }
  h.dist = ((SCALED ? dist_hull * s : dist_hull) - P.hull_margin) - r;
  h.hit = h.dist < thr;
  h.n = rotate(rk.q, nl);
  h.rr = fma3(-(r + h.dist), h.n, d);
  return h;
}

// sphere vs axis-aligned static box centred at the origin (court.urdf:19-24 ground,
// :43-47 net; both <origin>s sit inside <geometry> and are ignored by URDF parsers)
TB_DEV Hit sphere_vs_box(const KParams& P, float hx, float hy, float hz, vec3 c) {
  Hit h; h.hit = false; h.dist = 0.0f; h.n = mk(0, 0, 0); h.rr = mk(0, 0, 0);
  const float r = P.ball_radius, thr = P.contact_threshold;
  float sx = fabsf(c.x) - hx, sy = fabsf(c.y) - hy, sz = fabsf(c.z) - hz;
  if (sx - r >= thr || sy - r >= thr || sz - r >= thr) return h;
  float gx = c.x < 0.0f ? -1.0f : 1.0f, gy = c.y < 0.0f ? -1.0f : 1.0f, gz = c.z < 0.0f ? -1.0f : 1.0f;
  bool ox = sx > 0.0f, oy = sy > 0.0f, oz = sz > 0.0f;
  int nout = (int)ox + (int)oy + (int)oz;
  float ds;
  if (nout == 0) {
    if (sz >= sx && sz >= sy) { ds = sz; h.n = mk(0.0f, 0.0f, gz); }
    else if (sx >= sy) { ds = sx; h.n = mk(gx, 0.0f, 0.0f); }
    else { ds = sy; h.n = mk(0.0f, gy, 0.0f); }
  } else if (nout == 1) {
    if (oz) { ds = sz; h.n = mk(0.0f, 0.0f, gz); }
    else if (ox) { ds = sx; h.n = mk(gx, 0.0f, 0.0f); }
    else { ds = sy; h.n = mk(0.0f, gy, 0.0f); }
  } else {
    vec3 dl = mk(ox ? gx * sx : 0.0f, oy ? gy * sy : 0.0f, oz ? gz * sz : 0.0f);
    ds = sqrtf(dot(dl, dl));
    h.n = (1.0f / ds) * dl;
  }
  h.dist = ds - r;
  h.hit = h.dist < thr;
  return h;
}

// sphere vs the goal cylinder, axis z, centred at (gx, gy, 0) (simplegoal.urdf:17-22, objects.py:99-104)
TB_DEV Hit sphere_vs_goal(const KParams& P, float gx, float gy, vec3 c) {
  Hit h; h.hit = false; h.dist = 0.0f; h.n = mk(0, 0, 0); h.rr = mk(0, 0, 0);
  const float r = P.ball_radius, thr = P.contact_threshold;
  const float RG = P.goal_radius, hl = P.goal_half_len;
  float rx = c.x - gx, ry = c.y - gy, rz = c.z;
  float sz = fabsf(rz) - hl;
  if (sz - r >= thr) return h;
  float rad2 = FMA(rx, rx, ry * ry);
  float reach = (RG + r) + thr;
  if (rad2 > reach * reach) return h;
  float rad = sqrtf(rad2);
  float sr = rad - RG;
  float gz = rz < 0.0f ? -1.0f : 1.0f;
  vec3 radial = rad > 0.0f ? mk(rx / rad, ry / rad, 0.0f) : mk(1.0f, 0.0f, 0.0f);
  float ds;
  if (sr <= 0.0f && sz <= 0.0f) {
    if (sz >= sr) { ds = sz; h.n = mk(0.0f, 0.0f, gz); }
    else { ds = sr; h.n = radial; }
  } else if (sr <= 0.0f) { ds = sz; h.n = mk(0.0f, 0.0f, gz); }
  else if (sz <= 0.0f) { ds = sr; h.n = radial; }
  else {
    ds = sqrtf(FMA(sr, sr, sz * sz));
    float inv = 1.0f / ds;
    h.n = mk(radial.x * (sr * inv), radial.y * (sr * inv), gz * (sz * inv));
  }
  h.dist = ds - r;
  h.hit = h.dist < thr;
  return h;
}

// ---------------------------------------------------------------- sequential-impulse contact rows
// Sequential impulses as in Bullet's multibody solver: normal
// row with restitution / ERP / speculative margin, two friction rows along btPlaneSpace1(n)
// boxed by mu * j_n. Slot 0 = racket pair, slots 1..3 = ground, net, goal (visited in that order,
// inactive ones skipped = the order of a compacted list). The racket row lives in registers; the
// three static rows are indexed at run time on purpose, which places them in scratch: contacts are
// rare (a handful of substeps per episode), and ~100 VGPRs of static-row state would otherwise cap
// the occupancy of every substep, free flight included. What does not change during a solve is computed once per row: the racket's
// angular responses I_w^-1 (rr x dir) and, for static pairs, the ball-only effective masses
// (host-derived constants). Same arithmetic as recomputing them every time, fewer instructions.
struct RowS {  // ball vs static shape
  vec3 n, t1, t2;
  float mu, target, jn, jt1, jt2;
};
// a static row in its lane's LDS column (word w of row i at st[(14 i + w) * stride]).
// TWO: two slots instead of three -- the net's row and the goal's share the second (substep's SF_ESC form hands an env that is near both
// to the next phase kernel before anything is stored): 28 words per lane instead of 42, a fourth wave per SIMD for the kernel
// that runs nearly all of a large batch's fast-forward substeps.
template <bool TWO> TB_DEV constexpr int row_slot(int i) { return TWO && i == 2 ? 1 : i; }
template <bool TWO = false, typename MANI> TB_DEV RowS load_row(const MANI& M, int i) {
  const float* p = M.st + 14 * row_slot<TWO>(i) * M.stride;
  const int s = M.stride;
  RowS c;
  c.n = mk(p[0], p[s], p[2 * s]); c.t1 = mk(p[3 * s], p[4 * s], p[5 * s]); c.t2 = mk(p[6 * s], p[7 * s], p[8 * s]);
  c.mu = p[9 * s]; c.target = p[10 * s]; c.jn = p[11 * s]; c.jt1 = p[12 * s]; c.jt2 = p[13 * s];
  return c;
}
template <bool TWO = false, typename MANI> TB_DEV void store_row(const MANI& M, int i, const RowS& c) {
  float* p = M.st + 14 * row_slot<TWO>(i) * M.stride;
  const int s = M.stride;
  p[0] = c.n.x; p[s] = c.n.y; p[2 * s] = c.n.z; p[3 * s] = c.t1.x; p[4 * s] = c.t1.y; p[5 * s] = c.t1.z; p[6 * s] = c.t2.x; p[7 * s] = c.t2.y; p[8 * s] = c.t2.z;
  p[9 * s] = c.mu; p[10 * s] = c.target; p[11 * s] = c.jn; p[12 * s] = c.jt1; p[13 * s] = c.jt2;
}
template <bool TWO = false, typename MANI> TB_DEV void store_row_impulses(const MANI& M, int i, const RowS& c) {
  float* p = M.st + 14 * row_slot<TWO>(i) * M.stride;
  const int s = M.stride;
  p[11 * s] = c.jn; p[12 * s] = c.jt1; p[13 * s] = c.jt2;
}
struct RowR {  // ball vs racket
  vec3 n, t1, t2, rr;
  vec3 an, at1, at2;  // I_w^-1 (rr x n), I_w^-1 (rr x t1), I_w^-1 (rr x t2)
  float mu, target, kn, kt1, kt2, jn, jt1, jt2;
};

// I_w^-1 x for a racket built with globalScaling s: Bullet derives the inertia from the collision shape
// (params.bullet_shape_inertia), so a shape scaled by s has s^2 the inertia (mass is not scaled); P holds
// the scale-1 inverse inertia, inv_s2 = 1 / (s * s). SCALED = false (SwingRacket, s == 1) drops the
// multiplications by 1.
template <bool SCALED>
TB_DEV vec3 racket_invI(const KParams& P, quat q, vec3 x, float inv_s2) {
  vec3 b = rotate_inv(q, x);
  b = mk(b.x * P.racket_inv_inertia[0], b.y * P.racket_inv_inertia[1], b.z * P.racket_inv_inertia[2]);
  if (SCALED) b = mk(b.x * inv_s2, b.y * inv_s2, b.z * inv_s2);
  return rotate(q, b);
}
// 1 / sqrt(a) for the tangent basis. a == 1 exactly for every axis-aligned normal -- the court's and the net's faces, i.e. nearly every
// static contact -- and 1 / sqrtf(1) is exactly 1 under IEEE rounding: the same value without the ~30 dependent instructions of
// a correctly rounded square root and division, which a lone wave in the contact path pays one by one.
TB_DEV float inv_sqrt_unit(float a) {
  float k = 1.0f;
  if (a != 1.0f) k = 1.0f / sqrtf(a);
  return k;
}
TB_DEV void plane_space(vec3 n, vec3& p, vec3& q) {
  if (fabsf(n.z) > 0.7071067811865475244f) {
    float a = FMA(n.y, n.y, n.z * n.z);
    float k = inv_sqrt_unit(a);
    p = mk(0.0f, -(n.z * k), n.y * k);
    q = mk(a * k, -(n.x * p.z), n.x * p.y);
  } else {
    float a = FMA(n.x, n.x, n.y * n.y);
    float k = inv_sqrt_unit(a);
    p = mk(-(n.y * k), n.x * k, 0.0f);
    q = mk(-(n.z * p.y), n.z * p.x, a * k);
  }
}
TB_DEV float contact_target(const KParams& P, float vn, float dist, float e) {
  float rest = fabsf(vn) < P.rest_vel_threshold ? 0.0f : e * (-vn);
  if (rest < 0.0f) rest = 0.0f;
  float pos = dist > 0.0f ? -(dist * P.inv_dt) : -(dist * P.erp) * P.inv_dt;
  return rest + pos;
}
// velocity of the ball's contact point (relative to the static world)
TB_DEV vec3 ball_point_vel(const Ball& b, vec3 rb) { return b.v + cross(b.w, rb); }
TB_DEV vec3 rel_vel_racket(const RowR& c, const Racket& rk, const Ball& b, vec3 rb) {
  return ball_point_vel(b, rb) - (rk.v + cross(rk.w, c.rr));
}

TB_DEV void setup_static(const KParams& P, RowS& c, const Hit& h, float e, float mu, const Ball& b) {
  c.n = h.n; c.mu = mu; c.jn = 0.0f; c.jt1 = 0.0f; c.jt2 = 0.0f;
  plane_space(c.n, c.t1, c.t2);
  vec3 rb = (-P.ball_radius) * c.n;
  c.target = contact_target(P, dot(c.n, ball_point_vel(b, rb)), h.dist, e);
}
template <bool SCALED>
TB_DEV void setup_racket(const KParams& P, RowR& c, const Hit& h, const Racket& rk, const Ball& b, float scale) {
  const float r = P.ball_radius;
  const float inv_s2 = SCALED ? 1.0f / (scale * scale) : 1.0f;
  c.n = h.n; c.rr = h.rr; c.mu = P.fric_racket; c.jn = 0.0f; c.jt1 = 0.0f; c.jt2 = 0.0f;
  plane_space(c.n, c.t1, c.t2);
  float kt = FMA(P.ball_inv_inertia, r * r, P.ball_inv_mass);
  vec3 a;
  a = cross(c.rr, c.n);  c.an = racket_invI<SCALED>(P, rk.q, a, inv_s2);  c.kn = 1.0f / ((P.ball_inv_mass + P.racket_inv_mass) + dot(a, c.an));
  a = cross(c.rr, c.t1); c.at1 = racket_invI<SCALED>(P, rk.q, a, inv_s2); c.kt1 = 1.0f / ((kt + P.racket_inv_mass) + dot(a, c.at1));
  a = cross(c.rr, c.t2); c.at2 = racket_invI<SCALED>(P, rk.q, a, inv_s2); c.kt2 = 1.0f / ((kt + P.racket_inv_mass) + dot(a, c.at2));
  vec3 rb = (-r) * c.n;
  c.target = contact_target(P, dot(c.n, rel_vel_racket(c, rk, b, rb)), h.dist, P.rest_racket);
}

// jref: the largest normal impulse seen so far in this solve -- updates are judged against it, so
// that rows of a redundant manifold trading impulses that are tiny in absolute terms do not keep
// the solve running to the iteration cap
TB_DEV bool clamp_friction(float vt, float kt, float lim, float tol, float jref, float& acc, float& d) {
  float jt = FMA(-vt, kt, acc);
  jt = jt < -lim ? -lim : (jt > lim ? lim : jt);
  d = jt - acc;
  acc = jt;
  return fabsf(d) > tol * (fabsf(jt) > jref ? fabsf(jt) : jref);
}

TB_DEV bool normal_static(const KParams& P, RowS& c, Ball& b, float& jref) {
  vec3 rb = (-P.ball_radius) * c.n;
  float vn = dot(c.n, ball_point_vel(b, rb));
  float jn = FMA(c.target - vn, P.ball_kn, c.jn);
  if (jn < 0.0f) jn = 0.0f;
  float d = jn - c.jn;
  c.jn = jn;
  if (jn > jref) jref = jn;
  if (d == 0.0f) return false;
  b.v = fma3(d * P.ball_inv_mass, c.n, b.v);
  return fabsf(d) > P.solver_tol * jref;
}
TB_DEV bool friction_static(const KParams& P, RowS& c, Ball& b, float jref) {
  float lim = c.mu * c.jn;
  if (!(lim > 0.0f)) return false;
  bool moved = false;
  vec3 rb = (-P.ball_radius) * c.n;
  float d;
  bool m = clamp_friction(dot(c.t1, ball_point_vel(b, rb)), P.ball_kt, lim, P.solver_tol, jref, c.jt1, d);
  if (d != 0.0f) { moved |= m; b.v = fma3(d * P.ball_inv_mass, c.t1, b.v); b.w = fma3(d * P.ball_inv_inertia, cross(rb, c.t1), b.w); }
  m = clamp_friction(dot(c.t2, ball_point_vel(b, rb)), P.ball_kt, lim, P.solver_tol, jref, c.jt2, d);
  if (d != 0.0f) { moved |= m; b.v = fma3(d * P.ball_inv_mass, c.t2, b.v); b.w = fma3(d * P.ball_inv_inertia, cross(rb, c.t2), b.w); }
  return moved;
}
TB_DEV bool normal_racket(const KParams& P, RowR& c, Racket& rk, Ball& b, float& jref) {
  vec3 rb = (-P.ball_radius) * c.n;
  float vn = dot(c.n, rel_vel_racket(c, rk, b, rb));
  float jn = FMA(c.target - vn, c.kn, c.jn);
  if (jn < 0.0f) jn = 0.0f;
  float d = jn - c.jn;
  c.jn = jn;
  if (jn > jref) jref = jn;
  if (d == 0.0f) return false;
  b.v = fma3(d * P.ball_inv_mass, c.n, b.v);
  rk.v = fma3(-(d * P.racket_inv_mass), c.n, rk.v);
  rk.w = fma3(-d, c.an, rk.w);
  return fabsf(d) > P.solver_tol * jref;
}
TB_DEV bool friction_racket(const KParams& P, RowR& c, Racket& rk, Ball& b, float jref) {
  float lim = c.mu * c.jn;
  if (!(lim > 0.0f)) return false;
  bool moved = false;
  vec3 rb = (-P.ball_radius) * c.n;
  float d;
  bool m = clamp_friction(dot(c.t1, rel_vel_racket(c, rk, b, rb)), c.kt1, lim, P.solver_tol, jref, c.jt1, d);
  if (d != 0.0f) {
    moved |= m;
    b.v = fma3(d * P.ball_inv_mass, c.t1, b.v); b.w = fma3(d * P.ball_inv_inertia, cross(rb, c.t1), b.w);
    rk.v = fma3(-(d * P.racket_inv_mass), c.t1, rk.v); rk.w = fma3(-d, c.at1, rk.w);
  }
  m = clamp_friction(dot(c.t2, rel_vel_racket(c, rk, b, rb)), c.kt2, lim, P.solver_tol, jref, c.jt2, d);
  if (d != 0.0f) {
    moved |= m;
    b.v = fma3(d * P.ball_inv_mass, c.t2, b.v); b.w = fma3(d * P.ball_inv_inertia, cross(rb, c.t2), b.w);
    rk.v = fma3(-(d * P.racket_inv_mass), c.t2, rk.v); rk.w = fma3(-d, c.at2, rk.w);
  }
  return moved;
}

// Rolling friction (TbParams.roll_*; opt-in, part of the RG instantiations only so that the
// default kernels carry none of it): two angular rows per ball contact along the friction directions, target
// relative spin 0, boxed by roll * j_n; the angular impulse is judged against jref through the ball radius.
struct RollS { float roll, kr, jr1, jr2; };
struct RollR { vec3 ar1, ar2; float roll, kr1, kr2, jr1, jr2; };
TB_DEV void setup_roll_static(const KParams& P, RollS& q, float roll) {
  q.roll = roll; q.jr1 = 0.0f; q.jr2 = 0.0f; q.kr = 0.0f;
  if (roll > 0.0f) q.kr = 1.0f / P.ball_inv_inertia;
}
template <bool SCALED>
TB_DEV void setup_roll_racket(const KParams& P, RollR& q, const RowR& c, const Racket& rk, float scale) {
  q.roll = P.roll_racket; q.jr1 = 0.0f; q.jr2 = 0.0f; q.kr1 = 0.0f; q.kr2 = 0.0f;
  q.ar1 = mk(0.0f, 0.0f, 0.0f); q.ar2 = mk(0.0f, 0.0f, 0.0f);
  if (q.roll > 0.0f) {
    const float inv_s2 = SCALED ? 1.0f / (scale * scale) : 1.0f;
    q.ar1 = racket_invI<SCALED>(P, rk.q, c.t1, inv_s2); q.kr1 = 1.0f / (P.ball_inv_inertia + dot(c.t1, q.ar1));
    q.ar2 = racket_invI<SCALED>(P, rk.q, c.t2, inv_s2); q.kr2 = 1.0f / (P.ball_inv_inertia + dot(c.t2, q.ar2));
  }
}
TB_DEV bool rolling_static(const KParams& P, const RowS& c, RollS& q, Ball& b, float jref) {
  float lim = q.roll * c.jn;
  if (!(lim > 0.0f)) return false;
  bool moved = false;
  const float ref = jref * P.ball_radius;
  float d;
  bool m = clamp_friction(dot(c.t1, b.w), q.kr, lim, P.solver_tol, ref, q.jr1, d);
  if (d != 0.0f) { moved |= m; b.w = fma3(d * P.ball_inv_inertia, c.t1, b.w); }
  m = clamp_friction(dot(c.t2, b.w), q.kr, lim, P.solver_tol, ref, q.jr2, d);
  if (d != 0.0f) { moved |= m; b.w = fma3(d * P.ball_inv_inertia, c.t2, b.w); }
  return moved;
}
TB_DEV bool rolling_racket(const KParams& P, const RowR& c, RollR& q, Racket& rk, Ball& b, float jref) {
  float lim = q.roll * c.jn;
  if (!(lim > 0.0f)) return false;
  bool moved = false;
  const float ref = jref * P.ball_radius;
  float d;
  bool m = clamp_friction(dot(c.t1, b.w - rk.w), q.kr1, lim, P.solver_tol, ref, q.jr1, d);
  if (d != 0.0f) { moved |= m; b.w = fma3(d * P.ball_inv_inertia, c.t1, b.w); rk.w = fma3(-d, q.ar1, rk.w); }
  m = clamp_friction(dot(c.t2, b.w - rk.w), q.kr2, lim, P.solver_tol, ref, q.jr2, d);
  if (d != 0.0f) { moved |= m; b.w = fma3(d * P.ball_inv_inertia, c.t2, b.w); rk.w = fma3(-d, q.ar2, rk.w); }
  return moved;
}

// racket vs the court's ground box (TB_F_RACKET_GROUND; court.urdf:19-24) with a PERSISTENT manifold,
// the way Bullet's convex-convex pair works [3P-recalled]: per substep the narrowphase finds ONE point -- the deepest hull
// vertex, by walking the outline downhill from where the last query ended -- and adds it to a cache of at most 4 points;
// cached points are refreshed with the new pose and dropped once they are above the manifold threshold; a fifth point
// replaces the cached one whose loss keeps the largest contact area (never the deepest); every point keeps the impulses of
// the last solve and the next solve starts from them. A racket in flight costs the bounding-sphere test (or, near the
// ground, a short support walk); a racket at rest a refresh of 4 vertices and one or two solver sweeps -- not the 76-vertex
// scan + cold 4-point solve of the stateless round-1 manifold (30 x a free-flight substep). The CPU restatement's
// racket_vs_ground (the checker, test infrastructure only) is the same algorithm operation by operation.
// WHERE THE CACHE LIVES: in LDS, one private column per lane (word w of lane t at m[w * stride + t]: conflict-free), 48
// words: per point its vertex id, the three impulses, and what a solve needs of it (height, arm, effective masses, target).
// Not registers: 14 + 4 x 19 live VGPRs for a path few lanes take would cost every substep of every kernel an occupancy
// step. Not scratch either: measured, 1 KB of scratch per lane (cache + rows + an out-of-line function's frame) cut the
// step kernels' rate by a quarter at 4096 envs and by two thirds at 1 M with the flag OFF, just by being allocated.
// Only the count and the walk's start vertex are registers. The rows' angular responses W (rr x dir) are recomputed
// from the arm where they are used (W, 6 registers, lives for one solve): same values as the oracle's stored ones.
constexpr int TB_MAX_RG = 4;
constexpr int TB_MANI_LDS = 12 * TB_MAX_RG;  // words per lane
enum { MW_ID = 0, MW_JN, MW_JT1, MW_JT2, MW_H, MW_RX, MW_RY, MW_RZ, MW_KN, MW_KT1, MW_KT2, MW_TGT };
struct Manifold {
  int n, deep;     // cached points; outline vertex the last support walk ended at
  float* m;        // this lane's column of the workgroup's LDS scratchpad
  int stride;      // lanes per workgroup
  float* st;       // (rides along: this lane's column for the three static contact rows of the instantiations that do not keep
                   //  them in registers, TB_ROWS_LDS words -- LDS for the reason given above: as scratch they were 292 B per lane
                   //  and made the fast-forward kernel move 3 x its algorithmic bytes)
};
constexpr int TB_ROWS_LDS = 14 * 3;  // words per lane
constexpr int TB_ROWS_LDS_TWO = 14 * 2;  // ... of the kernel that keeps two row slots (load_row<TWO>)
TB_DEV float& mw(const Manifold& M, int j, int w) { return M.m[(12 * j + w) * M.stride]; }
TB_DEV vec3 hull_vertex(const KParams& P, const float4* hull, int k, float s) {
  float hx = P.racket_half_thick * s;
  float4 e0 = hull[2 * (k >> 1)];
  return mk((k & 1) ? hx : -hx, e0.x * s, e0.y * s);
}
TB_DEV float vertex_height(const KParams& P, const Racket& rk, vec3 zr, vec3 v) {
  float dz = FMA(zr.x, v.x, FMA(zr.z, v.z, zr.y * v.y));
  return ((rk.p.z + dz) - P.hull_margin) - P.ground_half[2];
}
TB_DEV float outline_height(const float4* hull, vec3 zr, int i, float s) {
  float4 e0 = hull[2 * i];
  return FMA(zr.z, e0.y * s, zr.y * (e0.x * s));
}
TB_DEV bool vertex_supported(const KParams& P, const Racket& rk, vec3 xr, vec3 yr, vec3 v, float h, float thr) {
  if (!(h < thr)) return false;  // above the manifold threshold (however deep below it: see the oracle's note on thin plates)
  float wx = rk.p.x + FMA(xr.x, v.x, FMA(xr.z, v.z, xr.y * v.y)), wy = rk.p.y + FMA(yr.x, v.x, FMA(yr.z, v.z, yr.y * v.y));
  return !(fabsf(wx) > P.ground_half[0] || fabsf(wy) > P.ground_half[1]);
}
TB_DEV float area4(vec3 p0, vec3 p1, vec3 p2, vec3 p3) {  // Bullet's calcArea4Points
  vec3 c0 = cross(p0 - p1, p2 - p3), c1 = cross(p0 - p2, p1 - p3), c2 = cross(p0 - p3, p1 - p2);
  float a0 = dot(c0, c0), a1 = dot(c1, c1), a2 = dot(c2, c2);
  float m = a0 > a1 ? a0 : a1;
  return m > a2 ? m : a2;
}
// the bounding sphere is clear of the manifold threshold, or the racket's COM is below the court: no contact, cache emptied
TB_DEV bool racket_clear_of_ground(const KParams& P, const Racket& rk, float s) {
  const float top = P.ground_half[2], thr = P.racket_ground_threshold * s;
  return (rk.p.z - (P.hull_bound_radius * s + P.hull_margin)) - top >= thr || !(rk.p.z > top);
}
// updates the cache for the racket's present pose (height and arm of every cached point included); returns the number of
// points. The caller has tested racket_clear_of_ground.
TB_DEV int racket_vs_ground(const KParams& P, const float4* hull, const Racket& rk, float s, Manifold& M) {
  const float top = P.ground_half[2], thr = P.racket_ground_threshold * s;
  const int nh = P.n_hull;
  const vec3 zr = rotate_inv(rk.q, mk(0.0f, 0.0f, 1.0f));
  int i = M.deep;
  float fi = outline_height(hull, zr, i, s);
  for (int it = 0; it < nh; ++it) {
    int j = i + 1 == nh ? 0 : i + 1;
    float fj = outline_height(hull, zr, j, s);
    if (fj < fi) { i = j; fi = fj; continue; }
    j = i == 0 ? nh - 1 : i - 1;
    fj = outline_height(hull, zr, j, s);
    if (fj < fi) { i = j; fi = fj; continue; }
    break;
  }
  M.deep = i;
  const int side = zr.x > 0.0f ? 0 : 1;  // the face that looks down
  const float hx = P.racket_half_thick * s;
  const float h_deep = ((rk.p.z + FMA(zr.x, side ? hx : -hx, fi)) - P.hull_margin) - top;
  if (M.n == 0 && !(h_deep < thr)) return 0;  // nothing cached, nothing near: the common case of a racket close to the ground
  const vec3 xr = rotate_inv(rk.q, mk(1.0f, 0.0f, 0.0f)), yr = rotate_inv(rk.q, mk(0.0f, 1.0f, 0.0f));
  int n = 0;
#pragma unroll 1
  for (int j = 0; j < M.n; ++j) {  // refresh the cached points, drop the ones that have left
    const int k = __float_as_int(mw(M, j, MW_ID));
    vec3 v = hull_vertex(P, hull, k, s);
    float hj = vertex_height(P, rk, zr, v);
    if (!vertex_supported(P, rk, xr, yr, v, hj, thr)) continue;
    if (n != j) { mw(M, n, MW_ID) = mw(M, j, MW_ID); mw(M, n, MW_JN) = mw(M, j, MW_JN); mw(M, n, MW_JT1) = mw(M, j, MW_JT1); mw(M, n, MW_JT2) = mw(M, j, MW_JT2); }
    mw(M, n, MW_H) = hj;
    ++n;
  }
  M.n = n;
  const int kd = 2 * i + side;
  bool known = false;
#pragma unroll 1
  for (int j = 0; j < n; ++j) known |= __float_as_int(mw(M, j, MW_ID)) == kd;
  const vec3 vd = hull_vertex(P, hull, kd, s);
  if (!known && vertex_supported(P, rk, xr, yr, vd, h_deep, thr)) {
    int slot = n;
    if (n == TB_MAX_RG) {  // full: the new point replaces the cached one (never the deepest) whose loss keeps the largest area
      int deepest = 0;
      float hd = mw(M, 0, MW_H);
#pragma unroll 1
      for (int j = 1; j < TB_MAX_RG; ++j) { float hj = mw(M, j, MW_H); if (hj < hd) { deepest = j; hd = hj; } }
      if (h_deep < hd) deepest = -1;
      const vec3 c0 = hull_vertex(P, hull, __float_as_int(mw(M, 0, MW_ID)), s), c1 = hull_vertex(P, hull, __float_as_int(mw(M, 1, MW_ID)), s),
                 c2 = hull_vertex(P, hull, __float_as_int(mw(M, 2, MW_ID)), s), c3 = hull_vertex(P, hull, __float_as_int(mw(M, 3, MW_ID)), s);
      float best = -1.0f;
      slot = -1;
#pragma unroll 1
      for (int j = 0; j < TB_MAX_RG; ++j) {
        if (j == deepest) continue;
        float a = area4(j == 0 ? vd : c0, j == 1 ? vd : c1, j == 2 ? vd : c2, j == 3 ? vd : c3);
        if (a > best) { best = a; slot = j; }
      }
    } else M.n = n + 1;
    mw(M, slot, MW_ID) = __int_as_float(kd); mw(M, slot, MW_JN) = 0.0f; mw(M, slot, MW_JT1) = 0.0f; mw(M, slot, MW_JT2) = 0.0f; mw(M, slot, MW_H) = h_deep;
  }
#pragma unroll 1
  for (int j = 0; j < M.n; ++j) {
    vec3 v = hull_vertex(P, hull, __float_as_int(mw(M, j, MW_ID)), s);
    vec3 r = rotate(rk.q, v);
    mw(M, j, MW_RX) = r.x; mw(M, j, MW_RY) = r.y; mw(M, j, MW_RZ) = r.z - P.hull_margin;  // the point on the inflated hull
  }
  return M.n;
}

// World-frame inverse inertia W = R diag(I^-1 / s^2) R^T (6 unique entries) for the racket<->court rows: with up to
// 4 points x 3 directions per solve, one matrix per substep is cheaper than 12 rotate-scale-rotate round trips.
struct Sym3 { float xx, xy, xz, yy, yz, zz; };
TB_DEV Sym3 world_inv_inertia(const KParams& P, quat q, float inv_s2) {
  float x2 = q.x + q.x, y2 = q.y + q.y, z2 = q.z + q.z;
  float xx = q.x * x2, xy = q.x * y2, xz = q.x * z2, yy = q.y * y2, yz = q.y * z2, zz = q.z * z2, wx = q.w * x2, wy = q.w * y2, wz = q.w * z2;
  float r00 = 1.0f - (yy + zz), r01 = xy - wz, r02 = xz + wy, r10 = xy + wz, r11 = 1.0f - (xx + zz), r12 = yz - wx, r20 = xz - wy, r21 = yz + wx, r22 = 1.0f - (xx + yy);
  float d0 = P.racket_inv_inertia[0] * inv_s2, d1 = P.racket_inv_inertia[1] * inv_s2, d2 = P.racket_inv_inertia[2] * inv_s2;
  Sym3 W;
  W.xx = FMA(r02 * d2, r02, FMA(r01 * d1, r01, (r00 * d0) * r00));
  W.xy = FMA(r02 * d2, r12, FMA(r01 * d1, r11, (r00 * d0) * r10));
  W.xz = FMA(r02 * d2, r22, FMA(r01 * d1, r21, (r00 * d0) * r20));
  W.yy = FMA(r12 * d2, r12, FMA(r11 * d1, r11, (r10 * d0) * r10));
  W.yz = FMA(r12 * d2, r22, FMA(r11 * d1, r21, (r10 * d0) * r20));
  W.zz = FMA(r22 * d2, r22, FMA(r21 * d1, r21, (r20 * d0) * r20));
  return W;
}
TB_DEV vec3 sym3_mul(const Sym3& W, vec3 a) {
  return mk(FMA(W.xz, a.z, FMA(W.xy, a.y, W.xx * a.x)), FMA(W.yz, a.z, FMA(W.yy, a.y, W.xy * a.x)), FMA(W.zz, a.z, FMA(W.yz, a.y, W.xz * a.x)));
}
// one racket<->court row: normal +z, friction directions btPlaneSpace1(+z) = (0,-1,0), (1,0,0); rr x n = (rr.y, -rr.x, 0),
// rr x t1 = (rr.z, 0, -rr.x), rr x t2 = (0, rr.z, -rr.y)
TB_DEV vec3 mani_arm(const Manifold& M, int j) { return mk(mw(M, j, MW_RX), mw(M, j, MW_RY), mw(M, j, MW_RZ)); }
TB_DEV void setup_ground_row(const KParams& P, const Manifold& M, int j, const Sym3& W, const Racket& rk) {
  const vec3 rr = mani_arm(M, j);
  const float dist = mw(M, j, MW_H);
  vec3 a = mk(rr.y, -rr.x, 0.0f);
  mw(M, j, MW_KN) = 1.0f / (P.racket_inv_mass + dot(a, sym3_mul(W, a)));
  a = mk(rr.z, 0.0f, -rr.x);
  mw(M, j, MW_KT1) = 1.0f / (P.racket_inv_mass + dot(a, sym3_mul(W, a)));
  a = mk(0.0f, rr.z, -rr.y);
  mw(M, j, MW_KT2) = 1.0f / (P.racket_inv_mass + dot(a, sym3_mul(W, a)));
  vec3 pv = rk.v + cross(rk.w, rr);
  float vn = pv.z;
  float rest = fabsf(vn) < P.rest_vel_threshold ? 0.0f : P.rest_racket_court * (-vn);
  if (rest < 0.0f) rest = 0.0f;
  float pos = dist > 0.0f ? -(dist * P.inv_dt) : -(dist * P.erp) * P.inv_dt;
  mw(M, j, MW_TGT) = rest + pos;
}
// The racket<->court rows of ONE solve, seen through an accessor: LdsGround reads and writes the lane's LDS column at every
// use (what every kernel did until round 3: no registers held for a path few lanes take); RegGround is a register copy made
// once per solve (GroundRegs, statically indexed from unrolled loops) and written back at its end. Same arithmetic, same
// order: bit-identical. Why both: a row update through LDS is ~12 DEPENDENT LDS round trips (~100 cycles each); a lane with a
// racket at rest on the court and the ball on top of it -- 9 of 4096 random-action envs run like that to the 800-substep
// limit, 3-5 sweeps x 5 rows x 3 directions per substep -- spent 22-60 us per substep on them, alone in its wave: fast-forward
// kernels of 17-47 ms at 4096 envs (profiles/r03_racket_ground.md). Where one wave per SIMD is all there is anyway (small
// batches) the registers cost nothing; the large-batch instantiations (BIG) keep the LDS form for occupancy.
struct GroundRegs { float rx[TB_MAX_RG], ry[TB_MAX_RG], rz[TB_MAX_RG], kn[TB_MAX_RG], kt1[TB_MAX_RG], kt2[TB_MAX_RG], tgt[TB_MAX_RG], jn[TB_MAX_RG], jt1[TB_MAX_RG], jt2[TB_MAX_RG]; };
struct LdsGround {
  const Manifold& M; int j;
  TB_DEV vec3 arm() const { return mani_arm(M, j); }
  TB_DEV float kn() const { return mw(M, j, MW_KN); }
  TB_DEV float kt1() const { return mw(M, j, MW_KT1); }
  TB_DEV float kt2() const { return mw(M, j, MW_KT2); }
  TB_DEV float tgt() const { return mw(M, j, MW_TGT); }
  TB_DEV float jn() const { return mw(M, j, MW_JN); }
  TB_DEV float jt1() const { return mw(M, j, MW_JT1); }
  TB_DEV float jt2() const { return mw(M, j, MW_JT2); }
  TB_DEV void set_jn(float v) const { mw(M, j, MW_JN) = v; }
  TB_DEV void set_jt1(float v) const { mw(M, j, MW_JT1) = v; }
  TB_DEV void set_jt2(float v) const { mw(M, j, MW_JT2) = v; }
};
template <int J> struct RegGround {
  GroundRegs& G;
  TB_DEV vec3 arm() const { return mk(G.rx[J], G.ry[J], G.rz[J]); }
  TB_DEV float kn() const { return G.kn[J]; }
  TB_DEV float kt1() const { return G.kt1[J]; }
  TB_DEV float kt2() const { return G.kt2[J]; }
  TB_DEV float tgt() const { return G.tgt[J]; }
  TB_DEV float jn() const { return G.jn[J]; }
  TB_DEV float jt1() const { return G.jt1[J]; }
  TB_DEV float jt2() const { return G.jt2[J]; }
  TB_DEV void set_jn(float v) const { G.jn[J] = v; }
  TB_DEV void set_jt1(float v) const { G.jt1[J] = v; }
  TB_DEV void set_jt2(float v) const { G.jt2[J] = v; }
};
template <int J> TB_DEV void load_ground_row(const Manifold& M, GroundRegs& G) {
  G.rx[J] = mw(M, J, MW_RX); G.ry[J] = mw(M, J, MW_RY); G.rz[J] = mw(M, J, MW_RZ);
  G.kn[J] = mw(M, J, MW_KN); G.kt1[J] = mw(M, J, MW_KT1); G.kt2[J] = mw(M, J, MW_KT2); G.tgt[J] = mw(M, J, MW_TGT);
  G.jn[J] = mw(M, J, MW_JN); G.jt1[J] = mw(M, J, MW_JT1); G.jt2[J] = mw(M, J, MW_JT2);
}
template <int J> TB_DEV void store_ground_row(const Manifold& M, const GroundRegs& G) {
  mw(M, J, MW_JN) = G.jn[J]; mw(M, J, MW_JT1) = G.jt1[J]; mw(M, J, MW_JT2) = G.jt2[J];
}
template <class Row>
TB_DEV void warm_start_ground(const KParams& P, const Row& c, const Sym3& W, Racket& rk, float& jref) {
  const vec3 rr = c.arm();
  const float jn = c.jn(), jt1 = c.jt1(), jt2 = c.jt2();
  if (jn > jref) jref = jn;
  rk.v.z = FMA(jn, P.racket_inv_mass, rk.v.z);   rk.w = fma3(jn, sym3_mul(W, mk(rr.y, -rr.x, 0.0f)), rk.w);
  rk.v.y = FMA(-jt1, P.racket_inv_mass, rk.v.y); rk.w = fma3(jt1, sym3_mul(W, mk(rr.z, 0.0f, -rr.x)), rk.w);
  rk.v.x = FMA(jt2, P.racket_inv_mass, rk.v.x);  rk.w = fma3(jt2, sym3_mul(W, mk(0.0f, rr.z, -rr.y)), rk.w);
}
template <class Row>
TB_DEV bool normal_ground(const KParams& P, const Row& c, const Sym3& W, Racket& rk, float& jref) {
  const vec3 rr = c.arm();
  const float old = c.jn();
  float vn = (rk.v + cross(rk.w, rr)).z;
  float jn = FMA(c.tgt() - vn, c.kn(), old);
  if (jn < 0.0f) jn = 0.0f;
  float d = jn - old;
  c.set_jn(jn);
  if (jn > jref) jref = jn;
  if (d == 0.0f) return false;
  rk.v.z = FMA(d, P.racket_inv_mass, rk.v.z);
  rk.w = fma3(d, sym3_mul(W, mk(rr.y, -rr.x, 0.0f)), rk.w);
  return fabsf(d) > P.solver_tol * jref;
}
template <class Row>
TB_DEV bool friction_ground(const KParams& P, const Row& c, const Sym3& W, Racket& rk, float jref) {
  float lim = P.fric_racket_court * c.jn();
  if (!(lim > 0.0f)) return false;
  const vec3 rr = c.arm();
  bool moved = false;
  float d, acc = c.jt1();
  bool m = clamp_friction(-((rk.v + cross(rk.w, rr)).y), c.kt1(), lim, P.solver_tol, jref, acc, d);
  c.set_jt1(acc);
  if (d != 0.0f) { moved |= m; rk.v.y = FMA(-d, P.racket_inv_mass, rk.v.y); rk.w = fma3(d, sym3_mul(W, mk(rr.z, 0.0f, -rr.x)), rk.w); }
  acc = c.jt2();
  m = clamp_friction((rk.v + cross(rk.w, rr)).x, c.kt2(), lim, P.solver_tol, jref, acc, d);
  c.set_jt2(acc);
  if (d != 0.0f) { moved |= m; rk.v.x = FMA(d, P.racket_inv_mass, rk.v.x); rk.w = fma3(d, sym3_mul(W, mk(0.0f, rr.z, -rr.y)), rk.w); }
  return moved;
}

// RG = the kernel was instantiated with the extended contact set: racket<->court contact (TB_F_RACKET_GROUND; its cache and
// rows live in LDS, see Manifold) and the rolling-friction rows (TbParams.roll_*); run-time flags / coefficients pick either.
// The default kernels carry neither: measured with both compiled in but switched off, the step kernels lost 3.5 % at 4096
// envs and SwingRacket 18 % at 1 M (195 instead of 166 VGPRs in the fast-forward loop: a wave per SIMD less).
template <bool RG> struct Rows;
template <> struct Rows<false> { RowR rk; RowS st[3]; int on; };  // on: bit k = row k is active (a dynamically indexed bool array would put the whole struct in scratch)
template <> struct Rows<true> { RowR rk; RowS st[3]; int on; RollR qrk; RollS qst[3]; };

// REGROWS: the three static rows statically indexed too, i.e. in registers (~40 VGPRs more): the right
// trade where balls bounce on the court all the time -- Tennisbot at every batch size (+8 ... +28 % in
// the steady state; tb_create decides) and SwingRacket's loop-free pipelined step kernel up to 131072
// envs (+2.7 % at 4096); in SwingRacket's fast-forward loop the registers cost more than they give
// (-6 % at 4096 envs, -17 % at 1 M). Same arithmetic either way.
// REGGROUND: the racket<->court rows of this solve in registers (RegGround) instead of the lane's LDS column (LdsGround)
#define TB_EACH_GROUND_ROW(CALL_REG, CALL_LDS)                                         \
  do {                                                                                  \
    if constexpr (REGGROUND) {                                                          \
      if (0 < nrg) { const RegGround<0> c{G}; CALL_REG; }                               \
      if (1 < nrg) { const RegGround<1> c{G}; CALL_REG; }                               \
      if (2 < nrg) { const RegGround<2> c{G}; CALL_REG; }                               \
      if (3 < nrg) { const RegGround<3> c{G}; CALL_REG; }                               \
    } else {                                                                            \
      _Pragma("unroll 1") for (int j = 0; j < nrg; ++j) { const LdsGround c{M, j}; CALL_LDS; } \
    }                                                                                   \
  } while (0)
template <bool RG, bool REGROWS, bool TWO = false, bool REGGROUND = false>
TB_DEV void solve_contacts(const KParams& P, Rows<RG>& R, const Manifold& M, int nrg, const Sym3& W, Racket& rk, Ball& b) {
  static_assert(TB_MAX_RG == 4, "TB_EACH_GROUND_ROW unrolls four rows");
  float jref = 0.0f;
  GroundRegs G;
  if constexpr (RG && REGGROUND) {
    if (0 < nrg) load_ground_row<0>(M, G);
    if (1 < nrg) load_ground_row<1>(M, G);
    if (2 < nrg) load_ground_row<2>(M, G);
    if (3 < nrg) load_ground_row<3>(M, G);
  }
  if constexpr (RG) {
    TB_EACH_GROUND_ROW(warm_start_ground(P, c, W, rk, jref), warm_start_ground(P, c, W, rk, jref));  // the cached impulses of the last solve
  }
  for (int it = 0; it < P.solver_iters; ++it) {
    bool moved = false;
    if ((R.on & 1)) moved |= normal_racket(P, R.rk, rk, b, jref);
    if constexpr (REGROWS) {
#pragma unroll
      for (int i = 0; i < 3; ++i)
        if (((R.on >> (i + 1)) & 1)) moved |= normal_static(P, R.st[i], b, jref);
    } else {
#pragma unroll 1
      for (int i = 0; i < 3; ++i)
        if (((R.on >> (i + 1)) & 1)) { RowS c = load_row<TWO>(M, i); moved |= normal_static(P, c, b, jref); store_row_impulses<TWO>(M, i, c); }
    }
    if constexpr (RG) {
      TB_EACH_GROUND_ROW(moved |= normal_ground(P, c, W, rk, jref), moved |= normal_ground(P, c, W, rk, jref));
    }
    if constexpr (RG) {  // rolling rows: after the normals, before sliding friction
      if ((R.on & 1)) moved |= rolling_racket(P, R.rk, R.qrk, rk, b, jref);
#pragma unroll 1
      for (int i = 0; i < 3; ++i)
        if (((R.on >> (i + 1)) & 1)) {
          if constexpr (REGROWS) moved |= rolling_static(P, R.st[i], R.qst[i], b, jref);
          else { RowS c = load_row<TWO>(M, i); moved |= rolling_static(P, c, R.qst[i], b, jref); }
        }
    }
    if ((R.on & 1)) moved |= friction_racket(P, R.rk, rk, b, jref);
    if constexpr (REGROWS) {
#pragma unroll
      for (int i = 0; i < 3; ++i)
        if (((R.on >> (i + 1)) & 1)) moved |= friction_static(P, R.st[i], b, jref);
    } else {
#pragma unroll 1
      for (int i = 0; i < 3; ++i)
        if (((R.on >> (i + 1)) & 1)) { RowS c = load_row<TWO>(M, i); moved |= friction_static(P, c, b, jref); store_row_impulses<TWO>(M, i, c); }
    }
    if constexpr (RG) {
      TB_EACH_GROUND_ROW(moved |= friction_ground(P, c, W, rk, jref), moved |= friction_ground(P, c, W, rk, jref));
    }
    if (!moved) break;
  }
  if constexpr (RG && REGGROUND) {
    if (0 < nrg) store_ground_row<0>(M, G);
    if (1 < nrg) store_ground_row<1>(M, G);
    if (2 < nrg) store_ground_row<2>(M, G);
    if (3 < nrg) store_ground_row<3>(M, G);
  }
}
#undef TB_EACH_GROUND_ROW

// ---------------------------------------------------------------- one 1/240 s substep
// wb_pre = rotate_inv(rk.q, rk.w), computed by the caller (rotate_inv2)
// LAZY (the large-batch fast-forward instantiations, where VALU issue slots count and not latency): the ball's spin rate is
// only taken where a ball spins -- after a contact, i.e. for a few lanes -- instead of next to the two speeds
template <bool LAZY = false>
TB_DEV void integrate_velocities(const KParams& P, Racket& rk, Ball& b, vec3 Fr, vec3 Tr, vec3 Fb, vec3 wb_pre) {
  const float dt = P.dt, g = P.gravity;
  // the three speeds that do not wait for anything are taken first, side by side: a correctly rounded sqrtf is a
  // ~16-instruction dependent chain, and three independent chains in one block interleave where three chains behind
  // three branches queue (the values and every operation on them are the same as before: bit-identical)
  const float speed_r = sqrtf(dot(rk.v, rk.v)), speed_b = sqrtf(dot(b.v, b.v));
  float spin_b = 0.0f;
  if constexpr (!LAZY) spin_b = sqrtf(dot(b.w, b.w));
  {
    float kd = FMA(P.lin_damp_quad, speed_r, P.lin_damp);
    vec3 a = mk(FMA(Fr.x, P.racket_inv_mass, -(rk.v.x * kd)), FMA(Fr.y, P.racket_inv_mass, -(rk.v.y * kd)),
                FMA(Fr.z, P.racket_inv_mass, -(rk.v.z * kd)) - g);
    rk.v = fma3(dt, a, rk.v);
    // a racket that neither spins nor is torqued has zero angular acceleration: skip the
    // body-frame round trip (Tennisbot rackets until they are hit; every fast-forward substep
    // of a racket that was never torqued)
    const bool torqued = nonzero3(Tr);
    bool active = nonzero3(rk.w) | torqued;
    TB_DIAG_ABLATE_ANGULAR(active);
    if (active) {
      vec3 wb = wb_pre, Tb = mk(0.0f, 0.0f, 0.0f);
      if (torqued) Tb = rotate_inv(rk.q, Tr);
      vec3 L = mk(P.racket_inertia[0] * wb.x, P.racket_inertia[1] * wb.y, P.racket_inertia[2] * wb.z);
      vec3 gy = cross(wb, L);
      float ka = FMA(P.ang_damp_quad, sqrtf(dot(wb, wb)), P.ang_damp);
      vec3 ab = mk(P.racket_inv_inertia[0] * ((Tb.x - gy.x) - L.x * ka), P.racket_inv_inertia[1] * ((Tb.y - gy.y) - L.y * ka),
                   P.racket_inv_inertia[2] * ((Tb.z - gy.z) - L.z * ka));
      rk.w = fma3(dt, rotate(rk.q, ab), rk.w);
    }
  }
  {
    if (P.magnus_k != 0.0f) Fb = fma3(P.magnus_k, cross(b.w, b.v), Fb);
    float kd = FMA(P.lin_damp_quad, speed_b, P.lin_damp);
    vec3 a = mk(FMA(Fb.x, P.ball_inv_mass, -(b.v.x * kd)), FMA(Fb.y, P.ball_inv_mass, -(b.v.y * kd)),
                FMA(Fb.z, P.ball_inv_mass, -(b.v.z * kd)) - g);
    b.v = fma3(dt, a, b.v);
    bool spinning = nonzero3(b.w);
    if (spinning) {
      if constexpr (LAZY) spin_b = sqrtf(dot(b.w, b.w));
      float ka = FMA(P.ang_damp_quad, spin_b, P.ang_damp);
      vec3 aw = mk(-(b.w.x * ka), -(b.w.y * ka), -(b.w.z * ka));
      b.w = fma3(dt, aw, b.w);
    }
  }
}

// exponential-map orientation update with Bullet's pi/4 clamp quirk (w is not rescaled, only
// z = x^2 is clamped); unclamped |q'|^2 = 1 + O(eps): one Newton step of 1/sqrt normalises
TB_DEV void integrate_pose(const KParams& P, Racket& rk, Ball& b) {
  const float dt = P.dt;
  rk.p = fma3(dt, rk.v, rk.p);
  b.p = fma3(dt, b.v, b.p);
  float w2 = dot(rk.w, rk.w);
  TB_DIAG_ABLATE_ORIENT(w2);
  if (w2 > 0.0f) {
    float h = 0.5f * dt, hm = 0.5f * P.max_ang_step;
    float z = (h * h) * w2, zc = hm * hm;
    bool clamped = z > zc;
    if (clamped) z = zc;
    float s = h * sinc_half(z);
    quat dq; dq.x = rk.w.x * s; dq.y = rk.w.y * s; dq.z = rk.w.z * s; dq.w = cos_half(z);
    quat q = qmul(dq, rk.q);
    float n2 = FMA(q.w, q.w, FMA(q.z, q.z, FMA(q.y, q.y, q.x * q.x)));
    float inv;
    if (__builtin_expect(clamped, 0)) inv = 1.0f / sqrtf(n2);
    else inv = FMA(-0.5f, n2, 1.5f);
    rk.q.x = q.x * inv; rk.q.y = q.y * inv; rk.q.z = q.z * inv; rk.q.w = q.w * inv;
  }
}

// Per-shape culls in front of the exact static tests (1 mm of slack against rounding: each is implied by the test's own early-out):
// the ball's lowest point clears the shape's top by the manifold threshold, or -- the net, 1 m high but 19 cm thick and 15 m
// away from where most balls fly -- its x separation alone does. One z test against the highest shape (the net: every ball
// below 0.53 m) sent a quarter of all fast-forward substeps through the three tests; the ground's own 4 cm band is two.
TB_DEV float ball_low_point(const KParams& P, const Ball& b) { return (b.p.z - P.ball_radius) - P.contact_threshold; }
TB_DEV bool near_ground(const KParams& P, float zlow) { return !(zlow >= P.ground_half[2] + 1.0e-3f); }
TB_DEV bool near_net(const KParams& P, const Ball& b, float zlow) {
  return (P.flags & TB_F_NET) && !(zlow >= P.net_half[2] + 1.0e-3f) && !(((fabsf(b.p.x) - P.net_half[0]) - P.ball_radius) >= P.contact_threshold + 1.0e-3f);
}
template <int KIND> TB_DEV bool near_goal(const KParams& P, float zlow) { return KIND == TB_ENV_SWING && !(zlow >= P.goal_half_len + 1.0e-3f); }

// returns the contact bits of this substep's manifold (the `len(getContactPoints) > 0` tests)
//
// Free flight is the common case (a SwingRacket fast-forward is <= 775 substeps of it), so
// the narrowphase is arranged as cheap per-lane culls + wave votes: a wave runs the outline
// sweep / the static tests / the impulse solver only if __any lane needs them, and those
// branches are wave-uniform (s_cbranch on the ballot), never if-converted into the hot path.
// SF_COLD: the contact path reads its constants from the LDS copy of the parameter block instead of holding them in SGPRs all the
// time. Pays where SGPRs are scarce and contacts rare (the policy rollout kernel: 97 -> 70 spill writes, collect +10 %); costs VGPRs and
// LDS reads where throughput counts (SwingRacket at 1 M envs -15 %, Tennisbot -4 %), so only that kernel asks for it.
// SF_ESC (first phase of the large-batch fast-forward): a lane whose ball gets past the racket's culls -- it needs the exact outline
// sweep, and probably a racket row in the solver next -- does not run them here, where 63 other lanes would wait for it (0.2 % of
// the lanes ask, but every ninth wave-substep has one: sweep + racket row + its solve are ~15 % of the loop's instructions).
// It returns CT_ESCAPE with the env UNTOUCHED (the culls read poses only and come first), the caller hands the env to the next
// phase kernel as a survivor, and that kernel repeats this substep with everything compiled in -- among lanes that mostly want
// the same. Same arithmetic, same order per env: bit-identical.
// SF_REGGROUND: see solve_contacts (the looping small-batch kernels ask for it)
// SF_LAZYTAB (the pipelined SwingRacket step kernel: ONE substep per launch): `hull`, the LDS copy of the outline table, is EMPTY on
// entry. Nothing reads it before a ball gets past the racket's slab test -- in the 25 short steps of a random-action episode none
// does -- so the copy (2.5 KB from `table_mem`, by the wave that needs it, no barrier: a wave's LDS operations complete in order)
// is made right there, behind a wave vote, instead of by every launch up front (0.4 us of a ~4 us launch, tools/diag/lanes_per_wave.hip).
// The forms are chosen by ONE template argument, a mask of SF_* bits (named at every call site; each form is explained above):
enum : unsigned {
  SF_RG = 1u,          // the extended contact set compiled in: racket<->court manifold, rolling-friction rows
  SF_REGROWS = 2u,     // the three static contact rows in registers instead of the lane's LDS column
  SF_COLD = 4u,        // contact-path constants from the LDS copy of the parameter block (the policy rollout kernels' env wave)
  SF_RELOAD = 8u,      // large-batch fast-forward: cull planes re-read per call, outline sweep shared by 8 helper lanes, lazy spin rate
  SF_ESC = 16u,        // first phase of the large-batch fast-forward: hand over instead of sweeping (CT_ESCAPE)
  SF_REGGROUND = 32u,  // the racket<->court rows of a solve in registers (small-batch kernels that loop)
  SF_LAZYTAB = 64u,    // the LDS outline table is copied by the first wave that reads it (pipelined SwingRacket step kernel)
  SF_WIDE = 128u,      // all 64 lanes of the wave are in the substep: the outline sweep is shared, four queries at a time (outline_sweep_rows)
};
template <int KIND, unsigned FORM>
TB_DEV int substep(const KParams& P, const float4* hull, Racket& rk, Ball& b, Manifold& M, vec3 Fr, vec3 Tr, vec3 Fb, float goal_x, float goal_y, float scale TB_STAMP_ARG,
                   const float4* table_mem = nullptr) {
  constexpr bool RG = (FORM & SF_RG) != 0, REGROWS = (FORM & SF_REGROWS) != 0, COLD = (FORM & SF_COLD) != 0, RELOAD = (FORM & SF_RELOAD) != 0,
                 ESC = (FORM & SF_ESC) != 0, REGGROUND = (FORM & SF_REGGROUND) != 0, LAZYTAB = (FORM & SF_LAZYTAB) != 0, WIDE = (FORM & SF_WIDE) != 0;
  static_assert(!LAZYTAB || (!ESC && !RELOAD && !COLD && !RG), "the lazily copied table serves the plain one-substep kernel only");
  static_assert(!WIDE || !RELOAD, "one form of shared sweep at a time");
  int bits = 0;
  TB_STAMP(st, 0);  // everything between two substeps (loop control, env logic)
  constexpr bool TWO = ESC && !RG && !REGROWS;  // the static rows in two LDS slots, see load_row
  if constexpr (TWO) {  // (poses only, like the racket's culls: the very tests the static narrowphase below starts with)
    const float zl = ball_low_point(P, b);
    if (near_net(P, b, zl) && near_goal<KIND>(P, zl)) return CT_ESCAPE;  // both rows of the shared slot could be wanted
  }
  vec3 wb0 = mk(0.0f, 0.0f, 0.0f);
  if constexpr (ESC) {
    const vec3 d0 = b.p - rk.p;
    const bool reach = (P.flags & TB_F_RACKET_BALL) && racket_in_reach(P, d0, scale);
    // the racket-frame ball offset (for the culls) and the body-frame spin (for the velocity update below) are rotations by the
    // same quaternion: done together, two per packed instruction
    vec3 l0;
    rotate_inv2(rk.q, d0, rk.w, l0, wb0);
    TB_LANES(2, reach);  // (census: this instantiation's near_racket below is constant false)
    if (__any(reach)) {
      vec3 ql = mk(0.0f, 0.0f, 0.0f);
      float qax = 0.0f;
      if (reach && racket_cull<KIND == TB_ENV_TENNIS, RELOAD>(P, hull, l0, scale, ql, qax)) return CT_ESCAPE;
    }
    TB_STAMP(st, 1);
  }
  // the velocity update touches velocities only and the narrowphase reads poses only: their order
  // is free, and running it first keeps the Hit records from staying live across it
  // racket <-> court: poses only, like every narrowphase, so it may run before the velocity update as well; the common case
  // is the bounding-sphere test failing (racket in flight, or hovering as in Tennisbot)
  int nrg = 0;
  if constexpr (RG) {  // the RG instantiations also serve rolling friction alone: the flag decides at run time
    if (P.flags & TB_F_RACKET_GROUND) {
      if (racket_clear_of_ground(P, rk, scale)) M.n = 0;
      else nrg = racket_vs_ground(P, hull, rk, scale, M);
    }
  }
  vec3 lp = mk(0.0f, 0.0f, 0.0f);
  if constexpr (!ESC) rotate_inv2(rk.q, b.p - rk.p, rk.w, lp, wb0);  // (as above; 4096 envs, same box: 612-619 -> 660-694 M env steps/s)
  integrate_velocities<RELOAD>(P, rk, b, Fr, Tr, Fb, wb0);
  TB_STAMP(st, 3);  // velocity update
  Hit hr, hg, hn, hc;
  hr.hit = false; hg.hit = false; hn.hit = false; hc.hit = false;

  vec3 d = b.p - rk.p;
  bool near_racket = !ESC && (P.flags & TB_F_RACKET_BALL) && racket_in_reach(P, d, scale);
  TB_DIAG_ABLATE_NARROW(near_racket);
  TB_LANES(0, true);          // [0] active lanes, [1] wave-substeps
  TB_LANES(2, near_racket);   // [2] lanes inside the racket's bounding sphere, [3] wave-substeps with one
  if (__any(near_racket)) {
    TB_DIAG_ADD_LEADER(12, 1);  // wave-substeps with a lane in reach
    vec3 ql = mk(0.0f, 0.0f, 0.0f);
    float qax = 0.0f;
    bool need = false;
    if constexpr (LAZYTAB) {
      bool past_slab = false;
      if (near_racket) past_slab = racket_slab<KIND == TB_ENV_TENNIS>(P, lp, scale, ql, qax);
      if (__any(past_slab)) {  // the first reader of the table in this launch: this wave copies it (another wave of the workgroup may be
                               // writing the same values to the same places)
        // (shared among the lanes that are HERE: the last wave of a batch that is no multiple of 64 has fewer than 64, and a copy
        //  strided by 64 left the entries of its missing lanes unwritten -- found by test_outline_sweep_with_other_outlines)
        float4* dst = const_cast<float4*>(hull);
        const unsigned long long here = __ballot(1);
        const int lane = (int)(threadIdx.x & 63), na = __popcll(here), rank = __popcll(here & ((1ull << lane) - 1ull));
        for (int k = rank; k < 2 * P.n_hull; k += na) dst[k] = table_mem[k];
        for (int k = TB_HULL_PLANES + rank; k < TB_HULL_LDS; k += na) dst[k] = table_mem[k];
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // every lane's writes are in LDS before any lane reads another's
        if (past_slab) {
          TB_LANES_ADD1(12);
          need = racket_planes<KIND == TB_ENV_TENNIS, false>(P, hull, ql, scale);
          if (need) { TB_DIAG_ADD_EACH(10, 1); TB_DIAG_ADD_LEADER(11, 1); }
        }
      }
    } else if (near_racket) need = racket_cull<KIND == TB_ENV_TENNIS, RELOAD>(P, hull, lp, scale, ql, qax);
    // the cooperative form where throughput counts (the large-batch fast-forward instantiation, RELOAD): +4.7 % at 1 M envs, same
    // box; at 4096 envs it shortens the fast-forward (0.47 -> 0.44 ms per lone episode) but its busier waves take more from the
    // step kernels beside them than that gives back (702 -> 655 M env steps/s), and in the loop-free step kernels its ballot masks
    // cost SGPR spills at kernel start (-10 %)
    TB_LANES(4, need);        // [4] lanes that need the outline sweep, [5] wave-substeps with one
    if constexpr (WIDE) {  // all 64 lanes are here (the policy rollout kernels' env wave: lanes without an env step a dummy) and share the sweeps, four queries at a time
      if (__any(need)) {
        const SweepOut so = outline_sweep_rows(hull, P.n_hull, need, ql.y, ql.z);
        if (need) hr = racket_finish<KIND == TB_ENV_TENNIS>(P, hull, rk, d, scale, ql, qax, so);
      }
    } else if constexpr (!RELOAD) {  // each lane for itself
      if (need) hr = racket_finish<KIND == TB_ENV_TENNIS>(P, hull, rk, d, scale, ql, qax, outline_sweep_serial(hull, P.n_hull, ql.y, ql.z));
    } else if (__any(need)) {  // the sweep is shared by the wave: every active lane goes in
      const SweepOut so = outline_sweep(hull, P.n_hull, need, ql.y, ql.z);
      if (need) hr = racket_finish<KIND == TB_ENV_TENNIS>(P, hull, rk, d, scale, ql, qax, so);
    }
  }
  TB_DIAG_ADD_LEADER(13, 1);  // wave-substeps
  TB_STAMP(st, 1);  // racket narrowphase
  const float zlow = ball_low_point(P, b);
  bool near_g = false, near_n = false, near_c = false;
  // every per-shape cull below needs the ball's low point under that shape's top: one wave vote against the highest of them
  // (host-derived static_top) skips the three tests for waves whose balls are all still up in the air
  if (__any(!(zlow >= P.static_top + 1.0e-3f))) {
    near_g = near_ground(P, zlow); near_n = near_net(P, b, zlow); near_c = near_goal<KIND>(P, zlow);
  }
  TB_DIAG_ABLATE_STATICS(near_g); TB_DIAG_ABLATE_STATICS(near_n); TB_DIAG_ABLATE_STATICS(near_c);
  TB_LANES(6, near_g | near_n | near_c);  // [6] lanes near a static shape, [7] wave-substeps with one
  if (__any(near_g | near_n | near_c)) {
    if (near_g) hg = sphere_vs_box(P, P.ground_half[0], P.ground_half[1], P.ground_half[2], b.p);
    if (near_n) hn = sphere_vs_box(P, P.net_half[0], P.net_half[1], P.net_half[2], b.p);
    if (near_c) hc = sphere_vs_goal(P, goal_x, goal_y, b.p);
  }
  TB_STAMP(st, 2);  // static narrowphase
  if (hr.hit) bits |= CT_RACKET;
  if (hg.hit) bits |= CT_GROUND;
  if (hn.hit) bits |= CT_NET;
  if (hc.hit) bits |= CT_GOAL;
  if (nrg) bits |= CT_RACKET_COURT;

  TB_LANES(8, bits != 0);     // [8] lanes with a contact, [9] wave-substeps with one
  TB_LANES(10, hr.hit);       // [10] lanes with a racket contact, [11] wave-substeps with one
  // A ball on the court and nothing else -- what nearly every solve of a Tennisbot batch is (the ball bounces on its way to the
  // racket): when no lane of the wave touches anything BUT the ground, its one row is set up and solved in registers by a loop of
  // its own, without the row slots, the per-row votes and the racket's bookkeeping of the general solver below. The operations on
  // the env, and their order, are those of solve_contacts with row 1 alone: bit-identical. Tennisbot only (4096 envs, same box:
  // 786-788 -> 810 M env steps/s, with inv_sqrt_unit 813-814): in SwingRacket's kernels a ground contact is the one substep that
  // ends an episode, and the second copy of the row code costs them more than it saves (32768 envs 6.01 -> 5.91 G, PPO collect under
  // the trained policy 581 -> 569 M; the pipelined step kernel 1125 -> 1105 M).
  bool solved = false;
  constexpr bool SOLO = !RG && KIND == TB_ENV_TENNIS;
  if constexpr (SOLO) {
    if (__any(bits != 0) && !__any((bits & ~CT_GROUND) != 0)) {
      solved = true;
      TB_DIAG_ADD_LEADER(6, 1);
      if (bits) {
        const KParams& PC = COLD ? *reinterpret_cast<const KParams*>(hull + TB_HULL_KP) : P;
        RowS c;
        setup_static(PC, c, hg, PC.rest_court, PC.fric_court, b);
        float jref = 0.0f;
        for (int it = 0; it < PC.solver_iters; ++it) {
          bool moved = normal_static(PC, c, b, jref);
          moved |= friction_static(PC, c, b, jref);
          if (!moved) break;
        }
      }
    }
  }
  if (!solved && __any(bits != 0)) {
    TB_DIAG_ADD_LEADER(6, 1);  // wave-substeps that enter the solver
    if (bits) {  // only lanes that touch something enter the solver
      const KParams& PC = COLD ? *reinterpret_cast<const KParams*>(hull + TB_HULL_KP) : P;
      Rows<RG> R;
      Sym3 W;
      W.xx = 0.0f; W.xy = 0.0f; W.xz = 0.0f; W.yy = 0.0f; W.yz = 0.0f; W.zz = 0.0f;
      if constexpr (RG) {
        if (nrg) {
          W = world_inv_inertia(PC, rk.q, 1.0f / (scale * scale));
#pragma unroll 1
          for (int j = 0; j < nrg; ++j) setup_ground_row(PC, M, j, W, rk);
        }
      }
      R.on = bits & (CT_RACKET | CT_GROUND | CT_NET | CT_GOAL);  // CT_* = 1 << row
      if ((R.on & 1)) setup_racket<KIND == TB_ENV_TENNIS>(PC, R.rk, hr, rk, b, scale);
      if constexpr (REGROWS) {
        if ((R.on & 2)) setup_static(PC, R.st[0], hg, PC.rest_court, PC.fric_court, b);
        if ((R.on & 4)) setup_static(PC, R.st[1], hn, PC.rest_court, PC.fric_court, b);
        if ((R.on & 8)) setup_static(PC, R.st[2], hc, PC.rest_goal, PC.fric_goal, b);
      } else {  // the rows go to the lane's LDS column, one at a time through registers
        RowS c;
        if ((R.on & 2)) { setup_static(PC, c, hg, PC.rest_court, PC.fric_court, b); store_row<TWO>(M, 0, c); }
        if ((R.on & 4)) { setup_static(PC, c, hn, PC.rest_court, PC.fric_court, b); store_row<TWO>(M, 1, c); }
        if ((R.on & 8)) { setup_static(PC, c, hc, PC.rest_goal, PC.fric_goal, b); store_row<TWO>(M, 2, c); }
      }
      if constexpr (RG) {
        if ((R.on & 1)) setup_roll_racket<KIND == TB_ENV_TENNIS>(PC, R.qrk, R.rk, rk, scale);
        if ((R.on & 2)) setup_roll_static(PC, R.qst[0], PC.roll_court);
        if ((R.on & 4)) setup_roll_static(PC, R.qst[1], PC.roll_court);
        if ((R.on & 8)) setup_roll_static(PC, R.qst[2], PC.roll_goal);
      }
      solve_contacts<RG, REGROWS, TWO, RG && REGGROUND>(PC, R, M, nrg, W, rk, b);
    }
  }
  TB_STAMP(st, 4);  // contact solve
  integrate_pose(P, rk, b);
  TB_STAMP(st, 5);  // pose update
  return bits;
}

}  // namespace tb

// tb_stepper.hip -- HIP kernels (gfx950 / MI355X) and the C ABI of include/tb_stepper.h.
//
// Data layout in HBM (DESIGN.md "Layout"): persistent state is structure-of-arrays along
// the env index, 32-bit words [W][N] (W = 30 Swing / 28 Tennisbot) plus one done byte [N],
// so lane i of a wave reads word k at base + (k*N + i)*4: every row access is one fully
// coalesced 256-B wave transaction. Per-call I/O keeps the caller's natural row-major
// shapes (actions [N][A], obs [N][O]); a lane's 8/24/48-byte row is read/written with
// 8- or 16-byte vector accesses and the rows of a wave are contiguous.
//
// Kernel shape: one lane = one world, state held in registers for the whole call
// (including the <= 775-substep SwingRacket fast-forward, swingracket_env.py:105-141, and
// the T steps of tb_rollout). No inter-lane communication except wave-level counter
// reductions; no inter-workgroup communication at all, so blockIdx -> XCD placement does
// not matter for correctness or reuse (there is no shared tile to keep in one L2).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>

#include "../../include/tb_stepper.h"
#include "tb_device.hpp"

using namespace tb;

namespace {

// ------------------------------------------------------------------------------------------
// kernel arguments
#define TB_COUNTER_SHARDS 64

struct KArgs {
  KParams P;
  uint32_t* words;        // [W][n]
  uint8_t* done_state;    // [n]
  const float4* hull;     // [2 * n_hull] edge records (device global), staged into LDS
  const float* actions;   // [T][n][A]
  float* obs;             // [T][n][O]
  float* reward;          // [T][n]
  uint8_t* done_out;      // [T][n]
  float* term_obs;        // [n][O] or null (T == 1 only)
  int32_t* substeps;      // [n] or null
  const uint8_t* mask;    // reset kernel: [n] or null
  unsigned long long* counters;  // [TB_COUNTER_SHARDS][TB_N_COUNTERS]
  uint32_t* mani;         // [TB_MANI_WORDS][n] racket<->court contact caches (Manifold), valid where mflag is set
  uint8_t* mflag;         // [n] 1 = env i has cached contact points
  unsigned long long seed, env_id_base;
  int n, T;
  // pipelined fast-forward (tb_set_pipeline): a step that starts a SwingRacket fast-forward parks
  // the env's pre-loop state in a slot and resets the env; tb_ff_kernel finishes it on a side stream
  float4* ff_rec;         // [n][ff_rec<RG>()] slot: one record per env (park_env)
  uint8_t* ff_flag;       // [n] 1 = env i is parked in the slot (null for the compacted / sorted lists: their records' own tag says so)
  int ff_lanes;           // tb_ff_kernel: parked envs per wave (a few per wave at small batch sizes)
  float4* ff_next;        // tb_ff_kernel: where envs still running when their budget is spent are compacted to (null = last phase: no budget)
  int* ff_next_count;     // ... and how many there are so far
  const int* ff_src_count;  // tb_ff_kernel, phases 2+: how many records ff_rec holds (null = A.n slots, parked or not)
  // deferred stragglers (tb_ff_kernel<.., POOL>): ff_next is then the handle's POOL, shared by every episode until the next flush
  int ff_cap;             // its capacity in records (a lane whose place does not fit finishes its loop in this launch)
  int ff_extra;           // substeps granted beyond the ballistic estimate before an env is deferred
  float** pool_dst_out;   // [ff_cap] where the deferred env's terminal reward goes: written next to its record
  float* const* pool_dst_in;  // the pool kernel reads it back (null: A.reward + env index)
  int defer;
  // fused policy inference (tb_policy_step): actions are computed in-kernel from pol_obs
  const float* pol_weights;  // packed SB3 MlpPolicy towers, see PolicyNet
  const float* pol_obs;      // [n][O] the observation each env acts on
  float* pol_actions;        // [n][A] what the env is stepped with (clipped to [-1, 1])
  float* pol_raw;            // [n][A] the unclipped sample (what the log-probability is of)
  float* pol_logp;           // [n]
  float* pol_value;          // [n]
  unsigned long long pol_seed;
  int pol_deterministic;
  // per-step strides (in elements) of the output arrays of a policy rollout launch (tb_policy_rollout)
  size_t st_obs, st_rew, st_done, st_act, st_raw, st_logp, st_val;
};

struct EnvRegs {
  Racket r;
  Ball b;
  float aux[6];  // swing: goal.x goal.y spawn.x spawn.y spawn.z d0 ; tennis: shoot force xyz, racket scale
  int step_count;
  uint32_t episode;
  uint32_t done;
};

template <int KIND> struct Dims;
template <> struct Dims<TB_ENV_SWING> { static constexpr int W = TB_SWING_WORDS, A = TB_SWING_ACT_DIM, O = TB_SWING_OBS_DIM, NAUX = 6; };
template <> struct Dims<TB_ENV_TENNIS> { static constexpr int W = TB_TENNIS_WORDS, A = TB_TENNIS_ACT_DIM, O = TB_TENNIS_OBS_DIM, NAUX = 4; };

// SwingRacket's state rows are addressed as ONE 64-bit base (scalar registers) + a 32-bit byte offset per row and lane: the
// global_load / store form with a scalar base, one v_add_u32 per row -- instead of a 64-bit pointer bump per row (a 64-bit vector add
// and two scalar adds: 108 scalar instructions per step launch for the 54 rows a step reads and writes, which a lone wave per SIMD
// pays for one by one). 4096 envs 1005 -> 1033 M env steps/s, 1 M 11.3 -> 11.8 G; Tennisbot gains nothing at 4096 envs and loses 2.6 %
// at 1 M: it keeps the 64-bit form. tb_create bounds n_envs so that the largest offset (30 rows x n x 4 bytes) fits 32 bits.
template <bool OFF32>
TB_DEV uint32_t row_word(const uint32_t* w, int row, int n, int i) {
  if constexpr (OFF32) {
    const uint32_t byte = ((uint32_t)row * (uint32_t)n + (uint32_t)i) * 4u;
    return *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(w) + byte);
  } else return w[(size_t)row * n + i];
}
template <bool OFF32>
TB_DEV void row_store(uint32_t* w, int row, int n, int i, uint32_t v) {
  if constexpr (OFF32) {
    const uint32_t byte = ((uint32_t)row * (uint32_t)n + (uint32_t)i) * 4u;
    *reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(w) + byte) = v;
  } else w[(size_t)row * n + i] = v;
}
template <int KIND> constexpr bool off32() { return KIND == TB_ENV_SWING; }
template <int KIND> TB_DEV float ld(const uint32_t* w, int row, int n, int i) { return __uint_as_float(row_word<off32<KIND>()>(w, row, n, i)); }
template <int KIND> TB_DEV void st(uint32_t* w, int row, int n, int i, float v) { row_store<off32<KIND>()>(w, row, n, i, __float_as_uint(v)); }

template <int KIND>
TB_DEV void load_env(const uint32_t* w, const uint8_t* done_state, int n, int i, EnvRegs& e) {
  e.r.p = mk(ld<KIND>(w, TB_W_RP, n, i), ld<KIND>(w, TB_W_RP + 1, n, i), ld<KIND>(w, TB_W_RP + 2, n, i));
  e.r.q.x = ld<KIND>(w, TB_W_RQ, n, i); e.r.q.y = ld<KIND>(w, TB_W_RQ + 1, n, i); e.r.q.z = ld<KIND>(w, TB_W_RQ + 2, n, i); e.r.q.w = ld<KIND>(w, TB_W_RQ + 3, n, i);
  e.r.v = mk(ld<KIND>(w, TB_W_RV, n, i), ld<KIND>(w, TB_W_RV + 1, n, i), ld<KIND>(w, TB_W_RV + 2, n, i));
  e.r.w = mk(ld<KIND>(w, TB_W_RW, n, i), ld<KIND>(w, TB_W_RW + 1, n, i), ld<KIND>(w, TB_W_RW + 2, n, i));
  e.b.p = mk(ld<KIND>(w, TB_W_BP, n, i), ld<KIND>(w, TB_W_BP + 1, n, i), ld<KIND>(w, TB_W_BP + 2, n, i));
  e.b.v = mk(ld<KIND>(w, TB_W_BV, n, i), ld<KIND>(w, TB_W_BV + 1, n, i), ld<KIND>(w, TB_W_BV + 2, n, i));
  e.b.w = mk(ld<KIND>(w, TB_W_BW, n, i), ld<KIND>(w, TB_W_BW + 1, n, i), ld<KIND>(w, TB_W_BW + 2, n, i));
#pragma unroll
  for (int k = 0; k < 6; ++k) e.aux[k] = k < Dims<KIND>::NAUX ? ld<KIND>(w, 22 + k, n, i) : 0.0f;
  e.step_count = (int)row_word<off32<KIND>()>(w, Dims<KIND>::W - 2, n, i);
  e.episode = row_word<off32<KIND>()>(w, Dims<KIND>::W - 1, n, i);
  e.done = done_state[i];
}

// `all`: also the rows that only change on reset (goal / spawn / d0 / shoot force / episode)
template <int KIND>
TB_DEV void store_env(uint32_t* w, uint8_t* done_state, int n, int i, const EnvRegs& e, bool all) {
  st<KIND>(w, TB_W_RP, n, i, e.r.p.x); st<KIND>(w, TB_W_RP + 1, n, i, e.r.p.y); st<KIND>(w, TB_W_RP + 2, n, i, e.r.p.z);
  st<KIND>(w, TB_W_RQ, n, i, e.r.q.x); st<KIND>(w, TB_W_RQ + 1, n, i, e.r.q.y); st<KIND>(w, TB_W_RQ + 2, n, i, e.r.q.z); st<KIND>(w, TB_W_RQ + 3, n, i, e.r.q.w);
  st<KIND>(w, TB_W_RV, n, i, e.r.v.x); st<KIND>(w, TB_W_RV + 1, n, i, e.r.v.y); st<KIND>(w, TB_W_RV + 2, n, i, e.r.v.z);
  st<KIND>(w, TB_W_RW, n, i, e.r.w.x); st<KIND>(w, TB_W_RW + 1, n, i, e.r.w.y); st<KIND>(w, TB_W_RW + 2, n, i, e.r.w.z);
  st<KIND>(w, TB_W_BP, n, i, e.b.p.x); st<KIND>(w, TB_W_BP + 1, n, i, e.b.p.y); st<KIND>(w, TB_W_BP + 2, n, i, e.b.p.z);
  st<KIND>(w, TB_W_BV, n, i, e.b.v.x); st<KIND>(w, TB_W_BV + 1, n, i, e.b.v.y); st<KIND>(w, TB_W_BV + 2, n, i, e.b.v.z);
  st<KIND>(w, TB_W_BW, n, i, e.b.w.x); st<KIND>(w, TB_W_BW + 1, n, i, e.b.w.y); st<KIND>(w, TB_W_BW + 2, n, i, e.b.w.z);
  if (all) {
#pragma unroll
    for (int k = 0; k < Dims<KIND>::NAUX; ++k) st<KIND>(w, 22 + k, n, i, e.aux[k]);
    row_store<off32<KIND>()>(w, Dims<KIND>::W - 1, n, i, e.episode);
  }
  row_store<off32<KIND>()>(w, Dims<KIND>::W - 2, n, i, (uint32_t)e.step_count);
  done_state[i] = (uint8_t)e.done;
}

// The racket<->court contact cache between launches: handle-owned rows next to the state (not part of the state words: a
// restored state starts with an empty cache, like the oracle's). One flag byte per env is read by every launch; the 14 words
// behind it only by lanes that have cached points -- a racket on the ground.
#define TB_MANI_WORDS 14
// dynamic LDS of every kernel that steps envs, one column per lane: [TB_ROWS_LDS words: the static contact rows, unless the
// instantiation keeps them in registers (REGROWS)] [TB_MANI_LDS words: the racket<->court cache, RG instantiations only]
extern __shared__ float s_mani[];
TB_DEV void init_manifold(Manifold& M, int lane_in_block, int lanes, bool rows_in_lds) {
  M.n = 0; M.deep = 0; M.stride = lanes;
  M.st = s_mani + lane_in_block;
  M.m = s_mani + (rows_in_lds ? TB_ROWS_LDS * lanes : 0) + lane_in_block;
}
TB_DEV void load_manifold(const KArgs& A, int i, Manifold& M) {
  const uint32_t w0 = A.mani[i], w1 = A.mani[(size_t)A.n + i];
  M.n = (int)(w0 & 255u); M.deep = (int)(w0 >> 8);
#pragma unroll 1
  for (int j = 0; j < TB_MAX_RG; ++j) {
    mw(M, j, MW_ID) = __int_as_float((int)((w1 >> (8 * j)) & 255u));
    mw(M, j, MW_JN) = __uint_as_float(A.mani[(size_t)(2 + j) * A.n + i]);
    mw(M, j, MW_JT1) = __uint_as_float(A.mani[(size_t)(6 + j) * A.n + i]);
    mw(M, j, MW_JT2) = __uint_as_float(A.mani[(size_t)(10 + j) * A.n + i]);
  }
}
TB_DEV void store_manifold(const KArgs& A, int i, const Manifold& M, bool had) {
  if (M.n > 0) {
    uint32_t w1 = 0u;
#pragma unroll 1
    for (int j = 0; j < TB_MAX_RG; ++j) {
      const bool on = j < M.n;
      w1 |= (uint32_t)(on ? __float_as_int(mw(M, j, MW_ID)) : 0) << (8 * j);
      A.mani[(size_t)(2 + j) * A.n + i] = on ? __float_as_uint(mw(M, j, MW_JN)) : 0u;
      A.mani[(size_t)(6 + j) * A.n + i] = on ? __float_as_uint(mw(M, j, MW_JT1)) : 0u;
      A.mani[(size_t)(10 + j) * A.n + i] = on ? __float_as_uint(mw(M, j, MW_JT2)) : 0u;
    }
    A.mani[i] = (uint32_t)M.n | ((uint32_t)M.deep << 8);
    A.mani[(size_t)A.n + i] = w1;
    A.mflag[i] = 1;
  } else if (had) {
    A.mflag[i] = 0;
  }
}

template <int KIND>
TB_DEV void make_obs(const EnvRegs& e, float* o) {
  if (KIND == TB_ENV_SWING) {  // swingracket_env.py:143-144,184-185
    o[0] = e.r.p.x; o[1] = e.r.p.y; o[2] = e.b.p.x; o[3] = e.b.p.y; o[4] = e.aux[0]; o[5] = e.aux[1];
  } else {  // tennisbot_env.py:134-136,259-261
    o[0] = e.r.p.x; o[1] = e.r.p.y; o[2] = e.r.p.z; o[3] = e.r.v.x; o[4] = e.r.v.y; o[5] = e.r.v.z;
    o[6] = e.b.p.x; o[7] = e.b.p.y; o[8] = e.b.p.z; o[9] = e.b.v.x; o[10] = e.b.v.y; o[11] = e.b.v.z;
  }
}
template <int KIND>
TB_DEV void write_obs(float* dst, size_t row, const float* o) {
  if (KIND == TB_ENV_SWING) {  // 24-byte rows: three 8-byte stores
    float2* p = reinterpret_cast<float2*>(dst + row * 6);
    p[0] = make_float2(o[0], o[1]); p[1] = make_float2(o[2], o[3]); p[2] = make_float2(o[4], o[5]);
  } else {  // 48-byte rows: three 16-byte stores
    float4* p = reinterpret_cast<float4*>(dst + row * 12);
    p[0] = make_float4(o[0], o[1], o[2], o[3]); p[1] = make_float4(o[4], o[5], o[6], o[7]); p[2] = make_float4(o[8], o[9], o[10], o[11]);
  }
}

template <int KIND>
TB_DEV void load_actions(const float* actions, size_t row, float* a) {
  if (KIND == TB_ENV_SWING) {  // 24-byte rows: three 8-byte loads
    const float2* ap = reinterpret_cast<const float2*>(actions + row * 6);
    float2 a0 = ap[0], a1 = ap[1], a2 = ap[2];
    a[0] = a0.x; a[1] = a0.y; a[2] = a1.x; a[3] = a1.y; a[4] = a2.x; a[5] = a2.y;
  } else {
    float2 a0 = *reinterpret_cast<const float2*>(actions + row * 2);
    a[0] = a0.x; a[1] = a0.y;
  }
}

// reset(): swingracket_env.py:151-186 / tennisbot_env.py:217-261. The world rebuild
// (resetSimulation + 3-4 loadURDF + STL hull + texture) collapses to re-drawing the state.
// `KP`: the device-resident copy of the parameter block (LDS in the step kernels, global memory in the reset
// kernel). The racket scale is read from THERE, not from the kernel arguments: tb_set_racket_scale updates it
// with a stream-ordered store, so replays of a hipGraph captured earlier see the curriculum (train.py:164-176).
template <int KIND>
TB_DEV void reset_env(const KArgs& A, const float4* KP, int i, EnvRegs& e) {
  const KParams& P = A.P;
  unsigned long long id = A.env_id_base + (unsigned long long)i;
  uint32_t k0 = (uint32_t)A.seed, k1 = (uint32_t)(A.seed >> 32);
  uint32_t c0 = (uint32_t)id, c1 = (uint32_t)(id >> 32);
  uint32_t u[4], w[4];
  TB_DIAG_PHILOX(c0, c1, e.episode, 0u, k0, k1, u);
  vec3 zero = mk(0.0f, 0.0f, 0.0f);
  e.r.v = zero; e.r.w = zero; e.b.v = zero; e.b.w = zero;
  vec3 com = mk(P.racket_com[0], P.racket_com[1], P.racket_com[2]);
  uint32_t spin_block;
  if (KIND == TB_ENV_SWING) {
    // swingracket_env.py:161-170: link x~U(5.5,11), y~U(-4,4), z=.6, rpy=(0,.5,0); ball (x-.1, y, z+.8)
    float x = uniform(5.5f, 5.5f, u[0]), y = uniform(-4.0f, 8.0f, u[1]), z = 0.6f;
    quat q0; q0.x = 0.0f; q0.y = (float)0.24740395925452292; q0.z = 0.0f; q0.w = (float)0.96891242171064473;
    e.r.q = q0;
    e.r.p = mk(x, y, z) + rotate(q0, com);  // racket.py:131 reports the COM frame
    e.b.p = mk(x - 0.1f, y, z + 0.8f);
    float gx = uniform(-3.0f, -9.0f, u[2]), gy = uniform(-5.0f, 10.0f, u[3]);  // :173
    e.aux[0] = gx; e.aux[1] = gy; e.aux[2] = x; e.aux[3] = y; e.aux[4] = z;
    float dx = e.b.p.x - gx, dy = e.b.p.y - gy;
    e.aux[5] = sqrtf(FMA(dx, dx, dy * dy));  // :174-175
    spin_block = 1u;
  } else {
    // tennisbot_env.py:227-246; objects.py:82-96 (ball born at (-9,0,1))
    TB_DIAG_PHILOX(c0, c1, e.episode, 1u, k0, k1, w);
    float x = uniform(7.5f, 5.0f, u[0]), y = uniform(-5.0f, 10.0f, u[1]), z = uniform(0.2f, 0.21f - 0.2f, u[2]);
    quat q0; q0.x = 0.0f; q0.y = 0.0f; q0.z = 0.0f; q0.w = 1.0f;
    e.r.q = q0;
    e.aux[3] = reinterpret_cast<const KParams*>(KP)->racket_scale;  // Racket(..., scale=self.racket_scale), tennisbot_env.py:230-234
    e.r.p = mk(x, y, z) + e.aux[3] * com;
    e.aux[0] = uniform(25.0f, 12.5f, u[3]);
    e.aux[1] = uniform(-10.0f, 20.0f, w[0]);
    e.aux[2] = 20.0f;
    e.aux[4] = 0.0f; e.aux[5] = 0.0f;
    e.b.p = mk(uniform(-12.0f, 6.0f, w[1]), uniform(-1.0f, 2.0f, w[2]), uniform(1.0f, 0.5f, w[3]));
    spin_block = 2u;
  }
  if (P.ball_spin_max != 0.0f) {  // extension; 0 reproduces the reference
    TB_DIAG_PHILOX(c0, c1, e.episode, spin_block, k0, k1, w);
    float m = P.ball_spin_max;
    e.b.w = mk(uniform(-m, 2.0f * m, w[0]), uniform(-m, 2.0f * m, w[1]), uniform(-m, 2.0f * m, w[2]));
  }
  e.step_count = 0;
  e.done = TB_DONE_NO;
}

// Parked SwingRacket envs travel as ONE record each (array of structures, unlike the SoA state): the fast-forward kernel
// hands records to lanes in another order than the env index (compacted survivors, sorted by predicted flight length, or a
// few per wave), and a lane that fetches whole 128-byte lines wastes nothing, where a gather from the SoA rows would pull a
// 32-byte sector per word. The record is 8 float4 = 128 B = one line: the env's state. The RG instantiations (racket<->court
// contact compiled in) append the contact cache: 12 float4 = 192 B. (Until round 3 every record was 192 B: a third of the
// fast-forward's record traffic was a cache that the default kernels never look at.)
#define TB_FF_REC_MAX 12  // what the handle allocates per env and slot (the parameter block may switch the extended contacts on later)
template <bool RG> constexpr int ff_rec() { return RG ? TB_FF_REC_MAX : 8; }
template <bool RG>
TB_DEV void park_env(float4* rec, int i, const EnvRegs& e, const Manifold& M) {
  float4* r = rec + (size_t)i * ff_rec<RG>();
  r[0] = make_float4(e.r.p.x, e.r.p.y, e.r.p.z, e.r.q.x);
  r[1] = make_float4(e.r.q.y, e.r.q.z, e.r.q.w, e.r.v.x);
  r[2] = make_float4(e.r.v.y, e.r.v.z, e.r.w.x, e.r.w.y);
  r[3] = make_float4(e.r.w.z, e.b.p.x, e.b.p.y, e.b.p.z);
  r[4] = make_float4(e.b.v.x, e.b.v.y, e.b.v.z, e.b.w.x);
  r[5] = make_float4(e.b.w.y, e.b.w.z, e.aux[0], e.aux[1]);
  r[6] = make_float4(e.aux[2], e.aux[3], e.aux[4], e.aux[5]);
  r[7] = make_float4(__int_as_float(e.step_count), __uint_as_float(e.episode), __uint_as_float(1u), __int_as_float(i));
  if constexpr (!RG) return;
  // (statically indexed: registers; lanes without cached points -- nearly all -- skip the LDS reads)
  uint32_t ids = 0u;
  float imp[3 * TB_MAX_RG];
#pragma unroll
  for (int j = 0; j < 3 * TB_MAX_RG; ++j) imp[j] = 0.0f;
  if (M.n > 0) {
#pragma unroll
    for (int j = 0; j < TB_MAX_RG; ++j) {
      const bool on = j < M.n;
      ids |= (uint32_t)(on ? __float_as_int(mw(M, j, MW_ID)) : 0) << (8 * j);
      imp[j] = on ? mw(M, j, MW_JN) : 0.0f; imp[TB_MAX_RG + j] = on ? mw(M, j, MW_JT1) : 0.0f; imp[2 * TB_MAX_RG + j] = on ? mw(M, j, MW_JT2) : 0.0f;
    }
  }
  r[8] = make_float4(__uint_as_float((uint32_t)M.n | ((uint32_t)M.deep << 8)), __uint_as_float(ids), imp[0], imp[1]);
  r[9] = make_float4(imp[2], imp[3], imp[4], imp[5]);
  r[10] = make_float4(imp[6], imp[7], imp[8], imp[9]);
  r[11] = make_float4(imp[10], imp[11], 0.0f, 0.0f);
}
template <bool RG>
TB_DEV void unpark_env(const float4* r, EnvRegs& e, Manifold& M, int& env_index) {
  if constexpr (RG) {
    const uint32_t w0 = __float_as_uint(r[8].x), ids = __float_as_uint(r[8].y);
    const float imp[3 * TB_MAX_RG] = {r[8].z, r[8].w, r[9].x, r[9].y, r[9].z, r[9].w, r[10].x, r[10].y, r[10].z, r[10].w, r[11].x, r[11].y};
    M.n = (int)(w0 & 255u); M.deep = (int)(w0 >> 8);
    if (M.n > 0) {
#pragma unroll
      for (int j = 0; j < TB_MAX_RG; ++j) {
        mw(M, j, MW_ID) = __int_as_float((int)((ids >> (8 * j)) & 255u));
        mw(M, j, MW_JN) = imp[j]; mw(M, j, MW_JT1) = imp[TB_MAX_RG + j]; mw(M, j, MW_JT2) = imp[2 * TB_MAX_RG + j];
      }
    }
  }
  e.r.p = mk(r[0].x, r[0].y, r[0].z);
  e.r.q.x = r[0].w; e.r.q.y = r[1].x; e.r.q.z = r[1].y; e.r.q.w = r[1].z;
  e.r.v = mk(r[1].w, r[2].x, r[2].y);
  e.r.w = mk(r[2].z, r[2].w, r[3].x);
  e.b.p = mk(r[3].y, r[3].z, r[3].w);
  e.b.v = mk(r[4].x, r[4].y, r[4].z);
  e.b.w = mk(r[4].w, r[5].x, r[5].y);
  e.aux[0] = r[5].z; e.aux[1] = r[5].w; e.aux[2] = r[6].x; e.aux[3] = r[6].y; e.aux[4] = r[6].z; e.aux[5] = r[6].w;
  e.step_count = __float_as_int(r[7].x); e.episode = __float_as_uint(r[7].y);
  env_index = __float_as_int(r[7].w);
  e.done = TB_DONE_NO;  // a parked env was running
}
// How long will this parked env's fast-forward last? The ball's flight decides (the loop ends when it touches the court
// or the goal): vertical motion under gravity and Bullet's v (k1 + k2 |v|) drag, integrated with 4 substeps per
// iteration until the ball's lowest point reaches the court; the iteration count is the sort key. An ESTIMATE for
// scheduling only -- which lane computes which env never changes a result -- so the hardware's approximate square
// root is good enough, and a ball that is struck again, rolls onto the goal or the net first just lands in a
// neighbouring bin.
TB_DEV int predict_flight(const KParams& P, vec3 bp, vec3 bv) {
  const float dt4 = 4.0f * P.dt, z_land = (P.ground_half[2] + P.ball_radius) + P.contact_threshold;
  float z = bp.z, vz = bv.z, vh = __builtin_amdgcn_sqrtf(FMA(bv.x, bv.x, bv.y * bv.y));
  int k = 0;
  while (k < 200 && z > z_land) {
    float kd = FMA(P.lin_damp_quad, __builtin_amdgcn_sqrtf(FMA(vh, vh, vz * vz)), P.lin_damp);
    vz = FMA(dt4, -P.gravity - vz * kd, vz);
    vh = FMA(dt4, -(vh * kd), vh);
    z = FMA(dt4, vz, z);
    ++k;
  }
  return k;
}

// swingracket_env.py:63-73
TB_DEV float moved_dist_to_goal(const EnvRegs& e) {
  float dx = e.b.p.x - e.aux[0], dy = e.b.p.y - e.aux[1];
  float d = sqrtf(FMA(dx, dx, dy * dy));
  return ((e.aux[5] - d) / e.aux[5]) * 20.0f;
}
TB_DEV vec3 restoring_force(const EnvRegs& e) {  // swingracket_env.py:135-141
  return mk(-50.0f * (e.r.p.x - e.aux[2]), -2.0f * (e.r.p.y - e.aux[3]), -2.0f * ((e.r.p.z - e.aux[4]) - 4.0f));
}

// swingracket_env.py:75-145 as ONE loop around ONE substep call site (the substep is the bulk of
// the kernel's code and registers; two inlined copies cost occupancy):
//   iteration 0      the agent's substep (:76-83), contact bonus while step_count < 25 (:98-101)
//   iterations 1..   the fast-forward of :105-141 -- substeps until the ball touches the court or
//                    the goal or step_count > 800; no agent input enters it. The first of them runs
//                    with no force at all (the previous substep cleared the accumulators), the
//                    later ones with the restoring force of :135-141.
// FORM   = the substep's form, a mask of SF_* bits (tb_device.hpp). The ones that show here:
// `in_ff` = start inside the fast-forward (tb_ff_kernel resuming a parked env).
// `defer`  = leave the fast-forward to tb_ff_kernel: sets `parked` instead of looping.
// BUDGET  = tb_ff_kernel only: leave the loop after `budget` substeps with the env still running (`parked` again): the
//           next phase kernel resumes it from the saved state, with the restoring force recomputed from that state.
// SF_ESC  = first phase of the large-batch tb_ff_kernel: a lane that needs the racket's exact narrowphase leaves the loop BEFORE
//           that substep (`parked` again, nothing of the substep applied); see substep.
// SF_LAZYTAB = see substep (the pipelined step kernel: `hull` is filled from `table_mem` by the first wave that reads it)
template <unsigned FORM, bool BUDGET = false>
TB_DEV float swing_loop(const KParams& P, const float4* hull, EnvRegs& e, Manifold& M, vec3 F, vec3 T, bool in_ff, bool defer, bool& parked,
                        int& ns, uint32_t* cnt TB_STAMP_ARG, int budget = 0, const float4* table_mem = nullptr) {
  const vec3 zero = mk(0.0f, 0.0f, 0.0f);
  float reward = 0.0f;
  for (;;) {
    constexpr bool ESC = (FORM & SF_ESC) != 0;
    int bits = substep<TB_ENV_SWING, FORM>(P, hull, e.r, e.b, M, F, T, zero, e.aux[0], e.aux[1], 1.0f TB_STAMP_PASS, table_mem);  // :82 / :107
    if (ESC && (bits & CT_ESCAPE)) { parked = true; break; }
    e.step_count += 1; ns++;                                                                                   // :83 / :108
    if (bits & CT_RACKET) cnt[0]++;
    if (!in_ff) {
      if (e.step_count < 25 && (bits & CT_RACKET)) reward += 2.0f;  // :98-101
      if (!(e.step_count > 25) || e.done) break;                    // :105-106
      if (defer) { parked = true; break; }                          // reward so far is 0: the bonus window closed at step 25
      in_ff = true; F = zero; T = zero;                             // accumulators were cleared by the substep above
    } else {
      if (bits & (CT_GROUND | CT_NET)) { e.done = TB_DONE_PENDING_FORCE; reward += moved_dist_to_goal(e); cnt[1]++; }  // :111-114
      if (bits & CT_GOAL) { reward += moved_dist_to_goal(e); reward += 50.0f; e.done = TB_DONE_PENDING_FORCE; cnt[2]++; }  // :119-123
      if (e.step_count > 800) { if (!e.done) cnt[3]++; e.done = TB_DONE_PENDING_FORCE; }  // :127-128
      if (e.done) break;
      F = restoring_force(e);  // :135-141 (also issued when done just became true; it then waits in the accumulator: TB_DONE_PENDING_FORCE)
      if (BUDGET && --budget <= 0) { parked = true; break; }
    }
  }
  return reward;
}

template <unsigned FORM>
TB_DEV float swing_step(const KParams& P, const float4* hull, EnvRegs& e, Manifold& M, const float* a, int& ns, uint32_t* cnt, bool defer, bool& parked TB_STAMP_ARG,
                        const float4* table_mem = nullptr) {
  vec3 F = mk(a[0] * 400.0f, a[1] * 400.0f, FMA(a[2], 400.0f, 4.0f * 9.81f));  // :76-77
  vec3 T = mk(a[3] * 5.0f, a[4] * 5.0f, a[5] * 5.0f);                          // :78
  if (e.done == TB_DONE_PENDING_FORCE) {  // the force of :135-141 is still in the accumulator
    F = F + restoring_force(e);
    e.done = TB_DONE_YES;
  }
  ns = 0;
  const float rew = swing_loop<FORM>(P, hull, e, M, F, T, false, defer, parked, ns, cnt TB_STAMP_PASS, 0, table_mem);
  if (!parked && M.n == 0) M.deep = 0;  // an empty cache is not kept between env.step() calls (a parked env's call is not over: its record keeps it)
  return rew;
}

// tennisbot_env.py:90-102
TB_DEV float dist_to_reward(float d) {
  return d < 0.5f ? 20.0f : d < 1.0f ? 15.0f : d < 2.0f ? 10.0f : d < 3.0f ? 5.0f : d < 4.0f ? 1.0f : 0.0f;
}

// tennisbot_env.py:104-207 (the DELAY_MODE sleep at :124-126 is dropped on purpose)
template <unsigned FORM>
TB_DEV float tennis_step(const KParams& P, const float4* hull, EnvRegs& e, Manifold& M, const float* a, float* obs, bool& ret_done, uint32_t* cnt TB_STAMP_ARG) {
  const vec3 zero = mk(0.0f, 0.0f, 0.0f);
  vec3 F = mk(a[0] * 10.0f, a[1] * 10.0f, 4.0f * 9.81f);  // :112-115
  vec3 Fb = zero;
  if (e.step_count < 5) Fb = mk(e.aux[0], e.aux[1], e.aux[2]);  // :118-119
  int bits = substep<TB_ENV_TENNIS, FORM>(P, hull, e.r, e.b, M, F, zero, Fb, 0.0f, 0.0f, e.aux[3] TB_STAMP_PASS);  // :121
  if (M.n == 0) M.deep = 0;  // an empty cache is not kept between env.step() calls
  e.step_count += 1;                                                                // :122
  if (bits & CT_RACKET) cnt[0]++;
  make_obs<TB_ENV_TENNIS>(e, obs);  // :134-136
  float reward = 0.0f;
  ret_done = false;
  if (e.step_count < 5) return reward;  // :138-139 returns the literal False
  float dz = e.b.p.z - e.r.p.z, dy = e.b.p.y - e.r.p.y;
  float delta = sqrtf(FMA(dz, dz, dy * dy));  // :142-143
  if (bits & CT_RACKET) { reward += 25.0f; reward += dist_to_reward(delta); }  // :170-174
  if (!(e.b.p.x - e.r.p.x < 0.5f)) {  // :182-194
    if (!e.done) cnt[4]++;
    e.done = TB_DONE_YES;
    reward += dist_to_reward(delta);
  }
  // :197-198 `3 > x > 15` is never true
  if (e.step_count > 1000) { if (!e.done) cnt[3]++; e.done = TB_DONE_YES; }  // :201-203
  ret_done = e.done != TB_DONE_NO;
  return reward;
}

TB_DEV bool finite3(vec3 v) { return isfinite(v.x) && isfinite(v.y) && isfinite(v.z); }
// all 22 state values finite? x * 0 is (+-)0 for a finite x and NaN for an infinity or a NaN, and NaN survives every sum: four
// short fma chains and ONE comparison instead of 22 class tests and the scalar ands between them (the same verdict for every input)
TB_DEV bool state_is_finite(const EnvRegs& e) {
  float a0 = e.r.p.x * 0.0f, a1 = e.r.p.y * 0.0f, a2 = e.r.p.z * 0.0f, a3 = e.r.q.x * 0.0f;
  a0 = FMA(e.r.q.y, 0.0f, a0); a1 = FMA(e.r.q.z, 0.0f, a1); a2 = FMA(e.r.q.w, 0.0f, a2); a3 = FMA(e.r.v.x, 0.0f, a3);
  a0 = FMA(e.r.v.y, 0.0f, a0); a1 = FMA(e.r.v.z, 0.0f, a1); a2 = FMA(e.r.w.x, 0.0f, a2); a3 = FMA(e.r.w.y, 0.0f, a3);
  a0 = FMA(e.r.w.z, 0.0f, a0); a1 = FMA(e.b.p.x, 0.0f, a1); a2 = FMA(e.b.p.y, 0.0f, a2); a3 = FMA(e.b.p.z, 0.0f, a3);
  a0 = FMA(e.b.v.x, 0.0f, a0); a1 = FMA(e.b.v.y, 0.0f, a1); a2 = FMA(e.b.v.z, 0.0f, a2); a3 = FMA(e.b.w.x, 0.0f, a3);
  a0 = FMA(e.b.w.y, 0.0f, a0); a1 = FMA(e.b.w.z, 0.0f, a1);
  const float t = (a0 + a1) + (a2 + a3);
  return t == t;
}

// wave-level sum of per-lane event counts; one atomic per wave and counter that is non-zero, into
// one of TB_COUNTER_SHARDS copies (same-address atomics serialise at ~12 ns each: with one copy a
// 1 M-env launch, 16 K waves, spent 200 us queueing on the substep counter alone). The mandatory
// first substep of a step is not counted on the device at all: the host adds n_envs * T per launch (count_first_substeps).
TB_DEV void flush_counters(unsigned long long* counters, const uint32_t* cnt) {
  uint32_t any = 0u;
#pragma unroll
  for (int k = 0; k < TB_N_COUNTERS; ++k) any |= cnt[k];
  if (__ballot(any != 0u) == 0ull) return;  // nothing happened in this wave (most launches of a small batch: one test instead of nine)
  counters += (size_t)((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & (TB_COUNTER_SHARDS - 1)) * TB_N_COUNTERS;
#pragma unroll
  for (int k = 0; k < TB_N_COUNTERS; ++k) {
    uint32_t v = cnt[k];
    if (__ballot(v != 0) == 0ull) continue;  // wave-uniform skip: most counters are zero most steps
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(&counters[k], (unsigned long long)v);
  }
}

TB_DEV void stage_hull(float4* s_hull, const KArgs& A) {
  for (int k = threadIdx.x; k < 2 * A.P.n_hull; k += blockDim.x) s_hull[k] = A.hull[k];
  for (int k = TB_HULL_PLANES + threadIdx.x; k < TB_HULL_LDS; k += blockDim.x) s_hull[k] = A.hull[k];
  __syncthreads();
}
static_assert(sizeof(KParams) <= sizeof(float4) * TB_KP_ROWS, "the LDS copy of the parameter block needs more rows");

// ------------------------------------------------------------------------------------------
// step / rollout kernel: T agent steps of every env, state in registers throughout
// ------------------------------------------------------------------------------------------
}  // namespace
#include "tb_policy.hpp"
namespace {

// LEAN (SwingRacket only): every lane that would start a fast-forward is parked for tb_ff_kernel, so
// the loop is not compiled into this kernel at all -- the pipelined path's step kernel. Its code is
// a third of the full kernel's, which is worth ~1.5 us of a ~7 us launch at 4096 envs.
// MULTI: A.T agent steps in one launch (tb_rollout); otherwise exactly one (tb_step). A compile-time
// trip count of 1 is worth ~50-100 VGPRs (no loop-carried copies of the per-step bookkeeping), i.e.
// one to two more waves per SIMD for the kernel every RL step launches.
// POLICY: the actions are not read from memory but inferred in-kernel (tb_policy_step).
// REGROWS (Tennisbot, small batches): the static contact rows in registers, see solve_contacts.
// SCHEDULING HINTS. Three places below (and one in tb_device.hpp) steer where the compiler puts scalar argument loads, with empty
// `asm volatile` statements that only NAME values. They change no result; each was chosen by a same-box A/B on AMD clang 22 / ROCm 7.2
// (profiles/EXPERIMENTS.md) and is worth 1-4 % to a launch-bound kernel -- on THIS compiler. They are the only compile-time switches
// left in the product sources, kept so that tools/diag/r04_hint_recheck.py can re-measure every one of them (-DTB_HINT_x=0 against
// the default) after a toolchain change; a hint that no longer pays is deleted, not tuned. (Round 4's re-check deleted one: the
// pipelined SwingRacket step kernel's rare-branch arguments named early gave 4096 envs +1 % and cost 32768 envs 5 %.)
#ifndef TB_HINT_TENNIS_CONSTANTS
#define TB_HINT_TENNIS_CONSTANTS 1   // Tennisbot step kernel: the free-flight constants fetched beside the state loads
#endif
#ifndef TB_HINT_TENNIS_OUTPUTS
#define TB_HINT_TENNIS_OUTPUTS 1     // ... and its output pointers
#endif
#ifndef TB_HINT_POLICY_VGPR_PARAMS
#define TB_HINT_POLICY_VGPR_PARAMS 1 // policy rollout kernels: the substep's constants pinned in vector registers for the whole launch
#endif
template <int KIND, bool LEAN, bool MULTI, bool RG, bool POLICY = false, bool REGROWS = false>
__global__ void __launch_bounds__(256) tb_step_kernel(const uint32_t* __restrict__ k_words, const uint8_t* __restrict__ k_done, const float* __restrict__ k_actions,
                                                      const float4* __restrict__ k_hull, int k_n, int k_nhull, KArgs A) {
  // The leading arguments repeat A.words / done_state / actions / hull / n / P.n_hull as separate,
  // restrict-qualified kernel arguments: the compiler then knows that the state loads every launch starts
  // with cannot alias the stores it ends with. (Preloading them into SGPRs at wave launch,
  // -amdgpu-kernarg-preload-count, was measured too: no further gain for Tennisbot, -5 % for SwingRacket.)
  constexpr int NA = Dims<KIND>::A, NO = Dims<KIND>::O;
  // TABLE_IN_MEMORY (tb_step on Tennisbot): the outline table stays where it is. A launch that runs ONE substep reads an entry at
  // most once, and most lanes read none (edges and cull planes are for balls at the racket, the parameter block's copy for
  // resets) -- while copying 2.5 KB into LDS behind a barrier costs every launch up to 0.4 us (tools/diag/lanes_per_wave.hip).
  // Tennisbot 4096 envs: 689 -> 727 M env steps/s, 1 M: 19.2 -> 19.5 G. The pipelined SwingRacket step kernel, at its SGPR limit,
  // pays more for the table's addresses than the copy costs it (918 -> 899 M, 32768 envs 5.17 -> 4.80 G): it keeps the LDS copy,
  // like every kernel that loops (fast-forward, tb_rollout, the fused policy).
  constexpr bool TABLE_IN_MEMORY = !POLICY && !MULTI && KIND == TB_ENV_TENNIS;
  // LAZYTAB (tb_step on pipelined SwingRacket without the extended contact set): the LDS copy is made by the first wave that reads it (substep's SF_LAZYTAB form)
  constexpr bool LAZYTAB = !POLICY && !MULTI && KIND == TB_ENV_SWING && LEAN && !RG;
  constexpr unsigned FORM = (RG ? SF_RG : 0u) | (REGROWS ? SF_REGROWS : 0u) | (LAZYTAB ? SF_LAZYTAB : 0u);
  __shared__ float4 s_lds_hull[TABLE_IN_MEMORY ? 1 : TB_HULL_LDS];
  __shared__ __attribute__((aligned(16))) float s_mean[POLICY ? 64 * 8 : 4];
  // POLICY: 256-thread workgroups, four waves per 64 envs, each running both towers of a 16-env slice (see policy_towers); wave 0 steps the envs
  const int i = POLICY ? blockIdx.x * 64 + (threadIdx.x & 63) : blockIdx.x * blockDim.x + threadIdx.x;
  // measured at 4096 envs: Tennisbot +5.6 % (687 -> 726 M env steps/s); SwingRacket -6 % if it uses them too
  // (its kernels sit at the SGPR limit), so SwingRacket keeps reading the struct
  constexpr bool SEP = KIND == TB_ENV_TENNIS;
  const uint32_t* __restrict__ w_words = SEP ? k_words : A.words;
  const uint8_t* __restrict__ w_done = SEP ? k_done : A.done_state;
  const float* __restrict__ w_actions = SEP ? k_actions : A.actions;  // (SwingRacket: the compiler loads this pointer inside the `live` branch, a
                                                                     //  second scalar round trip in front of the action loads; forcing it into the first batch
                                                                     //  of kernel-argument loads was measured: 925 -> 908 M env steps/s at 4096 envs)
  const float4* __restrict__ w_hull = SEP ? k_hull : A.hull;
  const int w_n = SEP ? k_n : A.n, w_nhull = SEP ? k_nhull : A.P.n_hull;
  const float4* const s_hull = TABLE_IN_MEMORY ? w_hull : s_lds_hull;
  const bool live = i < w_n;
  EnvRegs e;
  TB_DIAG_NOW(t_entry);
  TB_DIAG_TRACE_ENTRY(trace_slot);
  TB_DIAG_CADENCE_ENTRY(cad_t0, cad_n);
  // issue every load this launch depends on back to back -- state rows, the first step's actions,
  // the outline table -- so that their latencies overlap instead of queueing behind the barrier
  float a[NA];
  if (live && !(POLICY && threadIdx.x >= 64)) {
    load_env<KIND>(w_words, w_done, w_n, i, e);
    if (!POLICY) load_actions<KIND>(w_actions, (size_t)i, a);
  }
#if TB_HINT_TENNIS_CONSTANTS
  if constexpr (KIND == TB_ENV_TENNIS && !POLICY && !MULTI) {
    // The constants of a free-flight substep are wanted HERE, i.e. fetched beside the state loads in flight: left alone the compiler
    // sinks some of their scalar loads to where they are used, each behind a wait that a lone wave cannot hide (~0.12 us: what a
    // build with every scalar load hoisted shows). An empty asm that names them is enough. Tennisbot 4096 envs 742 -> 768 M env
    // steps/s, larger batches unchanged; the same in the SwingRacket step kernel costs it 3 %, so it is not done there.
    asm volatile("" :: "s"(A.P.dt), "s"(A.P.gravity), "s"(A.P.lin_damp), "s"(A.P.lin_damp_quad), "s"(A.P.racket_inv_mass), "s"(A.P.ball_inv_mass),
                 "s"(A.P.hull_bound_radius), "s"(A.P.hull_margin), "s"(A.P.ball_radius), "s"(A.P.contact_threshold), "s"(A.P.static_top), "s"(A.P.max_ang_step));
  }
#endif
#if TB_HINT_TENNIS_OUTPUTS
  if constexpr (KIND == TB_ENV_TENNIS && !POLICY && !MULTI)
    asm volatile("" :: "s"(A.obs), "s"(A.reward), "s"(A.done_out), "s"(A.substeps));  // ... and where the outputs go: 769 -> 783 M (SwingRacket: 1125 -> 1119 M, not done there either)
#endif
  Manifold M;
  init_manifold(M, POLICY ? (int)(threadIdx.x & 63) : (int)threadIdx.x, POLICY ? 64 : (int)blockDim.x, !REGROWS);
  bool had_contacts = false;
  if constexpr (RG) {
    if (live && !(POLICY && threadIdx.x >= 64)) { had_contacts = A.mflag[i] != 0; if (had_contacts) load_manifold(A, i, M); }
  }
  if (POLICY) {
    // one barrier for both hand-offs (outline table, action means); the outline rows are requested
    // before the towers' operands and parked in a register meanwhile
    static_assert(TB_HULL_LDS <= 256, "one outline row per thread");
    const bool has_row = (int)threadIdx.x < 2 * w_nhull || ((int)threadIdx.x >= TB_HULL_PLANES && (int)threadIdx.x < TB_HULL_LDS);
    float4 row = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (has_row) row = w_hull[threadIdx.x];
    policy_towers<KIND>(A, s_mean);
    if (has_row) s_lds_hull[threadIdx.x] = row;
    __syncthreads();
    if (threadIdx.x >= 64) return;  // no barrier below this point
    if (live) policy_sample<KIND>(A, s_mean, i, e, a);
  } else if (!TABLE_IN_MEMORY && !LAZYTAB) {
    for (int k = threadIdx.x; k < 2 * w_nhull; k += blockDim.x) s_lds_hull[k] = w_hull[k];
    for (int k = TB_HULL_PLANES + threadIdx.x; k < TB_HULL_LDS; k += blockDim.x) s_lds_hull[k] = w_hull[k];
    __syncthreads();
  }
  TB_DIAG_WAIT_LOADS(live);
  TB_DIAG_NOW(t_loaded);

  uint32_t cnt[TB_N_COUNTERS];
#pragma unroll
  for (int k = 0; k < TB_N_COUNTERS; ++k) cnt[k] = 0u;

  TB_DIAG_STAMPS_BEGIN(st);
  TB_DIAG_NOW(t_kernel0);
  TB_DIAG_REALTIME(rt_kernel0);
  if (live) {
    bool any_reset = false;
    int ns_total = 0;
    const int n_steps = MULTI ? A.T : 1;
    for (int t = 0; t < n_steps; ++t) {
      const size_t row = (size_t)t * A.n + i;
      if (!POLICY && t > 0) load_actions<KIND>(A.actions, row, a);
      float o[NO];
      int ns = 1;
      bool d, parked = false;
      float rew;
      if (KIND == TB_ENV_SWING) {
        rew = swing_step<FORM>(A.P, s_hull, e, M, a, ns, cnt, LEAN || A.defer != 0, parked TB_STAMP_PASS, w_hull);
        make_obs<TB_ENV_SWING>(e, o);
        d = e.done != TB_DONE_NO;  // swingracket_env.py:145 returns self.done
        if (parked) {
          // Every SwingRacket episode ends inside this step (the loop only exits through done), so
          // done = 1 is known now; reward, terminal obs and substep count of this step are written
          // later by tb_ff_kernel from the parked state. The env itself restarts immediately.
          if (A.ff_rec) {
            park_env<RG>(A.ff_rec, i, e, M);
            if (A.ff_flag) A.ff_flag[i] = 1;  // (a byte array of its own: cleared by the fast-forward with one coalesced store per wave, where a 4-byte
                                              //  store into each record cost a 64-byte memory write per env; the record's own tag says "parked" too)
            if (A.pool_dst_out) A.pool_dst_out[i] = A.reward + row;  // parked straight into the pool (TbOptions.ff_defer = 2): where its reward will go
          } else {
            cnt[8]++;  // lockstep invariant broken (see launch_step): reported, never silent
          }
          d = true;
        }
      } else {
        rew = tennis_step<FORM>(A.P, s_hull, e, M, a, o, d, cnt TB_STAMP_PASS);
      }
      cnt[6] += (uint32_t)(ns - 1);  // substeps beyond the first of each agent step
      ns_total += ns;
      if (!state_is_finite(e))
        cnt[7]++;
      if (d && (A.P.flags & TB_F_AUTO_RESET)) {
        cnt[5]++;
        if (A.term_obs && !parked) write_obs<KIND>(A.term_obs, (size_t)i, o);
        e.episode += 1u;
        reset_env<KIND>(A, s_hull + TB_HULL_KP, i, e);
        M.n = 0; M.deep = 0;  // a rebuilt world has no contacts yet
        make_obs<KIND>(e, o);
        any_reset = true;
      }
      write_obs<KIND>(A.obs, row, o);
      A.reward[row] = rew;
      A.done_out[row] = d ? 1 : 0;
    }
    TB_DIAG_ADD_LANE0(15, stamp_now() - t_loaded);  // compute + output stores issued
    if (A.substeps) A.substeps[i] = ns_total;
    store_env<KIND>(A.words, A.done_state, A.n, i, e, any_reset);
    if constexpr (RG) { if (M.n > 0 || had_contacts) store_manifold(A, i, M, had_contacts); }
  }
  TB_DIAG_ADD_LANE0(7, t_loaded - t_entry);  // state + outline loads landed
  flush_counters(A.counters, cnt);
  // (the first substep of every env in every agent step is counted by the HOST, see count_first_substeps: one atomic per launch
  //  from one lane was 2.7 % of the 4096-env rate)
  TB_DIAG_STAMPS_END(st);
  TB_DIAG_ADD_LANE0(8, stamp_now() - t_kernel0);  // per-wave scalars: cycles in the kernel, waves, 100 MHz ticks
  TB_DIAG_ADD_LANE0(9, 1);
  TB_DIAG_ADD_LANE0(14, __builtin_amdgcn_s_memrealtime() - rt_kernel0);
  TB_DIAG_TRACE_EXIT(trace_slot);
  TB_DIAG_CADENCE_EXIT(cad_t0, cad_n);
}

// T agent steps with the policy inside, ONE launch: no launch boundary, no state round trip between the
// steps of an episode. A workgroup owns E = 16 S envs: waves 0 .. 2S-1 are the towers of tb_policy.hpp -- wave w is tower
// w / S (pi, vf) of the 16-env slice w % S -- their weight fragments loaded once and resident in registers for the whole
// launch; wave 2S holds the E envs' state in registers (lanes >= E idle) and steps them. Per step: towers (obs from LDS) ->
// barrier -> the env wave samples, steps, writes the step's outputs and the new observations to LDS -> barrier. The two
// role branches execute the same number of barriers. S = 1 (three waves per 16 envs) for batches that would otherwise
// leave CUs without a workgroup -- 4096 envs: 256 workgroups, one per CU, every tower alone on its SIMD's matrix pipe;
// S = 3 (seven waves per 48 envs) where the chip is full anyway and a 16-lane env wave would waste VALU issue slots
// (S = 4, nine waves, would put three waves on one SIMD: 168 VGPRs each, and the env wave's ~200 spill).
// SwingRacket episodes end at most once per launch, at its last step (the host cuts rollouts at episode ends): those
// lanes are parked for tb_ff_kernel exactly as in the pipelined step kernel. Same arithmetic per env as tb_policy_step,
// step after step: identical results.
// RG: the extended contact set compiled in (racket<->court manifold cache in the env wave's LDS columns, rolling-friction rows).
template <int KIND, int S, bool RG>
__global__ void __launch_bounds__((2 * S + 1) * 64) tb_policy_rollout_kernel(KArgs A) {
  constexpr int NA = Dims<KIND>::A, NO = Dims<KIND>::O, E = TB_POLICY_SLICE * S;
  __shared__ float4 s_hull[TB_HULL_LDS];
  __shared__ __attribute__((aligned(16))) float s_mean[E * 8];
  __shared__ __attribute__((aligned(16))) float s_obs[E * NO];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int k = threadIdx.x; k < 2 * A.P.n_hull; k += blockDim.x) s_hull[k] = A.hull[k];
  for (int k = TB_HULL_PLANES + threadIdx.x; k < TB_HULL_LDS; k += blockDim.x) s_hull[k] = A.hull[k];
  if (wave < 2 * S) {
    const int tower = wave / S, slice = wave % S, grp = lane >> 4;
    const int slot = slice * TB_POLICY_SLICE + (lane & 15), env = blockIdx.x * E + slot;
    const int env_c = env < A.n ? env : A.n - 1;
    TowerRegs<KIND> regs;
    regs.load(A.pol_weights + tower * tower_floats<KIND>(), lane);
    __syncthreads();
    for (int t = 0; t < A.T; ++t) {
      float x0[TowerRegs<KIND>::NC0], out[4];
      if (t == 0) policy_inputs<KIND>(A.pol_obs + (size_t)env_c * NO, lane, x0);
      else policy_inputs<KIND>(s_obs + slot * NO, lane, x0);
      regs.apply(x0, out);
      if (tower == 0) { if (grp < 2) *reinterpret_cast<float4*>(s_mean + slot * 8 + grp * 4) = make_float4(out[0], out[1], out[2], out[3]); }
      else if (lane < 16 && env < A.n) A.pol_value[(size_t)t * A.st_val + env] = out[0];
      __syncthreads();  // the action means of step t are in LDS
      __syncthreads();  // the observations after step t are in LDS
    }
    return;
  }
  const int i = blockIdx.x * E + lane;
  const bool live = lane < E && i < A.n;
  // Lanes without an env (48 of the 64 at S = 1) step a DUMMY: a racket hovering at rest, a ball a kilometre up, a step counter that
  // never reaches an episode end. It touches nothing, asks for nothing and is never stored -- but its lane is IN the substep, so the
  // racket narrowphase can hand every lane one edge of an asking env's outline sweep (outline_sweep_wide).
  EnvRegs e;
  {
    const vec3 z3 = mk(0.0f, 0.0f, 0.0f);
    e.r.p = mk(0.0f, 0.0f, 10.0f); e.r.q.x = 0.0f; e.r.q.y = 0.0f; e.r.q.z = 0.0f; e.r.q.w = 1.0f; e.r.v = z3; e.r.w = z3;
    e.b.p = mk(100.0f, 100.0f, 1000.0f); e.b.v = z3; e.b.w = z3;
#pragma unroll
    for (int k = 0; k < 6; ++k) e.aux[k] = 0.0f;
    e.aux[3] = 1.0f;  // (Tennisbot: the racket scale; SwingRacket: spawn y)
    e.aux[5] = 1.0f;  // (SwingRacket: d0)
    e.step_count = -(1 << 30); e.episode = 0u; e.done = TB_DONE_NO;
  }
  if (live) load_env<KIND>(A.words, A.done_state, A.n, i, e);
  Manifold M;
  init_manifold(M, lane, 64, KIND == TB_ENV_SWING);  // SwingRacket: static rows in LDS; Tennisbot keeps them in registers (REGROWS below)
  bool had_contacts = false;
  if constexpr (RG) {
    if (live) { had_contacts = A.mflag[i] != 0; if (had_contacts) load_manifold(A, i, M); }
  }
  uint32_t cnt[TB_N_COUNTERS];
#pragma unroll
  for (int k = 0; k < TB_N_COUNTERS; ++k) cnt[k] = 0u;
  bool any_reset = false;
  TB_DIAG_STAMPS_BEGIN(st);
  __syncthreads();
  float stdv[NA], lstd[NA];
#pragma unroll
  for (int k = 0; k < NA; ++k) { lstd[k] = A.pol_weights[2 * tower_floats<KIND>() + k]; stdv[k] = expf(lstd[k]); }
  // The free-flight constants of the substep as VECTOR registers for the whole launch. As kernel arguments they are scalar loads
  // that the compiler, at its SGPR limit in this kernel, re-issues inside the per-step loop (two dozen of them, each behind a wait
  // the lone env wave cannot hide); the env wave has ~300 vector registers to spare, and a value that went through an empty asm
  // cannot be fetched again.
  KParams Pl = A.P;
#if TB_HINT_POLICY_VGPR_PARAMS
#define TB_PIN(f) asm volatile("" : "+v"(Pl.f))
  TB_PIN(dt); TB_PIN(gravity); TB_PIN(lin_damp); TB_PIN(ang_damp); TB_PIN(lin_damp_quad); TB_PIN(ang_damp_quad); TB_PIN(max_ang_step); TB_PIN(contact_threshold);
  TB_PIN(racket_inv_mass); TB_PIN(racket_inertia[0]); TB_PIN(racket_inertia[1]); TB_PIN(racket_inertia[2]);
  TB_PIN(racket_inv_inertia[0]); TB_PIN(racket_inv_inertia[1]); TB_PIN(racket_inv_inertia[2]);
  TB_PIN(racket_half_thick); TB_PIN(hull_margin); TB_PIN(hull_bound_radius); TB_PIN(ball_inv_mass); TB_PIN(ball_radius); TB_PIN(magnus_k); TB_PIN(static_top);
#undef TB_PIN
#endif
  for (int t = 0; t < A.T; ++t) {
    float eps[NA];
    if (live) policy_draw<KIND>(A, i, e, eps);  // while the towers run
    __syncthreads();  // the action means of step t are in LDS
    {
      float a[NA], raw[NA], o[NO], logp = 0.0f;
#pragma unroll
      for (int k = 0; k < NA; ++k) { a[k] = 0.0f; raw[k] = 0.0f; }
      if (live) logp = policy_sample_regs<NA>(s_mean + lane * 8, eps, stdv, lstd, raw, a);
      int ns = 1;
      bool d = false, parked = false;
      float rew;
      // (every lane of the wave, dummies included: see above)
      // (the wide sweep where 48 of the 64 lanes are dummies; with 48 envs per wave -- S = 3 -- its one-query-at-a-time loop loses to every
      //  lane sweeping for itself: PPO collect under the trained policy, same box, S = 1: 570-574 -> 592-595 M env steps/s, S = 3: 461 -> 426 M)
      constexpr unsigned FORM = (RG ? SF_RG : 0u) | SF_COLD | (S == 1 ? SF_WIDE : 0u);
      if (KIND == TB_ENV_SWING) rew = swing_step<FORM>(Pl, s_hull, e, M, a, ns, cnt, true, parked TB_STAMP_PASS);  // never loops in here: see tb_step_kernel<LEAN>
      else rew = tennis_step<FORM | SF_REGROWS>(Pl, s_hull, e, M, a, o, d, cnt TB_STAMP_PASS);
      if (live) {
        if (KIND == TB_ENV_SWING) {
          make_obs<TB_ENV_SWING>(e, o);
          d = e.done != TB_DONE_NO;
          if (parked) {
            if (A.ff_rec) {
              park_env<RG>(A.ff_rec, i, e, M);
              if (A.ff_flag) A.ff_flag[i] = 1;
              if (A.pool_dst_out) A.pool_dst_out[i] = A.reward + (size_t)t * A.st_rew + i;
            } else {
              cnt[8]++;  // lockstep invariant broken (see launch_policy_rollout): reported, never silent
            }
            d = true;
          }
        }
        cnt[6] += (uint32_t)(ns - 1);
        if (!state_is_finite(e))
          cnt[7]++;
        if (d) {  // (rollouts require TB_F_AUTO_RESET)
          cnt[5]++;
          e.episode += 1u;
          reset_env<KIND>(A, s_hull + TB_HULL_KP, i, e);
          M.n = 0; M.deep = 0;
          make_obs<KIND>(e, o);
          any_reset = true;
        }
#pragma unroll
        for (int k = 0; k < NO; ++k) s_obs[lane * NO + k] = o[k];
      }
      // The towers wait for the observations only: THEY are in LDS now. The step's rows go to memory behind the barrier, while
      // the towers already run step t + 1 (the env wave's next stop is the barrier behind their means: ~1.2 us away).
      __syncthreads();  // the observations after step t are in LDS
      if (live) {
        store_row2<NA>(A.pol_raw + (size_t)t * A.st_raw, (size_t)i, raw);
        store_row2<NA>(A.pol_actions + (size_t)t * A.st_act, (size_t)i, a);
        A.pol_logp[(size_t)t * A.st_logp + i] = logp;
        write_obs<KIND>(A.obs + (size_t)t * A.st_obs, (size_t)i, o);
        A.reward[(size_t)t * A.st_rew + i] = rew;
        A.done_out[(size_t)t * A.st_done + i] = d ? 1 : 0;
      }
    }
  }
  if (live) {
    store_env<KIND>(A.words, A.done_state, A.n, i, e, any_reset);
    if constexpr (RG) { if (M.n > 0 || had_contacts) store_manifold(A, i, M, had_contacts); }
  }
  flush_counters(A.counters, cnt);
}

// progress mark (tb_mark_record): one thread bumps a counter in pinned host memory. Relaxed on purpose: the kernels this
// one is ordered behind have completed, their end-of-kernel release included, before it starts; a release of its own
// would only write the L2 back once more, under the step kernels that are running by then
__global__ void tb_mark_kernel(unsigned long long* count) {
  __hip_atomic_fetch_add(count, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// tb_set_racket_scale: one stream-ordered 4-byte store into the device-resident parameter block
__global__ void tb_poke_kernel(float* dst, float v) { *dst = v; }

// Orders the parked records of a slot by predicted flight length (predict_flight), 1024 at a time: counting sort over
// 256 bins in LDS, each thread then writes its own record to its sorted place in a second buffer, which tb_ff_kernel
// runs over 64 records per wave. Why: the fast-forward loop's cost is set by the slowest lane of a wave and by the contact
// paths ANY lane enters (wave votes). 64 random envs: mean flight 108 substeps, maximum ~170, every lane landing in a
// substep of its own (one contact solve per lane, paid by the whole wave). Sorted, the lanes of a wave finish together
// and are in the same phase of the flight. A kernel of its own (not a prologue of tb_ff_kernel) so that the fast-forward
// waves stay independent one-wave workgroups: a workgroup's registers are only released when its LAST wave ends, and
// sorted workgroups would hold their short-flight waves' slots idle until their longest flight has landed (measured:
// -20 % at 1 M envs). The source record's parked flag is cleared here; results never depend on which lane runs which env.
#define TB_FF_SORT_BLOCK 1024
template <bool RG>
__global__ void __launch_bounds__(TB_FF_SORT_BLOCK) tb_ff_sort_kernel(KArgs A, float4* sorted) {
  constexpr int TB_FF_REC = ff_rec<RG>();
  __shared__ int s_hist[256];
  const int lane = threadIdx.x & 63;
  const int src = blockIdx.x * TB_FF_SORT_BLOCK + threadIdx.x;
  if (threadIdx.x < 256) s_hist[threadIdx.x] = 0;
  float4 r[TB_FF_REC];
#pragma unroll
  for (int k = 0; k < TB_FF_REC; ++k) r[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  int key = 255;  // not parked / beyond the batch: behind every real flight
  if (src < A.n) {
    float4* g = A.ff_rec + (size_t)src * TB_FF_REC;
#pragma unroll
    for (int k = 0; k < TB_FF_REC; ++k) r[k] = g[k];
    const bool parked = A.ff_flag[src] != 0;
    r[7].z = __uint_as_float(parked ? 1u : 0u);  // in `sorted` the record's own tag says whether it is parked
    if (parked) {
      A.ff_flag[src] = 0;  // the copy in `sorted` is the parked one from here on
      const int it = predict_flight(A.P, mk(r[3].y, r[3].z, r[3].w), mk(r[4].x, r[4].y, r[4].z));
      key = it < 254 ? it : 254;
    }
  }
  __syncthreads();
  const int rank = atomicAdd(&s_hist[key], 1);
  __syncthreads();
  if (threadIdx.x < 64) {  // exclusive scan of the 256 bins by one wave: 4 bins per lane
    const int c0 = s_hist[4 * lane], c1 = s_hist[4 * lane + 1], c2 = s_hist[4 * lane + 2], c3 = s_hist[4 * lane + 3];
    const int sum = c0 + c1 + c2 + c3;
    int inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { int o = __shfl_up(inc, off, 64); if (lane >= off) inc += o; }
    const int base = inc - sum;
    s_hist[4 * lane] = base; s_hist[4 * lane + 1] = base + c0; s_hist[4 * lane + 2] = base + c0 + c1; s_hist[4 * lane + 3] = base + c0 + c1 + c2;
  }
  __syncthreads();
  float4* d = sorted + ((size_t)blockIdx.x * TB_FF_SORT_BLOCK + (size_t)(s_hist[key] + rank)) * TB_FF_REC;  // (the buffer is padded to whole groups)
#pragma unroll
  for (int k = 0; k < TB_FF_REC; ++k) d[k] = r[k];
}

// Finishes parked SwingRacket fast-forwards (side stream): same device code as the in-step loop, lane by lane.
// What this kernel adds is lane utilisation. A wave loops until its slowest lane is done, and with random actions the
// flights are 103 substeps for 80 % of the envs (balls that were never struck drop from the same height) but 111 at the
// 90th percentile, 167 at the 99th and up to 775 -- decided by what happens DURING the loop (the tumbling racket strikes
// the ball, it lands on the goal or the net), not by anything the parked state shows. 64 random lanes wait for their
// maximum: ~170 substeps per wave for a mean of 108 (63 %). So the loop runs in PHASES: every lane gets a budget --
// the ballistic estimate of its ball's remaining flight (predict_flight) plus a margin -- and a lane whose env is still
// running when the budget is spent writes its state to a compacted list and leaves; the next phase kernel runs those
// survivors, packed 64 to a wave again, with a new estimate; the last phase has no budget. Measured wave-substeps per
// 64 envs: 170 -> 112 + 0.1 x 60 + ... ~ 125. The state a survivor carries is exactly the loop's state (the restoring
// force is a function of it), so results are bit-identical however the phases cut (tests/test_gpu_parity.py).
//   phase 1: wave w takes the A.ff_lanes records [w L, (w+1) L) of the slot (L < 64: few envs per wave at small batches);
//   phases 2+: grid-stride over the *A.ff_src_count survivors of the previous phase, 64 per wave.
// BIG: the instantiation for batches that fill the chip several times over (occupancy counts: cull planes re-read from
// LDS, see racket_planes, and the outline sweep shared by the wave); below that the loop's latency counts and the planes stay in registers.
// ESC: the first phase of a BIG fast-forward with a phase behind it also hands over every env whose ball reaches the racket (substep's SF_ESC form).
// (ESC without the extended contact set is also built for four waves per SIMD: 125 VGPRs without spills in rounds 1-2, 128 with 6
//  spilled under round 3's build flags, where three waves at 129 VGPRs measure the same -- and, with the
//  two-slot static rows of substep's SF_ESC form, 9.5 KB of LDS per wave: 16 waves per CU instead of 12; 1 M envs, same box: 9.3-9.4 ->
//  10.0 G env steps/s)
// POOL (up to 131 072 envs, TbOptions.ff_defer): THE POOL. Two uses of the same instantiation: (1) the POOL RUN -- whole episodes
// that the step kernels parked straight into the pool (ff_defer = 2: the automatic choice up to 16 384 envs, see defer_mode), or
// stragglers that earlier launches moved on to it, finished by ONE launch when the caller joins: A.pool_dst_in gives every record its
// destination; (2) ff_defer = 1, DEFERRED STRAGGLERS: a fast-forward kernel lasts as long as its slowest env, and
// at most four of them run at once (one per hardware queue). With random actions that is 370 us for a mean flight of 108 substeps;
// under a trained policy struck balls fly 300-775 substeps (0.9-2.5 ms per kernel: the PPO collect was bound by it, 229 M env
// steps/s), and with racket<->court contact a ball at rest on a grounded racket runs to the 800-substep limit at 12-20 us per
// substep (15 ms per kernel: 21 M). So every env gets its ballistic estimate (at most an un-struck ball's) + ff_extra substeps, and one that is still running
// then is parked once more -- into a pool that all episodes share. The pool is run to its end by ONE launch of this kernel when the
// caller joins (tb_flush and everything that flushes): its hundreds of stragglers advance side by side, 64 to a wave, instead of
// one or two per kernel. Same arithmetic per env, same results, complete after the flush as before. A lane that finds the pool
// full (ff_cap records) finishes its loop here instead; lanes that pass that check together may overshoot the capacity by what
// all resident waves can hold, and the pool is allocated with that much slack.
constexpr int TB_PHASE_LANES = 64;    // survivors per wave in the phase kernels behind the first (32 measured: EXPERIMENTS.md)
constexpr int TB_PHASE_GRID_DIV = 256; // their grid: n / 256 one-wave workgroups (measured / 128 ... / 1024)
constexpr int TB_BUDGET_MARGIN = 8;
template <bool RG, bool BIG, bool ESC = false, bool POOL = false>
__global__ void __launch_bounds__(64, (ESC && !RG) ? 4 : 1) tb_ff_kernel(KArgs A) {
  static_assert(!POOL || !ESC, "the pool has no hand-over phase behind it");
  constexpr int TB_FF_REC = ff_rec<RG>();
  __shared__ float4 s_hull[TB_HULL_LDS];
  const int lane = threadIdx.x & 63;
  stage_hull(s_hull, A);
  uint32_t cnt[TB_N_COUNTERS];
#pragma unroll
  for (int k = 0; k < TB_N_COUNTERS; ++k) cnt[k] = 0u;
  TB_DIAG_STAMPS_BEGIN(st);
  int n_src = A.ff_src_count ? *A.ff_src_count : A.n;
  if (A.pool_dst_in && n_src > A.ff_cap) n_src = A.ff_cap;  // (the pool's counter runs on past its capacity; what did not fit was finished in place)
  int wave_lanes = A.ff_lanes;
  if (POOL && A.pool_dst_in) {
    // the pool run: one wave per SIMD before any wave gets a second record -- its lanes are the long, contact-heavy flights, every
    // one on a path of its own, and a wave pays for the sum of its lanes' paths
    const int want = (n_src + 1023) / 1024;
    wave_lanes = 4;
    while (wave_lanes < want && wave_lanes < 64) wave_lanes <<= 1;
  }
  for (int base = blockIdx.x * wave_lanes; base < n_src; base += gridDim.x * wave_lanes) {
    float4 r[TB_FF_REC];
    bool live = false;
    const int src = base + lane;
    if (lane < wave_lanes && src < n_src) {
      const float4* g = A.ff_rec + (size_t)src * TB_FF_REC;
#pragma unroll
      for (int k = 0; k < TB_FF_REC; ++k) r[k] = g[k];
      live = A.ff_flag ? A.ff_flag[src] != 0 : (__float_as_uint(r[7].z) & 255u) != 0u;
    }
    bool unfinished = false;
    EnvRegs e;
    Manifold M;
    init_manifold(M, lane, 64, true);
    int i = 0, ns = 1;  // fresh from the step kernel: it ran the first substep of this agent step
    if (live) {
      unpark_env<RG>(r, e, M, i);
      const uint32_t tag = __float_as_uint(r[7].z);
      if ((tag & 255u) == 2u) ns = (int)(tag >> 8);  // a survivor of an earlier phase: substeps so far
      const vec3 zero = mk(0.0f, 0.0f, 0.0f);
      // the first loop substep runs without any force (the accumulators were cleared by the agent's substep), every later
      // one with the restoring force of the state before it (swingracket_env.py:135-141): what a resumed env recomputes
      const vec3 F0 = e.step_count > 26 ? restoring_force(e) : zero;
      // POOL: the estimate is capped at an un-struck ball's flight (104 + 8 substeps): what flies longer -- under a trained policy
      // most balls -- is finished with everybody else's long flights at the join, not four kernels at a time
      int budget = 0x7fffffff;
      if (A.ff_next) {
        budget = 4 * predict_flight(A.P, e.b.p, e.b.v) + TB_BUDGET_MARGIN;  // substeps beyond the ballistic estimate before a lane is handed to the next phase
        if (POOL) budget = (budget < 112 ? budget : 112) + A.ff_extra;
      }
      const int ns0 = ns;
      // (small batches: the racket<->court rows of a solve in registers -- one wave per SIMD anyway, and a grounded racket's lane is alone in its wave)
      constexpr unsigned FORM = (RG ? SF_RG : 0u) | (BIG ? SF_RELOAD : 0u) | (ESC ? SF_ESC : 0u) | (RG && !BIG ? SF_REGGROUND : 0u);
      float rew = swing_loop<FORM, true>(A.P, s_hull, e, M, F0, zero, true, false, unfinished, ns, cnt TB_STAMP_PASS, budget);
      if constexpr (POOL) {
        // a full pool: finish here after all (the counter is read, not reserved: see the slack above)
        if (unfinished && A.ff_next && *reinterpret_cast<volatile int*>(A.ff_next_count) >= A.ff_cap) {
          unfinished = false;
          rew = swing_loop<FORM, true>(A.P, s_hull, e, M, restoring_force(e), zero, true, false, unfinished, ns, cnt TB_STAMP_PASS, 0x7fffffff);
        }
      }
      cnt[6] += (uint32_t)(ns - ns0);
      if (!unfinished) {
        if (!state_is_finite(e))
          cnt[7]++;
        float o[TB_SWING_OBS_DIM];
        make_obs<TB_ENV_SWING>(e, o);
        if (A.term_obs) write_obs<TB_ENV_SWING>(A.term_obs, (size_t)i, o);
        // (a survivor has earned nothing yet: every reward of the loop is paid in its last substep)
        if (POOL && A.pool_dst_in) *A.pool_dst_in[src] = rew;  // the pool kernel: each record brought its own destination
        else A.reward[i] = rew;
        if (A.substeps) A.substeps[i] = ns;
      }
      if (A.ff_flag) A.ff_flag[src] = 0;  // the record is free again (lists and sorted copies are rewritten whole before their next use)
      // the pool run: a consumed record says so itself. A region's records are expected to be rewritten whole by the next launch that
      // parks into it -- but an env that does NOT park there (the lockstep invariant broken: counters[8]) would leave this record, with
      // its destination pointer, to be run once more by the next pool run. One 4-byte store per episode end.
      if (POOL && A.pool_dst_in) reinterpret_cast<uint32_t*>(A.ff_rec + (size_t)src * TB_FF_REC + 7)[2] = 0u;
    }
    if (A.ff_next) {  // survivors: one atomic per wave reserves their places in the next phase's list
      const unsigned long long m = __ballot(unfinished);
      if (m) {
        int first = 0;
        if (lane == 0) first = atomicAdd(A.ff_next_count, __popcll(m));
        first = __shfl(first, 0, 64);
        if (unfinished) {
          park_env<RG>(A.ff_next, first + __popcll(m & ((1ull << lane) - 1ull)), e, M);
          uint32_t* w = reinterpret_cast<uint32_t*>(A.ff_next + (size_t)(first + __popcll(m & ((1ull << lane) - 1ull))) * TB_FF_REC + 7);
          w[2] = 2u | ((uint32_t)ns << 8);
          w[3] = (uint32_t)i;
          if (POOL && A.pool_dst_out) A.pool_dst_out[first + __popcll(m & ((1ull << lane) - 1ull))] = A.reward + i;
        }
      }
    }
  }
  flush_counters(A.counters, cnt);
  TB_DIAG_STAMPS_END(st);
  TB_DIAG_ADD_LANE0(9, 1);
}

// reset kernel (masked)
template <int KIND>
__global__ void __launch_bounds__(256) tb_reset_kernel(KArgs A) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= A.n) return;
  if (A.mask && !A.mask[i]) return;
  EnvRegs e;
  e.episode = A.words[(size_t)(Dims<KIND>::W - 1) * A.n + i] + 1u;
  reset_env<KIND>(A, A.hull + TB_HULL_KP, i, e);
  store_env<KIND>(A.words, A.done_state, A.n, i, e, true);
  A.mflag[i] = 0;  // a rebuilt world has no contacts yet
  if (A.obs) {
    float o[Dims<KIND>::O];
    make_obs<KIND>(e, o);
    write_obs<KIND>(A.obs, (size_t)i, o);
  }
}

// identity orientation, episode = -1 so that the first reset starts episode 0
__global__ void tb_init_kernel(uint32_t* words, uint8_t* done, int n, int nwords) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int k = 0; k < nwords; ++k) words[(size_t)k * n + i] = 0u;
  words[(size_t)(TB_W_RQ + 3) * n + i] = __float_as_uint(1.0f);
  if (nwords == TB_TENNIS_WORDS) words[(size_t)TB_W_TN_SCALE * n + i] = __float_as_uint(1.0f);
  words[(size_t)(nwords - 1) * n + i] = 0xFFFFFFFFu;
  done[i] = TB_DONE_NO;
}

// diagnostics: SoA dword copy with the step kernel's access pattern (PMC calibration)
__global__ void __launch_bounds__(256) tb_diag_copy_kernel(const uint32_t* src, uint32_t* dst, int n, int rows) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int r = 0; r < rows; ++r) dst[(size_t)r * n + i] = src[(size_t)r * n + i];
}

// diagnostics: `gridDim.x` one-wave workgroups that do nothing but stay resident (s_sleep) until the 100 MHz real-time counter has
// advanced by `ticks` -- what the fast-forward waves look like to the dispatcher, without their arithmetic (tools/diag/r03_idle_probe.py)
__global__ void __launch_bounds__(64) tb_diag_idle_kernel(unsigned long long ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

// ------------------------------------------------------------------------------------------
// host side
thread_local char g_err[512] = "";

int fail(int code, const char* what) {
  if (code > 0) snprintf(g_err, sizeof g_err, "%s: %s (%s)", what, hipGetErrorString((hipError_t)code), hipGetErrorName((hipError_t)code));
  else snprintf(g_err, sizeof g_err, "%s", what);
  return code;
}
#define HIP_TRY(expr)                                              \
  do {                                                             \
    hipError_t _e = (expr);                                        \
    if (_e != hipSuccess) return fail((int)_e, #expr);             \
  } while (0)

struct DeviceGuard {  // calls run on the handle's device without disturbing the caller's current device
  int prev = -1;
  bool switched = false;
  hipError_t err = hipSuccess;
  explicit DeviceGuard(int dev) {
    err = hipGetDevice(&prev);
    if (err == hipSuccess && prev != dev) { err = hipSetDevice(dev); switched = err == hipSuccess; }
  }
  ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
};

bool kind_ok(int k) { return k == TB_ENV_SWING || k == TB_ENV_TENNIS; }

int validate_params(const TbParams* p) {
  if (p->n_hull < 3 || p->n_hull > TB_MAX_HULL) return fail(TB_E_PARAMS, "TbParams.n_hull must be in [3, 64]");
  if (!(p->dt > 0.0f) || !(p->inv_dt > 0.0f)) return fail(TB_E_PARAMS, "TbParams.dt / inv_dt must be positive");
  if (!(p->racket_inv_mass > 0.0f) || !(p->ball_inv_mass > 0.0f) || !(p->ball_inv_inertia > 0.0f)) return fail(TB_E_PARAMS, "TbParams masses must be positive");
  for (int i = 0; i < 3; ++i)
    if (!(p->racket_inertia[i] > 0.0f) || !(p->racket_inv_inertia[i] > 0.0f)) return fail(TB_E_PARAMS, "TbParams.racket_inertia must be positive");
  if (!(p->ball_radius > 0.0f) || !(p->contact_threshold >= 0.0f)) return fail(TB_E_PARAMS, "TbParams.ball_radius / contact_threshold invalid");
  if (p->solver_iters < 1 || p->solver_iters > 1000) return fail(TB_E_PARAMS, "TbParams.solver_iters must be in [1, 1000]");
  if (!(p->solver_tol >= 0.0f)) return fail(TB_E_PARAMS, "TbParams.solver_tol must be >= 0");
  return TB_OK;
}

void to_kparams(const TbParams* p, KParams* k, float* planes) {
  k->dt = p->dt; k->inv_dt = p->inv_dt; k->gravity = p->gravity; k->lin_damp = p->lin_damp; k->ang_damp = p->ang_damp; k->lin_damp_quad = p->lin_damp_quad; k->ang_damp_quad = p->ang_damp_quad;
  k->max_ang_step = p->max_ang_step; k->rest_vel_threshold = p->rest_vel_threshold; k->erp = p->erp;
  k->contact_threshold = p->contact_threshold; k->solver_iters = p->solver_iters; k->flags = p->flags; k->solver_tol = p->solver_tol;
  k->racket_inv_mass = p->racket_inv_mass;
  for (int i = 0; i < 3; ++i) {
    k->racket_inertia[i] = p->racket_inertia[i]; k->racket_inv_inertia[i] = p->racket_inv_inertia[i];
    k->racket_com[i] = p->racket_com[i]; k->ground_half[i] = p->ground_half[i]; k->net_half[i] = p->net_half[i];
  }
  k->racket_half_thick = p->racket_half_thick; k->hull_margin = p->hull_margin; k->hull_bound_radius = p->hull_bound_radius; k->racket_scale = p->racket_scale;
  k->ball_inv_mass = p->ball_inv_mass; k->ball_inv_inertia = p->ball_inv_inertia; k->ball_radius = p->ball_radius;
  k->magnus_k = p->magnus_k; k->ball_spin_max = p->ball_spin_max;
  k->rest_racket = p->rest_racket; k->rest_court = p->rest_court; k->rest_goal = p->rest_goal;
  k->fric_racket = p->fric_racket; k->fric_court = p->fric_court; k->fric_goal = p->fric_goal;
  k->rest_racket_court = p->rest_racket_court; k->fric_racket_court = p->fric_racket_court; k->racket_ground_threshold = p->racket_ground_threshold;
  k->roll_racket = p->roll_racket; k->roll_court = p->roll_court; k->roll_goal = p->roll_goal;
  k->goal_radius = p->goal_radius; k->goal_half_len = p->goal_half_len;
  k->n_hull = p->n_hull;
  float top = p->ground_half[2] > p->goal_half_len ? p->ground_half[2] : p->goal_half_len;
  if ((p->flags & TB_F_NET) && p->net_half[2] > top) top = p->net_half[2];
  k->static_top = top;
  {  // cull planes: 8 fixed directions + the 4 longest edges, each pushed out by its own rounding
    int np = 0;
    for (int d = 0; d < 8; ++d) {
      double ang = d * 0.78539816339744830962, ny = cos(ang), nz = sin(ang), hmax = -1e30;
      for (int i = 0; i < p->n_hull; ++i) { double v = ny * p->hull_edges[i][0] + nz * p->hull_edges[i][1]; hmax = v > hmax ? v : hmax; }
      planes[3 * np] = (float)ny; planes[3 * np + 1] = (float)nz; planes[3 * np + 2] = (float)(hmax + 1e-6); ++np;
    }
    bool used[TB_MAX_HULL] = {false};
    for (int pick = 0; pick < TB_N_CULL - 8; ++pick) {
      int best = -1; double bl = -1.0;
      for (int i = 0; i < p->n_hull; ++i) {
        double l2 = (double)p->hull_edges[i][2] * p->hull_edges[i][2] + (double)p->hull_edges[i][3] * p->hull_edges[i][3];
        if (!used[i] && l2 > bl) { bl = l2; best = i; }
      }
      used[best] = true;
      double il = 1.0 / sqrt(bl), ny = p->hull_edges[best][3] * il, nz = -p->hull_edges[best][2] * il;  // outward normal of a CCW edge
      double hmax = -1e30;
      for (int i = 0; i < p->n_hull; ++i) { double v = ny * p->hull_edges[i][0] + nz * p->hull_edges[i][1]; hmax = v > hmax ? v : hmax; }
      planes[3 * np] = (float)ny; planes[3 * np + 1] = (float)nz; planes[3 * np + 2] = (float)(hmax + 1e-6); ++np;
    }
  }
  // same float operations as the rows would do per contact (oracle setup_row): bit-identical
  k->ball_kn = 1.0f / p->ball_inv_mass;
  k->ball_kt = 1.0f / fmaf(p->ball_inv_inertia, p->ball_radius * p->ball_radius, p->ball_inv_mass);
}

}  // namespace

#define TB_FF_SLOTS 8  // parked-state buffers + side streams: ~2.5 fast-forwards are in flight in steady state
#define TB_PIPELINE_MAX_ENVS (1 << 24)
#define TB_DEFER_MAX_ENVS 131072  // deferred stragglers: a small-batch scheme (large batches run the fast-forward in phases)

struct TbHandle {
  int device, kind, n, block;
  TbOptions opt;  // as given to tb_create (0 = auto)
  int reg_rows;  // Tennisbot step kernel with the static contact rows in registers
  int swing_reg_rows;  // the same for the pipelined SwingRacket step kernel (+2.7 % at 4096 envs; NOT for tb_ff_kernel)
  uint64_t seed, env_id_base;
  TbParams params;
  KParams kp;
  uint32_t* d_words;
  uint8_t* d_done;
  float4* d_hull;
  float4* h_hull;  // pinned staging copy of the outline table
  float cull_planes[TB_N_CULL][3];  // derived from the outline (to_kparams); they travel behind it in the same table
  unsigned long long* d_counters;     // [TB_COUNTER_SHARDS][TB_N_COUNTERS]
  uint32_t* d_mani;                   // [TB_MANI_WORDS][n] racket<->court contact caches
  uint8_t* d_mflag;                   // [n]
  // pipelined fast-forward
  int pipeline;            // enabled by tb_set_pipeline
  int phase, phase_valid;  // agent steps since the last full reset (SwingRacket episodes are exactly 26 steps)
  int phase_at_capture, phase_valid_at_capture;  // snapshot taken by tb_pipeline_sync(h, 1), restored by tb_pipeline_sync(h, 0) / tb_pipeline_recover
  int params_generation;   // tb_set_params count: captured launches carry the parameter block they were captured with
  hipStream_t side[TB_FF_SLOTS];  // one stream per slot: consecutive fast-forwards overlap each other too
  const void *last_term, *last_sub;  // shared late-written buffers force ordering between fast-forwards
  int last_slot;
  float4* d_ff_rec[TB_FF_SLOTS];  // [n][ff_rec<RG>()] parked records (park_env), allocated for TB_FF_REC_MAX
  uint8_t* d_ff_flag[TB_FF_SLOTS];  // [n] parked flags
  float4* d_ff_sorted[TB_FF_SLOTS];  // ff_sort: the slot's records in the order tb_ff_sort_kernel gives them, padded to whole sort groups
  float4* d_ff_list[TB_FF_SLOTS][2];  // survivors of fast-forward phases 1 and 2 (worst case: every env), compacted
  int* d_ff_count[TB_FF_SLOTS];       // [2] their numbers
  int ff_phases;                      // 1 = one kernel runs every loop to its end; 2, 3 = budgeted phases + survivor kernels
  unsigned long long first_substeps;  // counters[6], the host's share: n envs x agent steps of every launch that RAN (see count_first_substeps)
  int ff_lanes, ff_sort;          // how tb_ff_kernel hands records to lanes (TbOptions.ff_lanes_per_wave / ff_sort, or chosen from n)
  // deferred stragglers (tb_ff_kernel<.., POOL>; TbOptions.ff_defer): one pool for all episodes between two flushes
  float4* d_pool;                 // [pool_cap + pool_slack][TB_FF_REC_MAX]
  float** d_pool_dst;             // [pool_cap + pool_slack] where each deferred env's terminal reward goes
  int* d_pool_count;
  int pool_cap, pool_slack, pool_pending;  // pending: records may be waiting (the next flush runs the pool kernel)
  int pool_run_upto;              // ... of which the first pool_run_upto have had their launch already (at a progress mark)
  int pool_episodes;              // ff_defer = 2: episodes parked straight into the pool since the last flush (records [k n, (k + 1) n) each)
  hipEvent_t ev_direct;           // ... and the latest launch that did so (a flush on another stream waits for it)
  hipEvent_t ev_pool;             // the last pool run (+ the reset of its counter): later fast-forwards append behind it, whatever stream flushed
  int pool_ev_valid, direct_ev_valid;
  hipEvent_t ev_step[TB_FF_SLOTS], ev_ff[TB_FF_SLOTS];
  int ff_busy[TB_FF_SLOTS], next_slot;
  // progress marks (tb_mark_record). h_marks: pinned host counters written by tb_mark_kernel -- [k] firings of mark k
  // (a kernel on the caller's own stream), [TB_MAX_MARKS] fast-forwards finished (a kernel behind every tb_ff_kernel on
  // its side stream). No extra streams, no extra graph edges: a mark never makes anything wait. What a mark still has
  // to wait for -- the fast-forwards enqueued before it -- is host arithmetic over these two kinds of counters.
  unsigned long long* h_marks;
  unsigned long long snap_ff[TB_FF_SLOTS], snap_marks[TB_MAX_MARKS];  // the counters at tb_mark_begin (nothing of this handle in flight)
  int marks_on;                               // tb_mark_enable: fast-forwards are followed by their counting kernel
  // fast-forwards enqueued PER SLOT (= per side stream: only there is "the first k have finished" the same as "k have
  // finished" -- fast-forwards of different episodes overtake each other, one with a ball at rest on a grounded racket
  // runs five times as long as the next): inside the current / latest capture; eagerly since tb_mark_begin
  long long ff_cap[TB_FF_SLOTS], ff_eager[TB_FF_SLOTS];
  long long mark_ff_before[TB_MAX_MARKS][TB_FF_SLOTS];  // fast-forwards enqueued per slot before the mark: inside a capture ff_cap at that point, eagerly ff_eager (mark_in_capture says which)
  int mark_in_capture[TB_MAX_MARKS];
};

namespace {

int words_of(int kind) { return kind == TB_ENV_SWING ? TB_SWING_WORDS : TB_TENNIS_WORDS; }

// dynamic LDS of a stepping kernel (see init_manifold): per lane, the static rows unless in registers + the cache if RG
#ifdef TB_DIAG_LDS_PAD  // (tools/diag/r03_occupancy_probe.py: fewer workgroups per CU through a padded dynamic LDS request; tb_diag_set_lds_pad)
size_t g_diag_lds_pad = 0;
#else
constexpr size_t g_diag_lds_pad = 0;
#endif
size_t dyn_lds(bool regrows, bool rg, unsigned lanes) { return g_diag_lds_pad + sizeof(float) * lanes * ((regrows ? 0 : TB_ROWS_LDS) + (rg ? TB_MANI_LDS : 0)); }

int ensure_marks(TbHandle* h) {
  if (h->h_marks) return TB_OK;
  HIP_TRY(hipHostMalloc((void**)&h->h_marks, sizeof(unsigned long long) * (TB_MAX_MARKS + TB_FF_SLOTS), hipHostMallocDefault));
  memset(h->h_marks, 0, sizeof(unsigned long long) * (TB_MAX_MARKS + TB_FF_SLOTS));
  return TB_OK;
}

// 128-thread workgroups, measured with 64 / 128 / 256 alternated in one process (tools/diag/diag_blocks2.py; M env steps/s):
//   SwingRacket  4096: 662-679 / 673-684 / 657-682    32768: 4570 / 4760 / 3600    65536: 4900 / 4600 / 4250    131072: 5900 / 5760 / 5450
//                262144: 7630 / 7630 / 7390            1 M: 8980 / 9000 / 8830
//   Tennisbot    4096: 667 / 666 / 669    32768: 3736 / 3825 / 3800    65536: 6300 / 6350 / 6430    262144: 13250 / 13450 / 13390    1 M: 18400 / 18900 / 18300
// i.e. 128 everywhere but for SwingRacket between 64 K and 128 K envs, where one wave per workgroup wins by 2-6 %.
int pick_block(int kind, int n, const TbOptions& o) {
  if (o.block == 64 || o.block == 128 || o.block == 256) return o.block;
  // (round 3, final build, no barrier left in the one-substep kernels: one-wave workgroups win up to 16384 envs -- SwingRacket 4096 envs
  //  1052 against 1033 M env steps/s, 16384: 3.07 / 3.01 G, Tennisbot 4096: 738 / 733 M, 8192: 1.24 / 1.23 G; 32768: 5.65 / 5.68 and 3.78 / 3.82 G)
  if (n <= 16384) return 64;
  return kind == TB_ENV_SWING && n >= 49152 && n <= 131072 ? 64 : 128;
}

KArgs base_args(const TbHandle* h) {
  KArgs a;
  memset(&a, 0, sizeof a);
  a.P = h->kp; a.words = h->d_words; a.done_state = h->d_done; a.hull = h->d_hull; a.counters = h->d_counters;
  a.mani = h->d_mani; a.mflag = h->d_mflag;
  a.seed = h->seed; a.env_id_base = h->env_id_base; a.n = h->n; a.T = 1;
  return a;
}

int upload_hull(TbHandle* h, hipStream_t s) {
  memcpy(h->h_hull, h->params.hull_edges, sizeof(float) * TB_HULL_REC * TB_MAX_HULL);
  memcpy(reinterpret_cast<float*>(h->h_hull + TB_HULL_PLANES), h->cull_planes, sizeof h->cull_planes);
  memset(h->h_hull + TB_HULL_KP, 0, sizeof(float4) * TB_KP_ROWS);
  memcpy(h->h_hull + TB_HULL_KP, &h->kp, sizeof h->kp);
  HIP_TRY(hipMemcpyAsync(h->d_hull, h->h_hull, sizeof(float4) * TB_HULL_LDS, hipMemcpyHostToDevice, s));
  return TB_OK;
}

// make `s` wait for every fast-forward still running on the side stream
int wait_side(TbHandle* h, hipStream_t s) {
  for (int k = 0; k < TB_FF_SLOTS; ++k)
    if (h->ff_busy[k]) HIP_TRY(hipStreamWaitEvent(s, h->ev_ff[k], 0));
  return TB_OK;
}

// the RG template instantiations hold what the default kernels leave out: racket <-> court contact and rolling friction
bool extended_contacts(const KParams& kp) {
  return (kp.flags & TB_F_RACKET_GROUND) || kp.roll_racket > 0.0f || kp.roll_court > 0.0f || kp.roll_goal > 0.0f;
}

// finish the lanes parked in `slot` on that slot's side stream, ordered after everything issued to `s` so far
// (Measured and dropped in round 3: enqueueing the fast-forward one launch LATE, so that under stream capture the next step -- not
//  the fast-forward -- is the parking node's first successor. It does what was hoped for the chain -- all 2132 step kernels of two
//  replays on ONE hardware queue instead of 572 / 520 / 520 / 520 -- but a replayed graph then puts every second successor on the same
//  second queue: 79 of 82 fast-forwards in line behind each other, 209 M env steps/s instead of 700.)
int defer_mode(const TbHandle* h);
int launch_ff(TbHandle* h, int slot, const KArgs& a_in, const void* term, const void* substeps, hipStream_t s) {
  KArgs a = a_in;
  hipStream_t side = h->side[slot];
  // lockstep episodes (every env parks in the same launch): sorted, or a few envs per wave; without the host knowing the
  // phase every step is followed by this kernel and nearly every record is idle: plain 64 per wave, one flag test each
  const bool sort = h->ff_sort && h->phase_valid;
  a.ff_lanes = sort || !h->phase_valid ? 64 : h->ff_lanes;
  const int groups = (h->n + TB_FF_SORT_BLOCK - 1) / TB_FF_SORT_BLOCK;
  HIP_TRY(hipEventRecord(h->ev_step[slot], s));
  HIP_TRY(hipStreamWaitEvent(side, h->ev_step[slot], 0));
  // two fast-forwards that write the same terminal-obs / substeps buffer must finish in order
  if (h->last_slot >= 0 && h->last_slot != slot && ((term && term == h->last_term) || (substeps && substeps == h->last_sub)))
    HIP_TRY(hipStreamWaitEvent(side, h->ev_ff[h->last_slot], 0));
  if (sort) {
    if (extended_contacts(h->kp)) hipLaunchKernelGGL(tb_ff_sort_kernel<true>, dim3((unsigned)groups), dim3(TB_FF_SORT_BLOCK), 0, side, a, h->d_ff_sorted[slot]);
    else hipLaunchKernelGGL(tb_ff_sort_kernel<false>, dim3((unsigned)groups), dim3(TB_FF_SORT_BLOCK), 0, side, a, h->d_ff_sorted[slot]);
    HIP_TRY(hipGetLastError());
    a.ff_rec = h->d_ff_sorted[slot]; a.ff_flag = nullptr; a.n = groups * TB_FF_SORT_BLOCK;  // (outputs are addressed by the env index each record carries)
  }
  // phases: budgeted loop + survivor kernels (see tb_ff_kernel). Without the host knowing the episode phase nearly every
  // record is idle: one plain kernel.
  const int phases = h->phase_valid ? h->ff_phases : 1;
  const bool rg = extended_contacts(h->kp);
  // deferred stragglers: on request (TbOptions.ff_defer > 0), or by default with racket<->court contact, whose resting stacks run
  // to the 800-substep limit. Not with progress marks (a mark promises that the steps before it are FINAL), not with late-written
  // terminal observations / substep counts (the pool keeps one destination per record: the reward's)
  const bool defer = h->d_pool && phases == 1 && h->phase_valid && !sort && !term && !substeps && defer_mode(h) == 1;
  if (phases > 1) HIP_TRY(hipMemsetAsync(h->d_ff_count[slot], 0, 2 * sizeof(int), side));
  for (int ph = 0; ph < phases; ++ph) {
    KArgs k = a;
    dim3 grid((unsigned)((a.n + a.ff_lanes - 1) / a.ff_lanes)), block(64);
    if (ph > 0) {  // survivors of phase ph: a compacted list of unknown length, walked by a fixed grid
      k.ff_rec = h->d_ff_list[slot][ph - 1]; k.ff_flag = nullptr; k.ff_src_count = h->d_ff_count[slot] + (ph - 1); k.ff_lanes = TB_PHASE_LANES;
      int g = h->n / TB_PHASE_GRID_DIV; g = g < 64 ? 64 : g;  // (1 M envs, same box: / 512 9.37, / 256 9.56, / 128 9.41, / 1024 9.19 G env steps/s)
      grid = dim3((unsigned)g);
    }
    if (ph + 1 < phases) { k.ff_next = h->d_ff_list[slot][ph]; k.ff_next_count = h->d_ff_count[slot] + ph; }
    if (defer) {
      // with racket<->court contact every lane is on a path of its own (rackets land at different times, manifolds of different
      // sizes, solves of different lengths) and a wave pays for the union: 16 envs per wave (4096 envs, same box: 84-86 M env
      // steps/s with 64, 92-96 with 32, 94-98 with 16, 93-97 with 8, 89 with 4)
      if (rg && !h->opt.ff_lanes_per_wave && k.ff_lanes > 16) { k.ff_lanes = 16; grid = dim3((unsigned)((a.n + 15) / 16)); }
      k.ff_next = h->d_pool; k.ff_next_count = h->d_pool_count; k.ff_cap = h->pool_cap; k.pool_dst_out = h->d_pool_dst;
      k.ff_extra = h->opt.ff_defer_margin ? h->opt.ff_defer_margin : 16;
      h->pool_pending = 1;
      if (h->pool_ev_valid) HIP_TRY(hipStreamWaitEvent(side, h->ev_pool, 0));  // append behind the last pool run and its counter reset
      if (rg) hipLaunchKernelGGL((tb_ff_kernel<true, false, false, true>), grid, block, dyn_lds(false, true, 64), side, k);
      else hipLaunchKernelGGL((tb_ff_kernel<false, false, false, true>), grid, block, dyn_lds(false, false, 64), side, k);
      HIP_TRY(hipGetLastError());
      continue;
    }
    const bool big = h->n >= 131072;
    const bool esc = big && ph == 0 && phases > 1;
    const size_t lds = esc && !rg ? sizeof(float) * 64 * TB_ROWS_LDS_TWO : dyn_lds(false, rg, 64);
    if (esc) { if (rg) hipLaunchKernelGGL((tb_ff_kernel<true, true, true>), grid, block, lds, side, k); else hipLaunchKernelGGL((tb_ff_kernel<false, true, true>), grid, block, lds, side, k); }
    else if (rg) { if (big) hipLaunchKernelGGL((tb_ff_kernel<true, true>), grid, block, lds, side, k); else hipLaunchKernelGGL((tb_ff_kernel<true, false>), grid, block, lds, side, k); }
    else { if (big) hipLaunchKernelGGL((tb_ff_kernel<false, true>), grid, block, lds, side, k); else hipLaunchKernelGGL((tb_ff_kernel<false, false>), grid, block, lds, side, k); }
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipGetLastError());
  if (h->h_marks && h->marks_on) {  // progress marks: count this fast-forward as finished, in stream order behind it
    hipLaunchKernelGGL(tb_mark_kernel, dim3(1), dim3(1), 0, side, h->h_marks + TB_MAX_MARKS + slot);
    HIP_TRY(hipGetLastError());
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    HIP_TRY(hipStreamIsCapturing(s, &st));
    if (st == hipStreamCaptureStatusActive) h->ff_cap[slot]++; else h->ff_eager[slot]++;
  }
  HIP_TRY(hipEventRecord(h->ev_ff[slot], side));
  h->ff_busy[slot] = 1; h->last_slot = slot; h->last_term = term; h->last_sub = substeps;
  return TB_OK;
}

// ONE launch for what the pool holds: the whole episodes parked straight into it (ff_defer = 2) that no launch has been given to yet
// -- regions [pool_run_upto, pool_episodes): the host knows how many records -- or, without any, the stragglers that the episodes'
// own fast-forward kernels moved on to it (ff_defer = 1: their number is the pool's device counter)
int run_pool(TbHandle* h, hipStream_t q) {
  KArgs k = base_args(h);
  const bool rg = extended_contacts(h->kp);
  k.ff_rec = h->d_pool; k.ff_flag = nullptr; k.ff_src_count = h->d_pool_count; k.ff_lanes = 64;
  k.ff_cap = h->pool_cap + h->pool_slack; k.pool_dst_in = h->d_pool_dst;
  long long records = (long long)h->pool_cap + h->pool_slack;
  if (h->pool_episodes > 0) {
    const size_t first = (size_t)h->pool_run_upto * h->n;
    records = (long long)(h->pool_episodes - h->pool_run_upto) * h->n;
    k.ff_src_count = nullptr; k.n = (int)records;
    k.ff_rec = h->d_pool + first * (rg ? TB_FF_REC_MAX : 8); k.pool_dst_in = h->d_pool_dst + first;
  }
  long long g = (records + 63) / 64;
  g = g < 1024 ? 1024 : g > 16384 ? 16384 : g;  // (workgroups beyond the pool's fill exit at once; grid-stride beyond 1 M records)
  (void)hipGetLastError();
  // whole episodes in the pool make it a LARGE batch -- 43 episodes x 4096 envs = 2752 waves: the instantiation built for occupancy
  // (143 VGPRs, three waves per SIMD, wave-shared outline sweep) holds them all at once, the small-batch one (178 VGPRs, two per
  // SIMD) ran them in two rounds
  const bool big = !rg && h->pool_episodes > 0 && records >= 131072;
  if (rg) hipLaunchKernelGGL((tb_ff_kernel<true, false, false, true>), dim3((unsigned)g), dim3(64), dyn_lds(false, true, 64), q, k);
  else if (big) hipLaunchKernelGGL((tb_ff_kernel<false, true, false, true>), dim3((unsigned)g), dim3(64), dyn_lds(false, false, 64), q, k);
  else hipLaunchKernelGGL((tb_ff_kernel<false, false, false, true>), dim3((unsigned)g), dim3(64), dyn_lds(false, false, 64), q, k);
  HIP_TRY(hipGetLastError());
  h->pool_run_upto = h->pool_episodes;
  return TB_OK;
}

// every result of every fast-forward is in place once `s` gets past this point
int flush_all(TbHandle* h, hipStream_t s) {
  if (int rc = wait_side(h, s)) return rc;
  if (h->pool_pending) {  // what the episodes since the last flush left in the pool, side by side in one launch
    if (h->pool_episodes > 0 && h->direct_ev_valid) HIP_TRY(hipStreamWaitEvent(s, h->ev_direct, 0));  // (a flush on another stream than the steps')
    if (h->pool_episodes == 0 || h->pool_run_upto < h->pool_episodes) {
      if (int rc = run_pool(h, s)) return rc;
    }
    HIP_TRY(hipMemsetAsync(h->d_pool_count, 0, sizeof(int), s));
    HIP_TRY(hipEventRecord(h->ev_pool, s));
    h->pool_pending = 0; h->pool_ev_valid = 1; h->pool_episodes = 0; h->pool_run_upto = 0;
  }
  return TB_OK;
}

// Progress marks with ff_defer = 2: a mark promises that the steps before it are FINAL, so the episodes parked since the last mark
// (or flush) get their pool launch now -- on a side stream, beside the steps of the next chunk, counted like any fast-forward kernel.
// A graph of C chunks forks C times instead of once per episode.
int run_pool_for_mark(TbHandle* h, hipStream_t s) {
  if (!(h->pool_episodes > h->pool_run_upto)) return TB_OK;
  const int slot = h->next_slot;
  h->next_slot = (slot + 1) % TB_FF_SLOTS;
  hipStream_t side = h->side[slot];
  HIP_TRY(hipEventRecord(h->ev_step[slot], s));
  HIP_TRY(hipStreamWaitEvent(side, h->ev_step[slot], 0));
  if (int rc = run_pool(h, side)) return rc;
  if (h->h_marks && h->marks_on) {
    hipLaunchKernelGGL(tb_mark_kernel, dim3(1), dim3(1), 0, side, h->h_marks + TB_MAX_MARKS + slot);
    HIP_TRY(hipGetLastError());
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    HIP_TRY(hipStreamIsCapturing(s, &st));
    if (st == hipStreamCaptureStatusActive) h->ff_cap[slot]++; else h->ff_eager[slot]++;
  }
  HIP_TRY(hipEventRecord(h->ev_ff[slot], side));
  h->ff_busy[slot] = 1; h->last_slot = slot; h->last_term = nullptr; h->last_sub = nullptr;
  return TB_OK;
}

// TbOptions.ff_defer = 2: the launch that ends the episodes parks them STRAIGHT into the pool -- region [k n, (k + 1) n) for the
// k-th such launch since the last flush -- and no fast-forward kernel of its own follows: the pool run at the join does all of
// them at once. Returns whether `a` was set up that way (not with progress marks / late-written outputs / a full pool: then the
// ordinary slot + tb_ff_kernel path serves the launch).
// What TbOptions.ff_defer = 0 (auto) means for this handle: 2 -- every episode end straight into the pool -- up to 16384 envs, where
// the rollout is a chain of launch-bound step kernels and every fork of a replayed graph costs the CHAIN (the next step moves to
// another hardware queue: ~10 us per episode end, and 1.7-2.8 us between all other steps instead of ~1.2 us in a graph that is one
// single list): 4096 envs, same box, 679 -> 871 M env steps/s (1024: 160 -> 225 M, 8192: 1.30 -> 1.49 G, 16384: 2.58 -> 2.63 G, 32768:
// 4.80 -> 4.26 G; racket<->court contact at 4096 envs: 92 -> 115-127 M). Above that: 1 (stragglers only) with racket<->court contact up
// to the pool's size limit, else 0 -- large batches run their fast-forwards beside the steps, in phases.
// With progress marks on, the automatic choice stays with one kernel per episode end: a graph of 8 marked chunks whose chunks are
// all-gathered beside it (one rank, 4096 envs, same box) replays in 6.3 ms that way and in 6.6-7.1 ms with the pool run at each mark
// (ff_defer = 2 asks for that); form 1 never runs under marks (a mark promises final steps).
int defer_mode(const TbHandle* h) {
  if (!h->d_pool || h->opt.ff_defer < 0) return 0;
  if (h->opt.ff_defer > 0) return h->marks_on && h->opt.ff_defer == 1 ? 0 : h->opt.ff_defer;
  if (h->marks_on) return 0;
  if (h->n <= 16384) return 2;
  return (h->kp.flags & TB_F_RACKET_GROUND) ? 1 : 0;
}

bool park_direct(TbHandle* h, KArgs& a, const void* term, const void* substeps, hipStream_t s, int* rc) {
  *rc = TB_OK;
  if (!(h->d_pool && defer_mode(h) == 2 && h->phase_valid && !term && !substeps && h->pool_episodes < h->pool_cap / h->n)) return false;
  if (h->pool_ev_valid) {  // behind the last pool run (which may have been enqueued on another stream)
    hipError_t e = hipStreamWaitEvent(s, h->ev_pool, 0);
    if (e != hipSuccess) { *rc = fail((int)e, "hipStreamWaitEvent(s, h->ev_pool, 0)"); return false; }
  }
  const size_t rec = extended_contacts(h->kp) ? TB_FF_REC_MAX : 8;
  a.defer = 1; a.ff_rec = h->d_pool + (size_t)h->pool_episodes * h->n * rec; a.ff_flag = nullptr;
  a.pool_dst_out = h->d_pool_dst + (size_t)h->pool_episodes * h->n;
  return true;
}
int parked_direct(TbHandle* h, hipStream_t s) {
  HIP_TRY(hipEventRecord(h->ev_direct, s));
  h->direct_ev_valid = 1; h->pool_episodes++; h->pool_pending = 1;
  return TB_OK;
}

// The substep counter's host share. Every agent step runs at least one substep of every env: n x T per launch, known to the host --
// the kernels only count what goes beyond (fast-forward loops). Added when a launch is enqueued to run; a launch that is only being
// CAPTURED runs nothing: whoever replays the graph reports the replayed steps through tb_phase_advance, as it must for the episode
// phase anyway. (tb_counters joins the stream before it reads: what was enqueued has run by then.)
int count_first_substeps(TbHandle* h, int T, hipStream_t s) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  HIP_TRY(hipStreamIsCapturing(s, &st));
  if (st == hipStreamCaptureStatusNone) h->first_substeps += (unsigned long long)h->n * (unsigned long long)T;
  return TB_OK;
}

struct PolicyIO {  // non-null weights = fused policy step
  const float* weights; const float* obs_in; float* actions; float* raw; float* logp; float* value;
  unsigned long long seed; int deterministic;
};

// lean_multi (T > 1): the caller guarantees lockstep episodes (phase_valid) and that an episode can only
// end at the LAST of the T steps (phase + T <= 26), so the launch parks at most once per env and the
// pipelined kernel (no in-kernel fast-forward) can run several steps per launch too.
int launch_step(TbHandle* h, int T, const float* actions, float* obs, float* reward, uint8_t* done, float* term, int32_t* substeps, hipStream_t s,
                const PolicyIO* pol = nullptr, bool lean_multi = false) {
  if (int rc = count_first_substeps(h, T, s)) return rc;
  KArgs a = base_args(h);
  if (pol) {
    a.pol_weights = pol->weights; a.pol_obs = pol->obs_in; a.pol_actions = pol->actions; a.pol_raw = pol->raw; a.pol_logp = pol->logp;
    a.pol_value = pol->value; a.pol_seed = pol->seed; a.pol_deterministic = pol->deterministic;
  }
  a.actions = actions; a.obs = obs; a.reward = reward; a.done_out = done; a.term_obs = term; a.substeps = substeps; a.T = T;
  dim3 grid((unsigned)((h->n + h->block - 1) / h->block)), block((unsigned)h->block);
  if (pol) { grid = dim3((unsigned)((h->n + 63) / 64)); block = dim3(256); }  // four waves per 64 envs
  // Pipelined SwingRacket: the step kernel never loops (LEAN); a lane that starts a fast-forward is
  // parked and tb_ff_kernel finishes it on a side stream. When the host knows the episode phase (all
  // envs were reset together; episodes are exactly 26 steps) only the 26th call can park anything, so
  // only that call is followed by tb_ff_kernel; when it does not (masked reset, injected state), every
  // call gets a slot and a (then mostly idle) tb_ff_kernel.
  const bool piped = (T == 1 || lean_multi) && h->pipeline && h->kind == TB_ENV_SWING && (h->kp.flags & TB_F_AUTO_RESET);
  const bool may_park = piped && (T == 1 ? (!h->phase_valid || h->phase == 25) : h->phase + T - 1 == 25);
  int slot = -1, rc_direct = TB_OK;
  const bool direct = may_park && park_direct(h, a, term, substeps, s, &rc_direct);
  if (rc_direct) return rc_direct;
  if (may_park && !direct) {
    slot = h->next_slot;
    h->next_slot = (slot + 1) % TB_FF_SLOTS;
    if (h->ff_busy[slot]) HIP_TRY(hipStreamWaitEvent(s, h->ev_ff[slot], 0));  // slot still in use by an older fast-forward
    a.defer = 1; a.ff_rec = h->d_ff_rec[slot]; a.ff_flag = h->d_ff_flag[slot];
  }
  const bool rg = extended_contacts(h->kp);  // selects the instantiation that contains the rolling-friction rows
  const unsigned lanes = pol ? 64u : block.x;
  const size_t lds_rows = dyn_lds(false, rg, lanes), lds_regs = dyn_lds(true, false, lanes);  // instantiations with the static rows in LDS / in registers
  (void)hipGetLastError();  // the check below is about THIS launch, not about whatever another library left behind
#define TB_LAUNCH_STEP(KIND, LEAN, MULTI)                                                                      \
  do {                                                                                                         \
    if (pol && rg) hipLaunchKernelGGL((tb_step_kernel<KIND, LEAN, false, true, true>), grid, block, lds_rows, s, a.words, a.done_state, a.actions, a.hull, a.n, a.P.n_hull, a);   \
    else if (pol) hipLaunchKernelGGL((tb_step_kernel<KIND, LEAN, false, false, true>), grid, block, lds_rows, s, a.words, a.done_state, a.actions, a.hull, a.n, a.P.n_hull, a);   \
    else if (rg) hipLaunchKernelGGL((tb_step_kernel<KIND, LEAN, MULTI, true>), grid, block, lds_rows, s, a.words, a.done_state, a.actions, a.hull, a.n, a.P.n_hull, a);           \
    else hipLaunchKernelGGL((tb_step_kernel<KIND, LEAN, MULTI, false>), grid, block, lds_rows, s, a.words, a.done_state, a.actions, a.hull, a.n, a.P.n_hull, a);                  \
  } while (0)
  if (T > 1) {
    if (h->kind == TB_ENV_TENNIS) {
      if (h->reg_rows && !pol && !rg) hipLaunchKernelGGL((tb_step_kernel<TB_ENV_TENNIS, false, true, false, false, true>), grid, block, lds_regs, s, a.words, a.done_state, a.actions, a.hull, a.n, a.P.n_hull, a);
      else TB_LAUNCH_STEP(TB_ENV_TENNIS, false, true);
    } else if (piped) {
      if (!may_park) a.ff_rec = nullptr;
      TB_LAUNCH_STEP(TB_ENV_SWING, true, true);
    } else TB_LAUNCH_STEP(TB_ENV_SWING, false, true);
  } else if (h->kind == TB_ENV_TENNIS) {
    if (h->reg_rows && !rg && pol) hipLaunchKernelGGL((tb_step_kernel<TB_ENV_TENNIS, false, false, false, true, true>), grid, block, lds_regs, s, a.words, a.done_state, a.actions, a.hull, a.n, a.P.n_hull, a);
    else if (h->reg_rows && !rg) hipLaunchKernelGGL((tb_step_kernel<TB_ENV_TENNIS, false, false, false, false, true>), grid, block, lds_regs, s, a.words, a.done_state, a.actions, a.hull, a.n, a.P.n_hull, a);
    else TB_LAUNCH_STEP(TB_ENV_TENNIS, false, false);
  } else if (piped && h->swing_reg_rows && !pol && !rg) {
    if (!may_park) a.ff_rec = nullptr;  // (see the comment of the next branch but one)
    hipLaunchKernelGGL((tb_step_kernel<TB_ENV_SWING, true, false, false, false, true>), grid, block, lds_regs, s, a.words, a.done_state, a.actions, a.hull, a.n, a.P.n_hull, a);
  } else if (may_park) TB_LAUNCH_STEP(TB_ENV_SWING, true, false);
  else if (piped) {
    // phase known and not the 26th step: every env has step_count = phase < 25 (all were reset
    // together and every library call that could break lockstep clears phase_valid), so no lane
    // can start a fast-forward in this launch and the lean kernel needs no slot. Should the
    // invariant ever be broken, the lane is counted in counters[8] (lockstep violations) instead of being dropped silently.
    a.ff_rec = nullptr;
    TB_LAUNCH_STEP(TB_ENV_SWING, true, false);
  } else TB_LAUNCH_STEP(TB_ENV_SWING, false, false);
#undef TB_LAUNCH_STEP
  HIP_TRY(hipGetLastError());
  if (direct) { if (int rc = parked_direct(h, s)) return rc; }
  else if (may_park) {
    if (T > 1) a.reward = reward + (size_t)(T - 1) * h->n;  // the fast-forward owes its reward to the step that parked: the last one
    if (int rc = launch_ff(h, slot, a, term, substeps, s)) return rc;
  }
  if (h->phase_valid) h->phase = (h->phase + T) % 26;
  return TB_OK;
}

// one launch of tb_policy_rollout_kernel over T steps; SwingRacket: T ends where the episode does
int launch_policy_rollout(TbHandle* h, int T, const PolicyIO& pol, float* obs, float* reward, uint8_t* done, const size_t* st /*element strides*/,
                          hipStream_t s) {
  if (int rc = count_first_substeps(h, T, s)) return rc;
  KArgs a = base_args(h);
  a.pol_weights = pol.weights; a.pol_obs = pol.obs_in; a.pol_actions = pol.actions; a.pol_raw = pol.raw; a.pol_logp = pol.logp;
  a.pol_value = pol.value; a.pol_seed = pol.seed; a.pol_deterministic = pol.deterministic;
  a.obs = obs; a.reward = reward; a.done_out = done; a.T = T;
  a.st_act = st[0]; a.st_raw = st[1]; a.st_logp = st[2]; a.st_val = st[3]; a.st_obs = st[4]; a.st_rew = st[5]; a.st_done = st[6];
  const bool swing = h->kind == TB_ENV_SWING;
  const bool may_park = swing && h->phase + T - 1 == 25;  // (the caller checked pipeline, lockstep phase and phase + T <= 26)
  int slot = -1, rc_direct = TB_OK;
  const bool direct = may_park && park_direct(h, a, nullptr, nullptr, s, &rc_direct);
  if (rc_direct) return rc_direct;
  if (may_park && !direct) {
    slot = h->next_slot;
    h->next_slot = (slot + 1) % TB_FF_SLOTS;
    if (h->ff_busy[slot]) HIP_TRY(hipStreamWaitEvent(s, h->ev_ff[slot], 0));
    a.defer = 1; a.ff_rec = h->d_ff_rec[slot]; a.ff_flag = h->d_ff_flag[slot];
  }
  // 16 envs per workgroup (3 waves) while every workgroup still gets a CU of its own, else 48 (7 waves): see the kernel
  const bool narrow = h->opt.policy_slices ? h->opt.policy_slices == 1 : h->n <= 4096;
  const int E = narrow ? TB_POLICY_SLICE : 3 * TB_POLICY_SLICE;
  dim3 grid((unsigned)((h->n + E - 1) / E)), block(narrow ? 192 : 448);
  (void)hipGetLastError();
  const bool rg = extended_contacts(h->kp);
  const size_t lds = dyn_lds(!swing, rg, 64);  // the env wave's columns: static rows (SwingRacket) + the racket<->court cache (RG)
#define TB_LAUNCH_PR(KIND, SL)                                                                                 \
  do {                                                                                                         \
    if (rg) hipLaunchKernelGGL((tb_policy_rollout_kernel<KIND, SL, true>), grid, block, lds, s, a);            \
    else hipLaunchKernelGGL((tb_policy_rollout_kernel<KIND, SL, false>), grid, block, lds, s, a);              \
  } while (0)
  if (swing) { if (narrow) TB_LAUNCH_PR(TB_ENV_SWING, 1); else TB_LAUNCH_PR(TB_ENV_SWING, 3); }
  else { if (narrow) TB_LAUNCH_PR(TB_ENV_TENNIS, 1); else TB_LAUNCH_PR(TB_ENV_TENNIS, 3); }
#undef TB_LAUNCH_PR
  HIP_TRY(hipGetLastError());
  if (direct) { if (int rc = parked_direct(h, s)) return rc; }
  else if (may_park) {
    a.reward = reward + (size_t)(T - 1) * st[5];  // the fast-forward owes its reward to the step that parked: the last one
    if (int rc = launch_ff(h, slot, a, nullptr, nullptr, s)) return rc;
  }
  if (h->phase_valid) h->phase = (h->phase + T) % 26;
  return TB_OK;
}

}  // namespace

extern "C" {

int tb_abi_version(void) { return TB_ABI_VERSION; }
int tb_obs_dim(int k) { return k == TB_ENV_SWING ? TB_SWING_OBS_DIM : k == TB_ENV_TENNIS ? TB_TENNIS_OBS_DIM : TB_E_INVAL; }
int tb_act_dim(int k) { return k == TB_ENV_SWING ? TB_SWING_ACT_DIM : k == TB_ENV_TENNIS ? TB_TENNIS_ACT_DIM : TB_E_INVAL; }
int tb_state_words(int k) { return kind_ok(k) ? words_of(k) : TB_E_INVAL; }
const char* tb_last_error(void) { return g_err; }

int tb_create(const TbParams* params, const TbOptions* options, int env_kind, int n_envs, int device, uint64_t seed, uint64_t env_id_base, TbHandle** out) {
  if (!params || !out) return fail(TB_E_INVAL, "tb_create: null argument");
  *out = nullptr;
  if (!kind_ok(env_kind)) return fail(TB_E_INVAL, "tb_create: unknown env kind");
  if (n_envs <= 0 || n_envs > (1 << 25)) return fail(TB_E_INVAL, "tb_create: n_envs must be in [1, 2^25]");  // (32-bit row offsets: see row_word)
  if (int rc = validate_params(params)) return rc;
  TbOptions opt;
  memset(&opt, 0, sizeof opt);
  if (options) {  // a caller built against an older (shorter) TbOptions leaves the newer fields at "auto"
    if (options->struct_size < sizeof(uint32_t) || options->struct_size > 4096) return fail(TB_E_INVAL, "tb_create: TbOptions.struct_size is not set");
    memcpy(&opt, options, options->struct_size < sizeof opt ? options->struct_size : sizeof opt);
    if (opt.block != 0 && opt.block != 64 && opt.block != 128 && opt.block != 256) return fail(TB_E_INVAL, "tb_create: TbOptions.block must be 0, 64, 128 or 256");
    if (opt.ff_lanes_per_wave < 0 || opt.ff_lanes_per_wave > 64) return fail(TB_E_INVAL, "tb_create: TbOptions.ff_lanes_per_wave must be in [0, 64]");
    if (opt.ff_phases < 0 || opt.ff_phases > 3) return fail(TB_E_INVAL, "tb_create: TbOptions.ff_phases must be in [0, 3]");
    if (opt.policy_slices != 0 && opt.policy_slices != 1 && opt.policy_slices != 3) return fail(TB_E_INVAL, "tb_create: TbOptions.policy_slices must be 0, 1 or 3");
    if (opt.ff_defer < -1 || opt.ff_defer > 2) return fail(TB_E_INVAL, "tb_create: TbOptions.ff_defer must be -1, 0, 1 or 2");
    if (opt.ff_defer_margin < 0 || opt.ff_defer_margin > 800) return fail(TB_E_INVAL, "tb_create: TbOptions.ff_defer_margin must be in [0, 800]");
  }
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) return fail(TB_E_NODEVICE, "tb_create: no HIP device available (this library has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail(TB_E_NODEVICE, "tb_create: device index out of range");
  DeviceGuard g(device);
  if (g.err != hipSuccess) return fail((int)g.err, "hipSetDevice");

  TbHandle* h = (TbHandle*)calloc(1, sizeof(TbHandle));
  if (!h) return fail(TB_E_INVAL, "tb_create: out of host memory");
  h->device = device; h->kind = env_kind; h->n = n_envs; h->seed = seed; h->env_id_base = env_id_base;
  h->params = *params; to_kparams(params, &h->kp, &h->cull_planes[0][0]); h->block = pick_block(env_kind, n_envs, opt);
  h->opt = opt;
  // Tennisbot, measured in the steady state (envs past their first, synchronised episodes): +28 % at 4096 envs,
  // +8 % at 256 K, +16 % at 1 M, +12 % at 4 M; only the contact-free first episode after a common reset, where the
  // kernel runs at 70 % of HBM peak and occupancy counts, loses 3 % at 1 M envs
  h->reg_rows = env_kind == TB_ENV_TENNIS && (opt.tennis_reg_rows ? opt.tennis_reg_rows > 0 : 1);
  // (with the unpacked build of round 3 the register-row step kernel is 154 VGPRs, three waves per SIMD: it wins at every size now --
  //  131072 envs 8.13 against 7.68 G env steps/s, 262144: 8.35 / 8.21, 1 M: 11.64 / 11.11; until then it was chosen up to 131072 envs)
  h->swing_reg_rows = env_kind == TB_ENV_SWING && (opt.swing_reg_rows ? opt.swing_reg_rows > 0 : 1);
  // fast-forward: sort the lanes of large batches by predicted flight length; below 4096 envs a few envs per wave
  h->ff_sort = opt.ff_sort > 0;  // opt-in: pays when flight lengths can be told from the parked state (a trained policy's struck balls)
  // measured on one box: 3 phases +11 % at 1 M envs, +-0 at 256 K, -16 % at 32 K and 4096 (two more kernels in every episode's chain)
  h->ff_phases = opt.ff_phases >= 1 && opt.ff_phases <= 3 ? opt.ff_phases : (n_envs >= 262144 ? 3 : 1);
  h->ff_lanes = opt.ff_lanes_per_wave;
  if (!h->ff_lanes) { h->ff_lanes = 4; while (h->ff_lanes < 64 && (long long)h->ff_lanes * 64 < n_envs) h->ff_lanes <<= 1; }
  const int nw = words_of(env_kind);
  hipError_t err;
#define CREATE_TRY(expr) if ((err = (expr)) != hipSuccess) { int rc = fail((int)err, #expr); tb_destroy(h); return rc; }
  CREATE_TRY(hipMalloc((void**)&h->d_words, sizeof(uint32_t) * (size_t)nw * n_envs));
  CREATE_TRY(hipMalloc((void**)&h->d_done, (size_t)n_envs));
  CREATE_TRY(hipMalloc((void**)&h->d_hull, sizeof(float4) * TB_HULL_LDS));
  CREATE_TRY(hipHostMalloc((void**)&h->h_hull, sizeof(float4) * TB_HULL_LDS, hipHostMallocDefault));
  CREATE_TRY(hipMalloc((void**)&h->d_counters, sizeof(unsigned long long) * TB_N_COUNTERS * TB_COUNTER_SHARDS));
  CREATE_TRY(hipMemsetAsync(h->d_counters, 0, sizeof(unsigned long long) * TB_N_COUNTERS * TB_COUNTER_SHARDS, 0));
  CREATE_TRY(hipMalloc((void**)&h->d_mani, sizeof(uint32_t) * (size_t)TB_MANI_WORDS * n_envs));
  CREATE_TRY(hipMalloc((void**)&h->d_mflag, (size_t)n_envs));
  CREATE_TRY(hipMemsetAsync(h->d_mflag, 0, (size_t)n_envs, 0));
  hipLaunchKernelGGL(tb_init_kernel, dim3((unsigned)((n_envs + 255) / 256)), dim3(256), 0, 0, h->d_words, h->d_done, n_envs, nw);
  CREATE_TRY(hipGetLastError());
  if (int rc = upload_hull(h, 0)) { tb_destroy(h); return rc; }
  CREATE_TRY(hipStreamSynchronize(0));
#undef CREATE_TRY
  *out = h;
  return TB_OK;
}

static void release_pipeline(TbHandle* h);
static void release_pool(TbHandle* h);

int tb_destroy(TbHandle* h) {
  if (!h) return TB_OK;
  DeviceGuard g(h->device);
  (void)hipDeviceSynchronize();
  if (h->d_words) (void)hipFree(h->d_words);
  if (h->d_done) (void)hipFree(h->d_done);
  if (h->d_hull) (void)hipFree(h->d_hull);
  if (h->h_hull) (void)hipHostFree(h->h_hull);
  if (h->d_counters) (void)hipFree(h->d_counters);
  if (h->d_mani) (void)hipFree(h->d_mani);
  if (h->d_mflag) (void)hipFree(h->d_mflag);
  release_pipeline(h);
  if (h->h_marks) (void)hipHostFree(h->h_marks);
  free(h);
  return TB_OK;
}

// the pipeline's streams, events and buffers: all of them, or none (what release_pipeline leaves behind is the state of a
// handle whose pipeline was never enabled)
static void release_pipeline(TbHandle* h) {
  for (int k = 0; k < TB_FF_SLOTS; ++k) {
    if (h->d_ff_rec[k]) (void)hipFree(h->d_ff_rec[k]);
    if (h->d_ff_flag[k]) (void)hipFree(h->d_ff_flag[k]);
    if (h->d_ff_sorted[k]) (void)hipFree(h->d_ff_sorted[k]);
    if (h->d_ff_list[k][0]) (void)hipFree(h->d_ff_list[k][0]);
    if (h->d_ff_list[k][1]) (void)hipFree(h->d_ff_list[k][1]);
    if (h->d_ff_count[k]) (void)hipFree(h->d_ff_count[k]);
    if (h->ev_step[k]) (void)hipEventDestroy(h->ev_step[k]);
    if (h->ev_ff[k]) (void)hipEventDestroy(h->ev_ff[k]);
    if (h->side[k]) (void)hipStreamDestroy(h->side[k]);
    h->d_ff_rec[k] = nullptr; h->d_ff_flag[k] = nullptr; h->d_ff_sorted[k] = nullptr; h->d_ff_list[k][0] = nullptr; h->d_ff_list[k][1] = nullptr;
    h->d_ff_count[k] = nullptr; h->ev_step[k] = nullptr; h->ev_ff[k] = nullptr; h->side[k] = nullptr; h->ff_busy[k] = 0;
  }
  release_pool(h);
  h->pipeline = 0;
}

// test hook (tb_diag_fail_alloc): the n-th device allocation of the next tb_set_pipeline fails with hipErrorOutOfMemory
static int g_fail_alloc_countdown = 0;
static hipError_t pipeline_malloc(void** p, size_t bytes) {
  if (g_fail_alloc_countdown > 0 && --g_fail_alloc_countdown == 0) { *p = nullptr; return hipErrorOutOfMemory; }
  return hipMalloc(p, bytes);
}

// The pool (TbOptions.ff_defer): 64 n records (two reference-sized rollouts of 1100 steps with EVERY episode end in it) + the slack
// all resident fast-forward waves could overshoot it by (slots x n), 192 B each + an 8-byte destination pointer: 14 KB per env.
// Allocated only for handles whose defer_mode can be non-zero: on request, up to 16384 envs, or above that (to 131072) once the
// parameter block turns racket<->court contact on -- tb_set_params calls this again. (Until round 4 every pipelined handle up
// to 131072 envs got one: 1.9 GB at that size that the default kernels never touched.) Zeroed: a record's tag word says whether
// it holds a parked env, and no launch may ever find a tag it did not write.
static bool pool_wanted(const TbHandle* h) {
  if (h->n > TB_DEFER_MAX_ENVS || h->opt.ff_defer < 0) return false;
  return h->opt.ff_defer > 0 || h->n <= 16384 || (h->kp.flags & TB_F_RACKET_GROUND);
}
static void release_pool(TbHandle* h) {
  if (h->d_pool) (void)hipFree(h->d_pool);
  if (h->d_pool_dst) (void)hipFree(h->d_pool_dst);
  if (h->d_pool_count) (void)hipFree(h->d_pool_count);
  if (h->ev_pool) (void)hipEventDestroy(h->ev_pool);
  if (h->ev_direct) (void)hipEventDestroy(h->ev_direct);
  h->ev_pool = nullptr; h->ev_direct = nullptr; h->pool_ev_valid = 0; h->direct_ev_valid = 0; h->pool_episodes = 0; h->pool_run_upto = 0;
  h->d_pool = nullptr; h->d_pool_dst = nullptr; h->d_pool_count = nullptr; h->pool_cap = 0; h->pool_slack = 0; h->pool_pending = 0;
}
static int alloc_pool_parts(TbHandle* h, float4** pool, float*** dst, int** count, hipEvent_t* ev_pool, hipEvent_t* ev_direct, size_t recs) {
  HIP_TRY(pipeline_malloc((void**)pool, sizeof(float4) * (size_t)TB_FF_REC_MAX * recs));
  HIP_TRY(hipMemset(*pool, 0, sizeof(float4) * (size_t)TB_FF_REC_MAX * recs));
  HIP_TRY(pipeline_malloc((void**)dst, sizeof(float*) * recs));
  HIP_TRY(hipMemset(*dst, 0, sizeof(float*) * recs));
  HIP_TRY(pipeline_malloc((void**)count, sizeof(int)));
  HIP_TRY(hipMemset(*count, 0, sizeof(int)));
  HIP_TRY(hipEventCreateWithFlags(ev_pool, hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(ev_direct, hipEventDisableTiming));
  return TB_OK;
}
static int alloc_pool(TbHandle* h) {  // all of it or none: defer_mode takes a non-null d_pool for a usable pool
  if (h->d_pool || !pool_wanted(h)) return TB_OK;
  const size_t recs = (size_t)64 * h->n + (size_t)TB_FF_SLOTS * h->n;
  float4* pool = nullptr; float** dst = nullptr; int* count = nullptr; hipEvent_t e1 = nullptr, e2 = nullptr;
  if (int rc = alloc_pool_parts(h, &pool, &dst, &count, &e1, &e2, recs)) {
    if (pool) (void)hipFree(pool);
    if (dst) (void)hipFree(dst);
    if (count) (void)hipFree(count);
    if (e1) (void)hipEventDestroy(e1);
    if (e2) (void)hipEventDestroy(e2);
    return rc;
  }
  h->pool_cap = 64 * h->n; h->pool_slack = TB_FF_SLOTS * h->n;
  h->d_pool = pool; h->d_pool_dst = dst; h->d_pool_count = count; h->ev_pool = e1; h->ev_direct = e2;
  return TB_OK;
}

static int alloc_pipeline(TbHandle* h) {
  const size_t wb = sizeof(float4) * (size_t)TB_FF_REC_MAX * h->n;
  for (int k = 0; k < TB_FF_SLOTS; ++k) {
    HIP_TRY(hipStreamCreateWithFlags(&h->side[k], hipStreamNonBlocking));
    HIP_TRY(pipeline_malloc((void**)&h->d_ff_rec[k], wb));
    HIP_TRY(hipMemset(h->d_ff_rec[k], 0, wb));
    HIP_TRY(pipeline_malloc((void**)&h->d_ff_flag[k], (size_t)h->n));
    HIP_TRY(hipMemset(h->d_ff_flag[k], 0, (size_t)h->n));
    for (int ph = 0; ph + 1 < h->ff_phases; ++ph) HIP_TRY(pipeline_malloc((void**)&h->d_ff_list[k][ph], wb));
    HIP_TRY(pipeline_malloc((void**)&h->d_ff_count[k], 2 * sizeof(int)));
    HIP_TRY(hipMemset(h->d_ff_count[k], 0, 2 * sizeof(int)));
    if (h->ff_sort) {
      const size_t sb = sizeof(float4) * (size_t)TB_FF_REC_MAX * TB_FF_SORT_BLOCK * ((h->n + TB_FF_SORT_BLOCK - 1) / TB_FF_SORT_BLOCK);
      HIP_TRY(pipeline_malloc((void**)&h->d_ff_sorted[k], sb));
      HIP_TRY(hipMemset(h->d_ff_sorted[k], 0, sb));
    }
    HIP_TRY(hipEventCreateWithFlags(&h->ev_step[k], hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&h->ev_ff[k], hipEventDisableTiming));
  }
  if (int rc = alloc_pool(h)) return rc;
  HIP_TRY(hipDeviceSynchronize());
  return TB_OK;
}

int tb_set_pipeline(TbHandle* h, int enable) {
  if (!h) return fail(TB_E_INVAL, "tb_set_pipeline: null handle");
  DeviceGuard g(h->device);
  if (enable && !h->side[0]) {
    if (h->kind != TB_ENV_SWING) return fail(TB_E_UNSUPPORTED, "tb_set_pipeline: only SwingRacket-v0 has a fast-forward to overlap");
    // 8 slots x (records + up to two survivor lists) x 192 B per env: 4.6 KB per env, 77 GB at the cap (of 288)
    if (h->n > TB_PIPELINE_MAX_ENVS) return fail(TB_E_INVAL, "tb_set_pipeline: more than 2^24 envs (the parked-record slots would not fit next to the state)");
    h->last_slot = -1;
    if (int rc = alloc_pipeline(h)) {
      // all or nothing: a half-built pipeline would pass the `side[0]` test above on the next call and the step kernel would
      // then park into a null slot. fail() has already recorded what went wrong.
      release_pipeline(h);
      (void)hipGetLastError();
      return rc;
    }
  }
  h->pipeline = enable ? 1 : 0;
  return TB_OK;
}

int tb_diag_fail_alloc(int nth) {
  g_fail_alloc_countdown = nth > 0 ? nth : 0;
  return TB_OK;
}

int tb_pipeline_sync(TbHandle* h, int host_wait) {
  if (!h) return fail(TB_E_INVAL, "tb_pipeline_sync: null handle");
  DeviceGuard g(h->device);
  for (int k = 0; k < TB_FF_SLOTS; ++k) {
    if (host_wait && h->side[k]) HIP_TRY(hipStreamSynchronize(h->side[k]));
    h->ff_busy[k] = 0;
  }
  h->pool_ev_valid = 0; h->direct_ev_valid = 0;  // (an event recorded on one side of a capture boundary means nothing on the other)
  h->last_slot = -1;
  if (host_wait) {
    memset(h->ff_cap, 0, sizeof h->ff_cap);
    h->phase_at_capture = h->phase; h->phase_valid_at_capture = h->phase_valid;
  } else {  // the captured tb_step calls advanced the host's episode phase, but none of them ran: the replays will (tb_phase_advance)
    h->phase = h->phase_at_capture; h->phase_valid = h->phase_valid_at_capture;
  }
  return TB_OK;
}

int tb_pipeline_recover(TbHandle* h) {
  if (!h) return fail(TB_E_INVAL, "tb_pipeline_recover: null handle");
  DeviceGuard g(h->device);
  (void)hipGetLastError();  // the abandoned capture leaves a sticky hipErrorStreamCaptureInvalidated behind
  // the captured tb_step calls advanced the host's episode phase, but none of them ran
  h->phase = h->phase_at_capture; h->phase_valid = h->phase_valid_at_capture;
  h->pool_ev_valid = 0; h->direct_ev_valid = 0;
  // episode ends that the abandoned capture "parked" into the pool never ran: nothing is pending on their account
  h->pool_episodes = 0; h->pool_run_upto = 0; h->pool_pending = 0;
  for (int k = 0; k < TB_FF_SLOTS; ++k) {
    h->ff_busy[k] = 0;
    if (!h->side[k]) continue;
    // a side stream that was forked into the capture stays invalidated: replace it and its events
    (void)hipStreamDestroy(h->side[k]);
    (void)hipEventDestroy(h->ev_step[k]);
    (void)hipEventDestroy(h->ev_ff[k]);
    (void)hipGetLastError();
    HIP_TRY(hipStreamCreateWithFlags(&h->side[k], hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&h->ev_step[k], hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&h->ev_ff[k], hipEventDisableTiming));
  }
  h->last_slot = -1; h->last_term = nullptr; h->last_sub = nullptr;
  memset(h->ff_cap, 0, sizeof h->ff_cap);  // nothing of the abandoned capture will ever run
  return TB_OK;
}

int tb_flush(TbHandle* h, void* stream) {
  if (!h) return fail(TB_E_INVAL, "tb_flush: null handle");
  DeviceGuard g(h->device);
  return flush_all(h, (hipStream_t)stream);
}

int tb_phase(TbHandle* h) {
  if (!h) return fail(TB_E_INVAL, "tb_phase: null handle");
  return h->phase_valid ? h->phase : -1;
}

int tb_pipeline_form(TbHandle* h) {
  if (!h) return fail(TB_E_INVAL, "tb_pipeline_form: null handle");
  if (h->kind != TB_ENV_SWING || !h->pipeline) return 0;
  const int mode = defer_mode(h);
  return mode == 2 ? 3 : mode == 1 && h->ff_phases == 1 && !h->ff_sort ? 2 : 1;
}

int tb_phase_advance(TbHandle* h, int n_steps) {
  if (!h || n_steps < 0) return fail(TB_E_INVAL, "tb_phase_advance: bad argument");
  if (h->phase_valid) h->phase = (h->phase + n_steps) % 26;
  h->first_substeps += (unsigned long long)h->n * (unsigned long long)n_steps;  // the replayed steps' share of the substep counter
  return TB_OK;
}

int tb_mark_record(TbHandle* h, int k, void* stream) {
  if (!h || k < 0 || k >= TB_MAX_MARKS) return fail(TB_E_INVAL, "tb_mark_record: bad handle or mark index");
  DeviceGuard g(h->device);
  hipStream_t s = (hipStream_t)stream;
  if (!h->marks_on) return fail(TB_E_UNSUPPORTED, "tb_mark_record needs tb_mark_enable(h, 1) before the steps it covers (their fast-forwards must be counted)");
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  HIP_TRY(hipStreamIsCapturing(s, &st));
  if (int rc = run_pool_for_mark(h, s)) return rc;  // (counted among the fast-forwards enqueued before the mark)
  hipLaunchKernelGGL(tb_mark_kernel, dim3(1), dim3(1), 0, s, h->h_marks + k);
  HIP_TRY(hipGetLastError());
  h->mark_in_capture[k] = st == hipStreamCaptureStatusActive;
  for (int q = 0; q < TB_FF_SLOTS; ++q) h->mark_ff_before[k][q] = h->mark_in_capture[k] ? h->ff_cap[q] : h->ff_eager[q];
  return TB_OK;
}

int tb_mark_enable(TbHandle* h, int on) {
  if (!h) return fail(TB_E_INVAL, "tb_mark_enable: null handle");
  if (on) { if (int rc = ensure_marks(h)) return rc; }
  h->marks_on = on ? 1 : 0;
  return TB_OK;
}

int tb_mark_begin(TbHandle* h) {
  if (!h) return fail(TB_E_INVAL, "tb_mark_begin: null handle");
  if (int rc = ensure_marks(h)) return rc;
  for (int q = 0; q < TB_FF_SLOTS; ++q) h->snap_ff[q] = __atomic_load_n(h->h_marks + TB_MAX_MARKS + q, __ATOMIC_ACQUIRE);
  for (int k = 0; k < TB_MAX_MARKS; ++k) h->snap_marks[k] = __atomic_load_n(h->h_marks + k, __ATOMIC_ACQUIRE);
  memset(h->ff_eager, 0, sizeof h->ff_eager);
  return TB_OK;
}

long long tb_mark_count(TbHandle* h, int k) {
  if (!h || k < 0 || k >= TB_MAX_MARKS) return fail(TB_E_INVAL, "tb_mark_count: bad handle or mark index");
  if (!h->h_marks) return 0;
  return (long long)__atomic_load_n(h->h_marks + k, __ATOMIC_ACQUIRE);
}

int tb_mark_host_wait(TbHandle* h, int k, int timeout_ms) {
  if (!h || k < 0 || k >= TB_MAX_MARKS || !h->h_marks) return fail(TB_E_INVAL, "tb_mark_host_wait: bad handle, mark index, or no mark recorded yet");
  // fired once more than at tb_mark_begin, and every fast-forward enqueued before the mark has finished: those of the
  // graph that holds it (counted at capture time) plus whatever was launched eagerly since (over-waiting at worst)
  unsigned long long ff_target[TB_FF_SLOTS];
  for (int q = 0; q < TB_FF_SLOTS; ++q)
    ff_target[q] = h->snap_ff[q] + (unsigned long long)(h->mark_in_capture[k] ? h->mark_ff_before[k][q] + h->ff_eager[q] : h->mark_ff_before[k][q]);
  const unsigned long long count = h->snap_marks[k] + 1;
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 0;; ++spins) {
    bool ok = __atomic_load_n(h->h_marks + k, __ATOMIC_ACQUIRE) >= count;
    for (int q = 0; ok && q < TB_FF_SLOTS; ++q) ok = __atomic_load_n(h->h_marks + TB_MAX_MARKS + q, __ATOMIC_ACQUIRE) >= ff_target[q];
    if (ok) return TB_OK;
    if ((spins & 1023u) == 1023u &&
        std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() > timeout_ms)
      return fail(TB_E_TIMEOUT, "tb_mark_host_wait: the mark did not fire in time");
    __builtin_ia32_pause();
  }
}

int tb_set_params(TbHandle* h, const TbParams* params, void* stream) {
  if (!h || !params) return fail(TB_E_INVAL, "tb_set_params: null argument");
  if (int rc = validate_params(params)) return rc;
  DeviceGuard g(h->device);
  hipStream_t s = (hipStream_t)stream;
  // the staging buffer may still feed an earlier async copy on another stream: settle it first
  if (int rc = flush_all(h, s)) return rc;
  HIP_TRY(hipStreamSynchronize(s));
  if ((params->flags ^ h->params.flags) & TB_F_AUTO_RESET) h->phase_valid = 0;  // episodes may stop / start restarting
  h->params = *params;
  to_kparams(params, &h->kp, &h->cull_planes[0][0]);
  if (int rc = upload_hull(h, s)) return rc;
  HIP_TRY(hipStreamSynchronize(s));
  if (h->side[0]) {  // a pipelined handle whose new parameter block asks for the pool (racket<->court contact above 16384 envs)
    if (int rc = alloc_pool(h)) return rc;
    HIP_TRY(hipDeviceSynchronize());
  }
  h->params_generation++;
  return TB_OK;
}

int tb_params_generation(TbHandle* h) {
  if (!h) return fail(TB_E_INVAL, "tb_params_generation: null handle");
  return h->params_generation;
}

int tb_set_racket_scale(TbHandle* h, float scale, void* stream) {
  if (!h) return fail(TB_E_INVAL, "tb_set_racket_scale: null handle");
  if (!(scale > 0.0f)) return fail(TB_E_PARAMS, "tb_set_racket_scale: scale must be positive");
  DeviceGuard g(h->device);
  h->params.racket_scale = scale; h->kp.racket_scale = scale;  // what a later tb_set_params re-uploads
  float* dst = reinterpret_cast<float*>(h->d_hull + TB_HULL_KP) + offsetof(KParams, racket_scale) / sizeof(float);
  hipLaunchKernelGGL(tb_poke_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, dst, scale);
  HIP_TRY(hipGetLastError());
  return TB_OK;
}

int tb_reset(TbHandle* h, const uint8_t* mask_dev, float* obs_dev, void* stream) {
  if (!h) return fail(TB_E_INVAL, "tb_reset: null handle");
  DeviceGuard g(h->device);
  if (int rc = flush_all(h, (hipStream_t)stream)) return rc;
  if (mask_dev) h->phase_valid = 0;  // episodes are no longer in lockstep
  else { h->phase_valid = 1; h->phase = 0; }
  KArgs a = base_args(h);
  a.mask = mask_dev; a.obs = obs_dev;
  dim3 grid((unsigned)((h->n + 255) / 256)), block(256);
  if (h->kind == TB_ENV_SWING) hipLaunchKernelGGL(tb_reset_kernel<TB_ENV_SWING>, grid, block, 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(tb_reset_kernel<TB_ENV_TENNIS>, grid, block, 0, (hipStream_t)stream, a);
  HIP_TRY(hipGetLastError());
  return TB_OK;
}

int tb_step(TbHandle* h, const float* actions_dev, float* obs_dev, float* reward_dev, uint8_t* done_dev, float* terminal_obs_dev,
            int32_t* substeps_dev, void* stream) {
  if (!h || !actions_dev || !obs_dev || !reward_dev || !done_dev) return fail(TB_E_INVAL, "tb_step: null argument");
  DeviceGuard g(h->device);
  return launch_step(h, 1, actions_dev, obs_dev, reward_dev, done_dev, terminal_obs_dev, substeps_dev, (hipStream_t)stream);
}

int tb_step_sequence(TbHandle* h, int n_steps, const float* actions_dev, float* obs_dev, float* reward_dev, uint8_t* done_dev,
                     size_t actions_stride, size_t obs_stride, size_t reward_stride, size_t done_stride, void* stream) {
  if (!h || !actions_dev || !obs_dev || !reward_dev || !done_dev) return fail(TB_E_INVAL, "tb_step_sequence: null argument");
  if (n_steps < 1) return fail(TB_E_INVAL, "tb_step_sequence: n_steps must be >= 1");
  DeviceGuard g(h->device);
  for (int t = 0; t < n_steps; ++t) {
    const size_t k = (size_t)t;
    if (int rc = launch_step(h, 1, reinterpret_cast<const float*>(reinterpret_cast<const char*>(actions_dev) + k * actions_stride),
                             reinterpret_cast<float*>(reinterpret_cast<char*>(obs_dev) + k * obs_stride),
                             reinterpret_cast<float*>(reinterpret_cast<char*>(reward_dev) + k * reward_stride), done_dev + k * done_stride, nullptr, nullptr,
                             (hipStream_t)stream))
      return rc;
  }
  return TB_OK;
}

int tb_policy_floats(int env_kind) {
  return env_kind == TB_ENV_SWING ? policy_floats<TB_ENV_SWING>() : env_kind == TB_ENV_TENNIS ? policy_floats<TB_ENV_TENNIS>() : TB_E_INVAL;
}

int tb_policy_step(TbHandle* h, const float* weights_dev, const float* obs_in_dev, float* actions_dev, float* raw_actions_dev, float* logp_dev,
                   float* value_dev, float* obs_dev, float* reward_dev, uint8_t* done_dev, uint64_t noise_seed, int deterministic, void* stream) {
  if (!h || !weights_dev || !obs_in_dev || !actions_dev || !raw_actions_dev || !logp_dev || !value_dev || !obs_dev || !reward_dev || !done_dev)
    return fail(TB_E_INVAL, "tb_policy_step: null argument");
  DeviceGuard g(h->device);
  PolicyIO pol = {weights_dev, obs_in_dev, actions_dev, raw_actions_dev, logp_dev, value_dev, noise_seed, deterministic};
  return launch_step(h, 1, nullptr, obs_dev, reward_dev, done_dev, nullptr, nullptr, (hipStream_t)stream, &pol);
}

int tb_policy_rollout(TbHandle* h, int n_steps, const float* weights_dev, const float* obs_in_dev, float* actions_dev, float* raw_actions_dev,
                      float* logp_dev, float* value_dev, float* obs_dev, float* reward_dev, uint8_t* done_dev, const size_t* step_strides_bytes,
                      uint64_t noise_seed, int deterministic, void* stream) {
  if (!h || !weights_dev || !obs_in_dev || !actions_dev || !raw_actions_dev || !logp_dev || !value_dev || !obs_dev || !reward_dev || !done_dev)
    return fail(TB_E_INVAL, "tb_policy_rollout: null argument");
  if (n_steps < 1) return fail(TB_E_INVAL, "tb_policy_rollout: n_steps must be >= 1");
  if (!(h->kp.flags & TB_F_AUTO_RESET)) return fail(TB_E_UNSUPPORTED, "tb_policy_rollout needs TB_F_AUTO_RESET (episodes must restart inside the launch)");
  const bool swing = h->kind == TB_ENV_SWING;
  if (swing && !(h->pipeline && h->phase_valid))
    return fail(TB_E_UNSUPPORTED, "tb_policy_rollout on SwingRacket-v0 needs tb_set_pipeline(h, 1) and episodes in lockstep (every env reset together): "
                                  "the fast-forward that ends an episode cannot run inside a multi-step launch");
  const size_t n = (size_t)h->n, A = swing ? TB_SWING_ACT_DIM : TB_TENNIS_ACT_DIM, O = swing ? TB_SWING_OBS_DIM : TB_TENNIS_OBS_DIM;
  size_t st[7] = {n * A, n * A, n, n, n * O, n, n};  // elements per step: actions, raw, logp, value, obs, reward, done
  if (step_strides_bytes) {
    for (int k = 0; k < 7; ++k) {
      const size_t el = k == 6 ? 1 : sizeof(float);
      if (step_strides_bytes[k] % el) return fail(TB_E_INVAL, "tb_policy_rollout: a step stride is not a multiple of its element size");
      if (step_strides_bytes[k]) st[k] = step_strides_bytes[k] / el;
    }
  }
  // action rows are written 8 bytes at a time (like the action rows tb_step reads): bases and step strides must keep them aligned
  if ((reinterpret_cast<uintptr_t>(actions_dev) | reinterpret_cast<uintptr_t>(raw_actions_dev)) % 8 || (st[0] * sizeof(float)) % 8 || (st[1] * sizeof(float)) % 8)
    return fail(TB_E_INVAL, "tb_policy_rollout: actions / raw_actions and their step strides must be 8-byte aligned");
  DeviceGuard g(h->device);
  hipStream_t s = (hipStream_t)stream;
  const float* obs_in = obs_in_dev;
  for (int t = 0; t < n_steps;) {
    int chunk = n_steps - t;
    if (swing) { const int room = 26 - h->phase; chunk = chunk < room ? chunk : room; }
    PolicyIO pol = {weights_dev, obs_in, actions_dev + (size_t)t * st[0], raw_actions_dev + (size_t)t * st[1], logp_dev + (size_t)t * st[2],
                    value_dev + (size_t)t * st[3], noise_seed, deterministic};
    if (int rc = launch_policy_rollout(h, chunk, pol, obs_dev + (size_t)t * st[4], reward_dev + (size_t)t * st[5], done_dev + (size_t)t * st[6], st, s)) return rc;
    t += chunk;
    obs_in = obs_dev + (size_t)(t - 1) * st[4];  // the next launch acts on what this one observed last
  }
  return TB_OK;
}

int tb_rollout(TbHandle* h, int n_steps, const float* actions_dev, float* obs_dev, float* reward_dev, uint8_t* done_dev,
               int32_t* substeps_total_dev, void* stream) {
  if (!h || !actions_dev || !obs_dev || !reward_dev || !done_dev) return fail(TB_E_INVAL, "tb_rollout: null argument");
  if (n_steps < 1) return fail(TB_E_INVAL, "tb_rollout: n_steps must be >= 1");
  if (!(h->kp.flags & TB_F_AUTO_RESET)) return fail(TB_E_UNSUPPORTED, "tb_rollout needs TB_F_AUTO_RESET (episodes must restart inside the launch)");
  DeviceGuard g(h->device);
  hipStream_t s = (hipStream_t)stream;
  if (h->pipeline && h->kind == TB_ENV_SWING && h->phase_valid && !substeps_total_dev) {
    // pipelined: launches that end where the episodes end (<= 26 steps each), every one followed by its
    // fast-forward on a side stream instead of stalling its waves on it
    const size_t n = (size_t)h->n;
    for (int t = 0; t < n_steps;) {
      const int room = 26 - h->phase, chunk = n_steps - t < room ? n_steps - t : room;
      if (int rc = launch_step(h, chunk, actions_dev + (size_t)t * n * TB_SWING_ACT_DIM, obs_dev + (size_t)t * n * TB_SWING_OBS_DIM, reward_dev + (size_t)t * n,
                               done_dev + (size_t)t * n, nullptr, nullptr, s, nullptr, chunk > 1))
        return rc;
      t += chunk;
    }
    return TB_OK;
  }
  return launch_step(h, n_steps, actions_dev, obs_dev, reward_dev, done_dev, nullptr, substeps_total_dev, s);
}

int tb_get_state(TbHandle* h, uint32_t* words, uint8_t* done, int on_device, void* stream) {
  if (!h || !words) return fail(TB_E_INVAL, "tb_get_state: null argument");
  DeviceGuard g(h->device);
  hipStream_t s = (hipStream_t)stream;
  if (int rc = flush_all(h, s)) return rc;
  const size_t wb = sizeof(uint32_t) * (size_t)words_of(h->kind) * h->n;
  hipMemcpyKind k = on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
  HIP_TRY(hipMemcpyAsync(words, h->d_words, wb, k, s));
  if (done) HIP_TRY(hipMemcpyAsync(done, h->d_done, (size_t)h->n, k, s));
  if (!on_device) HIP_TRY(hipStreamSynchronize(s));
  return TB_OK;
}

int tb_set_state(TbHandle* h, const uint32_t* words, const uint8_t* done, int on_device, void* stream) {
  if (!h || !words) return fail(TB_E_INVAL, "tb_set_state: null argument");
  DeviceGuard g(h->device);
  hipStream_t s = (hipStream_t)stream;
  if (int rc = flush_all(h, s)) return rc;
  h->phase_valid = 0;  // injected states need not be in lockstep
  const size_t wb = sizeof(uint32_t) * (size_t)words_of(h->kind) * h->n;
  hipMemcpyKind k = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  HIP_TRY(hipMemcpyAsync(h->d_words, words, wb, k, s));
  if (done) HIP_TRY(hipMemcpyAsync(h->d_done, done, (size_t)h->n, k, s));
  else HIP_TRY(hipMemsetAsync(h->d_done, 0, (size_t)h->n, s));
  HIP_TRY(hipMemsetAsync(h->d_mflag, 0, (size_t)h->n, s));  // the racket<->court contact caches are not part of the state words
  if (h->kind == TB_ENV_SWING && h->pipeline) {
    // the pipelined kernels need to know which launch ends the episodes: the injected envs are in lockstep again
    // when every one is running (done = 0) at the same step count s < 26 -- then the phase is s (e.g. a checkpoint
    // of a training run restored into a fresh handle). Costs one small device-to-host copy and a stream
    // synchronisation (so: not capturable, and not asynchronous even with on_device); only paid with the pipeline on,
    // the one mode that uses the phase -- without it the call stays fully asynchronous for on_device buffers.
    const size_t n = (size_t)h->n;
    uint32_t* steps = (uint32_t*)malloc(n * sizeof(uint32_t));
    uint8_t* dn = (uint8_t*)malloc(n);
    if (!steps || !dn) { free(steps); free(dn); return fail(TB_E_INVAL, "tb_set_state: out of host memory"); }
    hipError_t e1 = hipMemcpyAsync(steps, h->d_words + (size_t)TB_W_SW_STEP * n, n * sizeof(uint32_t), hipMemcpyDeviceToHost, s);
    hipError_t e2 = e1 == hipSuccess ? hipMemcpyAsync(dn, h->d_done, n, hipMemcpyDeviceToHost, s) : e1;
    hipError_t e3 = e2 == hipSuccess ? hipStreamSynchronize(s) : e2;
    if (e3 == hipSuccess) {
      bool same = (int32_t)steps[0] >= 0 && (int32_t)steps[0] < 26;
      for (size_t i = 0; same && i < n; ++i) same = steps[i] == steps[0] && dn[i] == TB_DONE_NO;
      if (same) { h->phase_valid = 1; h->phase = (int)steps[0]; }
    }
    free(steps); free(dn);
    if (e3 != hipSuccess) return fail((int)e3, "tb_set_state: reading back the step counters");
  }
  if (!on_device) HIP_TRY(hipStreamSynchronize(s));
  return TB_OK;
}

int tb_counters(TbHandle* h, uint64_t* out, void* stream) {
  if (!h || !out) return fail(TB_E_INVAL, "tb_counters: null argument");
  DeviceGuard g(h->device);
  if (int rc = flush_all(h, (hipStream_t)stream)) return rc;
  static_assert(sizeof(uint64_t) == sizeof(unsigned long long), "counter width");
  uint64_t shards[TB_COUNTER_SHARDS][TB_N_COUNTERS];
  HIP_TRY(hipMemcpyAsync(shards, h->d_counters, sizeof shards, hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  for (int k = 0; k < TB_N_COUNTERS; ++k) {
    out[k] = 0;
    for (int sh = 0; sh < TB_COUNTER_SHARDS; ++sh) out[k] += shards[sh][k];
  }
  out[6] += h->first_substeps;
  return TB_OK;
}

int tb_diag_stream_copy(const uint32_t* src_dev, uint32_t* dst_dev, int n, int rows, int device, void* stream) {
  if (!src_dev || !dst_dev || n <= 0 || rows <= 0) return fail(TB_E_INVAL, "tb_diag_stream_copy: bad argument");
  DeviceGuard g(device);
  if (g.err != hipSuccess) return fail((int)g.err, "hipSetDevice");
  hipLaunchKernelGGL(tb_diag_copy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src_dev, dst_dev, n, rows);
  HIP_TRY(hipGetLastError());
  return TB_OK;
}

int tb_diag_idle(int waves, int microseconds, int device, void* stream) {
  if (waves <= 0 || microseconds <= 0) return fail(TB_E_INVAL, "tb_diag_idle: bad argument");
  DeviceGuard g(device);
  if (g.err != hipSuccess) return fail((int)g.err, "hipSetDevice");
  hipLaunchKernelGGL(tb_diag_idle_kernel, dim3((unsigned)waves), dim3(64), 0, (hipStream_t)stream, (unsigned long long)microseconds * 100ull);
  HIP_TRY(hipGetLastError());
  return TB_OK;
}

// (diagnostic builds only: tb_diag_read_stamps / tb_diag_read_lanes)
#define TB_DIAG_HOST_SECTION
#include "tb_diag.hpp"
#undef TB_DIAG_HOST_SECTION

int tb_counters_reset(TbHandle* h, void* stream) {
  if (!h) return fail(TB_E_INVAL, "tb_counters_reset: null handle");
  DeviceGuard g(h->device);
  HIP_TRY(hipMemsetAsync(h->d_counters, 0, sizeof(uint64_t) * TB_N_COUNTERS * TB_COUNTER_SHARDS, (hipStream_t)stream));
  h->first_substeps = 0;
  return TB_OK;
}

}  // extern "C"

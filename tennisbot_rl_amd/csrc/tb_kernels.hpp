// tb_kernels.hpp -- the HIP kernels of libtb_stepper.so (gfx950 / MI355X): kernel arguments, state rows, reset, the env logic of both
// gym envs around tb_device.hpp's substep, and the __global__ entry points (step / rollout, fused policy rollout, fast-forward with its
// sort, reset, init, marks, diagnostics). Included once, by tb_stepper.hip, which holds the host side (handle, launches, C ABI).
// Everything here lives in an anonymous namespace of that one translation unit.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

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
  unsigned long long* ff_sealed;  // the pool's sealed-fate exit (fate_sealed): where the substeps it did not run are counted; null = exit off
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

// an empty asm that reads every state register: whatever loaded them has landed behind it (the compiler places the waits)
TB_DEV void landed_env(const EnvRegs& e) {
  asm volatile("" :: "v"(e.r.p.x), "v"(e.r.p.y), "v"(e.r.p.z), "v"(e.r.q.x), "v"(e.r.q.y), "v"(e.r.q.z), "v"(e.r.q.w), "v"(e.r.v.x), "v"(e.r.v.y), "v"(e.r.v.z),
               "v"(e.r.w.x), "v"(e.r.w.y), "v"(e.r.w.z));
  asm volatile("" :: "v"(e.b.p.x), "v"(e.b.p.y), "v"(e.b.p.z), "v"(e.b.v.x), "v"(e.b.v.y), "v"(e.b.v.z), "v"(e.b.w.x), "v"(e.b.w.y), "v"(e.b.w.z));
  asm volatile("" :: "v"(e.aux[0]), "v"(e.aux[1]), "v"(e.aux[2]), "v"(e.aux[3]), "v"(e.aux[4]), "v"(e.aux[5]), "v"(e.step_count), "v"(e.episode), "v"(e.done));
}

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

// THE SEALED FATE of a fast-forward (the pool's instantiations only, TbOptions.ff_seal). Under a trained policy a struck ball often
// leaves the court: it then falls until the 800-substep limit ends the episode with reward 0 (swingracket_env.py:127-128), and
// with auto-reset nothing else of that flight is ever read -- not the state (the env restarts), not an observation (the pool
// writes none), only the reward, the done flag and the substep count. A wave pays for its longest lane, so a quarter of the pool's
// waves ran ~400 substeps for one such ball. This test names the flights whose outcome CANNOT be anything but the timeout; the
// caller then books the remaining substeps (counters[6], the substeps output and step_count are what the full flight gives) and
// leaves. It claims, from this substep on:
//   (1) the ball is below every static shape -- all of them are centred on z = 0 and reach at most static_top from it -- and is
//       falling: gravity and the drag -v kd (kd dt < 1: a velocity component keeps its sign) leave v.z < 0 and the horizontal
//       velocity components their signs for as long as nothing touches the ball (no Magnus force: magnus_k == 0 is required);
//   (2) the racket cannot touch it either: on one horizontal axis the ball is beyond the racket's reach and moving away (or
//       still), and the racket's centre on that axis is a damped oscillator about its anchor (force -k xi of :135-141, drag
//       -v kd with kd >= 0 varying) whose amplitude stays under 1.25 sqrt(xi^2 + v^2 / w^2) + 0.05 m. (The discrete
//       recurrence v' = v (1 - c) - w^2 dt xi, xi' = xi + dt v' with an adversary choosing c in [0, 0.2] every substep stays under
//       1.06 of that root for w dt <= 0.2; the host checks both limits on the parameters: seal_params_ok. The racket's own
//       contact with the court is the extended contact set, whose instantiations do not have this exit.)
//   So no contact ever happens again: (1) and (2) hold at the next substep by the same argument, the termination tests of
//   :111-123 stay false and :127 fires at step_count 801. Rounding is no part of the argument: every margin above is orders of
//   magnitude beyond an ulp. Checked every 8th substep, by waves with a ball under the court only.
// The oracle does not have this exit: every pool-form parity test compares against the flight run to its end.
TB_DEV bool state_is_finite(const EnvRegs& e);
TB_DEV bool fate_sealed(const KParams& P, const EnvRegs& e) {
  if (!(e.b.v.z < 0.0f) || !(dot(e.b.v, e.b.v) < 1.0e6f) || !(dot(e.r.v, e.r.v) < 1.0e6f) || !state_is_finite(e)) return false;
  if (!(P.flags & TB_F_RACKET_BALL)) return true;
  const float reach = (((P.hull_bound_radius + P.hull_margin) + P.ball_radius) + P.contact_threshold) + 0.01f;
  const float xi = e.r.p.x - e.aux[2], yi = e.r.p.y - e.aux[3];
  const float bx = 1.25f * sqrtf(xi * xi + (e.r.v.x * e.r.v.x) / (50.0f * P.racket_inv_mass)) + 0.05f;
  const float by = 1.25f * sqrtf(yi * yi + (e.r.v.y * e.r.v.y) / (2.0f * P.racket_inv_mass)) + 0.05f;
  const float dx = e.b.p.x - e.aux[2], dy = e.b.p.y - e.aux[3];
  const bool away_x = (e.b.v.x >= 0.0f && dx - bx > reach) || (e.b.v.x <= 0.0f && -dx - bx > reach);
  const bool away_y = (e.b.v.y >= 0.0f && dy - by > reach) || (e.b.v.y <= 0.0f && -dy - by > reach);
  return away_x || away_y;
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
// SEAL    = tb_ff_kernel's pool instantiations: the sealed-fate exit (fate_sealed); `sealed` counts the substeps it books without running
template <unsigned FORM, bool BUDGET = false, bool SEAL = false>
TB_DEV float swing_loop(const KParams& P, const float4* hull, EnvRegs& e, Manifold& M, vec3 F, vec3 T, bool in_ff, bool defer, bool& parked,
                        int& ns, uint32_t* cnt TB_STAMP_ARG, int budget = 0, const float4* table_mem = nullptr, bool seal = false,
                        uint32_t* sealed = nullptr) {
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
      if constexpr (SEAL) {
        if (seal && (e.step_count & 7) == 0) {
          const bool under = ((e.b.p.z + P.ball_radius) + P.contact_threshold) < -P.static_top - 1.0e-3f;
          if (__any(under)) {
            if (under && fate_sealed(P, e)) {  // nothing but the timeout of :127-128 can end this flight: book it
              const int left = 801 - e.step_count;
              ns += left; *sealed += (uint32_t)left;
              e.step_count = 801; cnt[3]++; e.done = TB_DONE_PENDING_FORCE;
              break;
            }
          }
        }
      }
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
    // MULTI: every state load has landed before the loop is entered. Left alone, the loop's first use of each state value carries a
    // wait that the compiler must place for the first trip -- and that, in every later trip, waits for the previous step's STORES
    // and for the action prefetch below (the memory counter is one, and in order): a full round trip per step with nothing to hide it.
    if constexpr (MULTI && !POLICY) landed_env(e);
    for (int t = 0; t < n_steps; ++t) {
      const size_t row = (size_t)t * A.n + i;
      // MULTI: the actions of step t + 1 are requested before step t is computed -- nothing else of a step waits for memory (the state
      // is in registers), so a load at the top of each step was a whole memory round trip per step that nothing hid
      // (tb_rollout at 4096 envs, same box: SwingRacket 2.0 -> 3.x G env steps/s, see profiles/r04_rollout_rate.json)
      float a_next[NA];
      if constexpr (MULTI && !POLICY) {
        if (t + 1 < n_steps) load_actions<KIND>(A.actions, row + (size_t)A.n, a_next);
      }
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
      if constexpr (MULTI && !POLICY) {
        if (t + 1 < n_steps) {
#pragma unroll
          for (int k = 0; k < NA; ++k) a[k] = a_next[k];
        }
      }
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
#ifdef TB_DIAG_NO_TOWERS  // timing-only (tools/diag/r04_policy_ablate.py): RESULTS ARE WRONG
      out[0] = x0[0] * 0.1f; out[1] = x0[0] * 0.2f; out[2] = x0[0] * 0.3f; out[3] = x0[0] * 0.4f;
#else
      regs.apply(x0, out);
#endif
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
  // racket narrowphase can share the outline sweeps of the asking envs among all 64 lanes (outline_sweep_rows).
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
      // (the shared outline sweep, four asking lanes at a time: outline_sweep_rows. PPO collect under the trained policy, same box, 16 envs per
      //  env wave, 4096 envs: each lane sweeping for itself 570-574, one query at a time over 64 lanes 592-595 -> 622, four at a time 644 M env
      //  steps/s; 48 envs per wave, 16384 envs: for itself 1036 -> 1057 M -- there the one-query form had lost, 461 -> 426 M at 4096 envs)
      //  (not with 48 envs per wave AND the extended contact set: at that instantiation's 256-VGPR limit the shared sweep's edge records spill)
      constexpr unsigned FORM = (RG ? SF_RG : 0u) | SF_COLD | (S == 1 || !RG ? SF_WIDE : 0u);
#ifdef TB_DIAG_NO_ENVSTEP  // timing-only (tools/diag/r04_policy_ablate.py): RESULTS ARE WRONG
      rew = a[0]; e.step_count += 1;
      if (KIND == TB_ENV_TENNIS) make_obs<KIND>(e, o);
#else
      if (KIND == TB_ENV_SWING) rew = swing_step<FORM>(Pl, s_hull, e, M, a, ns, cnt, true, parked TB_STAMP_PASS);  // never loops in here: see tb_step_kernel<LEAN>
      else rew = tennis_step<FORM | SF_REGROWS>(Pl, s_hull, e, M, a, o, d, cnt TB_STAMP_PASS);
#endif
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
  TB_DIAG_STAMPS_END(st);
  TB_DIAG_ADD_LANE0(9, 1);
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
  uint32_t n_sealed = 0u;
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
      constexpr bool SEAL = POOL && !RG;
      const bool seal = SEAL && A.ff_sealed != nullptr && !A.term_obs;
      float rew = swing_loop<FORM, true, SEAL>(A.P, s_hull, e, M, F0, zero, true, false, unfinished, ns, cnt TB_STAMP_PASS, budget, nullptr, seal, &n_sealed);
      if constexpr (POOL) {
        // a full pool: finish here after all (the counter is read, not reserved: see the slack above)
        if (unfinished && A.ff_next && *reinterpret_cast<volatile int*>(A.ff_next_count) >= A.ff_cap) {
          unfinished = false;
          rew = swing_loop<FORM, true, SEAL>(A.P, s_hull, e, M, restoring_force(e), zero, true, false, unfinished, ns, cnt TB_STAMP_PASS, 0x7fffffff, nullptr, seal, &n_sealed);
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
  if constexpr (POOL && !RG) {
    if (__ballot(n_sealed != 0u) != 0ull) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) n_sealed += __shfl_xor(n_sealed, off, 64);
      if (lane == 0) atomicAdd(A.ff_sealed, (unsigned long long)n_sealed);
    }
  }
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


}  // namespace

// Policy inference for the fused policy + step kernels (tb_policy_step, tb_policy_rollout). Device code only; included by
// tb_kernels.hpp after KArgs / EnvRegs / Dims / philox4x32 are defined. This file is part of the HIP
// library's single translation unit (everything lives in one anonymous namespace there).
#pragma once

namespace {

// Policy inference fused into the step kernel: SB3's MlpPolicy with separate pi / vf
// towers as the reference configures it -- SwingRacket 6 -> 32 -> 64 -> 32 (train_swing.py:80-82),
// Tennisbot 12 -> 64 -> 64 (SB3 default, train.py:104-110) -- tanh hidden layers, linear action mean,
// state-independent log_std, a = mean + std * eps.
//
// The towers run on the matrix cores in fp32 (v_mfma_f32_16x16x4_f32: ONE VGPR per operand fragment instead of a
// broadcast weight per FMA -- a first VALU version spent its time re-reading weights out of LDS). Each layer is computed
// TRANSPOSED, H^T[out][env] = W^T[out][k] * X^T[k][env], 16 envs per wave: the weight fragment is the A operand
// (lane l: out l % 16, k-slot l / 16), the activations the B operand (lane l: k-slot l / 16, env l % 16), and -- the point of
// the transposition -- the C/D layout of one layer's output (lane l, register r: row 4 (l / 16) + r of the 16-row tile,
// env l % 16) IS the B layout of the next layer's input if the k-chunks are taken in that order: register r of output tile
// t feeds "chunk (t, r)" = k in {16 t + 4 g + r : g = 0..3}, lane group g holding its own k. pack_policy() permutes the
// weights to match, so activations never leave the registers: no LDS, no shuffles, no barrier between layers. The k-sum
// order is that permutation (a fixed order; vs. torch within 2e-5, tests/test_gpu_policy.py).
//
// Round 3: 16-env slices with the 16x16x4 shape replace 32-env halves with 32x32x2. The matrix pipe of a SIMD issues one
// MFMA at a time -- a tower for 32 envs was 83 x 64 = 5312 pipe cycles per step whatever the schedule, 67 of the 83 on
// one dependent chain -- and at 4096 envs only 64 of the 256 CUs had a workgroup at all. A 16-env slice is 76 x 32 = 2432
// pipe cycles in chains of at most 16 with 2-4 independent tiles side by side, half the tanh evaluations per lane, and a
// workgroup per 16 envs puts a tower on every CU (tb_policy_rollout_kernel<KIND, 1>).
// Blob, per tower and layer: bias tiles [ceil(out/16)][4 lane groups][4 regs], then weight fragments
// [ceil(out/16)][chunks][64 lanes] -- i.e. exactly what lane l loads at index l; heads padded to 16 outputs,
// the first layer's k padded to a multiple of 4 with zeros.
template <int KIND> struct PolicyNet;
template <> struct PolicyNet<TB_ENV_SWING> { static constexpr int NH = 3, H0 = 32, H1 = 64, H2 = 32, LAST = 32; };
template <> struct PolicyNet<TB_ENV_TENNIS> { static constexpr int NH = 2, H0 = 64, H1 = 64, H2 = 64, LAST = 64; };
constexpr int layer_floats(int in, int out) { return ((out + 15) / 16) * (16 + ((in + 3) / 4) * 64); }
template <int KIND> constexpr int tower_floats() {  // hidden layers + the (padded) head
  using N = PolicyNet<KIND>;
  return layer_floats(Dims<KIND>::O, N::H0) + layer_floats(N::H0, N::H1) + (N::NH == 3 ? layer_floats(N::H1, N::H2) : 0) + layer_floats(N::LAST, 16);
}
template <int KIND> constexpr int policy_floats() { return 2 * tower_floats<KIND>() + (Dims<KIND>::A + 3) / 4 * 4; }
constexpr int TB_POLICY_SLICE = 16;  // envs per tower wave

typedef float f32x4 __attribute__((ext_vector_type(4)));
// tanh(x) = 1 - 2 / (e^(2x) + 1) on the hardware exp2 / rcp units (1 ulp each): absolute error < 3e-7,
// saturates correctly at +-inf; 5 instructions instead of libm's ~40
TB_DEV float fast_tanh(float x) {
  float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
  return FMA(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
}

// one layer's operands for this lane: NT bias tiles (4 floats each) and NT * NC weight fragments
template <int NT, int NC>
struct LayerRegs {
  f32x4 bias[NT];
  float frag[NT * NC];
  TB_DEV void load(const float* g, int lane) {
    const int grp = lane >> 4;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float4 v = *reinterpret_cast<const float4*>(g + t * 16 + grp * 4);
      bias[t][0] = v.x; bias[t][1] = v.y; bias[t][2] = v.z; bias[t][3] = v.w;
    }
    g += NT * 16;
#pragma unroll
    for (int f = 0; f < NT * NC; ++f) frag[f] = g[f * 64 + lane];
  }
  // y[4 t + r] = act(bias + sum over chunks): the next layer's B operands, in place. Chunk-outer, tile-inner: the NT
  // accumulator chains are independent and written side by side, so the matrix pipe always has a ready MFMA
  template <bool TANH>
  TB_DEV void apply(const float (&x)[NC], float (&y)[NT * 4]) const {
    f32x4 c[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) c[t] = bias[t];
#pragma unroll
    for (int ch = 0; ch < NC; ++ch) {
#pragma unroll
      for (int t = 0; t < NT; ++t) c[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(frag[t * NC + ch], x[ch], c[t], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) y[t * 4 + r] = TANH ? fast_tanh(c[t][r]) : c[t][r];
    }
  }
};

// one tower for 16 envs: lane l works on env (l & 15); out[0..3] = head rows 4 (l >> 4) + 0..3 of that env. Every operand
// of the tower is requested by load() up front (one VGPR per fragment): the loads of the later layers land while the
// earlier ones compute (policy_tower), or stay resident across the steps of a rollout launch (tb_policy_rollout_kernel).
template <int KIND> struct TowerRegs;
template <> struct TowerRegs<TB_ENV_SWING> {  // 6 -> 32 -> 64 -> 32 -> head
  using N = PolicyNet<TB_ENV_SWING>;
  static constexpr int O = Dims<TB_ENV_SWING>::O, NC0 = (O + 3) / 4;
  LayerRegs<2, NC0> l0;
  LayerRegs<4, 8> l1;
  LayerRegs<2, 16> l2;
  LayerRegs<1, 8> lh;
  TB_DEV void load(const float* g, int lane) {
    l0.load(g, lane); g += layer_floats(O, N::H0);
    l1.load(g, lane); g += layer_floats(N::H0, N::H1);
    l2.load(g, lane); g += layer_floats(N::H1, N::H2);
    lh.load(g, lane);
  }
  TB_DEV void apply(const float (&x0)[NC0], float (&out)[4]) const {
    float h0[8], h1[16], h2[8];
    l0.template apply<true>(x0, h0);
    l1.template apply<true>(h0, h1);
    l2.template apply<true>(h1, h2);
    lh.template apply<false>(h2, out);
  }
};
template <> struct TowerRegs<TB_ENV_TENNIS> {  // 12 -> 64 -> 64 -> head
  using N = PolicyNet<TB_ENV_TENNIS>;
  static constexpr int O = Dims<TB_ENV_TENNIS>::O, NC0 = (O + 3) / 4;
  LayerRegs<4, NC0> l0;
  LayerRegs<4, 16> l1;
  LayerRegs<1, 16> lh;
  TB_DEV void load(const float* g, int lane) {
    l0.load(g, lane); g += layer_floats(O, N::H0);
    l1.load(g, lane); g += layer_floats(N::H0, N::H1);
    lh.load(g, lane);
  }
  TB_DEV void apply(const float (&x0)[NC0], float (&out)[4]) const {
    float h0[16], h1[16];
    l0.template apply<true>(x0, h0);
    l1.template apply<true>(h0, h1);
    lh.template apply<false>(h1, out);
  }
};

// the first layer's B operand of this lane: obs[4 ch + (lane >> 4)] of its env, 0 beyond the observation
template <int KIND>
TB_DEV void policy_inputs(const float* obs_row, int lane, float (&x0)[TowerRegs<KIND>::NC0]) {
  constexpr int NO = Dims<KIND>::O;
  const int grp = lane >> 4;
#pragma unroll
  for (int ch = 0; ch < TowerRegs<KIND>::NC0; ++ch) {
    if ((NO & 3) == 0 || ch < NO / 4) x0[ch] = obs_row[4 * ch + grp];
    else x0[ch] = 4 * ch + grp < NO ? obs_row[4 * ch + grp] : 0.0f;
  }
}

// standard normals from Philox bits (Box-Muller); keyed by (seed, global env id, episode, step):
// no host-side counter, so a captured graph draws fresh noise on every replay
template <int NA>
TB_DEV void policy_noise(unsigned long long seed, unsigned long long env_id, uint32_t episode, int step_count, float* eps) {
  uint32_t u[4];
#pragma unroll
  for (int blk = 0; blk < (NA + 3) / 4; ++blk) {
    philox4x32((uint32_t)env_id, (uint32_t)(env_id >> 32), episode, (uint32_t)step_count * 4u + (uint32_t)blk, (uint32_t)seed,
               (uint32_t)(seed >> 32) ^ 0x504F4C49u, u);
#pragma unroll
    for (int pair = 0; pair < 2; ++pair) {
      float u1 = ((float)(u[2 * pair] >> 8) + 1.0f) * 5.9604644775390625e-08f;  // (0, 1]
      float u2 = (float)(u[2 * pair + 1] >> 8) * 5.9604644775390625e-08f;         // [0, 1)
      float r = sqrtf(-2.0f * logf(u1)), th = 6.283185307179586f * u2;
      if (4 * blk + 2 * pair < NA) eps[4 * blk + 2 * pair] = r * cosf(th);
      if (4 * blk + 2 * pair + 1 < NA) eps[4 * blk + 2 * pair + 1] = r * sinf(th);
    }
  }
}
// the tower part of the one-step kernel (tb_policy_step: 256-thread workgroups, 64 envs): wave w runs BOTH towers of the
// 16-env slice w -- two independent chains the scheduler interleaves -- leaves the action means in s_mean[64][8] and
// writes the values
template <int KIND>
TB_DEV void policy_towers(const KArgs& A, float* s_mean) {
  constexpr int NO = Dims<KIND>::O;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, grp = lane >> 4;
  const int slot = wave * TB_POLICY_SLICE + (lane & 15), env = blockIdx.x * 64 + slot;
  const int env_c = env < A.n ? env : A.n - 1;
  float x0[TowerRegs<KIND>::NC0], mean[4], val[4];
  policy_inputs<KIND>(A.pol_obs + (size_t)env_c * NO, lane, x0);
  TowerRegs<KIND> pi, vf;
  pi.load(A.pol_weights, lane);
  vf.load(A.pol_weights + tower_floats<KIND>(), lane);
  __builtin_amdgcn_sched_barrier(0);  // keeps the scheduler from sinking each load down to its MFMA
  pi.apply(x0, mean);
  vf.apply(x0, val);
  if (grp < 2) *reinterpret_cast<float4*>(s_mean + slot * 8 + grp * 4) = make_float4(mean[0], mean[1], mean[2], mean[3]);
  if (lane < 16 && env < A.n) A.pol_value[env] = val[0];
}
// the env wave, after the workgroup barrier: sample, report, and hand the clipped actions to the env step
// `t`: the step of a rollout launch (tb_policy_rollout_kernel) whose output rows are written; 0 otherwise
// The noise of a step depends on the env's (episode, step) only, not on the policy's output: a caller
// with idle time before the means arrive (the env wave of tb_policy_rollout_kernel, while the towers
// run) draws it early with policy_draw and passes it in; `eps` = nullptr draws it here.
template <int KIND>
TB_DEV void policy_draw(const KArgs& A, int i, const EnvRegs& e, float* eps) {
  constexpr int NA = Dims<KIND>::A;
  if (!A.pol_deterministic) {
    policy_noise<NA>(A.pol_seed, A.env_id_base + (unsigned long long)i, e.episode, e.step_count, eps);
  } else {
#pragma unroll
    for (int k = 0; k < NA; ++k) eps[k] = 0.0f;
  }
}
template <int KIND>
TB_DEV void policy_sample(const KArgs& A, const float* s_mean, int i, const EnvRegs& e, float* a, size_t t = 0, const float* drawn = nullptr,
                          const float* stdv = nullptr /* exp(log_std), when the caller keeps it across steps */) {
  constexpr int NA = Dims<KIND>::A;
  float* out_act = A.pol_actions + t * A.st_act;
  float* out_raw = A.pol_raw + t * A.st_raw;
  const float* log_std = A.pol_weights + 2 * tower_floats<KIND>();
  const float* mean = s_mean + (threadIdx.x & 63) * 8;
  float eps[NA], logp = 0.0f;
  if (drawn) {
#pragma unroll
    for (int k = 0; k < NA; ++k) eps[k] = drawn[k];
  } else {
    policy_draw<KIND>(A, i, e, eps);
  }
#pragma unroll
  for (int k = 0; k < NA; ++k) {
    float ek = eps[k];
    float raw = FMA(stdv ? stdv[k] : expf(log_std[k]), ek, mean[k]);
    logp += FMA(-0.5f * ek, ek, -log_std[k]) - 0.9189385332046727f;  // -(eps^2)/2 - log_std - ln(2 pi)/2
    out_raw[(size_t)i * NA + k] = raw;
    a[k] = fminf(fmaxf(raw, -1.0f), 1.0f);  // SB3 clips Box actions before env.step
    out_act[(size_t)i * NA + k] = a[k];
  }
  A.pol_logp[t * A.st_logp + i] = logp;
}

// The same sampling for the rollout kernels' env wave, split in two so that nothing but arithmetic sits between the towers' means
// and the substep: `policy_sample_regs` works on registers only (exp(log_std) and log_std kept by the caller for the whole launch:
// the one-step form re-reads log_std from memory per step, six dependent loads in front of the step), `policy_store` writes the
// step's rows afterwards -- 8-byte stores, a row of NA floats being 8-byte aligned like the action rows tb_step reads.
// Same operations on the same values as policy_sample: bit-identical outputs.
template <int NA>
TB_DEV float policy_sample_regs(const float* mean, const float* eps, const float* stdv, const float* lstd, float* raw, float* a) {
  float logp = 0.0f;
#pragma unroll
  for (int k = 0; k < NA; ++k) {
    const float ek = eps[k];
    raw[k] = FMA(stdv[k], ek, mean[k]);
    logp += FMA(-0.5f * ek, ek, -lstd[k]) - 0.9189385332046727f;
    a[k] = fminf(fmaxf(raw[k], -1.0f), 1.0f);
  }
  return logp;
}
template <int NA>
TB_DEV void store_row2(float* dst, size_t row, const float* v) {
  static_assert(NA % 2 == 0, "action rows are written two floats at a time");
  float2* p = reinterpret_cast<float2*>(dst + row * NA);
#pragma unroll
  for (int k = 0; k < NA / 2; ++k) p[k] = make_float2(v[2 * k], v[2 * k + 1]);
}

}  // namespace

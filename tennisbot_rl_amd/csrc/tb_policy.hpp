// Policy inference for the fused policy + step kernel (tb_policy_step). Device code only; included by
// tb_stepper.hip after KArgs / EnvRegs / Dims / philox4x32 are defined. This file is part of the HIP
// library's single translation unit (everything lives in one anonymous namespace there).
#pragma once

namespace {

// Policy inference fused into the step kernel: SB3's MlpPolicy with separate pi / vf
// towers as the reference configures it -- SwingRacket 6 -> 32 -> 64 -> 32 (train_swing.py:80-82),
// Tennisbot 12 -> 64 -> 64 (SB3 default, train.py:104-110) -- tanh hidden layers, linear action mean,
// state-independent log_std, a = mean + std * eps.
//
// The towers run on the matrix cores in exact fp32 (v_mfma_f32_32x32x2_f32: bitwise a k-ordered fmaf
// chain, same peak as packed VALU FMA, but ONE VGPR per operand fragment instead of a broadcast weight
// per FMA -- a first VALU version spent its time re-reading weights out of LDS). Each layer is computed
// TRANSPOSED, H^T[out][env] = W^T[out][k] * X^T[k][env], 32 envs per wave: the weight fragment is the A
// operand, the activations the B operand, and -- the point of the transposition -- the C/D layout of one
// layer's output (lane = env, 16 registers = rows (r&3) + 8(r>>2) + 4(lane>>5)) IS the B layout of the
// next layer's input if the k-pairs are taken in that row order: register r of the output tile feeds
// "pair r" (k = row(r) on lanes 0-31, row(r)+4 on lanes 32-63). pack_policy() permutes the weights to
// match, so activations never leave the registers: no LDS, no shuffles, no barrier between layers. The
// k-sum order is that permutation (a fixed order; vs. torch within 1e-6).
// Four waves per 64 envs: {pi, vf} x {envs 0-31, 32-63}, independent until the pi waves hand the action
// means to wave 0 through LDS; wave 0 then samples and steps all 64 envs, the others retire.
// Blob, per tower and layer: bias tiles [out/32][2 halves][16 regs], then weight fragments
// [out/32][pairs][2][32] -- i.e. exactly what lane l loads at index l; heads padded to 32 outputs.
template <int KIND> struct PolicyNet;
template <> struct PolicyNet<TB_ENV_SWING> { static constexpr int NH = 3, H0 = 32, H1 = 64, H2 = 32, LAST = 32; };
template <> struct PolicyNet<TB_ENV_TENNIS> { static constexpr int NH = 2, H0 = 64, H1 = 64, H2 = 64, LAST = 64; };
constexpr int layer_floats(int in, int out) { return ((out + 31) / 32) * (32 + (in / 2) * 64); }
template <int KIND> constexpr int tower_floats() {  // hidden layers + the (padded) head
  using N = PolicyNet<KIND>;
  return layer_floats(Dims<KIND>::O, N::H0) + layer_floats(N::H0, N::H1) + (N::NH == 3 ? layer_floats(N::H1, N::H2) : 0) + layer_floats(N::LAST, 32);
}
template <int KIND> constexpr int policy_floats() { return 2 * tower_floats<KIND>() + (Dims<KIND>::A + 3) / 4 * 4; }

typedef float f32x16 __attribute__((ext_vector_type(16)));
// tanh(x) = 1 - 2 / (e^(2x) + 1) on the hardware exp2 / rcp units (1 ulp each): absolute error < 3e-7,
// saturates correctly at +-inf; 5 instructions instead of libm's ~40
TB_DEV float fast_tanh(float x) {
  float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
  return FMA(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
}

// one layer's operands for this lane: NT bias tiles (16 floats each) and NT * NP weight fragments
template <int NT, int NP>
struct LayerRegs {
  f32x16 bias[NT];
  float frag[NT * NP];
  TB_DEV void load(const float* g, int lane) {
    const int h = lane >> 5;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float4* p = reinterpret_cast<const float4*>(g + t * 32 + h * 16);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float4 v = p[q];
        bias[t][4 * q] = v.x; bias[t][4 * q + 1] = v.y; bias[t][4 * q + 2] = v.z; bias[t][4 * q + 3] = v.w;
      }
    }
    g += NT * 32;
#pragma unroll
    for (int f = 0; f < NT * NP; ++f) frag[f] = g[f * 64 + lane];
  }
  // y[t * 16 + r] = act(bias + sum over pairs): the next layer's B operands, in place
  template <bool TANH>
  TB_DEV void apply(const float (&x)[NP], float (&y)[NT * 16]) const {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      f32x16 c = bias[t];
#pragma unroll
      for (int pr = 0; pr < NP; ++pr) c = __builtin_amdgcn_mfma_f32_32x32x2f32(frag[t * NP + pr], x[pr], c, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 16; ++r) y[t * 16 + r] = TANH ? fast_tanh(c[r]) : c[r];
    }
  }
};

// one tower for 32 envs: lane l works on env (l & 31); out[0..3] = head rows 0-3 (lanes 0-31) or 4-7
// (lanes 32-63) of that env. Every operand of the tower is requested by load() up front (one VGPR per
// fragment): the loads of the later layers land while the earlier ones compute (policy_tower), or stay
// resident across the steps of a rollout launch (tb_policy_rollout_kernel).
template <int KIND> struct TowerRegs;
template <> struct TowerRegs<TB_ENV_SWING> {  // 6 -> 32 -> 64 -> 32 -> head
  using N = PolicyNet<TB_ENV_SWING>;
  static constexpr int O = Dims<TB_ENV_SWING>::O, NP0 = O / 2;
  LayerRegs<1, NP0> l0;
  LayerRegs<2, 16> l1;
  LayerRegs<1, 32> l2;
  LayerRegs<1, 16> lh;
  TB_DEV void load(const float* g, int lane) {
    l0.load(g, lane); g += layer_floats(O, N::H0);
    l1.load(g, lane); g += layer_floats(N::H0, N::H1);
    l2.load(g, lane); g += layer_floats(N::H1, N::H2);
    lh.load(g, lane);
  }
  TB_DEV void apply(const float (&x0)[NP0], float (&out)[4]) const {
    float h0[16], h1[32], h2[16], y[16];
    l0.template apply<true>(x0, h0);
    l1.template apply<true>(h0, h1);
    l2.template apply<true>(h1, h2);
    lh.template apply<false>(h2, y);
    out[0] = y[0]; out[1] = y[1]; out[2] = y[2]; out[3] = y[3];
  }
};
template <> struct TowerRegs<TB_ENV_TENNIS> {  // 12 -> 64 -> 64 -> head
  using N = PolicyNet<TB_ENV_TENNIS>;
  static constexpr int O = Dims<TB_ENV_TENNIS>::O, NP0 = O / 2;
  LayerRegs<2, NP0> l0;
  LayerRegs<2, 32> l1;
  LayerRegs<1, 32> lh;
  TB_DEV void load(const float* g, int lane) {
    l0.load(g, lane); g += layer_floats(O, N::H0);
    l1.load(g, lane); g += layer_floats(N::H0, N::H1);
    lh.load(g, lane);
  }
  TB_DEV void apply(const float (&x0)[NP0], float (&out)[4]) const {
    float h0[32], h1[32], y[16];
    l0.template apply<true>(x0, h0);
    l1.template apply<true>(h0, h1);
    lh.template apply<false>(h1, y);
    out[0] = y[0]; out[1] = y[1]; out[2] = y[2]; out[3] = y[3];
  }
};

// `obs_row`: this lane's env's observation (clamped to a valid env)
template <int KIND>
TB_DEV void policy_tower(const float* g, const float* obs_row, int lane, float (&out)[4]) {
  constexpr int NP0 = TowerRegs<KIND>::NP0;
  float x0[NP0];
#pragma unroll
  for (int pr = 0; pr < NP0; ++pr) x0[pr] = obs_row[2 * pr + (lane >> 5)];
  TowerRegs<KIND> regs;
  regs.load(g, lane);
  __builtin_amdgcn_sched_barrier(0);  // keeps the scheduler from sinking each load down to its MFMA
  regs.apply(x0, out);
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
// the tower part: wave w of the workgroup = (tower w >> 1, env half w & 1); the pi waves leave the
// action means in s_mean[64][8], the vf waves write the values
template <int KIND>
TB_DEV void policy_towers(const KArgs& A, float* s_mean) {
  constexpr int NO = Dims<KIND>::O;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, tower = wave >> 1, half = wave & 1;
  const int slot = half * 32 + (lane & 31), env = blockIdx.x * 64 + slot;
  const int env_c = env < A.n ? env : A.n - 1;
  float out[4];
  policy_tower<KIND>(A.pol_weights + tower * tower_floats<KIND>(), A.pol_obs + (size_t)env_c * NO, lane, out);
  if (tower == 0) {
    *reinterpret_cast<float4*>(s_mean + slot * 8 + (lane >> 5) * 4) = make_float4(out[0], out[1], out[2], out[3]);
  } else if (lane < 32 && env < A.n) {
    A.pol_value[env] = out[0];
  }
}
// wave 0, after the workgroup barrier: sample, report, and hand the clipped actions to the env step
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

}  // namespace

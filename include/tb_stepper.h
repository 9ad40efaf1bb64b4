/*
 * tb_stepper.h -- C ABI of the MI355X-native batched physics stepper for the
 * SwingRacket-v0 / Tennisbot-v0 environments of youliangtan/tennisbot-rl.
 *
 * This is the drop-in boundary. The reference has no FFI of its
 * own: its envs call the third-party `pybullet` C-API from Python. Each entry point
 * below therefore cites the reference call sites (file:line under the reference root)
 * whose work it replaces for a whole batch of N independent worlds.
 *
 * Conventions
 *   - plain C types only; no torch / HIP types in any signature. `stream` is a
 *     hipStream_t passed as void* (NULL = the default stream); every call is
 *     asynchronous on that stream unless stated otherwise.
 *   - all `*_dev` pointers are DEVICE pointers owned by the caller (for example
 *     torch tensors' data_ptr()). Layouts are row-major: actions [N][A], obs [N][O].
 *   - return value: 0 = OK, negative = TB_E_* below, positive = a hipError_t.
 *     Nothing throws across this boundary. tb_last_error() gives a thread-local
 *     message for the last non-zero return on the calling thread.
 *   - a handle is bound to one device and is not re-entrant.
 *   - there is NO CPU fallback in this library: without a usable HIP device
 *     tb_create fails.
 */
#ifndef TB_STEPPER_H
#define TB_STEPPER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TB_ABI_VERSION 4

/* library error codes (negative); positive return values are hipError_t */
#define TB_OK 0
#define TB_E_INVAL (-1)    /* bad argument (null pointer, n_envs <= 0, unknown env kind, ...) */
#define TB_E_NODEVICE (-2) /* no usable HIP device / device index out of range */
#define TB_E_PARAMS (-3)   /* TbParams failed validation (n_hull, dt, masses, ...) */
#define TB_E_UNSUPPORTED (-4)
#define TB_E_TIMEOUT (-5)    /* tb_mark_host_wait: the mark did not fire in time */

/* env kinds: tennisbot/__init__.py:3-11 registers exactly these two ids */
#define TB_ENV_SWING 0  /* SwingRacket-v0 -> tennisbot/envs/swingracket_env.py */
#define TB_ENV_TENNIS 1 /* Tennisbot-v0   -> tennisbot/envs/tennisbot_env.py   */

#define TB_SWING_ACT_DIM 6  /* swingracket_env.py:29-31 */
#define TB_SWING_OBS_DIM 6  /* swingracket_env.py:34-39,143-144 */
#define TB_TENNIS_ACT_DIM 2 /* tennisbot_env.py:43-44 */
#define TB_TENNIS_OBS_DIM 12 /* tennisbot_env.py:52-55,134-136 */

/* persistent state, structure-of-arrays: 32-bit words [TB_*_WORDS][N] + one byte [N] */
#define TB_SWING_WORDS 30
#define TB_TENNIS_WORDS 28
/* rows shared by both envs */
#define TB_W_RP 0   /* racket COM position (3)  racket.py:131 returns the COM frame */
#define TB_W_RQ 3   /* racket orientation quaternion x,y,z,w (4) */
#define TB_W_RV 7   /* racket linear velocity (3)  racket.py:142 */
#define TB_W_RW 10  /* racket angular velocity, world frame (3) */
#define TB_W_BP 13  /* ball position (3)  objects.py:57 */
#define TB_W_BV 16  /* ball linear velocity (3)  objects.py:64 */
#define TB_W_BW 19  /* ball angular velocity (3) */
/* SwingRacket-v0 rows */
#define TB_W_SW_GOAL 22   /* goal x,y (2)             swingracket_env.py:173 */
#define TB_W_SW_SPAWN 24  /* racket LINK spawn pos (3) swingracket_env.py:168 */
#define TB_W_SW_D0 27     /* initial_dist_to_goal      swingracket_env.py:174-175 */
#define TB_W_SW_STEP 28   /* step_count (int32)        swingracket_env.py:83,108 */
#define TB_W_SW_EPISODE 29 /* episode index (uint32), keys the reset RNG */
/* Tennisbot-v0 rows */
#define TB_W_TN_SHOOT 22  /* ball_shoot_force (3)      tennisbot_env.py:237-241 */
#define TB_W_TN_SCALE 25  /* racket globalScaling this episode was built with  tennisbot_env.py:230-234 */
#define TB_W_TN_STEP 26   /* step_count (int32)        tennisbot_env.py:122 */
#define TB_W_TN_EPISODE 27

/* done byte: 0 running; 1 done, and (Swing) the restoring force issued at the end
 * of the last fast-forward substep (swingracket_env.py:135-141) is still pending in
 * the engine's force accumulator; 2 done, nothing pending. */
#define TB_DONE_NO 0
#define TB_DONE_PENDING_FORCE 1
#define TB_DONE_YES 2

/* TbParams.flags */
#define TB_F_AUTO_RESET 0x1u          /* VecEnv semantics: a done env is reset inside the step */
#define TB_F_NET 0x2u                 /* court.urdf:43-47 second collision box */
#define TB_F_RACKET_BALL 0x4u         /* racket<->ball narrowphase + impulse; clear = BASELINE configs[1] "no ball contact" bench mode */
#define TB_F_RACKET_GROUND 0x8u       /* racket<->court-ground contact (court.urdf:19-24): a persistent manifold of
                                       * up to 4 hull vertices per env, one support point added per substep, warm-started rows
                                       * (DESIGN.md section 3). Opt-in. The reference's court does collide with the racket, and in
                                       * the SwingRacket fast-forward the racket is not gravity-compensated: it lands. What keeps
                                       * the flag out of the default is not the contact's own cost any more but the episodes it
                                       * creates: 0.08 % of random-action envs end with the ball at rest on the grounded racket and
                                       * run to the 800-substep limit (776 substeps of stacked resting contact, >= 6 us each, one
                                       * lane) -- at 4096 envs that is nearly every episode end of the batch, i.e. >= 4 ms behind
                                       * every rollout's join (the rollout itself takes 6.5 ms); rewards are not affected
                                       * (DESIGN.md section 3 has the measured distributions) */
#define TB_F_DEFAULT (TB_F_NET | TB_F_RACKET_BALL)

#define TB_MAX_HULL 64
#define TB_HULL_REC 8 /* floats per hull edge record */

/*
 * Scene + engine parameters. Every engine-semantics value recalled from Bullet
 *  is a named field so a host that has pybullet can calibrate.
 * All derived values (inverses, edge records) are computed once by the host side in
 * float64, rounded to float32 and consumed as given by the kernels.
 */
typedef struct TbParams {
  /* engine (Appendix B.1/B.2) */
  float dt;                 /* 1/240, racket.py:24 */
  float inv_dt;             /* 240 */
  float gravity;            /* 9.81, swingracket_env.py:154 setGravity(0,0,-9.81) */
  float lin_damp;           /* k1 = 0.04 of the damping force -m v (k1 + k2 |v|) */
  float ang_damp;           /* 0.04, the same form on the angular momentum */
  float lin_damp_quad;      /* k2 (ABI v4). Bullet's multibody path uses ONE damping value for both terms (k1 = k2 = 0.04): a field of its
                             * own so that the speed-proportional term can be switched off on its own (profiles/r03_pin_sensitivity.md) */
  float ang_damp_quad;
  float max_ang_step;       /* pi/4 per substep rotation clamp */
  float rest_vel_threshold; /* 0.2 m/s: below it restitution is 0 */
  float erp;                /* contact ERP (Baumgarte): 0.08 = PyBullet's world default (Bullet's library default is 0.2) */
  float contact_threshold;  /* 0.02 * ball radius: manifold keeps points closer than this */
  int32_t solver_iters;     /* sequential-impulse iteration cap (Bullet default 50) */
  float solver_tol;         /* stop early once every impulse update of a sweep is <= tol * |impulse|
                             * (float32 PGS otherwise oscillates by a few ulp and always runs to the cap) */
  uint32_t flags;           /* TB_F_* */
  /* racket: racket.urdf:17-21, racket.py:43-45 */
  float racket_mass, racket_inv_mass;
  float racket_inertia[3], racket_inv_inertia[3]; /* body-frame diagonal at scale 1; default: derived from the collision shape as Bullet does */
  float racket_com[3];       /* inertial origin in the link frame, at scale 1 */
  float racket_half_thick;   /* at scale 1 */
  float hull_margin;         /* URDF convex-hull collision margin, 0.001 (not scaled) */
  float hull_bound_radius;   /* max distance COM -> hull vertex at scale 1, for the cull */
  float racket_scale;        /* tennisbot_env.py:213-215: the globalScaling an env's racket is rebuilt
                              * with at its NEXT reset (tennisbot_env.py:230-234); each Tennisbot env
                              * keeps the scale of its current episode in its own state word */
  /* ball: ball.urdf:11-15,27-32, objects.py:48-50,67-72 */
  float ball_mass, ball_inv_mass, ball_inv_inertia, ball_radius;
  float magnus_k;            /* F = k (w x v); 0 reproduces the reference (BASELINE configs[4] extension) */
  float ball_spin_max;       /* reset: w0 ~ U(-max, max)^3; 0 reproduces the reference */
  /* pair coefficients: product rule, objects.py:16-18,29-31,48-50; goal keeps defaults */
  float rest_racket, rest_court, rest_goal;
  float fric_racket, fric_court, fric_goal;
  /* rolling friction of a ball contact (racket.py:43-45, objects.py:29-31,48-50 set rollingFriction .001):
   * combined coefficient of the pair, Bullet's rule rolling_a * friction_b + rolling_b * friction_a
   * (4e-4 with the racket and the court, 5e-4 with the goal: params.reference_rolling_friction()).
   * > 0 adds two angular rows per ball contact, along the friction directions, each boxed by
   * roll * j_n, solved between the normal and the sliding-friction rows. 0 (default) = no such rows:
   * whether Bullet's multibody solver visits them in PyBullet's default solver mode is not known here */
  float roll_racket, roll_court, roll_goal;
  /* racket <-> court (TB_F_RACKET_GROUND): product rule again (0.81, 0.04); manifold threshold
   * 0.02 * the racket's bounding radius at scale 1 (Bullet: 0.02 * angular motion disc) */
  float rest_racket_court, fric_racket_court, racket_ground_threshold;
  /* statics: court.urdf:19-24,43-47; simplegoal.urdf:17-22 (origins inside <geometry> are ignored) */
  float ground_half[3];
  float net_half[3];
  float goal_radius, goal_half_len;
  /* racket collision outline: CCW convex polygon in the COM frame (y, z), at scale 1.
   * record i = { a.y, a.z, e.y, e.z, 1/|e|^2, 1/|e|, 0, 0 } with e = v[i+1] - v[i] */
  int32_t n_hull;
  float hull_edges[TB_MAX_HULL][TB_HULL_REC];
} TbParams;

typedef struct TbHandle TbHandle;

/*
 * Kernel-selection options (since ABI v3; they replace the TB_BLOCK / TB_TENNIS_REG_ROWS / TB_SWING_REG_ROWS
 * environment variables of v2; v4 names the former `reserved` field and changes the policy blob's fragment order; `ff_seal` was appended
 * within v4: `struct_size` tells the library whether a caller has it, and an older caller gets the automatic choice). Every field: 0 = let the library choose from the batch size. None of
 * them changes any result -- the variants are bit-identical (tests/test_gpu_parity.py runs them all) --
 * only which instantiation of the same arithmetic is launched.
 */
typedef struct TbOptions {
  uint32_t struct_size;     /* sizeof(TbOptions): lets a newer library read an older caller's struct */
  int32_t block;            /* threads per workgroup of the step kernels: 64, 128 or 256 (auto: 64 up to 16384 envs, above that 128; 64 for SwingRacket-v0 between 49152 and 131072 envs) */
  int32_t tennis_reg_rows;  /* Tennisbot static contact rows in registers: 1 on, -1 off (auto: on) */
  int32_t swing_reg_rows;   /* the same for the pipelined SwingRacket step kernel: 1 on, -1 off (auto: on) */
  int32_t ff_lanes_per_wave; /* parked envs per wave in the first fast-forward phase, 1..64 (auto: 64 from 4096 envs on, fewer below) */
  int32_t ff_sort;          /* order parked envs by their ball's ballistic flight estimate before the fast-forward: 1 on (auto: off --
                             * with random actions the flight lengths are decided by events inside the loop, not by the parked state) */
  int32_t ff_phases;        /* the fast-forward as 1, 2 or 3 kernels: budgeted loop, then its compacted survivors (auto: 3 from 262144 envs on, else 1; from 131072 envs on the first of several also hands over every env whose ball reaches the racket) */
  int32_t ff_defer;         /* deferred fast-forwards (SwingRacket-v0 pipeline, up to 131072 envs). A fast-forward kernel lasts as long as its
                             * slowest env, at most four run at once (one per hardware queue), and every one is a FORK in a replayed graph that
                             * moves the chain of steps to another queue. 2: every episode end is parked straight into a pool (up to 64 episodes
                             * between two joins) and ONE launch runs all of them to their end when the caller joins (tb_flush and every call
                             * that flushes): the rollout is a single chain of step kernels. 1: each episode keeps its own fast-forward kernel,
                             * but an env still running after its ballistic flight estimate (capped at an un-struck ball's) + ff_defer_margin
                             * substeps moves on to the pool. -1: off. 0 = auto: 2 up to 16384 envs (4096 envs: 679 -> 871 M env steps/s; with
                             * racket<->court contact 92 -> 115-127 M), above that 1 with TB_F_RACKET_GROUND, else off (large batches run their
                             * fast-forwards beside the steps, in phases). Results do not change; terminal rewards are complete after the join,
                             * as with every pipelined path. Progress marks (a mark promises final steps): an explicit 2 gives the episodes parked since
                             * the previous mark their launch at tb_mark_record, on a side stream; auto and 1 fall back to one kernel per episode
                             * end while marks are enabled (measured: faster beside the chunks' all-gathers). Neither form is used for steps that
                             * ask for terminal-observation / substep outputs. */
  int32_t ff_defer_margin;  /* substeps beyond the estimate before an env is deferred (auto: 16) */
  int32_t policy_slices;    /* tb_policy_rollout: 16-env slices per workgroup, 1 (three waves per 16 envs) or 3 (seven waves per 48 envs) (auto: 1 up to 4096 envs) */
  int32_t ff_seal;          /* the pool's sealed-fate exit: 1 / 0 on (auto), -1 off. A SwingRacket-v0 flight whose ball has fallen below the court,
                             * out of the racket's reach for good, can only end in the 800-substep timeout with reward 0
                             * (swingracket_env.py:127-128); with auto-reset nothing else of it is ever read, so the pool books the remaining
                             * substeps (counters[6], substeps and timeouts are those of the full flight) instead of running them. Every
                             * result is the one the full flight gives (the oracle has no such exit and the parity tests compare against
                             * it); the pool's waves stop paying ~400 substeps for one such ball (a trained policy's collect: +5-8 %).
                             * Only without the Magnus extension and the extended contact set, with parameters inside the limits the
                             * argument needs (tb_kernels.hpp, fate_sealed), and never for launches that write terminal observations.
                             * tb_sealed_substeps reports how many substeps were booked this way. */
} TbOptions;

/* library identity / shape queries (host only, no device touched) */
int tb_abi_version(void);
int tb_obs_dim(int env_kind);
int tb_act_dim(int env_kind);
int tb_state_words(int env_kind);
const char *tb_last_error(void);

/*
 * Create a batch of n_envs independent worlds on `device`.
 * Replaces, per world: p.connect (swingracket_env.py:44-47, tennisbot_env.py:65-68),
 * and the one-time part of every loadURDF/changeDynamics (racket.py:35-45,
 * objects.py:25-36,43-50,102-104). State is NOT initialised: call tb_reset.
 * seed / env_id_base key the counter-based reset RNG: env i draws from
 * (seed, env_id_base + i, episode_index), so results do not depend on how a
 * global batch is sharded over GPUs.
 */
int tb_create(const TbParams *params, const TbOptions *options_or_null, int env_kind, int n_envs, int device,
              uint64_t seed, uint64_t env_id_base, TbHandle **out);

/* Replaces p.disconnect (swingracket_env.py:189, tennisbot_env.py:291). Synchronous. */
int tb_destroy(TbHandle *h);

/* Replace the parameter block. Takes effect for launches ENQUEUED after it on `stream`. The scalar
 * parameters travel in the kernel-argument block of each launch, so a hipGraph captured earlier keeps
 * the values it was captured with: tb_params_generation() goes up by one on every tb_set_params, and a
 * caller that replays captured steps compares it with the value at capture time and recaptures when it
 * differs (tennisbot_rl_amd.stepper.StepGraph does; a stale replay is refused, never silent). */
int tb_set_params(TbHandle *h, const TbParams *params, void *stream);
int tb_params_generation(TbHandle *h);
/* set_racket_scale (tennisbot_env.py:213-215; the curriculum callback train.py:164-176 calls it at every
 * rollout start): the globalScaling an env's racket is rebuilt with at ITS next reset. Unlike the rest of
 * TbParams this value is read by the reset code from the device-resident parameter block, not from the
 * kernel arguments: the update is one asynchronous 4-byte copy on `stream`, it does not bump
 * tb_params_generation, and REPLAYS OF GRAPHS CAPTURED EARLIER SEE IT (row f2 under hipGraph replay). */
int tb_set_racket_scale(TbHandle *h, float scale, void *stream);

/*
 * reset(): swingracket_env.py:151-186 / tennisbot_env.py:217-261 for every env i with
 * mask_dev == NULL or mask_dev[i] != 0. Starts a new episode (episode index + 1) and
 * writes the initial observation to obs_dev[i] if obs_dev != NULL.
 */
int tb_reset(TbHandle *h, const uint8_t *mask_dev, float *obs_dev, void *stream);

/*
 * step(action): swingracket_env.py:75-145 / tennisbot_env.py:104-207 for all envs,
 * including every p.stepSimulation() (swingracket_env.py:82,107; tennisbot_env.py:121),
 * the p.getContactPoints queries (swingracket_env.py:99,111,119; tennisbot_env.py:170),
 * Racket.apply_target_action (racket.py:92-100), Ball.apply_force (objects.py:67-72)
 * and the pose/velocity reads (racket.py:124-143, objects.py:52-65).
 *   actions_dev  [N][A] float32 in
 *   obs_dev      [N][O] float32 out
 *   reward_dev   [N]    float32 out
 *   done_dev     [N]    uint8   out (0/1)
 *   terminal_obs_dev [N][O] or NULL: with TB_F_AUTO_RESET, the last observation of an
 *                episode that ended in this step (untouched for other envs)
 *   substeps_dev [N] int32 or NULL: physics substeps executed for env i in this call
 */
int tb_step(TbHandle *h, const float *actions_dev, float *obs_dev, float *reward_dev,
            uint8_t *done_dev, float *terminal_obs_dev, int32_t *substeps_dev, void *stream);

/*
 * T consecutive step() calls in ONE launch with the state kept in registers:
 * actions [T][N][A] in; obs [T][N][O], reward [T][N], done [T][N] out; substeps_total
 * [N] int32 or NULL. Requires TB_F_AUTO_RESET. Same results as T calls of tb_step.
 * With the pipeline on (tb_set_pipeline), lockstep episodes and substeps_total_dev == NULL, the T steps
 * are issued as launches that end where the episodes end (<= 26 steps each), each followed by its
 * fast-forward on a side stream; the rewards of those terminal steps are then complete after tb_flush.
 */
int tb_rollout(TbHandle *h, int n_steps, const float *actions_dev, float *obs_dev,
               float *reward_dev, uint8_t *done_dev, int32_t *substeps_total_dev, void *stream);

/*
 * n_steps consecutive tb_step calls issued from one host call: step t reads actions at
 * actions_dev + t * actions_stride (bytes) and writes obs / reward / done at their bases + t * their
 * strides -- e.g. the records of a packed rollout buffer. Same launches, same results and the same
 * pipelining as n_steps calls of tb_step (no terminal_obs / substeps outputs); it only takes the host
 * language's per-call cost out of the loop (a ctypes call costs about as much as the step kernel runs).
 * Strides must keep every row as aligned as tb_step requires.
 */
int tb_step_sequence(TbHandle *h, int n_steps, const float *actions_dev, float *obs_dev, float *reward_dev,
                     uint8_t *done_dev, size_t actions_stride, size_t obs_stride, size_t reward_stride,
                     size_t done_stride, void *stream);

/*
 * step() with the policy inside ("policy inference on device ... so collect never
 * leaves the GPU"): SB3's MlpPolicy as the reference configures it (train_swing.py:80-82: pi = vf =
 * [32, 64, 32], tanh; Tennisbot-v0: SB3's default [64, 64], train.py:104-110), a = mean + exp(log_std)
 * * eps, clipped to the action space before the env sees it, as SB3 does. Per env i:
 *   obs_in_dev [N][O] the observation acted on  ->  actions_dev [N][A] (clipped), raw_actions_dev [N][A],
 *   logp_dev [N] (log-probability of the raw sample), value_dev [N]; then exactly tb_step on those actions.
 * The towers run on the matrix cores in fp32 (v_mfma_f32_16x16x4_f32, 16 envs per wave),
 * one layer's accumulator tile feeding the next layer's operand registers directly; tanh is evaluated
 * as 1 - 2/(exp(2x)+1) on the hardware exp2/rcp units (absolute error < 3e-7).
 * weights_dev: tb_policy_floats(kind) floats, 16-byte aligned, in the fragment order the kernel loads:
 * per tower (pi, then vf) its hidden layers and its head (padded to 16 outputs), each layer as
 *   bias tiles     [ceil(out/16)][4][4]: value (t, g, r) = bias[16t + 4g + r]
 *   weight frags   [ceil(out/16)][chunks][4][16]: value (t, c, g, j) = W[out 16t + j][in k(c, g)]
 * with k(c, g) = 4c + g for the first layer (zero beyond the observation) and k(4u + r, g) = 16u + 4g + r
 * after it (W = torch's nn.Linear.weight, zero beyond `out`); then log_std[A] padded to a multiple of 4.
 * tennisbot_rl_amd/ppo.py pack_policy() is the reference packer. eps is drawn in-kernel from Philox
 * keyed by (noise_seed, global env id, episode, step), so
 * a captured graph draws fresh noise on every replay; deterministic != 0 uses the mean.
 */
int tb_policy_floats(int env_kind);
int tb_policy_step(TbHandle *h, const float *weights_dev, const float *obs_in_dev, float *actions_dev,
                   float *raw_actions_dev, float *logp_dev, float *value_dev, float *obs_dev,
                   float *reward_dev, uint8_t *done_dev, uint64_t noise_seed, int deterministic, void *stream);

/*
 * n_steps of tb_policy_step without a launch boundary between them: whole episodes per launch, the
 * policy's weights resident in registers, the envs' state too; the observation a step produces is the
 * one the next step acts on (obs_in_dev is only read for the first). Output arrays are [n_steps][N][..];
 * step_strides_bytes (or NULL = contiguous) gives the distance in bytes between consecutive steps of
 * {actions, raw_actions, logp, value, obs, reward, done} -- e.g. the record size of a packed rollout
 * buffer; a 0 entry means contiguous for that array. Per env the arithmetic and the noise (keyed by
 * env, episode, step) are those of tb_policy_step: identical results. actions_dev / raw_actions_dev and their
 * step strides must be 8-byte aligned (rows are written two floats at a time). Requires TB_F_AUTO_RESET; on
 * SwingRacket-v0 also tb_set_pipeline(h, 1) and lockstep episodes (the launches end where the episodes
 * end, each followed by its fast-forward on a side stream; terminal rewards complete after tb_flush).
 */
int tb_policy_rollout(TbHandle *h, int n_steps, const float *weights_dev, const float *obs_in_dev,
                      float *actions_dev, float *raw_actions_dev, float *logp_dev, float *value_dev,
                      float *obs_dev, float *reward_dev, uint8_t *done_dev, const size_t *step_strides_bytes,
                      uint64_t noise_seed, int deterministic, void *stream);

/*
 * Pipelined fast-forward (SwingRacket-v0 with TB_F_AUTO_RESET; HIP streams, no reference
 * counterpart). The <= 775-substep fast-forward of swingracket_env.py:105-141 takes no
 * agent input, and the next episode does not depend on its outcome. With the pipeline
 * enabled, the 26th tb_step after a full tb_reset parks each env's pre-loop state, resets
 * the env and returns at once with obs (first observation of the new episode) and done = 1;
 * the loop itself runs on a stream owned by the handle, overlapping the following steps, and
 * then writes THAT step's reward, terminal observation and substep count into the buffers
 * that were passed to that tb_step call. Those buffers must therefore stay valid and must not
 * be reused until tb_flush; results are bit-identical to the unpipelined path.
 * tb_flush makes `stream` wait for all outstanding fast-forwards (tb_get_state, tb_set_state,
 * tb_counters, tb_reset, tb_set_params and tb_destroy do so themselves).
 * Replays of captured graphs are launched by the caller, not by this library: a caller that mixes eager
 * pipelined steps with graph replays calls tb_flush on the replaying stream first (the graph's launches bake in
 * slot indices and cannot wait for an eager fast-forward still reading its slot; stepper.StepGraph.replay does).
 * At most 2^24 envs per handle with the pipeline on. Device memory per env: 8 slots of parked records (192 B) and flags, with
 * ff_phases > 1 also up to two survivor lists per slot: 1.5-4.6 KB; handles that use the pool (TbOptions.ff_defer: on request, by
 * default up to 16384 envs, above that -- to 131072 -- only with TB_F_RACKET_GROUND, then allocated by the tb_set_params call that
 * turns it on) another 72 records + destination pointers: 14 KB.
 */
int tb_set_pipeline(TbHandle *h, int enable);
int tb_flush(TbHandle *h, void *stream);
/*
 * Progress marks (for callers that ship a rollout in chunks while ONE hipGraph is still producing it). A mark
 * tells the HOST that everything tb_step was asked to do before it is final in the caller's buffers -- the steps
 * themselves AND the late fast-forward writes they are still owed -- without making any stream wait for
 * anything: nothing is forked or joined, the graph keeps the shape it has without marks.
 * tb_mark_enable(h, 1): from now on every fast-forward kernel is followed on its side stream by a one-thread
 * kernel that counts it as finished in pinned host memory (off by default: plain graphs carry nothing extra);
 * call it before issuing -- or capturing -- the steps that marks will cover.
 * tb_mark_record(h, k, stream): the same one-thread kernel on `stream`, incrementing counter k; the library
 * remembers how many fast-forwards were enqueued before the mark (with TbOptions.ff_defer form 2 it first enqueues, on a side
 * stream, the ONE fast-forward launch of the episodes parked since the previous mark). Captured into a graph these are ordinary kernel nodes: every replay fires them again.
 * tb_mark_begin(h): snapshot of the counters; call it right before launching the work that contains the marks,
 * with nothing of this handle in flight (e.g. after a stream synchronize).
 * tb_mark_host_wait(h, k, timeout_ms): spin on the host until mark k has fired since tb_mark_begin and the
 * fast-forwards enqueued before it have finished; TB_E_TIMEOUT otherwise. The host waits, never a stream: a
 * stream-side wait queued on a hardware queue that the graph's own later kernels share would stall the very work
 * it waits for. One marked graph per handle at a time; 0 <= k < TB_MAX_MARKS.
 * tb_mark_count(h, k): how often mark k has fired since tb_create (a host read, no HIP call).
 */
#define TB_MAX_MARKS 64
int tb_mark_enable(TbHandle *h, int on);
int tb_mark_record(TbHandle *h, int k, void *stream);
int tb_mark_begin(TbHandle *h);
int tb_mark_host_wait(TbHandle *h, int k, int timeout_ms);
long long tb_mark_count(TbHandle *h, int k);
/* Stream-capture support (hipGraph): events recorded inside a capture are meaningless outside
 * it and vice versa. Call with host_wait = 1 right BEFORE beginning a capture that will contain
 * tb_step calls (drains the side streams on the host and forgets their events), capture
 * ... tb_step x K ... tb_flush on the capturing stream, end the capture, then call with
 * host_wait = 0 (forget the capture-local events). */
int tb_pipeline_sync(TbHandle *h, int host_wait);
/* The host-side episode phase of a SwingRacket handle: agent steps since the last common reset, modulo 26
 * (every episode is exactly 26 steps, swingracket_env.py:105-129), or -1 when the envs are not known to be
 * in lockstep (masked reset, injected state whose step counters differ). The pipelined kernels use it to
 * know which launch ends the episodes. A capture advances it by the captured steps although nothing ran:
 * tb_pipeline_sync(h, 0) therefore puts it back to the value of the matching tb_pipeline_sync(h, 1), and
 * whoever replays the captured steps calls tb_phase_advance(h, K) per replay -- after checking that
 * tb_phase(h) is the phase the steps were captured at (a graph bakes in WHICH of its steps end an episode).
 * tb_set_state re-derives the phase from the step-count row when every env agrees.
 * tb_phase_advance(h, K) is also how replayed steps reach the substep counter: every agent step runs at least one substep of
 * every env, and that share (n_envs x steps) is counted by the host -- per launch when a launch is enqueued to run, not at all
 * while it is only being captured, and per replay through this call. (One atomic per launch from one lane was 2.7 % of the
 * 4096-env rate.) */
int tb_phase(TbHandle *h);
int tb_phase_advance(TbHandle *h, int n_steps);
/* Which form the SwingRacket pipeline takes on this handle as it stands (TbOptions.ff_defer, the batch size, the contact flags,
 * progress marks on or off) for steps that ask for no terminal-observation / substep outputs: 0 = no pipeline (the fast-forward runs
 * inside the 26th step's kernel), 1 = one fast-forward kernel per episode end on a side stream, 2 = the same, its stragglers moving on
 * to the pool, 3 = every episode end parked into the pool, one fast-forward launch at the join (and at each progress mark). For
 * reports (bench.py names it). */
int tb_pipeline_form(TbHandle *h);
/* After a capture that contained tb_step calls was ABANDONED (it failed, e.g. because something else
 * in it was not capturable): the handle's side streams were forked into that capture and stay
 * invalidated, and the host's episode-phase hint ran ahead of the device. Replaces the side streams
 * and their events, restores the phase of the last tb_pipeline_sync(h, 1), clears the sticky HIP
 * error. The env state itself is untouched (nothing captured ever ran). */
int tb_pipeline_recover(TbHandle *h);

/* Snapshot / restore the persistent state (the reference never checkpoints env state;
 * SURVEY.md section 5). words: [tb_state_words][N] uint32 bit patterns, done: [N] bytes.
 * `on_device` != 0: the buffers are device memory (async on stream); 0: host memory
 * (the call synchronises the stream before returning). tb_set_state on a SwingRacket handle
 * with the pipeline ON also synchronises the stream in the on_device case (it reads the
 * step-count row back to re-derive the episode phase, see tb_phase) and so cannot be captured
 * into a graph; with the pipeline off an on_device restore is fully asynchronous. The racket<->court
 * contact caches are not part of the state words: a restored state starts with empty caches. */
int tb_get_state(TbHandle *h, uint32_t *words, uint8_t *done, int on_device, void *stream);
int tb_set_state(TbHandle *h, const uint32_t *words, const uint8_t *done, int on_device, void *stream);

/* per-handle event counters since create or the last tb_counters_reset: the batched
 * stand-in for the reference's per-step prints (swingracket_env.py:102,115,124,129;
 * tennisbot_env.py:172,191,202). out[0] racket-ball contact substeps, [1] ball-court
 * terminations, [2] goal hits, [3] timeouts, [4] pass-racket terminations, [5] episodes
 * finished, [6] substeps, [7] non-finite state detections, [8] lockstep violations: lanes that reached
 * an episode end in a launch the host had not given a fast-forward slot (only possible when a captured
 * graph is replayed at another phase than it was captured at; their terminal reward is lost). Synchronises
 * the stream.
 * CONTRACT for callers that replay captured graphs themselves: out[6] counts the first substep of every agent step on the HOST,
 * when tb_step / tb_rollout / tb_policy_* ENQUEUE work that runs (n_envs x steps per call; nothing during stream capture, where
 * nothing runs). Each replay of a graph holding K captured steps must therefore be reported with tb_phase_advance(h, K) -- the
 * call the episode phase needs anyway -- or out[6] under-reports by n_envs x K per replay (every other counter is device-side
 * and needs nothing). */
#define TB_N_COUNTERS 9
int tb_counters(TbHandle *h, uint64_t *out, void *stream);
int tb_counters_reset(TbHandle *h, void *stream);
/* how many of counters[6]'s substeps the pool's sealed-fate exit (TbOptions.ff_seal) booked without running them, since create or
 * the last tb_counters_reset. Joins the pipeline and synchronises the stream like tb_counters. */
int tb_sealed_substeps(TbHandle *h, uint64_t *out, void *stream);

/* Diagnostics (not part of the env surface): copies `rows` SoA rows of n 32-bit words,
 * src[r*n + i] -> dst[r*n + i], one dword per lane exactly like the step kernel's state
 * accesses. A known byte count in the kernel's own access pattern, used to calibrate the
 * rocprofv3 FETCH_SIZE / WRITE_SIZE counters (MI355X_MICROARCH.md, HBM section). */
int tb_diag_stream_copy(const uint32_t *src_dev, uint32_t *dst_dev, int n, int rows, int device, void *stream);
/* Diagnostics: `waves` one-wave workgroups that only stay resident (s_sleep) for `microseconds` on `stream`: the footprint of
 * the fast-forward waves without their arithmetic (tools/diag/r03_idle_probe.py: do resident waves shorten the dispatch gap between
 * the dependent launches of a graph?). */
int tb_diag_idle(int waves, int microseconds, int device, void *stream);
/* Test hook: the nth device allocation inside the NEXT tb_set_pipeline(h, 1) fails with hipErrorOutOfMemory
 * (0 = off). tb_set_pipeline is all-or-nothing: after a failure the handle is as if the pipeline had never
 * been enabled (nothing half-allocated for a later step to park into), and a second call starts over. */
int tb_diag_fail_alloc(int nth);

#ifdef __cplusplus
}
#endif
#endif /* TB_STEPPER_H */

// tb_stepper.hip -- the host side of libtb_stepper.so: the handle, the launches (which kernel instantiation runs when, on which
// stream, behind which event) and the C ABI of include/tb_stepper.h. The kernels themselves are in tb_kernels.hpp (env logic,
// __global__ entry points), tb_device.hpp (one substep of the rigid-body model) and tb_policy.hpp (MlpPolicy towers on MFMA).
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

#include "tb_kernels.hpp"

using namespace tb;

namespace {

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
  unsigned long long* d_counters;     // [TB_COUNTER_SHARDS][TB_N_COUNTERS] + one word: substeps booked by the pool's sealed-fate exit
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

// the parameter side of the pool's sealed-fate exit (fate_sealed in tb_kernels.hpp states the argument these limits belong to)
bool seal_params_ok(const TbHandle* h) {
  const KParams& k = h->kp;
  if (h->opt.ff_seal < 0 || h->kind != TB_ENV_SWING || !(k.flags & TB_F_AUTO_RESET) || k.magnus_k != 0.0f) return false;
  if (!(k.dt > 0.0f) || !(k.gravity > 0.0f) || !(k.racket_inv_mass > 0.0f) || !(k.lin_damp >= 0.0f) || !(k.lin_damp_quad >= 0.0f)) return false;
  const double wdt2 = 50.0 * (double)k.racket_inv_mass * (double)k.dt * (double)k.dt;      // (w dt)^2 of the stiffer axis
  const double c_max = (double)k.dt * ((double)k.lin_damp + 1000.0 * (double)k.lin_damp_quad);  // drag per substep at the 1000 m/s the test admits
  return wdt2 <= 0.04 && c_max <= 0.2;
}

KArgs base_args(const TbHandle* h) {
  KArgs a;
  memset(&a, 0, sizeof a);
  a.P = h->kp; a.words = h->d_words; a.done_state = h->d_done; a.hull = h->d_hull; a.counters = h->d_counters;
  a.ff_sealed = seal_params_ok(h) ? h->d_counters + TB_N_COUNTERS * TB_COUNTER_SHARDS : nullptr;
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
  // (153 VGPRs, three waves per SIMD, wave-shared outline sweep) holds them all at once, the small-batch one (188 VGPRs, two per
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
    if (opt.ff_seal < -1 || opt.ff_seal > 1) return fail(TB_E_INVAL, "tb_create: TbOptions.ff_seal must be -1, 0 or 1");
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
  CREATE_TRY(hipMalloc((void**)&h->d_counters, sizeof(unsigned long long) * (TB_N_COUNTERS * TB_COUNTER_SHARDS + 1)));
  CREATE_TRY(hipMemsetAsync(h->d_counters, 0, sizeof(unsigned long long) * (TB_N_COUNTERS * TB_COUNTER_SHARDS + 1), 0));
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

int tb_sealed_substeps(TbHandle* h, uint64_t* out, void* stream) {
  if (!h || !out) return fail(TB_E_INVAL, "tb_sealed_substeps: null argument");
  DeviceGuard g(h->device);
  if (int rc = flush_all(h, (hipStream_t)stream)) return rc;
  HIP_TRY(hipMemcpyAsync(out, h->d_counters + TB_N_COUNTERS * TB_COUNTER_SHARDS, sizeof(uint64_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
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
  HIP_TRY(hipMemsetAsync(h->d_counters, 0, sizeof(uint64_t) * (TB_N_COUNTERS * TB_COUNTER_SHARDS + 1), (hipStream_t)stream));
  h->first_substeps = 0;
  return TB_OK;
}

}  // extern "C"

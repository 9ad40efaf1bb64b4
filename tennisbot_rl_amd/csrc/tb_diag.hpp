// tb_diag.hpp -- everything the DIAGNOSTIC builds add to the kernels, behind one include.
//
// The product build (tennisbot_rl_amd/build.py) defines none of the TB_DIAG_* macros: every macro
// below then expands to nothing and the kernels in tb_device.hpp / tb_kernels.hpp read -- and
// compile -- as if this file did not exist. The diagnostic builds are made by tools/diag/*.py into
// /tmp and loaded through stepper.use_library(); they are never the in-tree libtb_stepper.so.
//
//   -DTB_DIAG_STAMPS         s_memtime stamps: cycles per substep segment and per kernel phase, summed per
//                            wave into g_diag_cycles[16] (tools/diag/diag_stamps*.py, diag_ff_sort.py)
//   -DTB_DIAG_TRACE          entry / exit real-time of every step-kernel launch into g_diag_trace (tools/diag/r03_cadence_probe.py)
//   -DTB_DIAG_CADENCE        the lean launch trace: entry / exit real-time of every step-kernel launch without any atomic
//                            (tools/diag/r04_cadence.py: the kernel_us / gap_us of bench.py's roofline)
//   -DTB_DIAG_LANES          lane census of the substep's wave votes into g_diag_lanes[16]
//                            (tools/diag/diag_lanes.py)
//   -DTB_DIAG_NO_ANGULAR / _NO_ORIENT / _NO_NARROW (/ _NO_RACKET / _NO_STATICS: its halves)
//                            timing-only ablations (tools/diag/diag_substep.py); RESULTS ARE WRONG
//   -DTB_DIAG_NO_PHILOX      timing-only: the reset's Philox draws replaced by a trivial hash (RESULTS ARE WRONG)
//   -DTB_DIAG_NO_TOWERS / _NO_ENVSTEP  timing-only: the policy rollout kernels without the MLP towers / without the env step
//                            (tools/diag/r04_policy_ablate.py; RESULTS ARE WRONG)
//   -DTB_DIAG_SWEEP_HELPERS=k  helper lanes of the wave-cooperative outline sweep (default 8)
//   -DTB_DIAG_LDS_PAD        (host side, tb_stepper.hip dyn_lds) pad every step launch's dynamic LDS by tb_diag_set_lds_pad(bytes):
//                            fewer workgroups per CU (tools/diag/r03_occupancy_probe.py)
//
// Included twice by design: once near the top of tb_device.hpp (device side, inside namespace tb)
// and once at the end of tb_stepper.hip with TB_DIAG_HOST_SECTION defined (the C entry points
// that read the counters back).
#ifndef TB_DIAG_HOST_SECTION
#ifndef TB_DIAG_HPP_DEVICE
#define TB_DIAG_HPP_DEVICE

#ifndef TB_DIAG_SWEEP_HELPERS
#define TB_DIAG_SWEEP_HELPERS 8
#endif

// ---- lane census ---------------------------------------------------------------------------
#ifdef TB_DIAG_LANES
__device__ unsigned long long g_diag_lanes[16];
// [k] += lanes for which `pred` holds, [k + 1] += 1 if any does (per wave-substep)
#define TB_LANES(k, pred) do { const unsigned long long m_ = __ballot(pred), a_ = __ballot(1); \
  if ((threadIdx.x & 63) == (unsigned)__ffsll((long long)a_) - 1u) { atomicAdd(&g_diag_lanes[k], (unsigned long long)__popcll(m_)); if (m_) atomicAdd(&g_diag_lanes[(k) + 1], 1ull); } } while (0)
#define TB_LANES_ADD1(k) atomicAdd(&g_diag_lanes[k], 1ull)
#else
#define TB_LANES(k, pred)
#define TB_LANES_ADD1(k)
#endif

// ---- cycle stamps --------------------------------------------------------------------------
#ifdef TB_DIAG_STAMPS
__device__ unsigned long long g_diag_cycles[16];
struct Stamps { unsigned long long t; unsigned int acc[8]; };
TB_DEV unsigned long long stamp_now() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
// a lane stops accumulating when its own env leaves the substep loop: the lane that stayed longest has
// the wave's complete account (every stamp adds the same scalar delta to all lanes still in the loop)
TB_DEV void diag_flush_stamps(const Stamps& st) {
  unsigned int mine = 0u;
  for (int k = 0; k < 6; ++k) mine += st.acc[k];
  unsigned int best = mine;
  for (int off = 32; off > 0; off >>= 1) { unsigned int o = __shfl_xor(best, off, 64); best = o > best ? o : best; }
  const unsigned long long holders = __ballot(mine == best);
  if ((threadIdx.x & 63) == (unsigned)__ffsll((long long)holders) - 1u)
    for (int k = 0; k < 6; ++k) atomicAdd(&g_diag_cycles[k], (unsigned long long)st.acc[k]);
}
#define TB_STAMP(st, k) do { unsigned long long _n = stamp_now(); (st).acc[k] += (unsigned int)(_n - (st).t); (st).t = _n; } while (0)
#define TB_STAMP_ARG , Stamps& st
#define TB_STAMP_PASS , st
#define TB_DIAG_STAMPS_BEGIN(st) Stamps st; for (int k_ = 0; k_ < 8; ++k_) st.acc[k_] = 0u; st.t = stamp_now()
#define TB_DIAG_STAMPS_END(st) diag_flush_stamps(st)
#define TB_DIAG_NOW(var) const unsigned long long var = stamp_now()
#define TB_DIAG_REALTIME(var) const unsigned long long var = __builtin_amdgcn_s_memrealtime()  /* 100 MHz ticks */
#define TB_DIAG_WAIT_LOADS(live) do { if (live) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); } while (0)
// g_diag_cycles[slot] += expr, once per wave (lane 0) / by the wave's first ACTIVE lane
#define TB_DIAG_ADD_LANE0(slot, expr) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_diag_cycles[slot], (unsigned long long)(expr)); } while (0)
#define TB_DIAG_ADD_LEADER(slot, expr) do { if ((threadIdx.x & 63) == (unsigned)__ffsll((long long)__ballot(1)) - 1u) atomicAdd(&g_diag_cycles[slot], (unsigned long long)(expr)); } while (0)
#define TB_DIAG_ADD_EACH(slot, expr) atomicAdd(&g_diag_cycles[slot], (unsigned long long)(expr))
#else
#define TB_STAMP(st, k) do { } while (0)
#define TB_STAMP_ARG
#define TB_STAMP_PASS
#define TB_DIAG_STAMPS_BEGIN(st) do { } while (0)
#define TB_DIAG_STAMPS_END(st) do { } while (0)
#define TB_DIAG_NOW(var) do { } while (0)
#define TB_DIAG_REALTIME(var) do { } while (0)
#define TB_DIAG_WAIT_LOADS(live) do { } while (0)
#define TB_DIAG_ADD_LANE0(slot, expr) do { } while (0)
#define TB_DIAG_ADD_LEADER(slot, expr) do { } while (0)
#define TB_DIAG_ADD_EACH(slot, expr) do { } while (0)
#endif

// ---- launch trace (-DTB_DIAG_TRACE, also part of -DTB_DIAG_STAMPS) ---------------------------
// the first thread of a step-kernel launch logs the 100 MHz real-time counter at entry and exit: the cadence of the launches
// inside a graph replay WITHOUT a profiler serialising them, at two scalar instructions and one atomic per launch
// (tools/diag/r03_cadence_probe.py)
#if defined(TB_DIAG_TRACE) || defined(TB_DIAG_STAMPS)
// (-DTB_DIAG_TRACE alone also logs, per launch, when the LAST workgroup of the launch before it left: every workgroup's first
//  thread raises g_diag_last_exit on its way out, the next launch's first thread collects it)
__device__ unsigned long long g_diag_trace[2 * 8192];
__device__ unsigned long long g_diag_trace_prev_all_out[8192];
__device__ unsigned long long g_diag_last_exit;
__device__ unsigned int g_diag_trace_n;
#ifdef TB_DIAG_TRACE
#define TB_DIAG_TRACE_ALL_OUT(tr) g_diag_trace_prev_all_out[tr] = atomicExch(&g_diag_last_exit, 0ull)
#define TB_DIAG_TRACE_LEAVE() do { if (threadIdx.x == 0) atomicMax(&g_diag_last_exit, (unsigned long long)__builtin_amdgcn_s_memrealtime()); } while (0)
#else
#define TB_DIAG_TRACE_ALL_OUT(tr) do { } while (0)
#define TB_DIAG_TRACE_LEAVE() do { } while (0)
#endif
#define TB_DIAG_TRACE_ENTRY(tr) unsigned int tr = 0xffffffffu; do { if (blockIdx.x == 0 && threadIdx.x == 0) { \
  tr = atomicAdd(&g_diag_trace_n, 1u) & 8191u; g_diag_trace[2 * tr] = __builtin_amdgcn_s_memrealtime(); TB_DIAG_TRACE_ALL_OUT(tr); } } while (0)
#define TB_DIAG_TRACE_EXIT(tr) do { if (tr != 0xffffffffu) g_diag_trace[2 * tr + 1] = __builtin_amdgcn_s_memrealtime(); TB_DIAG_TRACE_LEAVE(); } while (0)
#else
#define TB_DIAG_TRACE_ENTRY(tr) do { } while (0)
#define TB_DIAG_TRACE_EXIT(tr) do { } while (0)
#endif

// ---- launch cadence (-DTB_DIAG_CADENCE) ------------------------------------------------------
// The lean form of the launch trace, for the figures bench.py quotes (tools/diag/r04_cadence.py -> profiles/r04_cadence.json): the
// first thread of a step-kernel launch reads the 100 MHz real-time counter on entry and on exit and stores both at the end -- no
// atomic, nothing any other thread does, one extra load (the launch's running number, requested beside the state loads) and three
// stores by one lane. A build with it replays the headline graph at the product's rate (the -DTB_DIAG_TRACE build above, with its
// returning atomic at entry and one atomic per workgroup at exit, runs 20 % slower: its gaps are not the product's).
#ifdef TB_DIAG_CADENCE
__device__ unsigned long long g_diag_cadence[2 * 8192];
__device__ unsigned int g_diag_cadence_n;
#define TB_DIAG_CADENCE_ENTRY(t0, nn) const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(); unsigned int nn = 0u; \
  do { if (blockIdx.x == 0 && threadIdx.x == 0) nn = *reinterpret_cast<volatile unsigned int*>(&g_diag_cadence_n); } while (0)
#define TB_DIAG_CADENCE_EXIT(t0, nn) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const unsigned long long t1_ = __builtin_amdgcn_s_memrealtime(); \
  g_diag_cadence[2 * (nn & 8191u)] = t0; g_diag_cadence[2 * (nn & 8191u) + 1] = t1_; g_diag_cadence_n = nn + 1u; } } while (0)
#else
#define TB_DIAG_CADENCE_ENTRY(t0, nn) do { } while (0)
#define TB_DIAG_CADENCE_EXIT(t0, nn) do { } while (0)
#endif

// ---- timing-only ablations (results are wrong) ---------------------------------------------
#ifdef TB_DIAG_NO_ANGULAR
#define TB_DIAG_ABLATE_ANGULAR(flag) flag = false
#else
#define TB_DIAG_ABLATE_ANGULAR(flag) do { } while (0)
#endif
#ifdef TB_DIAG_NO_ORIENT
#define TB_DIAG_ABLATE_ORIENT(w2) w2 = 0.0f
#else
#define TB_DIAG_ABLATE_ORIENT(w2) do { } while (0)
#endif
#if defined(TB_DIAG_NO_NARROW) || defined(TB_DIAG_NO_RACKET)  // (_NO_RACKET / _NO_STATICS: one half of _NO_NARROW each)
#define TB_DIAG_ABLATE_NARROW(flag) flag = false
#else
#define TB_DIAG_ABLATE_NARROW(flag) do { } while (0)
#endif
#ifdef TB_DIAG_NO_PHILOX  // (the reset's random draws replaced by a cheap hash of the same inputs: what does Philox cost a launch?)
#define TB_DIAG_PHILOX(c0, c1, c2, c3, k0, k1, out) do { uint32_t h_ = (c0) * 2654435761u ^ (c2) * 40503u ^ (c3); \
  (out)[0] = h_; (out)[1] = h_ * 3u + (k0); (out)[2] = h_ * 5u + (k1); (out)[3] = h_ * 7u + (c1); } while (0)
#else
#define TB_DIAG_PHILOX(c0, c1, c2, c3, k0, k1, out) philox4x32(c0, c1, c2, c3, k0, k1, out)
#endif
#if defined(TB_DIAG_NO_NARROW) || defined(TB_DIAG_NO_STATICS)
#define TB_DIAG_ABLATE_STATICS(flag) flag = false
#else
#define TB_DIAG_ABLATE_STATICS(flag) do { } while (0)
#endif

#endif  // TB_DIAG_HPP_DEVICE
#else   // TB_DIAG_HOST_SECTION: inside tb_stepper.hip's extern "C" block, after HIP_TRY / fail()

#ifdef TB_DIAG_STAMPS
int tb_diag_read_stamps(unsigned long long* out16, int reset) {
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_diag_cycles), sizeof(unsigned long long) * 16));
  if (reset) { unsigned long long z[16] = {0}; HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_diag_cycles), z, sizeof z)); }
  return TB_OK;
}
#endif
#if defined(TB_DIAG_TRACE) || defined(TB_DIAG_STAMPS)
int tb_diag_read_trace(unsigned long long* out_pairs, int max_pairs, int reset) {
  HIP_TRY(hipDeviceSynchronize());
  unsigned int n = 0;
  HIP_TRY(hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_diag_trace_n), sizeof n));
  int k = (int)(n < 8192u ? n : 8192u);
  if (k > max_pairs) k = max_pairs;
  if (k > 0) HIP_TRY(hipMemcpyFromSymbol(out_pairs, HIP_SYMBOL(g_diag_trace), sizeof(unsigned long long) * 2 * (size_t)k));
  if (reset) { unsigned int z = 0; HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_diag_trace_n), &z, sizeof z)); }
  return k;
}
// per traced launch: when the last workgroup of the launch BEFORE it left (0: unknown)
int tb_diag_read_trace_all_out(unsigned long long* out, int max_entries) {
  HIP_TRY(hipDeviceSynchronize());
  const int k = max_entries < 8192 ? max_entries : 8192;
  if (k > 0) HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_diag_trace_prev_all_out), sizeof(unsigned long long) * (size_t)k));
  return k;
}
#endif
#ifdef TB_DIAG_CADENCE
// the ring of entry / exit pairs (100 MHz ticks; launch k of the count since the last reset sits at pair k % 8192); returns that count
int tb_diag_read_cadence(unsigned long long* out_8192_pairs, int reset) {
  HIP_TRY(hipDeviceSynchronize());
  unsigned int n = 0;
  HIP_TRY(hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_diag_cadence_n), sizeof n));
  HIP_TRY(hipMemcpyFromSymbol(out_8192_pairs, HIP_SYMBOL(g_diag_cadence), sizeof(unsigned long long) * 2 * 8192));
  if (reset) { unsigned int z = 0; HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_diag_cadence_n), &z, sizeof z)); }
  return (int)n;
}
#endif
#ifdef TB_DIAG_LDS_PAD
int tb_diag_set_lds_pad(long bytes) { g_diag_lds_pad = bytes > 0 ? (size_t)bytes : 0; return TB_OK; }
#endif
#ifdef TB_DIAG_LANES
int tb_diag_read_lanes(unsigned long long* out16, int reset) {
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_diag_lanes), sizeof(unsigned long long) * 16));
  if (reset) { unsigned long long z[16] = {0}; HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_diag_lanes), z, sizeof z)); }
  return TB_OK;
}
#endif

#endif  // TB_DIAG_HOST_SECTION

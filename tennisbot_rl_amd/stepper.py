"""Tensor API over the C ABI (include/tb_stepper.h): N independent (racket, ball) worlds
stepped in lockstep on one MI355X.

This is the measured path: `step(actions)` takes and returns device tensors and never
synchronises with the host. torch is used for device memory and streams only; all
arithmetic happens in libtb_stepper.so (hand-written HIP: csrc/tb_kernels.hpp, tb_device.hpp, tb_policy.hpp; host side csrc/tb_stepper.hip). There is
no CPU or eager-PyTorch fallback: if the library or a GPU is missing, construction raises.

Reference counterparts: `SwingRacketEnv` (tennisbot/envs/swingracket_env.py:24-192) and
`TennisbotEnv` (tennisbot/envs/tennisbot_env.py:27-291), one PyBullet world each.
"""
import ctypes
import os

import numpy as np

from .params import (ACT_DIM, COUNTER_NAMES, ENV_SWING, ENV_TENNIS, F_AUTO_RESET, N_COUNTERS, OBS_DIM,
                     STATE_ROWS, STATE_WORDS, TbOptions, TbParams, default_params, make_options)

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libtb_stepper.so")
_LIB = None
ABI_VERSION = 4

ENV_IDS = {"SwingRacket-v0": ENV_SWING, "Tennisbot-v0": ENV_TENNIS}  # tennisbot/__init__.py:3-11


class StepperError(RuntimeError):
    pass


def lib_path():
    return _LIB_PATH


def use_library(path):
    """Load another BUILD of the same sources instead of the in-tree one (the -DTB_DIAG_* diagnostic builds of
    tools/diag_*.py). An explicit call before the first BatchedEnv, not an environment variable: nothing
    outside the caller's own code can redirect the product path."""
    global _LIB_PATH, _LIB
    if _LIB is not None:
        raise StepperError("use_library() must be called before the library is first loaded")
    _LIB_PATH = os.path.abspath(path)


def load_library():
    """dlopen the HIP library. Fails loudly: the product has no fallback path."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(_LIB_PATH):
        raise StepperError(
            "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % _LIB_PATH)
    L = ctypes.CDLL(_LIB_PATH)
    vp, i32, u64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64
    L.tb_abi_version.restype = i32
    L.tb_obs_dim.argtypes = [i32]
    L.tb_act_dim.argtypes = [i32]
    L.tb_state_words.argtypes = [i32]
    L.tb_last_error.restype = ctypes.c_char_p
    L.tb_create.argtypes = [ctypes.POINTER(TbParams), ctypes.POINTER(TbOptions), i32, i32, i32, u64, u64, ctypes.POINTER(vp)]
    L.tb_destroy.argtypes = [vp]
    L.tb_set_params.argtypes = [vp, ctypes.POINTER(TbParams), vp]
    L.tb_reset.argtypes = [vp, vp, vp, vp]
    L.tb_step.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    L.tb_rollout.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp]
    L.tb_get_state.argtypes = [vp, vp, vp, i32, vp]
    L.tb_set_state.argtypes = [vp, vp, vp, i32, vp]
    L.tb_counters.argtypes = [vp, vp, vp]
    L.tb_counters_reset.argtypes = [vp, vp]
    L.tb_sealed_substeps.argtypes = [vp, vp, vp]
    L.tb_set_pipeline.argtypes = [vp, i32]
    L.tb_set_pipeline.restype = i32
    L.tb_flush.argtypes = [vp, vp]
    L.tb_flush.restype = i32
    L.tb_pipeline_sync.argtypes = [vp, i32]
    L.tb_pipeline_sync.restype = i32
    L.tb_policy_rollout.argtypes = [vp, i32] + [vp] * 9 + [ctypes.POINTER(ctypes.c_size_t), u64, i32, vp]
    L.tb_policy_rollout.restype = i32
    L.tb_step_sequence.argtypes = [vp, i32, vp, vp, vp, vp, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, vp]
    L.tb_step_sequence.restype = i32
    L.tb_policy_floats.argtypes = [i32]
    L.tb_policy_floats.restype = i32
    L.tb_policy_step.argtypes = [vp] * 10 + [u64, i32, vp]
    L.tb_policy_step.restype = i32
    L.tb_phase.argtypes = [vp]
    L.tb_phase.restype = i32
    L.tb_phase_advance.argtypes = [vp, i32]
    L.tb_phase_advance.restype = i32
    L.tb_pipeline_form.argtypes = [vp]
    L.tb_pipeline_form.restype = i32
    L.tb_params_generation.argtypes = [vp]
    L.tb_params_generation.restype = i32
    L.tb_set_racket_scale.argtypes = [vp, ctypes.c_float, vp]
    L.tb_set_racket_scale.restype = i32
    L.tb_pipeline_recover.argtypes = [vp]
    L.tb_pipeline_recover.restype = i32
    L.tb_mark_record.argtypes = [vp, i32, vp]
    L.tb_mark_record.restype = i32
    L.tb_mark_count.argtypes = [vp, i32]
    L.tb_mark_count.restype = ctypes.c_longlong
    L.tb_mark_host_wait.argtypes = [vp, i32, i32]
    L.tb_mark_host_wait.restype = i32
    L.tb_mark_begin.argtypes = [vp]
    L.tb_mark_begin.restype = i32
    L.tb_mark_enable.argtypes = [vp, i32]
    L.tb_mark_enable.restype = i32
    L.tb_diag_stream_copy.argtypes = [vp, vp, i32, i32, i32, vp]
    L.tb_diag_stream_copy.restype = i32
    L.tb_diag_idle.argtypes = [i32, i32, i32, vp]
    L.tb_diag_idle.restype = i32
    L.tb_diag_fail_alloc.argtypes = [i32]
    L.tb_diag_fail_alloc.restype = i32
    for f in ("tb_create", "tb_destroy", "tb_set_params", "tb_reset", "tb_step", "tb_rollout", "tb_get_state",
              "tb_set_state", "tb_counters", "tb_counters_reset", "tb_obs_dim", "tb_act_dim", "tb_state_words"):
        getattr(L, f).restype = i32
    if L.tb_abi_version() != ABI_VERSION:
        raise StepperError("libtb_stepper.so ABI version %d, expected %d" % (L.tb_abi_version(), ABI_VERSION))
    for kind in (ENV_SWING, ENV_TENNIS):
        assert L.tb_obs_dim(kind) == OBS_DIM[kind] and L.tb_act_dim(kind) == ACT_DIM[kind]
        assert L.tb_state_words(kind) == STATE_WORDS[kind]
    _LIB = L
    return L


def _check(L, rc, what):
    if rc != 0:
        raise StepperError("%s failed (%d): %s" % (what, rc, L.tb_last_error().decode()))


class StepGraph:
    """K captured agent steps of one BatchedEnv (BatchedEnv.capture). replay() runs them with one launch --
    after checking what a captured launch cannot check for itself:
      * the parameter block travels in the kernel arguments, so a graph captured before set_params() would
        step with the old parameters: refused (recapture); set_racket_scale() is exempt, the kernels read
        the scale from device memory;
      * a pipelined SwingRacket graph bakes in WHICH of its steps end an episode (and fork a fast-forward):
        it is only valid from the episode phase it was captured at. K % 26 != 0, or steps taken outside the
        graph in between, move the phase: refused, instead of episode ends falling into launches that have
        no fast-forward slot (their terminal rewards would be lost, counters()['lockstep_violations'])."""

    def __init__(self, env, graph, n_steps, phase, generation):
        self.env, self.graph, self.n_steps, self.phase, self.generation = env, graph, int(n_steps), phase, generation

    @property
    def repeatable(self):
        return self.phase < 0 or self.n_steps % 26 == 0

    def valid(self):
        """may replay() run now? (same parameter block, same episode phase as at capture time)"""
        env = self.env
        return env.L.tb_params_generation(env._h) == self.generation and (self.phase < 0 or env.L.tb_phase(env._h) == self.phase)

    def replay(self):
        env = self.env
        if env.L.tb_params_generation(env._h) != self.generation:
            raise StepperError("this graph was captured before set_params(): its launches carry the old parameter block; capture it again")
        if self.phase >= 0:
            now = env.L.tb_phase(env._h)
            if now != self.phase:
                raise StepperError("this graph was captured at episode phase %d and can only be replayed from there; the envs are at phase %d "
                                   "(captured steps %% 26 = %d; or steps were taken outside the graph)" % (self.phase, now, self.n_steps % 26))
        if env.pipeline:
            # eager pipelined steps may still have fast-forwards running on the handle's side streams; the captured launches bake
            # in slot indices and hold no wait on them (the capture began with every slot idle). Up to 8 hipStreamWaitEvent
            # calls, nothing when no slot is busy.
            env.flush()
        self.graph.replay()
        _check(env.L, env.L.tb_phase_advance(env._h, self.n_steps), "tb_phase_advance")


class BatchedEnv:
    """N lockstep worlds of one env kind on one GPU.

    step(actions) -> (obs, reward, done): float32 [N, O], float32 [N], uint8 [N] device tensors.
    With auto_reset (default, SB3 VecEnv semantics) an env that finishes is reset inside the
    same kernel: `obs` then holds the first observation of the new episode and
    `terminal_obs()` the last one of the old episode. With auto_reset=False `done` is sticky
    until reset(), exactly like the reference's `self.done`.
    """

    def __init__(self, env_kind, num_envs, device=None, seed=0, env_id_base=0, params=None, auto_reset=True,
                 reuse_buffers=False, track_terminal_obs=True, pipeline=False, options=None):
        import torch
        self.torch = torch
        if isinstance(env_kind, str):
            env_kind = ENV_IDS[env_kind]
        if env_kind not in (ENV_SWING, ENV_TENNIS):
            raise ValueError("unknown env kind %r" % (env_kind,))
        self.L = load_library()
        if not torch.cuda.is_available():
            raise StepperError("no GPU visible to torch: the batched stepper is HIP-only (no CPU fallback)")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.type != "cuda":
            raise StepperError("device must be a cuda (ROCm) device, got %s" % self.device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.kind, self.num_envs = env_kind, int(num_envs)
        self.obs_dim, self.act_dim, self.words = OBS_DIM[env_kind], ACT_DIM[env_kind], STATE_WORDS[env_kind]
        self.seed, self.env_id_base = int(seed), int(env_id_base)
        self._row_align = 8 if env_kind == ENV_SWING else 16  # float2 / float4 row accesses
        p = (params or default_params()).copy()
        p.flags = (p.flags | F_AUTO_RESET) if auto_reset else (p.flags & ~F_AUTO_RESET)
        self.params = p
        self.auto_reset = bool(auto_reset)
        self.reuse_buffers = bool(reuse_buffers)
        self._h = ctypes.c_void_p()
        # options: a TbOptions, or a dict of make_options() keywords (kernel variants; results never depend on them)
        self.options = options if isinstance(options, TbOptions) else make_options(**(options or {}))
        _check(self.L, self.L.tb_create(ctypes.byref(p), ctypes.byref(self.options), env_kind, self.num_envs, self.device.index, self.seed,
                                        self.env_id_base, ctypes.byref(self._h)), "tb_create")
        self._steps_issued = 0  # agent steps asked of the library so far (capture() measures its K with it)
        n, o = self.num_envs, self.obs_dim
        self._term = torch.zeros((n, o), dtype=torch.float32, device=self.device) if (auto_reset and track_terminal_obs) else None
        self._substeps = torch.zeros(n, dtype=torch.int32, device=self.device)
        self._bufs = None
        # pipelined fast-forward (tb_set_pipeline): terminal rewards of SwingRacket episodes are
        # written later, from a side stream, into the buffers of the step that ended the episode
        self.pipeline = bool(pipeline)
        self._inflight = []
        if self.pipeline:
            if not (auto_reset and env_kind == ENV_SWING):
                raise ValueError("pipeline=True needs SwingRacket-v0 with auto_reset")
            if reuse_buffers:
                raise ValueError("pipeline=True writes late into each step's own buffers: reuse_buffers must be off")
            _check(self.L, self.L.tb_set_pipeline(self._h, 1), "tb_set_pipeline")

    # ------------------------------------------------------------------ plumbing
    def _stream(self):
        return ctypes.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def _out(self, T=None):
        t, n, o = self.torch, self.num_envs, self.obs_dim
        lead = (n,) if T is None else (T, n)
        if self.reuse_buffers and T is None:
            if self._bufs is None:
                self._bufs = (t.empty(lead + (o,), dtype=t.float32, device=self.device),
                              t.empty(lead, dtype=t.float32, device=self.device),
                              t.empty(lead, dtype=t.uint8, device=self.device))
            return self._bufs
        return (t.empty(lead + (o,), dtype=t.float32, device=self.device), t.empty(lead, dtype=t.float32, device=self.device),
                t.empty(lead, dtype=t.uint8, device=self.device))

    def _check_tensor(self, x, shape, dtype, name):
        t = self.torch
        if not isinstance(x, t.Tensor):
            raise TypeError("%s must be a torch tensor" % name)
        if x.device != self.device:
            raise ValueError("%s is on %s, the env batch lives on %s" % (name, x.device, self.device))
        if x.dtype != dtype:
            raise TypeError("%s must be %s, got %s" % (name, dtype, x.dtype))
        if tuple(x.shape) != tuple(shape):
            raise ValueError("%s must have shape %s, got %s" % (name, tuple(shape), tuple(x.shape)))
        if not x.is_contiguous():
            raise ValueError("%s must be contiguous" % name)
        if dtype == t.float32 and x.dim() >= 2 and x.data_ptr() % self._row_align:
            raise ValueError("%s must be %d-byte aligned (the kernels use vector accesses per row)" % (name, self._row_align))
        return x

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self.L.tb_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ env surface
    def reset(self, mask=None):
        """Start a new episode in every env (or those with mask != 0). Returns obs [N, O]."""
        t = self.torch
        obs = t.empty((self.num_envs, self.obs_dim), dtype=t.float32, device=self.device)
        if mask is not None:
            mask = self._check_tensor(mask.to(t.uint8) if mask.dtype == t.bool else mask, (self.num_envs,), t.uint8, "mask")
            # unmasked rows keep their current observation
            obs.copy_(self.observe())
        _check(self.L, self.L.tb_reset(self._h, None if mask is None else mask.data_ptr(), obs.data_ptr(), self._stream()), "tb_reset")
        return obs

    def step(self, actions, out=None):
        """out: optional (obs, reward, done) tensors to write in place (e.g. slices of a
        RolloutBuffer), shapes [N, O] f32, [N] f32, [N] u8, contiguous, on this device."""
        t = self.torch
        a = self._check_tensor(actions, (self.num_envs, self.act_dim), t.float32, "actions")
        if out is None:
            obs, rew, done = self._out()
        else:
            obs = self._check_tensor(out[0], (self.num_envs, self.obs_dim), t.float32, "out[0]")
            rew = self._check_tensor(out[1], (self.num_envs,), t.float32, "out[1]")
            done = self._check_tensor(out[2], (self.num_envs,), t.uint8, "out[2]")
        _check(self.L, self.L.tb_step(self._h, a.data_ptr(), obs.data_ptr(), rew.data_ptr(), done.data_ptr(),
                                      None if self._term is None else self._term.data_ptr(),
                                      None if self.pipeline else self._substeps.data_ptr(), self._stream()), "tb_step")
        self._steps_issued += 1
        if self.pipeline and out is None:
            self._inflight.append(rew)  # keep the late-written buffer alive until flush()
            if len(self._inflight) > 4096:
                self.flush()
        return obs, rew, done

    def step_ptrs(self, actions_ptr, obs_ptr, reward_ptr, done_ptr):
        """Unchecked fast path for callers that validated their buffers once (RolloutBuffer.bind):
        raw device addresses of [N, A] f32 actions and [N, O] f32 / [N] f32 / [N] u8 outputs."""
        rc = self.L.tb_step(self._h, actions_ptr, obs_ptr, reward_ptr, done_ptr, None, None,
                            self.torch.cuda.current_stream(self.device).cuda_stream)
        if rc:
            _check(self.L, rc, "tb_step")
        self._steps_issued += 1

    def step_sequence_ptrs(self, n_steps, actions_ptr, obs_ptr, reward_ptr, done_ptr, strides):
        """n_steps consecutive steps from ONE host call (tb_step_sequence): step t uses the four
        device addresses advanced by t * strides[k] bytes (actions, obs, reward, done). Unchecked
        fast path, like step_ptrs; RolloutBuffer.step_range is the validated caller."""
        rc = self.L.tb_step_sequence(self._h, int(n_steps), actions_ptr, obs_ptr, reward_ptr, done_ptr,
                                     strides[0], strides[1], strides[2], strides[3], self.torch.cuda.current_stream(self.device).cuda_stream)
        if rc:
            _check(self.L, rc, "tb_step_sequence")
        self._steps_issued += int(n_steps)

    def policy_floats(self):
        """length of the packed MlpPolicy blob tb_policy_step expects for this env kind"""
        return int(self.L.tb_policy_floats(self.kind))

    def policy_step(self, weights, obs_in, seed=0, deterministic=False, out=None, policy_out=None):
        """step() with SB3's MlpPolicy evaluated inside the step kernel (tb_policy_step): each env
        acts on its row of `obs_in`. weights: the packed blob of `ppo.pack_policy`. Returns ((obs, reward, done), (actions, raw_actions, logp, value));
        `out` / `policy_out` are optional tuples of preallocated tensors of those shapes."""
        t, n = self.torch, self.num_envs
        nf = self.policy_floats()
        w = self._check_tensor(weights, (nf,), t.float32, "weights")
        if w.data_ptr() % 16:
            raise ValueError("weights must be 16-byte aligned")
        oi = self._check_tensor(obs_in, (n, self.obs_dim), t.float32, "obs_in")
        obs, rew, done = self._out() if out is None else out
        self._check_tensor(obs, (n, self.obs_dim), t.float32, "out[0]")
        self._check_tensor(rew, (n,), t.float32, "out[1]")
        self._check_tensor(done, (n,), t.uint8, "out[2]")
        if policy_out is None:
            policy_out = (t.empty((n, self.act_dim), dtype=t.float32, device=self.device), t.empty((n, self.act_dim), dtype=t.float32, device=self.device),
                          t.empty(n, dtype=t.float32, device=self.device), t.empty(n, dtype=t.float32, device=self.device))
        act, raw, logp, value = policy_out
        self._check_tensor(act, (n, self.act_dim), t.float32, "policy_out[0]")
        self._check_tensor(raw, (n, self.act_dim), t.float32, "policy_out[1]")
        self._check_tensor(logp, (n,), t.float32, "policy_out[2]")
        self._check_tensor(value, (n,), t.float32, "policy_out[3]")
        self.policy_step_ptrs(w.data_ptr(), oi.data_ptr(), act.data_ptr(), raw.data_ptr(), logp.data_ptr(), value.data_ptr(),
                              obs.data_ptr(), rew.data_ptr(), done.data_ptr(), seed, deterministic)
        if self.pipeline and out is None:
            self._inflight.append(rew)
        return (obs, rew, done), (act, raw, logp, value)

    def policy_step_ptrs(self, weights_ptr, obs_in_ptr, act_ptr, raw_ptr, logp_ptr, value_ptr, obs_ptr, reward_ptr, done_ptr, seed, deterministic=False):
        """unchecked fast path of policy_step (raw device addresses, validated once by the caller)"""
        rc = self.L.tb_policy_step(self._h, weights_ptr, obs_in_ptr, act_ptr, raw_ptr, logp_ptr, value_ptr, obs_ptr, reward_ptr, done_ptr,
                                   int(seed) & 0xFFFFFFFFFFFFFFFF, 1 if deterministic else 0, self.torch.cuda.current_stream(self.device).cuda_stream)
        if rc:
            _check(self.L, rc, "tb_policy_step")
        self._steps_issued += 1

    def mark(self, k):
        """progress mark k (tb_mark_record) at the current stream's position: a counter in pinned host memory goes
        up by one when the stream gets there; together with the library's count of finished fast-forwards it
        tells the host (mark_host_wait) that every step issued so far, late fast-forward writes included, is
        final in the caller's buffers. No stream waits for anything. Inside capture() it is a kernel node that
        every replay fires again."""
        _check(self.L, self.L.tb_mark_record(self._h, int(k), self._stream()), "tb_mark_record")

    def mark_enable(self, on=True):
        """count finished fast-forwards from now on (tb_mark_enable): needed before the steps that marks cover
        are issued or captured; off by default, so plain graphs carry nothing extra"""
        _check(self.L, self.L.tb_mark_enable(self._h, 1 if on else 0), "tb_mark_enable")

    def mark_begin(self):
        """snapshot of the mark counters: call right before launching the work that contains the marks, with
        nothing of this env in flight"""
        _check(self.L, self.L.tb_mark_begin(self._h), "tb_mark_begin")

    def mark_count(self, k):
        """how often mark k has fired so far (a host read)"""
        c = self.L.tb_mark_count(self._h, int(k))
        if c < 0:
            _check(self.L, int(c), "tb_mark_count")
        return int(c)

    def mark_host_wait(self, k, timeout_ms=10000):
        """spin on the host (GIL released) until mark k has fired since mark_begin() and the fast-forwards
        enqueued before it have finished"""
        _check(self.L, self.L.tb_mark_host_wait(self._h, int(k), int(timeout_ms)), "tb_mark_host_wait")

    def policy_rollout_ptrs(self, n_steps, weights_ptr, obs_in_ptr, act_ptr, raw_ptr, logp_ptr, value_ptr, obs_ptr, reward_ptr, done_ptr,
                            strides_bytes, seed, deterministic=False):
        """n_steps of policy_step_ptrs in as few launches as the episodes allow (tb_policy_rollout): the
        towers' weights and the envs' state stay in registers from step to step. strides_bytes: distance
        between consecutive steps of (actions, raw, logp, value, obs, reward, done), 0 = contiguous.
        Unchecked fast path; terminal SwingRacket rewards are complete after flush()."""
        st = (ctypes.c_size_t * 7)(*[int(x) for x in strides_bytes])
        rc = self.L.tb_policy_rollout(self._h, int(n_steps), weights_ptr, obs_in_ptr, act_ptr, raw_ptr, logp_ptr, value_ptr, obs_ptr, reward_ptr, done_ptr,
                                      st, int(seed) & 0xFFFFFFFFFFFFFFFF, 1 if deterministic else 0, self.torch.cuda.current_stream(self.device).cuda_stream)
        if rc:
            _check(self.L, rc, "tb_policy_rollout")
        self._steps_issued += int(n_steps)

    def policy_rollout(self, weights, obs_in, n_steps, seed=0, deterministic=False):
        """T = n_steps agent steps with the MlpPolicy inside the kernel, whole episodes per launch.
        Returns ((obs [T,N,O], reward [T,N], done [T,N]), (actions [T,N,A], raw [T,N,A], logp [T,N], value [T,N]))."""
        t, n, T = self.torch, self.num_envs, int(n_steps)
        w = self._check_tensor(weights, (self.policy_floats(),), t.float32, "weights")
        oi = self._check_tensor(obs_in, (n, self.obs_dim), t.float32, "obs_in")
        f = dict(dtype=t.float32, device=self.device)
        obs, rew, done = t.empty((T, n, self.obs_dim), **f), t.empty((T, n), **f), t.empty((T, n), dtype=t.uint8, device=self.device)
        act, raw, logp, value = t.empty((T, n, self.act_dim), **f), t.empty((T, n, self.act_dim), **f), t.empty((T, n), **f), t.empty((T, n), **f)
        self.policy_rollout_ptrs(T, w.data_ptr(), oi.data_ptr(), act.data_ptr(), raw.data_ptr(), logp.data_ptr(), value.data_ptr(),
                                 obs.data_ptr(), rew.data_ptr(), done.data_ptr(), (0,) * 7, seed, deterministic)
        if self.pipeline:
            self._inflight.append(rew)
        return (obs, rew, done), (act, raw, logp, value)

    def capture(self, fn):
        """Capture `fn()` -- a fixed sequence of step()/step_ptrs()/RolloutBuffer.step_into calls on
        fixed buffers -- into a HIP graph and return it as a StepGraph; `graph.replay()` then runs the
        whole sequence with one launch (no per-step host work). In pipelined mode the side-stream
        fast-forwards are captured as forked branches and joined by the final flush(). Nothing runs
        during the capture: the env (and the library's episode phase) are where they were, and the
        first replay() is the first time the steps happen."""
        t = self.torch
        if self.pipeline:
            self.flush()  # (also runs the pool of deferred stragglers, if any: nothing of this env may be pending across the capture boundary)
        t.cuda.current_stream(self.device).synchronize()
        _check(self.L, self.L.tb_pipeline_sync(self._h, 1), "tb_pipeline_sync")
        phase = self.L.tb_phase(self._h) if self.pipeline else -1
        gen = self.L.tb_params_generation(self._h)
        issued = self._steps_issued
        g = t.cuda.CUDAGraph()
        try:
            # thread_local: other threads (e.g. the RCCL watchdog of torch.distributed) may keep
            # issuing HIP calls while this thread captures
            with t.cuda.graph(g, capture_error_mode="thread_local"):
                fn()
                self.flush()
        except BaseException:
            # nothing captured ever ran; put the handle's streams and phase hint back (the env is intact)
            try:
                t.cuda.synchronize(self.device)
            except Exception:  # the capture error may surface once more through torch
                pass
            self.L.tb_pipeline_recover(self._h)
            self._steps_issued = issued
            raise
        _check(self.L, self.L.tb_pipeline_sync(self._h, 0), "tb_pipeline_sync")  # also puts the phase back: nothing ran
        n_steps, self._steps_issued = self._steps_issued - issued, issued
        return StepGraph(self, g, n_steps, phase, gen)

    def flush(self):
        """Pipelined mode: make the current stream wait until every outstanding fast-forward has
        written its step's reward / terminal observation / substep count."""
        _check(self.L, self.L.tb_flush(self._h, self._stream()), "tb_flush")
        self._inflight = []

    def rollout(self, actions):
        """T steps in one launch: actions [T, N, A] -> obs [T, N, O], reward [T, N], done [T, N]."""
        t = self.torch
        if actions.dim() != 3:
            raise ValueError("actions must be [T, N, A]")
        T = int(actions.shape[0])
        a = self._check_tensor(actions, (T, self.num_envs, self.act_dim), t.float32, "actions")
        obs, rew, done = self._out(T)
        _check(self.L, self.L.tb_rollout(self._h, T, a.data_ptr(), obs.data_ptr(), rew.data_ptr(), done.data_ptr(),
                                         None if self.pipeline else self._substeps.data_ptr(), self._stream()), "tb_rollout")
        self._steps_issued += T
        if self.pipeline:
            self._inflight.append(rew)  # terminal rewards are written late, from the side streams: flush() before reading
        return obs, rew, done

    def terminal_obs(self):
        """[N, O]: for envs whose episode ended in the latest step, its final observation."""
        if self._term is None:
            raise StepperError("terminal observations are only tracked with auto_reset=True")
        if self.pipeline:
            self.flush()
        return self._term

    def last_substeps(self):
        """int32 [N]: physics substeps executed by the latest step()/rollout() call."""
        if self.pipeline:
            raise StepperError("per-step substep counts are not tracked in pipelined mode (use counters()['substeps'])")
        return self._substeps

    def observe(self):
        """Current observation rebuilt from the state (no stepping)."""
        t = self.torch
        w, _ = self.get_state_words()
        f = w.view(t.float32)
        if self.kind == ENV_SWING:
            rows = [0, 1, 13, 14, 22, 23]
        else:
            rows = [0, 1, 2, 7, 8, 9, 13, 14, 15, 16, 17, 18]
        return f[rows].t().contiguous()

    def set_params(self, params):
        p = params.copy()
        p.flags = (p.flags | F_AUTO_RESET) if self.auto_reset else (p.flags & ~F_AUTO_RESET)
        _check(self.L, self.L.tb_set_params(self._h, ctypes.byref(p), self._stream()), "tb_set_params")
        self.params = p

    def set_racket_scale(self, scale):
        """tennisbot_env.py:213-215: takes effect at the next reset of each env (the reference
        rebuilds the racket with globalScaling=scale in reset(), :230-234); every env keeps the
        scale of its current episode in its own state word. One stream-ordered store into the
        device-resident parameter block (tb_set_racket_scale): graphs captured earlier see it."""
        _check(self.L, self.L.tb_set_racket_scale(self._h, float(scale), self._stream()), "tb_set_racket_scale")
        self.params.racket_scale = float(scale)

    def phase(self):
        """agent steps since the last common reset modulo 26, or -1 when the envs are not in lockstep"""
        return int(self.L.tb_phase(self._h))

    PIPELINE_FORMS = ("none", "slots", "slots+pool", "pool")

    def pipeline_form(self):
        """tb_pipeline_form as a word: "none" (fast-forward inside the 26th step), "slots" (one fast-forward kernel per episode end on a
        side stream), "slots+pool" (its stragglers deferred to the join), "pool" (every episode end parked, ONE fast-forward at the join)"""
        return self.PIPELINE_FORMS[int(self.L.tb_pipeline_form(self._h))]

    # ------------------------------------------------------------------ state save / restore
    def get_state_words(self):
        t = self.torch
        w = t.empty((self.words, self.num_envs), dtype=t.int32, device=self.device)
        d = t.empty(self.num_envs, dtype=t.uint8, device=self.device)
        _check(self.L, self.L.tb_get_state(self._h, w.data_ptr(), d.data_ptr(), 1, self._stream()), "tb_get_state")
        return w, d

    def set_state_words(self, words, done=None):
        t = self.torch
        w = words if isinstance(words, t.Tensor) else t.from_numpy(np.ascontiguousarray(words).view(np.int32))
        w = self._check_tensor(w.to(self.device).contiguous(), (self.words, self.num_envs), t.int32, "words")
        dptr = None
        if done is not None:
            d = done if isinstance(done, t.Tensor) else t.from_numpy(np.ascontiguousarray(done, np.uint8))
            d = self._check_tensor(d.to(self.device).contiguous(), (self.num_envs,), t.uint8, "done")
            dptr = d.data_ptr()
        _check(self.L, self.L.tb_set_state(self._h, w.data_ptr(), dptr, 1, self._stream()), "tb_set_state")
        self.torch.cuda.current_stream(self.device).synchronize()  # the source tensors may be temporaries

    def get_state(self):
        """dict of named numpy arrays (host copy): the env checkpoint the reference never had."""
        w, d = self.get_state_words()
        w = w.cpu().numpy().view(np.uint32)
        names = STATE_ROWS[self.kind]
        out, i = {}, 0
        fl = w.view(np.float32)
        while i < len(names):
            j = i
            while j < len(names) and names[j] == names[i]:
                j += 1
            out[names[i]] = np.ascontiguousarray(fl[i:j].T)
            i = j
        out["step_count"] = w[self.words - 2].view(np.int32).copy()
        out["episode"] = w[self.words - 1].copy()
        if "init_dist" in out:
            out["init_dist"] = out["init_dist"][:, 0]
        out["done"] = d.cpu().numpy()
        return out

    def save_state(self, path):
        """on-disk checkpoint of the env batch (npz): raw SoA words + done bytes + identity"""
        w, d = self.get_state_words()
        np.savez_compressed(path, words=w.cpu().numpy(), done=d.cpu().numpy(), env_kind=self.kind, num_envs=self.num_envs,
                            seed=self.seed, env_id_base=self.env_id_base, rows=np.array(STATE_ROWS[self.kind]))

    def load_state(self, path):
        z = np.load(path, allow_pickle=False)
        if int(z["env_kind"]) != self.kind or int(z["num_envs"]) != self.num_envs:
            raise ValueError("checkpoint is for env kind %d x %d envs" % (int(z["env_kind"]), int(z["num_envs"])))
        self.set_state_words(z["words"], z["done"])

    def counters(self):
        c = (ctypes.c_uint64 * N_COUNTERS)()
        _check(self.L, self.L.tb_counters(self._h, c, self._stream()), "tb_counters")
        return dict(zip(COUNTER_NAMES, [int(x) for x in c]))

    def sealed_substeps(self):
        """how many of counters()['substeps'] the pool's sealed-fate exit booked without running them (TbOptions.ff_seal)"""
        c = ctypes.c_uint64(0)
        _check(self.L, self.L.tb_sealed_substeps(self._h, ctypes.byref(c), self._stream()), "tb_sealed_substeps")
        return int(c.value)

    def counters_reset(self):
        _check(self.L, self.L.tb_counters_reset(self._h, self._stream()), "tb_counters_reset")

"""Rollout storage for the PPO collect boundary, sharded over GPUs.

Each rank steps its own contiguous block of envs (global env ids [rank*N, (rank+1)*N),
SURVEY.md 8e) and writes every step's obs / reward / done straight into one packed device
buffer (the step kernel's output pointers point into it: no copies). At the collect
boundary -- SB3's `collect_rollouts` end, the counterpart of the hook the reference uses
in train.py:164 -- the ranks exchange their shards with ONE all-gather (RCCL over xGMI:
`torch.distributed` backend "nccl" on ROCm; "gloo" in the CPU tests).

Packed layout per rank: T per-step records, each `obs f32 [N][O] | act f32 [N][A] | rew f32 [N] |
done u8 [N]` (every part padded to 16 B), so any range of steps is one contiguous byte range:
the whole buffer goes out in ONE all-gather, and begin_gather / gather_chunk / finish_gather can
instead ship it in a few step-chunks from a side stream while later steps are still being computed
(capture_marked keeps the rollout ONE hipGraph and tells the host when each chunk is final). n_steps
defaults to the reference's 1100 (train_swing.py:49-50).
"""
import numpy as np

from .params import ACT_DIM, OBS_DIM


def _align(x, a=16):
    return (x + a - 1) // a * a


class RolloutBuffer:
    def __init__(self, env_kind, n_steps, num_envs, device):
        import torch
        self.torch = torch
        self.T, self.N = int(n_steps), int(num_envs)
        self.O, self.A = OBS_DIM[env_kind], ACT_DIM[env_kind]
        N, O, A = self.N, self.O, self.A
        parts = [N * O * 4, N * A * 4, N * 4, N]
        self.part_offsets = np.cumsum([0] + [_align(x) for x in parts])  # within one step record
        self.record = int(self.part_offsets[-1])
        self.nbytes = self.record * self.T
        self.device = torch.device(device)
        self.raw = torch.zeros(self.nbytes, dtype=torch.uint8, device=self.device)
        self.obs, self.actions, self.rewards, self.dones = self.views(self.raw, self.T)

    def views(self, raw, n_steps):
        """typed [n_steps, N, ...] views into a packed run of step records (strided over steps,
        contiguous within a step)"""
        t, (N, O, A), off, rec = self.torch, (self.N, self.O, self.A), self.part_offsets, self.record
        f = raw.view(t.float32)
        fo, bo = f.storage_offset(), raw.storage_offset()  # as_strided offsets are absolute in the storage
        obs = f.as_strided((n_steps, N, O), (rec // 4, O, 1), fo + int(off[0]) // 4)
        act = f.as_strided((n_steps, N, A), (rec // 4, A, 1), fo + int(off[1]) // 4)
        rew = f.as_strided((n_steps, N), (rec // 4, 1), fo + int(off[2]) // 4)
        done = raw.as_strided((n_steps, N), (rec, 1), bo + int(off[3]))
        return obs, act, rew, done

    def bind(self, env):
        """validate this buffer against `env` once and cache the per-step device addresses, so the
        collect loop is one C call per step (no tensor slicing, no per-step checks)"""
        t = self.torch
        env._check_tensor(self.actions[0], (self.N, self.A), t.float32, "rollout actions[t]")
        env._check_tensor(self.obs[0], (self.N, self.O), t.float32, "rollout obs[t]")
        env._check_tensor(self.rewards[0], (self.N,), t.float32, "rollout rewards[t]")
        env._check_tensor(self.dones[0], (self.N,), t.uint8, "rollout dones[t]")
        base, off, rec = self.raw.data_ptr(), [int(x) for x in self.part_offsets], self.record
        self._ptrs = [(base + k * rec + off[1], base + k * rec + off[0], base + k * rec + off[2], base + k * rec + off[3])
                      for k in range(self.T)]
        self._bound = env
        return self

    def step_into(self, env, t):
        """run env.step on actions[t], with outputs written in place at slot t"""
        if getattr(self, "_bound", None) is env:
            env.step_ptrs(*self._ptrs[t])
            return None
        return env.step(self.actions[t], out=(self.obs[t], self.rewards[t], self.dones[t]))

    def step_range(self, env, t0, t1):
        """steps t0 .. t1-1 into their slots with ONE host call (tb_step_sequence): the record
        stride is the same for all four parts. Needs bind(env)."""
        if getattr(self, "_bound", None) is not env:
            raise ValueError("step_range needs bind(env) first")
        if not 0 <= t0 < t1 <= self.T:
            raise ValueError("step range [%d, %d) outside the buffer's %d steps" % (t0, t1, self.T))
        a, o, r, d = self._ptrs[t0]
        rec = self.record
        env.step_sequence_ptrs(t1 - t0, a, o, r, d, (rec, rec, rec, rec))

    def capture_marked(self, env, n_chunks):
        """The T steps of this buffer as ONE hipGraph with a progress mark (BatchedEnv.mark) after each of
        n_chunks step-chunks: the graph keeps its shape and speed -- a mark is one tiny kernel in the step
        stream, nothing forks or joins -- while replay_marked ships chunk c from a side stream as soon as
        the host sees mark c and the fast-forwards that chunk is owed."""
        if getattr(self, "_bound", None) is not env:
            raise ValueError("capture_marked needs bind(env) first")
        if self.T % n_chunks:
            raise ValueError("n_steps %d is not divisible into %d chunks" % (self.T, n_chunks))
        seg = self.T // n_chunks

        def body():
            for c in range(n_chunks):
                self.step_range(env, c * seg, (c + 1) * seg)
                env.mark(c)
        env.mark_enable(True)
        try:
            return env.capture(body)
        finally:
            env.mark_enable(False)  # what was captured keeps its counting kernels; later plain captures get none

    def replay_marked(self, graph, env, n_chunks, gather=False, force=False):
        """replay capture_marked's graph; gather=True (after begin_gather(n_chunks)): the host waits for each
        chunk's mark in turn and issues that chunk's all-gather on the gather stream while the graph goes on
        stepping. The wait is on the host on purpose (see tb_mark_host_wait): nothing queued on the device
        ever waits for the graph."""
        env.mark_begin()  # nothing of env is in flight here: the caller has joined the previous rollout
        graph.replay()
        if gather:
            for c in range(n_chunks):
                env.mark_host_wait(c)
                self.gather_chunk(c, force=force, after_mark=True)

    def _world(self, group):
        dist = self.torch.distributed
        if not (dist.is_available() and dist.is_initialized()):
            return 1
        return dist.get_world_size(group)

    def all_gather(self, group=None, force=False):
        """ONE collective for the whole rollout: returns a list of (obs, act, rew, done) views,
        one per rank, in rank order (= global env id order). Single process: the local shard
        (force=True issues the collective anyway: one-GPU rehearsal of the RCCL call)."""
        world = self._world(group)
        self._last_gather = "single"
        if world == 1 and not (force and self.torch.distributed.is_initialized()):
            return [(self.obs, self.actions, self.rewards, self.dones)]
        t = self.torch
        if getattr(self, "_gathered", None) is None or self._gathered.numel() != world * self.nbytes:
            self._gathered = t.empty(world * self.nbytes, dtype=t.uint8, device=self.device)
        if getattr(self, "rehearsal_total_cycles", 0):  # one-GPU rehearsal: stand in for the time a real exchange takes
            t.cuda._sleep(int(self.rehearsal_total_cycles))
        t.distributed.all_gather_into_tensor(self._gathered, self.raw, group=group)
        return [self.views(self._gathered[r * self.nbytes: (r + 1) * self.nbytes], self.T) for r in range(world)]

    # -- overlapped variant: the same bytes in a few step-chunks. Each chunk's all-gather is issued
    #    from a side stream that waits for (a) the chunk's step kernels and (b) the fast-forwards
    #    that still owe rewards to those steps -- so the stream that steps the envs never stalls:
    #    it neither waits for the fast-forwards nor for the collective. finish_gather() joins.
    def begin_gather(self, n_chunks, group=None, force=False, priority=-1):
        """priority: of the stream that carries the chunks' collectives. The runtime pools hardware queues per
        priority, so -1 (high) keeps a collective that runs for a while out of the queues the step kernels and
        the side-stream fast-forwards use; which setting overlaps best is measured, not assumed (bench.py tries both)."""
        if self.T % n_chunks:
            raise ValueError("n_steps %d is not divisible into %d chunks" % (self.T, n_chunks))
        world, t = self._world(group), self.torch
        self._chunks, self._chunk_steps, self._works, self._group = int(n_chunks), self.T // int(n_chunks), [], group
        self._last_gather = "chunks"
        cb = self._chunk_steps * self.record
        if (world > 1 or force) and (getattr(self, "_gath_chunks", None) is None or len(self._gath_chunks) != n_chunks
                                     or self._gath_chunks[0].numel() != world * cb):
            self._gath_chunks = [t.empty(world * cb, dtype=t.uint8, device=self.device) for _ in range(n_chunks)]
        if self.device.type == "cuda":
            streams = self.__dict__.setdefault("_gather_streams", {})
            if priority not in streams:
                streams[priority] = t.cuda.Stream(device=self.device, priority=int(priority))
            self._gather_stream = streams[priority]

    def gather_chunk(self, c, env=None, force=False, after_mark=None):
        """call after the steps [c*S, (c+1)*S) have been enqueued on the current stream. `env`: the
        pipelined BatchedEnv whose fast-forwards write late into this buffer (its flush is put on
        the gather stream, not on the caller's). force: issue the collective for a single rank too
        (rehearsal of the multi-rank path on one GPU). after_mark=True: the caller has already waited for the
        chunk's progress mark on the host (replay_marked): the gather stream starts at once."""
        world, t = self._world(self._group), self.torch
        if world == 1 and not force:
            return
        cb = self._chunk_steps * self.record
        src = self.raw[c * cb:(c + 1) * cb]
        gs = getattr(self, "_gather_stream", None)
        if gs is None:  # CPU tensors (gloo tests)
            self._works.append(t.distributed.all_gather_into_tensor(self._gath_chunks[c], src, group=self._group, async_op=True))
            return
        if not after_mark:
            gs.wait_stream(t.cuda.current_stream(self.device))
        with t.cuda.stream(gs):
            if env is not None and getattr(env, "pipeline", False):
                env.flush()  # makes `gs` wait for the outstanding fast-forwards
            if getattr(self, "rehearsal_total_cycles", 0):  # one-GPU rehearsal: stand in for the time a real exchange takes
                t.cuda._sleep(int(self.rehearsal_total_cycles) // self._chunks)
            self._works.append(t.distributed.all_gather_into_tensor(self._gath_chunks[c], src, group=self._group, async_op=True))

    def finish_gather(self):
        """wait for every chunk; returns, per rank, its list of per-chunk (obs, act, rew, done) views
        (no copy; `concatenated_chunks` builds [T, ...] tensors when a learner wants them)"""
        world, t = self._world(self._group), self.torch
        if not self._works:
            return [[(self.obs, self.actions, self.rewards, self.dones)]]
        for w in self._works:
            w.wait()  # the current stream waits for the collective
        self._works = []
        gs = getattr(self, "_gather_stream", None)
        if gs is not None:
            t.cuda.current_stream(self.device).wait_stream(gs)
        cb = self._chunk_steps * self.record
        return [[self.views(g[r * cb:(r + 1) * cb], self._chunk_steps) for g in self._gath_chunks] for r in range(world)]

    def check_gathered(self, group=None):
        """does the latest gather's output hold this rank's own shard, byte for byte? (a cheap
        self-test for callers that time the exchange without consuming it)"""
        t, dist = self.torch, self.torch.distributed
        rank = dist.get_rank(group) if (dist.is_available() and dist.is_initialized()) else 0
        if getattr(self, "_last_gather", None) == "chunks" and getattr(self, "_gath_chunks", None):
            cb = self._chunk_steps * self.record
            return all(t.equal(g[rank * cb:(rank + 1) * cb], self.raw[c * cb:(c + 1) * cb]) for c, g in enumerate(self._gath_chunks))
        if getattr(self, "_gathered", None) is not None:
            return t.equal(self._gathered[rank * self.nbytes:(rank + 1) * self.nbytes], self.raw)
        return True

    def concatenated_chunks(self, shards):
        """per-rank [T, N, ...] tensors from finish_gather()'s chunk views (this one copies)"""
        t = self.torch
        return [tuple(t.cat([p[k] for p in parts], dim=0) for k in range(4)) for parts in shards]

    def concatenated(self, shards):
        """[T, world*N, ...] tensors in global env id order (what a learner consumes)"""
        t = self.torch
        return tuple(t.cat([s[k] for s in shards], dim=1) for k in range(4))

"""Rollout storage for the PPO collect boundary, sharded over GPUs.

Each rank steps its own contiguous block of envs (global env ids [rank*N, (rank+1)*N),
SURVEY.md 8e) and writes every step's obs / reward / done straight into one packed device
buffer (the step kernel's output pointers point into it: no copies). At the collect
boundary -- SB3's `collect_rollouts` end, the counterpart of the hook the reference uses
in train.py:164 -- the ranks exchange their shards with ONE all-gather (RCCL over xGMI:
`torch.distributed` backend "nccl" on ROCm; "gloo" in the CPU tests).

Packed layout per rank: T per-step records, each `obs f32 [N][O] | act f32 [N][A] | rew f32 [N] |
done u8 [N]` (every part padded to 16 B), so any range of steps is one contiguous byte range:
the whole buffer goes out in ONE all-gather, and `all_gather_chunked` can instead ship it in a
few step-chunks on RCCL's stream while later steps are still being computed. n_steps defaults
to the reference's 1100 (train_swing.py:49-50).
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

    def _world(self, group):
        dist = self.torch.distributed
        if not (dist.is_available() and dist.is_initialized()):
            return 1
        return dist.get_world_size(group)

    def all_gather(self, group=None):
        """ONE collective for the whole rollout: returns a list of (obs, act, rew, done) views,
        one per rank, in rank order (= global env id order). Single process: the local shard."""
        world = self._world(group)
        if world == 1:
            return [(self.obs, self.actions, self.rewards, self.dones)]
        t = self.torch
        if getattr(self, "_gathered", None) is None or self._gathered.numel() != world * self.nbytes:
            self._gathered = t.empty(world * self.nbytes, dtype=t.uint8, device=self.device)
        t.distributed.all_gather_into_tensor(self._gathered, self.raw, group=group)
        return [self.views(self._gathered[r * self.nbytes: (r + 1) * self.nbytes], self.T) for r in range(world)]

    # -- overlapped variant: the same bytes in a few step-chunks, each issued as soon as its steps
    #    are enqueued (async on the collective's own stream), joined by finish_gather()
    def begin_gather(self, n_chunks, group=None):
        if self.T % n_chunks:
            raise ValueError("n_steps %d is not divisible into %d chunks" % (self.T, n_chunks))
        world, t = self._world(group), self.torch
        self._chunks, self._chunk_steps, self._works, self._group = int(n_chunks), self.T // int(n_chunks), [], group
        cb = self._chunk_steps * self.record
        if world > 1 and (getattr(self, "_gath_chunks", None) is None or len(self._gath_chunks) != n_chunks
                          or self._gath_chunks[0].numel() != world * cb):
            self._gath_chunks = [t.empty(world * cb, dtype=t.uint8, device=self.device) for _ in range(n_chunks)]

    def gather_chunk(self, c):
        """call after the steps [c*S, (c+1)*S) have been enqueued on the current stream"""
        world = self._world(self._group)
        if world == 1:
            return
        cb = self._chunk_steps * self.record
        src = self.raw[c * cb:(c + 1) * cb]
        self._works.append(self.torch.distributed.all_gather_into_tensor(self._gath_chunks[c], src, group=self._group, async_op=True))

    def finish_gather(self):
        """wait for every chunk; returns per-rank (obs, act, rew, done) over all T steps"""
        world, t = self._world(self._group), self.torch
        if world == 1:
            return [(self.obs, self.actions, self.rewards, self.dones)]
        for w in self._works:
            w.wait()
        cb = self._chunk_steps * self.record
        out = []
        for r in range(world):
            parts = [self.views(g[r * cb:(r + 1) * cb], self._chunk_steps) for g in self._gath_chunks]
            out.append(tuple(t.cat([p[k] for p in parts], dim=0) for k in range(4)))
        return out

    def concatenated(self, shards):
        """[T, world*N, ...] tensors in global env id order (what a learner consumes)"""
        t = self.torch
        return tuple(t.cat([s[k] for s in shards], dim=1) for k in range(4))

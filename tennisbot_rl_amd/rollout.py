"""Rollout storage for the PPO collect boundary, sharded over GPUs.

Each rank steps its own contiguous block of envs (global env ids [rank*N, (rank+1)*N),
SURVEY.md 8e) and writes every step's obs / reward / done straight into one packed device
buffer (the step kernel's output pointers point into it: no copies). At the collect
boundary -- SB3's `collect_rollouts` end, the counterpart of the hook the reference uses
in train.py:164 -- the ranks exchange their shards with ONE all-gather (RCCL over xGMI:
`torch.distributed` backend "nccl" on ROCm; "gloo" in the CPU tests).

Packed layout per rank (bytes): obs f32 [T][N][O] | act f32 [T][N][A] | rew f32 [T][N] | done u8 [T][N],
padded to 16 B. n_steps defaults to the reference's 1100 (train_swing.py:49-50).
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
        T, N, O, A = self.T, self.N, self.O, self.A
        sizes = [T * N * O * 4, T * N * A * 4, T * N * 4, T * N]
        self.offsets = np.cumsum([0] + [_align(s) for s in sizes])
        self.nbytes = int(self.offsets[-1])
        self.device = torch.device(device)
        self.raw = torch.zeros(self.nbytes, dtype=torch.uint8, device=self.device)
        self.obs, self.actions, self.rewards, self.dones = self.views(self.raw)

    def views(self, raw):
        """typed views into one packed shard"""
        t, (T, N, O, A), off = self.torch, (self.T, self.N, self.O, self.A), self.offsets
        obs = raw[off[0]: off[0] + T * N * O * 4].view(t.float32).view(T, N, O)
        act = raw[off[1]: off[1] + T * N * A * 4].view(t.float32).view(T, N, A)
        rew = raw[off[2]: off[2] + T * N * 4].view(t.float32).view(T, N)
        done = raw[off[3]: off[3] + T * N].view(T, N)
        return obs, act, rew, done

    def bind(self, env):
        """validate this buffer against `env` once and cache the per-step device addresses, so the
        collect loop is one C call per step (no tensor slicing, no per-step checks)"""
        t = self.torch
        env._check_tensor(self.actions[0], (self.N, self.A), t.float32, "rollout actions[t]")
        env._check_tensor(self.obs[0], (self.N, self.O), t.float32, "rollout obs[t]")
        env._check_tensor(self.rewards[0], (self.N,), t.float32, "rollout rewards[t]")
        env._check_tensor(self.dones[0], (self.N,), t.uint8, "rollout dones[t]")
        if self.N * self.O * 4 % env._row_align or self.N * self.A * 4 % env._row_align:
            raise ValueError("rollout rows are not %d-byte aligned for this batch size" % env._row_align)
        N, O, A = self.N, self.O, self.A
        self._ptrs = [(self.actions.data_ptr() + k * N * A * 4, self.obs.data_ptr() + k * N * O * 4,
                       self.rewards.data_ptr() + k * N * 4, self.dones.data_ptr() + k * N) for k in range(self.T)]
        self._bound = env
        return self

    def step_into(self, env, t):
        """run env.step on actions[t], with outputs written in place at slot t"""
        if getattr(self, "_bound", None) is env:
            env.step_ptrs(*self._ptrs[t])
            return None
        return env.step(self.actions[t], out=(self.obs[t], self.rewards[t], self.dones[t]))

    def all_gather(self, group=None):
        """one collective: returns a list of (obs, act, rew, done) views, one per rank, in
        rank order (= global env id order). Single process: returns the local shard."""
        dist = self.torch.distributed
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            return [(self.obs, self.actions, self.rewards, self.dones)]
        world = dist.get_world_size(group)
        if getattr(self, "_gathered", None) is None or self._gathered.numel() != world * self.nbytes:
            self._gathered = self.torch.empty(world * self.nbytes, dtype=self.torch.uint8, device=self.device)
        dist.all_gather_into_tensor(self._gathered, self.raw, group=group)
        return [self.views(self._gathered[r * self.nbytes: (r + 1) * self.nbytes]) for r in range(world)]

    def concatenated(self, shards):
        """[T, world*N, ...] tensors in global env id order (what a learner consumes)"""
        t = self.torch
        return tuple(t.cat([s[k] for s in shards], dim=1) for k in range(4))

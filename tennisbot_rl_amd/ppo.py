"""On-device PPO for the batched envs: the caller of the hot path.

Counterpart of what `train_swing.py:80-122` / `train.py:104-151` obtain from
stable-baselines3 (absent here; the learner itself is out of scope):
same policy architecture and hyper-parameters, but collection never leaves the GPU --
observations, actions, rewards and dones live in the rank's `RolloutBuffer`, the step kernel
writes into it in place, the SwingRacket fast-forward overlaps on side streams, and with
several ranks the shards are exchanged by ONE all-gather per rollout.

The network is torch.nn (6 -> 32 -> 64 -> 32 -> A, tanh; library GEMMs): it is 10^4 flops
per env-step, plumbing around the stepper, not a kernel this repository optimises.
"""
import math
import time

import numpy as np

from .params import ACT_DIM, ENV_SWING, OBS_DIM
from .rollout import RolloutBuffer
from .stepper import ENV_IDS, BatchedEnv, StepperError

# hyper-parameters of the reference scripts (and SB3 1.8.0 defaults where they are silent)
SWING_DEFAULTS = dict(net_arch=(32, 64, 32), ent_coef=0.002, learning_rate=3e-4)   # train_swing.py:80-91
TENNIS_DEFAULTS = dict(net_arch=(64, 64), ent_coef=0.01, learning_rate=3e-4)        # train.py:4-33,104-110
COMMON = dict(gamma=0.99, gae_lambda=0.95, clip_range=0.2, vf_coef=0.5, max_grad_norm=0.5, n_epochs=10)


def build_actor_critic(obs_dim, act_dim, net_arch=(32, 64, 32)):
    """SB3 `MlpPolicy` with separate pi / vf towers (net_arch=dict(pi=..., vf=...),
    train_swing.py:80-82): tanh MLPs, linear action mean, state-independent log_std."""
    import torch
    from torch import nn

    class ActorCritic(nn.Module):
        def __init__(self):
            super().__init__()

            def tower():
                layers, d = [], obs_dim
                for h in net_arch:
                    layers += [nn.Linear(d, h), nn.Tanh()]
                    d = h
                return nn.Sequential(*layers)
            self.policy_net, self.value_net_body = tower(), tower()
            self.action_net = nn.Linear(net_arch[-1], act_dim)
            self.value_net = nn.Linear(net_arch[-1], 1)
            self.log_std = nn.Parameter(torch.zeros(act_dim))
            for m in list(self.policy_net) + list(self.value_net_body):  # SB3 ortho_init gains
                if isinstance(m, nn.Linear):
                    nn.init.orthogonal_(m.weight, gain=math.sqrt(2)); nn.init.zeros_(m.bias)
            nn.init.orthogonal_(self.action_net.weight, gain=0.01); nn.init.zeros_(self.action_net.bias)
            nn.init.orthogonal_(self.value_net.weight, gain=1.0); nn.init.zeros_(self.value_net.bias)

        def forward(self, obs):
            return self.action_net(self.policy_net(obs)), self.value_net(self.value_net_body(obs)).squeeze(-1)

        def act(self, obs, deterministic=False):
            mean, value = self(obs)
            std = self.log_std.exp()
            a = mean if deterministic else mean + std * torch.randn_like(mean)
            logp = (-0.5 * ((a - mean) / std) ** 2 - self.log_std - 0.5 * math.log(2 * math.pi)).sum(-1)
            return a, value, logp

        def evaluate(self, obs, actions):
            mean, value = self(obs)
            std = self.log_std.exp()
            logp = (-0.5 * ((actions - mean) / std) ** 2 - self.log_std - 0.5 * math.log(2 * math.pi)).sum(-1)
            entropy = (0.5 + 0.5 * math.log(2 * math.pi) + self.log_std).sum()
            return value, logp, entropy

        def load_sb3_arrays(self, arrays):
            """weights in SB3's `policy.pth` naming (as exported by tools/export_reference_policy.py)"""
            sd = {}
            for k, v in arrays.items():
                k = k.replace("__", ".")
                k = k.replace("mlp_extractor.policy_net.", "policy_net.").replace("mlp_extractor.value_net.", "value_net_body.")
                sd[k] = torch.as_tensor(np.asarray(v))
            self.load_state_dict(sd)
            return self

    return ActorCritic()


def _fragment_indices(n_in, n_out, first_layer):
    """Gather indices into [bias (n_out), W^T (n_in x n_out) row-major, one trailing 0.0] that produce
    one layer of the blob `tb_policy_step` reads (include/tb_stepper.h; csrc/tb_policy.hpp, LayerRegs): bias tiles
    [out/16][4 lane groups][4 regs], then weight fragments [out/16][chunks][64 lanes] of v_mfma_f32_16x16x4_f32 -- lane l of
    a fragment holds W[out 16 t + l % 16][k(chunk, l // 16)]. The first layer takes k in natural order, k = 4 c + g (zero
    beyond the observation); every later layer takes it in the order the previous layer's accumulator registers hold it:
    chunk 4 u + r = register r of output tile u, whose lane group g holds row 16 u + 4 g + r."""
    zero = n_out + n_in * n_out
    if first_layer:
        chunks = [[4 * c + g for g in range(4)] for c in range((n_in + 3) // 4)]
    else:
        assert n_in % 16 == 0
        chunks = [[16 * u + 4 * g + r for g in range(4)] for u in range(n_in // 16) for r in range(4)]
    n_tiles = (n_out + 15) // 16
    idx = []
    for t in range(n_tiles):
        for g in range(4):
            for r in range(4):
                o = 16 * t + 4 * g + r
                idx.append(o if o < n_out else zero)
    for t in range(n_tiles):
        for ks in chunks:
            for g in range(4):
                for j in range(16):
                    o, k = 16 * t + j, ks[g]
                    idx.append(n_out + k * n_out + o if (o < n_out and k < n_in) else zero)
    return idx


def pack_policy(policy, out=None):
    """The blob `tb_policy_step` reads: pi tower layers, action head, vf tower layers, value head (each
    in MFMA fragment order, see _fragment_indices), then log_std padded to a multiple of 4 floats.
    `out`: a preallocated device tensor to refresh in place (its address is baked into captured graphs).
    ONE gather: the parameters are concatenated as they lie in memory ([bias, weight (out x in) row-major] per layer, log_std,
    one 0.0) and a cached index vector -- _fragment_indices composed with the weight's transpose -- picks the blob out of that
    (two kernels per call; the per-layer cat / transpose / index form was ~40 launches, 0.2 ms of a 4.5 ms collect)."""
    import torch
    dev = policy.log_std.device
    layers = []
    for body, head in ((policy.policy_net, policy.action_net), (policy.value_net_body, policy.value_net)):
        layers += [m for m in body if isinstance(m, torch.nn.Linear)] + [head]
    cache = getattr(policy, "_pack_index", None)
    if cache is None or cache.device != dev:
        n_body = (len(layers) - 2) // 2
        total = sum(m.out_features * (m.in_features + 1) for m in layers) + policy.log_std.numel()  # ... and the 0.0 lies at `total`
        gidx, off = [], 0
        for li, m in enumerate(layers):
            n_in, n_out = m.in_features, m.out_features
            for k in _fragment_indices(n_in, n_out, li % (n_body + 1) == 0):
                if k < n_out:                      # bias
                    gidx.append(off + k)
                elif k < n_out + n_in * n_out:     # W^T[i][o] = weight[o][i]
                    i, o = divmod(k - n_out, n_out)
                    gidx.append(off + n_out + o * n_in + i)
                else:
                    gidx.append(total)
            off += n_out * (n_in + 1)
        gidx += [off + k for k in range(policy.log_std.numel())] + [total] * (-policy.log_std.numel() % 4)
        cache = torch.tensor(gidx, dtype=torch.long, device=dev)
        policy._pack_index = cache
    with torch.no_grad():
        src = torch.cat([t.detach().float().reshape(-1) for m in layers for t in (m.bias, m.weight)] + [policy.log_std.detach().float().reshape(-1), torch.zeros(1, device=dev)])
        if out is None:
            return src.index_select(0, cache)
        torch.index_select(src, 0, cache, out=out)
    return out


class PPOTrainer:
    """clipped-surrogate PPO over a BatchedEnv; one process per GPU when distributed"""

    def __init__(self, env_id="SwingRacket-v0", num_envs=4096, n_steps=104, device=None, seed=0, batch_size=None,
                 pipeline=True, graph=True, fused=True, rollout_launch=True, params=None, ff_defer="all", options=None, **hp):
        import torch
        self.torch = torch
        kind = ENV_IDS[env_id]
        d = dict(SWING_DEFAULTS if kind == ENV_SWING else TENNIS_DEFAULTS)
        d.update(COMMON)
        d.update(hp)
        self.hp = d
        dist = torch.distributed
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank() if self.world > 1 else 0
        # params: a TbParams (default_params(flags=..., **overrides)), e.g. the reference's full contact set
        # (TB_F_RACKET_GROUND, rolling friction): the fused policy kernels are instantiated for it too
        # options: further TbOptions fields for the env (make_options: e.g. policy_slices)
        # ff_defer: a trained policy's struck balls fly 300-775 substeps, and at most four fast-forward kernels run at once: the
        # collect was bound by them (229 M env steps/s with the reference's policy). "all": every episode end of a rollout is
        # parked into one pool that a single launch finishes at the join -- the rollout kernels run undisturbed, the long flights
        # side by side (TbOptions.ff_defer = 2; same results). Up to 64 episodes per join: beyond, the ordinary path takes over.
        self.env = BatchedEnv(kind, num_envs, device=device, seed=seed, env_id_base=self.rank * num_envs, params=params,
                              track_terminal_obs=False, pipeline=pipeline and kind == ENV_SWING, options=dict({"ff_defer": ff_defer}, **(options or {})))  # (an explicit options['ff_defer'] wins over the keyword's default)
        self.device = self.env.device
        self.n_steps, self.num_envs = int(n_steps), int(num_envs)
        self.buf = RolloutBuffer(kind, self.n_steps, num_envs, self.device).bind(self.env)
        torch.manual_seed(seed)  # identical initial weights on every rank
        self.policy = build_actor_critic(OBS_DIM[kind], ACT_DIM[kind], tuple(d["net_arch"])).to(self.device)
        torch.manual_seed(seed + 1000 * (self.rank + 1))  # ... but rank-local exploration noise
        self.opt = torch.optim.Adam(self.policy.parameters(), lr=d["learning_rate"], eps=1e-5)
        # minibatches of <= 65536 rows. (The reference's batch_size = n_steps = 1100 is its whole 1-env rollout, ONE minibatch per
        # epoch; with 4096 envs per rollout that many rows per gradient step learns slower per timestep -- measured: reward 17.5
        # instead of 26.6 after 26 M timesteps -- although the update then takes 0.08 s instead of 0.17 s. Replaying the gradient
        # step as one hipGraph was measured too: 0.17 s either way, the step is bound by ~150 microsecond-sized kernels, not by
        # their launches.)
        self.batch_size = batch_size or min(self.n_steps * num_envs, 65536)
        self.values = torch.zeros((self.n_steps, num_envs), device=self.device)
        self.logps = torch.zeros((self.n_steps, num_envs), device=self.device)
        self.obs_seq = torch.zeros((self.n_steps, num_envs, self.env.obs_dim), device=self.device)
        self._raw_actions = torch.zeros((self.n_steps, num_envs, self.env.act_dim), device=self.device)
        self.last_value = torch.zeros(num_envs, device=self.device)
        self.obs_in = self.env.reset().clone()  # static input of the (captured) rollout
        self.use_graph, self._graph = bool(graph), None
        if self.use_graph and self.env.pipeline and self.n_steps % 26:
            # a pipelined SwingRacket graph bakes in which of its steps end an episode: it can only be replayed back
            # to back when the rollout is a whole number of 26-step episodes. The reference's n_steps = 1100
            # (train_swing.py:49-50) is not: such rollouts are issued eagerly -- with whole episodes per launch
            # (tb_policy_rollout) that is 43 launches + 43 fast-forwards per 1100 steps, nothing a graph would save
            self.use_graph = False
        self.num_timesteps = 0
        self._aux_stream = None  # the collect's bookkeeping, beside the join (see _collect_fused)
        # fused=True: the policy runs inside the step kernel (tb_policy_step); the torch module is
        # then only the learner's view of the same weights, repacked once per rollout
        self.fused = bool(fused) and tuple(d["net_arch"]) == tuple((SWING_DEFAULTS if kind == ENV_SWING else TENNIS_DEFAULTS)["net_arch"])
        self.noise_seed = (seed * 1000003 + 7919 * (self.rank + 1)) & 0xFFFFFFFF
        # whole episodes per launch need the pipelined fast-forward on SwingRacket (always on for Tennisbot)
        self.rollout_launch = bool(rollout_launch) and self.fused and (kind != ENV_SWING or self.env.pipeline)
        self._rollouts = 0
        if self.fused:
            self.packed = pack_policy(self.policy)
            assert self.packed.numel() == self.env.policy_floats()

    # ------------------------------------------------------------------ collect
    def _collect_fused(self):
        """the rollout through the fused policy+step kernels (plus the side-stream fast-forwards): no
        library GEMM, no sampling kernels, nothing between two env steps. rollout_launch=True: whole
        episodes per launch (tb_policy_rollout: weights and env state resident in registers, no launch
        boundary between steps); otherwise one launch per step (tb_policy_step). Same results."""
        buf, env = self.buf, self.env
        n, O, A = self.num_envs, env.obs_dim, env.act_dim
        wp, cur = self.packed.data_ptr(), self.obs_in.data_ptr()
        self.obs_seq[0].copy_(self.obs_in)
        # whole-episode launches need the library to know the episode phase (every env restored / reset in lockstep)
        by_launch = self.rollout_launch and (not env.pipeline or env.phase() >= 0)
        if by_launch:
            rec = buf.record
            env.policy_rollout_ptrs(self.n_steps, wp, cur, buf.actions[0].data_ptr(), self._raw_actions.data_ptr(), self.logps.data_ptr(),
                                    self.values.data_ptr(), buf.obs[0].data_ptr(), buf.rewards[0].data_ptr(), buf.dones[0].data_ptr(),
                                    (rec, 0, 0, 0, rec, rec, rec), self.noise_seed)
        for k in range(0 if by_launch else self.n_steps):
            obs_k = buf.obs[k]
            # exploration noise is keyed by (seed, env, episode, step) inside the kernel: replaying
            # the captured graph draws fresh noise because episodes / steps advance
            env.policy_step_ptrs(wp, cur, buf.actions[k].data_ptr(), self._raw_actions[k].data_ptr(), self.logps[k].data_ptr(),
                                 self.values[k].data_ptr(), obs_k.data_ptr(), buf.rewards[k].data_ptr(), buf.dones[k].data_ptr(), self.noise_seed)
            cur = obs_k.data_ptr()
        # observations are final when the step kernels are; only terminal rewards are still missing. The join (in the pool form: ONE
        # launch over every episode end of the rollout, ~0.5-0.8 ms of the main stream) and the bookkeeping -- a dozen small torch
        # kernels -- run side by side: the bookkeeping on a stream of its own, forked behind the rollout kernels and joined at the end
        t = self.torch
        main = t.cuda.current_stream(self.device)
        if self._aux_stream is None:
            self._aux_stream = t.cuda.Stream(device=self.device)
        aux = self._aux_stream
        aux.wait_stream(main)
        env.flush()
        with t.cuda.stream(aux):
            self.obs_seq[1:].copy_(buf.obs[:-1])
            last = buf.obs[self.n_steps - 1]
            self.last_value.copy_(self.policy(last)[1])
            self.obs_in.copy_(last)
        main.wait_stream(aux)

    def _collect_body(self):
        t = self.torch
        if self.fused:
            return self._collect_fused()
        buf, env = self.buf, self.env
        cur = self.obs_in
        for k in range(self.n_steps):
            a, v, lp = self.policy.act(cur)
            buf.actions[k].copy_(a.clamp(-1.0, 1.0))  # SB3 clips Box actions before env.step
            self.values[k], self.logps[k] = v, lp
            # log-prob is of the unclipped sample, as in SB3; the buffer keeps what the env saw
            self._raw_actions[k] = a
            buf.step_into(env, k)
            self.obs_seq[k] = cur
            cur = buf.obs[k]
        env.flush()  # terminal rewards of pipelined fast-forwards are in place from here on
        self.last_value.copy_(self.policy(cur)[1])
        self.obs_in.copy_(cur)

    def collect(self):
        """n_steps of every env into the rollout buffer, entirely on the device. With graph=True
        the whole rollout -- policy forward, sampling, env step, side-stream fast-forwards -- is
        captured once as a hipGraph and replayed (fixed buffers, in-place parameter updates)."""
        t = self.torch
        if self.fused:
            pack_policy(self.policy, out=self.packed)
        with t.no_grad():
            if self._graph is not None and not self._graph.valid():
                self._graph = None  # the envs are at another episode phase (load(), evaluate()) or set_params() ran: capture again
            if self._graph is not None:
                self._graph.replay()
            else:
                self._collect_body()  # a real rollout on the real stream; its errors are the caller's to see
                if self.use_graph:
                    t.cuda.synchronize(self.device)
                    try:
                        self._graph = self.env.capture(self._collect_body)  # runs nothing; replayed from the next collect on
                    except (StepperError, RuntimeError) as exc:  # the capture itself failed: the rollout above stands, go on eagerly
                        print("hipGraph capture of the rollout failed (%s); collecting eagerly" % exc)
                        self.use_graph = False
        self.num_timesteps += self.n_steps * self.num_envs * self.world
        return self.last_value

    def advantages(self, last_value):
        t, hp = self.torch, self.hp
        rew, done = self.buf.rewards, self.buf.dones.float()
        adv = t.zeros_like(rew)
        gae = t.zeros(self.num_envs, device=self.device)
        for k in reversed(range(self.n_steps)):
            nonterminal = 1.0 - done[k]
            next_value = last_value if k == self.n_steps - 1 else self.values[k + 1]
            # auto-reset: after a done the next stored value belongs to a new episode -> no bootstrap
            delta = rew[k] + hp["gamma"] * next_value * nonterminal - self.values[k]
            gae = delta + hp["gamma"] * hp["gae_lambda"] * nonterminal * gae
            adv[k] = gae
        return adv, adv + self.values

    # ------------------------------------------------------------------ update
    def update(self, adv, returns):
        t, hp = self.torch, self.hp
        dist = t.distributed
        n = self.n_steps * self.num_envs
        obs = self.obs_seq.reshape(n, -1); act = self._raw_actions.reshape(n, -1)
        old_lp = self.logps.reshape(n); adv = adv.reshape(n); returns = returns.reshape(n)
        stats = {}
        for epoch in range(hp["n_epochs"]):
            perm = t.randperm(n, device=self.device)
            for s in range(0, n, self.batch_size):
                idx = perm[s:s + self.batch_size]
                a = adv[idx]
                a = (a - a.mean()) / (a.std() + 1e-8)
                value, logp, entropy = self.policy.evaluate(obs[idx], act[idx])
                ratio = (logp - old_lp[idx]).exp()
                pg = -t.min(a * ratio, a * ratio.clamp(1 - hp["clip_range"], 1 + hp["clip_range"])).mean()
                vl = ((returns[idx] - value) ** 2).mean()
                loss = pg + hp["vf_coef"] * vl - hp["ent_coef"] * entropy
                self.opt.zero_grad(set_to_none=True)
                loss.backward()
                if self.world > 1:  # data-parallel: average gradients over ranks (RCCL all-reduce)
                    for p in self.policy.parameters():
                        dist.all_reduce(p.grad)
                        p.grad /= self.world
                t.nn.utils.clip_grad_norm_(self.policy.parameters(), hp["max_grad_norm"])
                self.opt.step()
            stats = {"policy_loss": float(pg.detach()), "value_loss": float(vl.detach()), "entropy": float(entropy.detach())}
        return stats

    def learn(self, total_timesteps, log=print, gather_rollouts=True):
        history = []
        while self.num_timesteps < total_timesteps:
            t0 = time.perf_counter()
            last_value = self.collect()
            shards = None
            if gather_rollouts and self.world > 1:
                # the collect-boundary exchange of SURVEY.md 8e (one collective). What consumes it HERE: the statistics below are
                # the whole node's, as the reference's single-process learner reports them (ep_rew_mean over all its envs) -- the
                # update itself stays data-parallel on the rank's own shard with gradient all-reduces (it is 300 x the collect's
                # time, and every rank repeating it on the gathered batch would throw the other GPUs away); a single-learner design
                # would take RolloutBuffer.concatenated(shards) instead. gather_rollouts=False skips the exchange.
                shards = self.buf.all_gather()
            self.torch.cuda.synchronize(self.device)
            t1 = time.perf_counter()
            adv, returns = self.advantages(last_value)
            stats = self.update(adv, returns)
            self.torch.cuda.synchronize(self.device)
            t2 = time.perf_counter()
            c = self.env.counters()
            if c["nonfinite_states"] or c["lockstep_violations"]:
                raise StepperError("rollout %d: %d env states went non-finite, %d episode ends fell outside the pipelined launches' slots "
                                   "(their terminal rewards are lost)" % (len(history), c["nonfinite_states"], c["lockstep_violations"]))
            if shards is None:
                ep, rew_sum = float(self.buf.dones.sum()), float(self.buf.rewards.sum())
            else:  # every rank holds every shard: global figures without another collective
                ep, rew_sum = sum(float(sh[3].sum()) for sh in shards), sum(float(sh[2].sum()) for sh in shards)
            stats.update(timesteps=self.num_timesteps, episodes=ep,
                         mean_episode_reward=rew_sum / max(ep, 1.0),
                         collect_steps_per_s=self.n_steps * self.num_envs * self.world / (t1 - t0), update_s=t2 - t1)
            history.append(stats)
            if log and self.rank == 0:
                log("timesteps %10d  episodes %7d  mean episode reward %8.3f  collect %.1f M steps/s  update %.2f s"
                    % (stats["timesteps"], ep, stats["mean_episode_reward"], stats["collect_steps_per_s"] / 1e6, stats["update_s"]))
        return history

    def evaluate(self, n_episodes_steps=26, deterministic=False):
        """EvalCallback counterpart (train_swing.py:111-114, deterministic=False there): mean
        episode reward over one more rollout of the same envs"""
        t = self.torch
        total = t.zeros((), device=self.device)
        eps = t.zeros((), device=self.device)
        with t.no_grad():
            for k in range(n_episodes_steps):
                a, _, _ = self.policy.act(self.obs_in, deterministic=deterministic)
                obs, r, d = self.env.step(a.clamp(-1.0, 1.0))
                self.env.flush()
                self.obs_in.copy_(obs)
                total += r.sum(); eps += d.float().sum()
        return float(total) / max(float(eps), 1.0)

    def save(self, path):
        """policy + optimizer + the env batch itself (the reference checkpoints only the learner,
        train_swing.py:115-117; SURVEY.md section 5 asks for env state as well)"""
        w, d = self.env.get_state_words()
        self.torch.save({"policy": self.policy.state_dict(), "optimizer": self.opt.state_dict(), "num_timesteps": self.num_timesteps,
                         "env_words": w.cpu(), "env_done": d.cpu(), "hp": self.hp}, path)

    def load(self, path):
        ck = self.torch.load(path, map_location=self.device, weights_only=True)
        self.policy.load_state_dict(ck["policy"]); self.opt.load_state_dict(ck["optimizer"])
        self.num_timesteps = int(ck["num_timesteps"])
        self.env.set_state_words(ck["env_words"].to(self.device), ck["env_done"].to(self.device))
        self.obs_in.copy_(self.env.observe())
        return self

#!/usr/bin/env python3
"""Headline benchmark: env steps/sec (whole node), SwingRacket-v0 @ 4096 envs/GPU.

  python bench.py --gpus N --steps K --warmup W
  N > 1: either under the launcher (python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py
  --gpus N ...: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment), or plain `python bench.py --gpus N`:
  the process then IS the launcher (self_launch: it starts the N ranks as children before anything touches a GPU,
  relays rank 0's one JSON line and exits with the children's status).

A "step" is one agent-level env.step() of every env of the batch (BASELINE.json metric):
one launch of the HIP step kernel over N envs, i.e. 1 physics substep for agent steps 1-25 of an
episode and 1 + (up to 775) substeps on the 26th (the fast-forward of swingracket_env.py:105-141,
~80 % of the workload's substeps), plus the in-kernel auto-reset.

WHAT IS TIMED. The cost of a SwingRacket step is only defined over whole 26-step episodes, and the
fast-forward of a rollout's last episode (its "tail") is part of the rollout. The unit of work is
therefore one ROLLOUT: --rollout-steps agent steps (default 1040 = 40 episodes; the reference collects
n_steps = 1100, train_swing.py:49-50) written in place into the rank's rollout buffer, ended by the join
of every outstanding fast-forward, and -- with N > 1 -- the exchange of the rollout shards that the PPO
collect boundary needs (RCCL all-gather over xGMI). `--steps K` is rounded UP to whole rollouts:
steps_timed = rollout_steps * ceil(K / rollout_steps) (so K = 20 times 1040 steps, not 20 one-substep
steps between two episode ends), and to as many more whole rollouts as it takes to time --min-timed-ms
(50 ms: the replay rate of a process wanders by +-5 % over seconds and is 8 % lower during its first
second about every second time -- hence also --settle-seconds of untimed replays first); `steps` echoes K. The episode phase is aligned (warm-up rounded up to
whole episodes) and the line carries substeps_per_agent_step; a value whose substeps per step are not
the workload's (within 5 %) is refused (value = null, "invalid" says why).
Inputs (state, synthetic U(-1,1) actions from PCG64) are resident in HBM before the timed region.

Also in that line:
  roofline     -- HBM roofline of the step kernel: algorithmic bytes per launch (267 B/env Swing,
                  263 B/env Tennisbot: DESIGN.md) / average launch duration from HIP events on the
                  launch stream; `read_only` prices the same launch with the 145 / 117 B a step READS.
                  Launch- and ALU-latency-bound at 4096 envs; `sweep` shows 4096 and
                  1 M envs for both envs (--sweep: the whole 4096 .. 4 M ladder).
  cpu_baseline -- the float32 CPU oracle (a port, not PyBullet, which is not installable here) timed on
                  this host's cores on a bounded sample of the same workload (whole episodes: the same
                  substeps per step), and `reference_record`: the wall-clock the reference's own PyBullet
                  training run recorded (tests/golden/ppo_swing_reference_episodes.json).
  exchange     -- N > 1: ranks_seen, bytes per rank, rollout_ms / exchange_ms measured apart, and what
                  of the exchange stayed exposed in the timed region.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)
ALGO_BYTES = {"swing": {"read": 145, "write": 122}, "tennis": {"read": 117, "write": 146}}  # SURVEY.md 8d / DESIGN.md
# physics substeps per agent step of the random-action workload (whole episodes): what a valid SwingRacket
# measurement must show within 5 %; Tennisbot is 1 by construction. Measured over >= 4096 x 1040 steps
# (profiles/r02*), the same on the CPU oracle; keyed by racket<->court contact on/off.
EXPECTED_SUBSTEPS = {("swing", True): None, ("swing", False): 5.09, ("tennis", True): 1.0, ("tennis", False): 1.0}


def pmc_traffic(env_name, n_envs):
    """HBM bytes per launch of the step kernel from the committed rocprofv3 PMC summary
    (profiles/*_pmc_traffic.json, produced by tools/run_pmc.sh + tools/summarize_pmc.py:
    separate --pmc passes, gfx950 FETCH_SIZE correction calibrated on this access pattern).
    PMC counters cannot be read from inside this process, so the figure is the latest
    profiled one for the same workload, or None."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json"))):
        try:
            w = json.load(open(f))["workloads"].get("%s_%d" % (env_name, n_envs))
        except Exception:
            w = None
        if w:
            best = (w["traffic_bytes_per_launch"], os.path.basename(f))
    return best


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1040, help="agent steps to time; rounded UP to whole rollouts (see --rollout-steps)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps after reset (default 52 SwingRacket / 1040 Tennisbot: its envs are reset together and their first episode -- every ball still in flight -- steps 12 %% faster than the steady state); rounded up to whole episodes; the captured rollout is also replayed once before the clock starts")
    ap.add_argument("--rollout-steps", type=int, default=1040, help="agent steps per rollout = per hipGraph replay, ended by the join of the fast-forwards (and the exchange); SwingRacket: a multiple of 26")
    ap.add_argument("--env", choices=["swing", "tennis"], default="swing")
    ap.add_argument("--envs-per-gpu", type=int, default=4096)
    ap.add_argument("--contact-off", action="store_true", help="BASELINE configs[1] bench mode: racket<->ball pair disabled")
    ap.add_argument("--racket-ground", action="store_true", help="also simulate racket<->court contact (TB_F_RACKET_GROUND, row f3)")
    ap.add_argument("--rolling-friction", action="store_true", help="also solve the rolling-friction rows of every ball contact (TbParams.roll_*, row f3; opt-in)")
    ap.add_argument("--magnus", type=float, default=0.0, help="BASELINE configs[4] extension: Magnus coefficient k_M in F = k_M w x v (0 = the reference)")
    ap.add_argument("--spin-max", type=float, default=0.0, help="BASELINE configs[4] extension: initial ball spin ~ U(-w, w)^3 rad/s at reset (0 = the reference)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU work per baseline leg (1 core, all cores)")
    ap.add_argument("--sweep", action="store_true", help="the whole N ladder (4096 .. 4 M envs) for this env instead of the default 4096 + 1 M of both envs")
    ap.add_argument("--no-sweep", action="store_true")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--gather-chunks", type=int, default=8, help="multi-rank exchange of the rollout: C > 1 (default 8) = C step-chunks, each chunk's all-gather issued on a high-priority side stream as soon as the chunk is final, overlapped with the later chunks' steps; 1 = ONE all-gather after the rollout")
    ap.add_argument("--no-graph", action="store_true", help="launch every step from the host instead of replaying one captured hipGraph per rollout")
    ap.add_argument("--fast-forward", choices=["auto", "slots", "pool"], default="auto", help="pin the pipeline's form (TbOptions.ff_defer) for A/B runs: slots = one fast-forward kernel per episode end on a side stream, pool = every episode end parked, one launch at the join; auto = the library's choice by batch size")
    ap.add_argument("--no-pipeline", action="store_true", help="run the SwingRacket fast-forward inside the step kernel instead of a side stream")
    ap.add_argument("--min-timed-ms", type=float, default=50.0, help="time at least this long: more whole rollouts than --steps asks for if need be (the replay rate of one process wanders by +-5 %% over seconds, tools/diag/diag_ramp.py; one 6.5 ms rollout samples that, eight average it); steps_timed reports what was timed")
    ap.add_argument("--settle-seconds", type=float, default=1.5, help="untimed replays of the rollout before the clock starts, for this long: a fresh process replays the SwingRacket graph at 645-655 M env steps/s for its first 0.6-1.3 s about every second time and at 700+ M from then on (tools/diag/diag_ramp.py); counted in warmup_run")
    args = ap.parse_args()
    if args.warmup is None:
        args.warmup = 52 if args.env == "swing" else 1040
    return args


def fill_actions(buf_actions, seed, torch):
    """synthetic U(-1,1) float32 actions from numpy PCG64, uploaded once; the
    host draw is bounded (one 104-step block, tiled) so set-up stays short at large N"""
    import numpy as np
    T, N, A = buf_actions.shape
    block = min(T, 104)
    rng = np.random.Generator(np.random.PCG64(seed))
    a = torch.from_numpy(rng.uniform(-1.0, 1.0, (block, N, A)).astype(np.float32)).to(buf_actions.device)
    for t0 in range(0, T, block):
        n = min(block, T - t0)
        buf_actions[t0:t0 + n].copy_(a[:n])


class Rollouts:
    """One rank's measured object: an env batch, its rollout buffer, and ONE way to run a rollout
    (hipGraph replay or host-issued steps) and ONE way to exchange it (chunked + overlapped, or a single
    all-gather)."""

    def __init__(self, env, buf, torch, dist_on, use_graph, chunks, force_collective=False, exchange=True):
        self.env, self.buf, self.torch, self.dist_on = env, buf, torch, dist_on
        self.force = force_collective
        self.collective = (dist_on and exchange) or force_collective  # exchange=False: replicas only (barriers and clocks still joined)
        self.T = buf.T
        self.chunks = chunks if (self.collective and chunks > 1 and self.T % chunks == 0) else 1
        self.graph = None
        self.use_graph = use_graph
        self.note = ""

    def prepare(self):
        """capture the rollout (nothing runs), then run it once untimed: RCCL's first use of each collective
        shape (channels, buffers) and the allocator's first touches stay outside the timed region"""
        env, buf, torch = self.env, self.buf, self.torch
        if self.use_graph:
            try:
                self.graph = buf.capture_marked(env, self.chunks) if self.chunks > 1 else env.capture(lambda: buf.step_range(env, 0, self.T))
            except Exception as exc:  # fall back to host-issued launches and say so
                print("hipGraph capture failed (%s: %s); issuing the steps from the host" % (type(exc).__name__, exc), file=sys.stderr)
                self.graph, self.use_graph = None, False
        if self.chunks > 1:
            try:
                self.run_once()
            except Exception as exc:  # a chunked exchange that does not complete must not cost the run
                print("chunked exchange failed (%s: %s); falling back to ONE all-gather per rollout" % (type(exc).__name__, exc), file=sys.stderr)
                self.note = "chunked exchange failed on this node (%s): ONE all-gather per rollout was timed instead" % type(exc).__name__
                torch.cuda.synchronize(env.device)
                self.chunks = 1
                self.graph = env.capture(lambda: buf.step_range(env, 0, self.T)) if self.use_graph else None
                self.run_once()
        else:
            self.run_once()
        torch.cuda.synchronize(env.device)

    def settle(self, seconds):
        """untimed replays until `seconds` have passed (all ranks the same number: the count is agreed on through an all-reduce);
        returns how many rollouts that were. self.first_window_s: wall-clock of the first (up to) 8 of them -- the process's
        cold rate, reported next to the settled one."""
        torch = self.torch
        self.first_window_s, self.first_window_rollouts = None, 0
        if seconds <= 0:
            return 0
        t0, k = time.perf_counter(), 0
        while True:
            self.run_once()
            torch.cuda.synchronize(self.env.device)
            k += 1
            if k <= 8:
                self.first_window_s, self.first_window_rollouts = time.perf_counter() - t0, k
            go = torch.tensor([1.0 if time.perf_counter() - t0 < seconds else 0.0], device=self.env.device)
            if self.dist_on:
                torch.distributed.all_reduce(go, op=torch.distributed.ReduceOp.MIN)
            if go.item() < 0.5 or k >= 10000:
                return k

    def steps_only(self):
        env, buf = self.env, self.buf
        if self.graph is not None and self.chunks > 1:
            buf.replay_marked(self.graph, env, self.chunks, gather=False)
        elif self.graph is not None:
            self.graph.replay()
        else:
            buf.step_range(env, 0, self.T)
            env.flush()  # pipelined fast-forwards: every step's outputs are complete from here on

    def exchange_only(self):
        buf = self.buf
        if not self.collective:
            return
        if self.chunks > 1:
            buf.begin_gather(self.chunks, force=self.force, priority=-1)
            for c in range(self.chunks):
                buf.gather_chunk(c, force=self.force, after_mark=True)
            buf.finish_gather()
        else:
            buf.all_gather(force=self.force)

    def run_once(self):
        """one rollout + its exchange, as the timed region runs it"""
        env, buf = self.env, self.buf
        if self.chunks > 1:
            buf.begin_gather(self.chunks, force=self.force, priority=-1)
            if self.graph is not None:
                buf.replay_marked(self.graph, env, self.chunks, gather=True, force=self.force)
            else:  # host-issued variant (--no-graph, or capture failed): the side stream waits by event
                seg = self.T // self.chunks
                for c in range(self.chunks):
                    buf.step_range(env, c * seg, (c + 1) * seg)
                    buf.gather_chunk(c, env=env, force=self.force)
                env.flush()
            buf.finish_gather()
        else:
            self.steps_only()
            if self.collective:
                buf.all_gather(force=self.force)  # collect boundary: one collective

    def timed(self, fn, reps):
        """reps x fn() bracketed by barrier + synchronize on both sides; (wall seconds, HIP-event seconds)"""
        torch, dev = self.torch, self.env.device
        if self.dist_on:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record(torch.cuda.current_stream(dev))  # the stream the step kernels are launched on
        for _ in range(reps):
            fn()
        ev1.record(torch.cuda.current_stream(dev))
        if self.dist_on:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)
        return time.perf_counter() - t0, ev0.elapsed_time(ev1) * 1e-3


def pick_exchange_form(chunked_wall, chunked_ok, single_wall, single_ok):
    """which of the two timed exchange forms `value` reports: ALWAYS the default form (the chunked, overlapped exchange) -- the metric
    is one fixed workload on every node, not the faster of two noisy measurements; the single all-gather's figures stay
    beside it under exchange.single_all_gather. Only a chunked run whose gathered shards FAILED their check gives way to a single
    all-gather whose shards checked out (a question of validity, not of speed)."""
    if single_ok and not chunked_ok:
        return "single_all_gather"
    return "chunked"


# What the exchange should cost on a node, before any node has been measured. xGMI: 8 GPUs fully connected,
# 7 links x ~153 GB/s per GPU and direction (MI355X_MICROARCH.md / SURVEY.md 8e); an all-gather of B bytes per rank moves
# (world - 1) x B into every GPU. "direct": every rank writes its shard to all peers at once, one link each; "ring": world - 1 hops
# over one link. RCCL's choice is not ours to make; both are given.
XGMI_LINK_GBS = 153.0


def predict_exchange(world, bytes_per_rank, rollout_ms, chunks, rollout_ms_pool_form=None):
    """predicted rollout + exchange time per rank and the scaling efficiency it implies, for the single all-gather after the
    rollout (nothing overlapped) and for the chunked form (all but the last chunk's exchange hidden behind later chunks' steps, when
    a chunk's exchange is shorter than a chunk's steps). rollout_ms: the rank's rollout in the chunked form (marked graph, one
    fast-forward kernel per episode end); rollout_ms_pool_form: its rollout without marks (what the single all-gather follows, and
    what N = 1 runs): `efficiency` is against the form's own rollout, `efficiency_vs_n1` against the N = 1 rollout -- the figure the
    driver's value(N) / (N x value(1)) will show."""
    if world < 2:
        return None
    pool_ms = rollout_ms_pool_form if rollout_ms_pool_form else rollout_ms
    out = {"assumed": {"xgmi_link_GBs": XGMI_LINK_GBS, "links_per_gpu": 7, "world": world, "bytes_per_rank": int(bytes_per_rank), "rollout_ms_chunked_form": rollout_ms,
                       "rollout_ms_pool_form": pool_ms,
                       "note": "peak link rate, no protocol overhead: an upper bound on the efficiency; the first SCALE record is to be read against it"}}
    for name, hops in (("direct", 1), ("ring", world - 1)):
        ag_ms = hops * bytes_per_rank / (XGMI_LINK_GBS * 1e9) * 1e3
        single = pool_ms + ag_ms
        per_chunk = ag_ms / chunks
        steps_chunk = rollout_ms / chunks
        chunked = rollout_ms + per_chunk + max(0.0, per_chunk - steps_chunk) * (chunks - 1)
        out[name] = {"all_gather_ms": ag_ms,
                     "single_all_gather": {"ms_per_rollout": single, "efficiency": pool_ms / single, "efficiency_vs_n1": pool_ms / single},
                     "chunked": {"chunks": chunks, "ms_per_rollout": chunked, "efficiency": rollout_ms / chunked, "efficiency_vs_n1": pool_ms / chunked}}
    return out


def cadence_profile(env_name, n_envs):
    """kernel_us / gap_us of the step launches inside the replayed rollout graph, from the committed un-profiled probe
    (profiles/*cadence.json, tools/diag/r04_cadence.py: a build that runs at the product's rate logs the real-time counter at the entry
    and exit of every launch) -- the two committed numbers `roofline.frac` can be recomputed from. (rocprofv3's per-dispatch average is
    NOT such a number: the profiler serialises every dispatch and counts its start-up, 4.5 us against a 3.6 us cadence.)"""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*cadence.json"))):
        try:
            with open(f) as fh:
                d = json.load(fh)
        except (OSError, ValueError):
            continue
        w = d.get(env_name)
        if isinstance(w, dict) and w.get("envs") == n_envs and "kernel_us" in w and "gap_us" in w and "cadence_build_rate_M" in w:
            best = (w, os.path.basename(f))
    return best


def timed_pipeline_form(env, marked):
    """tb_pipeline_form of the graph that was timed: a marked graph (chunked exchange) was captured with progress marks enabled, and
    the library's automatic choice differs there"""
    if not marked:
        return env.pipeline_form()
    env.mark_enable(True)
    try:
        return env.pipeline_form()
    finally:
        env.mark_enable(False)


def init_distributed(torch, dev, world, rehearsal=False, force_collective=False):
    """One process group per run: RCCL ("nccl") over xGMI. Returns None, or -- SURVEY.md 8e's fallback, labelled, never silent -- the
    reason why RCCL cannot be used on this node: the ranks then step their shards as replicas (no rollout exchange) and only
    join their clocks, over gloo."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist = torch.distributed
    if force_collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group(backend="nccl", device_id=dev, rank=0, world_size=1)
        return None
    if rehearsal:
        dist.init_process_group(backend="gloo")
        return "rehearsal of the fallback (TB_BENCH_FAIL_NCCL=1)" if os.environ.get("TB_BENCH_FAIL_NCCL") == "1" else None
    try:
        if os.environ.get("TB_BENCH_FAIL_NCCL") == "1":
            raise RuntimeError("TB_BENCH_FAIL_NCCL=1")
        dist.init_process_group(backend="nccl", device_id=dev)
        probe = torch.ones(1, device=dev)
        dist.all_reduce(probe)  # the first collective builds the communicator: fail here, not inside the timed region
        if int(probe.item()) != world:
            raise RuntimeError("all-reduce of ones over %d ranks returned %d" % (world, int(probe.item())))
        return None
    except Exception as exc:
        why = "%s: %s" % (type(exc).__name__, str(exc).replace("\n", " ")[:240])
        print("RCCL is not usable here (%s): replicas only -- every rank steps its shard, no rollout exchange; the ranks' clocks are "
              "still joined (gloo)" % why, file=sys.stderr)
        try:
            if dist.is_initialized():
                dist.destroy_process_group()
        except Exception:
            pass
        try:
            dist.init_process_group(backend="gloo")  # same rendezvous store (under the launcher the agent hosts it): a new group gets a new key prefix
        except Exception:
            os.environ["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29500")) + 1)  # rank 0 hosted the first store itself and it still holds the port
            dist.init_process_group(backend="gloo")
        return why


def warm_up(env, buf, warmup, period):
    """W untimed steps, rounded up to whole episodes so that the rollout starts at an episode start"""
    w = -(-warmup // period) * period
    for t in range(w):
        buf.step_into(env, t % buf.T)
    env.flush()
    return w


def sweep_entry(kind, env_name, n, flags, pipeline, dev, seed, use_graph, torch):
    """one N of the ladder: whole rollouts as in the headline line (1040 steps up to 32768 envs, 104 above: the rollout
    buffer of 4 M envs x 104 steps is 24 of the 288 GB), replayed until >= ~5 ms are timed"""
    from tennisbot_rl_amd.params import ENV_SWING, default_params
    from tennisbot_rl_amd.rollout import RolloutBuffer
    from tennisbot_rl_amd.stepper import BatchedEnv
    T = 1040 if n <= 32768 else 104
    e2 = BatchedEnv(kind, n, device=dev, seed=seed, params=default_params(flags=flags), track_terminal_obs=False, pipeline=pipeline)
    b2 = RolloutBuffer(kind, T, n, dev)
    b2.actions.uniform_(-1.0, 1.0)  # device RNG: a host PCG64 draw for 1 M envs would dominate the set-up
    b2.bind(e2)
    e2.reset()
    warm_up(e2, b2, 26 if kind == ENV_SWING else 1040, 26 if kind == ENV_SWING else 1)  # Tennisbot: steady state, past the first episodes
    r = Rollouts(e2, b2, torch, False, use_graph, 1)
    r.prepare()
    r.settle(1.0 if n <= 32768 else 0.0)  # small batches: see --settle-seconds
    _, one = r.timed(r.run_once, 1)
    reps = max(1, min(64, int(5e-3 / max(one, 1e-6)) + 1))
    e2.counters_reset()
    w, evs = r.timed(r.run_once, reps)
    c = e2.counters()
    ab = ALGO_BYTES[env_name]
    k = T * reps
    out = {"env": env_name, "envs": n, "steps_per_s": n * k / w, "launch_us": evs / k * 1e6, "rollout_steps": T, "rollouts_timed": reps,
           "substeps_per_agent_step": c["substeps"] / (n * k),
           "achieved_GBs": (ab["read"] + ab["write"]) * n / (evs / k) / 1e9, "read_GBs": ab["read"] * n / (evs / k) / 1e9}
    out["frac"] = out["achieved_GBs"] / HBM_PEAK_GBS
    out["read_frac"] = out["read_GBs"] / HBM_PEAK_GBS
    e2.close()
    del b2, r
    torch.cuda.empty_cache()
    return out


def open_loop_rate(kind, N, flags, pipeline, dev, seed, torch):
    """Not the metric (an RL loop needs the observation before it can choose the next action) but the
    rate of the same kernels without the launch boundary between steps: tb_rollout, whole episodes per
    launch with the actions known up front, replayed as one hipGraph."""
    from tennisbot_rl_amd.params import default_params
    from tennisbot_rl_amd.stepper import BatchedEnv
    env = BatchedEnv(kind, N, device=dev, seed=seed, params=default_params(flags=flags), track_terminal_obs=False, pipeline=pipeline)
    K = 1040
    acts = torch.empty((K, N, env.act_dim), dtype=torch.float32, device=dev).uniform_(-1.0, 1.0)
    env.reset()
    env.rollout(acts[:52] if pipeline else acts)  # Tennisbot: past the first (synchronised, cheaper) episodes
    env.flush()
    g = env.capture(lambda: env.rollout(acts))
    g.replay()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    g.replay()
    torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t0
    env.close()
    return {"steps_per_s": N * K / wall, "us_per_step": wall / K * 1e6, "steps": K, "envs": N,
            "note": "tb_rollout: up to 26 agent steps per launch, actions known up front (not an RL loop; shows the cost of the per-step launch boundary)"}


def parity_text():
    """what the line may claim about parity, and -- from the committed sensitivity table (tools/pin_sensitivity.py ->
    profiles/r03_pin_sensitivity.json) -- which recalled engine constants the reference's one PyBullet record constrains at all"""
    txt = ("bit-exact vs the CPU restatement, which agrees with a third, independently written implementation (dense-matrix substep, numpy narrowphase, plain-Python "
           "env logic and Philox: tests/test_oracle_independent.py, test_independent_episode.py) through whole episodes of both envs; PyBullet parity UNPINNED at trajectory level (the reference ships no tests or golden vectors, PyBullet is "
           "not available offline); pinned statistically by what the reference's ppo_swing.zip holds: the returns of its last 100 PyBullet episodes "
           "(two-sample KS + double-bonus count, with a leave-half-out selection test) and its critic's predictions (state-conditional calibration on 16 384 episodes)")
    try:
        st = json.load(open(os.path.join(ROOT, "profiles", "r03_pin_sensitivity.json")))["status"]
        con = sorted(k for k, v in st.items() if v != "free")
        free = sorted(k for k, v in st.items() if v == "free")
        txt += "; constants that record CONSTRAINS: " + ", ".join(con) + "; constants it leaves FREE (recalled from Bullet, not checkable here): " + ", ".join(free)
    except Exception:
        txt += "; (sensitivity table profiles/r03_pin_sensitivity.json not found)"
    return txt + " (DESIGN.md section 2)"


def reference_record():
    """what the reference itself recorded: wall-clock of the last 100 PyBullet training episodes inside its shipped
    ppo_swing.zip (exported as data by tools/export_reference_episode_stats.py). Not this host, not this workload shape
    (1 env, SB3 policy inference + Monitor + numpy between steps): quoted for scale, never as the target."""
    import numpy as np
    try:
        rec = json.load(open(os.path.join(ROOT, "tests", "golden", "ppo_swing_reference_episodes.json")))
    except Exception:
        return None
    t = np.asarray(rec["episode_wallclock_s"], np.float64)
    n = np.asarray(rec["episode_lengths"], np.float64)
    dt = np.diff(t)
    plain = (n[1:] == 26) & (dt < 3 * np.median(dt))  # episodes not interrupted by a PPO update or an evaluation
    return {"agent_steps_per_s_collect": float(26.0 / np.median(dt[plain])), "agent_steps_per_s_overall": float(n[1:].sum() / (t[-1] - t[0])),
            "episodes": int(t.size), "envs": 1,
            "caveat": "the reference's own PyBullet run (SwingRacket-v0, DIRECT mode, 1 env, SB3 1.8.0 PPO on unnamed hardware): "
                      "collect = 26 / median wall-clock per uninterrupted episode, includes SB3's per-step policy inference, Monitor and the "
                      "per-episode world rebuild (4 loadURDF); overall also includes the PPO updates and EvalCallback episodes in between",
            "source": "backup_models/ppo_swing.zip ep_info_buffer['t'] -> tests/golden/ppo_swing_reference_episodes.json"}


def cpu_baseline(kind_name, n_envs, seconds, seed, flags):
    """the oracle (kind "port") on this host, same workload shape, bounded time"""
    import numpy as np
    from oracle import OracleBatch
    from tennisbot_rl_amd.params import ENV_SWING, ENV_TENNIS, F_AUTO_RESET, default_params
    kind = ENV_SWING if kind_name == "swing" else ENV_TENNIS
    A = 6 if kind_name == "swing" else 2
    rng = np.random.Generator(np.random.PCG64(seed))
    acts = rng.uniform(-1.0, 1.0, (104, n_envs, A)).astype(np.float32)
    # the GPU box gives one-GPU jobs a 16-CPU share of a much larger host: more threads than
    # that only oversubscribe (measured: 256 threads -> 60x slower than 1)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))
    out = {}
    try:  # the PyBullet path is the preferred baseline where it exists
        import pybullet  # noqa: F401
        out["pybullet"] = "importable, but the reference's asset files do not travel to this host: not run (tools/pybullet_crosscheck.py is the harness)"
    except ImportError:
        out["pybullet"] = "not importable on this host: CPU baseline is the float32 restatement (kind \"port\")"
    for label, threads in (("1core", 1), ("allcores", cores)):
        b = OracleBatch(default_params(flags=flags | F_AUTO_RESET), kind, n_envs, seed=seed, precision="f32", threads=threads)
        b.reset()
        done_steps, t0 = 0, time.perf_counter()
        while True:
            for t in range(26):  # whole Swing episodes so that the fast-forward share is the workload's
                b.step(acts[(done_steps + t) % 104])
            done_steps += 26
            el = time.perf_counter() - t0
            if el > seconds:
                break
        sub = float(b.counters()[6])
        out[label] = {"steps_per_s": done_steps * n_envs / el, "agent_steps": done_steps, "seconds": el,
                      "substeps_per_s": sub / el, "substeps_per_agent_step": sub / (done_steps * n_envs), "threads": threads}
        b.close()
    return out


def self_launch(n, argv):
    """`python bench.py --gpus N` (N > 1) without a launcher's environment: become the launcher. This process never imports
    torch and never touches a GPU (a process that has initialised the GPU must not exec or fork GPU children on this pool): it
    starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py <the same arguments>` as a CHILD, waits,
    prints the one JSON line rank 0 wrote (anything else the ranks put on stdout goes to stderr) and returns the launcher's
    exit status -- non-zero when any rank failed, and non-zero when no line came back."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC (RCCL's peer buffers)
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    p = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for ln in p.stdout.splitlines():
        try:
            d = json.loads(ln)
        except ValueError:
            d = None
        if isinstance(d, dict) and "metric" in d:
            line = ln
        elif ln.strip():
            print(ln, file=sys.stderr)
    if line is not None:
        print(line)
    if p.returncode:
        print("bench.py: the %d-rank run ended with status %d" % (n, p.returncode), file=sys.stderr)
        return p.returncode
    return 0 if line is not None else 1


def dry_run(args):
    """TB_BENCH_DRY_RUN=1 -- for hosts WITHOUT a GPU (the CPU test-suite): everything of an N-rank run except the envs. The ranks
    rendezvous (gloo), count each other, and exchange a small CPU rollout buffer of the real record layout in both forms of the
    exchange (8 step-chunks; ONE all-gather), checking the gathered bytes; rank 0 prints a line with the contract's keys whose
    value is null ("invalid" says why): no env is stepped -- the stepper is HIP-only -- so there is nothing to rate."""
    import torch
    from tennisbot_rl_amd.params import ENV_SWING, ENV_TENNIS
    from tennisbot_rl_amd.rollout import RolloutBuffer
    dist = torch.distributed
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if os.environ.get("TB_BENCH_DRY_FAIL_RANK") == str(rank):  # (the test of the launcher's exit status)
        sys.exit(3)
    if world > 1:
        dist.init_process_group(backend="gloo")
    seen = torch.ones(1)
    if world > 1:
        dist.all_reduce(seen)
    kind = ENV_SWING if args.env == "swing" else ENV_TENNIS
    T, N = 208, 64
    buf = RolloutBuffer(kind, T, N, "cpu")
    buf.raw.copy_(torch.arange(buf.nbytes, dtype=torch.int64).add_(7919 * (rank + 1)).remainder_(251).to(torch.uint8))
    forms, ok = {}, True
    for chunks in sorted({max(1, args.gather_chunks), 1}, reverse=True):
        t0 = time.perf_counter()
        if chunks > 1 and world > 1:
            buf.begin_gather(chunks)
            for c in range(chunks):
                buf.gather_chunk(c)
            shards = buf.finish_gather()
        else:
            shards = buf.all_gather()
        forms["%d" % chunks] = (time.perf_counter() - t0) * 1e3
        ok = ok and buf.check_gathered() and len(shards) == world
    okt = torch.tensor([1.0 if ok else 0.0])
    if world > 1:
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({
            "metric": "env steps/sec (whole node), SwingRacket-v0 @4096 envs/GPU", "value": None, "unit": "env steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "dry_run": True,
            "invalid": "dry run (TB_BENCH_DRY_RUN=1): launcher, rendezvous and rollout exchange over gloo only; no env was stepped (the stepper is HIP-only)",
            "config": {"workload": "none (dry run): %d ranks x a %d-step x %d-env CPU rollout buffer exchanged over gloo" % (world, T, N),
                       "parallelism": "env-sharded x%d" % world},
            "exchange": {"ranks_seen": int(seen.item()), "bytes_per_rank": int(buf.nbytes), "gathered_ok": bool(okt.item() > 0.5),
                         "exchange_ms_by_chunks": forms}}))
    return 0 if okt.item() > 0.5 else 1


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))  # before torch is imported: the launcher process never touches a GPU
    if os.environ.get("TB_BENCH_DRY_RUN") == "1":
        sys.exit(dry_run(args))
    import torch
    from tennisbot_rl_amd.params import ENV_SWING, ENV_TENNIS, F_DEFAULT, F_NET, F_RACKET_GROUND, default_params
    from tennisbot_rl_amd.rollout import RolloutBuffer
    from tennisbot_rl_amd.stepper import BatchedEnv

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist_on = world > 1
    if args.gpus != world:
        print("bench.py: --gpus %d but the launcher started %d rank(s): the line reports n_gpus = %d" % (args.gpus, world, world), file=sys.stderr)
    # TB_BENCH_REHEARSAL=1: several ranks share cuda:0 over gloo -- a one-GPU rehearsal of the
    # multi-rank control flow (the numbers mean nothing; RCCL refuses two ranks on one device)
    rehearsal = os.environ.get("TB_BENCH_REHEARSAL") == "1"
    # TB_BENCH_FORCE_COLLECTIVE=1: a single rank still initialises RCCL and issues every collective
    # of the multi-rank path (one-GPU rehearsal of the RCCL calls themselves; numbers mean little)
    force_collective = os.environ.get("TB_BENCH_FORCE_COLLECTIVE") == "1" and world == 1
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    replicas_only = init_distributed(torch, dev, world, rehearsal, force_collective) if (dist_on or force_collective) else None

    kind = ENV_SWING if args.env == "swing" else ENV_TENNIS
    period = 26 if kind == ENV_SWING else 1
    flags = F_NET if args.contact_off else F_DEFAULT
    if args.racket_ground:
        flags |= F_RACKET_GROUND
    N = args.envs_per_gpu
    pipeline = kind == ENV_SWING and not args.no_pipeline
    ext = dict(magnus_k=args.magnus, ball_spin_max=args.spin_max) if (args.magnus or args.spin_max) else {}
    if args.rolling_friction:
        from tennisbot_rl_amd.params import reference_rolling_friction
        ext.update(reference_rolling_friction())
    env = BatchedEnv(kind, N, device=dev, seed=args.seed, env_id_base=rank * N, params=default_params(flags=flags, **ext),
                     track_terminal_obs=False, pipeline=pipeline, options=dict(ff_defer={"auto": None, "slots": False, "pool": "all"}[args.fast_forward]))
    T_roll = -(-max(1, args.rollout_steps) // period) * period
    rollouts = -(-max(1, args.steps) // T_roll)
    steps_timed = rollouts * T_roll
    buf = RolloutBuffer(kind, T_roll, N, dev)
    fill_actions(buf.actions, args.seed + rank, torch)
    buf.bind(env)
    env.reset()
    warmup_run = warm_up(env, buf, args.warmup, period)
    fake_us = float(os.environ.get("TB_BENCH_FAKE_GATHER_US", "0"))
    if fake_us > 0 and force_collective:
        # one-GPU rehearsal of the overlap: every exchange of the whole rollout is preceded, on the stream that
        # carries it, by a kernel that just runs for fake_us in total (split over the chunks)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(1000); torch.cuda.synchronize(dev)
        e0.record(); torch.cuda._sleep(10_000_000); e1.record(); torch.cuda.synchronize(dev)
        cycles_per_us = 10_000_000 / (e0.elapsed_time(e1) * 1e3)
        buf.rehearsal_total_cycles = int(fake_us * cycles_per_us)
    R = Rollouts(env, buf, torch, dist_on, not args.no_graph, max(1, args.gather_chunks), force_collective, exchange=replicas_only is None)
    R.prepare()
    warmup_run += R.settle(args.settle_seconds) * T_roll + T_roll  # (+ the one untimed rollout of prepare())
    if args.min_timed_ms > 0:  # agreed on by all ranks: the slowest rank's untimed rollout decides
        one_t = torch.tensor([R.timed(R.run_once, 1)[0]], dtype=torch.float64, device=dev)
        if dist_on:
            torch.distributed.all_reduce(one_t, op=torch.distributed.ReduceOp.MAX)
        warmup_run += T_roll
        rollouts = max(rollouts, min(256, int(args.min_timed_ms * 1e-3 / max(float(one_t.item()), 1e-6)) + 1))
        steps_timed = rollouts * T_roll
    exch = None
    if R.collective:  # measured apart, untimed: what the rollout and the exchange cost on their own
        seen = torch.ones(1, device=dev)
        if dist_on:
            torch.distributed.all_reduce(seen)
        r_wall, _ = R.timed(R.steps_only, 1)
        x_wall, _ = R.timed(R.exchange_only, 1)
        exch = {"ranks_seen": int(seen.item()), "bytes_per_rank": int(buf.nbytes), "rollout_ms": r_wall * 1e3, "exchange_ms": x_wall * 1e3,
                "form": "%d step-chunks, each all-gathered on a high-priority side stream while later chunks step" % R.chunks if R.chunks > 1
                        else "ONE all-gather after the rollout"}
    env.counters_reset()  # from here on the counters hold the timed steps only
    wall, ev_s = R.timed(R.run_once, rollouts)
    gather_ok = True
    if R.collective:  # after the clock: every rank must hold every shard, starting with its own
        ok = torch.tensor([1.0 if buf.check_gathered() else 0.0], device=dev)
        if dist_on:  # one verdict for all ranks: whatever follows, they do it together
            torch.distributed.all_reduce(ok, op=torch.distributed.ReduceOp.MIN)
        gather_ok = bool(ok.item() > 0.5)
    c = env.counters()

    single = None
    if R.collective and R.chunks > 1:
        # north_star's wording of the exchange -- "a single RCCL all-gather at the PPO collect boundary" -- timed the same way next
        # to the default form: the same rollouts, then ONE all-gather of the whole rollout buffer, nothing overlapped
        R1 = Rollouts(env, buf, torch, dist_on, R.use_graph, 1, force_collective, exchange=True)
        R1.prepare()
        x1_wall, _ = R1.timed(R1.exchange_only, 1)
        r1_wall, _ = R1.timed(R1.steps_only, 1)  # (its own graph: no marks, one fast-forward launch at the join)
        env.counters_reset()
        w1, ev1_s = R1.timed(R1.run_once, rollouts)
        ok1 = torch.tensor([1.0 if buf.check_gathered() else 0.0], device=dev)
        w1_t = torch.tensor([w1, x1_wall, r1_wall], dtype=torch.float64, device=dev)
        if dist_on:
            torch.distributed.all_reduce(w1_t, op=torch.distributed.ReduceOp.MAX)
            torch.distributed.all_reduce(ok1, op=torch.distributed.ReduceOp.MIN)
        single = {"form": "ONE all-gather after the rollout (--gather-chunks 1)", "value": world * N * steps_timed / float(w1_t[0].item()),
                  "exchange_ms": float(w1_t[1].item()) * 1e3, "ms_per_rollout": float(w1_t[0].item()) / rollouts * 1e3, "rollouts_timed": rollouts,
                  "rollout_ms": float(w1_t[2].item()) * 1e3}
        c1 = env.counters()
        del R1

    wall_t = torch.tensor([wall], dtype=torch.float64, device=dev)
    if dist_on:
        torch.distributed.all_reduce(wall_t, op=torch.distributed.ReduceOp.MAX)
    # Two complete forms of the same job were timed the same way (K steps each, barrier + synchronize on both sides, max over ranks):
    # `value` is ALWAYS the default (chunked) form's, named in config.workload and in exchange.timed_form; the single all-gather's
    # figures stay beside it (pick_exchange_form: only a chunked run whose shards failed their check gives way).
    picked_single = single is not None and pick_exchange_form(float(wall_t.item()), gather_ok, float(w1_t[0].item()), bool(ok1.item() > 0.5)) == "single_all_gather"
    if picked_single:
        chunked = {"form": exch["form"], "value": world * N * steps_timed / float(wall_t.item()), "ms_per_rollout": float(wall_t.item()) / rollouts * 1e3,
                   "rollouts_timed": rollouts, "gather_ok": gather_ok}
        wall, ev_s, c, gather_ok = w1, ev1_s, c1, True
        wall_t = w1_t[:1].clone()
    sub_t = torch.tensor([float(c["substeps"]), float(c["nonfinite_states"] + c["lockstep_violations"])], dtype=torch.float64, device=dev)
    if dist_on:
        torch.distributed.all_reduce(sub_t, op=torch.distributed.ReduceOp.SUM)
    wall_max = float(wall_t.item())
    timed_substeps, bad_states = float(sub_t[0].item()), float(sub_t[1].item())

    result = None
    if rank == 0:
        ab = ALGO_BYTES[args.env]
        per_launch_bytes = (ab["read"] + ab["write"]) * N
        launch_s = ev_s / steps_timed
        achieved = per_launch_bytes / launch_s / 1e9
        traffic = None if args.contact_off else pmc_traffic(args.env, N)
        cadence = None if (args.contact_off or args.racket_ground or args.rolling_friction) else cadence_profile(args.env, N)
        sps = timed_substeps / (world * N * steps_timed)
        if exch is not None:
            exch["exposed_exchange_ms"] = max(0.0, wall_max / rollouts * 1e3 - exch["rollout_ms"])
            exch["note"] = R.note
            if single is not None:
                single["exposed_exchange_ms"] = max(0.0, single["ms_per_rollout"] - single["rollout_ms"])
                exch["single_all_gather"] = single
                exch["timed_form"] = "single_all_gather" if picked_single else "chunked"
                if picked_single:  # `value` is the single all-gather's: the chunked form's own figures stay here
                    chunked["exposed_exchange_ms"] = max(0.0, chunked["ms_per_rollout"] - exch["rollout_ms"])
                    exch["chunked"] = chunked
                    exch["exposed_exchange_ms"] = single["exposed_exchange_ms"]
        elif replicas_only is not None:
            exch = {"ranks_seen": world, "bytes_per_rank": 0, "form": "none: REPLICAS ONLY, the sum of the ranks' own rollouts (RCCL unusable: %s)" % replicas_only}
        gather_note = ("" if not R.collective else ", rollouts all-gathered (RCCL) in %d step-chunks overlapped with the steps (one hipGraph, progress marks watched by the host)" % R.chunks
                       if R.chunks > 1 and not picked_single else ", 1 RCCL all-gather of the rollout at the collect boundary"
                       + (" (the %d-chunk overlapped form timed beside it FAILED its gather check: exchange.chunked)" % R.chunks if picked_single else ""))
        result = {
            "metric": "env steps/sec (whole node), SwingRacket-v0 @4096 envs/GPU" if args.env == "swing" and N == 4096
                      else "env steps/sec (whole node), %s @%d envs/GPU" % ("SwingRacket-v0" if args.env == "swing" else "Tennisbot-v0", N),
            "value": world * N * steps_timed / wall_max,
            "unit": "env steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "steps_timed": steps_timed, "rollout_steps": T_roll, "rollouts_timed": rollouts, "warmup_run": warmup_run,
            "timed_region_ms": wall_max * 1e3,
            "ms_per_step": wall_max / steps_timed * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s, %d envs/GPU, %s, auto-reset, U(-1,1) actions, whole rollouts of %d agent steps (%s) each ended by the join of its fast-forwards%s%s%s" % (
                "SwingRacket-v0" if args.env == "swing" else "Tennisbot-v0", N,
                ("racket<->ball contact off (configs[1] bench mode)" if args.contact_off else "full contact semantics") + (" + racket<->court contact" if args.racket_ground else "")
                + (" + rolling-friction rows" if args.rolling_friction else "")
                + (" + Magnus k=%g, spin<=%g rad/s (extension, not in the reference)" % (args.magnus, args.spin_max) if (args.magnus or args.spin_max) else ""),
                T_roll, "%d episodes" % (T_roll // 26) if kind == ENV_SWING else "steady state",
                {"none": "", "slots": ", one fast-forward kernel per episode end on a side stream", "slots+pool": ", one fast-forward kernel per episode end on a side stream (stragglers deferred to the join)",
                 "pool": ", episode ends parked and run to their end by ONE fast-forward launch at %s" % ("the end of each exchanged chunk, on a side stream" if R.chunks > 1 and not picked_single else "the join")}[timed_pipeline_form(R.env, R.chunks > 1 and not picked_single)], ", one hipGraph replay per rollout" if R.graph is not None else ", steps issued from the host",
                gather_note),
                "envs_per_gpu": N, "global_envs": world * N, "parallelism": "env-sharded x%d" % world},
            "substeps_per_s": timed_substeps / wall_max,
            "substeps_per_agent_step": sps,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic[0] if traffic else None,
                         "traffic_unit": "bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE)", "traffic_source": traffic[1] if traffic else None,
                         "algorithmic_bytes_per_launch": per_launch_bytes,
                         "kernel": "tb_step_kernel<%d>" % kind, "launch_us": launch_s * 1e6,
                         "algorithmic_bytes_per_env_step": ab["read"] + ab["write"],
                         "read_only": {"achieved": ab["read"] * N / launch_s / 1e9, "frac": ab["read"] * N / launch_s / 1e9 / HBM_PEAK_GBS,
                                       "algorithmic_bytes_per_env_step": ab["read"], "note": "the HBM-READ roofline north_star words its 40 % target on"},
                         "note": "latency-bound at this batch size, not bandwidth-bound (see sweep / DESIGN.md)"},
            "parity": parity_text(),
        }
        if cadence is not None:
            cw, csrc = cadence
            # launch_us (HIP events over the whole rollouts / their steps) = kernel_us + gap_us (one step launch and the gap to the next
            # one in the chain) + join_share_us (the rollout's ONE fast-forward launch at the join, per step)
            result["roofline"].update({
                "kernel_us": cw["kernel_us"], "gap_us": cw["gap_us"], "join_share_us": cw.get("join_share_us"),
                "cadence_source": csrc, "cadence_build_rate_M": cw["cadence_build_rate_M"], "cadence_product_rate_same_box_M": cw.get("product_rate_same_box_M"),
                "cadence_note": "committed un-profiled probe (tools/diag/r04_cadence.py): real-time counter at entry / exit of every step launch of one "
                                "replayed rollout, on a build that replays at the product's rate; frac = algorithmic_bytes_per_launch / "
                                "((kernel_us + gap_us + join_share_us) us) / peak. rocprofv3's per-dispatch average (profiles/*kernel_stats.csv) is "
                                "longer than the un-profiled cadence and is not what frac rests on (profiles/README.md)"})
        if exch is not None:
            if R.collective and exch.get("rollout_ms"):
                pool_ms = single["rollout_ms"] if single is not None else None
                exch["predicted"] = predict_exchange(max(world, exch.get("ranks_seen", world)), exch["bytes_per_rank"], exch["rollout_ms"], max(R.chunks, 1), pool_ms)
                exch["predicted_for_8_gpus"] = predict_exchange(8, exch["bytes_per_rank"], exch["rollout_ms"], max(R.chunks, 1), pool_ms)
            result["exchange"] = exch
        if replicas_only is not None:  # never to be mistaken for the metric: the rollouts were NOT exchanged
            result["replicas_only"] = True
            result["config"]["workload"] += "; REPLICAS ONLY: no rollout exchange (RCCL unusable on this node)"
        if getattr(R, "first_window_s", None):  # the cold rate of this process, next to the settled one that `value` is
            result["settle_first_rollouts"] = {"rollouts": R.first_window_rollouts, "steps_per_s": world * N * T_roll * R.first_window_rollouts / R.first_window_s,
                                               "note": "the first untimed replays of the rollout in this process (rank 0's clock): a fresh process replays ~8 % slower "
                                                       "for its first second about every second time (profiles/EXPERIMENTS.md); `value` is the settled rate"}
        invalid = []
        want = EXPECTED_SUBSTEPS.get((args.env, bool(args.racket_ground)))
        if args.contact_off or args.magnus or args.spin_max or args.rolling_friction:
            want = None  # other workloads than the headline's: reported, not judged
        if want is not None and abs(sps / want - 1.0) > 0.05:
            invalid.append("substeps per agent step %.3f, the workload's is %.3f: the timed region did not hold whole episodes" % (sps, want))
        if bad_states:
            invalid.append("%d env states went non-finite or left the lockstep the pipelined kernels rely on" % bad_states)
        if not gather_ok:
            invalid.append("the gathered rollout does not match the local shard")
        if kind == ENV_SWING and steps_timed % 26:
            invalid.append("steps_timed is not a whole number of episodes")
        if invalid:
            result["invalid"] = "; ".join(invalid)
            result["value_refused"] = result["value"]
            result["value"] = None
    if not dist_on and not args.no_sweep and not force_collective:
        ladder = [(args.env, n, None) for n in (4096, 32768, 262144, 1048576, 4194304)] if args.sweep else \
                 [("swing", 4096, None), ("swing", 1048576, None), ("tennis", 4096, None), ("tennis", 1048576, None),
                  ("swing", 4096, "BASELINE configs[1] as worded: racket-only dynamics, racket<->ball contact off")]
        sweep = []
        for name, n, variant in ladder:
            k2 = ENV_SWING if name == "swing" else ENV_TENNIS
            entry = sweep_entry(k2, name, n, F_NET if variant else flags, k2 == ENV_SWING and not args.no_pipeline, dev, args.seed, not args.no_graph, torch)
            if variant:
                entry["variant"] = variant
            sweep.append(entry)
        if result is not None:
            result["sweep"] = sweep
            if args.sweep:
                result["open_loop"] = open_loop_rate(kind, N, flags, pipeline, dev, args.seed, torch)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:  # a reported baseline, timed once, next to the 1-GPU figure
        cb = cpu_baseline(args.env, N, args.cpu_seconds, args.seed, flags)
        best = cb["allcores"]
        result["cpu_baseline"] = {
            "value": best["steps_per_s"], "unit": "env steps/s", "cores": best["threads"], "kind": "port",
            "sample": "float32 CPU oracle (oracle/tb_oracle.c, OpenMP over envs), %d envs x %d agent steps (whole 26-step episodes incl. fast-forward), %.1f s"
                      % (N, best["agent_steps"], best["seconds"]),
            "substeps_per_agent_step": best["substeps_per_agent_step"],
            "value_1core": cb["1core"]["steps_per_s"], "substeps_per_s": best["substeps_per_s"], "pybullet": cb["pybullet"],
            "substeps_per_s_1core": cb["1core"]["substeps_per_s"],
            "reference_record": reference_record() if args.env == "swing" else None,
        }
        sps = result["substeps_per_agent_step"]
        if kind == ENV_SWING and abs(sps / best["substeps_per_agent_step"] - 1.0) > 0.05 and result["value"] is not None:
            result["invalid"] = "GPU leg %.3f substeps per agent step, CPU leg %.3f: not the same workload" % (sps, best["substeps_per_agent_step"])
            result["value_refused"], result["value"] = result["value"], None
    if dist_on or force_collective:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()

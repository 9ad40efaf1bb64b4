#!/usr/bin/env python3
"""Headline benchmark: env steps/sec (whole node), SwingRacket-v0 @ 4096 envs/GPU.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one agent-level env.step() of every env of the batch (BASELINE.json metric;
SURVEY.md 8d): one launch of the HIP step kernel over N envs, i.e. 1 physics substep for
agent steps 1-25 of an episode and 1 + (up to 775) substeps on the 26th (the fast-forward
of swingracket_env.py:105-141), plus the in-kernel auto-reset. Inputs (state, synthetic
U(-1,1) actions from PCG64) are resident in HBM before the timed region; every step writes
obs / reward / done straight into the rank's rollout buffer, and with N > 1 the timed
region contains the exchange of the rollout shards (RCCL all-gather over xGMI) that the PPO
collect boundary needs: ONE collective after the K steps, or the same bytes in 8 step-chunks
overlapped with the steps -- whichever was faster on this node in untimed trials before the
clock started (--gather-chunks pins a form; the line says which ran). Rank 0 prints ONE JSON line.

Also in that line:
  roofline     -- HBM roofline of the step kernel: algorithmic bytes per launch (267 B/env
                  Swing, 263 B/env Tennisbot: DESIGN.md) / average launch duration from HIP
                  events on the launch stream. This path is launch- and ALU-latency-bound at
                  4096 envs, and the number says so; `sweep` shows the
                  asymptote at large N.
  cpu_baseline -- the float32 CPU oracle (a port, not PyBullet, which is not installable
                  here) timed on this host's cores on a bounded sample of the same workload.
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
    ap.add_argument("--steps", type=int, default=1040, help="timed agent steps (default 40 Swing episodes of 26)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps after reset (default 52 SwingRacket / 1040 Tennisbot: its envs are reset together and their first episode -- every ball still in flight -- steps 12 %% faster than the steady state; with a graph, its K steps are also replayed once before the clock starts)")
    ap.add_argument("--env", choices=["swing", "tennis"], default="swing")
    ap.add_argument("--envs-per-gpu", type=int, default=4096)
    ap.add_argument("--contact-off", action="store_true", help="BASELINE configs[1] bench mode: racket<->ball pair disabled")
    ap.add_argument("--racket-ground", action="store_true", help="also simulate racket<->court contact (TB_F_RACKET_GROUND, row f3; opt-in)")
    ap.add_argument("--rolling-friction", action="store_true", help="also solve the rolling-friction rows of every ball contact (TbParams.roll_*, row f3; opt-in)")
    ap.add_argument("--magnus", type=float, default=0.0, help="BASELINE configs[4] extension: Magnus coefficient k_M in F = k_M w x v (0 = the reference)")
    ap.add_argument("--spin-max", type=float, default=0.0, help="BASELINE configs[4] extension: initial ball spin ~ U(-w, w)^3 rad/s at reset (0 = the reference)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU work per baseline leg (1 core, all cores)")
    ap.add_argument("--sweep", action="store_true", help="also time N = 4096 .. 4M envs on one GPU (extra JSON key)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--gather-chunks", type=int, default=0, help="multi-rank exchange of the rollout: 1 = ONE all-gather at the collect boundary; C > 1 = C step-chunks, each all-gather overlapped with later chunks' steps; 0 (default) = measure both on the node, untimed, and time the faster")
    ap.add_argument("--no-graph", action="store_true", help="launch every step from Python instead of replaying one captured hipGraph")
    ap.add_argument("--no-pipeline", action="store_true", help="run the SwingRacket fast-forward inside the step kernel instead of a side stream")
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


GRAPH_STATE = {"used": True, "chunks": 1, "priority": 0, "gather_ok": None, "tuning": None}


def time_steps(env, buf, steps, warmup, torch, dist_on, tail_gather=True, graph=False, gather_chunks=0, force_collective=False):
    """W untimed + K timed steps; returns (wall seconds, HIP-event seconds of the K launches).
    graph=True: the K timed steps are captured once into a hipGraph (outside the timed region) and
    the timed region replays it -- same kernels, same buffers, no per-step host work.
    The exchange at the collect boundary (multi-rank only) has two forms: ONE all-gather after the K
    steps, or C step-chunks whose all-gathers run on a side stream while the main stream goes on
    stepping (the K steps stay ONE hipGraph with a progress mark behind every chunk,
    RolloutBuffer.capture_marked; the host watches the marks and issues each chunk's collective as
    soon as its records are final; only this library's kernels are captured, the collectives are
    plain asynchronous RCCL calls). gather_chunks = 1 / C > 1 picks a form; 0 = try ONE, and 8
    chunks at both stream priorities, untimed, on the node itself, and time the fastest (all ranks
    agree through an all-reduce): which one overlaps best depends on how the runtime maps streams to
    hardware queues, and is measured rather than assumed."""
    dev = env.device
    T = buf.T
    for t in range(warmup):
        buf.step_into(env, t % T)
    collective = (dist_on or force_collective) and tail_gather
    chunkable = collective and steps == T

    def usable(c):
        return c > 1 and chunkable and steps % c == 0
    # A pipelined SwingRacket graph bakes in which of its steps end an episode (and fork a fast-forward): it can be
    # replayed again and again only if K is a whole number of 26-step episodes. Otherwise it runs exactly once --
    # the timed run -- and nothing is tuned.
    repeatable = not getattr(env, "pipeline", False) or steps % 26 == 0
    if gather_chunks == 0:
        modes = [(1, 0)] + ([(8, -1), (8, 0)] if (usable(8) and repeatable) else [])
    elif usable(gather_chunks):
        modes = [(gather_chunks, -1 if getattr(env, "pipeline", False) else 0)]
    else:
        modes = [(1, 0)]

    plain, marked = None, {}
    if graph:
        try:
            def body():
                for t in range(steps):
                    buf.step_into(env, t % T)
            # a capture advances the library's episode-phase hint by K steps without running them: replay each
            # graph once before anything else is captured or stepped
            if any(c == 1 for c, _ in modes):
                plain = env.capture(body)
                if repeatable:
                    plain.replay()
                    torch.cuda.synchronize(dev)
            for c in sorted({c for c, _ in modes if c > 1}):
                marked[c] = buf.capture_marked(env, c)
                if repeatable:
                    marked[c].replay()
                    torch.cuda.synchronize(dev)
        except Exception as exc:  # fall back to host-issued launches and say so
            print("hipGraph capture failed (%s: %s); issuing the steps from the host" % (type(exc).__name__, exc), file=sys.stderr)
            plain, marked, graph = None, {}, False
    GRAPH_STATE["used"] = bool(graph)

    def run(mode):
        chunks, prio = mode
        seg = steps // chunks
        if chunks > 1:
            buf.begin_gather(chunks, force=force_collective, priority=prio)
        if dist_on:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record(torch.cuda.current_stream(dev))  # the stream the step kernel is launched on
        if chunks > 1 and graph:
            buf.replay_marked(marked[chunks], env, chunks, gather=True, force=force_collective)
        elif chunks > 1:  # host-issued variant (--no-graph, or capture failed): the side stream waits by event
            for c in range(chunks):
                buf.step_range(env, c * seg, (c + 1) * seg)
                buf.gather_chunk(c, env=env, force=force_collective)
            env.flush()
        elif graph:
            plain.replay()
        else:
            for t0_ in range(0, steps, T):
                buf.step_range(env, 0, min(T, steps - t0_))
            env.flush()  # pipelined fast-forwards: every step's outputs are complete from here on
        ev1.record(torch.cuda.current_stream(dev))
        if chunks > 1:
            buf.finish_gather()
        elif tail_gather:
            buf.all_gather(force=force_collective)  # collect boundary: one collective (no-op for a single rank)
        if dist_on:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)
        return time.perf_counter() - t0, ev0.elapsed_time(ev1) * 1e-3

    if collective and not (repeatable or not graph):  # the graph runs once: warm the collective up without it
        chunks, prio = modes[0]
        if chunks > 1:
            buf.begin_gather(chunks, force=force_collective, priority=prio)
            for c in range(chunks):
                buf.gather_chunk(c, env=env, force=force_collective)
            buf.finish_gather()
        else:
            buf.all_gather(force=force_collective)
        torch.cuda.synchronize(dev)
    elif collective:  # RCCL's first use of each collective shape (channels, buffers) stays outside the timed region; so does the tuning
        trial = []
        for mode in modes:
            run(mode)
            best = min(run(mode)[0] for _ in range(2)) if len(modes) > 1 else 0.0
            trial.append(best)
        if len(modes) > 1:
            tt = torch.tensor(trial, dtype=torch.float64, device=dev)
            if dist_on:
                torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            trial = [float(x) for x in tt.tolist()]
            GRAPH_STATE["tuning"] = {"%d chunk%s, stream priority %d" % (c, "" if c == 1 else "s", p): round(x * 1e3, 3) for (c, p), x in zip(modes, trial)}
            modes = [modes[trial.index(min(trial))]]
    env.counters_reset()  # from here on the counters hold the timed steps only
    wall, ev_s = run(modes[0])
    GRAPH_STATE["chunks"], GRAPH_STATE["priority"] = modes[0]
    if collective:  # after the clock: every rank must hold every shard, starting with its own
        ok = torch.tensor([1.0 if buf.check_gathered() else 0.0], device=dev)
        if dist_on:  # one verdict for all ranks: whatever follows, they do it together
            torch.distributed.all_reduce(ok, op=torch.distributed.ReduceOp.MIN)
        GRAPH_STATE["gather_ok"] = bool(ok.item() > 0.5)
    return wall, ev_s


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
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    g.replay()
    torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t0
    env.close()
    return {"steps_per_s": N * K / wall, "us_per_step": wall / K * 1e6, "steps": K, "envs": N,
            "note": "tb_rollout: up to 26 agent steps per launch, actions known up front (not an RL loop; shows the cost of the per-step launch boundary)"}


def cpu_baseline(kind_name, n_envs, seconds, seed):
    """the oracle (kind "port") on this host, same workload shape, bounded time"""
    import numpy as np
    from oracle import OracleBatch
    from tennisbot_rl_amd.params import ENV_SWING, ENV_TENNIS, F_AUTO_RESET, F_DEFAULT, default_params
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
    cores = max(1, min(cores, int(os.environ.get("TB_CPU_THREADS", "16"))))
    out = {}
    try:  # the PyBullet path is the preferred baseline where it exists
        import pybullet  # noqa: F401
        out["pybullet"] = "importable, but the reference's asset files do not travel to this host: not run (tools/pybullet_crosscheck.py is the harness)"
    except ImportError:
        out["pybullet"] = "not importable on this host: CPU baseline is the float32 restatement (kind \"port\")"
    for label, threads in (("1core", 1), ("allcores", cores)):
        b = OracleBatch(default_params(flags=F_DEFAULT | F_AUTO_RESET), kind, n_envs, seed=seed, precision="f32", threads=threads)
        b.reset()
        done_steps, t0 = 0, time.perf_counter()
        while True:
            for t in range(26):  # whole Swing episodes so that the fast-forward share is the workload's
                b.step(acts[(done_steps + t) % 104])
            done_steps += 26
            el = time.perf_counter() - t0
            if el > seconds:
                break
        out[label] = {"steps_per_s": done_steps * n_envs / el, "agent_steps": done_steps, "seconds": el,
                      "substeps_per_s": float(b.counters()[6]) / el, "threads": threads}
        b.close()
    return out


def main():
    args = parse()
    import torch
    from tennisbot_rl_amd.params import ENV_SWING, ENV_TENNIS, F_DEFAULT, F_NET, default_params
    from tennisbot_rl_amd.rollout import RolloutBuffer
    from tennisbot_rl_amd.stepper import BatchedEnv

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist_on = world > 1
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("--gpus %d needs the torch.distributed launcher (one process per GPU)" % args.gpus)
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
    if dist_on or force_collective:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if force_collective:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            torch.distributed.init_process_group(backend="nccl", device_id=dev, rank=0, world_size=1)
        elif rehearsal:
            torch.distributed.init_process_group(backend="gloo")
        else:
            torch.distributed.init_process_group(backend="nccl", device_id=dev)

    kind = ENV_SWING if args.env == "swing" else ENV_TENNIS
    flags = F_NET if args.contact_off else F_DEFAULT
    if args.racket_ground:
        from tennisbot_rl_amd.params import F_RACKET_GROUND
        flags |= F_RACKET_GROUND
    N = args.envs_per_gpu
    pipeline = kind == ENV_SWING and not args.no_pipeline
    ext = dict(magnus_k=args.magnus, ball_spin_max=args.spin_max) if (args.magnus or args.spin_max) else {}
    if args.rolling_friction:
        from tennisbot_rl_amd.params import reference_rolling_friction
        ext.update(reference_rolling_friction())
    env = BatchedEnv(kind, N, device=dev, seed=args.seed, env_id_base=rank * N, params=default_params(flags=flags, **ext),
                     track_terminal_obs=False, pipeline=pipeline)
    T_buf = min(args.steps, 1100)  # rollout length of the reference: n_steps = 1100 (train_swing.py:49-50)
    buf = RolloutBuffer(kind, T_buf, N, dev)
    fill_actions(buf.actions, args.seed + rank, torch)
    buf.bind(env)
    env.reset()
    env.counters_reset()
    use_graph = not args.no_graph
    chunks = max(0, args.gather_chunks)
    fake_us = float(os.environ.get("TB_BENCH_FAKE_GATHER_US", "0"))
    if fake_us > 0 and force_collective:
        # one-GPU rehearsal of the overlap: every exchange of the whole rollout is preceded, on the stream that
        # carries it, by a kernel that just runs for fake_us in total (split over the chunks)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(1000); torch.cuda.synchronize(dev)
        e0.record(); torch.cuda._sleep(10_000_000); e1.record(); torch.cuda.synchronize(dev)
        cycles_per_us = 10_000_000 / (e0.elapsed_time(e1) * 1e3)
        buf.rehearsal_total_cycles = int(fake_us * cycles_per_us)
    wall, ev_s = time_steps(env, buf, args.steps, args.warmup, torch, dist_on, graph=use_graph, gather_chunks=chunks, force_collective=force_collective)
    gather_note = ""
    if GRAPH_STATE["gather_ok"] is False and GRAPH_STATE["chunks"] > 1:
        # never seen, but an overlapped exchange that delivered stale bytes must not cost the run: say so, fall back to ONE exchange
        print("chunked exchange: the gathered rollout does not match the local shard; timing ONE all-gather instead", file=sys.stderr)
        wall, ev_s = time_steps(env, buf, args.steps, 0, torch, dist_on, graph=use_graph, gather_chunks=1, force_collective=force_collective)
    if GRAPH_STATE["gather_ok"] is False:
        sys.exit("the gathered rollout does not match the local shard: result discarded")
    if dist_on or force_collective:
        gather_note = (", 1 RCCL all-gather of rollouts at the collect boundary" if GRAPH_STATE["chunks"] == 1 else
                       ", rollouts all-gathered (RCCL) in %d step-chunks, each overlapped with the next chunk's steps%s" % (
                           GRAPH_STATE["chunks"], " (one hipGraph, progress marks watched by the host)" if GRAPH_STATE["used"] else " (steps enqueued by tb_step_sequence)"))
        if GRAPH_STATE["tuning"]:
            gather_note += "; exchange form chosen on this node before the clock started, ms per K steps + exchange: %s" % json.dumps(GRAPH_STATE["tuning"])
    c = env.counters()
    if c["nonfinite_states"]:
        sys.exit("%d env states went non-finite or left the lockstep the pipelined kernels rely on: result discarded" % c["nonfinite_states"])

    wall_t = torch.tensor([wall], dtype=torch.float64, device=dev)
    sub_t = torch.tensor([float(c["substeps"])], dtype=torch.float64, device=dev)
    if dist_on:
        torch.distributed.all_reduce(wall_t, op=torch.distributed.ReduceOp.MAX)
        torch.distributed.all_reduce(sub_t, op=torch.distributed.ReduceOp.SUM)
    wall_max = float(wall_t.item())
    timed_substeps = float(sub_t.item())  # time_steps resets the counters right before the timed steps

    result = None
    if rank == 0:
        ab = ALGO_BYTES[args.env]
        per_launch_bytes = (ab["read"] + ab["write"]) * N
        launch_s = ev_s / args.steps
        achieved = per_launch_bytes / launch_s / 1e9
        traffic = None if args.contact_off else pmc_traffic(args.env, N)
        result = {
            "metric": "env steps/sec (whole node), SwingRacket-v0 @4096 envs/GPU" if args.env == "swing" and N == 4096
                      else "env steps/sec (whole node), %s @%d envs/GPU" % ("SwingRacket-v0" if args.env == "swing" else "Tennisbot-v0", N),
            "value": world * N * args.steps / wall_max,
            "unit": "env steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall_max / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s, %d envs/GPU, %s, auto-reset, U(-1,1) actions, rollout buffer %d steps%s%s%s" % (
                "SwingRacket-v0" if args.env == "swing" else "Tennisbot-v0", N,
                ("racket<->ball contact off (configs[1] bench mode)" if args.contact_off else "full contact semantics") + (" + racket<->court contact" if args.racket_ground else "")
                + (" + rolling-friction rows" if args.rolling_friction else "")
                + (" + Magnus k=%g, spin<=%g rad/s (extension, not in the reference)" % (args.magnus, args.spin_max) if (args.magnus or args.spin_max) else ""),
                T_buf, ", fast-forward pipelined on side streams" if pipeline else "", ", K steps replayed as one hipGraph" if (use_graph and GRAPH_STATE["used"]) else "",
                gather_note),
                "envs_per_gpu": N, "global_envs": world * N, "parallelism": "env-sharded x%d" % world},
            "substeps_per_s": timed_substeps / wall_max,
            "substeps_per_agent_step": timed_substeps / (world * N * args.steps),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic[0] if traffic else None,
                         "traffic_unit": "bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE)", "traffic_source": traffic[1] if traffic else None,
                         "algorithmic_bytes_per_launch": per_launch_bytes,
                         "kernel": "tb_step_kernel<%d>" % kind, "launch_us": launch_s * 1e6,
                         "algorithmic_bytes_per_env_step": ab["read"] + ab["write"],
                         "note": "latency-bound at this batch size, not bandwidth-bound (see sweep / DESIGN.md)"},
            "parity": "bit-exact vs the CPU restatement; PyBullet parity unpinned at trajectory level (the reference ships no tests or golden vectors, PyBullet is not available offline), pinned statistically by the 100 PyBullet episodes recorded in the reference's ppo_swing.zip (DESIGN.md section 2)",
        }
    if args.sweep and not dist_on:
        sweep = []
        for n in (4096, 32768, 262144, 1048576, 4194304):
            e2 = BatchedEnv(kind, n, device=dev, seed=args.seed, params=default_params(flags=flags), track_terminal_obs=False, pipeline=pipeline)
            b2 = RolloutBuffer(kind, 26, n, dev)
            fill_actions(b2.actions, args.seed, torch)
            b2.bind(e2)
            e2.reset()
            k = 52 if n >= 1048576 else 104
            w, evs = time_steps(e2, b2, k, 26 if kind == ENV_SWING else 1040, torch, False, tail_gather=False, graph=use_graph)  # Tennisbot: steady state, past the first episodes
            ab = ALGO_BYTES[args.env]
            sweep.append({"envs": n, "steps_per_s": n * k / w, "launch_us": evs / k * 1e6,
                          "achieved_GBs": (ab["read"] + ab["write"]) * n / (evs / k) / 1e9})
            e2.close()
            del b2
            torch.cuda.empty_cache()
        if result is not None:
            result["sweep"] = sweep
            result["open_loop"] = open_loop_rate(kind, N, flags, pipeline, dev, args.seed, torch)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:  # a reported baseline, timed once, next to the 1-GPU figure
        cb = cpu_baseline(args.env, N, args.cpu_seconds, args.seed)
        best = cb["allcores"]
        result["cpu_baseline"] = {
            "value": best["steps_per_s"], "unit": "env steps/s", "cores": best["threads"], "kind": "port",
            "sample": "float32 CPU oracle (oracle/tb_oracle.c, OpenMP over envs), %d envs x %d agent steps (whole 26-step episodes incl. fast-forward), %.1f s"
                      % (N, best["agent_steps"], best["seconds"]),
            "value_1core": cb["1core"]["steps_per_s"], "substeps_per_s": best["substeps_per_s"], "pybullet": cb["pybullet"],
            "substeps_per_s_1core": cb["1core"]["substeps_per_s"],
        }
    if dist_on or force_collective:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()

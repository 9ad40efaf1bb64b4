#!/usr/bin/env python3
"""Where does a chunked rollout lose time? One GPU, 1040 steps at 4096 envs: one hipGraph against C
chunk graphs (RolloutBuffer.capture_chunks), with per-graph HIP-event times and host enqueue times."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from tennisbot_rl_amd.params import ENV_SWING, ENV_TENNIS
from tennisbot_rl_amd.rollout import RolloutBuffer
from tennisbot_rl_amd.stepper import BatchedEnv


def make(kind, pipeline, n=4096, T=1040):
    env = BatchedEnv(kind, n, device="cuda:0", seed=0, track_terminal_obs=False, pipeline=pipeline)
    buf = RolloutBuffer(kind, T, n, "cuda:0").bind(env)
    buf.actions.uniform_(-1, 1)
    env.reset()
    for t in range(52):
        buf.step_into(env, t)
    return env, buf


def run(kind, pipeline, C, window):
    env, buf = make(kind, pipeline)
    graphs, tail = buf.capture_chunks(env, C, defer_window=window)
    seq = graphs + ([tail] if tail is not None else [])
    best = None
    for rep in range(3):
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(seq) + 1)]
        host = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        evs[0].record()
        for k, g in enumerate(seq):
            h0 = time.perf_counter()
            g.replay()
            host.append((time.perf_counter() - h0) * 1e3)
            evs[k + 1].record()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) * 1e3
        gpu = [evs[k].elapsed_time(evs[k + 1]) for k in range(len(seq))]
        if best is None or wall < best[0]:
            best = (wall, gpu, host)
    print("%s pipeline=%d C=%d window=%d: wall %.2f ms | per-graph GPU ms %s | host enqueue ms %s" % (
        "swing" if kind == ENV_SWING else "tennis", pipeline, C, window, best[0],
        " ".join("%.2f" % x for x in best[1]), " ".join("%.2f" % x for x in best[2])))


def main():
    for kind, pipe in ((ENV_TENNIS, False), (ENV_SWING, True)):
        env, buf = make(kind, pipe)
        g = env.capture(lambda: [buf.step_into(env, t) for t in range(buf.T)])
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter(); g.replay(); h = time.perf_counter() - t0
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        print("%s one graph: %.2f ms (host enqueue %.2f ms)" % ("swing" if kind == ENV_SWING else "tennis", best * 1e3, h * 1e3))
    run(ENV_TENNIS, False, 8, 0)
    for C, w in ((4, 78), (8, 78), (8, 52), (16, 65), (8, 104)):
        run(ENV_SWING, True, C, w)


if __name__ == "__main__" and os.environ.get("TB_DIAG_GATHER") != "1":
    main()


def gather_probe():
    """host cost and GPU-side effect of the chunk all-gathers with ONE RCCL rank (TB_DIAG_GATHER=1)"""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0), rank=0, world_size=1)
    for kind, pipe in ((ENV_TENNIS, False), (ENV_SWING, True)):
        env, buf = make(kind, pipe)
        C = 8
        graphs, tail = buf.capture_chunks(env, C)
        buf.begin_gather(C, force=True)
        for c in range(C):
            buf.gather_chunk(c, force=True)
        buf.finish_gather()
        for mode in ("no gather", "gather"):
            best = None
            for rep in range(3):
                torch.cuda.synchronize()
                host, t0 = [], time.perf_counter()
                lag = 1 if tail is not None else 0
                for c, g in enumerate(graphs):
                    g.replay()
                    if mode == "gather" and c >= lag:
                        h0 = time.perf_counter(); buf.gather_chunk(c - lag, force=True); host.append((time.perf_counter() - h0) * 1e3)
                if tail is not None:
                    tail.replay()
                    if mode == "gather":
                        h0 = time.perf_counter(); buf.gather_chunk(C - 1, force=True); host.append((time.perf_counter() - h0) * 1e3)
                t_enq = (time.perf_counter() - t0) * 1e3
                if mode == "gather":
                    buf.finish_gather()
                torch.cuda.synchronize()
                wall = (time.perf_counter() - t0) * 1e3
                if best is None or wall < best[0]:
                    best = (wall, t_enq, host)
            print("%s C=8 %s: wall %.2f ms, host enqueue done at %.2f ms, gather_chunk host ms %s" % (
                "swing" if kind == ENV_SWING else "tennis", mode, best[0], best[1], " ".join("%.2f" % x for x in best[2])))
    dist.destroy_process_group()


if __name__ == "__main__" and os.environ.get("TB_DIAG_GATHER") == "1":
    gather_probe()

#!/usr/bin/env python3
"""Diagnostic: what do progress marks cost a rollout graph, and does side-stream work overlap with it?
  plain graph | marked graph, nobody watching | marked + host waits | marked + host waits + side-stream kernels"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tennisbot_rl_amd.params import ENV_SWING, ENV_TENNIS
from tennisbot_rl_amd.rollout import RolloutBuffer
from tennisbot_rl_amd.stepper import BatchedEnv


def run(kind, chunks, side_us, prio):
    piped = kind == ENV_SWING
    env = BatchedEnv(kind, 4096, device="cuda:0", seed=8, track_terminal_obs=False, pipeline=piped)
    T = 1040
    buf = RolloutBuffer(kind, T, 4096, "cuda:0").bind(env)
    buf.actions.uniform_(-1, 1)
    main = torch.cuda.Stream()
    side = torch.cuda.Stream(priority=prio)
    res = {}
    with torch.cuda.stream(main):
        env.reset()
        for t in range(26):
            buf.step_into(env, t)
        plain = env.capture(lambda: buf.step_range(env, 0, T))
        marked = buf.capture_marked(env, chunks)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(1000); main.synchronize()
        e0.record(); torch.cuda._sleep(10_000_000); e1.record(); main.synchronize()
        cyc = int(side_us * 10_000_000 / (e0.elapsed_time(e1) * 1e3))

        def timed(fn, reps=5):
            best = 1e9
            for _ in range(reps):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                fn()
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            return round(best * 1e3, 2)

        res["plain"] = timed(plain.replay)
        res["marked"] = timed(marked.replay)

        def watched(side_work):
            env.mark_begin()
            marked.replay()
            for c in range(chunks):
                env.mark_host_wait(c)
                if side_work:
                    with torch.cuda.stream(side):
                        torch.cuda._sleep(cyc)
        res["marked+host"] = timed(lambda: watched(False))
        res["marked+host+side %d us x %d" % (side_us, chunks)] = timed(lambda: watched(True))

        def serial():
            plain.replay()
            torch.cuda._sleep(cyc * chunks)
        res["plain then %d us" % (side_us * chunks)] = timed(serial)
    env.close()
    return res


for kind in (ENV_SWING, ENV_TENNIS):
    for prio in (0, -1):
        print("kind", kind, "side priority", prio, run(kind, 8, 500, prio), flush=True)
os._exit(0)

#!/usr/bin/env python3
"""Diagnostic: the usual SwingRacket rollout graph (fast-forwards forked inside the graph) against the linear graph with
hosted fast-forwards (RolloutBuffer.capture_hosted / replay_hosted): bit identity and rate."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tennisbot_rl_amd.params import ENV_SWING
from tennisbot_rl_amd.rollout import RolloutBuffer
from tennisbot_rl_amd.stepper import BatchedEnv

T, N = int(os.environ.get("T", "1040")), int(os.environ.get("N", "4096"))
ref_env = BatchedEnv(ENV_SWING, N, device="cuda:0", seed=8, track_terminal_obs=False, pipeline=True)
env = BatchedEnv(ENV_SWING, N, device="cuda:0", seed=8, track_terminal_obs=False, pipeline=True)
ref = RolloutBuffer(ENV_SWING, T, N, "cuda:0").bind(ref_env)
buf = RolloutBuffer(ENV_SWING, T, N, "cuda:0").bind(env)
buf.actions.uniform_(-1, 1)
ref.actions.copy_(buf.actions)
main = torch.cuda.Stream()
with torch.cuda.stream(main):
    ref_env.reset(); env.reset()
    forked = ref_env.capture(lambda: ref.step_range(ref_env, 0, T))
    hosted = buf.capture_hosted(env)
    print("jobs", env.ff_jobs(), flush=True)
    for rnd in range(6):
        torch.cuda.synchronize()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        t0 = time.perf_counter()
        env.ff_arm(); e0.record(); hosted.replay(); e1.record(); ta = time.perf_counter(); (torch.cuda.current_stream().synchronize() if os.environ.get('SERVICE_LATE') else None); env.ff_service(); tb = time.perf_counter(); env.flush(); e2.record()
        torch.cuda.synchronize(); th = time.perf_counter() - t0
        print("   graph alone %.3f ms on the GPU, with the join %.3f ms; host: replay() returned at %.3f ms, service done at %.3f ms" % (
            e0.elapsed_time(e1), e0.elapsed_time(e2), (ta - t0) * 1e3, (tb - t0) * 1e3))
        t0 = time.perf_counter(); forked.replay(); torch.cuda.synchronize(); tf = time.perf_counter() - t0
        print("round %d: hosted %.3f ms, forked %.3f ms, bit-identical %s, lockstep/non-finite %d" % (
            rnd, th * 1e3, tf * 1e3, bool(torch.equal(ref.raw, buf.raw)), env.counters()["nonfinite_states"]), flush=True)
    print("env steps/s: hosted %.1f M, forked %.1f M; counters equal %s" % (N * T / th / 1e6, N * T / tf / 1e6, env.counters() == ref_env.counters()))
if not os.environ.get("NO_EXIT"):
    os._exit(0)

#!/usr/bin/env python3
"""Does the launch gap of a chain of small dependent kernels depend on whether ANOTHER queue keeps the GPU busy?
SwingRacket with racket<->ball contact off (BASELINE configs[1] as worded) replays slower than the full-contact workload although
both of its kernels are shorter (profiles/r03a*): what differs is how long its fast-forwards keep a side stream busy. Here: the same
1040-step graph replayed (a) plainly, (b) with a kernel that just waits (torch.cuda._sleep) running beside it on another stream for the
whole replay. Also Tennisbot (no side streams at all) and the full-contact workload."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from tennisbot_rl_amd.params import ENV_SWING, ENV_TENNIS, F_DEFAULT, F_NET, default_params
from tennisbot_rl_amd.rollout import RolloutBuffer
from tennisbot_rl_amd.stepper import BatchedEnv

dev = torch.device("cuda", 0)
side = torch.cuda.Stream(device=dev)
# cycles per microsecond of torch.cuda._sleep
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda._sleep(1000); torch.cuda.synchronize()
e0.record(); torch.cuda._sleep(10_000_000); e1.record(); torch.cuda.synchronize()
cyc_per_us = 10_000_000 / (e0.elapsed_time(e1) * 1e3)
out = {}
for name, kind, flags in (("swing_contact_off", ENV_SWING, F_NET), ("swing_full", ENV_SWING, F_DEFAULT), ("tennis", ENV_TENNIS, F_DEFAULT)):
    env = BatchedEnv(kind, 4096, device=dev, seed=0, params=default_params(flags=flags), track_terminal_obs=False, pipeline=kind == ENV_SWING)
    buf = RolloutBuffer(kind, 1040, 4096, dev); buf.actions.uniform_(-1, 1); buf.bind(env); env.reset()
    for t in range(1040): buf.step_into(env, t)
    env.flush()
    g = env.capture(lambda: buf.step_range(env, 0, 1040))
    for _ in range(150): g.replay()
    torch.cuda.synchronize()
    res = {}
    for rnd in range(3):
        for mode in ("plain", "busy_side_stream", "plain2"):
            ts = []
            for k in range(20):
                torch.cuda.synchronize()
                if mode == "busy_side_stream":
                    with torch.cuda.stream(side):
                        torch.cuda._sleep(int(9000 * cyc_per_us))  # ~9 ms: longer than the replay
                t0 = time.perf_counter(); g.replay(); torch.cuda.current_stream().synchronize(); ts.append(time.perf_counter() - t0)
            ts.sort()
            res.setdefault(mode, []).append(round(4096 * 1040 / ts[10] / 1e6, 1))
    out[name] = res
    print(name, json.dumps(res), flush=True)
    env.close()
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r03_busy_probe.json"), "w"), indent=1)

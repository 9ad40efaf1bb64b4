#!/usr/bin/env python3
"""Do RESIDENT WAVES shorten the dispatch gap between the dependent launches of a graph? (GPU) The full-contact SwingRacket workload
replays faster than the same envs with racket<->ball contact off although the latter's kernels are shorter: un-profiled, the gap
between consecutive step kernels is 1.7-2.8 us with full contacts (fast-forward waves resident for 370 us after every episode end)
and a flat 3.0 us without (they leave after 94 us), 3.5 us for Tennisbot (no fast-forward at all) -- tools/diag/r03_cadence_probe.py.
Here: the Tennisbot and contact-off graphs replayed with tb_diag_idle waves (s_sleep only) resident beside them."""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from tennisbot_rl_amd import stepper
from tennisbot_rl_amd.params import ENV_SWING, ENV_TENNIS, F_DEFAULT, F_NET, default_params
from tennisbot_rl_amd.rollout import RolloutBuffer
from tennisbot_rl_amd.stepper import BatchedEnv
L = stepper.load_library()
dev = torch.device("cuda", 0)
side = torch.cuda.Stream(device=dev)
out = {}
for name, kind, flags in (("tennis", ENV_TENNIS, F_DEFAULT), ("swing_contact_off", ENV_SWING, F_NET), ("swing_full", ENV_SWING, F_DEFAULT)):
    env = BatchedEnv(kind, 4096, device=dev, seed=0, params=default_params(flags=flags), track_terminal_obs=False, pipeline=kind == ENV_SWING)
    buf = RolloutBuffer(kind, 1040, 4096, dev); buf.actions.uniform_(-1, 1); buf.bind(env); env.reset()
    for t in range(1040): buf.step_into(env, t)
    env.flush()
    g = env.capture(lambda: buf.step_range(env, 0, 1040))
    for _ in range(150): g.replay()
    torch.cuda.synchronize()
    res = {}
    for rnd in range(2):
        for waves in (0, 16, 64, 256, 1024, 0):
            ts = []
            for k in range(16):
                torch.cuda.synchronize()
                if waves:
                    rc = L.tb_diag_idle(waves, 8000, 0, ctypes.c_void_p(side.cuda_stream))
                    assert rc == 0
                t0 = time.perf_counter(); g.replay(); torch.cuda.current_stream().synchronize(); ts.append(time.perf_counter() - t0)
            ts.sort()
            res.setdefault("idle_waves_%d" % waves, []).append(round(4096 * 1040 / ts[8] / 1e6, 1))
            torch.cuda.synchronize()
    out[name] = res
    print(name, json.dumps(res), flush=True)
    env.close()
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r03_idle_probe.json"), "w"), indent=1)

#!/usr/bin/env python3
"""Same-box A/B of the library's two kernel builds (TbOptions.kernel_build: 1 packed, 2 unpacked, 3 packed steps + unpacked loops,
0 auto) over the workloads DESIGN.md quotes, each in a process of its own. Run on the GPU box."""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 2 and sys.argv[1] == "--child":
    build, what = int(sys.argv[2]), sys.argv[3]
    import torch
    from tennisbot_rl_amd.params import ENV_SWING, ENV_TENNIS, F_DEFAULT, F_RACKET_GROUND, default_params
    from tennisbot_rl_amd.rollout import RolloutBuffer
    from tennisbot_rl_amd.stepper import BatchedEnv
    dev = torch.device("cuda", 0)
    kind, n, T, flags = {"swing4096": (ENV_SWING, 4096, 1040, F_DEFAULT), "swing1024": (ENV_SWING, 1024, 1040, F_DEFAULT), "swing8192": (ENV_SWING, 8192, 1040, F_DEFAULT),
                         "swing16k": (ENV_SWING, 16384, 1040, F_DEFAULT), "swing32k": (ENV_SWING, 32768, 1040, F_DEFAULT), "swing64k": (ENV_SWING, 65536, 104, F_DEFAULT),
                         "swing128k": (ENV_SWING, 131072, 104, F_DEFAULT), "swing256k": (ENV_SWING, 262144, 104, F_DEFAULT), "swing1m": (ENV_SWING, 1048576, 104, F_DEFAULT),
                         "tennis4096": (ENV_TENNIS, 4096, 1040, F_DEFAULT), "tennis32k": (ENV_TENNIS, 32768, 1040, F_DEFAULT), "tennis1m": (ENV_TENNIS, 1048576, 104, F_DEFAULT),
                         "rg4096": (ENV_SWING, 4096, 1040, F_DEFAULT | F_RACKET_GROUND)}[what]
    env = BatchedEnv(kind, n, device=dev, seed=0, params=default_params(flags=flags), track_terminal_obs=False, pipeline=kind == ENV_SWING, options=dict(kernel_build=build))
    buf = RolloutBuffer(kind, T, n, dev); torch.manual_seed(0); buf.actions.uniform_(-1, 1); buf.bind(env); env.reset()
    for t in range(T): buf.step_into(env, t)
    env.flush()
    g = env.capture(lambda: buf.step_range(env, 0, T))
    t_end = time.perf_counter() + (1.0 if n <= 32768 else 0.2)
    while time.perf_counter() < t_end: g.replay(); torch.cuda.synchronize()
    ts = []
    for k in range(9 if n <= 32768 else 5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts.sort()
    print(json.dumps({"what": what, "rate_M": round(n * T / ts[len(ts) // 2] / 1e6, 1)})); sys.exit(0)
for what in sys.argv[1:] or ("swing1024", "swing4096", "swing8192", "swing16k", "swing32k", "swing64k", "swing128k", "swing256k", "swing1m", "tennis4096", "tennis32k", "tennis1m", "rg4096"):
    row = {}
    for build in (1, 2, 3, 0):
        r = subprocess.run([sys.executable, __file__, "--child", str(build), what], capture_output=True, text=True)
        try:
            row[{1: "packed", 2: "unpacked", 3: "packed steps + unpacked loops", 0: "auto"}[build]] = json.loads(r.stdout.strip().splitlines()[-1])["rate_M"]
        except Exception:
            row[build] = "failed: " + r.stderr[-300:]
    print(what, json.dumps(row), flush=True)

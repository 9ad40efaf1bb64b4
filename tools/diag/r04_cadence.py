#!/usr/bin/env python3
"""kernel_us and gap_us of bench.py's roofline: what a step launch spends INSIDE the kernel and BETWEEN two dependent launches of the
replayed rollout graph, measured un-profiled on a build that runs at the product's rate (-DTB_DIAG_CADENCE: the first thread of a
launch reads the 100 MHz real-time counter on entry and exit and stores both at its end; no atomic -- tb_diag.hpp).
Each workload in a process of its own, per build: the product library first and last (what the box gives without the trace, twice:
its spread), the cadence build in between. Writes gpurun_out/r04_cadence.json; the judged copy is profiles/r04_cadence.json.
Run on the GPU box:   python tools/diag/r04_cadence.py"""
import ctypes, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
T, N = 1040, 4096

if len(sys.argv) > 3 and sys.argv[1] == "--child":
    lib, what = sys.argv[2], sys.argv[3]
    import numpy as np
    import torch
    from tennisbot_rl_amd import stepper
    if lib != "product":
        stepper.use_library(lib)
    from tennisbot_rl_amd.params import ENV_SWING, ENV_TENNIS
    from tennisbot_rl_amd.rollout import RolloutBuffer
    from tennisbot_rl_amd.stepper import BatchedEnv
    L = stepper.load_library()
    dev = torch.device("cuda", 0)
    kind = ENV_TENNIS if what == "tennis" else ENV_SWING
    env = BatchedEnv(kind, N, device=dev, seed=0, track_terminal_obs=False, pipeline=kind == ENV_SWING)
    buf = RolloutBuffer(kind, T, N, dev); torch.manual_seed(0); buf.actions.uniform_(-1, 1); buf.bind(env); env.reset()
    for t in range(T): buf.step_into(env, t)
    env.flush()
    g = env.capture(lambda: buf.step_range(env, 0, T))
    t_end = time.perf_counter() + 1.5  # the settle time of bench.py
    while time.perf_counter() < t_end: g.replay(); torch.cuda.synchronize()
    tr = (ctypes.c_ulonglong * (2 * 8192))()
    if lib != "product":
        L.tb_diag_read_cadence.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.tb_diag_read_cadence(tr, 1)
    ts = []
    for _ in range(15):
        torch.cuda.synchronize(); t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    last = ts[-1]
    ts.sort()
    res = {"build": "product" if lib == "product" else "cadence", "workload": what, "rate_M": N * T / ts[len(ts) // 2] / 1e6, "replay_us_per_step": ts[len(ts) // 2] / T * 1e6,
           "pipeline_form": env.pipeline_form()}
    if lib != "product":
        # the launches of the LAST of the timed replays (the ring holds the latest 8192 launches)
        n = L.tb_diag_read_cadence(tr, 1)
        assert n == 15 * T, n
        ring = np.array(list(tr), dtype=np.float64).reshape(8192, 2) * 0.01  # microseconds
        a = ring[np.arange(n - T, n) % 8192]
        dur, s2s, gap = a[:, 1] - a[:, 0], np.diff(a[:, 0]), a[1:, 0] - a[:-1, 1]
        res.update(launches_traced=int(T), traced_replay_us_per_step=last / T * 1e6, kernel_us=float(dur.mean()), kernel_us_p50=float(np.median(dur)),
                   start_to_start_us=float(s2s.mean()), gap_us=float(gap.mean()), gap_us_p50=float(np.median(gap)), replay_span_us=float(a[-1, 1] - a[0, 0]),
                   join_share_us=last / T * 1e6 - float(s2s.mean()),
                   clock="s_memrealtime, 100 MHz: 0.01 us resolution per stamp; kernel_us = entry to exit of the launch's first thread")
        if kind == ENV_SWING:
            res["kernel_us_steps_1_to_25"] = float(dur.reshape(-1, 26)[:, :25].mean())
            res["kernel_us_parking_step"] = float(dur.reshape(-1, 26)[:, 25].mean())
    print(json.dumps(res)); sys.exit(0)

from tennisbot_rl_amd.build import HIPCC_FLAGS, SOURCES, hipcc
lib = "/tmp/libtb_cadence.so"
subprocess.check_call([hipcc()] + HIPCC_FLAGS + ["-DTB_DIAG_CADENCE", "-o", lib] + SOURCES)
out = {}
for what in ("swing", "tennis"):
    rows = []
    for which in ("product", lib, "product"):
        r = subprocess.run([sys.executable, __file__, "--child", which, what], capture_output=True, text=True)
        try:
            rows.append(json.loads(r.stdout.strip().splitlines()[-1]))
        except Exception:
            rows.append({"failed": r.stderr[-400:]})
        print(what, json.dumps(rows[-1]), flush=True)
    cad = rows[1]
    prod = [x["rate_M"] for x in (rows[0], rows[2]) if "rate_M" in x]
    if "kernel_us" in cad and prod:
        out[what] = {"envs": N, "rollout_steps": T, "kernel_us": cad["kernel_us"], "gap_us": cad["gap_us"], "start_to_start_us": cad["start_to_start_us"],
                     "join_share_us": cad["join_share_us"],
                     "cadence_build_rate_M": cad["rate_M"], "product_rate_same_box_M": prod, "cadence_build_vs_product": cad["rate_M"] / (sum(prod) / len(prod)),
                     "detail": cad}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r04_cadence.json"), "w"), indent=1)

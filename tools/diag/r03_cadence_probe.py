#!/usr/bin/env python3
"""Why does SwingRacket with racket<->ball contact OFF (BASELINE configs[1] as worded) replay slower than the full-contact workload
although both of its kernels are shorter under rocprof? The cadence of the step-kernel launches inside ONE un-profiled graph replay
(diagnostic build: the first thread of each launch logs the 100 MHz real-time counter at entry and exit) and the shader clock
while they ran (s_memtime / s_memrealtime), both workloads in one process."""
import ctypes, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from tennisbot_rl_amd import stepper
from tennisbot_rl_amd.build import HIPCC_FLAGS, SOURCES, hipcc
STAMPS = "stamps" in sys.argv[1:]  # also the in-kernel cycle stamps (shader clock; perturbs the timing a lot)
lib = "/tmp/libtb_trace.so"
subprocess.check_call([hipcc()] + HIPCC_FLAGS + ["-DTB_DIAG_STAMPS" if STAMPS else "-DTB_DIAG_TRACE", "-o", lib] + SOURCES)
stepper.use_library(lib)
from tennisbot_rl_amd.params import ENV_SWING, ENV_TENNIS, F_DEFAULT, F_NET, default_params
from tennisbot_rl_amd.rollout import RolloutBuffer
from tennisbot_rl_amd.stepper import BatchedEnv
L = stepper.load_library()
if STAMPS:
    L.tb_diag_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
L.tb_diag_read_trace.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
if not STAMPS:
    L.tb_diag_read_trace_all_out.argtypes = [ctypes.c_void_p, ctypes.c_int]
dev = torch.device("cuda", 0)
T = 1040
out = {}
for name, kind, flags in (("swing_full", ENV_SWING, F_DEFAULT), ("swing_contact_off", ENV_SWING, F_NET), ("tennis", ENV_TENNIS, F_DEFAULT), ("swing_full_again", ENV_SWING, F_DEFAULT)):
    env = BatchedEnv(kind, 4096, device=dev, seed=0, params=default_params(flags=flags), track_terminal_obs=False, pipeline=kind == ENV_SWING)
    buf = RolloutBuffer(kind, T, 4096, dev); buf.actions.uniform_(-1, 1); buf.bind(env); env.reset()
    for t in range(T): buf.step_into(env, t)
    env.flush()
    g = env.capture(lambda: buf.step_range(env, 0, T))
    for _ in range(100): g.replay()
    torch.cuda.synchronize()
    ts = []
    for k in range(15):
        torch.cuda.synchronize(); t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts.sort()
    st = (ctypes.c_ulonglong * 16)()
    tr = (ctypes.c_ulonglong * (2 * 8192))()
    if STAMPS: L.tb_diag_read_stamps(st, 1)
    L.tb_diag_read_trace(tr, 8192, 1)
    g.replay(); torch.cuda.synchronize()
    if STAMPS: L.tb_diag_read_stamps(st, 1)
    allout = (ctypes.c_ulonglong * 8192)()
    if not STAMPS:
        L.tb_diag_read_trace_all_out(allout, 8192)
    n = L.tb_diag_read_trace(tr, 8192, 1)
    a = np.array(list(tr[: 2 * n]), dtype=np.float64).reshape(n, 2) * 0.01  # microseconds
    ao = np.array(list(allout[:n]), dtype=np.float64) * 0.01
    order = np.argsort(a[:, 0])
    a, ao = a[order], ao[order]
    dur = a[:, 1] - a[:, 0]
    s2s = np.diff(a[:, 0])
    gap = a[1:, 0] - a[:-1, 1]
    res = {"rate_M": 4096 * T / ts[len(ts) // 2] / 1e6, "launches_traced": int(n), "replay_span_us": float(a[-1, 1] - a[0, 0]),
           "duration_us_mean": float(dur.mean()), "start_to_start_us_mean": float(s2s.mean()), "gap_us_mean": float(gap.mean()), "gap_us_p50": float(np.median(gap))}
    if not STAMPS:
        ok = ao[1:] > 0
        # launch k+1's first thread saw when the LAST workgroup of launch k left: how long the launch's slowest workgroup outlives its first,
        # and what is left of the gap once every workgroup has gone
        res.update(last_workgroup_after_first_us_mean=float((ao[1:] - a[:-1, 1])[ok].mean()), gap_after_last_workgroup_us_mean=float((a[1:, 0] - ao[1:])[ok].mean()))
    if STAMPS:
        res.update(shader_clock_GHz=st[8] / max(st[14], 1) * 0.1, kernel_cycles_per_wave=st[8] / max(st[9], 1))
    if kind == ENV_SWING and n == T:
        res["duration_by_position"] = [round(float(x), 2) for x in dur.reshape(-1, 26).mean(0)]
        res["start_to_start_by_position"] = [round(float(x), 2) for x in np.append(s2s, s2s.mean()).reshape(-1, 26).mean(0)]
        res["gap_by_position"] = [round(float(x), 2) for x in np.append(gap, gap.mean()).reshape(-1, 26).mean(0)]
    out[name] = res
    print(name, json.dumps(res), flush=True)
    env.close()
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r03_cadence_probe.json"), "w"), indent=1)

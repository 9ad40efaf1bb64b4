#!/usr/bin/env python3
"""The floor under a chain of dependent kernel nodes in a replayed hipGraph on this GPU: 1040 minimal kernels (tb_diag_stream_copy of
one 64-word row: one wave that loads and stores 256 bytes) captured as one chain, against the SwingRacket step chain of bench.py.
Run on the GPU box."""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from tennisbot_rl_amd import stepper
L = stepper.load_library()
L.tb_diag_stream_copy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda", 0)
out = {}
for n, rows, label in ((64, 1, "one wave, 256 B"), (4096, 1, "64 waves, 16 KB"), (4096, 30, "64 waves, 30 rows = the SwingRacket state (480 KB)")):
    a = torch.zeros(rows * n, dtype=torch.int32, device=dev); b = torch.zeros_like(a)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for k in range(1040):
                src, dst = (a, b) if k % 2 == 0 else (b, a)  # (each node depends on the one before it)
                L.tb_diag_stream_copy(src.data_ptr(), dst.data_ptr(), n, rows, 0, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        for _ in range(50): g.replay()
        torch.cuda.synchronize(); ts = []
        for _ in range(15):
            t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts.sort()
    out[label] = ts[7] / 1040 * 1e6
    print("chain of 1040 copy kernels, %s: %.2f us per node" % (label, out[label]), flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r03_launch_floor.json"), "w"), indent=1)

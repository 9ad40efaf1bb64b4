#!/usr/bin/env python3
"""What the step kernels' access pattern (SoA rows, one dword per lane) can move: tb_diag_copy_kernel, 30 rows each way, timed
back to back at 1 M and 4 M envs (GPU box). The Tennisbot step kernel at 1 M envs moves 280 MB in ~53 us = 5.3 TB/s."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from tennisbot_rl_amd.stepper import load_library  # noqa: E402

L = load_library()
dev = torch.device("cuda:0")
for n, rows in ((1 << 20, 30), (1 << 20, 35), (1 << 22, 30)):
    a = torch.randint(0, 2 ** 31 - 1, (rows, n), dtype=torch.int32, device=dev)
    b = torch.empty_like(a)
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        L.tb_diag_stream_copy(a.data_ptr(), b.data_ptr(), n, rows, 0, s)
    torch.cuda.synchronize()
    reps = 50
    t0 = time.perf_counter()
    for _ in range(reps):
        L.tb_diag_stream_copy(a.data_ptr(), b.data_ptr(), n, rows, 0, s)
        L.tb_diag_stream_copy(b.data_ptr(), a.data_ptr(), n, rows, 0, s)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (2 * reps)
    print("n=%d rows=%d: %.1f us per launch, %.2f TB/s (read + write)" % (n, rows, dt * 1e6, 2 * rows * n * 4 / dt / 1e12))

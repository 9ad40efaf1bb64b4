#!/usr/bin/env python3
"""Known-byte-count launches in the step kernel's access pattern (one dword per lane, SoA
rows), to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 before they are read as
HBM traffic (MI355X_MICROARCH.md: FETCH_SIZE is only calibrated for 16 B/lane streams).
Run under `rocprofv3 --pmc FETCH_SIZE` and again under `--pmc WRITE_SIZE`; each launch of
tb_diag_copy_kernel moves exactly rows*n*4 bytes each way (printed below)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from tennisbot_rl_amd.stepper import load_library  # noqa: E402

L = load_library()
dev = torch.device("cuda:0")
for n, rows in ((4096, 30), (1 << 20, 30), (1 << 22, 30)):
    src = torch.randint(0, 2 ** 31 - 1, (rows, n), dtype=torch.int32, device=dev)
    dst = torch.empty_like(src)
    # flush caches between launches with a large unrelated fill
    junk = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    for rep in range(3):
        junk.fill_(rep)
        rc = L.tb_diag_stream_copy(src.data_ptr(), dst.data_ptr(), n, rows, 0, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(src, dst)
    print("n=%d rows=%d bytes_each_way=%d" % (n, rows, rows * n * 4))

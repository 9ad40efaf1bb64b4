#!/usr/bin/env python3
"""PCIe-inclusive rate of the numpy-facing SB3-VecEnv surface (actions up, obs / reward / done down
every step) next to the device-tensor path: python tools/time_vecenv.py [num_envs]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from tennisbot_rl_amd.envs import TennisbotVecEnv


def main(n):
    for env_id, A in (("SwingRacket-v0", 6), ("Tennisbot-v0", 2)):
        env = TennisbotVecEnv(env_id, n)
        env.reset()
        acts = np.random.default_rng(0).uniform(-1, 1, (64, n, A)).astype(np.float32)
        for t in range(30):
            env.step(acts[t % 64])
        steps = 260
        t0 = time.perf_counter()
        for t in range(steps):
            env.step(acts[t % 64])
        dt = time.perf_counter() - t0
        dev = torch.from_numpy(acts).cuda()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for t in range(steps):
            env.tensor_step(dev[t % 64])
        env.batch.flush() if env.batch.pipeline else None
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t1
        print("%s, %d envs: numpy VecEnv.step %.2f M env steps/s (%.0f us per batch step, PCIe both ways + host sync + infos); "
              "tensor_step from Python, no graph, no pipeline %.1f M env steps/s" % (env_id, n, n * steps / dt / 1e6, dt / steps * 1e6, n * steps / dt2 / 1e6))
        env.close()


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 4096)

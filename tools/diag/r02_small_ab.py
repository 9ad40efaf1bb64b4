import sys, time
sys.path.insert(0, "/root/repo")
import torch
from tennisbot_rl_amd import stepper
if len(sys.argv) > 1: stepper.use_library(sys.argv[1])
from tennisbot_rl_amd.params import ENV_SWING, ENV_TENNIS
from tennisbot_rl_amd.rollout import RolloutBuffer
from tennisbot_rl_amd.stepper import BatchedEnv
dev = torch.device("cuda", 0)
for kind in (ENV_SWING, ENV_TENNIS):
    env = BatchedEnv(kind, 4096, device=dev, seed=0, track_terminal_obs=False, pipeline=kind == ENV_SWING)
    buf = RolloutBuffer(kind, 1040, 4096, dev); buf.actions.uniform_(-1, 1); buf.bind(env); env.reset()
    for t in range(1040): buf.step_into(env, t)
    env.flush()
    g = env.capture(lambda: buf.step_range(env, 0, 1040))
    torch.cuda.synchronize()
    out = []
    for k in range(30):
        t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); out.append(time.perf_counter() - t0)
    print(" ".join("%.0f" % (4096 * 1040 / x / 1e6) for x in out))
    out.sort()
    print(sys.argv[1:] , "swing" if kind == ENV_SWING else "tennis", "median %.1f M, best %.1f M" % (4096 * 1040 / out[15] / 1e6, 4096 * 1040 / out[0] / 1e6))
    env.close()

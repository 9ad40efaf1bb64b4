import sys, time
sys.path.insert(0, "/root/repo")
import torch
from tennisbot_rl_amd.params import ENV_SWING
from tennisbot_rl_amd.rollout import RolloutBuffer
from tennisbot_rl_amd.stepper import BatchedEnv
dev = torch.device("cuda", 0)
env = BatchedEnv(ENV_SWING, 4096, device=dev, seed=0, track_terminal_obs=False, pipeline=True)
buf = RolloutBuffer(ENV_SWING, 1040, 4096, dev); buf.actions.uniform_(-1, 1); buf.bind(env); env.reset()
for t in range(26): buf.step_into(env, t)
env.flush()
g = env.capture(lambda: buf.step_range(env, 0, 1040))
torch.cuda.synchronize()
out = []
for k in range(40):
    t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); out.append(time.perf_counter() - t0)
print(" ".join("%.0f" % (4096 * 1040 / x / 1e6) for x in out))
time.sleep(1.0)
out = []
for k in range(10):
    t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); out.append(time.perf_counter() - t0)
print("after 1 s idle:", " ".join("%.0f" % (4096 * 1040 / x / 1e6) for x in out))

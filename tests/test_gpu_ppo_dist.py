"""Data-parallel PPO over two ranks (one process per rank, both on cuda:0, gloo for the control
flow -- RCCL refuses two ranks on one device): env shards keyed by global env id, gradients averaged by
all-reduce, so both ranks must finish with identical weights while having stepped different envs."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from tennisbot_rl_amd.ppo import PPOTrainer
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=2)
    tr = PPOTrainer("SwingRacket-v0", num_envs=512, n_steps=26, device="cuda:0", seed=5, batch_size=6656)
    assert tr.world == 2 and tr.rank == rank and tr.env.env_id_base == rank * 512
    hist = tr.learn(2 * 512 * 26 * 3, log=None)  # three rollouts of the 1024-env global batch
    flat = torch.cat([p.detach().reshape(-1) for p in tr.policy.parameters()]).cpu().numpy()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), weights=flat, obs=tr.buf.obs.cpu().numpy(), timesteps=tr.num_timesteps,
             reward=np.float64(hist[-1]["mean_episode_reward"]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_ppo_keeps_replicas_in_sync(tmp_path):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(_free_port(), str(tmp_path)), nprocs=2, join=True)
    a, b = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert a["timesteps"] == b["timesteps"] == 2 * 512 * 26 * 3
    assert np.array_equal(a["weights"], b["weights"]), "replicas diverged: gradients were not averaged identically"
    assert not np.array_equal(a["obs"], b["obs"]), "both ranks stepped the same envs"
    assert np.isfinite(a["weights"]).all()

#!/usr/bin/env python3
"""Lockstep soak at BASELINE's full batch sizes (run on the GPU box; the suite's largest lockstep case is 200 003 envs):
N envs stepped `steps` times with random actions, every observation / done / reward / substep counter and the final state
words compared BIT FOR BIT with the f32 CPU oracle (16 host threads). usage: tests/soak_parity.py <swing|tennis> <n_envs> <steps> [rg] [defer|defer_all]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import OracleBatch  # noqa: E402  (a checker script kept with the tests: not collected by pytest, run by hand on the GPU box)
from tennisbot_rl_amd.params import ENV_SWING, ENV_TENNIS, F_AUTO_RESET, F_DEFAULT, F_RACKET_GROUND, STATE_WORDS, default_params  # noqa: E402
from tennisbot_rl_amd.stepper import BatchedEnv  # noqa: E402


def same(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    ok = np.array_equal(a.view(np.uint32), b.view(np.uint32)) if a.dtype.kind == "f" else np.array_equal(a, b)
    if not ok:
        bad = np.argwhere(a != b)
        raise SystemExit("MISMATCH %s: %d values, first at %s: %r vs %r" % (what, len(bad), bad[0], a[tuple(bad[0])], b[tuple(bad[0])]))


def main():
    kind = ENV_SWING if sys.argv[1] == "swing" else ENV_TENNIS
    n, steps = int(sys.argv[2]), int(sys.argv[3])
    flags = F_DEFAULT | (F_RACKET_GROUND if "rg" in sys.argv[4:] else 0)
    p = default_params(flags=flags)
    # "defer" / "defer_all": TbOptions.ff_defer = 1 / 2 (deferred stragglers; auto-on with rg up to 131072 envs)
    opts = dict(ff_defer="all") if "defer_all" in sys.argv[4:] else dict(ff_defer=True) if "defer" in sys.argv[4:] else None
    env = BatchedEnv(kind, n, device="cuda:0", seed=77, params=p, pipeline=kind == ENV_SWING, track_terminal_obs=False, options=opts)
    pf = p.copy(); pf.flags |= F_AUTO_RESET
    ref = OracleBatch(pf, kind, n, seed=77, precision="f32")
    ref.L.tbo_set_threads(ref.h, 16)
    rng = np.random.default_rng(5)
    same(env.reset().cpu().numpy(), ref.reset(), "reset obs")
    late, t_gpu, t_cpu, compared = [], 0.0, 0.0, 0
    for t in range(steps):
        a = rng.uniform(-1, 1, (n, env.act_dim)).astype(np.float32)
        t0 = time.perf_counter()
        obs, rew, done = env.step(torch.from_numpy(a).cuda()); obs_h, done_h = obs.cpu().numpy(), done.cpu().numpy()
        t1 = time.perf_counter()
        o2, r2, d2, _ = ref.step(a)
        t_gpu += t1 - t0; t_cpu += time.perf_counter() - t1
        same(obs_h, o2, "obs at step %d" % t); same(done_h, d2, "done at step %d" % t)
        if kind == ENV_SWING:
            late.append((rew, r2.copy()))  # terminal rewards arrive late from the side streams (deferred stragglers: at the flush), into THIS buffer
        else:
            same(rew.cpu().numpy(), r2, "reward at step %d" % t)
        compared += n
    env.flush()
    for t, (rew, r2) in enumerate(late):
        same(rew.cpu().numpy(), r2, "reward at step %d" % t)
    w_gpu, d_gpu = env.get_state_words()
    w_cpu, d_cpu = ref.get_state_words()
    same(w_gpu.cpu().numpy().view(np.uint32), w_cpu.view(np.uint32), "final state words")
    same(d_gpu.cpu().numpy(), d_cpu, "final done bytes")
    got, want = env.counters(), [int(x) for x in ref.counters()]
    if list(got.values()) != want:
        raise SystemExit("MISMATCH counters: %r vs %r" % (got, want))
    print("%s, %d envs x %d steps (flags 0x%x%s): %d env-steps, %d substeps, %d episode ends, every output and the final state bit-identical to the f32 oracle "
          "(host loop incl. copies: GPU %.1f s, oracle on 16 threads %.1f s)" % (sys.argv[1], n, steps, flags, ", " + " ".join(sys.argv[5:] if "rg" in sys.argv[4:5] else sys.argv[4:]) if opts else "", compared, got["substeps"], got["episodes_finished"], t_gpu, t_cpu))


if __name__ == "__main__":
    main()

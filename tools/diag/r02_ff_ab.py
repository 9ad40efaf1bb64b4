#!/usr/bin/env python3
"""A/B of the fast-forward kernel's lane assignment on ONE box (boxes differ by several percent): whole-rollout
rate of SwingRacket-v0 for TbOptions.ff_lanes_per_wave / ff_sort variants, plus the one-episode rollout (26 steps +
join) that exposes the fast-forward kernel's own duration."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch


def measure(n, T, opts, reps=5, flags=None, kind=None):
    from tennisbot_rl_amd.params import ENV_SWING, ENV_TENNIS, F_DEFAULT, default_params
    ENV_SWING = ENV_SWING if kind is None else kind
    from tennisbot_rl_amd.rollout import RolloutBuffer
    from tennisbot_rl_amd.stepper import BatchedEnv
    dev = torch.device("cuda", 0)
    env = BatchedEnv(ENV_SWING, n, device=dev, seed=0, params=default_params(flags=F_DEFAULT if flags is None else flags), track_terminal_obs=False, pipeline=kind is None, options=opts)
    buf = RolloutBuffer(ENV_SWING, T, n, dev)
    torch.manual_seed(0)  # the same actions for every variant: their counters must agree
    buf.actions.uniform_(-1.0, 1.0)
    buf.bind(env)
    env.reset()
    for t in range(26 if kind is None else 1040):
        buf.step_into(env, t % T)
    env.flush()
    g = env.capture(lambda: buf.step_range(env, 0, T))
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    # one episode + join: 26 step launches, then nothing to hide the fast-forward behind
    g1 = env.capture(lambda: buf.step_range(env, 0, 26))
    g1.replay(); torch.cuda.synchronize()
    one = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); g1.replay(); torch.cuda.synchronize()
        one = min(one, time.perf_counter() - t0)
    c = env.counters()
    env.close(); del buf
    torch.cuda.empty_cache()
    return {"n": n, "T": T, "opts": opts, "steps_per_s": n * T / best, "us_per_step": best / T * 1e6, "one_episode_ms": one * 1e3, "timeouts": c["timeouts"]}


def main():
    """usage: r02_ff_ab.py <small|large|all|lanes|tennis1m|opts1m|n=<envs>[:<steps>]> [lib=<path to another build of libtb_stepper.so>] [rg] [-D...]
    lib= measures that build instead of the in-tree one (e.g. an earlier commit's, built into /tmp: the same-box A/B); rg sets
    TB_F_RACKET_GROUND; -D... builds that variant of the present sources into /tmp and measures it"""
    from tennisbot_rl_amd.params import F_NET, F_RACKET_BALL, F_DEFAULT
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    from tennisbot_rl_amd.params import F_RACKET_GROUND
    flags = F_DEFAULT | (F_RACKET_GROUND if "rg" in sys.argv else 0)
    other = [x[4:] for x in sys.argv[2:] if x.startswith("lib=")]
    if other:
        from tennisbot_rl_amd import stepper
        stepper.use_library(other[0])
    extra = [x for x in sys.argv[2:] if x.startswith("-D")]
    if extra:  # extra hipcc flags: build that variant of the library into /tmp and measure it instead (same-box A/B)
        import subprocess
        from tennisbot_rl_amd import stepper
        from tennisbot_rl_amd.build import HIPCC_FLAGS, SOURCES, hipcc
        lib = "/tmp/libtb_variant.so"
        subprocess.check_call([hipcc()] + HIPCC_FLAGS + extra + ["-o", lib] + SOURCES)
        stepper.use_library(lib)
    print("variant", sys.argv[2:], "flags", hex(flags), flush=True)
    out = []
    if which == "lanes":  # parked envs per fast-forward wave at the headline batch size
        for o in ({}, dict(ff_lanes_per_wave=32), dict(ff_lanes_per_wave=16), {}, dict(ff_lanes_per_wave=32), dict(ff_lanes_per_wave=48)):
            out.append(measure(4096, 1040, o, flags=flags)); print(json.dumps(out[-1]), flush=True)
    if which == "defer":  # the random-action headline with the stragglers' pool: off (auto), margin form, every episode end deferred (a pure chain of steps)
        for o in ({}, dict(ff_defer=True), dict(ff_defer="all"), {}, dict(ff_defer=True), dict(ff_defer="all")):
            out.append(measure(4096, 1040, o, flags=flags)); print(json.dumps(out[-1]), flush=True)
    if which == "defer_ladder":  # where does deferring every episode end to one pool run at the join stop paying?
        for n in (1024, 4096, 8192, 16384, 32768, 65536, 131072):
            for o in ({}, dict(ff_defer="all"), dict(ff_defer=False)):
                out.append(measure(n, 1040 if n <= 32768 else 104, o, reps=3, flags=flags)); print(json.dumps(out[-1]), flush=True)
    if which == "retune":  # the size thresholds of tb_create / defer_mode once more (after a change of build flags): static rows in registers, fast-forward phases
        for n in (65536, 131072, 262144, 1048576):
            for o in (dict(swing_reg_rows=True), dict(swing_reg_rows=False)):
                out.append(measure(n, 104, o, reps=3, flags=flags)); print(json.dumps(out[-1]), flush=True)
        for n in (65536, 131072, 262144):
            for o in (dict(ff_phases=1), dict(ff_phases=2), dict(ff_phases=3)):
                out.append(measure(n, 104, o, reps=3, flags=flags)); print(json.dumps(out[-1]), flush=True)
    if which == "blocks4096":  # workgroup size of the step kernels at the headline batch size (and Tennisbot's)
        from tennisbot_rl_amd.params import ENV_TENNIS
        for o in (dict(block=64), dict(block=128), dict(block=256), dict(block=64), dict(block=128)):
            out.append(measure(4096, 1040, o, flags=flags)); print(json.dumps(out[-1]), flush=True)
        for o in (dict(block=64), dict(block=128), dict(block=256), dict(block=64), dict(block=128)):
            out.append(measure(4096, 1040, o, flags=flags, kind=ENV_TENNIS)); print(json.dumps(out[-1]), flush=True)
    if which == "blocks_ladder":
        from tennisbot_rl_amd.params import ENV_TENNIS
        for n in (1024, 8192, 16384, 32768, 65536):
            for o in (dict(block=64), dict(block=128), dict(block=64), dict(block=128)):
                out.append(measure(n, 1040 if n <= 32768 else 104, o, reps=3, flags=flags)); print(json.dumps(out[-1]), flush=True)
        for n in (8192, 32768):
            for o in (dict(block=64), dict(block=128), dict(block=64), dict(block=128)):
                out.append(measure(n, 1040, o, reps=3, flags=flags, kind=ENV_TENNIS)); print(json.dumps(out[-1]), flush=True)
    if which == "one":
        for o in ({}, {}, {}):
            out.append(measure(4096, 1040, o, flags=flags)); print(json.dumps(out[-1]), flush=True)
    if which == "lanes_few":  # ... fewer still (racket<->court contact with deferred stragglers: every lane is on a path of its own)
        for o in (dict(ff_lanes_per_wave=16), dict(ff_lanes_per_wave=8), dict(ff_lanes_per_wave=4), dict(ff_lanes_per_wave=16), dict(ff_lanes_per_wave=8), dict(ff_lanes_per_wave=4),
                  dict(ff_lanes_per_wave=8, ff_defer_margin=1), dict(ff_lanes_per_wave=8, ff_defer_margin=64)):
            out.append(measure(4096, 1040, o, flags=flags)); print(json.dumps(out[-1]), flush=True)
    if which in ("all", "small"):
        for o in ({}, dict(ff_phases=3), dict(ff_phases=2), {}):
            out.append(measure(4096, 1040, o, flags=flags)); print(json.dumps(out[-1]), flush=True)
    if which == "tennis1m":
        from tennisbot_rl_amd.params import ENV_TENNIS
        for o in (dict(), dict(tennis_reg_rows=False), dict(), dict(tennis_reg_rows=False), dict(block=256), dict(block=256, tennis_reg_rows=False), dict(block=64, tennis_reg_rows=False)):
            out.append(measure(1048576, 104, o, reps=3, flags=flags, kind=ENV_TENNIS)); print(json.dumps(out[-1]), flush=True)
    if which == "opts1m":
        for o in (dict(), dict(swing_reg_rows=False), dict(block=128), dict(block=64), dict(block=256), dict(ff_lanes_per_wave=32), dict()):
            out.append(measure(1048576, 104, o, reps=3, flags=flags)); print(json.dumps(out[-1]), flush=True)
    if which.startswith("n="):  # any batch size, one phase against three
        n, _, T = which[2:].partition(":")  # n=<envs>[:<rollout steps>]
        n = int(n)
        for o in ((dict(),) if T else (dict(ff_phases=1), dict(ff_phases=3), dict(ff_phases=2))):
            out.append(measure(n, int(T) if T else (104 if n > 32768 else 1040), o, reps=3, flags=flags)); print(json.dumps(out[-1]), flush=True)
    if which in ("all", "large"):
        for o in (dict(ff_phases=1), dict(ff_phases=3)):
            out.append(measure(1048576, 104, o, reps=3, flags=flags)); print(json.dumps(out[-1]), flush=True)


if __name__ == "__main__":
    main()

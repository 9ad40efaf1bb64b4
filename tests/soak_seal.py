#!/usr/bin/env python3
"""The sealed-fate exit of the pool (TbOptions.ff_seal) against the oracle's full flights, many seeds (GPU box; tools/soak.sh):
tests/test_gpu_parity.py::test_sealed_fate_exit_books_exactly_what_the_full_flight_gives with other random states.
  python tests/soak_seal.py [seeds=12] [n=32768]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import test_gpu_parity as T

seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
n = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
for k in range(seeds):
    for options in (dict(ff_defer="all"), dict(ff_defer=True, ff_defer_margin=40)):
        T.test_sealed_fate_exit_books_exactly_what_the_full_flight_gives(torch, options, seed=1000 + k, n=n)
print("sealed-fate exit: %d seeds x %d envs x 27 steps x (pool, stragglers), with the exit and without it: every observation, reward, done flag and counter "
      "bit-identical to the f32 oracle's full flights" % (seeds, n))

"""The CPU restatement under AddressSanitizer + UBSan (GPU sanitizers are
not available on the pool, so the sanitised build is the CPU one). A plain C driver
(oracle/selftest.c) runs whole episodes of both envs instrumented."""
import ctypes
import os
import subprocess

import pytest

from tennisbot_rl_amd.params import ENV_SWING, ENV_TENNIS, default_params

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def selftest(tmp_path_factory):
    d = tmp_path_factory.mktemp("asan")
    exe = str(d / "selftest")
    cmd = ["gcc", "-O1", "-g", "-std=c11", "-fopenmp", "-ffp-contract=off", "-mfma", "-fsanitize=address,undefined",
           "-fno-sanitize-recover=undefined", "-o", exe, os.path.join(ROOT, "oracle", "selftest.c"), os.path.join(ROOT, "oracle", "tb_oracle.c"), "-lm"]
    subprocess.check_call(cmd)
    p = default_params(magnus_k=1e-4, ball_spin_max=50.0)
    pbin = str(d / "params.bin")
    open(pbin, "wb").write(ctypes.string_at(ctypes.byref(p), ctypes.sizeof(p)))
    return exe, pbin


@pytest.mark.parametrize("kind,n,steps", [(ENV_SWING, 96, 78), (ENV_TENNIS, 64, 1100)])
def test_oracle_runs_clean_under_asan_ubsan(selftest, kind, n, steps):
    exe, pbin = selftest
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe, pbin, str(kind), str(n), str(steps)], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ERROR" not in r.stderr and "runtime error" not in r.stderr, r.stderr
    assert "episodes" in r.stdout

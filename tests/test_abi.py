"""C-ABI checks that need no GPU: the library loads, exports every symbol the header
declares, its shape queries agree with the Python mirror, TbParams layouts agree, and
the product path refuses to run without a device (no CPU fallback)."""
import ctypes
import os
import re
import subprocess

import pytest

from tennisbot_rl_amd.params import ACT_DIM, ENV_SWING, ENV_TENNIS, OBS_DIM, STATE_WORDS, TbOptions, TbParams, default_params, make_options

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "tb_stepper.h")


@pytest.fixture(scope="module")
def lib():
    from tennisbot_rl_amd.build import build_library
    from tennisbot_rl_amd.stepper import load_library
    build_library()  # hipcc cross-compiles gfx950 without a GPU
    return load_library()


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tb_[a-z_0-9]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(lib):
    names = declared_functions()
    assert {"tb_create", "tb_destroy", "tb_reset", "tb_step", "tb_rollout", "tb_get_state", "tb_set_state",
            "tb_set_params", "tb_counters", "tb_counters_reset", "tb_last_error", "tb_abi_version"} <= set(names)
    for n in names:
        assert hasattr(lib, n), "libtb_stepper.so does not export %s" % n


def test_shape_queries_match_python_mirror(lib):
    for k in (ENV_SWING, ENV_TENNIS):
        assert lib.tb_obs_dim(k) == OBS_DIM[k] and lib.tb_act_dim(k) == ACT_DIM[k] and lib.tb_state_words(k) == STATE_WORDS[k]
    assert lib.tb_obs_dim(7) < 0 and lib.tb_state_words(-1) < 0


@pytest.mark.parametrize("struct", [TbParams, TbOptions])
def test_struct_layouts_match_header(tmp_path, struct):
    """sizeof / offsetof from the C compiler == the ctypes mirrors (TbParams, TbOptions)"""
    name = struct.__name__
    fields = [f[0] for f in struct._fields_]
    prog = ['#include <stdio.h>', '#include <stddef.h>', '#include "%s"' % HEADER, "int main(void){",
            'printf("%%zu\\n", sizeof(%s));' % name]
    prog += ['printf("%%zu\\n", offsetof(%s, %s));' % (name, f) for f in fields]
    prog += ["return 0;}"]
    c = tmp_path / "layout.c"
    c.write_text("\n".join(prog))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-o", str(exe), str(c)])
    out = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert out[0] == ctypes.sizeof(struct)
    for f, off in zip(fields, out[1:]):
        assert getattr(struct, f).offset == off, f


def test_no_device_means_loud_failure_not_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = ctypes.c_void_p()
    p = default_params()
    rc = lib.tb_create(ctypes.byref(p), None, ENV_SWING, 16, 0, 0, 0, ctypes.byref(h))
    assert rc == -2 and not h  # TB_E_NODEVICE
    assert b"no CPU fallback" in lib.tb_last_error()
    from tennisbot_rl_amd.stepper import BatchedEnv, StepperError
    with pytest.raises(StepperError):
        BatchedEnv(ENV_SWING, 16)


def test_bad_arguments_are_rejected(lib):
    h = ctypes.c_void_p()
    p = default_params()
    assert lib.tb_create(ctypes.byref(p), None, 5, 16, 0, 0, 0, ctypes.byref(h)) == -1
    assert lib.tb_create(ctypes.byref(p), None, ENV_SWING, 0, 0, 0, 0, ctypes.byref(h)) == -1
    bad = default_params()
    bad.n_hull = 2
    assert lib.tb_create(ctypes.byref(bad), None, ENV_SWING, 16, 0, 0, 0, ctypes.byref(h)) == -3
    assert b"n_hull" in lib.tb_last_error()
    # kernel-selection options travel through the ABI (since v3), not through environment variables
    assert lib.tb_abi_version() == 4
    o = make_options(block=96)
    assert lib.tb_create(ctypes.byref(p), ctypes.byref(o), ENV_SWING, 16, 0, 0, 0, ctypes.byref(h)) == -1 and b"block" in lib.tb_last_error()
    o = TbOptions()  # struct_size left at 0
    assert lib.tb_create(ctypes.byref(p), ctypes.byref(o), ENV_SWING, 16, 0, 0, 0, ctypes.byref(h)) == -1 and b"struct_size" in lib.tb_last_error()
    src = "".join(open(os.path.join(ROOT, "tennisbot_rl_amd", "csrc", f)).read() for f in os.listdir(os.path.join(ROOT, "tennisbot_rl_amd", "csrc")))
    assert "getenv" not in src
    assert lib.tb_step(None, None, None, None, None, None, None, None) == -1
    assert lib.tb_destroy(None) == 0


def test_product_never_imports_the_oracle():
    """the shipped package must not route through oracle/ (or any CPU fallback)"""
    pkg = os.path.join(ROOT, "tennisbot_rl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "tb_oracle" not in text and "libtb_oracle" not in text, f

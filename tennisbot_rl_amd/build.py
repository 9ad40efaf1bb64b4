"""Build libtb_stepper.so with hipcc for gfx950 (MI355X). In-tree, no JIT cache:
the .so travels to the GPU box with the repo snapshot."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = [os.path.join(_HERE, "csrc", "tb_stepper.hip")]
HEADERS = [os.path.join(_HERE, "csrc", "tb_device.hpp"), os.path.join(_HERE, "csrc", "tb_policy.hpp"), os.path.join(_HERE, "csrc", "tb_diag.hpp"), os.path.join(_HERE, "..", "include", "tb_stepper.h")]
OUTPUT = os.path.join(_HERE, "libtb_stepper.so")

# -ffp-contract=off: the only fused multiply-adds are the explicit __builtin_fmaf calls
# (DESIGN.md "Arithmetic contract"); IEEE divide / sqrt are hipcc's default.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found (needed to build the gfx950 kernels)")
    return exe


def needs_build():
    if not os.path.exists(OUTPUT):
        return True
    t = os.path.getmtime(OUTPUT)
    return any(os.path.getmtime(f) > t for f in SOURCES + HEADERS)


def build_library(force=False, verbose=False, extra_flags=()):
    if not force and not needs_build():
        return OUTPUT
    cmd = [hipcc()] + HIPCC_FLAGS + list(extra_flags) + ["-o", OUTPUT] + SOURCES
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUTPUT


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))

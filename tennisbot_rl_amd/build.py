"""Build libtb_stepper.so with hipcc for gfx950 (MI355X). In-tree, no JIT cache:
the .so travels to the GPU box with the repo snapshot."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = [os.path.join(_HERE, "csrc", "tb_stepper.hip")]
HEADERS = [os.path.join(_HERE, "csrc", "tb_kernels.hpp"), os.path.join(_HERE, "csrc", "tb_device.hpp"), os.path.join(_HERE, "csrc", "tb_policy.hpp"), os.path.join(_HERE, "csrc", "tb_diag.hpp"),
           os.path.join(_HERE, "..", "include", "tb_stepper.h")]
OUTPUT = os.path.join(_HERE, "libtb_stepper.so")

# -ffp-contract=off: the only fused multiply-adds are the explicit __builtin_fmaf calls
# (DESIGN.md "Arithmetic contract"); IEEE divide / sqrt are hipcc's default.
# -fno-slp-vectorize, -amdgpu-sched-strategy=max-ilp: measured, tools/diag/r03_flag_ab.py (profiles/EXPERIMENTS.md, round 3). clang's SLP
# vectoriser packs pairs of fp32 operations (v_pk_mul / v_pk_fma: the same bits) -- fewer instructions, but operand pairs need
# aligned register pairs: +30-40 VGPRs per kernel and a v_mov per operand; without it the large-batch step kernels drop from 119 /
# 118 to 100 / 94 VGPRs and the kernels with loops gain a wave per SIMD. The max-ILP scheduling strategy then orders the unpacked
# code for a lone wave's latency, which is what small batches are bound by. Same box, M env steps/s, packed / packed+ILP /
# unpacked / unpacked+ILP: SwingRacket 4096 envs 861 / 815 / 859 / 924, 32768: 4428 / 4544 / 4270 / 5174, 131072: 5777 / 5862 /
# 7996 / 8239, 1 M: 9779 / 9503 / 10613 / 10797; Tennisbot 4096: 651 / 668 / 655 / 687, 1 M: 19396 / 19581 / 20344 / 20125.
# Neither flag changes a result: every lockstep test runs on this build.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-fPIC", "-shared"]


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

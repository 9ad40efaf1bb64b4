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


# The product library holds the kernels twice (csrc/tb_stepper.hip, TbAltTable): the source as it stands, and once more without
# clang's SLP (packed fp32) vectoriser -- two objects from one source, linked into one shared library. A single-command build
# ([hipcc()] + HIPCC_FLAGS + ["-o", lib] + SOURCES: the diagnostic variants of tools/diag) is the first build alone.
ALT_FLAGS = ["-fno-slp-vectorize", "-DTB_TU_ALT"]


def build_library(force=False, verbose=False, extra_flags=(), output=None):
    out = output or OUTPUT
    if output is None and not force and not needs_build():
        return out
    compile_flags = [f for f in HIPCC_FLAGS if f != "-shared"] + list(extra_flags) + ["-DTB_DUAL_TU", "-c"]
    objs = [out + ".packed.o", out + ".unpacked.o"]
    cmds = [[hipcc()] + compile_flags + ["-o", objs[0]] + SOURCES,
            [hipcc()] + compile_flags + ALT_FLAGS + ["-o", objs[1]] + SOURCES]
    procs = []
    for cmd in cmds:  # (side by side: each takes ~40 s)
        if verbose:
            print(" ".join(cmd))
        procs.append(subprocess.Popen(cmd))
    rcs = [p.wait() for p in procs]
    if any(rcs):
        raise subprocess.CalledProcessError(max(rcs), cmds[rcs.index(max(rcs))])
    link = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link)
    for o in objs:
        os.remove(o)
    return out


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))

#!/usr/bin/env python3
"""Registers, spills, scratch, LDS and occupancy of every kernel of libtb_stepper.so, as the compiler reports them
(-Rpass-analysis=kernel-resource-usage; no GPU needed):   python tools/kernel_resources.py [extra hipcc flags]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tennisbot_rl_amd.build import HIPCC_FLAGS, SOURCES, hipcc  # noqa: E402


def main():
    p = subprocess.run([hipcc()] + HIPCC_FLAGS + sys.argv[1:] + ["-Rpass-analysis=kernel-resource-usage", "-o", "/tmp/tb_resources.so"] + SOURCES,
                       capture_output=True, text=True)
    if p.returncode:
        sys.exit(p.stderr[-3000:])
    blocks = re.split(r"remark: [^\n]*Function Name: ", p.stderr)[1:]
    names = [b.split("\n")[0].split(" [")[0].strip() for b in blocks]
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    for b, dn in zip(blocks, dem):
        def g(k):
            m = re.search(k + r": (\d+)", b)
            return int(m.group(1)) if m else -1
        dn = dn.replace("(anonymous namespace)::", "").replace("void ", "")
        dn = dn[:dn.index("(")] if "(" in dn else dn
        print("%-64s VGPR %3d AGPR %3d SGPR %3d spillV %3d spillS %3d scratch %4d occ %2d LDS %6d" % (
            dn[:64], g("VGPRs"), g("AGPRs"), g("SGPRs"), g("VGPRs Spill"), g("SGPRs Spill"), g(r"ScratchSize \[bytes/lane\]"),
            g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))


if __name__ == "__main__":
    main()

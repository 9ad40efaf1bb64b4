#!/usr/bin/env python3
"""Re-check the product's four scheduling hints after a toolchain change. Each hint is an empty `asm volatile`
that only NAMES values, so that the compiler fetches scalar arguments where a lone wave can hide the wait (tb_kernels.hpp, "SCHEDULING
HINTS"; tb_device.hpp racket_planes). They were chosen on AMD clang 22 / ROCm 7.2; a compiler bump can undo or invert any of them.
For every hint: the library built with -DTB_HINT_<x>=0 (A) against the default build (B), same box, each workload a replayed rollout
graph in a process of its own (tools/diag/r03_flag_ab.py does the builds and the timing). A hint whose B is not ahead of its A by
more than the box's run-to-run spread (~1 %) should be deleted.   Run on the GPU box:   python tools/diag/r04_hint_recheck.py"""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
HINTS = [  # (macro, workloads that the hint was chosen on)
    ("TB_HINT_TENNIS_CONSTANTS", ["tennis4096", "tennis1m"]),
    ("TB_HINT_TENNIS_OUTPUTS", ["tennis4096"]),
    ("TB_HINT_POLICY_VGPR_PARAMS", ["collect_untrained", "collect_ref"]),
    ("TB_HINT_RELOAD_PLANES", ["swing1m", "swing128k"]),
]
only = set(sys.argv[1:])
for macro, work in HINTS:
    if only and macro not in only:
        continue
    print("==", macro, "(A: hint off, B: product)", flush=True)
    r = subprocess.run([sys.executable, os.path.join(HERE, "r03_flag_ab.py"), "-D%s=0" % macro, ""] + work + work[:1], capture_output=True, text=True)
    sys.stdout.write(r.stdout)
    if r.returncode:
        sys.stdout.write(r.stderr[-1500:])
        sys.exit(1)
    sys.stdout.flush()

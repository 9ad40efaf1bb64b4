#!/usr/bin/env python3
"""Per-kernel averages of the SQ counter passes of tools/run_pmc_sq.sh (per launch, by kernel and grid size)."""
import collections
import csv
import glob
import os
import sys

d = sys.argv[1]
acc = collections.defaultdict(list)
for path in sorted(glob.glob(os.path.join(d, "sq_*_counter_collection.csv"))):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        name = "ff<%s>" % k.split("tb_ff_kernel<")[1].split(">")[0] if "tb_ff_kernel" in k else "step" if "tb_step_kernel" in k else None
        if name:
            acc[(name, int(r["Grid_Size"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
for (name, grid, c), v in sorted(acc.items()):
    print("%-18s grid %9d  %-22s launches %4d  mean %16.1f" % (name, grid, c, len(v), sum(v) / len(v)))

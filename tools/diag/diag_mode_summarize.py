#!/usr/bin/env python3
"""usage: diag_mode_summarize.py <kernel_trace.csv>: step-kernel durations, gaps between consecutive step kernels, fast-forward durations"""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    kind = "ff" if "tb_ff_kernel" in k else "step" if "tb_step_kernel" in k else None
    if kind:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind, r.get("Queue_Id", "?")))
rows.sort()
steps = [r for r in rows if r[2] == "step"][-1040 * 5:]
ffs = [r for r in rows if r[2] == "ff"][-40 * 5:]
d = sorted(e - s for s, e, _, _ in steps)
gaps = sorted(steps[i + 1][0] - steps[i][1] for i in range(len(steps) - 1))
fd = sorted(e - s for s, e, _, _ in ffs)
q = lambda v, p: v[int(p * (len(v) - 1))] / 1e3
print("step kernels: %d, duration p10/p50/p90 %.2f / %.2f / %.2f us; gap to the next p10/p50/p90 %.2f / %.2f / %.2f us; mean period %.2f us" % (
    len(steps), q(d, .1), q(d, .5), q(d, .9), q(gaps, .1), q(gaps, .5), q(gaps, .9), (steps[-1][0] - steps[0][0]) / (len(steps) - 1) / 1e3))
print("fast-forward kernels: %d, duration p10/p50/p90 %.0f / %.0f / %.0f us" % (len(ffs), q(fd, .1), q(fd, .5), q(fd, .9)))
print("queues: step kernels on %s, fast-forwards on %s" % (sorted(set(r[3] for r in steps)), sorted(set(r[3] for r in ffs))))
# the last replay: which queue each step kernel ran on (run-length compressed), and where the fast-forwards sat
last = steps[-1040:]
seq, prev, run = [], None, 0
for r in last:
    if r[3] == prev: run += 1
    else:
        if prev is not None: seq.append("%sx%d" % (prev, run))
        prev, run = r[3], 1
seq.append("%sx%d" % (prev, run))
print("step-kernel queue sequence of the last replay:", " ".join(seq[:80]), "..." if len(seq) > 80 else "")
lo, hi = last[0][0], last[-1][1]
print("fast-forward queues in that replay:", " ".join(r[3] for r in ffs if lo <= r[0] <= hi))

#!/usr/bin/env python3
"""Timeline of a kernel trace (rocprofv3 --kernel-trace CSV): per step kernel its duration and how many fast-forward kernels ran
beside it; per fast-forward kernel its span. usage: r02_timeline.py <kernel_trace.csv>"""
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    kind = "ff" if ("tb_ff_kernel" in k or "tb_ff_flight_kernel" in k) else "step" if "tb_step_kernel" in k else None
    if kind:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind, int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0), ("flight" + k.split("tb_ff_flight_kernel")[-1][:8] if "tb_ff_flight_kernel" in k else k.split("tb_ff_kernel")[-1][:20]) if kind == "ff" else ""))
rows.sort()
t0 = rows[0][0]
ffs = [r for r in rows if r[2] == "ff"]
steps = [r for r in rows if r[2] == "step"]
# the last full rollout: take the last 104 step kernels
steps = steps[-104:]
lo = steps[0][0]
print("window: %.3f ms, %d step kernels" % ((steps[-1][1] - lo) / 1e6, len(steps)))
prev_end = None
for i, (s, e, _, g, _) in enumerate(steps):
    beside = [f for f in ffs if f[0] < e and f[1] > s]
    ov = sum(min(e, f[1]) - max(s, f[0]) for f in beside) / max(e - s, 1)
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    prev_end = e
    print("step %3d  start %9.1f us  dur %7.1f us  gap before %6.1f us  ff beside: %d (coverage %.2f) %s" % (
        i, (s - lo) / 1e3, (e - s) / 1e3, gap, len(beside), ov, " ".join("%s:%d" % (f[4].split("(")[0], f[3]) for f in beside)))
print("fast-forward kernels in the window:")
for f in ffs:
    if f[1] > lo and f[0] < steps[-1][1]:
        print("   %s grid %8d  start %9.1f us  dur %8.1f us" % (f[4], f[3], (f[0] - lo) / 1e3, (f[1] - f[0]) / 1e3))

#!/usr/bin/env python3
"""Histogram of per-launch durations of one kernel from a rocprofv3 --kernel-trace CSV:
python tools/trace_hist.py <kernel_trace.csv> <name substring>"""
import csv
import sys

import numpy as np


def main(path, needle):
    d = []
    for r in csv.DictReader(open(path)):
        if needle in r["Kernel_Name"]:
            d.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    d.sort()
    dur = np.array([x[1] for x in d], dtype=np.float64) / 1e3
    gaps = np.array([d[k + 1][0] - (d[k][0] + d[k][1]) for k in range(len(d) - 1)], dtype=np.float64) / 1e3
    print("%d launches of *%s*: duration us mean %.2f  p10 %.2f  p50 %.2f  p90 %.2f  p99 %.2f  max %.2f" % (
        len(dur), needle, dur.mean(), *np.percentile(dur, [10, 50, 90, 99]), dur.max()))
    print("gap to the next launch us: mean %.2f  p10 %.2f  p50 %.2f  p90 %.2f" % (gaps.mean(), *np.percentile(gaps, [10, 50, 90])))
    hist, edges = np.histogram(dur, bins=[0, 3, 4, 5, 6, 7, 8, 9, 10, 12, 15, 20, 1e9])
    print("histogram (us): " + "  ".join("<%g: %d" % (edges[k + 1], hist[k]) for k in range(len(hist) - 1)) + "  more: %d" % hist[-1])
    # Swing: position of the launch within the 26-step episode
    if len(dur) % 26 == 0:
        ph = dur.reshape(-1, 26).mean(0)
        print("mean duration by launch index mod 26: " + " ".join("%.1f" % x for x in ph))
    # the last 1040 launches = bench.py's timed replay: start-to-start intervals by position in the 26-step episode
    # (the launch at index 25 ends the episode and forks the fast-forward)
    if len(d) >= 1041:
        st = np.array([x[0] for x in d[-1041:]], dtype=np.float64) / 1e3
        iv = np.diff(st)
        print("timed replay: start-to-start us mean %.2f p10 %.2f p50 %.2f p90 %.2f; sum %.0f us" % (iv.mean(), *np.percentile(iv, [10, 50, 90]), iv.sum()))
        print("mean start-to-start by launch index mod 26 (interval FOLLOWING launch k): " + " ".join("%.1f" % x for x in iv.reshape(-1, 26).mean(0)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])

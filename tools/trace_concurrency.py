#!/usr/bin/env python3
"""How many fast-forward kernels run at once, and on which hardware queues? From a rocprofv3 --kernel-trace CSV:
python tools/trace_concurrency.py <kernel_trace.csv>"""
import csv
import sys
from collections import Counter

import numpy as np


def main(path):
    ff, st = [], []
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        rec = (int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r.get("Stream_Id", "?"))
        if "tb_ff_kernel" in k:
            ff.append(rec)
        elif "tb_step_kernel" in k or "tb_policy_rollout_kernel" in k:
            st.append(rec)
    ff.sort(); st.sort()
    if not ff:
        print("no fast-forward kernels in the trace")
        return
    d = np.array([e - s for s, e, _, _ in ff], np.float64) / 1e3
    print("%d fast-forward kernels: duration us mean %.0f  p10 %.0f  p50 %.0f  p90 %.0f  max %.0f" % (len(d), d.mean(), *np.percentile(d, [10, 50, 90]), d.max()))
    print("queues of the fast-forward kernels:", dict(Counter(q for _, _, q, _ in ff)), " streams:", len(set(s for _, _, _, s in ff)))
    print("queues of the step kernels:        ", dict(Counter(q for _, _, q, _ in st)), " streams:", len(set(s for _, _, _, s in st)))
    # time-weighted concurrency over the span from the first fast-forward's start to the last one's end
    ev = sorted([(s, 1) for s, _, _, _ in ff] + [(e, -1) for _, e, _, _ in ff])
    t_prev, level, acc = ev[0][0], 0, Counter()
    for t, dl in ev:
        acc[level] += t - t_prev
        t_prev, level = t, level + dl
    tot = sum(acc.values())
    print("fast-forwards in flight (share of the time): " + "  ".join("%d: %.1f %%" % (k, 100.0 * v / tot) for k, v in sorted(acc.items())))
    print("mean in flight %.2f" % (sum(k * v for k, v in acc.items()) / tot))
    # does a fast-forward start when its step kernel ends, or later (queued behind another kernel on its hardware queue)?
    ends = np.array([e for _, e, _, _ in st], np.int64)
    lag = []
    for s, _, _, _ in ff:
        j = np.searchsorted(ends, s, side="right") - 1
        if j >= 0:
            lag.append((s - ends[j]) / 1e3)
    lag = np.array(lag)
    print("start of a fast-forward after the end of the latest step kernel before it, us: p10 %.1f p50 %.1f p90 %.1f max %.1f" % (*np.percentile(lag, [10, 50, 90]), lag.max()))
    byq = {}
    for s, e, q, _ in ff:
        byq.setdefault(q, []).append((s, e))
    for q, v in byq.items():
        v.sort()
        ov = sum(1 for (s0, e0), (s1, e1) in zip(v, v[1:]) if s1 < e0)
        print("  queue %s: %d fast-forwards, %d of them started before the previous one on the same queue had ended" % (q, len(v), ov))


if __name__ == "__main__":
    main(sys.argv[1])

#!/usr/bin/env python3
"""Turn the rocprofv3 --pmc CSVs of tools/run_pmc.sh into the per-launch HBM traffic figure
that bench.py reports as roofline.traffic.

Corrections follow MI355X_MICROARCH.md (HBM section) and are CHECKED against this kernel's
own access pattern by the calibration launches (tools/pmc_calibrate.py): FETCH_SIZE and
WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of a coalesced stream
(measured here for one-dword-per-lane rows: ratio printed as `fetch_calibration`), WRITE_SIZE
reads exact."""
import collections
import csv
import json
import os
import sys


def per_kernel(path, counter, needle):
    vals = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and needle in r["Kernel_Name"]:
            vals[int(r["Grid_Size"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in vals.items()}, {k: len(v) for k, v in vals.items()}


def main(d, out):
    cal_f, _ = per_kernel(os.path.join(d, "calib_FETCH_SIZE_counter_collection.csv"), "FETCH_SIZE", "tb_diag_copy_kernel")
    cal_w, _ = per_kernel(os.path.join(d, "calib_WRITE_SIZE_counter_collection.csv"), "WRITE_SIZE", "tb_diag_copy_kernel")
    rows = 30
    big = max(cal_f)
    truth_kib = rows * big * 4 / 1024.0
    fetch_ratio = cal_f[big] / truth_kib
    write_ratio = cal_w[big] / truth_kib
    res = {"source": os.path.basename(d.rstrip("/")), "fetch_calibration": fetch_ratio, "write_calibration": write_ratio,
           "fetch_correction": 1.0 / fetch_ratio, "workloads": {}}
    for tag in ("swing4096", "swing1m", "tennis4096", "tennis1m"):
        env = "swing" if tag.startswith("swing") else "tennis"
        if not os.path.exists(os.path.join(d, "%s_FETCH_SIZE_counter_collection.csv" % tag)):
            continue
        f, nf = per_kernel(os.path.join(d, "%s_FETCH_SIZE_counter_collection.csv" % tag), "FETCH_SIZE", "tb_step_kernel")
        w, _ = per_kernel(os.path.join(d, "%s_WRITE_SIZE_counter_collection.csv" % tag), "WRITE_SIZE", "tb_step_kernel")
        for n in f:
            fb = f[n] * 1024.0 / fetch_ratio
            wb = w[n] * 1024.0 / write_ratio
            res["workloads"]["%s_%d" % (env, n)] = {
                "launches": nf[n], "FETCH_SIZE_KiB_raw": f[n], "WRITE_SIZE_KiB_raw": w[n],
                "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "traffic_bytes_per_launch": fb + wb,
                "traffic_bytes_per_env_step": (fb + wb) / n, "algorithmic_bytes_per_env_step": 267 if env == "swing" else 263}
        # pipelined runs: the fast-forward is its own kernel chain (per 26 agent steps: one launch over all parked envs and, from
        # 256 K envs on, two launches over the compacted survivors of the budgeted phases, on a smaller fixed grid)
        ff, nff = per_kernel(os.path.join(d, "%s_FETCH_SIZE_counter_collection.csv" % tag), "FETCH_SIZE", "tb_ff_kernel")
        fw, _ = per_kernel(os.path.join(d, "%s_WRITE_SIZE_counter_collection.csv" % tag), "WRITE_SIZE", "tb_ff_kernel")
        for n in f:
            if n not in ff and ff and env == "swing" and len(f) == 1:
                # the pool form (up to 16384 envs): ONE fast-forward launch per rollout over every episode end parked since the last
                # join, on a grid of its own -- all of its bytes over the episode ends the step launches of the same pass produced
                episodes = nf[n] / 26.0
                fb = sum(ff[g] * nff[g] for g in ff) * 1024.0 / fetch_ratio / episodes
                wb = sum(fw[g] * nff[g] for g in fw) * 1024.0 / write_ratio / episodes
                res["workloads"]["%s_%d" % (env, n)]["ff_kernel"] = {
                    "launches": {str(g): nff[g] for g in nff}, "form": "pool: one launch per rollout", "fetch_bytes_per_episode_end": fb, "write_bytes_per_episode_end": wb,
                    "traffic_bytes_per_env_per_episode_end": (fb + wb) / n, "algorithmic_bytes_per_env_per_episode_end": 140,
                    "note": "per episode end: reads each parked env's 128-byte record and its 8-byte destination pointer, writes its reward (4 B); the parking step wrote both"}
                continue
            if n not in ff:
                continue
            fb = sum(ff[g] * nff[g] for g in ff) / nff[n] * 1024.0 / fetch_ratio   # all phase kernels, per episode end
            wb = sum(fw[g] * nff[g] for g in fw) / nff[n] * 1024.0 / write_ratio
            res["workloads"]["%s_%d" % (env, n)]["ff_kernel"] = {
                "launches": {str(g): nff[g] for g in nff}, "fetch_bytes_per_episode_end": fb, "write_bytes_per_episode_end": wb,
                "traffic_bytes_per_env_per_episode_end": (fb + wb) / n, "algorithmic_bytes_per_env_per_episode_end": 136,
                "excess_bytes_per_env_per_episode_end": (fb + wb) / n - 136.0,
                "note": "per episode end (26 tb_step_kernel launches): reads each parked env's 128-byte record (the state; 192 B with the racket<->court cache in the RG instantiations, and in every kernel until round 3), writes its reward and clears its parked flag (4 + 4 B). The excess at 4096 envs (one kernel, no hand-overs) is table staging, counters and partial lines; with phases an env handed from one phase kernel to the next (it outlived its budget, or -- first phase of the large-batch chain -- its ball reached the racket) is written and read once more, 128 B each way: at 1 M envs roughly 0.3 hand-overs per env"}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])

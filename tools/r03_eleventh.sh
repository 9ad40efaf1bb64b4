#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03m
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o rg -- python3 $R/bench.py --racket-ground --no-cpu-baseline --no-sweep --settle-seconds 0 --min-timed-ms 0 --steps 1040 > $OUT/rg.log 2>&1 || exit 1
rm -f $OUT/rg_kernel_trace.csv
python3 - <<PY
import csv
for r in csv.DictReader(open("$OUT/rg_kernel_stats.csv")):
    if "tb_" in r["Name"]: print(r["Name"][:90], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["Percentage"])
PY
cd $R
for k in 1 2 3; do
  python3 ab/r02/bench.py --no-cpu-baseline --no-sweep > $OUT/ab_r02_$k.json 2>> $OUT/ab.err || exit 1
  python3 bench.py --no-cpu-baseline --no-sweep > $OUT/ab_new_$k.json 2>> $OUT/ab.err || exit 1
done
python3 - <<'PY'
import json, glob, os
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out", "r03m")
for tag in ("r02", "new"):
    print(tag, [round(json.load(open(f))["value"] / 1e6, 1) for f in sorted(glob.glob(os.path.join(out, "ab_%s_*.json" % tag)))])
PY
timeout -k 10 300 python3 tools/diag/r03_priority_probe.py > $OUT/priority.log 2>&1; grep -v amdgpu.ids $OUT/priority.log

#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03l
mkdir -p $OUT
export TMPDIR=/tmp
cd $R || exit 1
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1 || { tail -40 $OUT/pytest.log; exit 1; }
tail -2 $OUT/pytest.log
for k in 1 2 3; do
  python3 ab/r02/bench.py --no-cpu-baseline --no-sweep > $OUT/ab_r02_$k.json 2>> $OUT/ab.err || exit 1
  python3 bench.py --no-cpu-baseline --no-sweep > $OUT/ab_new_$k.json 2>> $OUT/ab.err || exit 1
done
python3 bench.py --no-cpu-baseline --no-sweep --contact-off > $OUT/new_off.json 2>> $OUT/ab.err || exit 1
python3 bench.py --no-cpu-baseline --no-sweep --env tennis > $OUT/new_tennis.json 2>> $OUT/ab.err || exit 1
python3 bench.py --no-cpu-baseline --no-sweep --racket-ground --settle-seconds 0.3 --min-timed-ms 0 --steps 2080 > $OUT/new_rg.json 2>> $OUT/ab.err || exit 1
python3 - <<'PY'
import json, glob, os
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out", "r03l")
for tag in ("r02", "new"):
    print(tag, [round(json.load(open(f))["value"] / 1e6, 1) for f in sorted(glob.glob(os.path.join(out, "ab_%s_*.json" % tag)))])
for k in ("off", "tennis", "rg"):
    d = json.load(open(os.path.join(out, "new_%s.json" % k))); print(k, round((d["value"] or d["value_refused"]) / 1e6, 1), "launch_us", round(d["roofline"]["launch_us"], 2))
PY
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o swing -- python3 $R/bench.py --no-cpu-baseline --no-sweep --settle-seconds 0 --min-timed-ms 0 --steps 1040 > $OUT/trace.log 2>&1 || exit 1
python3 $R/tools/trace_concurrency.py $OUT/prof/swing_kernel_trace.csv; rm -f $OUT/prof/swing_kernel_trace.csv
python3 $R/tools/diag/r03_cadence_probe.py > $OUT/cadence.log 2>&1; grep -v amdgpu $OUT/cadence.log | cut -c1-400

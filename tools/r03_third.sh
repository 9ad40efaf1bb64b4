#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03c
mkdir -p $OUT
cd $R || exit 1
python3 -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1 || { tail -40 $OUT/pytest.log; exit 1; }
tail -2 $OUT/pytest.log
python3 tools/diag/r03_collect_breakdown.py > $OUT/collect_breakdown.log 2>&1 || { tail -20 $OUT/collect_breakdown.log; exit 1; }
cat $OUT/collect_breakdown.log
python3 tools/diag/r03_busy_probe.py > $OUT/busy_probe.log 2>&1 || { tail -20 $OUT/busy_probe.log; exit 1; }
cat $OUT/busy_probe.log
for k in 1 2 3; do
  python3 ab/r02/bench.py --no-cpu-baseline --no-sweep > $OUT/ab_r02_$k.json 2>> $OUT/ab.err || exit 1
  python3 bench.py --no-cpu-baseline --no-sweep > $OUT/ab_new_$k.json 2>> $OUT/ab.err || exit 1
done
python3 - <<'PY'
import json, glob, os
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out", "r03c")
for tag in ("r02", "new"):
    print(tag, [round(json.load(open(f))["value"] / 1e6, 1) for f in sorted(glob.glob(os.path.join(out, "ab_%s_*.json" % tag)))])
PY
python3 tools/pin_sensitivity.py > $OUT/pin_sensitivity.log 2>&1 || { tail -30 $OUT/pin_sensitivity.log; exit 1; }
tail -12 $OUT/pin_sensitivity.log

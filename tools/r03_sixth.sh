#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03f
mkdir -p $OUT
cd $R || exit 1
python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q > $OUT/pytest.log 2>&1 || { tail -40 $OUT/pytest.log; exit 1; }
tail -2 $OUT/pytest.log
python3 tools/diag/r03_idle_probe.py > $OUT/idle_probe.log 2>&1 || { tail -20 $OUT/idle_probe.log; exit 1; }
cat $OUT/idle_probe.log
for q in 4 8 16; do
  GPU_MAX_HW_QUEUES=$q python3 bench.py --no-cpu-baseline --no-sweep > $OUT/q${q}_swing.json 2>> $OUT/q.err || exit 1
  GPU_MAX_HW_QUEUES=$q python3 bench.py --no-cpu-baseline --no-sweep --racket-ground --settle-seconds 0 --min-timed-ms 0 --steps 1040 > $OUT/q${q}_swing_rg.json 2>> $OUT/q.err || exit 1
  GPU_MAX_HW_QUEUES=$q python3 tools/diag/r03_collect_breakdown.py > $OUT/q${q}_collect.log 2>&1 || exit 1
done
python3 - <<'PY'
import json, glob, os
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out", "r03f")
for q in (4, 8, 16):
    a = json.load(open(os.path.join(out, "q%d_swing.json" % q))); b = json.load(open(os.path.join(out, "q%d_swing_rg.json" % q)))
    print("GPU_MAX_HW_QUEUES", q, "swing %.1f M" % (a["value"] / 1e6), "racket-ground %.1f M" % ((b["value"] or b.get("value_refused")) / 1e6))
    print(open(os.path.join(out, "q%d_collect.log" % q)).read())
PY

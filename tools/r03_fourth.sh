#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03d
mkdir -p $OUT
cd $R || exit 1
python3 tools/diag/r03_cadence_probe.py > $OUT/cadence_probe.log 2>&1 || { tail -20 $OUT/cadence_probe.log; exit 1; }
cat $OUT/cadence_probe.log
python3 tools/pin_sensitivity.py > $OUT/pin_sensitivity.log 2>&1 || { tail -30 $OUT/pin_sensitivity.log; exit 1; }
grep "^critic\|^leave" $OUT/pin_sensitivity.log

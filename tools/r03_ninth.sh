#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03k
mkdir -p $OUT
export TMPDIR=/tmp
cd $R || exit 1
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1 || { tail -40 $OUT/pytest.log; exit 1; }
tail -2 $OUT/pytest.log
python3 bench.py --no-cpu-baseline --no-sweep --racket-ground --settle-seconds 0.3 --min-timed-ms 0 --steps 2080 > $OUT/rg.json 2>> $OUT/err.log || exit 1
python3 -c "import json; d=json.load(open('$OUT/rg.json')); print('racket-ground, deferred stragglers:', (d['value'] or d['value_refused'])/1e6, 'M env steps/s', d.get('invalid'))"
python3 tools/diag/r03_collect_breakdown.py > $OUT/collect.log 2>&1 || { tail $OUT/collect.log; exit 1; }
grep -v amdgpu.ids $OUT/collect.log
python3 bench.py --no-cpu-baseline --no-sweep > $OUT/swing.json 2>> $OUT/err.log || exit 1
python3 -c "import json; d=json.load(open('$OUT/swing.json')); print('headline:', d['value']/1e6)"

#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03e
mkdir -p $OUT
export TMPDIR=/tmp
cd $R || exit 1
python3 -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1 || { tail -40 $OUT/pytest.log; exit 1; }
tail -2 $OUT/pytest.log
python3 tools/diag/r03_cadence_probe.py > $OUT/cadence_probe.log 2>&1 || { tail -20 $OUT/cadence_probe.log; exit 1; }
cat $OUT/cadence_probe.log
cd /tmp
for v in "--racket-ground" ""; do
  tag=trace_swing4096${v:+_racket_ground}
  rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o $tag -- python3 $R/bench.py $v --no-cpu-baseline --no-sweep --settle-seconds 0 --min-timed-ms 0 --steps 1040 > $OUT/$tag.log 2>&1 || exit 1
  python3 $R/tools/trace_concurrency.py $OUT/prof/${tag}_kernel_trace.csv > $OUT/$tag.concurrency.txt 2>&1
  rm -f $OUT/prof/${tag}_kernel_trace.csv
  echo "== $tag"; cat $OUT/$tag.concurrency.txt
done

#!/bin/bash
# round-3 first GPU pass: the GPU suite, the driver's bench line, and a same-box kernel trace of the headline against
# BASELINE configs[1] as worded (racket<->ball contact off)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03a
mkdir -p $OUT
export TMPDIR=/tmp
cd $R || exit 1
python3 -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log; exit 1; }
tail -3 $OUT/pytest.log
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_driver_line.json 2> $OUT/bench_driver.err || exit 1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o swing4096 -- python3 $R/bench.py --no-cpu-baseline --no-sweep > $OUT/prof_swing4096.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o swing4096_contact_off -- python3 $R/bench.py --contact-off --no-cpu-baseline --no-sweep > $OUT/prof_swing4096_contact_off.log 2>&1 || exit 1
rm -f $OUT/prof/*kernel_trace.csv
ls -la $OUT $OUT/prof

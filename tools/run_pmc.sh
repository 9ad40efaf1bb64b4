#!/bin/bash
# PMC passes for the roofline `traffic` figure. One counter per pass (FETCH_SIZE and
# WRITE_SIZE do not fit one pass on gfx950), never combined with tracing options.
# usage (on the GPU box): bash tools/run_pmc.sh <tag>
set -o pipefail
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
B="--rollout-steps 104 --steps 104 --warmup 26 --no-cpu-baseline --no-sweep"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT -o calib_$C -- python3 $R/tools/pmc_calibrate.py > $OUT/calib_$C.log 2>&1 || exit 1
  rocprofv3 --pmc $C --output-format csv -d $OUT -o swing4096_$C -- python3 $R/bench.py $B > $OUT/swing4096_$C.log 2>&1 || exit 1
  rocprofv3 --pmc $C --output-format csv -d $OUT -o swing1m_$C -- python3 $R/bench.py --envs-per-gpu 1048576 $B > $OUT/swing1m_$C.log 2>&1 || exit 1
  rocprofv3 --pmc $C --output-format csv -d $OUT -o tennis4096_$C -- python3 $R/bench.py --env tennis $B > $OUT/tennis4096_$C.log 2>&1 || exit 1
  rocprofv3 --pmc $C --output-format csv -d $OUT -o tennis1m_$C -- python3 $R/bench.py --env tennis --envs-per-gpu 1048576 $B > $OUT/tennis1m_$C.log 2>&1 || exit 1
done
ls -la $OUT

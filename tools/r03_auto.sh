#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03w; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
B="--envs-per-gpu 1048576 --rollout-steps 104 --steps 104 --warmup 26 --no-cpu-baseline --no-sweep"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o swing1m -- python3 $R/bench.py $B > $O/prof_swing1m.log 2>&1 || exit 1
rm -f $O/prof/*kernel_trace.csv
grep -h "tb_" $O/prof/*kernel_stats.csv | cut -c28-75,150-330
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $O/pmc_sq -o sq_a -- python3 $R/bench.py $B > $O/sq_a.log 2>&1 || exit 1
python3 $R/tools/summarize_pmc_sq.py $O/pmc_sq > $O/sq_counters_1m.txt 2>&1; cat $O/sq_counters_1m.txt
cd $R && timeout -k 10 300 python3 tools/diag/r02_ff_ab.py lazy1m 2>&1 | grep -v amdgpu.ids | cut -c1-220 | head -4

#!/bin/bash
# SQ-side PMC passes for the large-batch SwingRacket kernels: is the fast-forward VALU-issue bound or latency bound?
# usage (on the GPU box): bash tools/run_pmc_sq.sh <tag>   (never combined with tracing options; one small counter set per pass)
set -o pipefail
TAG=${1:-r02sq}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
B="--envs-per-gpu 1048576 --rollout-steps 104 --steps 104 --warmup 26 --no-cpu-baseline --no-sweep"
rocprofv3 -L > $OUT/counters.txt 2>&1
for C in "VALUBusy" "VALUUtilization" "SALUBusy" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "LDSBankConflict" "MemUnitStalled"; do
  N=$(echo $C | tr ' ' '_')
  rocprofv3 --pmc $C --output-format csv -d $OUT -o sq_$N -- python3 $R/bench.py $B > $OUT/sq_$N.log 2>&1 || echo "pass $N failed" >> $OUT/failed.txt
  echo "pass $N done"
done
ls $OUT

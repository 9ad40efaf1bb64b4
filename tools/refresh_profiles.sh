#!/bin/bash
# One GPU-box pass that regenerates every figure DESIGN.md section 5 quotes:
#   bash tools/refresh_profiles.sh <tag>      (writes under gpurun_out/<tag>/, copy what is judged into profiles/)
# Steps are joined so that a failing GPU step stops the pass.
set -o pipefail
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R || exit 1
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_swing4096_driver_line.json 2> $OUT/bench_driver.err || exit 1
python3 bench.py --sweep > $OUT/bench_swing4096_sweep.json 2> $OUT/bench_swing.err || exit 1
python3 bench.py --env tennis --sweep --no-cpu-baseline > $OUT/bench_tennis4096_sweep.json 2> $OUT/bench_tennis.err || exit 1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o swing4096 -- python3 $R/bench.py --no-cpu-baseline --no-sweep > $OUT/prof_swing4096.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o swing1m -- python3 $R/bench.py --envs-per-gpu 1048576 --rollout-steps 104 --steps 104 --no-cpu-baseline --no-sweep > $OUT/prof_swing1m.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o tennis4096 -- python3 $R/bench.py --env tennis --no-cpu-baseline --no-sweep > $OUT/prof_tennis4096.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o ppo_fused -- python3 $R/train_swing.py --total-timesteps 4e6 --save /tmp/ppo_%s.pt > $OUT/prof_ppo_fused.log 2>&1 || exit 1
rm -f $OUT/prof/*kernel_trace.csv
bash $R/tools/run_pmc.sh $TAG > $OUT/pmc.log 2>&1 || exit 1
python3 $R/tools/summarize_pmc.py $R/gpurun_out/pmc_$TAG $OUT/pmc_traffic.json > $OUT/pmc_summary.log 2>&1 || exit 1
ls -la $OUT $OUT/prof

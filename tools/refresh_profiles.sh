#!/bin/bash
# GPU-box passes that regenerate every figure DESIGN.md section 5 quotes (each part fits one gpurun call):
#   bash tools/refresh_profiles.sh <tag> a     bench lines, sweeps, rocprofv3 kernel stats of the same commands
#   bash tools/refresh_profiles.sh <tag> b     PMC traffic passes, SQ counters at 1 M envs
#   bash tools/refresh_profiles.sh <tag> c     PPO probes (collect rate, learning curve), pin analysis, racket<->court rates
# (writes under gpurun_out/<tag>/; copy what is judged into profiles/). EVERY GPU step ends in `|| exit 1`: a step that fails or
# faults stops the pass -- it is looked into from its log, never run past.
set -o pipefail
TAG=${1:-r04}
PART=${2:-a}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R || exit 1
if [ "$PART" = "a" ]; then
  python3 bench.py --steps 20 --warmup 5 > $OUT/bench_swing4096_driver_line.json 2> $OUT/bench_driver.err || exit 1
  python3 bench.py --sweep > $OUT/bench_swing4096_sweep.json 2> $OUT/bench_swing.err || exit 1
  python3 bench.py --env tennis --sweep --no-cpu-baseline > $OUT/bench_tennis4096_sweep.json 2> $OUT/bench_tennis.err || exit 1
  TB_BENCH_REHEARSAL=1 python3 bench.py --gpus 2 --steps 20 --warmup 5 > $OUT/bench_two_rank_rehearsal_one_gpu.json 2> $OUT/bench_two_rank.err || exit 1
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o swing4096 -- python3 $R/bench.py --no-cpu-baseline --no-sweep > $OUT/prof_swing4096.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o swing4096_contact_off -- python3 $R/bench.py --contact-off --no-cpu-baseline --no-sweep > $OUT/prof_swing4096_contact_off.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o swing4096_racket_ground -- python3 $R/bench.py --racket-ground --no-cpu-baseline --no-sweep --settle-seconds 0.3 --min-timed-ms 0 --steps 2080 > $OUT/prof_swing4096_racket_ground.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o swing1m -- python3 $R/bench.py --envs-per-gpu 1048576 --rollout-steps 104 --steps 104 --no-cpu-baseline --no-sweep > $OUT/prof_swing1m.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o tennis4096 -- python3 $R/bench.py --env tennis --no-cpu-baseline --no-sweep > $OUT/prof_tennis4096.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o ppo_fused -- python3 $R/train_swing.py --total-timesteps 4e6 --n-steps 1040 --save /tmp/ppo_%s.pt > $OUT/prof_ppo_fused.log 2>&1 || exit 1
  rm -f $OUT/prof/*kernel_trace.csv
  ls -la $OUT $OUT/prof
elif [ "$PART" = "b" ]; then
  bash $R/tools/run_pmc.sh $TAG > $OUT/pmc.log 2>&1 || exit 1
  python3 $R/tools/summarize_pmc.py $R/gpurun_out/pmc_$TAG $OUT/pmc_traffic.json > $OUT/pmc_summary.log 2>&1 || exit 1
  cat $OUT/pmc_summary.log
  cd /tmp
  B="--envs-per-gpu 1048576 --rollout-steps 104 --steps 104 --warmup 26 --no-cpu-baseline --no-sweep"
  for C in "VALUBusy" "VALUUtilization" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
    N=$(echo $C | tr ' ' '_')
    rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_sq -o sq_$N -- python3 $R/bench.py $B > $OUT/sq_$N.log 2>&1 || { echo "pass $N failed"; tail -20 $OUT/sq_$N.log; exit 1; }
  done
  python3 $R/tools/summarize_pmc_sq.py $OUT/pmc_sq > $OUT/sq_counters_1m.txt 2>&1 || exit 1
  cat $OUT/sq_counters_1m.txt
else
  python3 tools/diag/r03_ppo_probe.py curve > $OUT/ppo_probe.log 2>&1 || { tail -20 $OUT/ppo_probe.log; exit 1; }
  cp $R/gpurun_out/r03_ppo_probe.json $OUT/ppo_probe.json
  python3 tools/diag/r03_collect_breakdown.py > $OUT/collect_breakdown.log 2>&1 || exit 1
  cp $R/gpurun_out/r03_collect_breakdown.json $OUT/collect_breakdown.json
  python3 tools/pin_sensitivity.py > $OUT/pin_sensitivity.log 2>&1 || { tail -20 $OUT/pin_sensitivity.log; exit 1; }
  cp $R/gpurun_out/r03_pin_sensitivity.json $R/gpurun_out/r03_pin_sensitivity.md $OUT/
  python3 tools/diag/r02_ff_ab.py lanes rg > $OUT/racket_ground_lanes.log 2>&1 || { tail -20 $OUT/racket_ground_lanes.log; exit 1; }
  python3 tools/diag/r04_cadence.py > $OUT/cadence.log 2>&1 || { tail -20 $OUT/cadence.log; exit 1; }
  cp $R/gpurun_out/r04_cadence.json $OUT/cadence.json
  python3 tools/diag/r04_seal_ab.py > $OUT/seal_ab.log 2>&1 || { tail -20 $OUT/seal_ab.log; exit 1; }
  cp $R/gpurun_out/r04_seal_ab.json $OUT/seal_ab.json
  python3 tools/diag/r04_policy_stamps.py 2>&1 | grep -v amdgpu.ids > $OUT/policy_stamps.txt || exit 1
  python3 tools/diag/r03_policy_census.py 2>&1 | grep -v amdgpu.ids > $OUT/policy_census.txt || exit 1
  python3 tools/diag/r04_policy_ablate.py 2>&1 | grep -v amdgpu.ids > $OUT/policy_ablate.txt || exit 1
  grep -v amdgpu.ids $OUT/ppo_probe.log $OUT/collect_breakdown.log | cut -c1-600
fi

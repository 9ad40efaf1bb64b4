#!/bin/bash
# round-3 second GPU pass: the policy tests with the 16-env tower slices, PPO collect rate, the rollout kernel under rocprof,
# and start-to-start intervals of the step kernels by position in the episode: headline vs racket<->ball contact off
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03b
mkdir -p $OUT
export TMPDIR=/tmp
cd $R || exit 1
python3 -m pytest tests/test_gpu_policy.py tests/test_ppo.py tests/test_gpu_ppo_dist.py -m gpu -x -q > $OUT/pytest_policy.log 2>&1 || { tail -40 $OUT/pytest_policy.log; exit 1; }
tail -3 $OUT/pytest_policy.log
python3 tools/diag/r03_ppo_probe.py > $OUT/ppo_probe.log 2>&1 || { tail -20 $OUT/ppo_probe.log; exit 1; }
cat $OUT/ppo_probe.log
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o ppo_fused -- python3 $R/train_swing.py --total-timesteps 4e6 --save /tmp/ppo_%s.pt > $OUT/prof_ppo_fused.log 2>&1 || exit 1
rm -f $OUT/prof/ppo_fused_kernel_trace.csv
for v in "" "--contact-off"; do
  tag=trace_swing4096${v:+_contact_off}
  rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o $tag -- python3 $R/bench.py $v --no-cpu-baseline --no-sweep --settle-seconds 0.3 --min-timed-ms 20 > $OUT/$tag.log 2>&1 || exit 1
  python3 $R/tools/trace_hist.py $OUT/prof/${tag}_kernel_trace.csv tb_step_kernel > $OUT/$tag.hist.txt 2>&1
  python3 $R/tools/trace_hist.py $OUT/prof/${tag}_kernel_trace.csv tb_ff_kernel >> $OUT/$tag.hist.txt 2>&1
  rm -f $OUT/prof/${tag}_kernel_trace.csv
  cat $OUT/$tag.hist.txt
done
ls $OUT $OUT/prof

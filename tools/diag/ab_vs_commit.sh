#!/bin/bash
# Same-box comparison of the kernels of an EARLIER COMMIT with the working tree's, workload by workload (tools/diag/r03_flag_ab.py's
# workload names), each library in a process of its own, alternated twice.
#   here (has .git):   bash tools/diag/ab_vs_commit.sh export <commit> <tag>      -> ab/<tag>/ (git-ignored, travels with gpurun)
#   on the GPU box:    bash tools/diag/ab_vs_commit.sh run <tag> swing4096 tennis4096 swing1m ...
# Round 4's check against round 3 (6d9d2c3): large batches equal (Tennisbot 1 M 20.7-20.8 G, 4 M 19.3 G, SwingRacket 1 M 12.1 G both),
# Tennisbot 4096 envs 781-783 -> 809-810 M, SwingRacket 4096 envs 1114-1120 -> 1101-1105 M (a scheduling hint was deleted on purpose).
R=${GRAFT_REPO_ROOT:-/root/repo}
if [ "$1" = "export" ]; then
  rm -rf $R/ab/$3 && mkdir -p $R/ab/$3 && git -C $R archive $2 tennisbot_rl_amd/csrc include | tar -x -C $R/ab/$3 && echo "exported $2 to ab/$3"
  exit $?
fi
TAG=$2; shift 2
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp -fPIC -shared"
/opt/rocm/bin/hipcc $F -o /tmp/libtb_$TAG.so $R/ab/$TAG/tennisbot_rl_amd/csrc/tb_stepper.hip || exit 1
for w in "$@"; do
  for lib in /tmp/libtb_$TAG.so $R/tennisbot_rl_amd/libtb_stepper.so /tmp/libtb_$TAG.so $R/tennisbot_rl_amd/libtb_stepper.so; do
    echo -n "$w $(basename $lib) "; python3 $R/tools/diag/r03_flag_ab.py --child $lib $w 2>/dev/null | tail -1 || exit 1
  done
done

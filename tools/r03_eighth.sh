#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03i
mkdir -p $OUT
export TMPDIR=/tmp
cd $R || exit 1
python3 tools/diag/r02_ff_ab.py small rg > $OUT/ff_ab_small_rg.log 2>&1; cat $OUT/ff_ab_small_rg.log | grep -v amdgpu.ids
python3 tools/diag/r02_ff_ab.py lanes rg > $OUT/ff_ab_lanes_rg.log 2>&1; cat $OUT/ff_ab_lanes_rg.log | grep -v amdgpu.ids
for k in 1 2; do
  python3 ab/r02/bench.py --no-cpu-baseline --no-sweep --envs-per-gpu 1048576 --rollout-steps 104 --steps 104 > $OUT/ab1m_r02_$k.json 2>> $OUT/ab.err || exit 1
  python3 bench.py --no-cpu-baseline --no-sweep --envs-per-gpu 1048576 --rollout-steps 104 --steps 104 > $OUT/ab1m_new_$k.json 2>> $OUT/ab.err || exit 1
done
python3 - <<'PY'
import json, glob, os
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out", "r03i")
for tag in ("r02", "new"):
    print("1 M envs", tag, [round(json.load(open(f))["value"] / 1e9, 2) for f in sorted(glob.glob(os.path.join(out, "ab1m_%s_*.json" % tag)))], "G env steps/s")
PY
cd /tmp
B="--envs-per-gpu 1048576 --rollout-steps 104 --steps 104 --warmup 26 --no-cpu-baseline --no-sweep"
for C in "VALUBusy" "VALUUtilization" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU"; do
  N=$(echo $C | tr ' ' '_')
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc -o sq_$N -- python3 $R/bench.py $B > $OUT/sq_$N.log 2>&1 || echo "pass $N failed"
done
python3 $R/tools/summarize_pmc_sq.py $OUT/pmc > $OUT/sq_summary.txt 2>&1; cat $OUT/sq_summary.txt
